/*
 * tile_launch_f32h.hip -- instantiates and launches the fp32 "f16x3" split tile kernels (lssvm_tile_f32_split.hip.hpp: two f16 planes,
 * three plane products on v_mfma_f32_16x16x32_f16) and their set-up kernels.  A translation unit of its own so that it builds beside the
 * bf16x6 instantiations.  Compiled for gfx950 only.
 */
#include "tile_launch.hip.hpp"

#include "lssvm_tile_f32_split.hip.hpp"

/* compiled as TWO translation units (LSSVM_TU_HALF 1: *_sym.hip, the symmetric instantiations; 2: *_full.hip, the full-square ones, the set-up kernels of
 * the planes and the entry point) so that the build spreads over more cores */
#ifndef LSSVM_TU_HALF
#error "compile the _sym / _full wrapper of this file"
#endif

namespace lssvm {

#if LSSVM_TU_HALF != 1  // (the set-up kernels of the planes: in the full-square half only)
/* "f16x3" planes of y = scale * x (scale = 2^k, exact).  One wave per row.
 *   shift = 0 (linear, polynomial):  hi = f16(y), mid = f16(y - hi); planes [2][rows][ldx16] = (hi, mid).
 *   shift = s > 0 (rbf):             P0 = f16(2^-s y), P1 = f16(2^s (y - 2^s P0)), P2 = 2^(2s) P0 (exact); planes [3][rows][ldx16] = (P0, P1, P2).
 *     hi = 2^s P0 = 2^-s P2 and mid = 2^-s P1 are defined by the STORED values, so the pair is consistent whatever P0 loses to f16's subnormals;
 *     |y| must stay below 2^(16 - 2s) (P2 would overflow: reported as a NaN statistic).
 * Besides the planes the kernel leaves what the set-up needs to decide whether two f16 planes represent this data as well as fp32 does:
 * stats[0] = max over the rows of |rest|^2 / |y|^2 (rest = y - hi - mid: relative representation error of a row, squared), stats[1] = max |rest|^2,
 * stats[2] = max |y|^2 -- as float bit patterns combined with atomicMax (non-negative floats order like their bit patterns; a NaN stays on top).
 * X: [rows][ldx] fp32, features in natural order; planes zero padded. */
/* row_inv != NULL (round 6, linear kernel): every ROW gets a power-of-two scale of its own -- 2^k_i moves the row's largest entry to [2^14, 2^15) -- and row_inv[row]
 * receives 2^-k_i: K = D (Xs Xs^T) D with Xs the scaled rows and D = diag(2^-k_i), so the caller multiplies the vector by D in front of the product and the result by D
 * behind it (Problem<float>: row_scaled_).  One scale for the whole matrix loses the small ROWS of data whose points differ by orders of magnitude. */
__global__ void k_split_f16x2(const float *__restrict__ X, int ldx, int dfeat, size_t rows, int ldx16, float scale, int shift, uint16_t *__restrict__ planes,
                              size_t plane_stride, unsigned *__restrict__ stats, float *__restrict__ row_inv) {
    const size_t row = static_cast<size_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    if (row_inv != nullptr) {
        float m = 0.0f;
        for (int f = lane; f < dfeat; f += 64) m = fmaxf(m, fabsf(X[row * ldx + f]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        int k = 0;
        if (m > 0.0f && m <= 3.0e38f) k = min(max(14 - ilogbf(m), -40), 40);  // (F16_TARGET_EXP, F16_MAX_SHIFT of lssvm_problem.hip)
        scale = ldexpf(1.0f, k);
        if (lane == 0) row_inv[row] = ldexpf(1.0f, -k);
    }
    const float up = __builtin_ldexpf(1.0f, shift), down = __builtin_ldexpf(1.0f, -shift);
    float sr = 0.0f, sx = 0.0f;
    bool overflow = false;
    for (int f = lane; f < ldx16; f += 64) {
        const float y = f < dfeat ? X[row * ldx + f] * scale : 0.0f;
        const _Float16 p0 = static_cast<_Float16>(y * down);
        const float hi = static_cast<float>(p0) * up;
        const float r1 = y - hi;
        const _Float16 p1 = static_cast<_Float16>(r1 * up);
        const float r2 = r1 - static_cast<float>(p1) * down;
        planes[row * ldx16 + f] = __builtin_bit_cast(uint16_t, p0);
        planes[plane_stride + row * ldx16 + f] = __builtin_bit_cast(uint16_t, p1);
        if (shift != 0) {
            const float p2 = hi * up;
            overflow = overflow || !(fabsf(p2) <= 65504.0f);
            planes[2 * plane_stride + row * ldx16 + f] = __builtin_bit_cast(uint16_t, static_cast<_Float16>(p2));
        }
        sr = fmaf(r2, r2, sr);
        sx = fmaf(y, y, sx);
    }
    if (overflow) sr = __builtin_nanf("");
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sr += __shfl_xor(sr, off);
        sx += __shfl_xor(sx, off);
    }
    if (lane == 0 && (sx > 0.0f || sx != sx || sr != sr)) {
        // (a million rows hammering three addresses serialise: 34 ms at 1 000 000 x 128.  The maxima only grow, so a row that does not exceed
        // what it reads -- possibly a moment old, never too large -- has nothing to add)
        const unsigned v0 = __builtin_bit_cast(unsigned, sr / sx), v1 = __builtin_bit_cast(unsigned, sr), v2 = __builtin_bit_cast(unsigned, sx);
        if (v0 > __hip_atomic_load(stats + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(stats + 0, v0);
        if (v1 > __hip_atomic_load(stats + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(stats + 1, v1);
        if (v2 > __hip_atomic_load(stats + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(stats + 2, v2);
    }
}

/* GRID planes of the centred, scaled rbf data y (round 5, KT_RBFG; lssvm_tile_f32_split.hip.hpp): h = g rint(y / g), s = y - h (exact), planes [3][rows][ldx16] =
 * (f16(sigma h) -- exact: |h / g| <= 2048 --, s1 = f16(sigma s), s2 = f16(sigma s - s1)); per row chg = sigma^2 ch with ch = -|h|^2 / 2 (a multiple of g^2 / 2: exact in
 * fp32 while (R2 + 160) / (g^2 / 2) <= 2^24, the host's choice of g) and E = 2^(c - ch), c = -|y|^2 / 2 in double: the part of the exponent the grid norms leave out.
 * stats[0] becomes a NaN pattern if a plane overflows f16 (the host then falls back).  One wave per row. */
__global__ void k_split_grid_f16(const float *__restrict__ X, int ldx, int dfeat, size_t rows, int ldx16, float g, float sigma, uint16_t *__restrict__ planes, size_t plane_stride,
                                 float *__restrict__ chg, float *__restrict__ efac, unsigned *__restrict__ stats) {
    const size_t row = static_cast<size_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float inv_g = 1.0f / g;  // (a power of two)
    double sh = 0.0, sy = 0.0;
    bool overflow = false;
    for (int f = lane; f < ldx16; f += 64) {
        const float y = f < dfeat ? X[row * ldx + f] : 0.0f;
        const float h = rintf(y * inv_g) * g;
        const float sres = (y - h) * sigma;
        const float hs = h * sigma;
        const _Float16 p0 = static_cast<_Float16>(hs);
        const _Float16 p1 = static_cast<_Float16>(sres);
        const _Float16 p2 = static_cast<_Float16>(sres - static_cast<float>(p1));
        overflow = overflow || !(fabsf(hs) <= 65504.0f) || static_cast<float>(p0) != hs;
        planes[row * ldx16 + f] = __builtin_bit_cast(uint16_t, p0);
        planes[plane_stride + row * ldx16 + f] = __builtin_bit_cast(uint16_t, p1);
        planes[2 * plane_stride + row * ldx16 + f] = __builtin_bit_cast(uint16_t, p2);
        sh = fma(static_cast<double>(h), static_cast<double>(h), sh);
        sy = fma(static_cast<double>(y), static_cast<double>(y), sy);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sh += __shfl_xor(sh, off);
        sy += __shfl_xor(sy, off);
    }
    if (lane == 0) {
        const double ch = -0.5 * sh, c = -0.5 * sy;
        const float chs = static_cast<float>(static_cast<double>(sigma) * static_cast<double>(sigma) * ch);
        chg[row] = chs;
        efac[row] = static_cast<float>(exp2(c - ch));
        if (static_cast<double>(chs) != static_cast<double>(sigma) * static_cast<double>(sigma) * ch) overflow = true;  // (the start value must be exact)
    }
    if (__any(overflow) && lane == 0) atomicMax(stats + 0, 0x7FC00000u);
}

/* max |x| over the valid entries of X (as a float bit pattern, atomicMax) */
__global__ void k_absmax(const float *__restrict__ X, int ldx, int dfeat, size_t rows, unsigned *__restrict__ out) {
    const size_t total = rows * static_cast<size_t>(ldx);
    float m = 0.0f;
    for (size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total; idx += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const int f = static_cast<int>(idx % ldx);
        if (f < dfeat) {
            const float v = fabsf(X[idx]);
            m = (v > m || v != v) ? v : m;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(m, off);
        m = (o > m || o != o) ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, __builtin_bit_cast(unsigned, m));
}
#endif


template <int KT, bool SYM>
static void launch_f3_kt(const TileArgs<float> &a, dim3 grid, hipStream_t s) {
    const dim3 block(TILE_THREADS);
    // ONE kernel per (kernel function, feature count, variant) -- the instantiations no default path reaches were retired in round 4:
    //   <= 128 features: the hand-scheduled groups (tile_matvec_f32_f3h); the run-time integer power, whose epilogue does not fit their capped
    //                    register budget without spills, stays on the compiler-scheduled groups (tile_matvec_f32_f3w) at every width;
    //   beyond:          tile_matvec_f32_f3w -- except the linear kernel in the symmetric variant, which runs one pass of the 128-feature kernels
    //                    per feature panel (Problem<float>: wide_linear_) and never comes here with more.
#define LSSVM_F3_CASE(N)                                                                                  \
    case N:                                                                                               \
        if constexpr (N > f16_max_nk64(KT)) {                                                             \
            throw Error(LSSVM_ERR_INTERNAL, "no f16x3 rbf tile kernel for this number of features");       \
        } else if constexpr (KT != KT_POLY && N <= F16_HAND_MAX_NK64) {                                   \
            ensure_dynamic_lds(tile_matvec_f32_f3h<KT, N, SYM>, V2_LDS_BYTES);                            \
            hipLaunchKernelGGL((tile_matvec_f32_f3h<KT, N, SYM>), grid, block, V2_LDS_BYTES, s, a);       \
        } else if constexpr (KT == KT_LINEAR && SYM) {                                                    \
            throw Error(LSSVM_ERR_INTERNAL, "the symmetric linear kernel takes feature-panel passes beyond 128 features"); \
        } else {                                                                                          \
            ensure_dynamic_lds(tile_matvec_f32_f3w<KT, N, SYM>, V2_LDS_BYTES);                            \
            hipLaunchKernelGGL((tile_matvec_f32_f3w<KT, N, SYM>), grid, block, V2_LDS_BYTES, s, a);       \
        }                                                                                                 \
        break;
    switch (a.nk64) {
#ifdef LSSVM_DEV_SUBSET  // development builds (make DEV=1): 128 and 256 features only, a quarter of the compile time
        LSSVM_F3_CASE(2) LSSVM_F3_CASE(4)
#else
        LSSVM_F3_CASE(1) LSSVM_F3_CASE(2) LSSVM_F3_CASE(3) LSSVM_F3_CASE(4) LSSVM_F3_CASE(5) LSSVM_F3_CASE(6) LSSVM_F3_CASE(7) LSSVM_F3_CASE(8)
#endif
        default: throw Error(LSSVM_ERR_INTERNAL, "no f16x3 tile kernel for this number of features");
    }
#undef LSSVM_F3_CASE
}

template <bool SYM>
static void launch_f3(const TileArgs<float> &a, int kernel_type, dim3 grid, hipStream_t s) {
    switch (kernel_type) {
        case KT_LINEAR: launch_f3_kt<KT_LINEAR, SYM>(a, grid, s); break;
        case KT_POLY:
            if (a.degree == 3) {
                launch_f3_kt<KT_POLY3, SYM>(a, grid, s);
            } else if (a.degree == 2) {
                launch_f3_kt<KT_POLY2, SYM>(a, grid, s);
            } else {
                launch_f3_kt<KT_POLY, SYM>(a, grid, s);
            }
            break;
        default:
            if (a.rbf_grid != 0) {  // grid planes (KT_RBFG): hand-scheduled groups, at most 128 features
                const dim3 block(TILE_THREADS);
                switch (a.nk64) {
#ifndef LSSVM_DEV_SUBSET
                    case 1:
                        ensure_dynamic_lds(tile_matvec_f32_g6h<1, SYM>, V2_LDS_BYTES);
                        hipLaunchKernelGGL((tile_matvec_f32_g6h<1, SYM>), grid, block, V2_LDS_BYTES, s, a);
                        break;
#endif
                    case 2:
                        ensure_dynamic_lds(tile_matvec_f32_g6h<2, SYM>, V2_LDS_BYTES);
                        hipLaunchKernelGGL((tile_matvec_f32_g6h<2, SYM>), grid, block, V2_LDS_BYTES, s, a);
                        break;
#ifndef LSSVM_DEV_SUBSET
#define LSSVM_G6W_CASE(N)                                                                         \
    case N:                                                                                       \
        ensure_dynamic_lds(tile_matvec_f32_g6w<N, SYM>, V2_LDS_BYTES);                            \
        hipLaunchKernelGGL((tile_matvec_f32_g6w<N, SYM>), grid, block, V2_LDS_BYTES, s, a);       \
        break;
                    LSSVM_G6W_CASE(3) LSSVM_G6W_CASE(4) LSSVM_G6W_CASE(5) LSSVM_G6W_CASE(6)
#undef LSSVM_G6W_CASE
#endif
                    default: throw Error(LSSVM_ERR_INTERNAL, "no grid-plane rbf tile kernel for this number of features");
                }
            } else if (a.dc_folded != 0) {
                launch_f3_kt<KT_RBFF, SYM>(a, grid, s);
            } else {
                launch_f3_kt<KT_RBF, SYM>(a, grid, s);
            }
            break;
    }
}

void launch_f16_tile_kernel_sym(const TileArgs<float> &a, int kernel_type, hipStream_t s);  // tile_launch_f32h_sym.hip

#if LSSVM_TU_HALF == 1
void launch_f16_tile_kernel_sym(const TileArgs<float> &a, int kernel_type, hipStream_t s) {
    launch_f3<true>(a, kernel_type, dim3(static_cast<unsigned>(a.num_items)), s);
}
#else
void launch_f16_tile_kernel(const TileArgs<float> &a, int kernel_type, dim3 grid, hipStream_t s) {
    if (a.items != nullptr) {
        launch_f16_tile_kernel_sym(a, kernel_type, s);
    } else {
        launch_f3<false>(a, kernel_type, grid, s);
    }
    LSSVM_HIP_CHECK(hipGetLastError());
}

void split_f16_planes(const float *X, int ldx, int dfeat, size_t rows, int ldx16, float scale, int shift, uint16_t *planes, size_t plane_stride, unsigned *stats, hipStream_t s,
                      float *row_inv_scale) {
    hipLaunchKernelGGL(k_split_f16x2, dim3(static_cast<unsigned>((rows + 3) / 4)), dim3(256), 0, s, X, ldx, dfeat, rows, ldx16, scale, shift, planes, plane_stride, stats, row_inv_scale);
    LSSVM_HIP_CHECK(hipGetLastError());
}

void split_grid_planes(const float *X, int ldx, int dfeat, size_t rows, int ldx16, float g, float sigma, uint16_t *planes, size_t plane_stride, float *chg, float *efac, unsigned *stats,
                       hipStream_t s) {
    hipLaunchKernelGGL(k_split_grid_f16, dim3(static_cast<unsigned>((rows + 3) / 4)), dim3(256), 0, s, X, ldx, dfeat, rows, ldx16, g, sigma, planes, plane_stride, chg, efac, stats);
    LSSVM_HIP_CHECK(hipGetLastError());
}

void absmax_f32(const float *X, int ldx, int dfeat, size_t rows, unsigned *out, hipStream_t s) {
    hipLaunchKernelGGL(k_absmax, dim3(1024), dim3(256), 0, s, X, ldx, dfeat, rows, out);
    LSSVM_HIP_CHECK(hipGetLastError());
}

#endif

}  // namespace lssvm
