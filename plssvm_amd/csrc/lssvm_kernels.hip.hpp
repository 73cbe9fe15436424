/*
 * lssvm_kernels.hip.hpp -- hand-written gfx950 (CDNA4 / MI355X) device kernels of the LS-SVM CG hot path.
 *
 * What is computed (citations relative to the reference tree SC-SGS/PLSSVM):
 *   the implicit matrix-vector product  ret += add * Abar * d,  Abar_ij = k(x_i,x_j) + delta_ij/C + QA_cost - q_i - q_j
 *   (src/plssvm/backends/OpenMP/svm_kernel.cpp:33-54, include/plssvm/backends/HIP/svm_kernel.hip.hpp:38-270),
 *   the q vector (q_kernel.cpp:18-55 / HIP/q_kernel.hip.hpp:33-85) and the BLAS-1 of the CG loop (csvm.cpp:101-163).
 *
 * How (MI355X-first, NOT the reference's 16x16-thread / 6x6-register tiling):
 *   Abar * d = K d + d/C + (QA_cost*S - q.d) 1 - S q   with S = sum(d)  (rank-1 terms peeled off).
 *   Only K d is O(n^2 d).  It is evaluated as a flash-style sweep: a 256-thread workgroup (4 wave64) owns a 128-row
 *   block of the implicit matrix and walks a chunk of 128-column tiles.  Per tile the 128x128 Gram block X_I X_J^T is
 *   contracted on the matrix cores (v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64: exact IEEE fma chains in k order,
 *   at the full f32/f64 vector rate) from k-chunks staged through LDS (padded rows: conflict-free ds_read_b128 /
 *   ds_read_b64); the kernel function (pow / exp) and the multiplication with d_j are fused into the epilogue on the
 *   vector ALU, and row sums stay in registers across the whole chunk.  No symmetry trick and no atomics: every
 *   (row block, column chunk) work item writes its partial row sums to its own slab, a second kernel adds the slabs
 *   in a fixed order => bit-reproducible and independent of the number of GPUs.
 *   RBF uses |x_i - x_j|^2 = |x_i|^2 + |x_j|^2 - 2 x_i.x_j on data centred by the column means (distances are translation
 *   invariant; centring bounds the cancellation error); the accumulator is initialised with -(|x_i|^2+|x_j|^2)/2 so the
 *   MFMA chain leaves -|x_i-x_j|^2/2 and the epilogue is one mul + v_exp_f32 + fma.
 */
#pragma once

#include "lssvm_types.hpp"

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <type_traits>

/* timing-only ablations of the fp32 tile kernel (option "debug_ablate"); compiled in only with -DLSSVM_ENABLE_ABLATION so that the
 * shipped kernel carries no extra branches */
#ifdef LSSVM_ENABLE_ABLATION
#define LSSVM_DBG(a, bit) (((a).dbg & (bit)) != 0)
#else
#define LSSVM_DBG(a, bit) false
#endif

namespace lssvm {

/* integer power by repeated squaring; the reference uses pow(real, int) on the GPU (HIP/svm_kernel.hip.hpp:178) and
 * std::pow(real, real(degree)) on the CPU (kernel_function_types.hpp:86-89): equal up to rounding for integer degrees. */
template <typename T>
__device__ __forceinline__ T ipow(T base, int degree) {
    unsigned e = degree < 0 ? static_cast<unsigned>(-(long) degree) : static_cast<unsigned>(degree);
    T result = T(1);
    T b = base;
    while (e != 0u) {
        if (e & 1u) result *= b;
        b *= b;
        e >>= 1u;
    }
    return degree < 0 ? T(1) / result : result;
}

/* blockIdx.x -> (local row block, column chunk).
 * map_mode 0: consecutive blocks walk the row blocks of one column chunk.
 * map_mode 1 (XCD aware): the hardware deals consecutive workgroup ids round-robin over the 8 XCDs, each with a private
 *   4 MiB L2 (placement is a speed matter only, never correctness).  The ids that land on one XCD are grouped into 8 x 8
 *   super-tiles (8 row blocks x 8 column chunks), so that the ~64 workgroups resident on an XCD at a time re-read only 8
 *   row panels (8 x d x 128 x s bytes) and share every column tile 8 ways -- instead of 64 distinct row panels that alone
 *   overflow the L2. */
template <typename T>
__device__ __forceinline__ bool decode_work_item(const TileArgs<T> &a, int &ibl, int &jc) {
    const int id = blockIdx.x;
    if (a.map_mode == 0) {
        ibl = id % a.num_ib;
        jc = id / a.num_ib;
        return true;
    }
    const int x = id & 7;
    const int k = id >> 3;
    const int l = k & 63;
    const int S = (k >> 6) * 8 + x;  // super-tile index
    const int si = S % a.super_i;
    const int sj = S / a.super_i;
    ibl = si * 8 + (l & 7);
    jc = sj * 8 + (l >> 3);
    return ibl < a.num_ib && jc < a.num_jc;
}

/* exp(x) in double for the rbf epilogue: 2^k * p(r), k = rint(x log2 e), r = x - k ln2 (two-part ln2), p = degree-13 Taylor
 * polynomial on |r| <= 0.347 (truncation 4e-18), 19 double-precision VALU operations instead of libm's ~40 with its
 * special-case branches; v_ldexp_f64 handles underflow to 0 for very negative x.  Relative error < 2 ulp. */
__device__ __forceinline__ double fast_exp_f64(double x) {
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;
    p = fma(p, r, 2.08767569878681e-09);
    p = fma(p, r, 2.505210838544172e-08);
    p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.7557319223985893e-06);
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 1.984126984126984e-04);
    p = fma(p, r, 1.388888888888889e-03);
    p = fma(p, r, 8.333333333333333e-03);
    p = fma(p, r, 4.1666666666666664e-02);
    p = fma(p, r, 1.6666666666666666e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_ldexp(p, static_cast<int>(k));
}

/* DEG: polynomial degree class resolved OUTSIDE the per-element loop (a uniform switch around the whole epilogue):
 * 3 = cube, 2 = square, 0 = generic integer power.  Ignored for the other kernels. */
template <int KT, int DEG, typename T>
__device__ __forceinline__ T apply_kernel_function(T acc, const TileArgs<T> &a) {
    if constexpr (KT == KT_LINEAR) {
        return acc;
    } else if constexpr (KT == KT_POLY) {
        const T v = acc * a.gamma + a.coef0;  // contracted to one fma, = std::fma(gamma, dot, coef0)
        if constexpr (DEG == 3) {
            return v * v * v;
        } else if constexpr (DEG == 2) {
            return v * v;
        } else {
            return ipow(v, a.degree);
        }
    } else {
        if constexpr (std::is_same_v<T, float>) {
            // fp32: the data was pre-scaled by sqrt(2*gamma*log2(e)) at set-up, so acc = -gamma*log2(e)*|xi-xj|^2 already
            return __builtin_amdgcn_exp2f(acc);
        } else {
            return fast_exp_f64(acc * a.gamma);  // acc = -|xi-xj|^2 / 2 ; gamma field = 2*gamma
        }
    }
}

/* v^degree for the degree class DEG (3, 2, or 0 = any integer degree) */
template <int DEG, typename T>
__device__ __forceinline__ T poly_power(T v, int degree) {
    if constexpr (DEG == 3) {
        return v * v * v;
    } else if constexpr (DEG == 2) {
        return v * v;
    } else {
        return ipow(v, degree);
    }
}

/* runs `body(std::integral_constant<int, DEG>)` with the polynomial degree class of `a` (one uniform branch per tile) */
template <int KT, typename T, typename F>
__device__ __forceinline__ void with_degree_class(const TileArgs<T> &a, F &&body) {
    if constexpr (KT == KT_POLY) {
        if (a.degree == 3) {
            body(std::integral_constant<int, 3>{});
        } else if (a.degree == 2) {
            body(std::integral_constant<int, 2>{});
        } else {
            body(std::integral_constant<int, 0>{});
        }
    } else {
        body(std::integral_constant<int, 0>{});
    }
}

/* Pins a uniform pointer into an SGPR pair at this point of the program: "uniform base + 32-bit lane offset" is then selected
 * as the saddr form of global_load_lds / global_store (no 64-bit vector address arithmetic, one VGPR per lane offset), and the
 * compiler cannot re-associate the base into several vector adds. */
__device__ __forceinline__ const char *sgpr_ptr(const void *p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<unsigned>(v))));  // (the builtin returns int:
    const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<unsigned>(v >> 32))));  // no sign extension)
    unsigned long long u = (static_cast<unsigned long long>(hi) << 32) | lo;
    asm volatile("" : "+s"(u));
    return reinterpret_cast<const char *>(u);
}

/* Keeps the 32-bit -> 64-bit extension of a lane offset in the basic block of its use (instruction selection is per block:
 * a zext hoisted out of the loop hides the "SGPR base + 32-bit VGPR offset" addressing mode from it). */
__device__ __forceinline__ unsigned lane_off(unsigned v) {
    asm volatile("" : "+v"(v));
    return v;
}

/* The v2 kernels take the polynomial degree class as part of their kernel-type template parameter, so every instantiation
 * carries ONE epilogue (the three-way runtime switch of with_degree_class made the register allocator budget for the generic
 * integer-power path and spill in the cube path). */
__host__ __device__ constexpr int v2_base_kt(int kt) { return (kt == KT_POLY2 || kt == KT_POLY3) ? KT_POLY : kt; }
__host__ __device__ constexpr int v2_degree_class(int kt) { return kt == KT_POLY3 ? 3 : (kt == KT_POLY2 ? 2 : 0); }

/* =====================================================================================================================
 * fp32 tile kernel: v_mfma_f32_32x32x2_f32
 *   operand maps (cdna_hip_programming.md section 3): lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
 *   the 32x32 result has column j = l&31 on the lane and rows (reg&3) + 8*(reg>>2) + 4*(l>>5) in its 16 registers.
 *   fp32 data is stored in HBM (and hence in LDS) with the features of every aligned group of 8 in the order
 *   k = 0,2,4,6,1,3,5,7 (k_interleave_features, applied once at set-up), so that ONE 16-byte read of lane-half h returns
 *   k = h, 2+h, 4+h, 6+h -- the operands of four consecutive MFMAs -- and the contraction runs through k in ascending
 *   order (bit-identical to the fma chain of the reference's dot product, include/plssvm/detail/operators.hpp:117-126).
 *   LDS image of a k-chunk (this kernel): [128 rows][32 floats + 4 pad].
 * ===================================================================================================================== */

template <int KT>
__global__ __launch_bounds__(TILE_THREADS, 2) void tile_matvec_f32(const TileArgs<float> a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *As = reinterpret_cast<float *>(smem_raw);  // [2][TILE * F32_LS]
    float *Bs = As + 2 * TILE * F32_LS;               // [2][TILE * F32_LS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1;  // wave row (0..1): rows wr*64 .. +63 of the tile
    const int wc = wave & 1;   // wave column
    const int r = lane & 31;
    const int h = lane >> 5;

    // work item -> (row block, column chunk); consecutive blocks share the column chunk (L2 reuse on every XCD)
    int ibl, jc;
    if (!decode_work_item(a, ibl, jc)) return;
    const int row0 = (a.ib_begin + ibl) * TILE;
    const int jt_begin = jc * a.jc_tiles;
    const int jt_end = min(jt_begin + a.jc_tiles, a.num_jt);
    const int ntiles = jt_end - jt_begin;
    if (ntiles <= 0) return;

    // staging: thread -> (row = tid/4 [+64], 8 consecutive floats at (tid%4)*8); 4 threads cover one 128-B line
    const int srow = tid >> 2;
    const int sseg = tid & 3;
    const float *Ag = a.Xr + static_cast<size_t>(row0 + srow) * a.ldx + sseg * 8;
    const size_t rstep = static_cast<size_t>(64) * a.ldx;
    const int lds_w = srow * F32_LS + sseg * 8;  // float offset of this thread's 8 floats in the LDS image

    float rowpart[2][16];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) rowpart[rb][i] = 0.0f;

    // rbf: c_i = -|x_i|^2/2 of the tile's 128 rows lives in LDS (re-read by every tile_init as 4-row float4 broadcasts);
    // keeping the lane's 32 values in registers instead pushes the kernel over the 256-VGPR budget of 2 waves per SIMD
    float *cis = Bs + 2 * TILE * F32_LS;  // [TILE]
    if constexpr (KT == KT_RBF) {
        if (tid < TILE) cis[tid] = a.cr[row0 + tid];
    }

    f32x16 acc[2][2];
    f32x4 sa[2][2], sb[2][2];

    auto stage_load = [&](int jt, int kc) {
        const float *Bg = a.Xc + static_cast<size_t>(jt * TILE + srow) * a.ldx + sseg * 8;
        const int ko = kc * F32_KC;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            sa[p][0] = *reinterpret_cast<const f32x4 *>(Ag + p * rstep + ko);
            sa[p][1] = *reinterpret_cast<const f32x4 *>(Ag + p * rstep + ko + 4);
            sb[p][0] = *reinterpret_cast<const f32x4 *>(Bg + p * rstep + ko);
            sb[p][1] = *reinterpret_cast<const f32x4 *>(Bg + p * rstep + ko + 4);
        }
    };
    auto stage_store = [&](int buf) {
        float *Aw = As + buf * TILE * F32_LS + lds_w;
        float *Bw = Bs + buf * TILE * F32_LS + lds_w;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            // the k-interleave (even k first, odd k second, see header comment) is already part of the HBM layout (k_interleave_features)
            *reinterpret_cast<f32x4 *>(Aw + p * 64 * F32_LS) = sa[p][0];
            *reinterpret_cast<f32x4 *>(Aw + p * 64 * F32_LS + 4) = sa[p][1];
            *reinterpret_cast<f32x4 *>(Bw + p * 64 * F32_LS) = sb[p][0];
            *reinterpret_cast<f32x4 *>(Bw + p * 64 * F32_LS + 4) = sb[p][1];
        }
    };

    // per-lane column data of a tile: d_j and (rbf) c_j = -|x_j|^2/2.  They are fetched ONE TILE AHEAD (col_prefetch at the
    // first k-chunk of the running tile, consumed by tile_init at its end) so their global-load latency is never exposed.
    float dj[2], cj[2], djn[2], cjn[2];
    bool padcol[2] = { false, false };  // polynomial with a negative degree only: (0*gamma+coef0)^degree may be inf on padding
    auto col_prefetch = [&](int jt) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int j = jt * TILE + wc * 64 + cb * 32 + r;
            djn[cb] = a.dvec[j];
            if constexpr (KT == KT_RBF) cjn[cb] = a.cc[j];
        }
    };
    auto tile_init = [&](int jt) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            dj[cb] = djn[cb];
            if constexpr (KT == KT_RBF) cj[cb] = cjn[cb];
            if constexpr (KT == KT_POLY) padcol[cb] = (a.degree < 0) && (jt * TILE + wc * 64 + cb * 32 + r >= a.ncols_valid);
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if constexpr (KT != KT_RBF) acc[rb][cb][i] = 0.0f;
                }
        if constexpr (KT == KT_RBF) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 civ = *reinterpret_cast<const f32x4 *>(cis + wr * 64 + rb * 32 + 8 * g4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[rb][0][4 * g4 + e] = civ[e] + cj[0];
                        acc[rb][1][4 * g4 + e] = civ[e] + cj[1];
                    }
                }
        }
    };

    const int nsteps = ntiles * a.kchunks;
    stage_load(jt_begin, 0);
    col_prefetch(jt_begin);
    stage_store(0);
    __syncthreads();  // also publishes cis
    tile_init(jt_begin);

    int jt = jt_begin;
    int kc = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        int njt = jt, nkc = kc + 1;
        if (nkc == a.kchunks) {
            nkc = 0;
            ++njt;
        }
        const bool has_next = (s + 1 < nsteps);
        const bool do_stage = has_next && !LSSVM_DBG(a, 1);
        if (do_stage) stage_load(njt, nkc);
        if (kc == 0 && jt + 1 < jt_end) col_prefetch(jt + 1);

        {
            const float *Ab = As + cur * TILE * F32_LS + (wr * 64 + r) * F32_LS + h * 4;
            const float *Bb = Bs + cur * TILE * F32_LS + (wc * 64 + r) * F32_LS + h * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a0 = *reinterpret_cast<const f32x4 *>(Ab + g * 8);
                const f32x4 a1 = *reinterpret_cast<const f32x4 *>(Ab + 32 * F32_LS + g * 8);
                const f32x4 b0 = *reinterpret_cast<const f32x4 *>(Bb + g * 8);
                const f32x4 b1 = *reinterpret_cast<const f32x4 *>(Bb + 32 * F32_LS + g * 8);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc[1][1], 0, 0, 0);
                }
            }
        }

        if (do_stage) stage_store(cur ^ 1);

        if (kc == a.kchunks - 1 && !LSSVM_DBG(a, 4)) {
            // epilogue of tile jt: K_ij = f(acc), row partial += K_ij * d_j  (vector ALU, fused; nothing is written)
            with_degree_class<KT>(a, [&](auto degc) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            float kv = LSSVM_DBG(a, 2) ? acc[rb][cb][i] : apply_kernel_function<KT, decltype(degc)::value>(acc[rb][cb][i], a);
                            if constexpr (KT == KT_POLY) {
                                if (padcol[cb]) kv = 0.0f;  // d_j is an exact zero there, but inf * 0 would be nan
                            }
                            rowpart[rb][i] = fmaf(kv, dj[cb], rowpart[rb][i]);
                        }
            });
            if (has_next) tile_init(njt);
        }
        if (!LSSVM_DBG(a, 8)) __syncthreads();
        jt = njt;
        kc = nkc;
    }

    // reduce the row partials over the 32 lanes that share the rows (same lane-half), then over the two wave columns
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = rowpart[rb][i];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            rowpart[rb][i] = v;
        }
    float *red = reinterpret_cast<float *>(smem_raw);  // [2][TILE]; the staging buffers are dead (barrier above)
    if (r == 0) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[wc * TILE + wr * 64 + rb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h] = rowpart[rb][i];
    }
    __syncthreads();
    if (tid < TILE) {
        a.partial[static_cast<size_t>(jc) * a.part_stride + ibl * TILE + tid] = red[tid] + red[TILE + tid];
    }
}

/* =====================================================================================================================
 * fp32 tile kernel, version 2 ("resident row panel"): for num_features <= 256.
 *   - the work item's 128-row panel of X stays in REGISTERS for the whole sweep (flash-style): wave w owns rows 32w..32w+31
 *     as MFMA A fragments (16 VGPRs per 32 features), so the panel is read from L2 once per work item instead of once
 *     per column tile, and only the column side is staged on chip;
 *   - column k-chunks (128 rows x 32 features = 16 KiB) travel HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no
 *     staging registers, no ds_write), into a 4-slot ring, two chunks ahead; the hand-over of chunk s+1 (counted
 *     s_waitcnt vmcnt(4) + raw s_barrier, cdna_hip_programming.md section 5 "Pipelining across barriers") is executed in
 *     the MIDDLE of step s, in the shadow of its MFMAs, so a step starts reading its chunk with no wait at its head;
 *   - the LDS image is lane-linear (128-byte rows); bank conflicts are avoided by XOR-swizzling the 16-byte slot with
 *     (row >> 1) & 7 on the SOURCE address of the DMA and on the read address (rule 21 of the guide);
 *   - d_j and c_j of a tile arrive the same way from a packed [tile][256] array (k_pack_dc), so no ordinary global load
 *     (whose use would drain the DMA queue) sits inside the loop.
 * Each wave multiplies its 32 rows with all 128 columns of the tile: 4 accumulators of 32x32, 64 MFMAs + 16 ds_read_b128
 * per chunk.
 * ===================================================================================================================== */
constexpr int V2_RING = 4;                       // chunk slots in LDS
constexpr int V2_SLOT_BYTES = TILE * 32 * 4;     // 16 KiB
constexpr int V2_DC_SLOTS = 4;                   // ring of per-tile (d_j | c_j) records, 1 KiB each
constexpr size_t V2_LDS_BYTES = static_cast<size_t>(V2_RING) * V2_SLOT_BYTES + V2_DC_SLOTS * 1024 + (2 * TILE + 2 * 4 * TILE) * sizeof(float);  // ring + records + cis, dis, colred

using lds_ptr_t = __attribute__((address_space(3))) void *;
using gbl_ptr_t = const __attribute__((address_space(1))) void *;

template <int KT, int NKC, bool SYM>
__global__ __launch_bounds__(TILE_THREADS, (NKC <= 4 ? 2 : 1)) void tile_matvec_f32_v2(const TileArgs<float> a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char *ring = smem_raw;                                                          // [V2_RING][128 rows][128 B]
    char *dcs = smem_raw + V2_RING * V2_SLOT_BYTES;                                 // [V2_DC_SLOTS][256 floats]
    float *cis = reinterpret_cast<float *>(dcs + V2_DC_SLOTS * 1024);               // [128] c_i of the row panel (rbf)
    float *dis = cis + TILE;                                                        // [128] d_i of the row panel (SYM)
    float *colred = dis + TILE;                                                     // [2][4 waves][128] column sums of a tile (SYM)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;

    // SYM: the kernel matrix is symmetric, so only the tiles on or below the diagonal are evaluated (as the reference does,
    // svm_kernel.cpp:39); an off-diagonal tile K_IJ contributes K_IJ d_J to the rows of I AND K_IJ^T d_I to the rows of J.
    // Work items come from a host-built list of the non-empty (row block, column chunk) pairs.
    int ibl, jc;
    if constexpr (SYM) {
        const int2 it = a.items[blockIdx.x];
        ibl = __builtin_amdgcn_readfirstlane(it.x);  // uniform, but loaded through the vector memory path: move to SGPRs so
        jc = __builtin_amdgcn_readfirstlane(it.y);   // that everything derived from it is scalar arithmetic
    } else {
        if (!decode_work_item(a, ibl, jc)) return;
    }
    const int ib = a.ib_begin + ibl;
    const int row0 = ib * TILE;
    const int jt_begin = jc * a.jc_tiles;
    const int jt_end = SYM ? min(jt_begin + a.jc_tiles, ib + 1) : min(jt_begin + a.jc_tiles, a.num_jt);
    const int ntiles = jt_end - jt_begin;
    if (ntiles <= 0) return;
    const int nsteps = ntiles * NKC;
    // record index of (ib, jt) in the packed strictly-lower-triangular column slab of this device
    const long rec0 = SYM ? (static_cast<long>(ib) * (ib - 1) / 2 - a.pair_origin) : 0;

    // ---- the row panel: A fragments of this wave's 32 rows, all features (HBM layout is k-interleaved) ----
    f32x4 afrag[4 * NKC];
    {
        const float *xr = a.Xr + static_cast<size_t>(row0 + wave * 32 + r) * a.ldx + 4 * h;
#pragma unroll
        for (int m = 0; m < 4 * NKC; ++m) afrag[m] = *reinterpret_cast<const f32x4 *>(xr + 8 * m);
    }
    if constexpr (KT == KT_RBF) {
        if (tid < TILE) cis[tid] = a.cr[row0 + tid];
    }
    if constexpr (SYM) {
        if (tid < TILE) dis[tid] = a.dvec[row0 + tid];
    }
    // make the compiler retire these ordinary loads HERE, before any LDS-DMA is in flight
#pragma unroll
    for (int m = 0; m < 4 * NKC; ++m) asm volatile("" : "+v"(afrag[m]));

    // ---- LDS-DMA addressing ----
    // instruction q = 4*wave + i moves rows 8q .. 8q+7 of a chunk; lane L -> row 8q + L/8, physical 16-B slot L%8, which
    // holds logical slot (L%8) ^ ((row >> 1) & 7)
    // The source address of a DMA is (uniform 64-bit base in SGPRs) + (32-bit per-lane byte offset): the saddr form of
    // global_load_lds, so a piece costs no 64-bit vector address arithmetic and one VGPR
    unsigned dma_off[4];  // byte offset of this lane's 16 bytes inside a (tile, chunk) = 4 * (row * ldx + 4 * logical_slot)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * (4 * wave + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        dma_off[i] = 4u * static_cast<unsigned>(row * a.ldx + 4 * c);
    }
    auto issue_chunk = [&](int step) {  // step = linear (tile, chunk) index of this work item
        if (LSSVM_DBG(a, 16) && step > 3) return;  // ablation: no DMA after the prologue
        const int t = LSSVM_DBG(a, 1) ? 0 : step / NKC;  // ablation bit 1: always the same (L2-resident) tile
        const int kc = LSSVM_DBG(a, 1) ? 0 : step - t * NKC;
        const char *base = sgpr_ptr(a.Xc + static_cast<size_t>(jt_begin + t) * TILE * a.ldx + kc * 32);
        char *slot = ring + (step % V2_RING) * V2_SLOT_BYTES + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t) (base + lane_off(dma_off[i])), (lds_ptr_t) (slot + i * 1024), 16, 0, 0);
        }
    };
    // one of the four DMA instructions of a chunk (steady state: spread over the MFMA groups that follow the hand-over, an
    // LDS-DMA issue costs the wave ~60-100 cycles, MI355X_MICROARCH.md "LDS-DMA piece issue cost")
    auto issue_chunk_part = [&](int step, int i) {
        const int t = step / NKC;
        const int kc = step - t * NKC;
        const char *base = sgpr_ptr(a.Xc + static_cast<size_t>(jt_begin + t) * TILE * a.ldx + kc * 32);
        char *slot = ring + (step % V2_RING) * V2_SLOT_BYTES + wave * 4096;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t) (base + lane_off(dma_off[i])), (lds_ptr_t) (slot + i * 1024), 16, 0, 0);
    };
    auto issue_dc = [&](int t) {  // (d_j | c_j) of tile jt_begin + t: 1 KiB, each wave moves a quarter with 16 lanes
        if (lane < 16) {
            const char *src = sgpr_ptr(a.dc + static_cast<size_t>(jt_begin + t) * 256) + __builtin_amdgcn_readfirstlane(wave * 256);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t) (src + 16u * (lane_off(threadIdx.x) & 15u)), (lds_ptr_t) (dcs + (t % V2_DC_SLOTS) * 1024 + wave * 256), 16, 0, 0);
        }
    };

    // ---- read addressing: lane (r, h) reads 16-B logical slot 2*mm + h of row cb*32 + r (swizzle depends on r only) ----
    int rd_off[4];
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) rd_off[mm] = r * 128 + (((2 * mm + h) ^ ((r >> 1) & 7)) << 4);

    float rowpart[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) rowpart[i] = 0.0f;
    f32x16 acc[4];
    float dj[4], cj[4];
    bool padcol[4] = { false, false, false, false };

    // ---- prologue: chunks 0, 1, 2 (each preceded by the record of the tile that starts with it) ----
    issue_dc(0);
    issue_chunk(0);
#pragma unroll
    for (int pre = 1; pre <= 2; ++pre) {
        if (pre < nsteps) {
            if (pre % NKC == 0) issue_dc(pre / NKC);
            issue_chunk(pre);
        }
    }
    // chunk 0 (and record 0, and cis) complete: everything but the DMA instructions of the younger chunks is done
    if (nsteps >= 3) {
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    } else if (nsteps == 2) {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    f32x4 bcur[4];  // B fragments of the group about to be multiplied (double buffered against bnext in the loop)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bcur[cb] = *reinterpret_cast<const f32x4 *>(ring + cb * 4096 + rd_off[0]);

    // ---- hand-over of the NEXT chunk, executed in the MIDDLE of a step (in the shadow of that step's MFMAs) ----
    // Called half-way through step `step`: this wave's DMA of chunk step+1 (issued 2 steps ago) is complete once all but its
    // 4 youngest DMA instructions (chunk step+2) are done; the barrier makes every wave's part visible, so the next step
    // starts reading at once, with no wait and no barrier at its head.  Ring of 4 slots: the DMA issued here (chunk step+3)
    // overwrites the slot of chunk step-1, which every wave finished reading before it arrived at this barrier.
    // CHECKED = false: steady state, step + 3 < nsteps is known, the code is branch free (one basic block per tile, so the
    // compiler can place the scalar address arithmetic and the DMA issue in the shadow of the MFMAs); CHECKED = true: the
    // last tiles of the work item.
    auto handover = [&](int step, int kc_plus3_mod, auto checked) {
        constexpr bool CHECKED = decltype(checked)::value;
        if constexpr (!CHECKED) {
            if (!LSSVM_DBG(a, 16)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (!LSSVM_DBG(a, 8)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // the record of a tile is issued right BEFORE the first chunk of that tile: "chunk landed" implies "record landed"
            if (kc_plus3_mod == 0) issue_dc((step + 3) / NKC);
            // the four DMA instructions of chunk step+3 follow one by one between the MFMAs of this step's second half
        } else {
            if (step + 1 < nsteps) {
                if (step + 2 < nsteps) {
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (step + 3 < nsteps) {
                    if (kc_plus3_mod == 0) issue_dc((step + 3) / NKC);
                    issue_chunk(step + 3);
                }
            }
        }
    };

    // SYM: the four waves' column sums of tile t (written to colred by its epilogue, made visible by the next barrier) are
    // added in a fixed order and stored to the tile's record of the column slab
    auto flush_cols = [&](int t) {
        if (tid < TILE) {
            const float *cr_ = colred + (t & 1) * 512;
            const float sum = (cr_[tid] + cr_[128 + tid]) + (cr_[256 + tid] + cr_[384 + tid]);
            float *rec = a.colslab + (rec0 + jt_begin + t) * TILE;  // uniform base + 32-bit lane offset
            rec[static_cast<unsigned>(tid)] = sum;
        }
    };

    auto tile_body = [&](int t, auto checked) {
        const int s0 = t * NKC;
        const bool tile_sym = SYM && (jt_begin + t < ib);  // strictly below the diagonal
        {
            // tile_init: per-lane column data + accumulator start values (the record became visible at the last hand-over)
            const float *dcr = reinterpret_cast<const float *>(dcs + (t % V2_DC_SLOTS) * 1024);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                dj[cb] = dcr[cb * 32 + r];
                if constexpr (KT == KT_RBF) cj[cb] = dcr[128 + cb * 32 + r];
                if constexpr (KT == KT_POLY) padcol[cb] = (a.degree < 0) && ((jt_begin + t) * TILE + cb * 32 + r >= a.ncols_valid);
            }
            // rbf: the accumulators start at c_i + c_j (vector adds; producing the sum with one extra MFMA per accumulator --
            // A = (c_i, 1), B = (1, c_j) -- was measured 0.8 % slower at c5: the adds overlap with the other workgroup's MFMAs)
            if constexpr (KT == KT_RBF) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 civ = *reinterpret_cast<const f32x4 *>(cis + wave * 32 + 8 * g4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) acc[cb][4 * g4 + e] = civ[e] + cj[cb];
                }
            }
            // the other kernels start the chain with the constant 0 as the C operand of the first MFMA (no v_mov per register)
        }
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const int step = s0 + kc;
            const char *slot = ring + (step % V2_RING) * V2_SLOT_BYTES;
            const char *slot_next = ring + ((step + 1) % V2_RING) * V2_SLOT_BYTES;
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                // software prefetch of the NEXT group's B fragments (next chunk for mm == 3: visible since this step's hand-over),
                // issued before the hand-over barrier so that LDS latency and barrier skew hide behind the 16 MFMAs below
                f32x4 bnext[4];
                if (mm < 3) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) bnext[cb] = *reinterpret_cast<const f32x4 *>(slot + cb * 4096 + rd_off[mm + 1]);
                }
                if (mm == 2) {
                    if constexpr (SYM) {
                        // the colred writes of the previous tile's epilogue must have completed before the barrier publishes them
                        if (kc == 0 && t > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    handover(step, (kc + 3) % NKC, checked);
                    if constexpr (SYM) {
                        if (kc == 0 && t > 0) flush_cols(t - 1);  // every tile before the last one of an item is off-diagonal
                    }
                }
                if (mm == 3) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) bnext[cb] = *reinterpret_cast<const f32x4 *>(slot_next + cb * 4096 + rd_off[0]);
                }
                const f32x4 av = afrag[4 * kc + mm];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        if (KT != KT_RBF && kc == 0 && mm == 0 && tt == 0) {
                            const f32x16 zero = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
                            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tt], bcur[cb][tt], zero, 0, 0, 0);
                        } else {
                            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tt], bcur[cb][tt], acc[cb], 0, 0, 0);
                        }
                    }
                    if constexpr (!decltype(checked)::value) {
                        if (mm >= 2 && (tt & 1) == 0) issue_chunk_part(step + 3, (mm - 2) * 2 + (tt >> 1));
                    }
                }
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) bcur[cb] = bnext[cb];
            }
        }
        // epilogue of the tile: K_ij = f(acc), row partial += K_ij * d_j; SYM, off-diagonal tile: column partial += K_ij * d_i
        // (vector ALU, fused; the Gram tile itself is never written)
        if (!LSSVM_DBG(a, 4))
        {  // (the polynomial degree class is a template parameter here: KT_POLY2 / KT_POLY3 / generic KT_POLY)
            auto epilogue = [&](auto with_cols) {
                constexpr bool COLS = decltype(with_cols)::value;
                float di[16];
                float colacc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
                if constexpr (COLS) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 dv = *reinterpret_cast<const f32x4 *>(dis + wave * 32 + 8 * g4 + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) di[4 * g4 + e] = dv[e];
                    }
                }
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float kv = apply_kernel_function<v2_base_kt(KT), v2_degree_class(KT)>(acc[cb][i], a);
                        if constexpr (KT == KT_POLY) {
                            if (padcol[cb]) kv = 0.0f;
                        }
                        rowpart[i] = fmaf(kv, dj[cb], rowpart[i]);
                        if constexpr (COLS) colacc[cb] = fmaf(kv, di[i], colacc[cb]);
                    }
                if constexpr (COLS) {
                    float *cw = colred + (t & 1) * 512 + wave * 128;
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        const float v = colacc[cb] + __shfl_xor(colacc[cb], 32);  // the two lane halves hold different rows
                        if (h == 0) cw[cb * 32 + r] = v;
                    }
                }
            };
            if (tile_sym) {
                epilogue(std::true_type{});
            } else {
                epilogue(std::false_type{});
            }
        }
    };

    // steady state: every tile whose last step still has step + 3 < nsteps; then the (1..3) tail tiles with the checked hand-over
    constexpr int TAIL_TILES = (3 + NKC - 1) / NKC;
    const int nmain = ntiles > TAIL_TILES ? ntiles - TAIL_TILES : 0;
    int t = 0;
    for (; t < nmain; ++t) tile_body(t, std::false_type{});
    for (; t < ntiles; ++t) tile_body(t, std::true_type{});
    if constexpr (SYM) {
        if (jt_begin + ntiles - 1 < ib) {  // the last tile was off-diagonal: publish its column sums
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            flush_cols(ntiles - 1);
        }
    }

    // every wave owns its rows: reduce over the 32 lanes of a lane-half and store
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float v = rowpart[i];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        rowpart[i] = v;
    }
    if (r == 0) {
        float *dst = a.partial + static_cast<size_t>(jc) * a.part_stride + ibl * TILE + wave * 32 + 4 * h;
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[(i & 3) + 8 * (i >> 2)] = rowpart[i];
    }
}

/* dc[jt][0..127] = d of tile jt, dc[jt][128..255] = c (rbf: -|x_j|^2/2, else unused): one 1-KiB LDS-DMA record per tile */
__global__ void k_pack_dc(const float *__restrict__ dvec, const float *__restrict__ cc, int ncols_padded, float *__restrict__ dc) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols_padded) return;
    const int jt = j >> 7, l = j & 127;
    dc[static_cast<size_t>(jt) * 256 + l] = dvec[j];
    dc[static_cast<size_t>(jt) * 256 + 128 + l] = (cc != nullptr) ? cc[j] : 0.0f;
}

/* in place: the features of every aligned group of 8 are reordered to 0,2,4,6,1,3,5,7 (fp32 HBM layout, see above) */
__global__ void k_interleave_features(float *__restrict__ X, size_t ngroups) {
    const size_t g = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    f32x4 *p = reinterpret_cast<f32x4 *>(X + 8 * g);
    const f32x4 lo = p[0], hi = p[1];
    p[0] = f32x4{ lo.x, lo.z, hi.x, hi.z };
    p[1] = f32x4{ lo.y, lo.w, hi.y, hi.w };
}

/* =====================================================================================================================
 * fp64 tile kernel: v_mfma_f64_16x16x4_f64
 *   lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15] (one f64 each); the 16x16 result has column
 *   j = l&15 on the lane and rows (l>>4) + 4*reg in its 4 registers (NOT the f32 row map).
 *   LDS image of a k-chunk: [128 rows][16 doubles + 2 pad] (144-B rows: conflict-free ds_read_b64).
 * ===================================================================================================================== */

template <int KT>
__global__ __launch_bounds__(TILE_THREADS, 1) void tile_matvec_f64(const TileArgs<double> a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *As = reinterpret_cast<double *>(smem_raw);  // [2][TILE * F64_LS]
    double *Bs = As + 2 * TILE * F64_LS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1;
    const int wc = wave & 1;
    const int r = lane & 15;
    const int qd = lane >> 4;

    int ibl, jc;
    if (!decode_work_item(a, ibl, jc)) return;
    const int row0 = (a.ib_begin + ibl) * TILE;
    const int jt_begin = jc * a.jc_tiles;
    const int jt_end = min(jt_begin + a.jc_tiles, a.num_jt);
    const int ntiles = jt_end - jt_begin;
    if (ntiles <= 0) return;

    // staging: thread -> (row = tid/8 [+32 p], 2 doubles at (tid%8)*2); 8 threads cover one 128-B line
    const int srow = tid >> 3;
    const int sseg = tid & 7;
    const double *Ag = a.Xr + static_cast<size_t>(row0 + srow) * a.ldx + sseg * 2;
    const size_t rstep = static_cast<size_t>(32) * a.ldx;
    const int lds_w = srow * F64_LS + sseg * 2;

    double rowpart[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) rowpart[mt][i] = 0.0;

    f64x4 acc[4][4];
    f64x2 sa[4], sb[4];

    auto stage_load = [&](int jt, int kc) {
        const double *Bg = a.Xc + static_cast<size_t>(jt * TILE + srow) * a.ldx + sseg * 2;
        const int ko = kc * F64_KC;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            sa[p] = *reinterpret_cast<const f64x2 *>(Ag + p * rstep + ko);
            sb[p] = *reinterpret_cast<const f64x2 *>(Bg + p * rstep + ko);
        }
    };
    auto stage_store = [&](int buf) {
        double *Aw = As + buf * TILE * F64_LS + lds_w;
        double *Bw = Bs + buf * TILE * F64_LS + lds_w;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<f64x2 *>(Aw + p * 32 * F64_LS) = sa[p];
            *reinterpret_cast<f64x2 *>(Bw + p * 32 * F64_LS) = sb[p];
        }
    };

    double ci[4][4];  // rbf: c_i of this lane's 16 rows (1 wave per SIMD: the register budget is 512)
    if constexpr (KT == KT_RBF) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) ci[mt][i] = a.cr[row0 + wr * 64 + mt * 16 + qd + 4 * i];
    }
    double dj[4], cj[4], djn[4], cjn[4];
    bool padcol[4] = { false, false, false, false };
    auto col_prefetch = [&](int jt) {  // one tile ahead, see the fp32 kernel
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int j = jt * TILE + wc * 64 + nt * 16 + r;
            djn[nt] = a.dvec[j];
            if constexpr (KT == KT_RBF) cjn[nt] = a.cc[j];
        }
    };
    auto tile_init = [&](int jt) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            dj[nt] = djn[nt];
            cj[nt] = 0.0;
            if constexpr (KT == KT_RBF) cj[nt] = cjn[nt];
            if constexpr (KT == KT_POLY) padcol[nt] = (a.degree < 0) && (jt * TILE + wc * 64 + nt * 16 + r >= a.ncols_valid);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double civ = 0.0;
                if constexpr (KT == KT_RBF) civ = ci[mt][i];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt][i] = civ + cj[nt];
            }
    };

    const int nsteps = ntiles * a.kchunks;
    stage_load(jt_begin, 0);
    col_prefetch(jt_begin);
    tile_init(jt_begin);
    stage_store(0);
    __syncthreads();

    int jt = jt_begin;
    int kc = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        int njt = jt, nkc = kc + 1;
        if (nkc == a.kchunks) {
            nkc = 0;
            ++njt;
        }
        const bool has_next = (s + 1 < nsteps);
        if (has_next) stage_load(njt, nkc);
        if (kc == 0 && jt + 1 < jt_end) col_prefetch(jt + 1);

        {
            const double *Ab = As + cur * TILE * F64_LS + (wr * 64 + r) * F64_LS + qd;
            const double *Bb = Bs + cur * TILE * F64_LS + (wc * 64 + r) * F64_LS + qd;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                double av[4], bv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    av[t] = Ab[t * 16 * F64_LS + ks * 4];
                    bv[t] = Bb[t * 16 * F64_LS + ks * 4];
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
            }
        }

        if (has_next) stage_store(cur ^ 1);

        if (kc == a.kchunks - 1) {
            with_degree_class<KT>(a, [&](auto degc) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            double kv = apply_kernel_function<KT, decltype(degc)::value>(acc[mt][nt][i], a);
                            if constexpr (KT == KT_POLY) {
                                if (padcol[nt]) kv = 0.0;
                            }
                            rowpart[mt][i] = fma(kv, dj[nt], rowpart[mt][i]);
                        }
            });
            if (has_next) tile_init(njt);
        }
        __syncthreads();
        jt = njt;
        kc = nkc;
    }

    // rows are shared by the 16 lanes of a quarter-wave (same l>>4)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double v = rowpart[mt][i];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            rowpart[mt][i] = v;
        }
    double *red = reinterpret_cast<double *>(smem_raw);  // [2][TILE]
    if (r == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wc * TILE + wr * 64 + mt * 16 + qd + 4 * i] = rowpart[mt][i];
    }
    __syncthreads();
    if (tid < TILE) {
        a.partial[static_cast<size_t>(jc) * a.part_stride + ibl * TILE + tid] = red[tid] + red[TILE + tid];
    }
}

/* =====================================================================================================================
 * fp64 tile kernel, version 2: the fp32 v2 pipeline (row panel resident in registers, LDS-DMA ring two chunks ahead, chunk
 * hand-over in the middle of a step, packed (d_j | c_j) records) on v_mfma_f64_16x16x4_f64, for num_features <= 128.
 *   A column tile of 128 is processed as two 64-column SUB-TILES so that a wave's accumulators (32 rows x 64 columns =
 *   8 tiles of 16x16 = 64 VGPRs) plus its row panel leave room for two workgroups per CU.
 *   Chunk = 64 columns x 16 features = 8 KiB = 4 k-steps of 8 MFMAs per wave.
 *   v_mfma_f64 does NOT overlap with vector ALU instructions (tests/tools/microbench_f64.hip: one integer VALU op per MFMA
 *   costs 9 % of the matrix-core rate, one v_fma_f64 15 %), so the chunk loop consists of MFMAs, LDS reads with immediate
 *   offsets, LDS-DMA with scalar base addresses and scalar instructions only, the accumulators start from the constant 0 as
 *   the C operand of the first MFMA, and the polynomial kernel runs on data pre-scaled by sqrt(gamma).
 * ===================================================================================================================== */
constexpr int V2D_RING = 4;
constexpr int V2D_SLOT_BYTES = 64 * 128;  // 8 KiB
constexpr int V2D_DC_SLOTS = 4;           // (64 d_j | 64 c_j) doubles = 1 KiB per sub-tile
constexpr size_t V2D_LDS_BYTES = static_cast<size_t>(V2D_RING) * V2D_SLOT_BYTES + V2D_DC_SLOTS * 1024 + (2 * TILE + 2 * 4 * 64) * sizeof(double);  // ring + records + cis, dis, colred

template <int KT, int NKC, bool SYM>
__global__ __launch_bounds__(TILE_THREADS, (NKC <= 4 ? 2 : 1)) void tile_matvec_f64_v2(const TileArgs<double> a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char *ring = smem_raw;
    char *dcs = smem_raw + V2D_RING * V2D_SLOT_BYTES;
    double *cis = reinterpret_cast<double *>(dcs + V2D_DC_SLOTS * 1024);  // [128] c_i of the row panel (rbf)
    double *dis = cis + TILE;                                              // [128] d_i of the row panel (SYM)
    double *colred = dis + TILE;                                           // [2][4 waves][64] column sums of a sub-tile (SYM)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15;
    const int q = lane >> 4;

    // SYM: see tile_matvec_f32_v2.  A sub-tile st is strictly below the diagonal block of row block ib iff st < 2 ib; the two
    // sub-tiles of the diagonal tile are evaluated in full and contribute to the rows only.
    int ibl, jc;
    if constexpr (SYM) {
        const int2 it = a.items[blockIdx.x];
        ibl = __builtin_amdgcn_readfirstlane(it.x);  // uniform, but loaded through the vector memory path: move to SGPRs so
        jc = __builtin_amdgcn_readfirstlane(it.y);   // that everything derived from it is scalar arithmetic
    } else {
        if (!decode_work_item(a, ibl, jc)) return;
    }
    const int ib = a.ib_begin + ibl;
    const int row0 = ib * TILE;
    const int jt_begin = jc * a.jc_tiles;
    const int jt_end = SYM ? min(jt_begin + a.jc_tiles, ib + 1) : min(jt_begin + a.jc_tiles, a.num_jt);
    const int nsub = 2 * (jt_end - jt_begin);  // 64-column sub-tiles
    if (nsub <= 0) return;
    const int st_begin = 2 * jt_begin;
    const int nsteps = nsub * NKC;
    // record index of (ib, st) in this device's packed column slab: row block b owns the 2 b sub-tiles below its diagonal
    const long rec0 = SYM ? (static_cast<long>(ib) * (ib - 1) - 2 * a.pair_origin) : 0;

    // row panel: A operand of lane (r, q) for k-step s is X[row][4 s + q]
    double afrag[2][4 * NKC];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const double *xr = a.Xr + static_cast<size_t>(row0 + wave * 32 + rb * 16 + r) * a.ldx + q;
#pragma unroll
        for (int s = 0; s < 4 * NKC; ++s) afrag[rb][s] = xr[4 * s];
    }
    if constexpr (KT == KT_RBF) {
        if (tid < TILE) cis[tid] = a.cr[row0 + tid];
    }
    if constexpr (SYM) {
        if (tid < TILE) dis[tid] = a.dvec[row0 + tid];
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int s = 0; s < 4 * NKC; ++s) asm volatile("" : "+v"(afrag[rb][s]));

    // LDS image of a chunk (64 columns x 16 features, 8 KiB): [column block cb = 0..3][16-byte k-slot ks = 0..7][column r = 0..15],
    // i.e. byte cb * 2048 + ks * 256 + r * 16 holds features 2 ks, 2 ks + 1 of column cb * 16 + r.  Piece 2 * wave + i of the DMA
    // (1 KiB, lane-linear in LDS) is block cb = wave, k-slots 4 i .. 4 i + 3: lane L fetches the 16 bytes of column L % 16, k-slot
    // 4 i + L / 16 -- a gather on the SOURCE side (16 rows x 64 contiguous bytes per piece).
    unsigned dma_off[2];  // byte offsets (saddr form: uniform base in SGPRs + 32-bit lane offset, see tile_matvec_f32_v2)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int col = wave * 16 + (lane & 15);
        const int ks = 4 * i + (lane >> 4);
        dma_off[i] = 8u * static_cast<unsigned>(col * a.ldx + 2 * ks);
    }
    auto issue_chunk = [&](int step) {
        if (LSSVM_DBG(a, 16) && step > 2) return;  // ablation: no DMA after the prologue
        const int t = step / NKC;
        const int kc = step - t * NKC;
        const char *base = sgpr_ptr(a.Xc + static_cast<size_t>(st_begin + t) * 64 * a.ldx + kc * 16);
        char *slot = ring + (step % V2D_RING) * V2D_SLOT_BYTES + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t) (base + lane_off(dma_off[i])), (lds_ptr_t) (slot + i * 1024), 16, 0, 0);
        }
    };
    auto issue_dc = [&](int t) {
        if (lane < 16) {
            const char *src = sgpr_ptr(a.dc + static_cast<size_t>(st_begin + t) * 128) + __builtin_amdgcn_readfirstlane(wave * 256);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t) (src + 16u * (lane_off(threadIdx.x) & 15u)), (lds_ptr_t) (dcs + (t % V2D_DC_SLOTS) * 1024 + wave * 256), 16, 0, 0);
        }
    };

    // Read addressing: lane (r, q) needs feature 4 s + q of column cb * 16 + r for k-step s = k-slot 2 s + q / 2, half q % 2:
    // byte (q / 2) * 256 + r * 16 + (q % 2) * 8 [per lane, constant] + cb * 2048 + s * 512 [immediates] + ring slot [one add per
    // chunk].  The 32 lanes of a ds_read_b64 group (q / 2 fixed) read 256 contiguous bytes: conflict free without a swizzle.
    const int lane_base = (q >> 1) * 256 + r * 16 + (q & 1) * 8;
    auto read_group = [&](const char *slot, int s, double (&b)[4]) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) b[cb] = *((const volatile __attribute__((address_space(3))) double *) (slot + cb * 2048 + s * 512));  // volatile: keeps ds_read_b64 (a fused ds_read2st64_b64 is banked modulo 32 dwords: 2-way conflicts here)
    };

    double rowpart[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i) rowpart[rb][i] = 0.0;
    f64x4 acc[2][4];
    double dj[4], cj[4];
    bool padcol[4] = { false, false, false, false };

    // ---- prologue: chunks 0, 1, 2 (each preceded by the record of the sub-tile that starts with it) ----
    issue_dc(0);
    issue_chunk(0);
#pragma unroll
    for (int pre = 1; pre <= 2; ++pre) {
        if (pre < nsteps) {
            if (pre % NKC == 0) issue_dc(pre / NKC);
            issue_chunk(pre);
        }
    }
    // chunk 0 (and record 0, cis, dis) complete: everything but the DMA pieces of the younger chunks is done
    if (nsteps >= 3) {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    } else if (nsteps == 2) {
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    double bcur[4];  // B fragments of the k-step about to be multiplied (double buffered against bnext in the loop)
    read_group(ring + lane_base, 0, bcur);

    // ---- hand-over of the NEXT chunk in the MIDDLE of a step (see tile_matvec_f32_v2): called in k-step 2 of chunk `step`.  This
    // wave's two pieces of chunk step + 1 are complete once all but its 2 youngest DMA instructions (chunk step + 2) are; the barrier
    // makes every wave's pieces visible, so k-step 3 can already prefetch the first fragments of chunk step + 1.  Ring of 4 slots:
    // the DMA issued here (chunk step + 3) overwrites the slot of chunk step - 1, which every wave finished before this barrier.
    auto handover = [&](int step, int kc_plus3_mod, auto checked) {
        constexpr bool CHECKED = decltype(checked)::value;
        if constexpr (!CHECKED) {
            if (!LSSVM_DBG(a, 16)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if (!LSSVM_DBG(a, 8)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kc_plus3_mod == 0) issue_dc((step + 3) / NKC);
            issue_chunk(step + 3);
        } else {
            if (step + 1 < nsteps) {
                if (step + 2 < nsteps) {
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (step + 3 < nsteps) {
                    if (kc_plus3_mod == 0) issue_dc((step + 3) / NKC);
                    issue_chunk(step + 3);
                }
            }
        }
    };

    auto flush_cols = [&](int t) {  // fixed-order sum of the four waves' column sums of sub-tile t -> its slab record
        if (tid < 64) {
            const double *cr_ = colred + (t & 1) * 256;
            double *rec = a.colslab + (rec0 + st_begin + t) * 64;  // uniform base + 32-bit lane offset
            rec[static_cast<unsigned>(tid)] = (cr_[tid] + cr_[64 + tid]) + (cr_[128 + tid] + cr_[192 + tid]);
        }
    };

    auto tile_body = [&](int t, auto checked) {
        const int s0 = t * NKC;
        const bool tile_sym = SYM && (st_begin + t < 2 * ib);
        const double *dcr = reinterpret_cast<const double *>(dcs + (t % V2D_DC_SLOTS) * 1024);
        // rbf: the accumulators start at c_i + c_j; the other kernels start the chain with the constant 0 as the C operand of the
        // first MFMA (no register initialisation: 64 v_mov per sub-tile would cost as much matrix-core time as the cube)
        if constexpr (KT == KT_RBF) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) cj[cb] = dcr[64 + cb * 16 + r];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double civ = cis[wave * 32 + rb * 16 + q + 4 * i];
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) acc[rb][cb][i] = civ + cj[cb];
                }
        }
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const int step = s0 + kc;
            const char *slot = ring + (step % V2D_RING) * V2D_SLOT_BYTES + lane_base;
            const char *slot_next = ring + ((step + 1) % V2D_RING) * V2D_SLOT_BYTES + lane_base;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                // software prefetch of the next k-step's B fragments (next chunk for s == 3: visible since this step's hand-over)
                double bnext[4];
                if (s < 3) read_group(slot, s + 1, bnext);
                if (s == 2) {
                    if constexpr (SYM) {
                        // the colred writes of the previous sub-tile's epilogue must have completed before the barrier publishes them
                        if (kc == 0 && t > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    handover(step, (kc + 3) % NKC, checked);
                    if constexpr (SYM) {
                        // sub-tile t - 1 was off-diagonal unless it is the first of the diagonal pair
                        if (kc == 0 && t > 0 && (st_begin + t - 1 < 2 * ib)) flush_cols(t - 1);
                    }
                }
                if (s == 3) read_group(slot_next, 0, bnext);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        if (KT != KT_RBF && kc == 0 && s == 0) {
                            const f64x4 zero = { 0.0, 0.0, 0.0, 0.0 };
                            acc[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[rb][0], bcur[cb], zero, 0, 0, 0);
                        } else {
                            acc[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[rb][4 * kc + s], bcur[cb], acc[rb][cb], 0, 0, 0);
                        }
                    }
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) bcur[cb] = bnext[cb];
            }
        }
        // ---- epilogue of the sub-tile (vector ALU; every instruction here costs matrix-core time, see the header) ----
        if (!LSSVM_DBG(a, 4)) {
            // d_j is fetched from the sub-tile's record only now: it need not occupy registers during the MFMA loop
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                dj[cb] = dcr[cb * 16 + r];
                if constexpr (KT == KT_POLY) padcol[cb] = (a.degree < 0) && ((st_begin + t) * 64 + cb * 16 + r >= a.ncols_valid);
            }
            if constexpr (v2_base_kt(KT) == KT_POLY) {
                if (a.coef0 != 0.0) {  // uniform; the common coef0 = 0 costs nothing
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[rb][cb][i] += a.coef0;
                }
            }
            auto epilogue = [&](auto with_cols) {
                constexpr bool COLS = decltype(with_cols)::value;
                double colacc[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        double di = 0.0;
                        if constexpr (COLS) di = dis[wave * 32 + rb * 16 + q + 4 * i];
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) {
                            double kv;
                            if constexpr (v2_base_kt(KT) == KT_POLY) {
                                // the data carries sqrt(gamma) (Problem<double> pre-scales it for this kernel) and coef0 was added above
                                kv = poly_power<v2_degree_class(KT)>(acc[rb][cb][i], a.degree);
                            } else {
                                kv = apply_kernel_function<KT, 0>(acc[rb][cb][i], a);
                            }
                            if constexpr (KT == KT_POLY) {
                                if (padcol[cb]) kv = 0.0;
                            }
                            rowpart[rb][i] = fma(kv, dj[cb], rowpart[rb][i]);
                            if constexpr (COLS) colacc[cb] = fma(kv, di, colacc[cb]);
                        }
                    }
                if constexpr (COLS) {
                    double *cw = colred + (t & 1) * 256 + wave * 64;
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        double v = colacc[cb];
                        v += __shfl_xor(v, 16);  // the four quarter-waves hold different rows of the same column
                        v += __shfl_xor(v, 32);
                        if (q == 0) cw[cb * 16 + r] = v;
                    }
                }
            };
            if (tile_sym) {
                epilogue(std::true_type{});
            } else {
                epilogue(std::false_type{});
            }
        }
    };

    // steady state: every sub-tile whose last step still has step + 3 < nsteps; then the tail sub-tiles with the checked hand-over
    constexpr int TAIL_TILES = (3 + NKC - 1) / NKC;
    const int nmain = nsub > TAIL_TILES ? nsub - TAIL_TILES : 0;
    int t = 0;
    for (; t < nmain; ++t) tile_body(t, std::false_type{});
    for (; t < nsub; ++t) tile_body(t, std::true_type{});
    if constexpr (SYM) {
        if (st_begin + nsub - 1 < 2 * ib) {  // the last sub-tile was off-diagonal: publish its column sums
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            flush_cols(nsub - 1);
        }
    }

    // rows are shared by the 16 lanes of a quarter-wave
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double v = rowpart[rb][i];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            rowpart[rb][i] = v;
        }
    if (r == 0) {
        double *dst = a.partial + static_cast<size_t>(jc) * a.part_stride + ibl * TILE + wave * 32 + q;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[rb * 16 + 4 * i] = rowpart[rb][i];
    }
}

/* fp64 records of the v2 kernel: per 64-column sub-tile st: dc[st][0..63] = d, dc[st][64..127] = c */
__global__ void k_pack_dc_f64(const double *__restrict__ dvec, const double *__restrict__ cc, int ncols_padded, double *__restrict__ dc) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols_padded) return;
    const int st = j >> 6, l = j & 63;
    dc[static_cast<size_t>(st) * 128 + l] = dvec[j];
    dc[static_cast<size_t>(st) * 128 + 64 + l] = (cc != nullptr) ? cc[j] : 0.0;
}

/* =====================================================================================================================
 * Direct-form RBF on the vector ALU (fp32): accumulates (x_i - x_j)^2 exactly as the reference does
 * (HIP/svm_kernel.hip.hpp:247, operators.hpp:161-171).  1 sub + 1 fma per (i, j, feature): at most half of the fp32 FMA
 * peak.  Kept as the formula-exact alternative to the matrix-core path (option "rbf_form" = 1) and as its on-device
 * cross-check.  Each thread owns an 8 x 8 register tile of a 128 x 128 workgroup tile; operands come from the same
 * k-chunked LDS images (no k interleave needed here; rows padded to 33 floats).
 * ===================================================================================================================== */
constexpr int DIR_KC = 32;
constexpr int DIR_LS = 33;

__global__ __launch_bounds__(TILE_THREADS, 2) void tile_matvec_rbf_direct_f32(const TileArgs<float> a) {
    __shared__ float As[TILE * DIR_LS];
    __shared__ float Bs[TILE * DIR_LS];
    __shared__ float red[16][TILE];

    const int tid = threadIdx.x;
    const int tx = tid & 15;  // column group: columns tx + 16*c
    const int ty = tid >> 4;  // row group:    rows    ty + 16*rr
    int ibl, jc;
    if (!decode_work_item(a, ibl, jc)) return;
    const int row0 = (a.ib_begin + ibl) * TILE;
    const int jt_begin = jc * a.jc_tiles;
    const int jt_end = min(jt_begin + a.jc_tiles, a.num_jt);
    if (jt_end <= jt_begin) return;

    float rowpart[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rowpart[i] = 0.0f;

    const int srow = tid >> 3;  // 0..31
    const int sseg = tid & 7;   // 4 floats each
    for (int jt = jt_begin; jt < jt_end; ++jt) {
        float acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = 0.0f;
        for (int kc = 0; kc < a.kchunks; ++kc) {
            __syncthreads();
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = srow + 32 * p;
                const f32x4 va = *reinterpret_cast<const f32x4 *>(a.Xr + static_cast<size_t>(row0 + row) * a.ldx + kc * DIR_KC + sseg * 4);
                const f32x4 vb = *reinterpret_cast<const f32x4 *>(a.Xc + static_cast<size_t>(jt * TILE + row) * a.ldx + kc * DIR_KC + sseg * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    As[row * DIR_LS + sseg * 4 + e] = va[e];
                    Bs[row * DIR_LS + sseg * 4 + e] = vb[e];
                }
            }
            __syncthreads();
#pragma unroll 4
            for (int k = 0; k < DIR_KC; ++k) {
                float av[8], bv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) av[i] = As[(ty + 16 * i) * DIR_LS + k];
#pragma unroll
                for (int j = 0; j < 8; ++j) bv[j] = Bs[(tx + 16 * j) * DIR_LS + k];
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float diff = av[i] - bv[j];
                        acc[i][j] = fmaf(diff, diff, acc[i][j]);
                    }
            }
        }
        // epilogue: exp(-gamma * dist^2) * d_j ; for THIS kernel the gamma field carries -gamma*log2(e)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float dv = a.dvec[jt * TILE + tx + 16 * j];
#pragma unroll
            for (int i = 0; i < 8; ++i) rowpart[i] = fmaf(__builtin_amdgcn_exp2f(acc[i][j] * a.gamma), dv, rowpart[i]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) red[tx][ty + 16 * i] = rowpart[i];
    __syncthreads();
    if (tid < TILE) {
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) s += red[c][tid];
        a.partial[static_cast<size_t>(jc) * a.part_stride + ibl * TILE + tid] = s;
    }
}

/* =====================================================================================================================
 * O(n) / O(n d) helper kernels
 * ===================================================================================================================== */

/* deterministic block reduction of `NV` doubles per thread; result valid in thread 0 */
template <int NV>
__device__ __forceinline__ void block_reduce(double (&v)[NV], double *lds /* [NV][4] for 256 threads */) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) lds[k * 4 + wave] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = (lds[k * 4 + 0] + lds[k * 4 + 1]) + (lds[k * 4 + 2] + lds[k * 4 + 3]);
    }
}

/* part[b][0] = sum v, part[b][1] = sum q*v over the block's grid-stride slice */
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void k_sum_and_qdot(const T *__restrict__ v, const T *__restrict__ q, int n, double *__restrict__ part) {
    __shared__ double lds[8];
    double acc[2] = { 0.0, 0.0 };
    for (int i = blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += RED_BLOCKS * RED_THREADS) {
        const double vi = static_cast<double>(v[i]);
        acc[0] += vi;
        acc[1] += vi * static_cast<double>(q[i]);
    }
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = acc[0];
        part[blockIdx.x * 2 + 1] = acc[1];
    }
}

/* single block: out[slot0] = sum part[.][0], out[slot1] = sum part[.][1]  (fixed tree order) */
__global__ __launch_bounds__(RED_THREADS) void k_finish2(const double *__restrict__ part, double *__restrict__ sc, int slot0, int slot1) {
    __shared__ double lds[8];
    double acc[2] = { part[threadIdx.x * 2 + 0], part[threadIdx.x * 2 + 1] };
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) {
        sc[slot0] = acc[0];
        if (slot1 >= 0) sc[slot1] = acc[1];
    }
}

/* Kv[row_begin + i] = sum over column chunks of partial[c][i], chunks in ascending order (rows of this device only) */
template <typename T>
__global__ void k_reduce_partials(const T *__restrict__ partial, long part_stride, int num_jc, int row_begin, int nrows, T *__restrict__ Kv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows) {
        T s = partial[i];
        for (int c = 1; c < num_jc; ++c) s += partial[static_cast<size_t>(c) * part_stride + i];
        Kv[row_begin + i] = s;
    }
}

/* SYM: Kv[column c] += sum over the row blocks ib (of this device) below column record c of colslab[(ib, c)], in a FIXED order.
 * Records are W columns wide, SUB = 128 / W records per 128-column tile (fp32: W = 128, fp64: W = 64); row block ib owns the
 * SUB * ib records below its diagonal tile, packed triangularly.  One block of 1024 threads per record column: 1024 / W groups
 * walk the row blocks with stride 1024 / W (four independent partial sums each, for memory-level parallelism), then the groups'
 * sums are added in group order -- the order depends only on the shape, never on timing. */
template <typename T, int W>
__global__ __launch_bounds__(1024) void k_reduce_colslab(const T *__restrict__ colslab, long pair_origin, int ib_begin, int ib_end, T *__restrict__ Kv) {
    constexpr int SUB = TILE / W;
    constexpr int G = 1024 / W;
    __shared__ T red[G][W];
    const int c = blockIdx.x;
    const int l = threadIdx.x % W;
    const int g = threadIdx.x / W;
    const int first = max(c / SUB + 1, ib_begin);
    auto rec = [&](int ib) { return colslab[(SUB * (static_cast<long>(ib) * (ib - 1) / 2 - pair_origin) + c) * W + l]; };
    T s0 = T(0), s1 = T(0), s2 = T(0), s3 = T(0);
    int ib = first + g;
    for (; ib + 3 * G < ib_end; ib += 4 * G) {
        s0 += rec(ib);
        s1 += rec(ib + G);
        s2 += rec(ib + 2 * G);
        s3 += rec(ib + 3 * G);
    }
    for (; ib < ib_end; ib += G) s0 += rec(ib);
    red[g][l] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0) {
        T s = red[0][l];
#pragma unroll
        for (int k = 1; k < G; ++k) s += red[k][l];
        Kv[c * W + l] += s;
    }
}

/* SYM: Kv[row_begin + i] = sum of the row slabs of the column chunks that exist for the row's block (chunks 0 .. ib / jc_tiles) */
template <typename T>
__global__ void k_reduce_partials_sym(const T *__restrict__ partial, long part_stride, int jc_tiles, int ib_begin, int nrows, T *__restrict__ Kv, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows) {
        const int ib = ib_begin + i / TILE;
        const int nchunks = ib / jc_tiles + 1;
        T s = partial[i];
        for (int c = 1; c < nchunks; ++c) s += partial[static_cast<size_t>(c) * part_stride + i];
        const int row = ib_begin * TILE + i;
        Kv[row] = accumulate ? Kv[row] + s : s;
    }
}

/* (Abar v)_i = Kv_i + v_i/C + (QA_cost*S - q.v) - S*q_i, evaluated in double */
template <typename T>
__device__ __forceinline__ double abar_row(const T *Kv, const T *v, const T *q, int i, double inv_cost, double QA_cost, double S, double QV) {
    return static_cast<double>(Kv[i]) + static_cast<double>(v[i]) * inv_cost + (QA_cost * S - QV) - S * static_cast<double>(q[i]);
}

/* ret_i += add * (Abar v)_i       (run_device_kernel semantics, csvm.cpp:283-306) */
template <typename T>
__global__ void k_apply_ret(const T *__restrict__ Kv, const T *__restrict__ v, const T *__restrict__ q, const double *__restrict__ sc, int n,
                            double inv_cost, double QA_cost, double add, T *__restrict__ ret) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double val = abar_row(Kv, v, q, i, inv_cost, QA_cost, sc[SC_S], sc[SC_QD]);
        ret[i] = static_cast<T>(static_cast<double>(ret[i]) + add * val);
    }
}

/* Ad_i = (Abar d)_i ; part[b][0] = sum d_i Ad_i     (csvm.cpp:131-135) */
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void k_Ad_and_dAd(const T *__restrict__ Kv, const T *__restrict__ d, const T *__restrict__ q,
                                                            const double *__restrict__ sc, int n, double inv_cost, double QA_cost,
                                                            T *__restrict__ Ad, double *__restrict__ part) {
    __shared__ double lds[8];
    const double S = sc[SC_S], QD = sc[SC_QD];
    double acc[2] = { 0.0, 0.0 };
    for (int i = blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += RED_BLOCKS * RED_THREADS) {
        const T adi = static_cast<T>(abar_row(Kv, d, q, i, inv_cost, QA_cost, S, QD));
        Ad[i] = adi;
        acc[0] += static_cast<double>(d[i]) * static_cast<double>(adi);
    }
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = acc[0];
        part[blockIdx.x * 2 + 1] = 0.0;
    }
}

/* alpha_cd = delta / (d^T Ad)      (csvm.cpp:135) */
__global__ __launch_bounds__(RED_THREADS) void k_finish_alpha(const double *__restrict__ part, double *__restrict__ sc) {
    __shared__ double lds[8];
    double acc[2] = { part[threadIdx.x * 2 + 0], 0.0 };
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) {
        sc[SC_DAD] = acc[0];
        sc[SC_ALPHA] = sc[SC_DELTA] / acc[0];
    }
}

/* x += alpha_cd d ; r -= alpha_cd Ad ; part = sum r^2      (csvm.cpp:138, :148, :153).  The scalar is rounded to T first,
 * as the reference's real_type alpha_cd is. */
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void k_update_x_r(T *__restrict__ x, T *__restrict__ r, const T *__restrict__ d, const T *__restrict__ Ad,
                                                            const double *__restrict__ sc, int n, int update_r, double *__restrict__ part) {
    __shared__ double lds[8];
    const T alpha = static_cast<T>(sc[SC_ALPHA]);
    double acc[2] = { 0.0, 0.0 };
    for (int i = blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += RED_BLOCKS * RED_THREADS) {
        x[i] += alpha * d[i];
        if (update_r) {
            const T ri = r[i] - alpha * Ad[i];
            r[i] = ri;
            acc[0] += static_cast<double>(ri) * static_cast<double>(ri);
        }
    }
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = acc[0];
        part[blockIdx.x * 2 + 1] = 0.0;
    }
}

/* r_i = b_i - (Abar x)_i ; part = sum r^2       (csvm.cpp:101-107 and the refresh :140-145).  Uses SC_SUMX / SC_QX. */
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void k_residual(const T *__restrict__ Kv, const T *__restrict__ x, const T *__restrict__ q, const T *__restrict__ b,
                                                          const double *__restrict__ sc, int n, double inv_cost, double QA_cost, T *__restrict__ r,
                                                          double *__restrict__ part) {
    __shared__ double lds[8];
    const double S = sc[SC_SUMX], QX = sc[SC_QX];
    double acc[2] = { 0.0, 0.0 };
    for (int i = blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += RED_BLOCKS * RED_THREADS) {
        const T ri = static_cast<T>(static_cast<double>(b[i]) - abar_row(Kv, x, q, i, inv_cost, QA_cost, S, QX));
        r[i] = ri;
        acc[0] += static_cast<double>(ri) * static_cast<double>(ri);
    }
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = acc[0];
        part[blockIdx.x * 2 + 1] = 0.0;
    }
}

/* delta_old = delta ; delta = sum r^2 ; beta = delta / delta_old ; publish delta to the host-mapped word  (csvm.cpp:152-161) */
__global__ __launch_bounds__(RED_THREADS) void k_finish_delta(const double *__restrict__ part, double *__restrict__ sc, double *__restrict__ host_delta, int is_initial) {
    __shared__ double lds[8];
    double acc[2] = { part[threadIdx.x * 2 + 0], 0.0 };
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) {
        const double delta_old = sc[SC_DELTA];
        sc[SC_DELTA_OLD] = delta_old;
        sc[SC_DELTA] = acc[0];
        if (is_initial) {
            sc[SC_DELTA0] = acc[0];
            sc[SC_BETA] = 0.0;
        } else {
            sc[SC_BETA] = acc[0] / delta_old;
        }
        *host_delta = acc[0];
    }
}

/* d = beta d + r ; part = (sum d, sum q d) for the next matvec     (csvm.cpp:163) */
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void k_update_d(T *__restrict__ d, const T *__restrict__ r, const T *__restrict__ q, const double *__restrict__ sc,
                                                          int n, int copy_only, double *__restrict__ part) {
    __shared__ double lds[8];
    const T beta = static_cast<T>(sc[SC_BETA]);
    double acc[2] = { 0.0, 0.0 };
    for (int i = blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += RED_BLOCKS * RED_THREADS) {
        const T di = copy_only ? r[i] : beta * d[i] + r[i];
        d[i] = di;
        acc[0] += static_cast<double>(di);
        acc[1] += static_cast<double>(di) * static_cast<double>(q[i]);
    }
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = acc[0];
        part[blockIdx.x * 2 + 1] = acc[1];
    }
}

template <typename T>
__global__ void k_fill(T *__restrict__ v, int n, T value) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = value;
}

/* b_i = y_i - y_last     (csvm.cpp:89-91) */
template <typename T>
__global__ void k_make_b(const T *__restrict__ y, int n, T *__restrict__ b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = y[i] - y[n];
}

/* q_i = k(x_i, x_last): ONE THREAD PER ROW, sequential fma chain over the features in ascending order, exactly the
 * reference's operators.hpp:117-126 / :161-171 chain => q is bit-identical to the OpenMP backend for the linear kernel
 * and differs only through pow/exp otherwise.  O(N d), once per solve (q_kernel.cpp:18-55, HIP/q_kernel.hip.hpp:33-85). */
template <int KT, typename T>
__global__ void k_q(const T *__restrict__ X, int ldx, int dfeat, int n, const T *__restrict__ xlast, int degree, T gamma, T coef0, T *__restrict__ q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T *xi = X + static_cast<size_t>(i) * ldx;
    T val = T(0);
    for (int f = 0; f < dfeat; ++f) {
        if constexpr (KT == KT_RBF) {
            const T diff = xi[f] - xlast[f];
            val = fma(diff, diff, val);
        } else {
            val = fma(xi[f], xlast[f], val);
        }
    }
    if constexpr (KT == KT_LINEAR) {
        q[i] = val;
    } else if constexpr (KT == KT_POLY) {
        q[i] = ipow(fma(gamma, val, coef0), degree);
    } else {
        q[i] = exp(-gamma * val);
    }
}

/* column sums in double, deterministic two-stage: stage 1 = one block per 256-row slab */
template <typename T>
__global__ void k_colsum_stage1(const T *__restrict__ X, int ldx, int nrows, int rows_per_block, double *__restrict__ part /* [gridDim.x][ldx] */) {
    const int f = threadIdx.x + blockIdx.y * blockDim.x;
    if (f >= ldx) return;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(r0 + rows_per_block, nrows);
    double s = 0.0;
    for (int i = r0; i < r1; ++i) s += static_cast<double>(X[static_cast<size_t>(i) * ldx + f]);
    part[static_cast<size_t>(blockIdx.x) * ldx + f] = s;
}
template <typename T>
__global__ void k_colsum_stage2(const double *__restrict__ part, int nblocks, int ldx, int nrows, T *__restrict__ mean) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= ldx) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += part[static_cast<size_t>(b) * ldx + f];
    mean[f] = static_cast<T>(s / static_cast<double>(nrows));
}
/* X[i][f] = (X[i][f] - mean[f]) * scale for the valid rows / features only (padding stays exactly zero) */
template <typename T>
__global__ void k_center(T *__restrict__ X, int ldx, int dfeat, int nrows, const T *__restrict__ mean, T scale) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (f < dfeat && i < nrows) X[static_cast<size_t>(i) * ldx + f] = (X[static_cast<size_t>(i) * ldx + f] - (mean != nullptr ? mean[f] : T(0))) * scale;
}
/* c_i = -0.5 * |x_i|^2 (one wave per row, coalesced) */
template <typename T>
__global__ void k_half_neg_norms(const T *__restrict__ X, int ldx, int nrows_total, T *__restrict__ c) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= nrows_total) return;
    const T *x = X + static_cast<size_t>(row) * ldx;
    T s = T(0);
    for (int f = lane; f < ldx; f += 64) s = fma(x[f], x[f], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) c[row] = T(-0.5) * s;
}

/* w[f] = sum_i alpha_i X[i][f]   (calculate_w, csvm.cpp:255-280 / HIP/predict_kernel.hip.hpp:34-45): sequential fma chain
 * over the points per feature like the reference; coalesced because consecutive threads own consecutive features */
template <typename T>
__global__ void k_calculate_w(const T *__restrict__ X, int ldx, int dfeat, int npoints, const T *__restrict__ alpha, T *__restrict__ w) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= dfeat) return;
    T s = T(0);
    for (int i = 0; i < npoints; ++i) s = fma(alpha[i], X[static_cast<size_t>(i) * ldx + f], s);
    w[f] = s;
}
/* out_p = w . x_p - rho  (linear predict, csvm.cpp:213): one thread per point, sequential fma chain */
template <typename T>
__global__ void k_predict_linear(const T *__restrict__ P, int ldx, int dfeat, int npoints, const T *__restrict__ w, T rho, T *__restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npoints) return;
    const T *x = P + static_cast<size_t>(p) * ldx;
    T s = T(0);
    for (int f = 0; f < dfeat; ++f) s = fma(w[f], x[f], s);
    out[p] = s - rho;
}
/* out_p = Kv_p - rho */
template <typename T>
__global__ void k_sub_rho(const T *__restrict__ Kv, int n, T rho, T *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = Kv[i] - rho;
}

}  // namespace lssvm
