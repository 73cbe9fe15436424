/*
 * tile_launch_f32s.hip -- instantiates and launches the fp32 "bf16x6" split tile kernel (lssvm_tile_f32_split.hip.hpp; option
 * gram_mode = 1) and the plane-splitting set-up kernel.  Compiled for gfx950 only.
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
/* x = hi + mid + lo, each rounded to nearest-even bf16 of the remainder (exact: the remainders are representable in fp32).
 * X: [rows][ldx] fp32, features in natural order; planes: [3][rows][ldx16] bf16, zero padded. */
__global__ void k_split_bf16x3(const float *__restrict__ X, int ldx, int dfeat, size_t rows, int ldx16, uint16_t *__restrict__ planes, size_t plane_stride) {
    const size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t total = rows * static_cast<size_t>(ldx16);
    if (idx >= total) return;
    const size_t row = idx / ldx16;
    const int f = static_cast<int>(idx - row * ldx16);
    const float x = f < dfeat ? X[row * ldx + f] : 0.0f;
    const __bf16 hi = static_cast<__bf16>(x);
    const float r1 = x - static_cast<float>(hi);
    const __bf16 mid = static_cast<__bf16>(r1);
    const float r2 = r1 - static_cast<float>(mid);
    const __bf16 lo = static_cast<__bf16>(r2);
    planes[idx] = __builtin_bit_cast(uint16_t, hi);
    planes[plane_stride + idx] = __builtin_bit_cast(uint16_t, mid);
    planes[2 * plane_stride + idx] = __builtin_bit_cast(uint16_t, lo);
}
#endif


template <int KT, bool SYM>
static void launch_s6_kt(const TileArgs<float> &a, dim3 grid, hipStream_t s) {
    const dim3 block(TILE_THREADS);
    // (<= 128 features: the hand-scheduled groups; the run-time integer power and everything wider: the compiler-scheduled groups -- one kernel per
    // case, see tile_launch_f32h.hip)
#define LSSVM_S6_CASE(N)                                                                                  \
    case N:                                                                                               \
        if constexpr (KT != KT_POLY && N <= 2) {                                                          \
            ensure_dynamic_lds(tile_matvec_f32_s6h<KT, N, SYM>, V2_LDS_BYTES);                            \
            hipLaunchKernelGGL((tile_matvec_f32_s6h<KT, N, SYM>), grid, block, V2_LDS_BYTES, s, a);       \
        } else {                                                                                          \
            ensure_dynamic_lds(tile_matvec_f32_s6w<KT, N, SYM>, V2_LDS_BYTES);                            \
            hipLaunchKernelGGL((tile_matvec_f32_s6w<KT, N, SYM>), grid, block, V2_LDS_BYTES, s, a);       \
        }                                                                                                 \
        break;
    switch (a.nk64) {
#ifdef LSSVM_DEV_SUBSET  // development builds (make DEV=1): 128 and 256 features only
        LSSVM_S6_CASE(2) LSSVM_S6_CASE(4)
#else
        LSSVM_S6_CASE(1) LSSVM_S6_CASE(2) LSSVM_S6_CASE(3) LSSVM_S6_CASE(4) LSSVM_S6_CASE(5) LSSVM_S6_CASE(6)
#endif
        default: throw Error(LSSVM_ERR_INTERNAL, "no split tile kernel for this number of features");
    }
#undef LSSVM_S6_CASE
}

template <bool SYM>
static void launch_s6(const TileArgs<float> &a, int kernel_type, dim3 grid, hipStream_t s) {
    switch (kernel_type) {
        case KT_LINEAR: launch_s6_kt<KT_LINEAR, SYM>(a, grid, s); break;
        case KT_POLY:
            if (a.degree == 3) {
                launch_s6_kt<KT_POLY3, SYM>(a, grid, s);
            } else if (a.degree == 2) {
                launch_s6_kt<KT_POLY2, SYM>(a, grid, s);
            } else {
                launch_s6_kt<KT_POLY, SYM>(a, grid, s);
            }
            break;
        default:
            if (a.dc_folded != 0) {
                launch_s6_kt<KT_RBFF, SYM>(a, grid, s);
            } else {
                launch_s6_kt<KT_RBF, SYM>(a, grid, s);
            }
            break;
    }
}

void launch_split_tile_kernel_sym(const TileArgs<float> &a, int kernel_type, hipStream_t s);  // tile_launch_f32s_sym.hip

#if LSSVM_TU_HALF == 1
void launch_split_tile_kernel_sym(const TileArgs<float> &a, int kernel_type, hipStream_t s) {
    launch_s6<true>(a, kernel_type, dim3(static_cast<unsigned>(a.num_items)), s);
}
#else
void launch_split_tile_kernel(const TileArgs<float> &a, int kernel_type, dim3 grid, hipStream_t s) {
    if (a.items != nullptr) {
        launch_split_tile_kernel_sym(a, kernel_type, s);
    } else {
        launch_s6<false>(a, kernel_type, grid, s);
    }
    LSSVM_HIP_CHECK(hipGetLastError());
}

void split_bf16_planes(const float *X, int ldx, int dfeat, size_t rows, int ldx16, uint16_t *planes, size_t plane_stride, hipStream_t s) {
    const size_t total = rows * static_cast<size_t>(ldx16);
    hipLaunchKernelGGL(k_split_bf16x3, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, X, ldx, dfeat, rows, ldx16, planes, plane_stride);
    LSSVM_HIP_CHECK(hipGetLastError());
}

#endif

}  // namespace lssvm
