/*
 * tile_launch_f32s.hip -- instantiates and launches the fp32 "bf16x6" split tile kernel (lssvm_tile_f32_split.hip.hpp; option
 * gram_mode = 1) and the plane-splitting set-up kernel.  Compiled for gfx950 only.
 */
#include "tile_launch.hip.hpp"

#include "lssvm_tile_f32_split.hip.hpp"

namespace lssvm {

template <int KT, bool SYM>
static void launch_s6_kt(const TileArgs<float> &a, dim3 grid, hipStream_t s) {
    const dim3 block(TILE_THREADS);
    const size_t V2_LDS_BYTES = lssvm::V2_LDS_BYTES + static_cast<size_t>(a.lds_extra_kb) * 1024;  // experiment knob: limits workgroups per CU
#define LSSVM_S6_CASE(N)                                                                                  \
    case N:                                                                                               \
        if (a.mfma_shape == 2) {                                                                          \
            if constexpr (N <= 2) {                                                                       \
                ensure_dynamic_lds(tile_matvec_f32_s6h<KT, N, SYM>, V2_LDS_BYTES);                        \
                hipLaunchKernelGGL((tile_matvec_f32_s6h<KT, N, SYM>), grid, block, V2_LDS_BYTES, s, a);   \
                break;                                                                                    \
            }                                                                                             \
        }                                                                                                 \
        if (a.mfma_shape >= 1) {                                                                          \
            ensure_dynamic_lds(tile_matvec_f32_s6w<KT, N, SYM>, V2_LDS_BYTES);                            \
            hipLaunchKernelGGL((tile_matvec_f32_s6w<KT, N, SYM>), grid, block, V2_LDS_BYTES, s, a);       \
        } else if constexpr (KT != KT_RBFF) {                                                             \
            ensure_dynamic_lds(tile_matvec_f32_s6<KT, N, SYM>, V2_LDS_BYTES);                             \
            hipLaunchKernelGGL((tile_matvec_f32_s6<KT, N, SYM>), grid, block, V2_LDS_BYTES, s, a);        \
        } else {                                                                                          \
            throw Error(LSSVM_ERR_INTERNAL, "folded rbf records need the 16x16x32 kernels");              \
        }                                                                                                 \
        break;
    switch (a.ldx16 / 64) {
        LSSVM_S6_CASE(1) LSSVM_S6_CASE(2) LSSVM_S6_CASE(3) LSSVM_S6_CASE(4) LSSVM_S6_CASE(5) LSSVM_S6_CASE(6)
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
                launch_s6_kt<KT_RBFF, SYM>(a, grid, s);  // only the 16x16x32 kernels understand the folded records (Problem sets the flag with mfma_shape >= 1)
            } else {
                launch_s6_kt<KT_RBF, SYM>(a, grid, s);
            }
            break;
    }
}

void launch_split_tile_kernel(const TileArgs<float> &a, int kernel_type, dim3 grid, hipStream_t s) {
    if (a.items != nullptr) {
        launch_s6<true>(a, kernel_type, dim3(static_cast<unsigned>(a.num_items)), s);
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

}  // namespace lssvm
