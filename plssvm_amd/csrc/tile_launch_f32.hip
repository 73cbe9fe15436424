/*
 * tile_launch_f32.hip -- instantiates and launches the fp32 tile kernels (launch_tile_kernel<float>, declared in
 * lssvm_problem.hip.hpp).  Compiled for gfx950 only.
 */
#include "tile_launch.hip.hpp"

#include "lssvm_tile_f32.hip.hpp"

namespace lssvm {

/* fp32 v2 kernel (row panel in registers, LDS-DMA ring): eligible for up to 16 k-chunks (num_features <= 512) */
template <int KT, bool SYM>
static void launch_v2_kt(const TileArgs<float> &a, dim3 grid, hipStream_t s) {
    const dim3 block(TILE_THREADS);
    const size_t V2_LDS_BYTES = lssvm::V2_LDS_BYTES + static_cast<size_t>(a.lds_extra_kb) * 1024;  // experiment knob: limits workgroups per CU
#define LSSVM_V2_CASE(N)                                                                                  \
    case N:                                                                                               \
        ensure_dynamic_lds(tile_matvec_f32_v2<KT, N, SYM>, V2_LDS_BYTES);                                 \
        hipLaunchKernelGGL((tile_matvec_f32_v2<KT, N, SYM>), grid, block, V2_LDS_BYTES, s, a);            \
        break;
    switch (a.kchunks) {
        LSSVM_V2_CASE(1) LSSVM_V2_CASE(2) LSSVM_V2_CASE(3) LSSVM_V2_CASE(4) LSSVM_V2_CASE(5) LSSVM_V2_CASE(6) LSSVM_V2_CASE(7) LSSVM_V2_CASE(8)
        LSSVM_V2_CASE(10) LSSVM_V2_CASE(12) LSSVM_V2_CASE(14) LSSVM_V2_CASE(16)
        default: throw Error(LSSVM_ERR_INTERNAL, "no v2 tile kernel for this number of k-chunks");
    }
#undef LSSVM_V2_CASE
}

template <>
void launch_tile_kernel<float>(TileArgs<float> &a, int kernel_type, bool rbf_direct, int num_jc, hipStream_t s) {
    const dim3 grid(a.num_ib > 0 && num_jc > 0 ? finish_mapping(a, num_jc) : 0u);
    const dim3 block(TILE_THREADS);
    if (grid.x == 0) return;
    constexpr size_t lds = static_cast<size_t>(4) * TILE * F32_LS * sizeof(float) + TILE * sizeof(float);  // staging ring + c_i of the row block
    if (a.dc != nullptr && a.Xc16 != nullptr) {  // the data exists as planes: two f16 planes (f16x3) or three bf16 planes (bf16x6)
        if (a.row_pair != 0) {
            launch_pair_tile_kernel(a, kernel_type, s);
        } else if (a.wide_panels != 0) {
            launch_wide_tile_kernel(a, kernel_type, grid, s);
        } else if (a.planes_f16 != 0) {
            launch_f16_tile_kernel(a, kernel_type, grid, s);
        } else {
            launch_split_tile_kernel(a, kernel_type, grid, s);
        }
        return;
    }
    if (a.dc != nullptr) {  // the records exist only where the v2 kernel was chosen when the data was prepared (v2_eligible)
        if (a.items != nullptr) {  // symmetric variant: one block per listed work item
            const dim3 sgrid(static_cast<unsigned>(a.num_items));
            switch (kernel_type) {
                case KT_LINEAR: launch_v2_kt<KT_LINEAR, true>(a, sgrid, s); break;
                case KT_POLY:
                    if (a.degree == 3) {
                        launch_v2_kt<KT_POLY3, true>(a, sgrid, s);
                    } else if (a.degree == 2) {
                        launch_v2_kt<KT_POLY2, true>(a, sgrid, s);
                    } else {
                        launch_v2_kt<KT_POLY, true>(a, sgrid, s);
                    }
                    break;
                default: launch_v2_kt<KT_RBF, true>(a, sgrid, s); break;
            }
        } else {
            switch (kernel_type) {
                case KT_LINEAR: launch_v2_kt<KT_LINEAR, false>(a, grid, s); break;
                case KT_POLY:
                    if (a.degree == 3) {
                        launch_v2_kt<KT_POLY3, false>(a, grid, s);
                    } else if (a.degree == 2) {
                        launch_v2_kt<KT_POLY2, false>(a, grid, s);
                    } else {
                        launch_v2_kt<KT_POLY, false>(a, grid, s);
                    }
                    break;
                default: launch_v2_kt<KT_RBF, false>(a, grid, s); break;
            }
        }
        LSSVM_HIP_CHECK(hipGetLastError());
        return;
    }
    switch (kernel_type) {
        case KT_LINEAR:
            ensure_dynamic_lds(tile_matvec_f32<KT_LINEAR>, lds);
            hipLaunchKernelGGL(tile_matvec_f32<KT_LINEAR>, grid, block, lds, s, a);
            break;
        case KT_POLY:
            ensure_dynamic_lds(tile_matvec_f32<KT_POLY>, lds);
            hipLaunchKernelGGL(tile_matvec_f32<KT_POLY>, grid, block, lds, s, a);
            break;
        default:
            if (rbf_direct) {
                hipLaunchKernelGGL(tile_matvec_rbf_direct_f32, grid, block, 0, s, a);
            } else {
                ensure_dynamic_lds(tile_matvec_f32<KT_RBF>, lds);
                hipLaunchKernelGGL(tile_matvec_f32<KT_RBF>, grid, block, lds, s, a);
            }
            break;
    }
    LSSVM_HIP_CHECK(hipGetLastError());
}

}  // namespace lssvm
