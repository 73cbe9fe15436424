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
    const size_t V2_LDS_BYTES = lssvm::V2_LDS_BYTES + static_cast<size_t>(options().lds_extra_kb) * 1024;  // experiment knob: limits workgroups per CU
    static size_t configured_for = 0;
    if (configured_for != V2_LDS_BYTES) {
        configured_for = V2_LDS_BYTES;
        ensure_dynamic_lds(tile_matvec_f32_v2<KT, 1, SYM>, V2_LDS_BYTES);
        ensure_dynamic_lds(tile_matvec_f32_v2<KT, 2, SYM>, V2_LDS_BYTES);
        ensure_dynamic_lds(tile_matvec_f32_v2<KT, 3, SYM>, V2_LDS_BYTES);
        ensure_dynamic_lds(tile_matvec_f32_v2<KT, 4, SYM>, V2_LDS_BYTES);
        ensure_dynamic_lds(tile_matvec_f32_v2<KT, 5, SYM>, V2_LDS_BYTES);
        ensure_dynamic_lds(tile_matvec_f32_v2<KT, 6, SYM>, V2_LDS_BYTES);
        ensure_dynamic_lds(tile_matvec_f32_v2<KT, 7, SYM>, V2_LDS_BYTES);
        ensure_dynamic_lds(tile_matvec_f32_v2<KT, 8, SYM>, V2_LDS_BYTES);
        ensure_dynamic_lds(tile_matvec_f32_v2<KT, 10, SYM>, V2_LDS_BYTES);
        ensure_dynamic_lds(tile_matvec_f32_v2<KT, 12, SYM>, V2_LDS_BYTES);
        ensure_dynamic_lds(tile_matvec_f32_v2<KT, 14, SYM>, V2_LDS_BYTES);
        ensure_dynamic_lds(tile_matvec_f32_v2<KT, 16, SYM>, V2_LDS_BYTES);
    }
    switch (a.kchunks) {
        case 1: hipLaunchKernelGGL((tile_matvec_f32_v2<KT, 1, SYM>), grid, block, V2_LDS_BYTES, s, a); break;
        case 2: hipLaunchKernelGGL((tile_matvec_f32_v2<KT, 2, SYM>), grid, block, V2_LDS_BYTES, s, a); break;
        case 3: hipLaunchKernelGGL((tile_matvec_f32_v2<KT, 3, SYM>), grid, block, V2_LDS_BYTES, s, a); break;
        case 4: hipLaunchKernelGGL((tile_matvec_f32_v2<KT, 4, SYM>), grid, block, V2_LDS_BYTES, s, a); break;
        case 5: hipLaunchKernelGGL((tile_matvec_f32_v2<KT, 5, SYM>), grid, block, V2_LDS_BYTES, s, a); break;
        case 6: hipLaunchKernelGGL((tile_matvec_f32_v2<KT, 6, SYM>), grid, block, V2_LDS_BYTES, s, a); break;
        case 7: hipLaunchKernelGGL((tile_matvec_f32_v2<KT, 7, SYM>), grid, block, V2_LDS_BYTES, s, a); break;
        case 8: hipLaunchKernelGGL((tile_matvec_f32_v2<KT, 8, SYM>), grid, block, V2_LDS_BYTES, s, a); break;
        case 10: hipLaunchKernelGGL((tile_matvec_f32_v2<KT, 10, SYM>), grid, block, V2_LDS_BYTES, s, a); break;
        case 12: hipLaunchKernelGGL((tile_matvec_f32_v2<KT, 12, SYM>), grid, block, V2_LDS_BYTES, s, a); break;
        case 14: hipLaunchKernelGGL((tile_matvec_f32_v2<KT, 14, SYM>), grid, block, V2_LDS_BYTES, s, a); break;
        case 16: hipLaunchKernelGGL((tile_matvec_f32_v2<KT, 16, SYM>), grid, block, V2_LDS_BYTES, s, a); break;
        default: throw Error(LSSVM_ERR_INTERNAL, "no v2 tile kernel for this number of k-chunks");
    }
}

template <>
void launch_tile_kernel<float>(TileArgs<float> &a, int kernel_type, bool rbf_direct, int num_jc, hipStream_t s) {
    const dim3 grid(a.num_ib > 0 && num_jc > 0 ? finish_mapping(a, num_jc) : 0u);
    const dim3 block(TILE_THREADS);
    if (grid.x == 0) return;
    constexpr size_t lds = static_cast<size_t>(4) * TILE * F32_LS * sizeof(float) + TILE * sizeof(float);  // staging ring + c_i of the row block
    static bool configured = false;
    if (!configured) {
        ensure_dynamic_lds(tile_matvec_f32<KT_LINEAR>, lds);
        ensure_dynamic_lds(tile_matvec_f32<KT_POLY>, lds);
        ensure_dynamic_lds(tile_matvec_f32<KT_RBF>, lds);
        configured = true;
    }
    if (a.dc != nullptr && a.Xc16 != nullptr) {  // option gram_mode = 1: the three-plane bf16 data exists
        launch_split_tile_kernel(a, kernel_type, grid, s);
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
        case KT_LINEAR: hipLaunchKernelGGL(tile_matvec_f32<KT_LINEAR>, grid, block, lds, s, a); break;
        case KT_POLY: hipLaunchKernelGGL(tile_matvec_f32<KT_POLY>, grid, block, lds, s, a); break;
        default:
            if (rbf_direct) {
                hipLaunchKernelGGL(tile_matvec_rbf_direct_f32, grid, block, 0, s, a);
            } else {
                hipLaunchKernelGGL(tile_matvec_f32<KT_RBF>, grid, block, lds, s, a);
            }
            break;
    }
    LSSVM_HIP_CHECK(hipGetLastError());
}

}  // namespace lssvm
