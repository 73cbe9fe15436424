/*
 * tile_launch_f32xd.hip -- instantiates and launches the panels-inside-a-tile kernels with 256-row workgroups (lssvm_tile_f32_wide_pair.hip.hpp:
 * rbf / polynomial on wide data, eight waves on a block pair, one column stream per CU).  Symmetric variant, both plane kinds.  Compiled for
 * gfx950 only.
 */
#include "tile_launch.hip.hpp"

#include "lssvm_tile_f32_wide_pair.hip.hpp"

namespace lssvm {

template <int KT>
static void launch_wide_pair_kt(const TileArgs<float> &a, hipStream_t s) {
    const dim3 grid(static_cast<unsigned>(a.num_items)), block(XP_THREADS);
    if (a.planes_f16 != 0) {
        ensure_dynamic_lds(tile_matvec_f32_wide_pair<KT, 2>, XP_LDS_BYTES);
        hipLaunchKernelGGL((tile_matvec_f32_wide_pair<KT, 2>), grid, block, XP_LDS_BYTES, s, a);
    } else {
        ensure_dynamic_lds(tile_matvec_f32_wide_pair<KT, 3>, XP_LDS_BYTES);
        hipLaunchKernelGGL((tile_matvec_f32_wide_pair<KT, 3>), grid, block, XP_LDS_BYTES, s, a);
    }
}

void launch_wide_pair_tile_kernel(const TileArgs<float> &a, int kernel_type, hipStream_t s) {
    if (a.items == nullptr || a.num_items <= 0) return;
    if (a.nk64 < 4 || a.nk64 % 2 != 0) throw Error(LSSVM_ERR_INTERNAL, "the wide split tile kernel needs planes padded to a multiple of 128 features");
    if (a.degree < 0 && kernel_type == KT_POLY) throw Error(LSSVM_ERR_INTERNAL, "the wide split tile kernel does not take a negative polynomial degree");
    switch (kernel_type) {
        case KT_POLY:
            if (a.degree == 3) {
                launch_wide_pair_kt<KT_POLY3>(a, s);
            } else if (a.degree == 2) {
                launch_wide_pair_kt<KT_POLY2>(a, s);
            } else {
                launch_wide_pair_kt<KT_POLY>(a, s);
            }
            break;
        case KT_RBF:
            if (a.dc_folded != 0) {
                launch_wide_pair_kt<KT_RBFF>(a, s);
            } else {
                launch_wide_pair_kt<KT_RBF>(a, s);
            }
            break;
        default: throw Error(LSSVM_ERR_INTERNAL, "the wide split tile kernel exists for the rbf and polynomial kernels");
    }
    LSSVM_HIP_CHECK(hipGetLastError());
}

}  // namespace lssvm
