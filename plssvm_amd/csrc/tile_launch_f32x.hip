/*
 * tile_launch_f32x.hip -- instantiates and launches the fp32 split tile kernels for rbf / polynomial problems whose feature count exceeds the
 * row panel a wave can hold in registers (lssvm_tile_f32_wide.hip.hpp: feature panels of 128 walked inside a tile).  Compiled for gfx950 only.
 */
#include "tile_launch.hip.hpp"

#include "lssvm_tile_f32_wide.hip.hpp"

namespace lssvm {

template <int KT>
static void launch_wide_kt(const TileArgs<float> &a, hipStream_t s) {
    const dim3 grid(static_cast<unsigned>(a.num_items)), block(TILE_THREADS);
    const size_t lds = lssvm::V2_LDS_BYTES + static_cast<size_t>(a.lds_extra_kb) * 1024;
    if (a.planes_f16 != 0) {
        ensure_dynamic_lds(tile_matvec_f32_wide<KT, 2>, lds);
        hipLaunchKernelGGL((tile_matvec_f32_wide<KT, 2>), grid, block, lds, s, a);
    } else {
        ensure_dynamic_lds(tile_matvec_f32_wide<KT, 3>, lds);
        hipLaunchKernelGGL((tile_matvec_f32_wide<KT, 3>), grid, block, lds, s, a);
    }
}

void launch_wide_tile_kernel(const TileArgs<float> &a, int kernel_type, hipStream_t s) {
    if (a.items == nullptr || a.nk64 < 4 || a.nk64 % 2 != 0) throw Error(LSSVM_ERR_INTERNAL, "the wide split tile kernel needs the symmetric variant and planes padded to a multiple of 128 features");
    if (a.num_items <= 0) return;
    switch (kernel_type) {
        case KT_POLY:
            if (a.degree == 3) {
                launch_wide_kt<KT_POLY3>(a, s);
            } else if (a.degree == 2) {
                launch_wide_kt<KT_POLY2>(a, s);
            } else {
                launch_wide_kt<KT_POLY>(a, s);
            }
            break;
        case KT_RBF:
            if (a.dc_folded != 0) {
                launch_wide_kt<KT_RBFF>(a, s);
            } else {
                launch_wide_kt<KT_RBF>(a, s);
            }
            break;
        default: throw Error(LSSVM_ERR_INTERNAL, "the wide split tile kernel exists for the rbf and polynomial kernels");
    }
}

}  // namespace lssvm
