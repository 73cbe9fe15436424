/*
 * tile_launch_f32x.hip -- instantiates and launches the fp32 split tile kernels for rbf / polynomial problems whose feature count exceeds the
 * row panel a wave can hold in registers (lssvm_tile_f32_wide.hip.hpp: feature panels of 128 walked inside a tile).  Compiled for gfx950 only.
 */
#include "tile_launch.hip.hpp"

#include "lssvm_tile_f32_wide.hip.hpp"

namespace lssvm {

template <int KT, bool SYM>
static void launch_wide_kt(const TileArgs<float> &a, dim3 grid, hipStream_t s) {
    const dim3 block(TILE_THREADS);
    const size_t lds = lssvm::V2_LDS_BYTES;
    if (a.planes_f16 != 0) {
        ensure_dynamic_lds(tile_matvec_f32_wide<KT, 2, SYM>, lds);
        hipLaunchKernelGGL((tile_matvec_f32_wide<KT, 2, SYM>), grid, block, lds, s, a);
    } else {
        ensure_dynamic_lds(tile_matvec_f32_wide<KT, 3, SYM>, lds);
        hipLaunchKernelGGL((tile_matvec_f32_wide<KT, 3, SYM>), grid, block, lds, s, a);
    }
}

template <bool SYM>
static void launch_wide(const TileArgs<float> &a, int kernel_type, dim3 grid, hipStream_t s) {
    switch (kernel_type) {
        case KT_POLY:
            if (a.degree == 3) {
                launch_wide_kt<KT_POLY3, SYM>(a, grid, s);
            } else if (a.degree == 2) {
                launch_wide_kt<KT_POLY2, SYM>(a, grid, s);
            } else {
                launch_wide_kt<KT_POLY, SYM>(a, grid, s);
            }
            break;
        case KT_RBF:
            if (a.rbf_grid != 0) {  // rbf on grid planes (KT_RBFG): f16 planes only
                const dim3 block(TILE_THREADS);
                if (a.planes_f16 == 0) throw Error(LSSVM_ERR_INTERNAL, "the grid-plane rbf kernel runs on f16 planes");
                ensure_dynamic_lds(tile_matvec_f32_wide<KT_RBFG, 2, SYM>, lssvm::V2_LDS_BYTES);
                hipLaunchKernelGGL((tile_matvec_f32_wide<KT_RBFG, 2, SYM>), grid, block, lssvm::V2_LDS_BYTES, s, a);
            } else if (a.dc_folded != 0) {
                launch_wide_kt<KT_RBFF, SYM>(a, grid, s);
            } else {
                launch_wide_kt<KT_RBF, SYM>(a, grid, s);
            }
            break;
        default: throw Error(LSSVM_ERR_INTERNAL, "the wide split tile kernel exists for the rbf and polynomial kernels");
    }
}

/* `grid` is used by the full-square variant only (the symmetric variant runs one workgroup per listed work item) */
void launch_wide_tile_kernel(const TileArgs<float> &a, int kernel_type, dim3 grid, hipStream_t s) {
    if (a.nk64 < 4 || a.nk64 % 2 != 0) throw Error(LSSVM_ERR_INTERNAL, "the wide split tile kernel needs planes padded to a multiple of 128 features");
    if (a.degree < 0 && kernel_type == KT_POLY) throw Error(LSSVM_ERR_INTERNAL, "the wide split tile kernel does not take a negative polynomial degree");
    if (a.items != nullptr) {
        if (a.Xr16f == nullptr) throw Error(LSSVM_ERR_INTERNAL, "the symmetric wide split tile kernel needs the fragment-major row planes");
        if (a.num_items > 0) launch_wide<true>(a, kernel_type, dim3(static_cast<unsigned>(a.num_items)), s);
    } else {
        launch_wide<false>(a, kernel_type, grid, s);
    }
    LSSVM_HIP_CHECK(hipGetLastError());  // (a failed launch surfaces HERE, not at an unrelated later call)
}

}  // namespace lssvm
