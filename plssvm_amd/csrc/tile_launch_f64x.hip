/*
 * tile_launch_f64x.hip -- instantiates and launches the fp64 tile kernel for rbf / polynomial problems with more than 256 features
 * (lssvm_tile_f64_wide.hip.hpp: feature panels of 64 walked inside a sub-tile).  Compiled for gfx950 only.
 */
#include "tile_launch.hip.hpp"

#include "lssvm_tile_f64_wide.hip.hpp"

namespace lssvm {

static_assert(V2D_LDS_BYTES <= 64 * 1024, "the launches below do not opt into more dynamic LDS than the default limit");

template <bool SYM>
static void launch_wide_f64(const TileArgs<double> &a, int kernel_type, dim3 grid, hipStream_t s) {
    const dim3 block(TILE_THREADS);
    switch (kernel_type) {
        case KT_POLY:
            if (a.degree == 3) {
                hipLaunchKernelGGL((tile_matvec_f64_wide<KT_POLY3, SYM>), grid, block, V2D_LDS_BYTES, s, a);
            } else if (a.degree == 2) {
                hipLaunchKernelGGL((tile_matvec_f64_wide<KT_POLY2, SYM>), grid, block, V2D_LDS_BYTES, s, a);
            } else {
                hipLaunchKernelGGL((tile_matvec_f64_wide<KT_POLY, SYM>), grid, block, V2D_LDS_BYTES, s, a);
            }
            break;
        case KT_RBF: hipLaunchKernelGGL((tile_matvec_f64_wide<KT_RBF, SYM>), grid, block, V2D_LDS_BYTES, s, a); break;
        default: throw Error(LSSVM_ERR_INTERNAL, "the wide fp64 tile kernel exists for the rbf and polynomial kernels");
    }
}

/* `grid` is used by the full-square variant only (the symmetric variant runs one workgroup per listed work item) */
void launch_wide_tile_kernel_f64(const TileArgs<double> &a, int kernel_type, dim3 grid, hipStream_t s) {
    if (a.kchunks < 8 || a.kchunks % 4 != 0) throw Error(LSSVM_ERR_INTERNAL, "the wide fp64 tile kernel needs data padded to a multiple of 64 features");
    if (a.degree < 0 && kernel_type == KT_POLY) throw Error(LSSVM_ERR_INTERNAL, "the wide fp64 tile kernel does not take a negative polynomial degree");
    if (a.items != nullptr) {
        if (a.Xrf == nullptr || a.frag_rows16 <= 0) throw Error(LSSVM_ERR_INTERNAL, "the symmetric wide fp64 tile kernel needs the fragment-major rows");
        if (a.num_items > 0) launch_wide_f64<true>(a, kernel_type, dim3(static_cast<unsigned>(a.num_items)), s);
    } else {
        launch_wide_f64<false>(a, kernel_type, grid, s);
    }
    LSSVM_HIP_CHECK(hipGetLastError());  // (a failed launch surfaces HERE, not at an unrelated later call)
}

}  // namespace lssvm
