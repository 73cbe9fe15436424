/*
 * tile_launch_f64.hip -- instantiates and launches the fp64 tile kernels (launch_tile_kernel<double>, declared in
 * lssvm_problem.hip.hpp).  Compiled for gfx950 only.
 */
#include "tile_launch.hip.hpp"

#include "lssvm_tile_f64.hip.hpp"

/* This source is compiled as TWO translation units (tile_launch_f64_sym.hip: LSSVM_TU_HALF 1 = the symmetric instantiations, tile_launch_f64_full.hip:
 * LSSVM_TU_HALF 2 = the full-square ones, the generic kernel and the entry point), so that the build spreads over more cores. */
#ifndef LSSVM_TU_HALF
#error "compile tile_launch_f64_sym.hip / tile_launch_f64_full.hip"
#endif

namespace lssvm {

template <int KT, bool SYM>
static void launch_v2d_kt(const TileArgs<double> &a, dim3 grid, hipStream_t s) {
    const dim3 block(TILE_THREADS);
    switch (a.kchunks) {
        case 1: hipLaunchKernelGGL((tile_matvec_f64_v2<KT, 1, SYM>), grid, block, V2D_LDS_BYTES, s, a); break;
        case 2: hipLaunchKernelGGL((tile_matvec_f64_v2<KT, 2, SYM>), grid, block, V2D_LDS_BYTES, s, a); break;
        case 3: hipLaunchKernelGGL((tile_matvec_f64_v2<KT, 3, SYM>), grid, block, V2D_LDS_BYTES, s, a); break;
        case 4: hipLaunchKernelGGL((tile_matvec_f64_v2<KT, 4, SYM>), grid, block, V2D_LDS_BYTES, s, a); break;
        case 5: hipLaunchKernelGGL((tile_matvec_f64_v2<KT, 5, SYM>), grid, block, V2D_LDS_BYTES, s, a); break;
        case 6: hipLaunchKernelGGL((tile_matvec_f64_v2<KT, 6, SYM>), grid, block, V2D_LDS_BYTES, s, a); break;
        case 7: hipLaunchKernelGGL((tile_matvec_f64_v2<KT, 7, SYM>), grid, block, V2D_LDS_BYTES, s, a); break;
        case 8: hipLaunchKernelGGL((tile_matvec_f64_v2<KT, 8, SYM>), grid, block, V2D_LDS_BYTES, s, a); break;
        case 10: hipLaunchKernelGGL((tile_matvec_f64_v2<KT, 10, SYM>), grid, block, V2D_LDS_BYTES, s, a); break;
        case 12: hipLaunchKernelGGL((tile_matvec_f64_v2<KT, 12, SYM>), grid, block, V2D_LDS_BYTES, s, a); break;
        case 14: hipLaunchKernelGGL((tile_matvec_f64_v2<KT, 14, SYM>), grid, block, V2D_LDS_BYTES, s, a); break;
        case 16: hipLaunchKernelGGL((tile_matvec_f64_v2<KT, 16, SYM>), grid, block, V2D_LDS_BYTES, s, a); break;
        default: throw Error(LSSVM_ERR_INTERNAL, "no v2 tile kernel for this number of k-chunks");
    }
}

void launch_v2d_sym(const TileArgs<double> &a, int kernel_type, hipStream_t s);  // tile_launch_f64_sym.hip

#if LSSVM_TU_HALF == 1
void launch_v2d_sym(const TileArgs<double> &a, int kernel_type, hipStream_t s) {
    const dim3 sgrid(static_cast<unsigned>(a.num_items));
    switch (kernel_type) {
        case KT_LINEAR: launch_v2d_kt<KT_LINEAR, true>(a, sgrid, s); break;
        case KT_POLY:
            if (a.degree == 3) {
                launch_v2d_kt<KT_POLY3, true>(a, sgrid, s);
            } else if (a.degree == 2) {
                launch_v2d_kt<KT_POLY2, true>(a, sgrid, s);
            } else {
                launch_v2d_kt<KT_POLY, true>(a, sgrid, s);
            }
            break;
        default: launch_v2d_kt<KT_RBF, true>(a, sgrid, s); break;
    }
}
#else
template <>
void launch_tile_kernel<double>(TileArgs<double> &a, int kernel_type, bool /*rbf_direct*/, int num_jc, hipStream_t s) {
    const dim3 grid(a.num_ib > 0 && num_jc > 0 ? finish_mapping(a, num_jc) : 0u);
    const dim3 block(TILE_THREADS);
    if (grid.x == 0) return;
    constexpr size_t lds = static_cast<size_t>(4) * TILE * F64_LS * sizeof(double);
    if (a.dc != nullptr && a.wide_panels != 0) {
        launch_wide_tile_kernel_f64(a, kernel_type, grid, s);
        return;
    }
    if (a.dc != nullptr) {  // the records exist only where the v2 kernel was chosen when the data was prepared; V2D_LDS_BYTES < 64 KiB
        if (a.items != nullptr) {
            launch_v2d_sym(a, kernel_type, s);
        } else {
            switch (kernel_type) {
                case KT_LINEAR: launch_v2d_kt<KT_LINEAR, false>(a, grid, s); break;
                case KT_POLY:
                    if (a.degree == 3) {
                        launch_v2d_kt<KT_POLY3, false>(a, grid, s);
                    } else if (a.degree == 2) {
                        launch_v2d_kt<KT_POLY2, false>(a, grid, s);
                    } else {
                        launch_v2d_kt<KT_POLY, false>(a, grid, s);
                    }
                    break;
                default: launch_v2d_kt<KT_RBF, false>(a, grid, s); break;
            }
        }
        LSSVM_HIP_CHECK(hipGetLastError());
        return;
    }
    switch (kernel_type) {
        case KT_LINEAR:
            ensure_dynamic_lds(tile_matvec_f64<KT_LINEAR>, lds);
            hipLaunchKernelGGL(tile_matvec_f64<KT_LINEAR>, grid, block, lds, s, a);
            break;
        case KT_POLY:
            ensure_dynamic_lds(tile_matvec_f64<KT_POLY>, lds);
            hipLaunchKernelGGL(tile_matvec_f64<KT_POLY>, grid, block, lds, s, a);
            break;
        default:
            ensure_dynamic_lds(tile_matvec_f64<KT_RBF>, lds);
            hipLaunchKernelGGL(tile_matvec_f64<KT_RBF>, grid, block, lds, s, a);
            break;
    }
    LSSVM_HIP_CHECK(hipGetLastError());
}
#endif

}  // namespace lssvm
