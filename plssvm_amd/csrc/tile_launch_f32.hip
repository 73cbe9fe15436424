/*
 * tile_launch_f32.hip -- instantiates and launches the fp32 tile kernels (launch_tile_kernel<float>, declared in
 * lssvm_problem.hip.hpp).  Compiled for gfx950 only.
 */
#include "tile_launch.hip.hpp"

#include "lssvm_tile_f32.hip.hpp"

/* compiled as TWO translation units: tile_launch_f32_sym.hip (LSSVM_TU_HALF 1: the symmetric instantiations of the native v2 kernel) and
 * tile_launch_f32_full.hip (LSSVM_TU_HALF 2: the full-square ones, the generic and the direct rbf kernel, the entry point) */
#ifndef LSSVM_TU_HALF
#error "compile tile_launch_f32_sym.hip / tile_launch_f32_full.hip"
#endif

namespace lssvm {

/* fp32 v2 kernel (row panel in registers, LDS-DMA ring): eligible for up to 16 k-chunks (num_features <= 512) */
template <int KT, bool SYM>
static void launch_v2_kt(const TileArgs<float> &a, dim3 grid, hipStream_t s) {
    const dim3 block(TILE_THREADS);
#define LSSVM_V2_CASE(N)                                                                                  \
    case N:                                                                                               \
        ensure_dynamic_lds(tile_matvec_f32_v2<KT, N, SYM>, V2_LDS_BYTES);                                 \
        hipLaunchKernelGGL((tile_matvec_f32_v2<KT, N, SYM>), grid, block, V2_LDS_BYTES, s, a);            \
        break;
    switch (a.kchunks) {
        // (1 ... 4 k-chunks of 32 features, then whole pairs: padded_features<float>.  Since round 3 this kernel runs where there are no operand planes
        // -- option gram_mode = 0, bench.py's native reference line, data that fails the f16 check on 385 ... 512 features -- and carries the
        // run-time integer power only: the degree-2 / -3 specialisations went with the other instantiations no default path reaches)
        LSSVM_V2_CASE(1) LSSVM_V2_CASE(2) LSSVM_V2_CASE(3) LSSVM_V2_CASE(4) LSSVM_V2_CASE(6) LSSVM_V2_CASE(8)
        LSSVM_V2_CASE(10) LSSVM_V2_CASE(12) LSSVM_V2_CASE(14) LSSVM_V2_CASE(16)
        default: throw Error(LSSVM_ERR_INTERNAL, "no v2 tile kernel for this number of k-chunks");
    }
#undef LSSVM_V2_CASE
}

void launch_v2_sym(const TileArgs<float> &a, int kernel_type, hipStream_t s);  // tile_launch_f32_sym.hip

#if LSSVM_TU_HALF == 1
void launch_v2_sym(const TileArgs<float> &a, int kernel_type, hipStream_t s) {
    const dim3 sgrid(static_cast<unsigned>(a.num_items));
    switch (kernel_type) {
        case KT_LINEAR: launch_v2_kt<KT_LINEAR, true>(a, sgrid, s); break;
        case KT_POLY: launch_v2_kt<KT_POLY, true>(a, sgrid, s); break;
        default: launch_v2_kt<KT_RBF, true>(a, sgrid, s); break;
    }
}
#else
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
            launch_v2_sym(a, kernel_type, s);
        } else {
            switch (kernel_type) {
                case KT_LINEAR: launch_v2_kt<KT_LINEAR, false>(a, grid, s); break;
                case KT_POLY: launch_v2_kt<KT_POLY, false>(a, grid, s); break;
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
                hipLaunchKernelGGL(tile_matvec_rbf_direct_f32<true>, grid, block, 0, s, a);
            } else {
                ensure_dynamic_lds(tile_matvec_f32<KT_RBF>, lds);
                hipLaunchKernelGGL(tile_matvec_f32<KT_RBF>, grid, block, lds, s, a);
            }
            break;
    }
    LSSVM_HIP_CHECK(hipGetLastError());
}
#endif

}  // namespace lssvm
