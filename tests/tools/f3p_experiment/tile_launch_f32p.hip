/*
 * tile_launch_f32p.hip -- instantiates and launches the software-pipelined f16x3 tile kernel (lssvm_tile_f32_pipe.hip.hpp; option
 * mfma_shape = 3).  A translation unit of its own: its generated asm statements compile in seconds.  Compiled for gfx950 only.
 */
#include "tile_launch.hip.hpp"

#include "lssvm_tile_f32_pipe.hip.hpp"

namespace lssvm {

/* true if a pipelined kernel exists for this launch (rbf with folded records, 65 ... 128 features, symmetric variant) and has been launched */
bool launch_f3p_tile_kernel(const TileArgs<float> &a, int kernel_type, hipStream_t s) {
    if (a.planes_f16 == 0 || a.items == nullptr || kernel_type != KT_RBF || a.dc_folded == 0 || a.ldx16 != 128 || a.nk64 != 2 || a.num_items <= 0) return false;
    const size_t lds_bytes = V2_LDS_BYTES;
    ensure_dynamic_lds(tile_matvec_f32_f3p_rbff_k2_sym, lds_bytes);
    hipLaunchKernelGGL(tile_matvec_f32_f3p_rbff_k2_sym, dim3(static_cast<unsigned>(a.num_items)), dim3(TILE_THREADS), lds_bytes, s, a);
    LSSVM_HIP_CHECK(hipGetLastError());
    return true;
}

}  // namespace lssvm
