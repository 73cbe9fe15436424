/*
 * plssvm_amd_testing.h -- entry points and option names of libplssvm_amd.so that are NOT part of the boundary a PLSSVM maintainer binds
 * (include/plssvm_amd.h): measurement and test aids used by bench.py, tests/ and tests/tools/ only.  They are exported by the same
 * library so that what is measured is the shipped code; nothing in include/plssvm_amd/csvm.hpp or plssvm_amd/csvm.py calls them.
 */
#ifndef PLSSVM_AMD_TESTING_H
#define PLSSVM_AMD_TESTING_H

#include "plssvm_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement utility, not on the solve path and without a counterpart in the reference: what a BARE loop of v_mfma_f32_16x16x32_bf16
 * (the instruction of the fp32 "bf16x6" Gram kernel; 64 x 64 wave tiles, two waves per SIMD, normal(0,1) operands, nothing else in the
 * loop) sustains on `device` after `settle_ms` of back-to-back launches.  b_from_lds bit 0: the B fragments are re-read from LDS every
 * pass, as the Gram kernel does; bit 1: v_mfma_f32_16x16x32_f16 on f16 operands (the instruction of the "f16x3" kernels) instead of the bf16 form.  Returns TFLOP/s, the in-kernel clock (s_memtime / s_memrealtime, median over workgroups) and the
 * nominal peak (4096 FLOP/clk/CU x CUs x nominal clock).  The chip lowers its clock under matrix-core load, so this -- not the
 * nominal peak -- is what a kernel on this device is up against; bench.py prints it beside roofline.frac. */
int lssvm_mi355_measure_bf16_mfma_ceiling(int device, int b_from_lds, double settle_ms, double *tflops_out, double *clock_ghz_out, double *nominal_tflops_out);

/* The file the library's RCCL entry points were resolved from (dladdr of ncclAllReduce after the lazy dlopen of "librccl.so.1"): a bench line or a
 * test can then say WHICH library carried the exchange -- the process's RCCL (PyTorch's or /opt/rocm's), or a stand-in with that SONAME which a test
 * harness loaded into the process first (tests/tools/; the library itself never looks for one). */
int lssvm_mi355_comm_library_path(char *buf, size_t buf_len);

/* The file readers and writers (lssvm_mi355_libsvm_*, _arff_*, _model_*) use at most this many host threads; 0 (default) = as many as the hardware has,
 * at most 32.  For the tests that a written file does not depend on the thread count. */
int lssvm_mi355_set_io_threads(int threads);

/* option names understood by lssvm_mi355_set_option / _get_option besides the fourteen documented in plssvm_amd.h:
 *   "force_collective" 1 = run the per-matvec RCCL collective even with a world of 1 (testing aid; default 0)
 *   "skip_collective"  1 = problems created with world > 1 need no communicator and do NOT exchange their partial K*v (testing aid:
 *                      lets one GPU evaluate every rank's share in turn; default 0)
 * and, accepted with the value 0 everywhere but effective in development builds only:
 *   "debug_ablate"  timing-only ablation bits of the fp32 tile kernels (-DLSSVM_ENABLE_ABLATION; results are wrong when != 0)
 *   "pair_lag"      256-row workgroups: plane-chunk steps waves 4-7 run behind waves 0-3 (make DEV=1: 0, 1, 3; the shipped library instantiates 0, lock step)
 */

#ifdef __cplusplus
}
#endif
#endif /* PLSSVM_AMD_TESTING_H */
