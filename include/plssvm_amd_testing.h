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

/* ---- EXPERIMENTAL: shares by weight for devices of unequal pace ----
 * Not part of the boundary: built in round 5 for hardware nobody has measured yet (two DIFFERENT devices behind one solve), exercised for self-consistency on one
 * time-shared GPU only, where "pace" is the scheduler's and not a device's.  Frozen until a multi-GPU node exists; the names may change. */
/* Devices of unequal pace (the MI355X boxes of one pool run the same kernel in 252 ... 277 ms): rank r of a sharded SYMMETRIC problem gets weights[r] / sum of the triangle's area
 * instead of 1 / world -- a process-wide default like the options below, snapshotted when a problem is created, applied when `count` equals the problem's world (any other
 * length, or count = 0: equal shares).  EVERY rank of a sharded solve must set the same weights (the partition is computed locally by every rank); the data is replicated on
 * every device, so a new partition costs a new problem, no data exchange.  bench.py --balance-shares measures the ranks' pace and sets them.  No counterpart in the reference
 * (its multi-device split is by features, gpu_csvm.hpp:283-299). */
int lssvm_mi355_set_shard_weights(const double *weights, int count);
/* The same for a LIVE problem, between two lssvm_mi355_cg_step calls: the shards' row blocks, work items and slabs are rebuilt for new shares; the data, the vectors and the
 * CG state stay (the implicit matrix does not change, only who evaluates which tiles).  weights == NULL, count == 0: shares by MEASURED pace -- every shard's tile-kernel time
 * per matvec so far against the area of its share; one process driving all devices knows them, one process per GPU gathers them over the library's RCCL communicator (every rank
 * must make the call; not over HIP IPC: explicit weights there, the same on every rank).  *changed_out = 0 where the times lie within 2 % of each other or the problem is not
 * sharded / not symmetric.  A solve that wants it: cg_begin, a few cg_step, problem_rebalance, the remaining cg_step. */
int lssvm_mi355_problem_rebalance(lssvm_mi355_problem *p, const double *weights, int count, int *changed_out);
/* option (lssvm_mi355_set_option / lssvm_mi355_options_set), experimental with the two entry points above:
 *   "rebalance_after" lssvm_mi355_solve_multi_*: after this many CG iterations the shards get new shares of the triangle by their measured pace
 *                   (lssvm_mi355_problem_rebalance with weights = NULL); 0 (default) = never -- devices are taken to run at one pace
 */

/* The file readers and writers (lssvm_mi355_libsvm_*, _arff_*, _model_*) use at most this many host threads; 0 (default) = as many as the hardware has,
 * at most 32.  For the tests that a written file does not depend on the thread count. */
int lssvm_mi355_set_io_threads(int threads);

/* option names understood by lssvm_mi355_set_option / _get_option besides the fourteen documented in plssvm_amd.h:
 *   "force_collective" 1 = run the per-matvec RCCL collective even with a world of 1 (testing aid; default 0)
 *   "skip_collective"  1 = problems created with world > 1 need no communicator and do NOT exchange their partial K*v (testing aid:
 *                      lets one GPU evaluate every rank's share in turn; default 0)
 *   ("rebalance_after": above)
 * and, accepted with the value 0 everywhere but effective in development builds only:
 *   "debug_ablate"  timing-only ablation bits of the fp32 tile kernels (-DLSSVM_ENABLE_ABLATION; results are wrong when != 0)
 *   "pair_lag"      256-row workgroups: plane-chunk steps waves 4-7 run behind waves 0-3 (make DEV=1: 0, 1, 3; the shipped library instantiates 0, lock step)
 */

#ifdef __cplusplus
}
#endif
#endif /* PLSSVM_AMD_TESTING_H */
