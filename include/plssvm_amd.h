/*
 * plssvm_amd.h -- C ABI of the MI355X-native LS-SVM Conjugate-Gradient backend (libplssvm_amd.so).
 *
 * This is the drop-in boundary for ONE path of SC-SGS/PLSSVM: the CG solve behind
 * plssvm::csvm::solve_system_of_linear_equations and the kernels it is built from.  Plain C: pointers + sizes only,
 * no C++ or torch types.  The C++ adaptor with the reference's virtual signatures lives in
 * include/plssvm_amd/csvm.hpp, the Python mirror of the reference's bindings in plssvm_amd/ (both sit ABOVE this ABI).
 *
 * Conventions
 *   - every function returns LSSVM_SUCCESS (0) or a negative lssvm_status; lssvm_mi355_last_error() returns the
 *     thread-local message of the last failure (the C++ adaptor rethrows it as plssvm::mi355::backend_exception,
 *     the counterpart of plssvm::hip::backend_exception, include/plssvm/backends/HIP/exceptions.hpp);
 *   - the caller owns every host pointer; data matrices are dense ROW-MAJOR `num_points x num_features`
 *     (the reference's std::vector<std::vector<T>> flattened, include/plssvm/csvm.hpp:188);
 *   - n = num_points - 1 is the size of the reduced system ("dept" in the reference);
 *   - `_f32` / `_f64` twins mirror the reference's float/double overload pairs (csvm.hpp:188-208);
 *   - kernel arithmetic runs ONLY on the GPU: there is no CPU fallback, a missing device is LSSVM_ERR_NO_DEVICE.
 *
 * All `file:line` citations are relative to the reference tree (SC-SGS/PLSSVM v2.0.0).
 */
#ifndef PLSSVM_AMD_H
#define PLSSVM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLSSVM_AMD_ABI_VERSION 4 /* 2: multi-device entry points (_multi), lssvm_cg_info grew local_devices / exchange; 3: lssvm_cg_info grew matvec_timed /
                                  * matvec_kernel_ms_total / rccl_nranks / rccl_rank / rccl_device / persistent_launches; 4: PER-CALL OPTIONS -- every entry point
                                  * that creates a problem takes a trailing `const lssvm_mi355_options *` (NULL = the process defaults), predict_values reports
                                  * lssvm_predict_info, lssvm_cg_info grew f16_row_rel_error, model / data file writers and the model reader, the shard-weight
                                  * entry points moved to plssvm_amd_testing.h (experimental) */

typedef enum lssvm_status {
    LSSVM_SUCCESS = 0,
    LSSVM_ERR_INVALID_ARGUMENT = -1, /* a violated precondition (the reference's PLSSVM_ASSERTs, csvm.cpp:73-78, svm_kernel.cpp:24-28) */
    LSSVM_ERR_NO_DEVICE = -2,        /* "HIP backend selected but no HIP capable devices were found!" (csvm.hip.cpp:70-72) */
    LSSVM_ERR_HIP = -3,              /* a HIP runtime call failed (PLSSVM_HIP_ERROR_CHECK, utility.hip.cpp:19-23) */
    LSSVM_ERR_COMM = -4,             /* RCCL missing or a collective failed */
    LSSVM_ERR_OUT_OF_MEMORY = -5,
    LSSVM_ERR_INTERNAL = -6
} lssvm_status;

/* plssvm::kernel_function_type (include/plssvm/kernel_function_types.hpp:31-38) */
typedef enum lssvm_kernel_type {
    LSSVM_KERNEL_LINEAR = 0,     /* u . v */
    LSSVM_KERNEL_POLYNOMIAL = 1, /* (gamma * u . v + coef0)^degree */
    LSSVM_KERNEL_RBF = 2         /* exp(-gamma * |u - v|^2) */
} lssvm_kernel_type;

/* plssvm::detail::parameter<T> (include/plssvm/parameter.hpp:156-165) with gamma ALREADY resolved
 * (the 1 / num_features default is applied by the caller exactly as csvm::fit does, csvm.hpp:303-307). */
typedef struct lssvm_params {
    int32_t kernel_type; /* lssvm_kernel_type */
    int32_t degree;      /* polynomial only */
    double gamma;        /* polynomial, rbf: must be > 0 (svm_kernel.cpp:68, :77) */
    double coef0;        /* polynomial only */
    double cost;         /* C; must be != 0 (svm_kernel.cpp:27 checks 1/C) */
} lssvm_params;

/* What the reference logs / tracks for a solve: keys cg/iterations, cg/max_iterations, cg/residuum, cg/target_residuum,
 * cg/avg_iteration_time, cg/epsilon, cg/total_runtime (csvm.cpp:167-176, csvm.hpp:318-320), plus device timings. */
typedef struct lssvm_cg_info {
    uint64_t iterations;     /* min(iter + 1, max_iter) */
    uint64_t max_iterations;
    double residuum;         /* delta = r^T r after the last iteration */
    double initial_residuum; /* delta0 */
    double target_residuum;  /* eps * eps * delta0 */
    double epsilon;
    double avg_iteration_ms; /* host wall clock per CG iteration */
    double total_ms;         /* host wall clock of begin + all steps + finish */
    double setup_ms;         /* host->device transfer of the data matrix, q vector, norms */
    double matvec_kernel_ms; /* average device time of the tile-kernel launches of ONE implicit matvec (HIP events on the solver stream): matvec_kernel_ms_total / matvec_timed */
    uint64_t matvec_launches; /* implicit matvecs enqueued since cg_begin */
    int32_t devices_used;    /* world size of the row-block sharding (1 = single GPU) */
    int32_t converged;       /* 1 if the stop test delta <= eps^2 * delta0 fired */
    int32_t symmetric;       /* 1 if the implicit matvec evaluated only the tiles on/below the diagonal (half the multiply-adds) */
    int32_t gram_mode;       /* fp32: how the Gram tiles ran: 0 = native v_mfma_f32 chains, 1 = "bf16x6" (exact 3-way bf16 split), 2 = "f16x3" (two f16 planes per operand),
                              * 3 = rbf on f16 GRID planes (three planes, six products: option rbf_form) */
    int32_t local_devices;   /* devices driven by THIS process (1 for a single GPU and for one process per GPU) */
    int32_t exchange;        /* how the partial K*v vectors were combined per matvec: 0 none, 1 RCCL (all-reduce / all-gather), 2 peer kernels over xGMI */
    int32_t tile_launches_per_matvec; /* tile-kernel launches per implicit matvec: row-block bands (option colslab_band_mb) x feature panels of a wide linear problem; matvec_kernel_ms is their SUM */
    int32_t rbf_direct;      /* fp32 rbf: 1 if the formula-exact (x_i - x_j)^2 kernel ran instead of the matrix-core norm expansion (option rbf_form) */
    double rbf_exponent_scale; /* fp32 rbf, rbf_form 0: 2 gamma log2(e) max|x - mean|^2, the quantity compared with rbf_direct_above */
    uint64_t matvec_timed;   /* of matvec_launches, the matvecs whose tile-kernel launches were bracketed by HIP events AND have been read back (short matvecs
                              * are sampled: launches 1 ... 4 and then every 8th, never the first after cg_begin) */
    double matvec_kernel_ms_total; /* summed device time of the tile-kernel launches of those matvec_timed matvecs (slowest shard of this process): differences of
                                    * (matvec_kernel_ms_total, matvec_timed) between two lssvm_mi355_problem_info calls give the average over the steps between them */
    int32_t rccl_nranks;     /* exchange == 1: ncclCommCount of the communicator the partial vectors travel over (what RCCL itself says, not what was asked for); else 0 */
    int32_t rccl_rank;       /* exchange == 1: ncclCommUserRank of this process's (first) communicator; else -1 */
    int32_t rccl_device;     /* exchange == 1: ncclCommCuDevice of that communicator; else -1 */
    int32_t persistent_launches; /* of tile_launches_per_matvec, the launches that are PERSISTENT: one workgroup per CU, the work items drawn from per-XCD counters instead of one
                                  * workgroup per item dealt by the hardware (256-row workgroups, launches of more items than CUs) */
    double f16_row_rel_error; /* fp32, gram_mode 2 / 3: what the set-up MEASURED on this data -- the largest relative error (2-norm) of a row represented as two f16 planes;
                               * mode 3 takes "f16x3" up to 2^-22 (rbf: or an absolute bound on the exponent), else "bf16x6".  -1 where the check did not run, NaN for a plane that overflows */
} lssvm_cg_info;

/* What a predict_values call reports (the reference tracks the whole call, gpu_csvm.hpp:656-730). */
typedef struct lssvm_predict_info {
    double total_ms;   /* host wall clock of the call: uploads, data preparation, the product, the read-back */
    double setup_ms;   /* ... of which before the product kernel was enqueued (uploads of both point sets, centring, norms, operand planes) */
    double kernel_ms;  /* device time of the kernel that evaluates the product (HIP events): the rectangular tile kernel, or w.x for the linear kernel */
    double rbf_exponent_scale; /* as lssvm_cg_info */
    double f16_row_rel_error;  /* as lssvm_cg_info, over support vectors and points */
    int32_t gram_mode;         /* as lssvm_cg_info */
    int32_t rbf_direct;        /* as lssvm_cg_info */
    int32_t resident;          /* lssvm_mi355_predictor_predict: 1 if the batch ran against the RESIDENT support vectors, 0 if it took the one-shot path (same result) */
    int32_t reserved;
} lssvm_predict_info;

/* ------------------------------------------------------------------------------------------------------------------ */
/* library / device queries                                                                                           */
/* ------------------------------------------------------------------------------------------------------------------ */
int lssvm_mi355_abi_version(void);
/* number of visible HIP devices (hip::detail::get_device_count, csvm.hip.cpp:66); 0 if none, <0 on error */
int lssvm_mi355_device_count(void);
/* writes "name (gcnArchName), CUs" of `device` into buf (csvm.hip.cpp:77-81 logs the same properties) */
int lssvm_mi355_device_name(int device, char *buf, size_t buf_len);
const char *lssvm_mi355_last_error(void);

/* ------------------------------------------------------------------------------------------------------------------ */
/* options of ONE caller (ABI 4)                                                                                       */
/* ------------------------------------------------------------------------------------------------------------------ */
/* The tuning knobs listed at the end of this header, held by the caller instead of the process: a plssvm::csvm object is a move-only object whose const
 * virtuals share no state with other objects beyond `verbosity` (include/plssvm/csvm.hpp:50-83) -- two backend objects with different settings must be able to
 * solve at the same time from two threads.  `create` copies the process-wide defaults of that moment (lssvm_mi355_set_option / LSSVM_MI355_OPTIONS); `set` / `get`
 * take the names and ranges of lssvm_mi355_set_option.  Every entry point below that creates a problem takes a trailing `const lssvm_mi355_options *`:
 * NULL = a snapshot of the process defaults (the behaviour of ABI 3); the object is read during the call only and may be changed or destroyed afterwards. */
typedef struct lssvm_mi355_options lssvm_mi355_options; /* opaque */
int lssvm_mi355_options_create(lssvm_mi355_options **out);
int lssvm_mi355_options_set(lssvm_mi355_options *options, const char *name, int64_t value);
int lssvm_mi355_options_get(const lssvm_mi355_options *options, const char *name, int64_t *value_out);
int lssvm_mi355_options_destroy(lssvm_mi355_options *options);

/* ------------------------------------------------------------------------------------------------------------------ */
/* one-shot entry points: exactly what plssvm::csvm's pure virtuals need (include/plssvm/csvm.hpp:188-208)           */
/* ------------------------------------------------------------------------------------------------------------------ */

/* csvm::solve_system_of_linear_equations (csvm.hpp:188, :192; recipe: backends/OpenMP/csvm.cpp:71-183).
 * X: N x d row-major, y: N labels (+-1), alpha_out: N entries (alpha[N-1] = -sum(alpha[0..N-1))), rho_out = -bias.
 * Runs on device 0 of the calling process.  info may be NULL. */
int lssvm_mi355_solve_f32(const lssvm_params *params, const float *X, size_t num_points, size_t num_features, const float *y,
                          float eps, uint64_t max_iter, float *alpha_out, float *rho_out, lssvm_cg_info *info, const lssvm_mi355_options *options);
int lssvm_mi355_solve_f64(const lssvm_params *params, const double *X, size_t num_points, size_t num_features, const double *y,
                          double eps, uint64_t max_iter, double *alpha_out, double *rho_out, lssvm_cg_info *info, const lssvm_mi355_options *options);

/* The same solve on SEVERAL devices of the calling process -- what a plssvm::csvm backend needs, since the reference drives all
 * its devices from one process behind csvm::fit (gpu_csvm.hpp:283-299 device split, :574-593 per-device launches, :449-475
 * device_reduction).  The implicit matrix is row-block sharded over the devices (the data matrix is replicated), one stream per
 * device is driven by the calling thread, and the partial K*v vectors are combined once per matvec with an RCCL all-reduce
 * (ncclCommInitAll) or, option "exchange" = 2, by peer kernels over xGMI.
 *   num_devices == 0: automatic = every visible device, but at least 4096 points per device; devices must then be NULL;
 *   num_devices >= 1: devices[0..num_devices) are HIP device ordinals, or NULL for 0 .. num_devices-1.  The same ordinal may be
 *   listed several times (shards then share that device; the exchange falls back to peer kernels): how the sharded path is
 *   exercised on a single-GPU machine. */
int lssvm_mi355_solve_multi_f32(const lssvm_params *params, const float *X, size_t num_points, size_t num_features, const float *y,
                                float eps, uint64_t max_iter, float *alpha_out, float *rho_out, lssvm_cg_info *info,
                                const int *devices, int num_devices, const lssvm_mi355_options *options);
int lssvm_mi355_solve_multi_f64(const lssvm_params *params, const double *X, size_t num_points, size_t num_features, const double *y,
                                double eps, uint64_t max_iter, double *alpha_out, double *rho_out, lssvm_cg_info *info,
                                const int *devices, int num_devices, const lssvm_mi355_options *options);

/* csvm::predict_values (csvm.hpp:204, :208; recipe: backends/OpenMP/csvm.cpp:188-227, HIP/predict_kernel.hip.hpp:34-117).
 * w_inout has num_features entries; *w_valid != 0 on entry means it already holds w (linear kernel only), on exit it is
 * set to 1 when w was computed (calculate_w, csvm.cpp:255-280).  out: num_predict_points decision values.  info (may be NULL): the timings of the call. */
int lssvm_mi355_predict_values_f32(const lssvm_params *params, const float *support_vectors, size_t num_support_vectors,
                                   size_t num_features, const float *alpha, float rho, float *w_inout, int *w_valid,
                                   const float *predict_points, size_t num_predict_points, float *out, lssvm_predict_info *info,
                                   const lssvm_mi355_options *options);
int lssvm_mi355_predict_values_f64(const lssvm_params *params, const double *support_vectors, size_t num_support_vectors,
                                   size_t num_features, const double *alpha, double rho, double *w_inout, int *w_valid,
                                   const double *predict_points, size_t num_predict_points, double *out, lssvm_predict_info *info,
                                   const lssvm_mi355_options *options);

/* The same prediction with the MODEL RESIDENT in HBM across calls (no counterpart in the reference, whose predict_values uploads the support vectors on every call --
 * gpu_csvm.hpp:656-730 -- which is the whole cost of a small batch): `create` uploads the support vectors and prepares them once (centring, norms, operand planes,
 * packed records; the linear kernel: w), `predict` uploads a batch of points (host memory, num_points x num_features row-major of the predictor's dtype), prepares
 * it alike and writes num_points decision values.  fp32 rbf / polynomial models of at most 128 features run against the resident form; everything else, and any batch
 * the resident form cannot take (it lies further from the support vectors' centre than the norm expansion allows, or two f16 planes do not represent it), goes through
 * the one-shot path (a resident model keeps no host copy: its support vectors come back from HBM the first time that happens) -- the values are the same either way,
 * lssvm_predict_info.resident says which.  `mem_kind` (LSSVM_MEM_HOST / LSSVM_MEM_DEVICE, as lssvm_mi355_problem_create's) says where BOTH `predict_points` and `out`
 * live: with LSSVM_MEM_DEVICE they are memory of device 0 (e.g. torch tensors' data_ptr) and a batch crosses no PCIe at all.  Device 0, like predict_values; calls on
 * one handle are not re-entrant. */
typedef struct lssvm_mi355_predictor lssvm_mi355_predictor; /* opaque */
int lssvm_mi355_predictor_create(lssvm_mi355_predictor **out, const lssvm_params *params, int dtype, const void *support_vectors, size_t num_support_vectors,
                                 size_t num_features, const void *alpha, double rho, const lssvm_mi355_options *options);
int lssvm_mi355_predictor_predict(lssvm_mi355_predictor *predictor, const void *predict_points, int mem_kind, size_t num_predict_points, void *out, lssvm_predict_info *info);
int lssvm_mi355_predictor_destroy(lssvm_mi355_predictor *predictor);

/* ------------------------------------------------------------------------------------------------------------------ */
/* fine-grained entry points for kernel-level parity tests: the protected members the reference's backend tests re-export  */
/* (tests/backends/HIP/mock_hip_csvm.hpp:22-44): generate_q, run_device_kernel, calculate_w                           */
/* ------------------------------------------------------------------------------------------------------------------ */

/* gpu_csvm::generate_q (gpu_csvm.hpp:349-384) / openmp generate_q (csvm.cpp:232-251): q_out has N-1 entries */
int lssvm_mi355_generate_q_f32(const lssvm_params *params, const float *X, size_t num_points, size_t num_features, float *q_out, const lssvm_mi355_options *options);
int lssvm_mi355_generate_q_f64(const lssvm_params *params, const double *X, size_t num_points, size_t num_features, double *q_out, const lssvm_mi355_options *options);

/* run_device_kernel (gpu_csvm.hpp:431-447 / csvm.cpp:283-306): ret[0..N-1) += add * Abar * d with
 * Abar_ij = k(x_i,x_j) + delta_ij / C + QA_cost - q_i - q_j; add must be +1 or -1 (svm_kernel.cpp:28). */
int lssvm_mi355_run_device_kernel_f32(const lssvm_params *params, const float *X, size_t num_points, size_t num_features,
                                      const float *q, const float *d, float *ret_inout, float QA_cost, float add, const lssvm_mi355_options *options);
int lssvm_mi355_run_device_kernel_f64(const lssvm_params *params, const double *X, size_t num_points, size_t num_features,
                                      const double *q, const double *d, double *ret_inout, double QA_cost, double add, const lssvm_mi355_options *options);

/* calculate_w (gpu_csvm.hpp:386-429 / csvm.cpp:255-280): w[f] = sum_i alpha_i * sv[i][f] */
int lssvm_mi355_calculate_w_f32(const float *support_vectors, size_t num_support_vectors, size_t num_features, const float *alpha, float *w_out);
int lssvm_mi355_calculate_w_f64(const double *support_vectors, size_t num_support_vectors, size_t num_features, const double *alpha, double *w_out);

/* ------------------------------------------------------------------------------------------------------------------ */
/* resident-problem API: what the one-shot calls are built from.  The data matrix is uploaded once and stays in HBM;  */
/* the CG recipe is split into begin / step / finish so that a caller (bench.py, a multi-rank launcher) can time the  */
/* iterations alone and can drive row-block sharding across several GPUs (one rank = one process = one GPU).          */
/* Replaces: gpu_csvm::setup_data_on_device (gpu_csvm.hpp:302-346) + the CG loop (gpu_csvm.hpp:477-654).             */
/* ------------------------------------------------------------------------------------------------------------------ */
typedef struct lssvm_mi355_problem lssvm_mi355_problem; /* opaque */

#define LSSVM_DTYPE_F32 0
#define LSSVM_DTYPE_F64 1

#define LSSVM_MEM_HOST 0   /* X points to host memory */
#define LSSVM_MEM_DEVICE 1 /* X points to memory of `device` (e.g. a torch tensor's data_ptr); it is copied into the padded layout */

/* Row-block sharding descriptor.  rank r of `world` owns a contiguous block of output rows of the
 * implicit matrix; the data matrix is replicated.  world == 1: single GPU, no communicator needed.
 * For world > 1 the ranks exchange their partial K*v once per implicit matvec: over RCCL (lssvm_mi355_comm_init on this rank first)
 * or over HIP IPC (lssvm_mi355_problem_ipc_export / _connect after the problem exists); option "exchange" selects. */
typedef struct lssvm_shard {
    int32_t rank;
    int32_t world;
} lssvm_shard;

/* Host only, no device needed: the 128-row blocks [*block_begin, *block_end) of the implicit matrix that shard `rank` of `world`
 * evaluates for a data set of `num_points` points.  symmetric != 0: only tiles on/below the diagonal are evaluated, blocks are
 * dealt by equal AREA (boundary r = round(blocks * sqrt(r / world))); else equal contiguous runs.  The partition every
 * Problem uses (plssvm_amd/sharding.py restates it for the flop accounting of bench.py). */
int lssvm_mi355_shard_blocks(size_t num_points, int world, int rank, int symmetric, int64_t *block_begin, int64_t *block_end);
/* RCCL bootstrap: rank 0 obtains a 128-byte unique id and hands it to the other ranks out of band
 * (bench.py / the Python launcher broadcast it with torch.distributed); every rank then calls comm_init.
 * One communicator per process.  Replaces the reference's host-staged device_reduction (gpu_csvm.hpp:449-475). */
#define LSSVM_UNIQUE_ID_BYTES 128
int lssvm_mi355_comm_get_unique_id(unsigned char id_out[LSSVM_UNIQUE_ID_BYTES]);
int lssvm_mi355_comm_init(int device, int rank, int world, const unsigned char id[LSSVM_UNIQUE_ID_BYTES]);
int lssvm_mi355_comm_destroy(void);

/* One process per GPU WITHOUT RCCL (option "exchange" = 2, or "exchange" = 0 and no communicator in this process): the ranks of one
 * node map each other's partial K*v vectors with HIP IPC and every rank sums (symmetric variant) or gathers (full square) them in
 * rank order with the peer kernel the single-process mode uses -- the same bits on every rank.  The ranks meet at host flags in
 * POSIX shared memory, twice per implicit matvec.  After lssvm_mi355_problem_create(..., shard) on every rank:
 *   1. every rank: lssvm_mi355_problem_ipc_export -> LSSVM_IPC_BLOB_BYTES bytes,
 *   2. the application hands every rank the blobs of ALL ranks, concatenated in rank order (bench.py: torch.distributed all_gather),
 *   3. every rank: lssvm_mi355_problem_ipc_connect.
 * HSA_ENABLE_IPC_MODE_LEGACY=0 must be set where the host driver only supports dmabuf IPC.  A rank that waits longer than option
 * "ipc_timeout_s" for its peers fails with LSSVM_ERR_COMM and makes the others fail too. */
#define LSSVM_IPC_BLOB_BYTES 256
int lssvm_mi355_problem_ipc_export(lssvm_mi355_problem *p, void *blob_out, size_t blob_bytes);
int lssvm_mi355_problem_ipc_connect(lssvm_mi355_problem *p, const void *blobs, size_t total_bytes);

/* upload X (N x d row-major, dtype per `dtype`), compute q, QA_cost and the per-row norms on `device`. */
int lssvm_mi355_problem_create(lssvm_mi355_problem **out, const lssvm_params *params, int dtype, const void *X, int mem_kind,
                               size_t num_points, size_t num_features, int device, const lssvm_shard *shard /* NULL = single GPU */, const lssvm_mi355_options *options);
/* the same on several devices of this process (see lssvm_mi355_solve_multi_*); mem_kind LSSVM_MEM_DEVICE: X may live on any of them.
 * Every lssvm_mi355_problem_* / lssvm_mi355_cg_* call below accepts the handle. */
int lssvm_mi355_problem_create_multi(lssvm_mi355_problem **out, const lssvm_params *params, int dtype, const void *X, int mem_kind,
                                     size_t num_points, size_t num_features, const int *devices, int num_devices, const lssvm_mi355_options *options);
int lssvm_mi355_problem_destroy(lssvm_mi355_problem *p);

/* read back q (N-1 entries, dtype of the problem) and QA_cost */
int lssvm_mi355_problem_get_q(lssvm_mi355_problem *p, void *q_out, double *QA_cost_out);

/* ret[0..N-1) += add * Abar * d (host vectors of the problem's dtype, N-1 entries EACH: the library reads and writes exactly
 * num_points - 1 elements of both).  With sharding every rank returns the full vector. */
int lssvm_mi355_problem_matvec(lssvm_mi355_problem *p, const void *d, void *ret_inout, double add);

/* CG, csvm.cpp:89-111: b = y[0..n) - y[n], x = 1, r = b - A x, delta0, d = r */
int lssvm_mi355_cg_begin(lssvm_mi355_problem *p, const void *y, double eps);
/* run at most `iterations` further CG iterations (csvm.cpp:125-166); stops early when delta <= eps^2 delta0.
 * *done_out (may be NULL) is set to 1 when the stop test fired. */
int lssvm_mi355_cg_step(lssvm_mi355_problem *p, uint64_t iterations, int *done_out);
/* csvm.cpp:179-182: bias, alpha[N-1] = -sum, rho = -bias; alpha_out has N entries of the problem's dtype */
int lssvm_mi355_cg_finish(lssvm_mi355_problem *p, void *alpha_out, double *rho_out, lssvm_cg_info *info);
/* block until all work queued on the problem's stream has finished (gpu_csvm::device_synchronize, gpu_csvm.hpp:208) */
int lssvm_mi355_problem_synchronize(lssvm_mi355_problem *p);
/* timing / counters accumulated since cg_begin (same struct as cg_finish fills); does not synchronize */
int lssvm_mi355_problem_info(lssvm_mi355_problem *p, lssvm_cg_info *info);

/* ---- LIBSVM data files (the input format of plssvm-train, include/plssvm/detail/io/libsvm_parsing.hpp:47-229) ----
 * Multi-threaded fast path for WELL-FORMED files: `open` reads and validates the file (labels on every line or on none; one-based,
 * strictly increasing indices; every token converts) and reports its shape, `fill` writes the dense row-major matrix (missing
 * features = 0; ldx >= num_features, in elements) and the labels (always double; may be NULL; ignored for unlabelled files).  Any irregularity
 * -> LSSVM_ERR_INVALID_ARGUMENT without a diagnosis: callers re-parse with their reference-exact parser for the error message
 * (plssvm_amd/io_libsvm.py does).  Host code only, no device needed. */
typedef struct lssvm_mi355_libsvm_file lssvm_mi355_libsvm_file;
int lssvm_mi355_libsvm_open(const char *path, uint64_t skipped_lines, lssvm_mi355_libsvm_file **file_out, uint64_t *num_points, uint64_t *num_features,
                            int *has_label);
int lssvm_mi355_libsvm_fill_f32(lssvm_mi355_libsvm_file *file, float *X, uint64_t ldx, double *labels);
int lssvm_mi355_libsvm_fill_f64(lssvm_mi355_libsvm_file *file, double *X, uint64_t ldx, double *labels);
int lssvm_mi355_libsvm_close(lssvm_mi355_libsvm_file *file);

/* ---- LIBSVM data files: writer (include/plssvm/detail/io/libsvm_parsing.hpp:244-296) ----
 * `header` (may be NULL) is written verbatim first -- the reference's two comment lines "# This data set has been created at ..." and "# NxD" are the caller's
 * to word.  Then one line per point: "label idx:val idx:val ... " -- every number as {:.10e}, zero features left out, one-based indices, a blank after every
 * token, like the reference.  The label column: int_labels (whole numbers, written as integers), OR label_text + label_offsets (point i's label is the
 * bytes [label_offsets[i], label_offsets[i + 1]) of label_text: string labels, or numbers the caller has already formatted), or neither: no labels.  The
 * points are written in their order by all host threads (chunks of rows formatted concurrently, flushed in order): the file does not depend on the thread
 * count (the reference's order is unspecified, :251).  Host code only, no device needed. */
int lssvm_mi355_libsvm_write_f32(const char *path, const char *header, const float *X, uint64_t num_points, uint64_t num_features, uint64_t ldx,
                                 const int64_t *int_labels, const char *label_text, const uint64_t *label_offsets);
int lssvm_mi355_libsvm_write_f64(const char *path, const char *header, const double *X, uint64_t num_points, uint64_t num_features, uint64_t ldx,
                                 const int64_t *int_labels, const char *label_text, const uint64_t *label_offsets);

/* ---- LIBSVM model files: what plssvm-train writes and plssvm-predict reads (include/plssvm/detail/io/libsvm_model_parsing.hpp) ----
 * Writer (:371-499): `header` -- the lines from the "#" time stamp to "SV" (:296-342; a dozen short lines, worded by the caller: plssvm_amd/model.py) -- is
 * written verbatim, then `count` lines "alpha idx:val idx:val ... " ({:.10e}, zeros left out, a blank after every token) for the support vectors
 * order[0 .. count) -- the caller lists them grouped by class in the order of the header's "label" line (:416-499); order == NULL: all of them as they lie
 * (count must equal num_support_vectors).  Multi-threaded like the data writer, the same bytes for any thread count. */
int lssvm_mi355_model_write_f32(const char *path, const char *header, const float *support_vectors, uint64_t num_support_vectors, uint64_t num_features,
                                uint64_t ldx, const float *alpha, const uint64_t *order, uint64_t count);
int lssvm_mi355_model_write_f64(const char *path, const char *header, const double *support_vectors, uint64_t num_support_vectors, uint64_t num_features,
                                uint64_t ldx, const double *alpha, const uint64_t *order, uint64_t count);
/* Reader (:64-262), the same contract as the data readers: a multi-threaded fast path for WELL-FORMED files -- every header key at most once and spelled
 * out ("svm_type c_svc", "kernel_type linear|polynomial|rbf", "degree", "gamma", "coef0", "nr_class", "total_sv", "rho", "label", "nr_sv", then "SV"; any
 * order and case), only the parameters its kernel uses, counts that add up, then exactly total_sv lines "alpha idx:val ..." that follow the data-file
 * rules.  `open` maps and validates the file and reports the header; `labels` hands out the "label" line's entries (separated by single blanks, NUL
 * terminated, lssvm_model_info.label_text_bytes bytes; the caller converts them to its label type and checks them for duplicates AFTER that conversion,
 * as the reference does) and the nr_sv counts (nr_class entries); `fill` writes the support vectors (dense row-major, missing features = 0) and their
 * weights.  Any irregularity -> LSSVM_ERR_INVALID_ARGUMENT without a diagnosis: callers re-parse with their reference-exact parser for the error message
 * (plssvm_amd/model.py does). */
typedef struct lssvm_model_info {
    int32_t kernel_type; /* lssvm_kernel_type */
    int32_t has_degree;  /* which of the kernel parameters the header states (an rbf / polynomial header may leave gamma to the 1 / num_features default) */
    int32_t has_gamma;
    int32_t has_coef0;
    int64_t degree;
    double gamma;
    double coef0;
    double rho;
    uint64_t nr_class;
    uint64_t total_sv;         /* = number of support-vector lines */
    uint64_t num_features;     /* largest feature index of the body */
    uint64_t label_text_bytes; /* bytes lssvm_mi355_model_labels writes into label_text_out, the terminating NUL included */
} lssvm_model_info;
typedef struct lssvm_mi355_model_file lssvm_mi355_model_file;
int lssvm_mi355_model_open(const char *path, lssvm_mi355_model_file **file_out, lssvm_model_info *info);
int lssvm_mi355_model_labels(lssvm_mi355_model_file *file, char *label_text_out, uint64_t label_text_bytes, uint64_t *nr_sv_out);
int lssvm_mi355_model_fill_f32(lssvm_mi355_model_file *file, float *support_vectors, uint64_t ldx, float *alpha);
int lssvm_mi355_model_fill_f64(lssvm_mi355_model_file *file, double *support_vectors, uint64_t ldx, double *alpha);
int lssvm_mi355_model_close(lssvm_mi355_model_file *file);

/* ---- ARFF data files (the second format of plssvm::data_set, include/plssvm/detail/io/arff_parsing.hpp:57-372) ----
 * The same contract as the LIBSVM reader above: a multi-threaded fast path for WELL-FORMED files -- "@RELATION name", "@ATTRIBUTE name NUMERIC" per feature, at
 * most one "@ATTRIBUTE class {l1,l2,...}" with numeric labels, "@DATA", then dense rows (one value per attribute) or sparse rows ("{index value,...}", zero-based
 * attribute indices) whose labels are among the header's.  int_labels != 0: the caller's label type is an integer, the labels must be written as plain integers.
 * `fill` writes the dense row-major matrix and the labels (always double; may be NULL).  Any irregularity -> LSSVM_ERR_INVALID_ARGUMENT without a diagnosis:
 * callers re-parse with their reference-exact parser for the error message (plssvm_amd/io_arff.py does).  Host code only, no device needed. */
typedef struct lssvm_mi355_arff_file lssvm_mi355_arff_file;
int lssvm_mi355_arff_open(const char *path, int int_labels, lssvm_mi355_arff_file **file_out, uint64_t *num_points, uint64_t *num_features, int *has_label);
int lssvm_mi355_arff_fill_f32(lssvm_mi355_arff_file *file, float *X, uint64_t ldx, double *labels);
int lssvm_mi355_arff_fill_f64(lssvm_mi355_arff_file *file, double *X, uint64_t ldx, double *labels);
int lssvm_mi355_arff_close(lssvm_mi355_arff_file *file);

/* tuning knobs, by name (all have defaults; unknown names -> LSSVM_ERR_INVALID_ARGUMENT).  set_option changes the process-wide DEFAULTS (thread safe);
 * every problem / solve that is not handed a lssvm_mi355_options of its own takes a snapshot of them when it is created, so later changes never affect a live problem.  The defaults can
 * be preset from the environment: LSSVM_MI355_OPTIONS="name=value,name=value" (read once when the library is loaded).  Thirteen options here, two
 * testing aids and one experimental option in plssvm_amd_testing.h -- who sets each besides the tests: DESIGN.md section 4.5 (round 4 retired xcd_map, lds_extra_kb, item_order,
 * linear_panel_features, check_shards, rbf_direct_above and mfma_shape = 1: measured, decided, constants now):
 *   "rbf_form"      fp32 rbf: 0 = automatic (default): the norm expansion c_i + c_j + x_i'.x_j' on the matrix cores, unless
 *                   R2 = 2 gamma log2(e) max|x - mean|^2 exceeds 32 -- the expansion's exponent carries an absolute error of
 *                   ~2^-24 R2 whatever the distance of the pair, which nearby pairs (K ~ 1) see as a relative error of K
 *                   ([-1,1]-scaled data with gamma = 1 / num_features has R2 <= 3); then, up to R2 = 4096 (times sqrt(128 / num_features) beyond 128 features), the matrix
 *                   cores on GRID planes (since round 5: x = h + s1 + s2 with h on a grid, the accumulators started from the exact grid norms and
 *                   fed the h.h products first, so that the large terms cancel exactly -- the direct form's accuracy at 1.6x the f16x3 time); beyond
 *                   that, and with 1 = always, the formula-exact (x_i - x_j)^2 kernel on the vector ALU (10x slower than the grid planes at
 *                   50 000 x 128); 2 = always the norm expansion; 3 = the grid planes wherever they exist (R2 within that limit)
 *   "rbf_fold"      fp32 rbf on the split kernels: 1 (default) = the column records carry (2^c_j d_j | 2^c_j) and the accumulators start
 *                   from c_i as the C operand of their first MFMA (256-row workgroups: from 0, the row's term folded too) -- no start-value
 *                   instructions, K_ij = 2^acc 2^c_j (one more rounding than 2^(acc + c_j); used while the exponent scale R2 <= 200 keeps both factors
 *                   far inside the fp32 range); 0 = start values c_i + c_j
 *   "j_chunk_tiles" number of 128-column tiles per work item; 0 = automatic (default: about 4096 work items per device, 2 ... 16 tiles each,
 *                   up to 64 for the split kernels; 256-row workgroups: the length whose replayed dispatch over the CUs finishes first)
 *   "j_chunk_head"  256-row workgroups (see "mfma_shape"): the FIRST `count` column chunks of every pair of row blocks have `tiles` tiles instead of j_chunk_tiles, value =
 *                   1024 count + tiles; their short work items are dispatched last and fill the final dispatch round of a launch that is only a few rounds long.
 *                   1 (default) = the replayed dispatch chooses a head together with the chunk length (j_chunk_tiles = 0; 3-5 % at 20 000-50 000 points since the
 *                   launches are persistent and draw their items from per-XCD counters); 0 = none
 *   "symmetric"     1 = evaluate only the kernel-matrix tiles on/below the diagonal and mirror them (default; any num_features -- beyond 512 (fp32) /
 *                   256 (fp64) features over feature panels -- except fp32 with gram_mode = 0 beyond 512 features and a negative polynomial
 *                   degree, which run the full square),
 *                   0 = full square (row-owned sums, results independent of the GPU count)
 *   "tile_kernel"   0 = automatic: the "resident row panel" kernels for num_features <= 512 (fp32) / 256 (fp64) and their feature-panel forms beyond
 *                   (default), 1 = always the generic kernel
 *   "gram_mode"     fp32 Gram tiles on the 16-bit matrix cores with fp32 accumulation, at fp32-equivalent accuracy:
 *                   3 (default) = "f16x3" where the data allows, else "bf16x6"; 2 = "f16x3": every operand as TWO f16 planes (hi + mid; rbf: shifted by
 *                   2^-6 / 2^6, others pre-scaled by a power of two), three plane products on v_mfma_f32_16x16x32_f16; num_features <= 512 (rbf 384)
 *                   in one pass, beyond that over feature panels of 128 (linear: one launch per panel; rbf / polynomial: inside a tile, symmetric variant)
 *                   -- without the check whether two f16 planes represent THIS data as well as fp32 does, which mode 3 makes at set-up (linear kernel, round 6: where ONE
 *                   power-of-two scale for the matrix fails that check -- points that differ by orders of magnitude -- mode 3 gives every ROW its own scale and stays f16x3);
 *                   1 = "bf16x6": exact split into THREE bf16 planes, six plane products on v_mfma_f32_16x16x32_bf16, num_features <= 384 in one
 *                   pass (rbf / polynomial beyond that: feature panels inside a tile);
 *                   0 = Gram tiles on v_mfma_f32_32x32x2_f32 (exact fmaf chains)
 *   "mfma_shape"    split kernels, symmetric variant, at most 128 features per pass: 3 (default) = 256-row workgroups -- eight waves on a pair of row
 *                   blocks share one column stream per CU -- from 64 row blocks (8 192 points) on; 2 = 128-row workgroups (four waves) throughout
 *   "colslab_band_mb" the symmetric variant leaves one record of 128 column sums per evaluated off-diagonal tile (256-row workgroups: per pair of row
 *                   blocks and column tile); the device's row blocks are
 *                   cut into bands of equal area whose records fit this many MiB (default 2048), the tile kernel runs band by band into one
 *                   slab and every band is folded into K*v before the next (1M points in fp32: 15.6 GB of records -> 8 bands, 2 GiB)
 *   "colslab_limit_mb" upper bound of the band slab (default 98304); 0 switches the symmetric variant off (full square)
 *   "exchange"      several devices in one process (the _multi entry points): 0 = automatic (RCCL when the listed devices are distinct, peer
 *                   kernels otherwise; default), 1 = RCCL all-reduce / all-gather, 2 = peer kernels: every device adds the partial vectors of all
 *                   devices through its xGMI peer mappings in rank order (bit-equal on all devices, deterministic).
 *                   One process per GPU (lssvm_shard): 0 = RCCL when lssvm_mi355_comm_init was called in this process, else HIP IPC;
 *                   1 = RCCL; 2 = HIP IPC + the peer kernel (lssvm_mi355_problem_ipc_export / _connect)
 *   "ipc_timeout_s" one process per GPU over HIP IPC: seconds a rank waits for its peers at an exchange before it fails (default 600)
 *   "enqueue_ahead_below_us" CG loop: while an implicit matvec is expected to take less than this many microseconds (default 5000; the expectation is a rule
 *                   on the shape, the real type and the kernel path -- never a measurement, every rank of a sharded solve must agree), the direction update and
 *                   the NEXT matvec are enqueued before the host reads the stop test of the current iteration, so the device never waits for
 *                   the host; they touch d and K*d only, so a converged solve ends exactly where the reference's does, one matvec is discarded.
 *                   0 = the host reads every stop test before it enqueues anything further
 */
int lssvm_mi355_set_option(const char *name, int64_t value);
int lssvm_mi355_get_option(const char *name, int64_t *value_out);

#ifdef __cplusplus
}
#endif
#endif /* PLSSVM_AMD_H */
