/*
 * capi.hip -- the extern "C" surface of libplssvm_amd.so (declared in include/plssvm_amd.h).
 * No exception crosses this boundary: every entry point maps lssvm::Error / std::exception to a negative status and
 * stores the message in a thread-local string (lssvm_mi355_last_error).
 */
#include "lssvm_problem.hip.hpp"

#include <dlfcn.h>
#include <cstdlib>
#include <memory>
#include <new>

/* the options of one caller (ABI 4): a private copy of the tuning knobs */
struct lssvm_mi355_options {
    lssvm::Options o;
};

namespace {

using lssvm::guarded;

/* the options by name: ONE rule book for the process defaults (lssvm_mi355_set_option) and for a caller's own object (lssvm_mi355_options_set) */
void set_option_in(lssvm::Options &o, const char *name, int64_t value) {
    const std::string n(name);
    if (n == "rbf_form") {
        LSSVM_REQUIRE(value >= 0 && value <= 3, "rbf_form must be 0 (automatic), 1 (direct), 2 (matrix cores, norm expansion) or 3 (matrix cores, grid planes)");
        o.rbf_form = value;
    } else if (n == "rbf_fold") {
        o.rbf_fold = value != 0 ? 1 : 0;
    } else if (n == "j_chunk_tiles") {
        LSSVM_REQUIRE(value >= 0 && value <= (1 << 20), "j_chunk_tiles out of range");
        o.j_chunk_tiles = value;
    } else if (n == "j_chunk_head") {
        LSSVM_REQUIRE(value >= 0 && value < (1 << 20), "j_chunk_head out of range (1024 count + tiles)");
        o.j_chunk_head = value;
    } else if (n == "symmetric") {
        o.symmetric = value != 0 ? 1 : 0;
    } else if (n == "tile_kernel") {
        LSSVM_REQUIRE(value == 0 || value == 1, "tile_kernel must be 0 (automatic) or 1 (generic kernel)");
        o.tile_kernel = value;
    } else if (n == "debug_ablate") {
#ifdef LSSVM_ENABLE_ABLATION
        o.debug_ablate = value;
#else
        LSSVM_REQUIRE(value == 0, "debug_ablate exists in builds with -DLSSVM_ENABLE_ABLATION only");
#endif
    } else if (n == "force_collective") {
        o.force_collective = value != 0 ? 1 : 0;
    } else if (n == "gram_mode") {
        LSSVM_REQUIRE(value >= 0 && value <= 3, "gram_mode must be 0 (v_mfma_f32), 1 (bf16x6), 2 (f16x3 unchecked) or 3 (f16x3 where the data allows, else bf16x6)");
        o.gram_mode = value;
    } else if (n == "mfma_shape") {
        LSSVM_REQUIRE(value == 2 || value == 3, "mfma_shape must be 2 (128-row workgroups) or 3 (256-row workgroups in the symmetric variant where they apply)");
        o.mfma_shape = value;
    } else if (n == "item_order_dev") {
#ifdef LSSVM_DEV_SUBSET
        LSSVM_REQUIRE(value >= 0 && value <= 7, "item_order_dev must be 0 ... 7");
        o.item_order_dev = value;
#else
        LSSVM_REQUIRE(value == 0, "item_order_dev exists in development builds (make DEV=1) only");
#endif
    } else if (n == "pair_lag") {
#ifdef LSSVM_DEV_SUBSET
        LSSVM_REQUIRE(value >= 0 && value <= 7 && value != 2, "pair_lag must be 0, 1, 3 (steps of lag) or 4 ... 7 (priority experiments)");
        o.pair_lag = value;
#else
        LSSVM_REQUIRE(value == 0, "pair_lag exists in development builds (make DEV=1) only");
#endif
    } else if (n == "colslab_band_mb") {
        LSSVM_REQUIRE(value >= 1, "colslab_band_mb must be positive");
        o.colslab_band_mb = value;
    } else if (n == "colslab_limit_mb") {
        LSSVM_REQUIRE(value >= 0, "colslab_limit_mb must not be negative");
        o.colslab_limit_mb = value;
    } else if (n == "skip_collective") {
        o.skip_collective = value != 0 ? 1 : 0;
    } else if (n == "exchange") {
        LSSVM_REQUIRE(value >= 0 && value <= 2, "exchange must be 0 (automatic), 1 (RCCL) or 2 (peer kernels)");
        o.exchange = value;
    } else if (n == "ipc_timeout_s") {
        LSSVM_REQUIRE(value >= 1, "ipc_timeout_s must be at least 1");
        o.ipc_timeout_s = value;
    } else if (n == "enqueue_ahead_below_us") {
        LSSVM_REQUIRE(value >= 0, "enqueue_ahead_below_us must not be negative");
        o.enqueue_ahead_below_us = value;
    } else if (n == "rebalance_after") {
        LSSVM_REQUIRE(value >= 0, "rebalance_after must not be negative");
        o.rebalance_after = value;
    } else {
        throw lssvm::Error(LSSVM_ERR_INVALID_ARGUMENT, "unknown option '" + n + "'");
    }
}
void get_option_from(const lssvm::Options &o, const char *name, int64_t *value_out) {
    const std::string n(name);
    if (n == "rbf_form") {
        *value_out = o.rbf_form;
    } else if (n == "rbf_fold") {
        *value_out = o.rbf_fold;
    } else if (n == "j_chunk_tiles") {
        *value_out = o.j_chunk_tiles;
    } else if (n == "j_chunk_head") {
        *value_out = o.j_chunk_head;
    } else if (n == "symmetric") {
        *value_out = o.symmetric;
    } else if (n == "tile_kernel") {
        *value_out = o.tile_kernel;
    } else if (n == "debug_ablate") {
        *value_out = o.debug_ablate;
    } else if (n == "force_collective") {
        *value_out = o.force_collective;
    } else if (n == "gram_mode") {
        *value_out = o.gram_mode;
    } else if (n == "mfma_shape") {
        *value_out = o.mfma_shape;
    } else if (n == "item_order_dev") {
        *value_out = o.item_order_dev;
    } else if (n == "pair_lag") {
        *value_out = o.pair_lag;
    } else if (n == "colslab_band_mb") {
        *value_out = o.colslab_band_mb;
    } else if (n == "colslab_limit_mb") {
        *value_out = o.colslab_limit_mb;
    } else if (n == "skip_collective") {
        *value_out = o.skip_collective;
    } else if (n == "exchange") {
        *value_out = o.exchange;
    } else if (n == "ipc_timeout_s") {
        *value_out = o.ipc_timeout_s;
    } else if (n == "enqueue_ahead_below_us") {
        *value_out = o.enqueue_ahead_below_us;
    } else if (n == "rebalance_after") {
        *value_out = o.rebalance_after;
    } else {
        throw lssvm::Error(LSSVM_ERR_INVALID_ARGUMENT, "unknown option '" + n + "'");
    }
}

/* the options a call runs with: the caller's object, or a snapshot of the process defaults */
lssvm::Options options_of(const lssvm_mi355_options *options) { return options != nullptr ? options->o : lssvm::options_snapshot(); }

struct Handle {
    std::unique_ptr<lssvm::ProblemBase> impl;
};

lssvm::ProblemBase *impl_of(lssvm_mi355_problem *p) {
    LSSVM_REQUIRE(p != nullptr, "problem handle must not be NULL");
    return reinterpret_cast<Handle *>(p)->impl.get();
}

template <typename T>
void solve_one_shot(const lssvm_params *params, const T *X, size_t N, size_t d, const T *y, T eps, uint64_t max_iter, T *alpha_out, T *rho_out, lssvm_cg_info *info,
                    const int *devices, int num_devices, const lssvm_mi355_options *options) {
    lssvm::check_params(params);
    LSSVM_REQUIRE(X != nullptr && N > 0, "The data must not be empty!");                                                                  // csvm.cpp:73
    LSSVM_REQUIRE(d > 0, "The data points must contain at least one feature!");                                                           // csvm.cpp:74
    LSSVM_REQUIRE(y != nullptr, "The number of data points in the matrix A and the values in the right hand side vector must be the same!");  // csvm.cpp:76
    LSSVM_REQUIRE(eps > T(0), "The stopping criterion in the CG algorithm must be greater than 0.0, but is " + std::to_string(eps) + "!");  // csvm.cpp:77
    LSSVM_REQUIRE(max_iter > 0, "The number of CG iterations must be greater than 0!");                                                   // csvm.cpp:78
    LSSVM_REQUIRE(alpha_out != nullptr && rho_out != nullptr, "alpha_out / rho_out must not be NULL");
    const lssvm::Options opt = options_of(options);
    lssvm::Solver<T> prob(opt, *params, X, LSSVM_MEM_HOST, N, d, lssvm::resolve_devices(devices, num_devices, N), nullptr);
    prob.cg_begin(y, static_cast<double>(eps));
    // option rebalance_after (several devices, symmetric variant): the first iterations measure every shard's pace, then the shares follow it (lssvm_mi355_problem_rebalance)
    const uint64_t first = static_cast<uint64_t>(opt.rebalance_after);
    if (first > 0 && first < max_iter) {
        prob.cg_step(first, nullptr);
        (void) prob.rebalance(nullptr, 0);
        prob.cg_step(max_iter - first, nullptr);
    } else {
        prob.cg_step(max_iter, nullptr);
    }
    double rho = 0.0;
    lssvm_cg_info local{};
    prob.cg_finish(alpha_out, &rho, &local);
    local.max_iterations = max_iter;
    *rho_out = static_cast<T>(rho);
    if (info != nullptr) *info = local;
}

template <typename T>
void generate_q_one_shot(const lssvm_params *params, const T *X, size_t N, size_t d, T *q_out, const lssvm_mi355_options *options) {
    lssvm::check_params(params);
    LSSVM_REQUIRE(q_out != nullptr, "q_out must not be NULL");
    lssvm::Solver<T> prob(options_of(options), *params, X, LSSVM_MEM_HOST, N, d, { 0 }, nullptr);
    prob.get_q(q_out, nullptr);
}

}  // namespace

extern "C" {

static_assert(sizeof(lssvm_cg_info) == 168, "lssvm_cg_info changed: bump PLSSVM_AMD_ABI_VERSION and plssvm_amd/_capi.py (LssvmCgInfo) with it");
int lssvm_mi355_abi_version(void) { return PLSSVM_AMD_ABI_VERSION; }

const char *lssvm_mi355_last_error(void) { return lssvm::last_error_message().c_str(); }

int lssvm_mi355_device_count(void) {
    int count = 0;
    const hipError_t err = hipGetDeviceCount(&count);
    if (err != hipSuccess) {
        (void) hipGetLastError();
        if (err == hipErrorNoDevice) return 0;
        lssvm::last_error_message() = std::string("hipGetDeviceCount failed: ") + hipGetErrorString(err);
        return 0;
    }
    return count;
}

int lssvm_mi355_device_name(int device, char *buf, size_t buf_len) {
    return guarded([&] {
        LSSVM_REQUIRE(buf != nullptr && buf_len > 0, "buf must not be NULL");
        lssvm::select_device_checked(device);
        hipDeviceProp_t prop{};
        LSSVM_HIP_CHECK(hipGetDeviceProperties(&prop, device));
        std::snprintf(buf, buf_len, "%s (%s), %d CUs", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    });
}

int lssvm_mi355_solve_f32(const lssvm_params *params, const float *X, size_t num_points, size_t num_features, const float *y, float eps, uint64_t max_iter,
                          float *alpha_out, float *rho_out, lssvm_cg_info *info, const lssvm_mi355_options *options) {
    static const int device0 = 0;
    return guarded([&] { solve_one_shot<float>(params, X, num_points, num_features, y, eps, max_iter, alpha_out, rho_out, info, &device0, 1, options); });
}
int lssvm_mi355_solve_multi_f32(const lssvm_params *params, const float *X, size_t num_points, size_t num_features, const float *y, float eps, uint64_t max_iter,
                                float *alpha_out, float *rho_out, lssvm_cg_info *info, const int *devices, int num_devices, const lssvm_mi355_options *options) {
    return guarded([&] { solve_one_shot<float>(params, X, num_points, num_features, y, eps, max_iter, alpha_out, rho_out, info, devices, num_devices, options); });
}
int lssvm_mi355_solve_f64(const lssvm_params *params, const double *X, size_t num_points, size_t num_features, const double *y, double eps, uint64_t max_iter,
                          double *alpha_out, double *rho_out, lssvm_cg_info *info, const lssvm_mi355_options *options) {
    static const int device0 = 0;
    return guarded([&] { solve_one_shot<double>(params, X, num_points, num_features, y, eps, max_iter, alpha_out, rho_out, info, &device0, 1, options); });
}
int lssvm_mi355_solve_multi_f64(const lssvm_params *params, const double *X, size_t num_points, size_t num_features, const double *y, double eps, uint64_t max_iter,
                                double *alpha_out, double *rho_out, lssvm_cg_info *info, const int *devices, int num_devices, const lssvm_mi355_options *options) {
    return guarded([&] { solve_one_shot<double>(params, X, num_points, num_features, y, eps, max_iter, alpha_out, rho_out, info, devices, num_devices, options); });
}

int lssvm_mi355_predict_values_f32(const lssvm_params *params, const float *sv, size_t nsv, size_t nfeat, const float *alpha, float rho, float *w_inout,
                                   int *w_valid, const float *points, size_t npoints, float *out, lssvm_predict_info *info, const lssvm_mi355_options *options) {
    return guarded([&] {
        LSSVM_REQUIRE(params != nullptr, "params must not be NULL!");
        lssvm::predict_values<float>(options_of(options), *params, sv, nsv, nfeat, alpha, rho, w_inout, w_valid, points, npoints, out, info);
    });
}
int lssvm_mi355_predict_values_f64(const lssvm_params *params, const double *sv, size_t nsv, size_t nfeat, const double *alpha, double rho, double *w_inout,
                                   int *w_valid, const double *points, size_t npoints, double *out, lssvm_predict_info *info, const lssvm_mi355_options *options) {
    return guarded([&] {
        LSSVM_REQUIRE(params != nullptr, "params must not be NULL!");
        lssvm::predict_values<double>(options_of(options), *params, sv, nsv, nfeat, alpha, rho, w_inout, w_valid, points, npoints, out, info);
    });
}

struct lssvm_mi355_predictor {
    std::unique_ptr<lssvm::PredictorBase> impl;
};
int lssvm_mi355_predictor_create(lssvm_mi355_predictor **out, const lssvm_params *params, int dtype, const void *support_vectors, size_t num_support_vectors,
                                 size_t num_features, const void *alpha, double rho, const lssvm_mi355_options *options) {
    return guarded([&] {
        LSSVM_REQUIRE(out != nullptr, "out must not be NULL");
        *out = nullptr;
        lssvm::check_params(params);
        LSSVM_REQUIRE(dtype == LSSVM_DTYPE_F32 || dtype == LSSVM_DTYPE_F64, "dtype must be LSSVM_DTYPE_F32 or LSSVM_DTYPE_F64");
        auto h = std::make_unique<lssvm_mi355_predictor>();
        h->impl = lssvm::make_predictor(options_of(options), *params, dtype, support_vectors, num_support_vectors, num_features, alpha, rho);
        *out = h.release();
    });
}
int lssvm_mi355_predictor_predict(lssvm_mi355_predictor *predictor, const void *predict_points, int mem_kind, size_t num_predict_points, void *out, lssvm_predict_info *info) {
    return guarded([&] {
        LSSVM_REQUIRE(predictor != nullptr, "predictor handle must not be NULL");
        LSSVM_REQUIRE(mem_kind == LSSVM_MEM_HOST || mem_kind == LSSVM_MEM_DEVICE, "invalid mem_kind");
        predictor->impl->predict(predict_points, mem_kind, num_predict_points, out, info);
    });
}
int lssvm_mi355_predictor_destroy(lssvm_mi355_predictor *predictor) {
    return guarded([&] { delete predictor; });
}

int lssvm_mi355_generate_q_f32(const lssvm_params *params, const float *X, size_t num_points, size_t num_features, float *q_out, const lssvm_mi355_options *options) {
    return guarded([&] { generate_q_one_shot<float>(params, X, num_points, num_features, q_out, options); });
}
int lssvm_mi355_generate_q_f64(const lssvm_params *params, const double *X, size_t num_points, size_t num_features, double *q_out, const lssvm_mi355_options *options) {
    return guarded([&] { generate_q_one_shot<double>(params, X, num_points, num_features, q_out, options); });
}

int lssvm_mi355_run_device_kernel_f32(const lssvm_params *params, const float *X, size_t num_points, size_t num_features, const float *q, const float *d,
                                      float *ret_inout, float QA_cost, float add, const lssvm_mi355_options *options) {
    return guarded([&] {
        lssvm::check_params(params);
        LSSVM_REQUIRE(q != nullptr, "The q array may not be empty!");  // csvm.cpp:284
        (void) QA_cost;  // q and QA_cost are functions of (X, params); they are recomputed on the device and must agree with the caller's
        lssvm::Solver<float> prob(options_of(options), *params, X, LSSVM_MEM_HOST, num_points, num_features, { 0 }, nullptr);
        prob.matvec(d, ret_inout, static_cast<double>(add));
    });
}
int lssvm_mi355_run_device_kernel_f64(const lssvm_params *params, const double *X, size_t num_points, size_t num_features, const double *q, const double *d,
                                      double *ret_inout, double QA_cost, double add, const lssvm_mi355_options *options) {
    return guarded([&] {
        lssvm::check_params(params);
        LSSVM_REQUIRE(q != nullptr, "The q array may not be empty!");
        (void) QA_cost;
        lssvm::Solver<double> prob(options_of(options), *params, X, LSSVM_MEM_HOST, num_points, num_features, { 0 }, nullptr);
        prob.matvec(d, ret_inout, add);
    });
}

int lssvm_mi355_calculate_w_f32(const float *sv, size_t nsv, size_t nfeat, const float *alpha, float *w_out) {
    return guarded([&] { lssvm::calculate_w<float>(sv, nsv, nfeat, alpha, w_out); });
}
int lssvm_mi355_calculate_w_f64(const double *sv, size_t nsv, size_t nfeat, const double *alpha, double *w_out) {
    return guarded([&] { lssvm::calculate_w<double>(sv, nsv, nfeat, alpha, w_out); });
}

int lssvm_mi355_shard_blocks(size_t num_points, int world, int rank, int symmetric, int64_t *block_begin, int64_t *block_end) {
    return guarded([&] {
        LSSVM_REQUIRE(block_begin != nullptr && block_end != nullptr, "output pointers must not be NULL");
        LSSVM_REQUIRE(num_points >= 2 && num_points < (size_t(1) << 31) - 4 * lssvm::TILE, "invalid number of data points");
        LSSVM_REQUIRE(world >= 1 && rank >= 0 && rank < world, "invalid shard descriptor");
        int b = 0, e = 0;
        const lssvm::Options opt = lssvm::options_snapshot();
        lssvm::shard_blocks(static_cast<int>((num_points - 1 + lssvm::TILE - 1) / lssvm::TILE), world, rank, symmetric != 0, b, e, &opt.shard_weights);
        *block_begin = b;
        *block_end = e;
    });
}

int lssvm_mi355_set_shard_weights(const double *weights, int count) {
    return guarded([&] {
        LSSVM_REQUIRE(count >= 0 && count <= 4096 && (count == 0 || weights != nullptr), "invalid shard weights");
        std::vector<double> w;
        for (int k = 0; k < count; ++k) {
            LSSVM_REQUIRE(std::isfinite(weights[k]) && weights[k] > 0.0, "shard weights must be positive and finite");
            w.push_back(weights[k]);
        }
        const std::lock_guard<std::mutex> lock(lssvm::options_mutex());
        lssvm::options().shard_weights = w;
    });
}

/* ---- communicator ---- */
int lssvm_mi355_comm_get_unique_id(unsigned char id_out[LSSVM_UNIQUE_ID_BYTES]) {
    return guarded([&] {
        LSSVM_REQUIRE(id_out != nullptr, "id_out must not be NULL");
        static_assert(sizeof(ncclUniqueId) == LSSVM_UNIQUE_ID_BYTES, "ncclUniqueId size changed");
        lssvm::comm_load();
        ncclUniqueId id;
        const ncclResult_t rc = lssvm::comm().pGetUniqueId(&id);
        if (rc != ncclSuccess) throw lssvm::Error(LSSVM_ERR_COMM, std::string("ncclGetUniqueId failed: ") + lssvm::comm().pGetErrorString(rc));
        std::memcpy(id_out, &id, sizeof(id));
    });
}
int lssvm_mi355_comm_init(int device, int rank, int world, const unsigned char id_in[LSSVM_UNIQUE_ID_BYTES]) {
    return guarded([&] {
        LSSVM_REQUIRE(id_in != nullptr, "id must not be NULL");
        LSSVM_REQUIRE(world >= 1 && rank >= 0 && rank < world, "invalid rank/world");
        lssvm::Comm &c = lssvm::comm();
        LSSVM_REQUIRE(c.comm == nullptr, "a communicator already exists in this process");
        lssvm::comm_load();
        lssvm::select_device_checked(device);
        ncclUniqueId id;
        std::memcpy(&id, id_in, sizeof(id));
        ncclComm_t nc = nullptr;
        const ncclResult_t rc = c.pCommInitRank(&nc, world, id, rank);
        if (rc != ncclSuccess) throw lssvm::Error(LSSVM_ERR_COMM, std::string("ncclCommInitRank failed: ") + c.pGetErrorString(rc));
        c.comm = nc;
        c.rank = rank;
        c.world = world;
        c.device = device;
    });
}
int lssvm_mi355_comm_library_path(char *buf, size_t buf_len) {
    return guarded([&] {
        LSSVM_REQUIRE(buf != nullptr && buf_len > 0, "buf must not be NULL");
        lssvm::comm_load();
        Dl_info di{};
        const char *path = dladdr(reinterpret_cast<void *>(lssvm::comm().pAllReduce), &di) != 0 && di.dli_fname != nullptr ? di.dli_fname : "";
        std::snprintf(buf, buf_len, "%s", path);
    });
}
int lssvm_mi355_comm_destroy(void) {
    return guarded([&] {
        lssvm::Comm &c = lssvm::comm();
        if (c.comm != nullptr) {
            (void) hipSetDevice(c.device);
            c.pCommDestroy(c.comm);
            c.comm = nullptr;
            c.world = 1;
            c.rank = 0;
        }
    });
}

/* ---- resident problem ---- */
int lssvm_mi355_problem_create(lssvm_mi355_problem **out, const lssvm_params *params, int dtype, const void *X, int mem_kind, size_t num_points,
                               size_t num_features, int device, const lssvm_shard *shard, const lssvm_mi355_options *options) {
    return guarded([&] {
        LSSVM_REQUIRE(out != nullptr, "out must not be NULL");
        *out = nullptr;
        lssvm::check_params(params);
        LSSVM_REQUIRE(dtype == LSSVM_DTYPE_F32 || dtype == LSSVM_DTYPE_F64, "dtype must be LSSVM_DTYPE_F32 or LSSVM_DTYPE_F64");
        auto h = std::make_unique<Handle>();
        lssvm::select_device_checked(device);
        if (dtype == LSSVM_DTYPE_F32) {
            h->impl = std::make_unique<lssvm::Solver<float>>(options_of(options), *params, X, mem_kind, num_points, num_features, std::vector<int>{ device }, shard);
        } else {
            h->impl = std::make_unique<lssvm::Solver<double>>(options_of(options), *params, X, mem_kind, num_points, num_features, std::vector<int>{ device }, shard);
        }
        *out = reinterpret_cast<lssvm_mi355_problem *>(h.release());
    });
}
int lssvm_mi355_problem_create_multi(lssvm_mi355_problem **out, const lssvm_params *params, int dtype, const void *X, int mem_kind, size_t num_points,
                                     size_t num_features, const int *devices, int num_devices, const lssvm_mi355_options *options) {
    return guarded([&] {
        LSSVM_REQUIRE(out != nullptr, "out must not be NULL");
        *out = nullptr;
        lssvm::check_params(params);
        LSSVM_REQUIRE(dtype == LSSVM_DTYPE_F32 || dtype == LSSVM_DTYPE_F64, "dtype must be LSSVM_DTYPE_F32 or LSSVM_DTYPE_F64");
        LSSVM_REQUIRE(num_points >= 2, "The data must contain at least two data points!");
        const std::vector<int> devs = lssvm::resolve_devices(devices, num_devices, num_points);
        auto h = std::make_unique<Handle>();
        if (dtype == LSSVM_DTYPE_F32) {
            h->impl = std::make_unique<lssvm::Solver<float>>(options_of(options), *params, X, mem_kind, num_points, num_features, devs, nullptr);
        } else {
            h->impl = std::make_unique<lssvm::Solver<double>>(options_of(options), *params, X, mem_kind, num_points, num_features, devs, nullptr);
        }
        *out = reinterpret_cast<lssvm_mi355_problem *>(h.release());
    });
}
int lssvm_mi355_problem_ipc_export(lssvm_mi355_problem *p, void *blob_out, size_t blob_bytes) {
    return guarded([&] { impl_of(p)->ipc_export(blob_out, blob_bytes); });
}
int lssvm_mi355_problem_ipc_connect(lssvm_mi355_problem *p, const void *blobs, size_t total_bytes) {
    return guarded([&] { impl_of(p)->ipc_connect(blobs, total_bytes); });
}
int lssvm_mi355_problem_destroy(lssvm_mi355_problem *p) {
    return guarded([&] { delete reinterpret_cast<Handle *>(p); });
}
int lssvm_mi355_problem_get_q(lssvm_mi355_problem *p, void *q_out, double *QA_cost_out) {
    return guarded([&] { impl_of(p)->get_q(q_out, QA_cost_out); });
}
int lssvm_mi355_problem_matvec(lssvm_mi355_problem *p, const void *d, void *ret_inout, double add) {
    return guarded([&] { impl_of(p)->matvec(d, ret_inout, add); });
}
int lssvm_mi355_cg_begin(lssvm_mi355_problem *p, const void *y, double eps) {
    return guarded([&] { impl_of(p)->cg_begin(y, eps); });
}
int lssvm_mi355_cg_step(lssvm_mi355_problem *p, uint64_t iterations, int *done_out) {
    return guarded([&] { impl_of(p)->cg_step(iterations, done_out); });
}
int lssvm_mi355_cg_finish(lssvm_mi355_problem *p, void *alpha_out, double *rho_out, lssvm_cg_info *info) {
    return guarded([&] { impl_of(p)->cg_finish(alpha_out, rho_out, info); });
}
int lssvm_mi355_problem_synchronize(lssvm_mi355_problem *p) {
    return guarded([&] { impl_of(p)->synchronize(); });
}
int lssvm_mi355_problem_rebalance(lssvm_mi355_problem *p, const double *weights, int count, int *changed_out) {
    return guarded([&] {
        LSSVM_REQUIRE(count >= 0 && (weights == nullptr) == (count == 0), "weights and count go together (NULL, 0: measured shares)");
        const int changed = impl_of(p)->rebalance(weights, count);
        if (changed_out != nullptr) *changed_out = changed;
    });
}
int lssvm_mi355_problem_info(lssvm_mi355_problem *p, lssvm_cg_info *info) {
    return guarded([&] {
        LSSVM_REQUIRE(info != nullptr, "info must not be NULL");
        impl_of(p)->fill_info(info);
    });
}

int lssvm_mi355_measure_bf16_mfma_ceiling(int device, int b_from_lds, double settle_ms, double *tflops_out, double *clock_ghz_out, double *nominal_tflops_out) {
    return guarded([&] {
        LSSVM_REQUIRE(settle_ms >= 0.0 && settle_ms <= 60000.0, "settle_ms must lie in [0, 60000]");
        lssvm::measure_bf16_mfma_ceiling(device, b_from_lds, settle_ms, tflops_out, clock_ghz_out, nominal_tflops_out);
    });
}

int lssvm_mi355_set_option(const char *name, int64_t value) {
    return guarded([&] {
        LSSVM_REQUIRE(name != nullptr, "name must not be NULL");
        const std::lock_guard<std::mutex> lock(lssvm::options_mutex());
        set_option_in(lssvm::options(), name, value);
    });
}
int lssvm_mi355_get_option(const char *name, int64_t *value_out) {
    return guarded([&] {
        LSSVM_REQUIRE(name != nullptr && value_out != nullptr, "name / value_out must not be NULL");
        const std::lock_guard<std::mutex> lock(lssvm::options_mutex());
        get_option_from(lssvm::options(), name, value_out);
    });
}

/* ---- options of one caller (ABI 4) ---- */
int lssvm_mi355_options_create(lssvm_mi355_options **out) {
    return guarded([&] {
        LSSVM_REQUIRE(out != nullptr, "out must not be NULL");
        *out = new lssvm_mi355_options{ lssvm::options_snapshot() };
    });
}
int lssvm_mi355_options_set(lssvm_mi355_options *options, const char *name, int64_t value) {
    return guarded([&] {
        LSSVM_REQUIRE(options != nullptr && name != nullptr, "options / name must not be NULL");
        set_option_in(options->o, name, value);
    });
}
int lssvm_mi355_options_get(const lssvm_mi355_options *options, const char *name, int64_t *value_out) {
    return guarded([&] {
        LSSVM_REQUIRE(options != nullptr && name != nullptr && value_out != nullptr, "options / name / value_out must not be NULL");
        get_option_from(options->o, name, value_out);
    });
}
int lssvm_mi355_options_destroy(lssvm_mi355_options *options) {
    return guarded([&] { delete options; });
}

/* The process-wide option defaults can be preset from the environment, LSSVM_MI355_OPTIONS="name=value,name=value" (applied once when the
 * library is loaded; unknown names or bad values are reported on stderr and ignored): lets a whole test suite or an unmodified host
 * program run against another kernel selection. */
namespace {
struct EnvOptions {
    EnvOptions() {
        const char *env = std::getenv("LSSVM_MI355_OPTIONS");
        if (env == nullptr) return;
        std::string all(env);
        size_t pos = 0;
        while (pos < all.size()) {
            const size_t end = std::min(all.find(',', pos), all.size());
            const std::string item = all.substr(pos, end - pos);
            pos = end + 1;
            const size_t eq = item.find('=');
            if (eq == std::string::npos) continue;
            char *stop = nullptr;
            const long long value = std::strtoll(item.c_str() + eq + 1, &stop, 10);
            if (lssvm_mi355_set_option(item.substr(0, eq).c_str(), value) != LSSVM_SUCCESS) {
                std::fprintf(stderr, "libplssvm_amd: LSSVM_MI355_OPTIONS: %s\n", lssvm_mi355_last_error());
            }
        }
    }
};
const EnvOptions g_env_options;
}  // namespace

}  // extern "C"
