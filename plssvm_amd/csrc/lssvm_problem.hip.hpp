/*
 * lssvm_problem.hip.hpp -- host side of the MI355X LS-SVM CG backend: device-resident problem + CG driver.
 *
 * Replaces (citations relative to the reference tree):
 *   gpu_csvm::setup_data_on_device        include/plssvm/backends/gpu_csvm.hpp:302-346   (SoA + 96 pad rows  -> row-major, k-chunk padded)
 *   gpu_csvm::generate_q                  gpu_csvm.hpp:349-384
 *   gpu_csvm::run_device_kernel           gpu_csvm.hpp:431-447
 *   gpu_csvm::device_reduction            gpu_csvm.hpp:449-475                           (host-staged sum -> RCCL all-gather over xGMI)
 *   gpu_csvm::solve_system_of_linear_equations_impl   gpu_csvm.hpp:477-654               (host BLAS-1 + 3 PCIe copies / iteration
 *                                                                                          -> everything device resident, one 8-byte read-back)
 * with the CG recipe of src/plssvm/backends/OpenMP/csvm.cpp:71-183 (x0 = 1, residual refresh every 50 iterations,
 * stop test delta <= eps^2 delta0 before the direction update, bias / rho / alpha_N epilogue).
 */
#pragma once

#include "lssvm_types.hpp"

#include "../../include/plssvm_amd.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is dlopen'ed lazily (single-GPU use needs no RCCL)

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace lssvm {

/* ------------------------------------------------------------------ errors ------------------------------------------------------------------ */
struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string &msg) : std::runtime_error(msg), status(st) {}
};

#define LSSVM_HIP_CHECK(expr)                                                                                                         \
    do {                                                                                                                              \
        const hipError_t lssvm_err_ = (expr);                                                                                         \
        if (lssvm_err_ != hipSuccess) {                                                                                               \
            throw ::lssvm::Error(lssvm_err_ == hipErrorOutOfMemory ? LSSVM_ERR_OUT_OF_MEMORY : LSSVM_ERR_HIP,                         \
                                 std::string("HIP assert '") + hipGetErrorName(lssvm_err_) + "' (" + std::to_string((int) lssvm_err_) + \
                                     "): " + hipGetErrorString(lssvm_err_) + " at " #expr);                                           \
        }                                                                                                                             \
    } while (0)

#define LSSVM_REQUIRE(cond, msg)                                              \
    do {                                                                      \
        if (!(cond)) throw ::lssvm::Error(LSSVM_ERR_INVALID_ARGUMENT, (msg)); \
    } while (0)

/* ------------------------------------------------------------------ options ------------------------------------------------------------------ */
struct Options {
    int64_t rbf_form = 0;        // 0: norm expansion on the matrix cores, 1: direct (x_i - x_j)^2 on the vector ALU (fp32 only)
    int64_t j_chunk_tiles = 0;   // 128-column tiles per work item; 0 = automatic (2 ... 16, about 4096 work items per device)
    int64_t symmetric = 1;         // 1: evaluate only the tiles on/below the diagonal and mirror them (fp32 v2 kernel), 0: full square
    int64_t tile_kernel = 0;       // 0: automatic (v2 'resident row panel' kernels when num_features <= 512 in fp32 / 256 in fp64), 1: always the generic v1 kernel
    int64_t xcd_map = 0;           // 1: XCD-aware work item mapping (8 x 8 super-tiles per XCD), 0: linear (default: measured equal, better balanced)
    int64_t lds_extra_kb = 0;      // experiment knob: extra dynamic LDS per workgroup of the fp32 v2 kernel (lowers workgroups per CU)
    int64_t debug_ablate = 0;      // diagnostic timing ablations of the fp32 tile kernel (results are wrong when != 0)
    int64_t item_order = 1;        // symmetric variant, order of the work items: 0 column-chunk major, 1 = 0 with the short (diagonal) items moved to the end, longest first
    int64_t gram_mode = 1;         // fp32, <= 256 features: 1 = exact 3-way bf16 split of the operands, six plane products on the bf16 MFMA (default), 0 = v_mfma_f32
    int64_t colslab_limit_mb = 98304;  // symmetric variant only while its column slab (per device) stays below this many MiB (96 GiB of the 288 GB)
    int64_t force_collective = 0;
    int64_t skip_collective = 0;   // testing aid: sharded problems (world > 1) need no communicator and leave their PARTIAL K*v un-exchanged  // testing aid: run the all-gather even for world == 1 (needs lssvm_mi355_comm_init(.., 0, 1, ..))
};
Options &options();

/* ------------------------------------------------------------------ RCCL (lazy) ------------------------------------------------------------------ */
struct Comm {
    void *lib = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    decltype(&ncclGetUniqueId) pGetUniqueId = nullptr;
    decltype(&ncclCommInitRank) pCommInitRank = nullptr;
    decltype(&ncclCommDestroy) pCommDestroy = nullptr;
    decltype(&ncclAllGather) pAllGather = nullptr;
    decltype(&ncclAllReduce) pAllReduce = nullptr;
    decltype(&ncclGetErrorString) pGetErrorString = nullptr;
};
Comm &comm();
void comm_load();

inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

/* ------------------------------------------------------------------ small RAII helpers ------------------------------------------------------------------ */
template <typename U>
struct DevBuf {
    U *p = nullptr;
    size_t count = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void alloc_zero(size_t n, hipStream_t s) {
        release();
        count = n;
        if (n == 0) return;
        LSSVM_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(U)));
        LSSVM_HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(U), s));
    }
    void release() {
        if (p != nullptr) (void) hipFree(p);
        p = nullptr;
        count = 0;
    }
};

inline int round_up(long v, int m) { return static_cast<int>(((v + m - 1) / m) * m); }

template <typename T>
constexpr int kchunk_of() {
    return std::is_same_v<T, float> ? F32_KC : F64_KC;
}

/* Padded feature count: a multiple of the k-chunk; between 8 and 16 chunks a multiple of TWO chunks, so that the v2 tile kernels
 * (instantiated for 1..8 and 10, 12, 14, 16 chunks) cover num_features <= 512 in fp32 / <= 256 in fp64. */
template <typename T>
inline int padded_features(size_t nfeat) {
    const int kc = kchunk_of<T>();
    const int ldx = round_up(static_cast<long>(nfeat), kc);
    return (ldx > 8 * kc && ldx <= 16 * kc) ? round_up(static_cast<long>(nfeat), 2 * kc) : ldx;
}

/* A dense row-major point set in HBM: rows padded to a multiple of 128, features padded to a multiple of the k-chunk
 * (zeros), so that every global load of the tile kernel is an aligned 16-byte load of a full 128-byte line. */
template <typename T>
struct DeviceMatrix {
    DevBuf<T> data;
    int rows = 0;        // valid rows
    int rows_alloc = 0;  // multiple of TILE
    int dfeat = 0;       // valid features
    int ldx = 0;         // padded features

    void upload(const void *src, int mem_kind, size_t nrows, size_t nfeat, size_t min_rows_alloc, hipStream_t s) {
        rows = static_cast<int>(nrows);
        dfeat = static_cast<int>(nfeat);
        ldx = padded_features<T>(nfeat);
        rows_alloc = std::max(round_up(static_cast<long>(nrows), TILE), round_up(static_cast<long>(min_rows_alloc), TILE));
        data.alloc_zero(static_cast<size_t>(rows_alloc) * ldx, s);
        LSSVM_HIP_CHECK(hipMemcpy2DAsync(data.p, static_cast<size_t>(ldx) * sizeof(T), src, nfeat * sizeof(T), nfeat * sizeof(T), nrows,
                                         mem_kind == LSSVM_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    }
};

/* ------------------------------------------------------------------ tile kernel launch ------------------------------------------------------------------ */
template <typename T>
void launch_tile_kernel(TileArgs<T> &a, int kernel_type, bool rbf_direct, int num_jc, hipStream_t s);
template <>
void launch_tile_kernel<float>(TileArgs<float> &a, int kernel_type, bool rbf_direct, int num_jc, hipStream_t s);   // tile_launch_f32.hip
template <>
void launch_tile_kernel<double>(TileArgs<double> &a, int kernel_type, bool rbf_direct, int num_jc, hipStream_t s);  // tile_launch_f64.hip

/* centre `M` (and optionally `M2` with the same means) by the column means of M's valid rows; rbf only */
template <typename T>
void center_columns(DeviceMatrix<T> &M, DeviceMatrix<T> *M2, T scale, hipStream_t s);
template <typename T>
void half_neg_norms(const DeviceMatrix<T> &M, DevBuf<T> &c, hipStream_t s);
template <typename T>
void interleave_features(DeviceMatrix<T> &M, hipStream_t s);
bool v2_eligible(int ldx, bool rbf_direct);
void split_bf16_planes(const float *X, int ldx, int dfeat, size_t rows, int ldx16, uint16_t *planes, size_t plane_stride, hipStream_t s);  // tile_launch_f32s.hip
bool v2_eligible_f64(int ldx);
int sym_block_boundary(int num_tiles, int r, int world);

/* ------------------------------------------------------------------ the resident problem ------------------------------------------------------------------ */
struct ProblemBase {
    int dtype = 0;
    virtual ~ProblemBase() = default;
    virtual void get_q(void *q_out, double *QA_cost_out) = 0;
    virtual void matvec(const void *d, void *ret_inout, double add) = 0;
    virtual void cg_begin(const void *y, double eps) = 0;
    virtual void cg_step(uint64_t iterations, int *done_out) = 0;
    virtual void cg_finish(void *alpha_out, double *rho_out, lssvm_cg_info *info) = 0;
    virtual void synchronize() = 0;
    virtual void fill_info(lssvm_cg_info *info) = 0;
};

template <typename T>
class Problem final : public ProblemBase {
  public:
    Problem(const lssvm_params &params, const void *X, int mem_kind, size_t num_points, size_t num_features, int device, const lssvm_shard *shard);
    ~Problem() override;

    void get_q(void *q_out, double *QA_cost_out) override;
    void matvec(const void *d, void *ret_inout, double add) override;
    void cg_begin(const void *y, double eps) override;
    void cg_step(uint64_t iterations, int *done_out) override;
    void cg_finish(void *alpha_out, double *rho_out, lssvm_cg_info *info) override;
    void synchronize() override;
    void fill_info(lssvm_cg_info *info) override;

  private:
    void apply_K(const T *v_dev);  // Kv_ <- K * v  (all rows, after the all-gather)
    void sum_and_qdot(const T *v_dev, int slot_sum, int slot_q);
    void drain_events();
    TileArgs<T> tile_args(const T *v_dev) const;

    lssvm_params params_{};
    int device_ = 0;
    int rank_ = 0, world_ = 1;
    hipStream_t stream_ = nullptr;

    size_t N_ = 0;  // data points
    int n_ = 0;     // N - 1
    int num_tiles_ = 0;  // ceil(n / TILE): row blocks == column tiles
    int ib_begin_ = 0, num_ib_ = 0, ib_per_rank_ = 0;
    int jc_tiles_ = 16, num_jc_ = 1;
    int nvec_ = 0;  // allocated vector length (multiple of TILE * world)
    bool rbf_direct_ = false;
    bool poly_prescaled_ = false;
    DevBuf<uint16_t> planes_;      // gram_mode 1: X as three bf16 planes [3][rows_alloc][ldx16]
    int ldx16_ = 0;  // fp64 polynomial on the v2 kernel: X_ carries sqrt(gamma), the kernel sees gamma = 1

    DeviceMatrix<T> X_;
    DevBuf<T> c_;  // -0.5 |x|^2 (rbf, centred data)
    DevBuf<T> q_, b_, x_, r_, d_, Ad_, Kv_, tmp_, ylast_;
    DevBuf<T> partial_;
    DevBuf<T> dc_;  // fp32 v2 kernel: packed (d_j | c_j) records
    // symmetric variant
    bool sym_ = false;
    DevBuf<int2> items_;
    int num_items_ = 0;
    DevBuf<T> colslab_;
    long pair_origin_ = 0;
    DevBuf<double> part_, sc_;
    double *host_sc_ = nullptr;  // pinned, SC_COUNT doubles
    double QA_cost_ = 0.0;
    double inv_cost_ = 1.0;
    double y_last_ = 0.0;

    // CG state
    double eps_ = 0.0;
    double delta0_ = 0.0, delta_ = 0.0;
    uint64_t iter_ = 0;        // iterations done
    bool converged_ = false;
    bool begun_ = false;

    // statistics
    double setup_ms_ = 0.0, cg_wall_ms_ = 0.0, cg_t0_ = 0.0;
    double matvec_ms_ = 0.0;
    uint64_t matvec_launches_ = 0;
    struct EvPair {
        hipEvent_t a = nullptr, b = nullptr;
        bool pending = false;
    };
    std::vector<EvPair> events_;
};

/* one-shot helpers used by the C ABI */
template <typename T>
void predict_values(const lssvm_params &params, const T *sv, size_t nsv, size_t nfeat, const T *alpha, T rho, T *w_inout, int *w_valid, const T *points,
                    size_t npoints, T *out);
template <typename T>
void calculate_w(const T *sv, size_t nsv, size_t nfeat, const T *alpha, T *w_out);

void check_params(const lssvm_params *params);
int select_device_checked(int device);

}  // namespace lssvm
