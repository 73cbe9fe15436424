/*
 * lssvm_problem.hip.hpp -- host side of the MI355X LS-SVM CG backend: device-resident problem + CG driver.
 *
 * Replaces (citations relative to the reference tree):
 *   gpu_csvm::setup_data_on_device        include/plssvm/backends/gpu_csvm.hpp:302-346   (SoA + 96 pad rows  -> row-major, k-chunk padded)
 *   gpu_csvm::generate_q                  gpu_csvm.hpp:349-384
 *   gpu_csvm::run_device_kernel           gpu_csvm.hpp:431-447
 *   gpu_csvm::device_reduction            gpu_csvm.hpp:449-475                           (host-staged sum -> RCCL all-reduce / peer kernels over xGMI)
 *   gpu_csvm::solve_system_of_linear_equations_impl   gpu_csvm.hpp:477-654               (host BLAS-1 + 3 PCIe copies / iteration
 *                                                                                          -> everything device resident, one 8-byte read-back)
 * with the CG recipe of src/plssvm/backends/OpenMP/csvm.cpp:71-183 (x0 = 1, residual refresh every 50 iterations,
 * stop test delta <= eps^2 delta0 before the direction update, bias / rho / alpha_N epilogue).
 */
#pragma once

#include "lssvm_error.hpp"
#include "lssvm_types.hpp"

#include "../../include/plssvm_amd.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is dlopen'ed lazily (single-GPU use needs no RCCL)

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace lssvm {

/* ------------------------------------------------------------------ errors (lssvm::Error, LSSVM_REQUIRE: lssvm_error.hpp) ------------------------------------------------------------------ */
#define LSSVM_HIP_CHECK(expr)                                                                                                         \
    do {                                                                                                                              \
        const hipError_t lssvm_err_ = (expr);                                                                                         \
        if (lssvm_err_ != hipSuccess) {                                                                                               \
            throw ::lssvm::Error(lssvm_err_ == hipErrorOutOfMemory ? LSSVM_ERR_OUT_OF_MEMORY : LSSVM_ERR_HIP,                         \
                                 std::string("HIP assert '") + hipGetErrorName(lssvm_err_) + "' (" + std::to_string((int) lssvm_err_) + \
                                     "): " + hipGetErrorString(lssvm_err_) + " at " #expr);                                           \
        }                                                                                                                             \
    } while (0)

/* ------------------------------------------------------------------ options ------------------------------------------------------------------ */
struct Options {
    // the 14 options of the product (lssvm_mi355_set_option; include/plssvm_amd.h documents them; round 4 retired xcd_map, lds_extra_kb, item_order,
    // linear_panel_features, check_shards, rbf_direct_above and mfma_shape = 1 -- measured, decided, now constants below)
    int64_t rbf_form = 0;        // fp32 rbf: 0 automatic (matrix cores: norm expansion up to the exponent scale RBF_DIRECT_ABOVE, grid planes up to RBF_GRID_MAX_R2 (times sqrt(128 / features) beyond 128 features); else direct), 3 = grid planes where they exist, 1 always the direct
                                 // (x_i - x_j)^2 kernel on the vector ALU, 2 always the norm expansion on the matrix cores
    int64_t rbf_fold = 1;        // fp32 rbf on the split kernels: 1 = folded column records (2^c_j d_j | 2^c_j), accumulators start from c_i as the C
                                 // operand of their first MFMA (default, while the exponent scale stays below 200); 0 = start values c_i + c_j by vector adds
    int64_t j_chunk_tiles = 0;   // 128-column tiles per work item; 0 = automatic (see Problem<T>'s constructor)
    std::vector<double> shard_weights;  // symmetric variant, several ranks: rank r's share of the triangle's area is weight[r] / sum (empty or of another length: equal shares); lssvm_mi355_set_shard_weights
    int64_t j_chunk_head = 1;    // 256-row workgroups: 0 = none; 1 = chosen by the replayed dispatch (with j_chunk_tiles = 0; default); 1024 count + tiles = the first `count` column chunks have `tiles` tiles
    int64_t symmetric = 1;         // 1: evaluate only the tiles on/below the diagonal and mirror them, 0: full square
    int64_t tile_kernel = 0;       // 0: automatic (resident-row-panel kernels where they exist), 1: always the generic v1 kernel (the cross-checks' yardstick)
    int64_t gram_mode = 3;         // fp32 Gram tiles: 0 = v_mfma_f32 chains; 1 = "bf16x6"; 2 = "f16x3" without the representability check; 3 (default) = f16x3
                                   // where the data passes that check, else bf16x6
    int64_t mfma_shape = 3;        // split kernels, symmetric variant, <= 128 features per pass: 2 = 128-row workgroups (four waves), 3 (default) = 256-row
                                   // workgroups on block pairs (eight waves, one column stream per CU) from 64 row blocks on
    int64_t colslab_band_mb = 2048;    // symmetric variant: the column-sum records of ONE row-block band may take this many MiB; the tile kernel runs band by band
    int64_t colslab_limit_mb = 98304;  // symmetric variant only while its column slab (per device) stays below this many MiB (96 GiB of the 288 GB); 0 = off
    int64_t force_collective = 0;  // testing aid: run the per-matvec collective even for a world of one (needs lssvm_mi355_comm_init(.., 0, 1, ..))
    int64_t skip_collective = 0;   // testing aid: sharded problems (world > 1) need no communicator and leave their PARTIAL K*v un-exchanged
    int64_t exchange = 0;          // several devices in ONE process: 0 = automatic (RCCL when the devices are distinct, else peer kernels), 1 = RCCL all-reduce / all-gather
                                   // (ncclCommInitAll), 2 = peer kernels: every device sums the partial vectors of all devices over xGMI in rank order
    int64_t rebalance_after = 0;           // one-shot solves on several devices: after this many iterations the shards are given new shares by their measured pace (lssvm_mi355_problem_rebalance); 0 = never
    int64_t enqueue_ahead_below_us = 5000;  // CG: implicit matvecs shorter than this are enqueued ahead of the previous iteration's stop test (0 = never)
    int64_t ipc_timeout_s = 600;   // one process per GPU over HIP IPC: how long a rank waits for its peers at an exchange before it gives up
    // development builds only (the setter refuses them elsewhere)
    int64_t debug_ablate = 0;      // -DLSSVM_ENABLE_ABLATION: timing ablations of the fp32 tile kernels (results are wrong when != 0)
    int64_t item_order_dev = 0;    // make DEV=1: order of the work items of the symmetric variant (0 = ITEM_ORDER; 3 = XCD-aware lanes)
    int64_t pair_lag = 0;          // make DEV=1: plane-chunk steps waves 4-7 of a 256-row workgroup run behind waves 0-3 (0 = lock step: the shipped form; 1, 3
                                   // measured slower)
};
constexpr double RBF_GRID_MAX_R2 = 4096.0;    // rbf_form 0: grid planes (KT_RBFG) between RBF_DIRECT_ABOVE and this exponent scale (at any width): their error grows with the cross
                                              // terms |h||s| ~ R2 sqrt(d) 2^-12 (7 eps of a row's summands at R2 = 12 600, d = 128 in the model of tests/tools/grid_planes_model.py)
constexpr double RBF_DIRECT_ABOVE = 32.0;     // rbf_form 0: the formula-exact kernel above this exponent scale 2 gamma log2(e) max|x - mean|^2 (absolute error of the
                                              // matrix-core exponent ~ 2^-24 x that; [-1, 1]-scaled data with gamma = 1 / num_features has <= 3)
constexpr int LINEAR_IN_TILE_BELOW = 10000;   // fp32 linear kernel on more than 256 features and fewer points than this: ONE launch of the polynomial tile kernels with degree 1
                                              // instead of launch-bound feature-panel passes (Problem<float>::tile_params_)
constexpr int LINEAR_PANEL_FEATURES = 128;    // linear kernel beyond this many features: one pass of the <= 128-feature kernels per feature panel (fp32 f16x3: beats the
                                              // wider one-wave kernels by 5 ... 12 % at every width measured; fp64: 64- and 256-feature panels within 2 %)
constexpr int ITEM_ORDER = 3;                 // symmetric variant: the work items of a column chunk on ONE XCD at a time (list position 8 k + x = lane x; the hardware deals
                                              // workgroups round-robin over the 8 XCDs), the items cut short by the diagonal last, longest first.  Against plain
                                              // column-chunk major order (1): bit-identical results, fabric traffic 116 -> 97 GB per matvec at 1 000 000 x 128, time -0.3 %
                                              // (profiles/r04_ab_xcd_item_order.log)
/* process-wide DEFAULTS (lssvm_mi355_set_option); every problem / solve that is not handed options of its own (lssvm_mi355_options, ABI 4) takes a snapshot when it is
 * created.  The defaults are read and written under options_mutex() only: options_snapshot() for readers. */
Options &options();
std::mutex &options_mutex();
inline Options options_snapshot() {
    const std::lock_guard<std::mutex> lock(options_mutex());
    return options();
}
/* fp32 rbf, rbf_form 0 (automatic): the grid planes were chosen from the exponent scale but do not represent THIS data -- the owner of the problem builds it again
 * with the formula-exact kernel (Solver's constructor, predict_values); with an explicit rbf_form = 3 the error reaches the caller */
struct GridPlanesUnfit : Error {
    GridPlanesUnfit() : Error(LSSVM_ERR_INTERNAL, "the grid planes of the rbf kernel do not represent this data (option rbf_form = 1 selects the direct kernel)") {}
};

/* ------------------------------------------------------------------ RCCL (lazy) ------------------------------------------------------------------ */
struct Comm {  // the RCCL entry points (dlopen'ed once) + the communicator of a one-process-per-GPU launch
    void *lib = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    decltype(&ncclGetUniqueId) pGetUniqueId = nullptr;
    decltype(&ncclCommInitRank) pCommInitRank = nullptr;
    decltype(&ncclCommInitAll) pCommInitAll = nullptr;
    decltype(&ncclCommDestroy) pCommDestroy = nullptr;
    decltype(&ncclAllGather) pAllGather = nullptr;
    decltype(&ncclAllReduce) pAllReduce = nullptr;
    decltype(&ncclGroupStart) pGroupStart = nullptr;
    decltype(&ncclGroupEnd) pGroupEnd = nullptr;
    decltype(&ncclGetErrorString) pGetErrorString = nullptr;
    decltype(&ncclCommCount) pCommCount = nullptr;
    decltype(&ncclCommCuDevice) pCommCuDevice = nullptr;
    decltype(&ncclCommUserRank) pCommUserRank = nullptr;
};
Comm &comm();
void comm_load();
void nccl_check(ncclResult_t rc, const char *what);  // throws LSSVM_ERR_COMM with RCCL's own message
constexpr int MAX_LOCAL_DEVICES = 16;                // shards one process drives / ranks the peer exchange sums

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

/* Padded feature count: a multiple of the k-chunk; between 8 (fp32: 4) and 16 chunks a multiple of TWO chunks, so that the v2 tile kernels
 * (fp64: instantiated for 1..8 and 10, 12, 14, 16 chunks; fp32: 1..4, 6, 8, 10, 12, 14, 16) cover num_features <= 512 in fp32 / <= 256 in fp64. */
template <typename T>
inline int padded_features(size_t nfeat) {
    const int kc = kchunk_of<T>();
    const int ldx = round_up(static_cast<long>(nfeat), kc);
    if (std::is_same_v<T, double> && ldx > 16 * kc) return round_up(static_cast<long>(nfeat), 64);  // fp64 beyond the one-pass kernels: whole feature panels of 64 (lssvm_tile_f64_wide.hip.hpp)
    // fp32: whole pairs of chunks already beyond 4 (the native v2 kernel -- the rare path without operand planes -- is instantiated for 1 ... 4, 6, 8 ... 16)
    const int single_up_to = std::is_same_v<T, float> ? 4 : 8;
    return (ldx > single_up_to * kc && ldx <= 16 * kc) ? round_up(static_cast<long>(nfeat), 2 * kc) : ldx;
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
                                         mem_kind == LSSVM_MEM_DEVICE ? hipMemcpyDefault : hipMemcpyHostToDevice, s));  // (a device source may live on another device of the process)
    }
};

/* ------------------------------------------------------------------ more RAII: stream, events, pinned host words ------------------------------------------------------------------ */
/* A non-blocking stream of the CURRENT device.  Creating one costs 2-4 ms on this runtime (a hardware queue behind it) -- more than everything else in the set-up of a
 * 50 000-point problem -- so an idle stream goes back to a small per-device pool instead of being destroyed, and the next problem takes it from there. */
struct Stream {
    hipStream_t s = nullptr;
    int device = -1;
    Stream() = default;
    Stream(const Stream &) = delete;
    Stream &operator=(const Stream &) = delete;
    ~Stream() {
        if (s == nullptr) return;
        if (hipStreamSynchronize(s) == hipSuccess) {  // (only an idle stream without a pending error is kept)
            const std::lock_guard<std::mutex> lock(pool_mutex());
            std::vector<hipStream_t> &idle = pool()[device];
            if (idle.size() < 8) {
                idle.push_back(s);
                return;
            }
        }
        (void) hipStreamDestroy(s);
    }
    void create() {
        LSSVM_HIP_CHECK(hipGetDevice(&device));
        {
            const std::lock_guard<std::mutex> lock(pool_mutex());
            std::vector<hipStream_t> &idle = pool()[device];
            if (!idle.empty()) {
                s = idle.back();
                idle.pop_back();
                return;
            }
        }
        LSSVM_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    }

  private:
    static std::mutex &pool_mutex() {
        static std::mutex m;
        return m;
    }
    static std::map<int, std::vector<hipStream_t>> &pool() {
        static std::map<int, std::vector<hipStream_t>> *p = new std::map<int, std::vector<hipStream_t>>();  // (never destroyed: streams outlive static destruction order)
        return *p;
    }
};
struct Event {
    hipEvent_t e = nullptr;
    Event() = default;
    Event(const Event &) = delete;
    Event &operator=(const Event &) = delete;
    Event(Event &&o) noexcept : e(o.e) { o.e = nullptr; }
    ~Event() {
        if (e != nullptr) (void) hipEventDestroy(e);
    }
    void create(bool timing) { LSSVM_HIP_CHECK(hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming)); }
};
template <typename U>
struct PinnedBuf {
    U *p = nullptr;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() {
        if (p != nullptr) (void) hipHostFree(p);
    }
    void alloc(size_t n) { LSSVM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&p), n * sizeof(U), hipHostMallocDefault)); }
    /* host memory a KERNEL writes (fine-grained, mapped): `dev` is the address the device uses; the host reads `p` behind an event of the stream */
    U *dev = nullptr;
    void alloc_mapped(size_t n) {
        LSSVM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&p), n * sizeof(U), hipHostMallocMapped | hipHostMallocCoherent));
        LSSVM_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&dev), p, 0));
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
bool v2_eligible(const Options &o, int ldx, bool rbf_direct);
void split_bf16_planes(const float *X, int ldx, int dfeat, size_t rows, int ldx16, uint16_t *planes, size_t plane_stride, hipStream_t s);  // tile_launch_f32s.hip
/* the f16 planes of `scale` * X (scale = a power of two; shift = 0: two planes, shift > 0: the three shifted planes of the rbf kernels) + the representation
 * statistics {max rel^2, max |rest|^2, max |y|^2} as float bits */
void split_grid_planes(const float *X, int ldx, int dfeat, size_t rows, int ldx16, float g, float sigma, uint16_t *planes, size_t plane_stride, float *chg, float *efac, unsigned *stats,
                       hipStream_t s);  // tile_launch_f32h.hip
void split_f16_planes(const float *X, int ldx, int dfeat, size_t rows, int ldx16, float scale, int shift, uint16_t *planes, size_t plane_stride, unsigned *stats, hipStream_t s,
                      float *row_inv_scale = nullptr);  // row_inv_scale != NULL: a power-of-two scale per ROW, its inverse stored there  // tile_launch_f32h.hip
void absmax_f32(const float *X, int ldx, int dfeat, size_t rows, unsigned *out, hipStream_t s);  // tile_launch_f32h.hip
bool v2_eligible_f64(const Options &o, int ldx);
int sym_block_boundary(int num_tiles, int r, int world, const std::vector<double> *weights = nullptr);
void shard_blocks(int num_tiles, int world, int rank, bool symmetric, int &begin, int &end, const std::vector<double> *weights = nullptr);

/* fp32: a data matrix once more as operand planes of the split tile kernels (make_planes in lssvm_problem.hip) */
struct PlaneSet {
    DevBuf<float> row_inv_scale;     // f16x3, linear kernel, round 6: 2^-k_i per row where the planes carry a scale PER ROW (empty: one scale for the matrix)
    double f16_row_rel_error = -1.0;  // what the representability check of the f16 planes measured (make_planes), -1 where it did not run
    DevBuf<uint16_t> buf;
    int ldx16 = 0;
    int mode = 0;   // 0 none, 1 bf16x6 (three bf16 planes), 2 f16x3 (two f16 planes)
    int shift = 0;  // f16x3: the planes carry 2^shift x
    int nplanes = 0;  // planes in `buf` (f16x3: 2, rbf 3 -- the shifted planes; bf16x6: 3)
};

/* ------------------------------------------------------------------ one device's share of the problem ------------------------------------------------------------------ */
/* Problem<T>: the data matrix resident on ONE device plus the CG vectors, and the row blocks `rank` of `world` of the implicit
 * matrix.  Every method only ENQUEUES work on the shard's stream (set-up excepted); the CG recipe and the exchange between
 * shards are driven by Solver<T>.  Replaces gpu_csvm::setup_data_on_device / generate_q / run_device_kernel (gpu_csvm.hpp:302-447). */
template <typename T>
class Solver;

template <typename T>
class Problem {
  public:
    Problem(const Options &opt, const lssvm_params &params, const void *X, int mem_kind, size_t num_points, size_t num_features, int device, int rank, int world);
    ~Problem();
    Problem(const Problem &) = delete;
    Problem &operator=(const Problem &) = delete;

    void activate() const { LSSVM_HIP_CHECK(hipSetDevice(device_)); }
    /* Kv_ <- this shard's part of K * v (complete for a world of one): tile kernel + fixed-order reductions */
    void enqueue_apply_K_local(const T *v_dev, bool zero_first);
    void reshard(const std::vector<double> &weights);  // new shares of the triangle (lssvm_mi355_problem_rebalance): the data, the vectors and the CG state stay
    void choose_shard_geometry();
    void build_shard_lists(hipStream_t st);
    PackDc<T> pack_for_d(bool zero_first);  // what k_update_d needs to pack the records of d_ (dc == NULL: this problem packs per matvec); marks them as present
    void enqueue_sum_and_qdot(const T *v_dev, int slot_sum, int slot_q);
    void drain_events();
    hipStream_t stream() const { return stream_.s; }
    /* tile-kernel passes per row-block band: the feature panels of a wide fp32 linear problem, else 1 */
    int passes_per_matvec() const {  // feature panels of a wide linear problem: fp32 over the planes, fp64 over the row-major data
        if (!wide_linear_) return 1;
        if constexpr (std::is_same_v<T, float>) {
            return (planes_.ldx16 + LINEAR_PANEL_FEATURES - 1) / LINEAR_PANEL_FEATURES;
        } else {
            return (X_.ldx + LINEAR_PANEL_FEATURES - 1) / LINEAR_PANEL_FEATURES;
        }
    }

    /* full-square-equivalent rate (flop/s, 2 n^2 d per matvec) this shard's kernel path is expected to sustain -- a rule on the shape, the real type and
     * the options only (never a measurement: every rank of a sharded solve must come to the same number).  Round numbers from profiles/r03 / r04:
     * fp32 split kernels in the symmetric variant ~900 T (c5: 930, c3: 1 000), full square half of that; native fp32 v2 ~250 T (sym) / 125 T; panels
     * inside a tile ~250 T; generic fp32 / direct rbf ~45 T; fp64 v2 ~130 T (sym, c4: 134) / 65 T; fp64 panels ~110 T; generic fp64 ~45 T. */
    double nominal_full_square_rate() const {
        const double sym = sym_ ? 1.0 : 0.5;
        if constexpr (std::is_same_v<T, float>) {
            if (rbf_direct_) return 45e12;
            if (planes_.mode != 0 && dc_.p != nullptr) return (wide_nl_ ? 250e12 : 900e12) * sym;
            return dc_.p != nullptr ? 250e12 * sym : 45e12;
        } else {
            if (wide_nl_) return 110e12 * sym;
            return dc_.p != nullptr ? 130e12 * sym : 45e12;
        }
    }

  private:
    friend class Solver<T>;
    TileArgs<T> tile_args(const T *v_dev) const;

    Options opt_{};
    lssvm_params params_{};
    int wide_order_ = 0;            // panels inside a tile, symmetric variant: the row-group major item order (band_items 4 / 5: groups of four / two row blocks); 0 elsewhere
    DevBuf<T> Xfrag_;               // fp64 panels-inside-a-sub-tile kernel, symmetric variant: the data once more, fragment-major, for the row side (TileArgs::Xrf)
    DevBuf<uint16_t> planes_frag_;  // panels-inside-a-tile kernel, symmetric variant: the planes once more, fragment-major, for the row side (TileArgs::Xr16f)
    lssvm_params tile_params_{};   // what the TILE kernels evaluate: params_, except that a linear kernel on few points and many features runs as the polynomial
                                   // kernel of degree 1 (constructor; same values, one launch instead of feature-panel passes)
    int device_ = 0;
    int rank_ = 0, world_ = 1;
    Stream stream_;

    size_t N_ = 0;  // data points
    int n_ = 0;     // N - 1
    int num_tiles_ = 0;  // ceil(n / TILE): row blocks == column tiles
    int ib_begin_ = 0, num_ib_ = 0, ib_per_rank_ = 0;
    int jc_tiles_ = 16, num_jc_ = 1;
    int jc_head_tiles_ = 0, jc_head_count_ = 0;  // 256-row workgroups: a head of short column chunks (choose_pair_chunk)
    int nvec_ = 0;  // allocated vector length (multiple of TILE * world)
    bool rbf_direct_ = false;
    double rbf_r2_ = 0.0;  // fp32 rbf: 2 gamma log2(e) max|x - mean|^2
    bool rbf_grid_ = false;      // fp32 rbf on GRID planes (KT_RBFG): large exponent scales on the matrix cores at the direct form's accuracy (Problem<T>'s constructor)
    float grid_sigma_ = 1.0f;    // ... the planes carry sigma x (a power of two), the chain sigma^2 t
    DevBuf<T> efac_;             // ... E_i = 2^(c_i - ch_i) per row (c_ then holds sigma^2 ch_i)
    bool dc_folded_ = false;  // the (d_j | c_j) records carry (2^c_j d_j | 2^c_j): rbf on the 16x16x32 bf16x6 kernels
    bool pair_ = false;            // fp32 symmetric variant on the split kernels, <= 128 features per pass: 256-row workgroups on block pairs (lssvm_tile_f32_pair.hip.hpp)
    int part_blocks() const { return pair_ ? round_up(std::max(num_ib_, 1), 2) : std::max(num_ib_, 1); }  // row blocks of a row slab (whole pairs)
    double f16_row_rel_error_ = -1.0;  // fp32, gram_mode 2 / 3: the largest relative error of a row under two f16 planes, as measured at set-up (-1: not measured)
    bool row_scaled_ = false;        // fp32 linear kernel: the f16 planes carry a power-of-two scale per ROW (planes_.row_inv_scale): K v = D (Xs Xs^T) (D v)
    DevBuf<T> vs_;                   // ... D v, the vector the tile kernel multiplies
    bool f16_probe_failed_ = false;  // the probe for the linear kernel's panel passes found the data unfit for two f16 planes
    bool wide_nl_ = false;         // fp32 rbf / polynomial on more features than the one-pass split kernels take: feature panels inside a tile (lssvm_tile_f32_wide.hip.hpp)
    bool wide_linear_ = false;     // linear kernel over feature panels, one tile-kernel pass per panel: fp32 f16x3 beyond linear_panel_features, fp64 beyond 256 features
    bool poly_prescaled_ = false;  // fp64 polynomial on the v2 kernel: X_ carries sqrt(gamma), the kernel sees gamma = 1
    PlaneSet planes_;              // fp32 split kernels: X as three bf16 planes (bf16x6) or two f16 planes (f16x3), [planes][rows_alloc][ldx16]

    DeviceMatrix<T> X_;
    DevBuf<T> c_;  // -0.5 |x|^2 (rbf, centred data)
    DevBuf<T> q_, b_, x_, r_, d_, Ad_, Kv_, Ksum_, tmp_, ylast_;
    T *Kres_ = nullptr;  // the exchanged K * v the O(n) kernels read: Kv_ itself, or Ksum_ when peer kernels do the exchange
    DevBuf<T> partial_;
    DevBuf<T> dc_;  // v2 kernels: packed (d_j | c_j) records
    bool d_packed_ = false;  // dc_ holds the records of d_ and K*v is cleared: k_update_d left them (pack_for_d), the next implicit matvec of d_ launches no k_pack_dc
    // symmetric variant
    bool sym_ = false;
    DevBuf<int2> items_;
    int num_items_ = 0;
    int ldx_probe_ = 0;        // the padded feature count the kernel paths were chosen with (constructor), kept for reshard
    size_t num_features_ = 0;
    DevBuf<unsigned> queue_;  // 256-row workgroups, persistent launches: two sets of eight item counters (a launch draws from one and zeroes the other)
    int queue_set_ = 0;
    int queue_min_items_ = 256;  // launches of more items than this (the CU count) are persistent
    DevBuf<T> colslab_;  // the records of the band in flight
    struct Band {
        int ib_begin, ib_end;       // row blocks [begin, end) of the band (global block indices)
        int item_begin, item_count; // its work items in items_
        long pair_origin;           // record index of (ib_begin, 0) in the packed triangle: ib_begin (ib_begin - 1) / 2
    };
    std::vector<Band> bands_;
    DevBuf<double> part_, sc_;
    double *part(PartSet s) const { return part_.p + static_cast<size_t>(s) * RED_BLOCKS * 2; }  // (lssvm_types.hpp)
    PinnedBuf<double> host_sc_;  // SC_COUNT doubles
    PinnedBuf<double> host_delta_;  // one word: k_finish_delta stores the iteration's delta straight into host memory (no copy kernel per iteration)
    double QA_cost_ = 0.0;
    double inv_cost_ = 1.0;
    double setup_ms_ = 0.0;

    // statistics: HIP events around the tile kernel
    double matvec_ms_ = 0.0;       // tile-kernel time of the TIMED matvecs (HIP events around the band launches)
    uint64_t matvec_launches_ = 0; // matvecs enqueued
    double pace_ms0_ = 0.0;        // matvec_ms_ / matvec_timed_ at the last reshard: a rebalance by measured pace looks at what came after it
    uint64_t pace_timed0_ = 0;
    uint64_t matvec_timed_ = 0;    // ... of which timed: all of them where a matvec is long, every 8th where it is short (two event records cost ~6 us per
                                   // matvec on the stream: 9 % of a 10 000-point CG iteration, profiles/r04_event_record_cost.log)
    /* a rule on the shape only, like the enqueue-ahead decision: below ~1 ms per matvec the events are sampled */
    int event_stride() const {
        const double n_d = static_cast<double>(n_);
        return 2.0 * n_d * n_d * static_cast<double>(X_.dfeat) / static_cast<double>(world_) / nominal_full_square_rate() < 1e-3 ? 8 : 1;
    }
    struct EvPair {
        Event a, b;
        bool pending = false;
        bool first_of_matvec = true;
    };
    std::vector<EvPair> events_;
    Event ev_ready_, ev_consumed_;  // peer exchange: partial vector written / all partial vectors read
};

/* ------------------------------------------------------------------ the resident problem (what a C handle points to) ------------------------------------------------------------------ */
struct ProblemBase {
    int dtype = 0;
    virtual ~ProblemBase() = default;
    virtual void get_q(void *q_out, double *QA_cost_out) = 0;
    virtual void matvec(const void *d, void *ret_inout, double add) = 0;
    virtual void cg_begin(const void *y, double eps) = 0;
    virtual void cg_step(uint64_t iterations, int *done_out) = 0;
    virtual void cg_finish(void *alpha_out, double *rho_out, lssvm_cg_info *info) = 0;
    virtual void synchronize() = 0;
    virtual int rebalance(const double *weights, int count) = 0;  // lssvm_mi355_problem_rebalance: 1 if the shares changed
    virtual void fill_info(lssvm_cg_info *info) = 0;
    virtual void ipc_export(void *blob_out, size_t blob_bytes) = 0;
    virtual void ipc_connect(const void *blobs, size_t total_bytes) = 0;
};

/* ------------------------------------------------------------------ one process per GPU without RCCL: HIP IPC + host flags ------------------------------------------------------------------ */
/* What one rank hands to the others (plain bytes; the application moves them, like the RCCL unique id). */
constexpr size_t IPC_BLOB_BYTES = 256;
struct IpcBlob {
    uint32_t magic;
    int32_t rank, world, device;
    uint64_t nvec;
    uint32_t real_size;
    int32_t pid;
    hipIpcMemHandle_t mem;  // this rank's partial K*v vector
    char shm_name[64];      // this rank's flag page (POSIX shared memory)
};
static_assert(sizeof(IpcBlob) <= IPC_BLOB_BYTES, "IpcBlob must fit the published blob size");

/* one rank's flag page: sequence numbers of the implicit matvecs, written by the owner only */
struct IpcFlags {
    std::atomic<uint64_t> ready;     // matvec s: my partial vector is complete in HBM
    std::atomic<uint64_t> consumed;  // matvec s: I have read every peer's partial vector
    std::atomic<uint32_t> abort;     // I gave up (error / timeout): peers stop waiting
};

class IpcPeers {
  public:
    IpcPeers(int rank, int world);
    ~IpcPeers();
    IpcPeers(const IpcPeers &) = delete;
    IpcPeers &operator=(const IpcPeers &) = delete;
    void connect(const IpcBlob *blobs, void *own_vector);
    /* block until every peer's `ready` (which = 0) or `consumed` (which = 1) counter has reached seq */
    void wait_all(int which, uint64_t seq, double timeout_s);

    int rank, world;
    std::string own_name;
    IpcFlags *own = nullptr;
    std::vector<IpcFlags *> flags;  // [world], own included
    std::vector<void *> vectors;    // [world]: the partial K*v vectors, own = the local pointer
    bool connected = false;
};

/* communicators of the devices of ONE process (ncclCommInitAll); cached per device list, destroyed at exit */
struct LocalComms {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    ~LocalComms();
};
std::shared_ptr<LocalComms> local_comms_for(const std::vector<int> &devices);  // lssvm_exchange.hip

/* Solver<T>: the CG driver (csvm.cpp:71-183 / gpu_csvm.hpp:477-654) over the shards that live in THIS process:
 *   - one shard, world 1: single GPU;
 *   - one shard, rank r of a world of processes: one process per GPU (torchrun), exchange = the process communicator (RCCL);
 *   - D shards on D devices: single process, one stream per device driven by the calling thread, exchange = RCCL (ncclCommInitAll,
 *     group calls) or peer kernels over xGMI.  This is the mode behind plssvm::csvm (the reference drives all its devices from one
 *     process too, gpu_csvm.hpp:283-299, :574-593).
 * Every shard runs the identical O(n) kernels on the identical exchanged vector, so all CG scalars are bit-equal on all shards and
 * the host reads the stop criterion from shard 0 only. */
template <typename T>
class Solver final : public ProblemBase {
  public:
    /* `opt`: the options of THIS solve (the caller's lssvm_mi355_options, or a snapshot of the process defaults) */
    Solver(const Options &opt, const lssvm_params &params, const void *X, int mem_kind, size_t num_points, size_t num_features, const std::vector<int> &devices, const lssvm_shard *shard);
    ~Solver() override;

    void get_q(void *q_out, double *QA_cost_out) override;
    void matvec(const void *d, void *ret_inout, double add) override;
    void cg_begin(const void *y, double eps) override;
    void cg_step(uint64_t iterations, int *done_out) override;
    int rebalance(const double *weights, int count) override;
    void cg_finish(void *alpha_out, double *rho_out, lssvm_cg_info *info) override;
    void synchronize() override;
    void fill_info(lssvm_cg_info *info) override;
    void ipc_export(void *blob_out, size_t blob_bytes) override;
    void ipc_connect(const void *blobs, size_t total_bytes) override;

  private:
    enum class Exchange { none, process_rccl, local_rccl, peer, process_peer };
    enum class Vec { d, x, tmp };
    void apply_K(Vec which);  // every shard: Kres_ <- K * v (all rows)
    void exchange();
    void sync_all();
    PackDc<T> pack_with_direction(Problem<T> &p);
    T *vec_of(Problem<T> &p, Vec which) const { return which == Vec::d ? p.d_.p : (which == Vec::x ? p.x_.p : p.tmp_.p); }

    Options opt_{};
    std::vector<std::unique_ptr<Problem<T>>> shards_;
    Exchange exchange_ = Exchange::none;
    std::shared_ptr<LocalComms> local_comms_;
    std::unique_ptr<IpcPeers> ipc_;  // Exchange::process_peer
    uint64_t xseq_ = 0;              // implicit matvecs exchanged so far
    Event ev_delta_;                 // shard 0's stream: delta of the iteration is on the host
    int world_ = 1;  // shards of the problem in total (all processes)
    int info_shard_ = -1;  // the shard whose tile-kernel times fill_info reports (-1: not chosen yet)

    // CG state (host side)
    double eps_ = 0.0;
    double delta0_ = 0.0, delta_ = 0.0;
    double delta_before_ = 0.0;  // the residuum one iteration before delta_ (0 = none yet): the stop test's forecast, cg_step
    int held_back_ = 0;          // iterations in a row whose next matvec was NOT enqueued ahead of the stop test because of that forecast
    double y_last_ = 0.0;
    uint64_t iter_ = 0;  // iterations done
    bool converged_ = false;
    bool begun_ = false;
    double setup_ms_ = 0.0, cg_wall_ms_ = 0.0;
};

/* measurement utility (mfma_ceiling.hip): TFLOP/s and held clock of a bare v_mfma_f32_16x16x32_bf16 loop on `device` */
void measure_bf16_mfma_ceiling(int device, int b_from_lds, double settle_ms, double *tflops_out, double *clock_ghz_out, double *nominal_tflops_out);

/* one-shot helpers used by the C ABI */
template <typename T>
void predict_values(const Options &opt, const lssvm_params &params, const T *sv, size_t nsv, size_t nfeat, const T *alpha, T rho, T *w_inout, int *w_valid, const T *points,
                    size_t npoints, T *out, lssvm_predict_info *info);
template <typename T>
void calculate_w(const T *sv, size_t nsv, size_t nfeat, const T *alpha, T *w_out);

/* The resident predictor (round 6; lssvm_mi355_predictor_*): a model kept in HBM across predict calls -- csvm::predict_values (csvm.hpp:204-208) uploads and prepares the
 * support vectors on every call, which is the whole cost of a small batch.  create() prepares them once (centring, norms, operand planes, packed records; the linear
 * kernel: w); predict() uploads a batch of points, prepares it alike and runs the product.  What a batch cannot do on the resident form (see Predictor<T>::predict) goes
 * through the one-shot path with the host copies kept here: the result is the same either way. */
struct PredictorBase {
    int dtype = 0;
    virtual ~PredictorBase() = default;
    virtual void predict(const void *points, int mem_kind, size_t npoints, void *out, lssvm_predict_info *info) = 0;  // points AND out of mem_kind
};
std::unique_ptr<PredictorBase> make_predictor(const Options &opt, const lssvm_params &params, int dtype, const void *sv, size_t nsv, size_t nfeat, const void *alpha, double rho);

void check_params(const lssvm_params *params);
int select_device_checked(int device);
/* `num_devices` == 0: automatic (all visible devices, but at least 32 row blocks per device); `devices` may be NULL (0 .. num_devices-1) */
std::vector<int> resolve_devices(const int *devices, int num_devices, size_t num_points);

/* ------------------------------------------------------------------ shared by lssvm_problem.hip and lssvm_predict.hip ------------------------------------------------------------------ */
constexpr double FOLD_MAX_R2 = 200.0;    // folded rbf records (KT_RBFF) only while |c| = R2 / 2 <= 100: 2^c and 2^acc stay far inside the fp32 range
constexpr double PAIR_FOLD_MAX_C = 32.0; // 256-row kernels, rbf: row AND column term folded (K = e_i 2^(x_i.x_j) e_j) only while |c| <= 32: the partial sums then carry at most 2^32 of
                                         // extra scale (RBF_DIRECT_ABOVE = 32 keeps the automatic choice at |c| <= 16)
constexpr int PAIR_MIN_TILES = 64;       // 256-row workgroups from this many row blocks on (a rule on the global shape: every rank decides alike)
constexpr int SPLIT_MAX_FEATURES = 384;  // the bf16x6 kernels exist for 1 ... 6 chunks of 64 features (row panel = 3 planes in registers)
constexpr int F16_MAX_FEATURES = 512;    // the f16x3 kernels exist for 1 ... 8 chunks of 64 features (row panel = 2 planes in registers)
constexpr int F16_LINEAR_MAX_FEATURES = 1 << 20;
constexpr int F16_RBF_MAX_FEATURES = 384;  // ... rbf: 1 ... 6 chunks (three row planes in registers: the shifted planes, see make_planes)
constexpr int F16_RBF_SHIFT = 6;         // rbf: the planes are (2^-6 hi, 2^6 mid, 2^6 hi)
constexpr int F16_TARGET_EXP = 14;       // f16x3, linear / polynomial: the planes carry 2^k x with max |2^k x| in [2^14, 2^15) (f16 overflows at 65504)
constexpr int F16_MAX_SHIFT = 40;        // |k| is clamped here (2^(-2k) must stay a normal float beside gamma)
constexpr float F16_REL2_MAX = 0x1p-44f; // accepted relative representation error of a row, squared: |x - (hi + mid)| <= 2^-22 |x| in the 2-norm
constexpr float F16_ABS_MAX = 0x1p-22f;  // rbf: accepted bound on the ABSOLUTE error of the exponent from the representation, 2 max|rest| max|x|

template <typename T>
T rbf_prescale(const lssvm_params &p, bool fp64_v2);
bool wide_nonlinear_f64(const Options &o, const lssvm_params &p, size_t num_features);
bool wide_nonlinear(const Options &o, const lssvm_params &p, bool rbf_direct, size_t num_features);
template <typename T>
void set_kernel_scalars(TileArgs<T> &a, const lssvm_params &p, bool rbf_direct);
template <typename T>
void set_launch_options(TileArgs<T> &a, const Options &o);
template <typename T>
void column_means(const DeviceMatrix<T> &M, DevBuf<T> &mean, hipStream_t s);
template <typename T>
double max_centred_sqnorm(const DeviceMatrix<T> &M, const DevBuf<T> &mean, hipStream_t s);
template <typename T>
bool rbf_wants_direct_form(const Options &o, const lssvm_params &p, const DeviceMatrix<T> &M, const DeviceMatrix<T> *M2, hipStream_t s, double *r2_out);
void make_planes(const Options &o, const lssvm_params &p, bool rbf_direct, const DeviceMatrix<float> &M, const DeviceMatrix<float> *M2, PlaneSet &out, PlaneSet *out2, hipStream_t s,
                 bool wide_nl = false, bool linear_panels = false, bool f16_known_bad = false);
bool rbf_wants_grid_planes(const Options &o, const lssvm_params &p, size_t num_features, double r2);
float make_grid_planes(const DeviceMatrix<float> &M, double r2_in, PlaneSet &out, float *chg, DevBuf<float> &efac, hipStream_t s, bool wide_nl);
void set_plane_args(TileArgs<float> &a, const lssvm_params &p, const PlaneSet &cols, const PlaneSet &rows, size_t col_rows_alloc, size_t row_rows_alloc);
std::vector<int2> xcd_lane_order(const std::vector<std::vector<int2>> &by_chunk);
void enqueue_pack_records(const float *dvec, const float *cc, int ncols_padded, float *dc, int folded, const float *efac, hipStream_t s);
void enqueue_pack_records(const double *dvec, const double *cc, int ncols_padded, double *dc, int folded, const double *efac, hipStream_t s);
void enqueue_planes_fragment_major(const uint16_t *planes, size_t plane_elems, int rows_alloc, int ldx16, int nplanes, uint16_t *frag, hipStream_t s);

}  // namespace lssvm
