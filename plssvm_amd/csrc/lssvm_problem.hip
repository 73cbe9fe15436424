/*
 * lssvm_problem.hip -- Problem<T>: one device's shard of the device-resident LS-SVM problem (see lssvm_problem.hip.hpp) -- the data in its HBM layouts, the
 * operand planes, the work-item geometry, the tile-kernel launches of one implicit matvec -- and the one-shot predict path, a rectangular instance of the same
 * tile kernels.  The CG driver over the shards is lssvm_solver.hip, the exchange between them lssvm_exchange.hip.  Compiled for gfx950 only.
 */
#include "lssvm_problem.hip.hpp"

#define LSSVM_KERNELS_SETUP
#include "lssvm_kernels.hip.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>

namespace lssvm {

Options &options() {
    static Options o;
    return o;
}
std::mutex &options_mutex() {
    static std::mutex m;
    return m;
}

void check_params(const lssvm_params *params) {
    LSSVM_REQUIRE(params != nullptr, "params must not be NULL!");
    LSSVM_REQUIRE(params->kernel_type == LSSVM_KERNEL_LINEAR || params->kernel_type == LSSVM_KERNEL_POLYNOMIAL || params->kernel_type == LSSVM_KERNEL_RBF,
                  "Invalid kernel function " + std::to_string(params->kernel_type) + " given!");  // csvm.hpp:379
    if (params->kernel_type != LSSVM_KERNEL_LINEAR) {
        LSSVM_REQUIRE(params->gamma > 0.0, "gamma must be greater than 0, but is " + std::to_string(params->gamma) + "!");  // svm_kernel.cpp:68, :77
    }
    LSSVM_REQUIRE(params->cost != 0.0 && std::isfinite(1.0 / params->cost), "cost must not be 0.0 since it is 1 / plssvm::cost!");  // svm_kernel.cpp:27
}

static int device_count_checked() {
    int count = 0;
    const hipError_t err = hipGetDeviceCount(&count);
    if (err != hipSuccess || count <= 0) {
        (void) hipGetLastError();
        throw Error(LSSVM_ERR_NO_DEVICE, "HIP backend selected but no HIP capable devices were found!");  // csvm.hip.cpp:70-72
    }
    return count;
}

int select_device_checked(int device) {
    const int count = device_count_checked();
    LSSVM_REQUIRE(device >= 0 && device < count, "Invalid device " + std::to_string(device) + " (" + std::to_string(count) + " available)!");
    LSSVM_HIP_CHECK(hipSetDevice(device));
    return count;
}

std::vector<int> resolve_devices(const int *devices, int num_devices, size_t num_points) {
    const int count = device_count_checked();
    LSSVM_REQUIRE(num_devices >= 0 && num_devices <= MAX_LOCAL_DEVICES, "num_devices must be between 0 (automatic) and " + std::to_string(MAX_LOCAL_DEVICES) + "!");
    std::vector<int> out;
    if (num_devices == 0) {
        // automatic: every visible device, as the reference's backends do (csvm.hip.cpp:66-75), but never fewer than 32 row blocks
        // (4096 points) per device -- below that the exchange and the launch latencies outweigh the split
        LSSVM_REQUIRE(devices == nullptr, "a device list needs its length (num_devices > 0)");
        const long tiles = (static_cast<long>(num_points) - 1 + TILE - 1) / TILE;
        const int use = static_cast<int>(std::max<long>(1, std::min<long>(std::min(count, MAX_LOCAL_DEVICES), tiles / 32)));
        for (int r = 0; r < use; ++r) out.push_back(r);
        return out;
    }
    for (int r = 0; r < num_devices; ++r) {
        const int dev = devices != nullptr ? devices[r] : r;
        LSSVM_REQUIRE(dev >= 0 && dev < count, "Invalid device " + std::to_string(dev) + " (" + std::to_string(count) + " available)!");
        out.push_back(dev);
    }
    return out;
}

/* ------------------------------------------------------------------ tile kernel selection ------------------------------------------------------------------ */
/* the v2 kernels exist for 1..8 and 10, 12, 14, 16 k-chunks (padded_features() never produces an odd count above 8) */
static bool v2_chunk_count_ok(int kchunks, int single_up_to = 8) {
    return kchunks <= single_up_to || (kchunks <= 16 && kchunks % 2 == 0);
}
bool v2_eligible(const Options &o, int ldx, bool rbf_direct) {
    return !rbf_direct && v2_chunk_count_ok(ldx / F32_KC, 4) && o.tile_kernel != 1;
}
bool v2_eligible_f64(const Options &o, int ldx) {
    return v2_chunk_count_ok(ldx / F64_KC) && o.tile_kernel != 1;
}

constexpr int F64_ONE_PASS_FEATURES = 256;

/* fp64 rbf / polynomial on more than 256 features: feature panels of 64 inside a sub-tile (lssvm_tile_f64_wide.hip.hpp; the data carries the
   kernel's scale as on the one-pass v2 kernel, so gamma > 0 -- a precondition of the kernels anyway -- and a non-negative degree are required) */
bool wide_nonlinear_f64(const Options &o, const lssvm_params &p, size_t num_features) {
    const bool nonlinear = p.kernel_type == LSSVM_KERNEL_RBF || (p.kernel_type == LSSVM_KERNEL_POLYNOMIAL && p.degree >= 0);
    return nonlinear && p.gamma > 0.0 && o.tile_kernel != 1 && padded_features<double>(num_features) > F64_ONE_PASS_FEATURES;
}

/* fp32 rbf / polynomial beyond the feature count the one-pass split kernels take (row panel in registers): the panel kernel applies */
bool wide_nonlinear(const Options &o, const lssvm_params &p, bool rbf_direct, size_t num_features) {
    const bool nonlinear = (p.kernel_type == LSSVM_KERNEL_RBF && !rbf_direct) || (p.kernel_type == LSSVM_KERNEL_POLYNOMIAL && p.degree >= 0);
    const long one_pass_limit = (p.kernel_type == LSSVM_KERNEL_RBF || o.gram_mode == 1) ? 384 : 512;  // SPLIT_MAX_FEATURES / F16_MAX_FEATURES (rbf: F16_RBF_MAX_FEATURES)
    return nonlinear && o.gram_mode != 0 && o.tile_kernel != 1 && round_up(static_cast<long>(num_features), 64) > one_pass_limit;
}

/* first row block of rank r when the lower triangle is dealt by equal area: tiles * sqrt(r / world), rounded to an EVEN block index -- the
 * 256-row workgroups of lssvm_tile_f32_pair.hip.hpp work on the block pairs (2p, 2p + 1), which must not straddle two devices (every symmetric
 * partition follows the rule, whichever kernel runs: one partition per problem shape) */
int sym_block_boundary(int num_tiles, int r, int world, const std::vector<double> *weights) {
    if (r <= 0) return 0;
    if (r >= world) return num_tiles;
    // the share of the triangle's AREA in front of rank r: r / world, or -- shard weights (lssvm_mi355_set_shard_weights: devices of unequal pace) -- the ranks' weights in front of it
    double share = static_cast<double>(r) / static_cast<double>(world);
    if (weights != nullptr && static_cast<int>(weights->size()) == world) {
        double before = 0.0, all = 0.0;
        for (int k = 0; k < world; ++k) {
            all += (*weights)[static_cast<size_t>(k)];
            if (k < r) before += (*weights)[static_cast<size_t>(k)];
        }
        share = before / all;
    }
    const int b = 2 * static_cast<int>(std::llround(0.5 * static_cast<double>(num_tiles) * std::sqrt(share)));
    return std::min(std::max(b, 0), num_tiles);
}

/* the row blocks [begin, end) of rank `rank`: equal (or weighted) AREAS of the lower triangle (symmetric variant: block ib costs ib + 1 tiles),
 * equal contiguous runs otherwise (every row costs the same, and the all-gather of the full-square variant needs equal slices: no weights there) */
void shard_blocks(int num_tiles, int world, int rank, bool symmetric, int &begin, int &end, const std::vector<double> *weights) {
    if (symmetric) {
        begin = sym_block_boundary(num_tiles, rank, world, weights);
        end = std::max(begin, sym_block_boundary(num_tiles, rank + 1, world, weights));
    } else {
        const int per_rank = (num_tiles + world - 1) / world;
        begin = std::min(rank * per_rank, num_tiles);
        end = std::min(begin + per_rank, num_tiles);
    }
}

/* kernel-function specific scalars of TileArgs */
template <typename T>
void set_kernel_scalars(TileArgs<T> &a, const lssvm_params &p, bool rbf_direct) {
    a.degree = p.degree;
    a.coef0 = static_cast<T>(p.coef0);
    constexpr double log2e = 1.4426950408889634073599246810019;
    if (p.kernel_type == LSSVM_KERNEL_RBF) {
        const T g = static_cast<T>(p.gamma);  // gamma is rounded to the real type first, as in parameter<T>
        if (rbf_direct) {
            a.gamma = static_cast<T>(-static_cast<double>(g) * log2e);
        } else if (std::is_same_v<T, float>) {
            a.gamma = T(1);  // folded into the data (rbf_prescale)
        } else {
            a.gamma = static_cast<T>(2.0 * static_cast<double>(g));
        }
    } else {
        a.gamma = static_cast<T>(p.gamma);
    }
}

/* the launch-related knobs travel inside TileArgs: the launchers never read the process-wide defaults */
template <typename T>
void set_launch_options(TileArgs<T> &a, const Options &o) {
    a.dbg = static_cast<int>(o.debug_ablate);
    a.mfma_shape = static_cast<int>(o.mfma_shape);
    a.pair_lag = static_cast<int>(o.pair_lag);
}

/* rbf on the matrix cores: the data is scaled so that the MFMA chain leaves the exponent in the unit the epilogue wants:
 * fp32: x' = sqrt(2 gamma log2 e) (x - mean)  =>  x_i'.x_j' - (|x_i'|^2 + |x_j'|^2)/2 = -gamma log2(e) |x_i - x_j|^2 (epilogue v_exp_f32);
 * fp64 on the v2 kernel (`fp64_v2`): the same (epilogue exp2_f64); fp64 on the generic kernel: unscaled (epilogue fast_exp_f64(acc * 2 gamma)). */
template <typename T>
T rbf_prescale(const lssvm_params &p, bool fp64_v2) {
    constexpr double log2e = 1.4426950408889634073599246810019;
    if constexpr (std::is_same_v<T, float>) {
        return static_cast<T>(std::sqrt(2.0 * static_cast<double>(static_cast<T>(p.gamma)) * log2e));
    } else {
        return fp64_v2 ? static_cast<T>(std::sqrt(2.0 * p.gamma * log2e)) : T(1);
    }
}

/* column means of M's valid rows, in double, deterministic two-stage sum */
template <typename T>
void column_means(const DeviceMatrix<T> &M, DevBuf<T> &mean, hipStream_t s) {
    const int rows_per_block = 256;
    const int nblocks = (M.rows + rows_per_block - 1) / rows_per_block;
    DevBuf<double> part;
    part.alloc_zero(static_cast<size_t>(nblocks) * M.ldx, s);
    mean.alloc_zero(M.ldx, s);
    const dim3 g1(nblocks, (M.ldx + 255) / 256);
    hipLaunchKernelGGL(k_colsum_stage1<T>, g1, dim3(256), 0, s, M.data.p, M.ldx, M.rows, rows_per_block, part.p);
    hipLaunchKernelGGL(k_colsum_stage2<T>, dim3((M.ldx + 255) / 256), dim3(256), 0, s, part.p, nblocks, M.ldx, M.rows, mean.p);
    LSSVM_HIP_CHECK(hipGetLastError());
    LSSVM_HIP_CHECK(hipStreamSynchronize(s));  // part is released on return
}

/* max_i |x_i - mean|^2 over the valid rows of M (the data is NOT modified) */
template <typename T>
double max_centred_sqnorm(const DeviceMatrix<T> &M, const DevBuf<T> &mean, hipStream_t s) {
    DevBuf<double> sq;
    sq.alloc_zero(static_cast<size_t>(M.rows), s);
    hipLaunchKernelGGL(k_centred_sqnorm<T>, dim3((M.rows + 3) / 4), dim3(256), 0, s, M.data.p, M.ldx, M.dfeat, M.rows, mean.p, sq.p);
    LSSVM_HIP_CHECK(hipGetLastError());
    std::vector<double> host(static_cast<size_t>(M.rows));
    LSSVM_HIP_CHECK(hipMemcpyAsync(host.data(), sq.p, host.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    LSSVM_HIP_CHECK(hipStreamSynchronize(s));
    return host.empty() ? 0.0 : *std::max_element(host.begin(), host.end());
}

template <typename T>
void center_columns(DeviceMatrix<T> &M, DeviceMatrix<T> *M2, T scale, hipStream_t s) {
    DevBuf<T> mean;
    column_means<T>(M, mean, s);
    hipLaunchKernelGGL(k_center<T>, dim3((M.dfeat + 255) / 256, M.rows), dim3(256), 0, s, M.data.p, M.ldx, M.dfeat, M.rows, mean.p, scale);
    if (M2 != nullptr) {
        hipLaunchKernelGGL(k_center<T>, dim3((M2->dfeat + 255) / 256, M2->rows), dim3(256), 0, s, M2->data.p, M2->ldx, M2->dfeat, M2->rows, mean.p, scale);
    }
    LSSVM_HIP_CHECK(hipGetLastError());
    LSSVM_HIP_CHECK(hipStreamSynchronize(s));  // mean is released on return
}

/* fp32 rbf, option rbf_form = 0 (automatic): the matrix-core form evaluates the exponent as c_i + c_j + x_i'.x_j' on data scaled by
 * sqrt(2 gamma log2 e), so its ABSOLUTE error is about 2^-24 times the size of the terms that cancel, R2 = 2 gamma log2(e) max |x - mean|^2,
 * whatever the distance of the pair -- for nearby points (K close to 1) that is a relative error of K of ~R2 * 2^-24, where the reference's
 * direct (x_i - x_j)^2 chain keeps all its digits.  [-1, 1]-scaled data with gamma = 1 / num_features has R2 <= 3; gamma = 1 at 128
 * features has R2 ~ 100.  Above RBF_DIRECT_ABOVE the formula-exact vector-ALU kernel is used instead (5x slower, same accuracy class as
 * the reference).  Returns true for the direct form.  `M2` (predict: the points beside the support vectors) may be NULL. */
template <typename T>
bool rbf_wants_direct_form(const Options &o, const lssvm_params &p, const DeviceMatrix<T> &M, const DeviceMatrix<T> *M2, hipStream_t s, double *r2_out) {
    if (r2_out != nullptr) *r2_out = 0.0;
    if (!std::is_same_v<T, float> || p.kernel_type != LSSVM_KERNEL_RBF) return false;
    if (o.rbf_form == 1) return true;
    DevBuf<T> mean;
    column_means<T>(M, mean, s);
    double sq = max_centred_sqnorm<T>(M, mean, s);
    if (M2 != nullptr) sq = std::max(sq, max_centred_sqnorm<T>(*M2, mean, s));
    const double r2 = 2.0 * static_cast<double>(static_cast<T>(p.gamma)) * 1.4426950408889634 * sq;
    if (r2_out != nullptr) *r2_out = r2;
    if (o.rbf_form == 2) return false;  // the norm expansion whatever the scale (r2 is still reported: it decides the record form)
    return r2 > RBF_DIRECT_ABOVE;       // (0 and 3: the caller then moves to the grid planes where they exist -- rbf_wants_grid_planes -- and stays here where they do not)
}

/* fp32: reorder the features of every group of 8 to 0,2,4,6,1,3,5,7 (the operand order of the MFMA kernels); fp64: nothing */
template <typename T>
void interleave_features(DeviceMatrix<T> &M, hipStream_t s) {
    if constexpr (std::is_same_v<T, float>) {
        const size_t ngroups = static_cast<size_t>(M.rows_alloc) * M.ldx / 8;
        hipLaunchKernelGGL(k_interleave_features, dim3(static_cast<unsigned>((ngroups + 255) / 256)), dim3(256), 0, s, M.data.p, ngroups);
        LSSVM_HIP_CHECK(hipGetLastError());
    }
}

template <typename T>
void half_neg_norms(const DeviceMatrix<T> &M, DevBuf<T> &c, hipStream_t s) {
    c.alloc_zero(M.rows_alloc, s);
    hipLaunchKernelGGL(k_half_neg_norms<T>, dim3((M.rows_alloc + 3) / 4), dim3(256), 0, s, M.data.p, M.ldx, M.rows_alloc, c.p);
    LSSVM_HIP_CHECK(hipGetLastError());
}

/* fp32: the (centred, scaled) data once more as operand planes of the split kernels; `M2` (predict: the points beside the support vectors)
 * may be NULL and gets planes of the same kind and scale.
 *   f16x3 (mode 2): two f16 planes, x ~ hi + mid (11 + 11 significant bits).  What limits them is f16's exponent range: mid ~ 2^-12 |x| turns
 *     subnormal (loses bits) for |x| < 2^-2.  Linear and polynomial kernels: the planes carry 2^k x, k moves the largest entry to [2^14, 2^15)
 *     so that mid stays a NORMAL f16 for entries down to 2^-17 of it -- undone exactly on the finished sums (TileArgs::out_scale) or inside
 *     gamma.  rbf cannot pre-scale (the chain must leave the exponent itself, and a multiply per element in the epilogue costs matrix-core
 *     time): it uses the SHIFTED planes (2^-6 hi, 2^6 mid, 2^6 hi) -- every product of a row plane and a column plane carries the net scale 1,
 *     mid is normal for |x| in [2^-8, 2^10), which covers everything the exponent scale allows (|x| <= sqrt(RBF_DIRECT_ABOVE)); the price is a
 *     third row plane in registers, none in the column stream.  Whether the planes carry THIS data as well as fp32 does is measured, not
 *     assumed: k_split_f16x2 returns the largest relative representation error of a row; above 2^-22 (data with a dynamic range beyond the
 *     planes', or planes that overflow) the answer is no -- for rbf an absolute bound on the exponent's error is accepted as well.
 *   bf16x6 (mode 1): three bf16 planes, exact for every fp32 input (8 exponent bits), twice the matrix-core work.
 * option gram_mode: 0 = none (native v_mfma_f32 kernels), 1 = bf16x6, 2 = f16x3 without the check (A/B, tests), 3 = f16x3 if the data passes, else bf16x6. */
void make_planes(const Options &o, const lssvm_params &p, bool rbf_direct, const DeviceMatrix<float> &M, const DeviceMatrix<float> *M2, PlaneSet &out, PlaneSet *out2, hipStream_t s,
                 bool wide_nl, bool linear_panels, bool f16_known_bad) {
    out.mode = 0;
    if (out2 != nullptr) out2->mode = 0;
    // (wide_nl: rbf / polynomial beyond the register-resident row panel -- the kernel walks feature panels of 128 inside a tile and exists for
    // both plane kinds, so neither feature limit below applies; planes padded to whole panels)
    const int ldx16 = static_cast<int>(round_up(static_cast<long>(M.dfeat), wide_nl ? 128 : 64));
    // linear_panels: the CALLER says whether the linear kernel's panel passes will run (symmetric variant, f16x3 only) -- the decision is not
    // re-derived here (a full-square problem of more than 512 features got planes that nobody read).  f16_known_bad: an earlier call
    // on the same data has already seen the representability check fail.
    const bool wide_linear = linear_panels && p.kernel_type == LSSVM_KERNEL_LINEAR && ldx16 > LINEAR_PANEL_FEATURES && o.tile_kernel != 1;
    if (o.gram_mode == 0 || rbf_direct || (!v2_eligible(o, M.ldx, false) && !wide_linear && !wide_nl)) return;
    auto alloc = [&](int nplanes) {
        out.ldx16 = ldx16;
        out.nplanes = nplanes;
        if (out2 != nullptr) out2->nplanes = nplanes;
        out.buf.alloc_zero(static_cast<size_t>(nplanes) * M.rows_alloc * ldx16, s);
        if (M2 != nullptr) {
            out2->ldx16 = ldx16;
            out2->buf.alloc_zero(static_cast<size_t>(nplanes) * M2->rows_alloc * ldx16, s);
        }
    };
    const bool rbf = p.kernel_type == LSSVM_KERNEL_RBF;
    const int f16_limit = rbf ? F16_RBF_MAX_FEATURES : (wide_linear ? F16_LINEAR_MAX_FEATURES : F16_MAX_FEATURES);
    if ((o.gram_mode == 2 || o.gram_mode == 3) && (ldx16 <= f16_limit || wide_nl) && !(f16_known_bad && o.gram_mode == 3)) {
        DevBuf<unsigned> stats;
        stats.alloc_zero(4, s);
        int shift = 0;
        unsigned host[4] = { 0, 0, 0, 0 };
        if (p.kernel_type != LSSVM_KERNEL_RBF) {
            absmax_f32(M.data.p, M.ldx, M.dfeat, static_cast<size_t>(M.rows), stats.p + 3, s);
            if (M2 != nullptr) absmax_f32(M2->data.p, M2->ldx, M2->dfeat, static_cast<size_t>(M2->rows), stats.p + 3, s);
            LSSVM_HIP_CHECK(hipMemcpyAsync(host, stats.p, sizeof(host), hipMemcpyDeviceToHost, s));
            LSSVM_HIP_CHECK(hipStreamSynchronize(s));
            float amax = 0.0f;
            std::memcpy(&amax, &host[3], sizeof(float));
            if (amax > 0.0f && std::isfinite(amax)) shift = std::min(std::max(F16_TARGET_EXP - std::ilogb(amax), -F16_MAX_SHIFT), F16_MAX_SHIFT);
        }
        alloc(rbf ? 3 : 2);
        const float scale = std::ldexp(1.0f, shift);
        const int pshift = rbf ? F16_RBF_SHIFT : 0;
        split_f16_planes(M.data.p, M.ldx, M.dfeat, static_cast<size_t>(M.rows_alloc), ldx16, scale, pshift, out.buf.p, static_cast<size_t>(M.rows_alloc) * ldx16, stats.p, s);
        if (M2 != nullptr) {
            split_f16_planes(M2->data.p, M2->ldx, M2->dfeat, static_cast<size_t>(M2->rows_alloc), ldx16, scale, pshift, out2->buf.p, static_cast<size_t>(M2->rows_alloc) * ldx16, stats.p, s);
        }
        LSSVM_HIP_CHECK(hipMemcpyAsync(host, stats.p, sizeof(host), hipMemcpyDeviceToHost, s));
        LSSVM_HIP_CHECK(hipStreamSynchronize(s));
        float rel2 = 0.0f, rest2 = 0.0f, x2 = 0.0f;
        std::memcpy(&rel2, &host[0], sizeof(float));
        std::memcpy(&rest2, &host[1], sizeof(float));
        std::memcpy(&x2, &host[2], sizeof(float));
        out.f16_row_rel_error = std::sqrt(static_cast<double>(rel2));  // (NaN: an overflowing plane)
        bool ok = rel2 <= F16_REL2_MAX;  // (false for a NaN: an overflowing plane)
        if (!ok && p.kernel_type == LSSVM_KERNEL_RBF) ok = std::isfinite(rel2) && 2.0 * std::sqrt(static_cast<double>(rest2) * static_cast<double>(x2)) <= static_cast<double>(F16_ABS_MAX);
        if (const char *dbg = std::getenv("LSSVM_MI355_DEBUG"); dbg != nullptr && dbg[0] == '1') {
            std::fprintf(stderr, "[plssvm_amd] f16 planes: shift %d, max relative representation error of a row %.3g (accepted up to %.3g), max |rest| %.3g, max |x| %.3g -> %s\n", shift,
                         std::sqrt(static_cast<double>(rel2)), std::sqrt(static_cast<double>(F16_REL2_MAX)), std::sqrt(static_cast<double>(rest2)), std::sqrt(static_cast<double>(x2)),
                         ok ? "f16x3" : (o.gram_mode == 2 ? "f16x3 (forced)" : "bf16x6"));
        }
        if (ok || o.gram_mode == 2) {
            out.mode = 2;
            out.shift = shift;
            if (out2 != nullptr) {
                out2->mode = 2;
                out2->shift = shift;
            }
            return;
        }
        // Round 6, linear kernel (training): what ONE scale for the matrix cannot hold is data whose ROWS differ by orders of magnitude (tests/tools/gram_mode_by_data.py:
        // the only shape that failed) -- the small rows' mid plane falls into f16's subnormals.  A power-of-two scale PER ROW holds them: K = D (Xs Xs^T) D with D = diag(2^-k_i),
        // so the tile kernels run unchanged on Xs and Problem<float> multiplies the vector by D in front of the product and the result by D behind it (two O(n) launches)
        // instead of running twice the matrix-core work as bf16x6.
        if (p.kernel_type == LSSVM_KERNEL_LINEAR && M2 == nullptr && o.gram_mode == 3 && std::isfinite(rel2)) {
            out.row_inv_scale.alloc_zero(static_cast<size_t>(M.rows_alloc), s);
            LSSVM_HIP_CHECK(hipMemsetAsync(stats.p, 0, 4 * sizeof(unsigned), s));
            split_f16_planes(M.data.p, M.ldx, M.dfeat, static_cast<size_t>(M.rows_alloc), ldx16, 1.0f, 0, out.buf.p, static_cast<size_t>(M.rows_alloc) * ldx16, stats.p, s, out.row_inv_scale.p);
            LSSVM_HIP_CHECK(hipMemcpyAsync(host, stats.p, sizeof(host), hipMemcpyDeviceToHost, s));
            LSSVM_HIP_CHECK(hipStreamSynchronize(s));
            float rel2_rows = 0.0f;
            std::memcpy(&rel2_rows, &host[0], sizeof(float));
            if (const char *dbg = std::getenv("LSSVM_MI355_DEBUG"); dbg != nullptr && dbg[0] == '1') {
                std::fprintf(stderr, "[plssvm_amd] f16 planes with a scale per row: max relative representation error of a row %.3g -> %s\n", std::sqrt(static_cast<double>(rel2_rows)),
                             rel2_rows <= F16_REL2_MAX ? "f16x3 (row scaled)" : "bf16x6");
            }
            if (rel2_rows <= F16_REL2_MAX) {
                out.mode = 2;
                out.shift = 0;
                out.f16_row_rel_error = std::sqrt(static_cast<double>(rel2_rows));
                return;
            }
            out.row_inv_scale.release();
        }
        out.buf.release();
        if (out2 != nullptr) out2->buf.release();
    }
    if ((o.gram_mode == 1 || o.gram_mode == 3) && (ldx16 <= SPLIT_MAX_FEATURES || wide_nl)) {
        alloc(3);
        split_bf16_planes(M.data.p, M.ldx, M.dfeat, static_cast<size_t>(M.rows_alloc), ldx16, out.buf.p, static_cast<size_t>(M.rows_alloc) * ldx16, s);
        out.mode = 1;
        if (M2 != nullptr) {
            split_bf16_planes(M2->data.p, M2->ldx, M2->dfeat, static_cast<size_t>(M2->rows_alloc), ldx16, out2->buf.p, static_cast<size_t>(M2->rows_alloc) * ldx16, s);
            out2->mode = 1;
        }
    }
}

/* rbf on GRID planes (KT_RBFG): the rule that chooses them, and the planes of one (centred, scaled) matrix */
bool rbf_wants_grid_planes(const Options &o, const lssvm_params &p, size_t num_features, double r2) {
    const bool shape = p.kernel_type == LSSVM_KERNEL_RBF && o.gram_mode != 0 && o.tile_kernel != 1;  // (<= 128 features: _g6h, <= 384: _g6w, beyond: the panels-inside-a-tile kernel)
    // (the cross terms |h||s| grow like R2 sqrt(d): the limit is RBF_GRID_MAX_R2 at 128 features and sqrt(128 / d) of it beyond -- 2 365 at 384; a 300-case random run at the
    // flat limit had its worst case, 15.7 eps, on wide data at the top of the range: profiles/r05_grid_stress_seed31.log)
    const double limit = RBF_GRID_MAX_R2 * std::sqrt(128.0 / static_cast<double>(std::max<size_t>(num_features, 128)));
    return shape && std::isfinite(r2) && r2 <= limit && (o.rbf_form == 3 || (o.rbf_form == 0 && r2 > RBF_DIRECT_ABOVE));
}
/* g from the exponent scale alone (max|x_k| <= sqrt(R2), so |h / g| <= 2048 holds with it; (R2 + 160) / (g^2 / 2) <= 2^24 keeps the h.h chain of every pair that matters -- |t| <= 150,
 * beyond that 2^t is 0 in fp32 -- exact); sigma moves the largest |h| below f16's maximum.  `chg` (rows_alloc floats, allocated) receives sigma^2 ch_i, `efac` is allocated here. */
float make_grid_planes(const DeviceMatrix<float> &M, double r2_in, PlaneSet &out, float *chg, DevBuf<float> &efac, hipStream_t s, bool wide_nl) {
    const double r2 = std::max(r2_in, 1.0);
    const double g = std::exp2(std::ceil(std::log2(std::max(std::sqrt((r2 + 160.0) * 0x1p-23), std::sqrt(r2) / 2048.0))));
    const float sigma = static_cast<float>(std::exp2(std::floor(std::log2(60000.0 / (std::sqrt(r2) + g)))));
    out.ldx16 = static_cast<int>(round_up(static_cast<long>(M.dfeat), wide_nl ? 128 : 64));  // (the panels-inside-a-tile kernel walks panels of 128 features)
    out.nplanes = 3;
    out.shift = 0;
    out.buf.alloc_zero(static_cast<size_t>(3) * M.rows_alloc * out.ldx16, s);
    efac.alloc_zero(M.rows_alloc, s);
    DevBuf<unsigned> stats;
    stats.alloc_zero(4, s);
    split_grid_planes(M.data.p, M.ldx, M.dfeat, static_cast<size_t>(M.rows_alloc), out.ldx16, static_cast<float>(g), sigma, out.buf.p, static_cast<size_t>(M.rows_alloc) * out.ldx16, chg, efac.p, stats.p, s);
    unsigned bad = 0;
    LSSVM_HIP_CHECK(hipMemcpyAsync(&bad, stats.p, sizeof(bad), hipMemcpyDeviceToHost, s));
    LSSVM_HIP_CHECK(hipStreamSynchronize(s));
    // (no valid input is known to get here -- |h / g| <= 2048 follows from the exponent scale the planes were sized with -- so the tests reach the path through this hook)
    if (const char *hook = std::getenv("LSSVM_MI355_TEST_GRID_UNFIT"); hook != nullptr && hook[0] == '1') bad = 1;
    if (bad != 0) throw GridPlanesUnfit();  // (rbf_form 0: the owner builds the problem again on the formula-exact kernel; rbf_form 3: the error is the caller's)
    out.mode = 2;  // f16 planes (the launcher picks the grid kernel from TileArgs::rbf_grid)
    if (const char *dbg = std::getenv("LSSVM_MI355_DEBUG"); dbg != nullptr && dbg[0] == '1') {
        std::fprintf(stderr, "[plssvm_amd] rbf on grid planes: exponent scale %.1f, g = 2^%d, sigma = 2^%d\n", r2_in, static_cast<int>(std::log2(g)), static_cast<int>(std::log2(sigma)));
    }
    return sigma;
}

/* the plane fields of TileArgs; `gamma` must hold the kernel's own gamma already */
void set_plane_args(TileArgs<float> &a, const lssvm_params &p, const PlaneSet &cols, const PlaneSet &rows, size_t col_rows_alloc, size_t row_rows_alloc) {
    a.Xr16 = rows.buf.p;
    a.Xc16 = cols.buf.p;
    a.plane_stride = col_rows_alloc * cols.ldx16;
    a.plane_stride_r = row_rows_alloc * rows.ldx16;
    a.ldx16 = cols.ldx16;
    a.nk64 = cols.ldx16 / 64;
    a.planes_f16 = cols.mode == 2 ? 1 : 0;
    a.out_scale = 1.0f;
    if (cols.mode == 2 && cols.shift != 0) {
        const float undo = std::ldexp(1.0f, -2 * cols.shift);
        if (p.kernel_type == LSSVM_KERNEL_LINEAR) a.out_scale = undo;
        if (p.kernel_type == LSSVM_KERNEL_POLYNOMIAL) a.gamma *= undo;
    }
}

/* host restatement of kernel_function(x, x) for QA_cost (csvm.cpp:86): the same fma chain in the same precision */
template <typename T>
static T host_self_kernel(const lssvm_params &p, const std::vector<T> &x) {
    T val = T(0);
    if (p.kernel_type == LSSVM_KERNEL_RBF) {
        for (const T v : x) {
            const T diff = v - v;
            val = std::fma(diff, diff, val);
        }
        return std::exp(-static_cast<T>(p.gamma) * val);
    }
    for (const T v : x) val = std::fma(v, v, val);
    if (p.kernel_type == LSSVM_KERNEL_LINEAR) return val;
    return std::pow(std::fma(static_cast<T>(p.gamma), val, static_cast<T>(p.coef0)), static_cast<T>(p.degree));
}

/* ------------------------------------------------------------------ work-item geometry of the symmetric variant ------------------------------------------------------------------ */
static long pairs_below(long b) { return b * (b - 1) / 2; }

/* Row-block BANDS of equal triangle AREA whose column-sum records fit the slab budget (see Problem's constructor). */
static std::vector<int> band_edges(int ib_begin, int ib_end_all, size_t real_size, const Options &o) {
    const double rec_bytes = static_cast<double>(TILE) * real_size;
    const double total_bytes = static_cast<double>(pairs_below(ib_end_all) - pairs_below(ib_begin)) * rec_bytes;
    const double band_bytes = static_cast<double>(std::min(std::max<int64_t>(o.colslab_band_mb, 1), o.colslab_limit_mb)) * 1048576.0;
    const int nbands = static_cast<int>(std::min<double>(std::max(1.0, std::ceil(total_bytes / band_bytes)), std::max(ib_end_all - ib_begin, 1)));
    // equal areas: band k of this device ends where the triangle area reaches (k + 1) / nbands of the device's share
    const double a0 = static_cast<double>(ib_begin) * ib_begin, a1 = static_cast<double>(ib_end_all) * ib_end_all;
    std::vector<int> edge(nbands + 1, ib_begin);
    for (int k = 1; k < nbands; ++k) {
        const int e = 2 * static_cast<int>(std::llround(0.5 * std::sqrt(a0 + (a1 - a0) * static_cast<double>(k) / nbands)));  // even: a block pair stays in one band
        edge[k] = std::min<int>(std::max(e, edge[k - 1]), ib_end_all);
    }
    edge[nbands] = ib_end_all;
    return edge;
}

/* The work items of a launch, given chunk by chunk, in XCD-LANE order: list position 8 k + x belongs to lane x (the persistent workgroups of XCD x draw from it first,
 * for_each_work_item; the hardware's own deal is round-robin too), and a lane works through one UNIT -- a run of items of ONE column chunk -- at a time, the units dealt
 * to the lanes as they run dry.  Round 6: a unit used to be a whole chunk.  A lane that found no chunk left gave its list positions to the others, which shifts every
 * later item to another lane: with FEWER CHUNKS THAN A FEW PER LANE the mapping was scrambled for most of the list -- predict_values at 200 000 x 50 000 has 7 chunks for
 * 8 lanes: L2 hit rate 0.19, 17 GB of fabric reads per 5 ms launch (profiles/r06_pmc_predict.txt); 50 000 x 128 training: 8 chunks of unequal length, 0.77.  Now a launch
 * of fewer than 64 chunks cuts its chunks into pieces (of at least 32 items: the CUs of one XCD) so that there are about 64 units: the lanes stay aligned until the last
 * unit of the list.  Launches of 64 chunks and more (1 000 000 x 128: 123 per band) are dealt as before. */
std::vector<int2> xcd_lane_order(const std::vector<std::vector<int2>> &by_chunk) {
    size_t total = 0;
    for (const auto &c : by_chunk) total += c.size();
    const size_t pieces = by_chunk.size() < 64 ? (64 + by_chunk.size() - 1) / std::max<size_t>(by_chunk.size(), 1) : 1;
    struct Unit {
        const int2 *begin;
        size_t count;
    };
    std::vector<Unit> units;
    for (const auto &c : by_chunk) {
        if (c.empty()) continue;
        const size_t n_pieces = std::max<size_t>(1, std::min(pieces, c.size() / 32));
        for (size_t k = 0; k < n_pieces; ++k) {
            const size_t lo = c.size() * k / n_pieces, hi = c.size() * (k + 1) / n_pieces;
            units.push_back({ c.data() + lo, hi - lo });
        }
    }
    std::vector<int2> out;
    out.reserve(total);
    size_t next_unit = 0;
    std::vector<long> lane_unit(8, -1);  // -1: none yet, -2: the list of units is exhausted
    std::vector<size_t> lane_pos(8, 0);
    while (out.size() < total) {
        for (int x = 0; x < 8 && out.size() < total; ++x) {
            while (lane_unit[x] == -1 || (lane_unit[x] >= 0 && lane_pos[x] >= units[static_cast<size_t>(lane_unit[x])].count)) {
                if (next_unit >= units.size()) {
                    lane_unit[x] = -2;
                    break;
                }
                lane_unit[x] = static_cast<long>(next_unit++);
                lane_pos[x] = 0;
            }
            if (lane_unit[x] == -2) continue;  // (this lane has run out of units: its positions go to the others -- the last round of the list only)
            out.push_back(units[static_cast<size_t>(lane_unit[x])].begin[lane_pos[x]++]);
        }
    }
    return out;
}

/* The (row block, column chunk) work items of one band in DISPATCH order: column chunk major (concurrent workgroups share the chunk; the
   hardware dispatches workgroups in item order as CU slots free up).  order >= 1 (ITEM_ORDER): the items cut short by the diagonal go last in
   their band, longest first, so that the final dispatch round is made of the shortest items.  .x = absolute row block, .y = chunk. */
static std::vector<int2> band_items(int band_begin, int band_end, int jc_tiles, int num_jc, int order, bool pairs, int head_tiles = 0, int head_count = 0) {
    std::vector<int2> full, cut;
    auto cbegin = [&](int jc) { return chunk_begin(jc, jc_tiles, head_tiles, head_count); };
    auto clen = [&](int jc) { return chunk_len(jc, jc_tiles, head_tiles, head_count); };
    const int rows = pairs ? 2 : 1;  // row blocks per work item: block pairs (2p, 2p + 1) for the 256-row workgroups (.x = the even block; the band edges are even)
    const int nrow_items = (band_end - band_begin + rows - 1) / rows;
    if (order >= 4) {
        // ROW-GROUP major, for the kernels that re-load their row panel at every tile (panels inside a tile): the 64 workgroups an XCD runs at a time
        // are the items of a GROUP of row blocks x their column chunks, so that the row panels they keep re-reading fit that XCD's L2 (with the
        // column-chunk major orders every concurrent workgroup has a row panel of its own: 60 000 x 640 rbf re-read 54 GB of row planes per matvec
        // from beyond L2) and a column tile is streamed for the whole group at once.  List position 8 k + x belongs to XCD x: runs of 64 consecutive items of the
        // sequence go to one lane.
        std::vector<int2> seq;
        const int GROUP = order == 4 ? 4 : (order == 5 ? 2 : (order == 6 ? 8 : 1));  // (4 and 5 ship; 6, 7: development builds, item_order_dev)
        for (int g0 = 0; g0 < nrow_items; g0 += GROUP) {
            for (int jc = 0; jc < num_jc; ++jc) {
                for (int k = g0; k < std::min(g0 + GROUP, nrow_items); ++k) {
                    const int ib = band_begin + rows * k;
                    if (cbegin(jc) > ib + rows - 1) continue;
                    seq.push_back(make_int2(ib, jc));
                }
            }
        }
        std::vector<std::vector<int2>> lane(8);
        for (size_t i = 0; i < seq.size(); ++i) lane[(i / 64) % 8].push_back(seq[i]);
        std::vector<int2> out;
        out.reserve(seq.size());
        std::vector<size_t> pos(8, 0);
        while (out.size() < seq.size()) {
            for (int x = 0; x < 8; ++x) {
                if (pos[x] < lane[x].size()) out.push_back(lane[x][pos[x]++]);
            }
        }
        return out;
    }
    for (int jc = 0; jc < num_jc; ++jc) {
        for (int k = 0; k < nrow_items; ++k) {
            const int ib = band_begin + rows * (order == 2 ? nrow_items - 1 - k : k);
            const int last = ib + rows - 1;  // the item's tiles end at the diagonal of its LAST block
            if (cbegin(jc) > last) continue;
            const bool is_cut = cbegin(jc) + clen(jc) > last + 1 || clen(jc) < jc_tiles;  // fewer than jc_tiles tiles: cut by the diagonal, or a chunk of the short head
            (order >= 1 && is_cut ? cut : full).push_back(make_int2(ib, jc));
        }
    }
    auto item_tiles = [&](const int2 &it) { return std::min(cbegin(it.y) + clen(it.y), it.x + rows) - cbegin(it.y); };
    std::stable_sort(cut.begin(), cut.end(), [&](const int2 &x, const int2 &y) { return item_tiles(x) > item_tiles(y); });
    if (order == 3) {
        // XCD-aware: the hardware deals consecutive workgroups round-robin over the 8 XCDs (observed, a speed matter only), each with an L2 of its own.
        // In column-chunk major order the workgroups that stream one chunk are spread over all eight L2s, which each fetch it from the fabric; here list
        // position 8 k + x belongs to "lane" x, and a lane works through ONE column chunk at a time, so that an XCD's workgroups share their column
        // stream in ITS L2 (xcd_lane_order below).
        std::vector<std::vector<int2>> by_chunk(static_cast<size_t>(num_jc));
        for (const int2 &it : full) by_chunk[static_cast<size_t>(it.y)].push_back(it);
        full = xcd_lane_order(by_chunk);
    }
    full.insert(full.end(), cut.begin(), cut.end());
    return full;
}

/* Chunk length of the 256-row workgroups (block pairs) when the launch is SMALL: one workgroup per CU means 256 slots, and a problem of 20 000
 * points has only ~6 000 pair-tiles -- whether the items come out as 0.9 or 1.6 or 8 rounds over the slots decides the launch time (measured at
 * 20 000 x 128: 0.186 ms with 3-tile items, 0.144 with 16, 0.136 with 32; 10 000 points: 0.070 / 0.049 with 2 / 8 tiles;
 * profiles/r04_chunk_sweep_pair.log).  The host therefore replays the hardware's dispatch (items in list order onto the slot that frees up
 * first) for a handful of candidate lengths with the cost model  item = tiles + 2.6  (the work-item prologue and tail in tile units, fitted to
 * that sweep) and takes the shortest makespan.  Large launches (more than 16 items per slot at the longest chunk) keep the longest chunk. */
struct PairChunks {
    int tiles = 64, head_tiles = 0, head_count = 0;
};
static int num_chunks(int num_tiles, int tiles, int head_tiles, int head_count) {
    const int head = head_count * head_tiles;
    return head_count + (std::max(num_tiles - head, 0) + tiles - 1) / tiles;
}
/* Round 5: the replay also tries a HEAD of short chunks -- the first `head_count` column chunks of every row pair have `head_tiles` tiles -- whose items are
 * dispatched last, longest first, together with the items cut short by the diagonal: long items for the bulk of a launch (fewer row-panel loads), short ones
 * to fill its end.  Searched where a launch is a few rounds long (at most 8 items per slot at the longest chunk: that is where the last round
 * decides; ADVICE r04: no long replays of large launches); a fixed head of two 16-tile chunks up to 32 items per slot, none beyond.
 * MEASURED, same box, against the replay's best uniform length (option j_chunk_head = 0):
 *   - with one workgroup per item, dealt by the hardware (profiles/r05_ab_chunk_head.log): -2.5 % at 40 000 points, -1 % at 70 000, but +2.5 % at 30 000 and +1 % at
 *     50 000 -- a wash: that deal is in order and static per XCD (tests/tools/item_trace.py: CUs wait 8 ... 40 us in front of the short items, and the launch ends
 *     with its slowest XCD whatever the items);
 *   - with the PERSISTENT launches that draw their items from per-XCD counters (pair_queue_fetch; profiles/r05_queue_chunk_sweep.log, r05_queue_head_large.log):
 *     -5.4 % at 20 000 points, -4.5 % at 30 000, -3.5 % at 40 000, -2.6 % at 50 000, -0.9 % at 70 000, -1.6 % at 100 000, -1.0 % at 150 000, nothing at 250 000.
 * So heads are the default since the launches are persistent. */
static PairChunks choose_pair_chunk_by_replay(int ib_begin, int ib_end, int num_tiles, size_t real_size, const Options &o, int slots);
/* The replay is a pure function of the shard's shape and three options, and it costs 1.6 of the 4 ms a 50 000-point problem takes to set up: a process that creates
 * problems of one shape again and again (a parameter search over C and gamma, predict after train) replays once. */
static PairChunks choose_pair_chunk(int ib_begin, int ib_end, int num_tiles, size_t real_size, const Options &o, int slots) {
    using Key = std::tuple<int, int, int, size_t, int, int64_t, int64_t, int64_t>;
    static std::mutex cache_mutex;
    static std::map<Key, PairChunks> cache;
    const Key key{ ib_begin, ib_end, num_tiles, real_size, slots, o.j_chunk_head, o.colslab_band_mb, o.colslab_limit_mb };
    {
        const std::lock_guard<std::mutex> lock(cache_mutex);
        if (const auto it = cache.find(key); it != cache.end()) return it->second;
    }
    const PairChunks chosen = choose_pair_chunk_by_replay(ib_begin, ib_end, num_tiles, real_size, o, slots);
    const std::lock_guard<std::mutex> lock(cache_mutex);
    if (cache.size() >= 4096) cache.clear();  // (bounded: shapes, not problems)
    cache.emplace(key, chosen);
    return chosen;
}
static PairChunks choose_pair_chunk_by_replay(int ib_begin, int ib_end, int num_tiles, size_t real_size, const Options &o, int slots) {
    const int cap = 64;
    PairChunks best_c;
    const long area = (static_cast<long>(ib_end) * (ib_end + 1) - static_cast<long>(ib_begin) * (ib_begin + 1)) / 4;  // pair-tiles, about
    if (area / cap > 16L * slots) {  // (beyond ~16 dispatch rounds the last round no longer decides, and the replay itself costs: 16 of 27 ms of set-up at 250 000 points)
        if (o.j_chunk_head == 1 && area / cap <= 32L * slots && num_tiles > 2 * 16 + cap) {
            best_c.head_tiles = 16;
            best_c.head_count = 2;
        }
        return best_c;
    }
    const std::vector<int> edge = band_edges(ib_begin, ib_end, real_size, o);
    const bool few_rounds = area / cap <= 8L * slots && edge.size() == 2;
    const double item_cost = 2.6;  // (with the persistent launches a traced item costs 1.2 tiles beside its tiles -- tests/tools/item_trace.py --; replays with 1.2 and 0.6 choose within
                                   // 1 % of this one, 2.6 a little better from 70 000 points on: profiles/r05_queue_chunk_sweep.log)
    struct Candidate {
        int jc, head_tiles, head_count;
        double makespan;
    };
    auto replay = [&](Candidate &c) {  // (a pure function of the candidate: the replays run side by side below)
        const int num_jc = num_chunks(num_tiles, c.jc, c.head_tiles, c.head_count);
        double total = 0.0;
        for (size_t k = 0; k + 1 < edge.size(); ++k) {
            if (edge[k + 1] <= edge[k]) continue;
            std::vector<double> slot(static_cast<size_t>(slots), 0.0);  // a min-heap of the slots' finish times
            auto later = [](double x, double y) { return x > y; };
            for (const int2 &it : band_items(edge[k], edge[k + 1], c.jc, num_jc, ITEM_ORDER, true, c.head_tiles, c.head_count)) {
                const int b = chunk_begin(it.y, c.jc, c.head_tiles, c.head_count);
                const int tiles = std::min(std::min(b + chunk_len(it.y, c.jc, c.head_tiles, c.head_count), it.x + 2), num_tiles) - b;
                std::pop_heap(slot.begin(), slot.end(), later);
                slot.back() += static_cast<double>(std::max(tiles, 0)) + item_cost;
                std::push_heap(slot.begin(), slot.end(), later);
            }
            total += *std::max_element(slot.begin(), slot.end());
        }
        c.makespan = total;
    };
    std::vector<Candidate> cands;
    for (const int jc : { 2, 3, 4, 6, 8, 10, 12, 16, 20, 24, 32, 40, 48, 64 }) cands.push_back({ jc, 0, 0, 0.0 });
    if (!few_rounds && o.j_chunk_head == 1 && edge.size() == 2) {
        for (const int jc : { 48, 64 }) {
            if (2 * 16 + jc <= num_tiles) cands.push_back({ jc, 16, 2, 0.0 });
        }
    }
    if (few_rounds && o.j_chunk_head == 1) {
        for (const int jc : { 24, 32, 40, 48, 56, 64 }) {
            for (const int hc : { 1, 2, 3, 4, 6, 8 }) {
                for (const int ht : { 4, 8, 12, 16, 20, 24 }) {
                    if (ht >= jc || hc * ht + jc > num_tiles) continue;
                    cands.push_back({ jc, ht, hc, 0.0 });
                }
            }
        }
    }
    // some 230 replays of a thousand items each: 8 ms of the set-up of a 50 000-point problem on one thread (a problem that iterates in 0.7 ms), 1 ms on eight; the
    // choice -- the first candidate of the list with the shortest makespan -- does not depend on how the replays are spread
    const unsigned nthreads = cands.size() >= 32 ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    std::atomic<size_t> next{ 0 };
    auto worker = [&] {
        for (size_t k = next.fetch_add(1); k < cands.size(); k = next.fetch_add(1)) replay(cands[k]);
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(worker);
    worker();
    for (std::thread &t : pool) t.join();
    const Candidate *best = &cands[0];
    for (const Candidate &c : cands) {
        if (c.makespan < best->makespan) best = &c;
    }
    best_c.tiles = best->jc;
    best_c.head_tiles = best->head_count > 0 ? best->head_tiles : 0;
    best_c.head_count = best->head_count;
    return best_c;
}

/* ------------------------------------------------------------------ Problem: one device's shard ------------------------------------------------------------------ */
/* The part of a problem that depends on WHICH row blocks this shard evaluates (constructor; reshard): the shard's blocks, the column chunks of its work items. */
template <typename T>
void Problem<T>::choose_shard_geometry() {
    const size_t num_features = num_features_;
    const int ldx_probe = ldx_probe_;
    {
        int ib_end = 0;
        shard_blocks(num_tiles_, world_, rank_, sym_, ib_begin_, ib_end, &opt_.shard_weights);
        num_ib_ = ib_end - ib_begin_;
    }
    if (opt_.j_chunk_tiles > 0) {
        jc_tiles_ = static_cast<int>(opt_.j_chunk_tiles);
    } else {
        // automatic: long chunks amortise a work item's prologue (row panel load) and keep its row sums in registers, but the grid
        // must fill 256 CUs x 2 workgroups several times over.  Measured optimum (tests/tools/gpu_probe.py --small, 3 000 ... 50 000
        // points): about 4096 work items, between 2 and 16 tiles each.
        const long ib_end = ib_begin_ + num_ib_;
        const long area = sym_ ? (ib_end * (ib_end + 1) - static_cast<long>(ib_begin_) * (ib_begin_ + 1)) / 2 : static_cast<long>(num_ib_) * num_tiles_;
        // the bf16x6 kernel has the costlier work-item prologue (three planes of the row panel) and the faster tiles: longer chunks
        // (measured 16 -> 64 tiles: +1.5 % at 100 000 points, +2 % at 300 000; the native kernels are flat or lose beyond 16)
        const bool split = wide_linear_ || wide_nl_
                           || (std::is_same_v<T, float> && opt_.gram_mode != 0 && v2_eligible(opt_, ldx_probe, rbf_direct_)
                               && round_up(static_cast<long>(num_features), 64) <= ((opt_.gram_mode == 1 || tile_params_.kernel_type == LSSVM_KERNEL_RBF) ? SPLIT_MAX_FEATURES : F16_MAX_FEATURES));
        const long cap = split ? 64 : 16;
        jc_tiles_ = static_cast<int>(std::min<long>(cap, std::max<long>(2, (area + 2048) / 4096)));
        if (pair_) {
            int cus = 256;  // one such workgroup per CU
            LSSVM_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_));
            const PairChunks pc = choose_pair_chunk(ib_begin_, static_cast<int>(ib_end), num_tiles_, sizeof(T), opt_, std::max(cus, 1));
            jc_tiles_ = pc.tiles;
            jc_head_tiles_ = pc.head_tiles;
            jc_head_count_ = pc.head_count;
        }
        // panels inside a tile, symmetric variant: SHORT items in row-group major order (wide_order_ below) -- the shorter the better at every shape
        // (12 > 8 > 4 > 2 tiles, profiles/r04_ab_wide_item_order_groups.log, r04_ab_wide_final.log); longer only where the partial slabs would grow
        // beyond 256 per row
        if (wide_nl_ && sym_) jc_tiles_ = static_cast<int>(std::min<long>(64, std::max<long>(2, (num_tiles_ + 255) / 256)));
    }
    if (wide_nl_ && sym_) {
        // These kernels re-load a work item's row panel at EVERY column tile.  In the column-chunk major orders each of the 64 workgroups an XCD runs
        // at a time has a row panel of its own -- 64 x 0.4 ... 1.6 MB against 4 MB of L2: the re-loads came from beyond L2 at HBM rate (60 000 x 640
        // rbf: 54 GB of row planes per matvec, 6 TB/s).  Row-group major (band_items, order 4 / 5): an XCD works on FOUR (two, where a panel is
        // beyond 640 KB) row blocks x their column chunks at a time, so their panels stay in its L2 and every column tile is streamed for the
        // whole group: 60 000 x 640 rbf 8.5 -> 6.2 ms, 100 000 x 385 rbf 19.4 -> 12.4 ms.  Results do not depend on the order.
        const double panel_bytes = std::is_same_v<T, float>
                                       ? 128.0 * static_cast<double>(round_up(static_cast<long>(num_features), 128)) * ((tile_params_.kernel_type == LSSVM_KERNEL_RBF || opt_.gram_mode == 1) ? 3.0 : 2.0) * 2.0
                                       : 128.0 * static_cast<double>(ldx_probe) * 8.0;
        wide_order_ = panel_bytes <= 640e3 ? 4 : 5;
    }
    if (pair_ && opt_.j_chunk_head >= 1024) {  // an explicit head (option j_chunk_head = 1024 count + tiles): tests, A/B runs
        jc_head_count_ = static_cast<int>(opt_.j_chunk_head / 1024);
        jc_head_tiles_ = static_cast<int>(opt_.j_chunk_head % 1024);
        if (jc_head_count_ <= 0 || jc_head_tiles_ <= 0 || jc_head_count_ * jc_head_tiles_ >= num_tiles_) jc_head_count_ = jc_head_tiles_ = 0;
    }
    if (!pair_) jc_head_count_ = jc_head_tiles_ = 0;
    num_jc_ = num_chunks(num_tiles_, jc_tiles_, jc_head_tiles_, jc_head_count_);
    if (const char *dbg = std::getenv("LSSVM_MI355_DEBUG"); dbg != nullptr && dbg[0] == '1') {
        std::fprintf(stderr, "[plssvm_amd] shard %d/%d on device %d: row blocks [%d, %d) of %d, %d tiles per work item (head: %d chunks of %d), symmetric %d\n", rank_, world_, device_,
                     ib_begin_, ib_begin_ + num_ib_, num_tiles_, jc_tiles_, jc_head_count_, jc_head_tiles_, sym_ ? 1 : 0);
    }
}

/* ... and the device-side lists for it: the row slabs, the work items band by band, the column slab, the events around the band launches. */
template <typename T>
void Problem<T>::build_shard_lists(hipStream_t st) {
    partial_.alloc_zero(static_cast<size_t>(std::max(num_jc_, 1)) * part_blocks() * TILE, st);
    bands_.clear();
    if (sym_) {
        // Row-block BANDS.  Every evaluated off-diagonal tile leaves a 128-entry record of column sums (no atomics: one writer per record,
        // k_reduce_colslab adds the records of a column in a fixed order).  All records of the triangle would be n_tiles^2 / 2 * 512 bytes
        // (15.6 GB at 1M points in fp32); instead the device's row blocks are cut into bands of equal AREA whose records fit
        // colslab_band_mb, the tile kernel runs band by band into the SAME slab and the band's records are folded into K*v before the
        // next band overwrites them.  One band for up to ~360 000 points per device at the default 2 GiB.
        const int ib_end_all = ib_begin_ + num_ib_;
        const std::vector<int> edge = band_edges(ib_begin_, ib_end_all, sizeof(T), opt_);
        std::vector<int2> items;
        long max_records = 1;
        for (size_t k = 0; k + 1 < edge.size(); ++k) {
            Band band{};
            band.ib_begin = edge[k];
            band.ib_end = edge[k + 1];
            band.item_begin = static_cast<int>(items.size());
            band.pair_origin = pairs_below(band.ib_begin);
            for (const int2 &it : band_items(band.ib_begin, band.ib_end, jc_tiles_, num_jc_, opt_.item_order_dev != 0 ? static_cast<int>(opt_.item_order_dev) : (wide_order_ != 0 ? wide_order_ : ITEM_ORDER), pair_, jc_head_tiles_, jc_head_count_)) items.push_back(make_int2(it.x - ib_begin_, it.y));
            band.item_count = static_cast<int>(items.size()) - band.item_begin;
            // (block pairs: the records of the pair's SECOND block, which is padding behind an odd last block)
            max_records = std::max(max_records, pairs_below(pair_ ? round_up(band.ib_end, 2) : band.ib_end) - band.pair_origin);
            if (band.ib_end > band.ib_begin) bands_.push_back(band);
        }
        num_items_ = static_cast<int>(items.size());
        items_.alloc_zero(std::max<size_t>(items.size(), 1), st);
        if (!items.empty()) LSSVM_HIP_CHECK(hipMemcpyAsync(items_.p, items.data(), items.size() * sizeof(int2), hipMemcpyHostToDevice, st));
        colslab_.alloc_zero(static_cast<size_t>(max_records) * TILE, st);
        LSSVM_HIP_CHECK(hipStreamSynchronize(st));  // `items` goes out of scope
    }
    // (two matvecs can be in flight -- enqueue-ahead -- and each issues bands x feature-panel passes tile launches: ADVICE r03, the later panels of a wide
    // linear problem went untimed and the reported kernel time came out too low)
    // (created HERE: a reshard that needs one more band than before grows the list -- pairs without events would be handed out by free_event)
    events_.resize(4 * std::max<size_t>(bands_.size(), 1) * static_cast<size_t>(std::max(passes_per_matvec(), 1)));
    for (EvPair &e : events_) {
        if (e.a.e == nullptr) e.a.create(true);
        if (e.b.e == nullptr) e.b.create(true);
    }
}

/* New shares for the ranks of a sharded symmetric problem (lssvm_mi355_problem_rebalance): the row blocks of this shard, its work items, slabs and bands are
 * rebuilt for `weights`; the data, the operand planes, the vectors and the CG state are what they were -- the implicit matrix does not change, only who evaluates
 * which of its tiles.  The caller has drained the stream. */
template <typename T>
void Problem<T>::reshard(const std::vector<double> &weights) {
    LSSVM_REQUIRE(sym_ && world_ > 1 && static_cast<int>(weights.size()) == world_, "shares by weight exist for the symmetric variant of a sharded problem, one weight per rank");
    select_device_checked(device_);
    LSSVM_HIP_CHECK(hipStreamSynchronize(stream_.s));
    drain_events();
    opt_.shard_weights = weights;
    choose_shard_geometry();
    build_shard_lists(stream_.s);
    d_packed_ = false;  // (the records of d_ are shard independent, but K*v is cleared again by the next matvec: the plain path)
    pace_ms0_ = matvec_ms_;  // the pace of the NEW share is measured from here on; the counters lssvm_cg_info reports keep running
    pace_timed0_ = matvec_timed_;
    LSSVM_HIP_CHECK(hipStreamSynchronize(stream_.s));
}

template <typename T>
Problem<T>::Problem(const Options &opt, const lssvm_params &params, const void *X, int mem_kind, size_t num_points, size_t num_features, int device, int rank, int world) :
    opt_(opt), params_(params), device_(device), rank_(rank), world_(world) {
    check_params(&params_);
    LSSVM_REQUIRE(X != nullptr, "The data must not be empty!");                                           // csvm.cpp:73
    LSSVM_REQUIRE(num_points >= 2, "The data must contain at least two data points!");
    LSSVM_REQUIRE(num_features >= 1, "The data points must contain at least one feature!");               // csvm.cpp:74
    LSSVM_REQUIRE(num_points < (size_t(1) << 31) - 4 * TILE && num_features < (size_t(1) << 24), "problem too large for 32-bit tile indices");
    LSSVM_REQUIRE(mem_kind == LSSVM_MEM_HOST || mem_kind == LSSVM_MEM_DEVICE, "invalid mem_kind");
    LSSVM_REQUIRE(world_ >= 1 && rank_ >= 0 && rank_ < world_, "invalid shard descriptor");
    select_device_checked(device_);
    const double t0 = now_ms();
    stream_.create();
    hipStream_t st = stream_.s;
    const char *dbg_env = std::getenv("LSSVM_MI355_DEBUG");
    const bool dbg_laps = dbg_env != nullptr && dbg_env[0] == '1' && rank_ == 0;
    double t_lap = t0;
    auto lap = [&](const char *what) {  // LSSVM_MI355_DEBUG=1: where the set-up's time goes (the stream is drained at every lap, so the laps add up)
        if (!dbg_laps) return;
        (void) hipStreamSynchronize(st);
        const double t = now_ms();
        std::fprintf(stderr, "[plssvm_amd] set-up: %-44s %8.3f ms\n", what, t - t_lap);
        t_lap = t;
    };

    lap("stream");
    N_ = num_points;
    n_ = static_cast<int>(num_points - 1);
    num_tiles_ = (n_ + TILE - 1) / TILE;
    ib_per_rank_ = (num_tiles_ + world_ - 1) / world_;
    nvec_ = ib_per_rank_ * world_ * TILE;
    // data matrix: all N points (the last one is row n; it takes part in q and QA_cost only)
    // (+ TILE: an odd number of row blocks ends in a block pair whose second block is zero padding -- the 256-row workgroups read its rows, d_i and c_i)
    X_.upload(X, mem_kind, num_points, num_features, static_cast<size_t>(nvec_) + TILE, st);
    lap("data into the padded layout");
    // fp32 rbf: matrix cores (norm expansion) or the formula-exact vector-ALU kernel?  (every shard sees the same data: same decision)
    rbf_direct_ = rbf_wants_direct_form<T>(opt_, params_, X_, nullptr, st, &rbf_r2_);
    // Large exponent scales (round 5): between RBF_DIRECT_ABOVE and RBF_GRID_MAX_R2, at any width, with operand planes allowed, the matrix cores run the
    // rbf kernel on GRID planes (KT_RBFG, lssvm_tile_f32_split.hip.hpp) -- the direct form's accuracy at about twice the f16x3 time instead of five times.  A rule on
    // the data's scale, the shape and the options: every shard sees the same data and decides alike; predict_values takes the same decision.
    if constexpr (std::is_same_v<T, float>) {
        if (rbf_wants_grid_planes(opt_, params_, num_features, rbf_r2_)) {
            rbf_grid_ = true;
            rbf_direct_ = false;
        }
    }
    // symmetric variant: v2 kernels only; a negative polynomial degree can give inf on zero-padded rows -> full square
    const int ldx_probe = padded_features<T>(num_features);
    ldx_probe_ = ldx_probe;
    num_features_ = num_features;
    bool v2_ok = std::is_same_v<T, float> ? v2_eligible(opt_, ldx_probe, rbf_direct_) : v2_eligible_f64(opt_, ldx_probe);
    // FEW points, many features, linear kernel (fp32): the feature-panel passes below are launch-bound there -- 3 000 x 16 384: 128 passes 4.75 ms, the
    // polynomial kernel's ONE launch over the same panels 2.89 (profiles/r04_very_wide_probe.log).  Below LINEAR_IN_TILE_BELOW points and beyond 256
    // features the tile kernels therefore run the linear kernel AS the polynomial kernel of degree 1 with gamma = 1, coef0 = 0: K_ij = (1 * acc + 0)^1,
    // bit for bit the linear kernel's value (the planes' power-of-two pre-scale is undone inside gamma, exactly); above, the passes win (twice the
    // matrix-core rate of the panels-inside-a-tile kernels).  q, QA_cost and the data preparation keep the caller's kernel (params_); a rule on the
    // shape and the options only, so every rank of a sharded solve decides alike.
    tile_params_ = params_;
    if constexpr (std::is_same_v<T, float>) {
        if (params_.kernel_type == LSSVM_KERNEL_LINEAR && opt_.gram_mode != 0 && opt_.tile_kernel != 1 && round_up(static_cast<long>(num_features), 64) > 256
            && n_ < LINEAR_IN_TILE_BELOW) {
            tile_params_.kernel_type = LSSVM_KERNEL_POLYNOMIAL;
            tile_params_.degree = 1;
            tile_params_.gamma = 1.0;
            tile_params_.coef0 = 0.0;
        }
    }
    if constexpr (std::is_same_v<T, float>) {
        // linear kernel on more than 512 features: K*v = sum over feature panels of (X_p X_p^T) v, every panel one launch of the f16x3 kernels
        // (enqueue_apply_K_local).  Whether the data allows f16 planes decides it, so the planes are built HERE (the linear kernel needs the raw data).
        if (tile_params_.kernel_type == LSSVM_KERNEL_LINEAR && opt_.gram_mode >= 2 && opt_.tile_kernel != 1
            && round_up(static_cast<long>(num_features), 64) > LINEAR_PANEL_FEATURES) {
            if (opt_.symmetric != 0 && opt_.colslab_limit_mb != 0) {  // (the panel passes exist for the symmetric variant: no probe, no planes otherwise)
                make_planes(opt_, tile_params_, false, X_, nullptr, planes_, nullptr, st, false, true);
                if (planes_.mode == 2) {
                    v2_ok = wide_linear_ = true;
                } else {
                    f16_probe_failed_ = true;  // (the planes of the one-pass kernels below do not repeat the split and the check)
                    planes_.buf.release();
                    planes_.mode = 0;
                }
            }
        }
    }
    if constexpr (std::is_same_v<T, double>) {
        // fp64 linear kernel on more than 256 features (the widest row panel the resident-row-panel kernel holds): the same sum over feature
        // panels, one pass of that kernel per panel of 128 features (a remainder of 1 ... 8 sixteen-feature chunks always has its instantiation)
        // -- the symmetric variant instead of the generic full-square kernel: 40 000 x 512 35.1 -> 12.3 ms per iteration, 20 000 x 2 000
        // 34.2 -> 13.4 ms.  The panel width hardly matters (64 and 256 measure within 2 %: the linear epilogue is two fmas per element);
        // up to 256 features the one pass stays.
        if (params_.kernel_type == LSSVM_KERNEL_LINEAR && opt_.tile_kernel != 1 && opt_.symmetric != 0 && opt_.colslab_limit_mb != 0 && ldx_probe > F64_ONE_PASS_FEATURES) {
            v2_ok = wide_linear_ = true;
        }
        // rbf / polynomial beyond 256 features: the panels inside a sub-tile, either variant
        if (wide_nonlinear_f64(opt_, params_, num_features)) v2_ok = wide_nl_ = true;
    }
    if constexpr (std::is_same_v<T, float>) {
        // rbf / polynomial on more features than the row panel of the split kernels holds in registers (f16x3: 384 rbf, 512 polynomial; bf16x6:
        // 384): feature panels of 128 walked inside a tile (lssvm_tile_f32_wide.hip.hpp), either variant, either plane kind -- decided from
        // the shape and the options alone, so every rank of a sharded solve decides alike.  (The linear kernel has its panel passes above.)
        if (wide_nonlinear(opt_, tile_params_, rbf_direct_, num_features)) v2_ok = wide_nl_ = true;
    }
    sym_ = opt_.symmetric != 0 && v2_ok && !(tile_params_.kernel_type == LSSVM_KERNEL_POLYNOMIAL && tile_params_.degree < 0);
    // the symmetric variant keeps one 128-entry record per evaluated off-diagonal tile of the row-block BAND in flight (see the bands
    // below); colslab_limit_mb = 0 switches the variant off (a rule in the options only, so every rank of a sharded solve decides alike)
    if (sym_ && opt_.colslab_limit_mb == 0) sym_ = false;
    if constexpr (std::is_same_v<T, float>) {
        // 256-row workgroups on block pairs (lssvm_tile_f32_pair.hip.hpp): the symmetric variant of the split kernels on at most 128 features per
        // pass with a hand-scheduled epilogue -- decided from the shape and the options alone (both plane kinds have the kernel), so that the
        // geometry below does not wait for the planes and every rank of a sharded solve decides alike
        const bool poly_generic = tile_params_.kernel_type == LSSVM_KERNEL_POLYNOMIAL && tile_params_.degree != 2 && tile_params_.degree != 3;
        const bool narrow = wide_linear_ ? true : (round_up(static_cast<long>(num_features), 64) <= 128 && v2_eligible(opt_, ldx_probe, rbf_direct_));
        // (rbf: that kernel folds BOTH exponent terms out of the chain -- only while |c| = R2 / 2 stays small and the folded records are on)
        // (round 6: rbf on GRID planes has its 256-row form too -- start values instead of folded terms, so neither condition applies to it)
        const bool rbf_ok = tile_params_.kernel_type != LSSVM_KERNEL_RBF || rbf_grid_ || (opt_.rbf_fold != 0 && rbf_r2_ <= 2.0 * PAIR_FOLD_MAX_C);
        // (below 64 row blocks -- 8 192 points -- the 128-row workgroups have more items to spread over the chip: 3 000 points 13.8 against 19.7 us)
        pair_ = sym_ && opt_.gram_mode != 0 && opt_.mfma_shape >= 3 && !wide_nl_ && !poly_generic && narrow && rbf_ok && num_tiles_ >= PAIR_MIN_TILES;
    }
    lap("exponent scale, path decisions");
    choose_shard_geometry();
    lap("shard geometry (chunk choice by replay)");
    // 256-row workgroups: PERSISTENT launches, the work items drawn from per-XCD counters (for_each_work_item, lssvm_device_common.hip.hpp) instead of one workgroup per
    // item dealt by the hardware -- whose deal is static per XCD (every eighth workgroup, whatever the XCD's pace: the eight clocks of one chip differ by 3-5 %) and in
    // order.  Same box, interleaved, bit-identical: 1 000 000 x 128 rbf 264.9 -> 256.8 ms per iteration (-3.0 %), 200 000 x 256 linear 20.05 -> 19.49 (-3.0 %),
    // 50 000 x 128 unchanged (profiles/r05_ab_pair_queue.log).  LSSVM_MI355_PAIR_QUEUE=0 in the environment: the former launches (A/B runs).
    if (const char *pq = std::getenv("LSSVM_MI355_PAIR_QUEUE"); pair_ && !(pq != nullptr && pq[0] == '0')) {
        queue_.alloc_zero(512, st);
        LSSVM_HIP_CHECK(hipDeviceGetAttribute(&queue_min_items_, hipDeviceAttributeMultiprocessorCount, device_));
    }
    inv_cost_ = static_cast<double>(T(1) / static_cast<T>(params_.cost));  // "1 / params.cost" in real_type, csvm.cpp:297

    // QA_cost = k(x_last, x_last) + 1/C, evaluated on the host in the real type (csvm.cpp:86)
    std::vector<T> last(num_features);
    LSSVM_HIP_CHECK(hipMemcpyAsync(last.data(), X_.data.p + static_cast<size_t>(n_) * X_.ldx, num_features * sizeof(T), hipMemcpyDeviceToHost, st));
    LSSVM_HIP_CHECK(hipStreamSynchronize(st));
    QA_cost_ = static_cast<double>(host_self_kernel<T>(params_, last) + T(1) / static_cast<T>(params_.cost));

    // vectors (zero padded to nvec_)
    for (DevBuf<T> *v : { &q_, &b_, &x_, &r_, &d_, &Ad_, &Kv_, &tmp_ }) v->alloc_zero(static_cast<size_t>(nvec_) + TILE, st);
    Kres_ = Kv_.p;
    ylast_.alloc_zero(num_points, st);
    part_.alloc_zero(static_cast<size_t>(PART_REGIONS) * RED_BLOCKS * 2, st);  // (four sets of partial sums: a kernel reduces its predecessor's while it writes its own)
    sc_.alloc_zero(SC_COUNT, st);
    host_sc_.alloc(SC_COUNT);
    host_delta_.alloc_mapped(1);

    lap("QA_cost, vectors");
    // q from the raw (un-centred) data: bit-compatible fma chains (q_kernel.cpp:18-55)
    {
        const T *xlast = X_.data.p + static_cast<size_t>(n_) * X_.ldx;
        const dim3 grid((n_ + 127) / 128), block(128);
        const T g = static_cast<T>(params_.gamma), c0 = static_cast<T>(params_.coef0);
        switch (params_.kernel_type) {
            case LSSVM_KERNEL_LINEAR: hipLaunchKernelGGL((k_q<KT_LINEAR, T>), grid, block, 0, st, X_.data.p, X_.ldx, X_.dfeat, n_, xlast, params_.degree, g, c0, q_.p); break;
            case LSSVM_KERNEL_POLYNOMIAL: hipLaunchKernelGGL((k_q<KT_POLY, T>), grid, block, 0, st, X_.data.p, X_.ldx, X_.dfeat, n_, xlast, params_.degree, g, c0, q_.p); break;
            default: hipLaunchKernelGGL((k_q<KT_RBF, T>), grid, block, 0, st, X_.data.p, X_.ldx, X_.dfeat, n_, xlast, params_.degree, g, c0, q_.p); break;
        }
        LSSVM_HIP_CHECK(hipGetLastError());
    }
    lap("q");
    // rbf on the matrix cores: centre the data, then c_i = -|x_i|^2 / 2
    if (params_.kernel_type == LSSVM_KERNEL_RBF && !rbf_direct_) {
        center_columns<T>(X_, nullptr, rbf_prescale<T>(params_, v2_eligible_f64(opt_, X_.ldx) || wide_nl_), st);
        half_neg_norms<T>(X_, c_, st);
    }
    lap("centring, norms");
    // polynomial in fp64 on the v2 kernel: fold gamma into the data (x' = sqrt(gamma) x, after q was computed from the raw data), so
    // that the MFMA chain leaves gamma * <x_i, x_j> and the epilogue is the bare integer power -- every vector ALU instruction
    // beside v_mfma_f64 costs matrix-core time (gamma > 0 is a precondition of the kernel, parameter.hpp / csvm.cpp:77)
    if constexpr (std::is_same_v<T, double>) {
        if (params_.kernel_type == LSSVM_KERNEL_POLYNOMIAL && (v2_eligible_f64(opt_, X_.ldx) || wide_nl_) && params_.gamma > 0.0) {
            hipLaunchKernelGGL(k_center<T>, dim3((X_.dfeat + 255) / 256, X_.rows), dim3(256), 0, st, X_.data.p, X_.ldx, X_.dfeat, X_.rows,
                               static_cast<const T *>(nullptr), static_cast<T>(std::sqrt(params_.gamma)));
            LSSVM_HIP_CHECK(hipGetLastError());
            poly_prescaled_ = true;
        }
    }
    if constexpr (std::is_same_v<T, float>) {
        // the (centred, scaled) data once more as operand planes of the split kernels (features in natural order): see make_planes
        if (rbf_grid_) {
            grid_sigma_ = make_grid_planes(X_, rbf_r2_, planes_, c_.p, efac_, st, wide_nl_);
        } else if (!wide_linear_) make_planes(opt_, tile_params_, rbf_direct_, X_, nullptr, planes_, nullptr, st, wide_nl_, false, f16_probe_failed_);
        f16_row_rel_error_ = planes_.f16_row_rel_error;
        row_scaled_ = planes_.row_inv_scale.p != nullptr;
        if (row_scaled_) vs_.alloc_zero(static_cast<size_t>(nvec_) + TILE, st);
        if ((wide_nl_ || pair_) && planes_.mode == 0) throw Error(LSSVM_ERR_INTERNAL, "no operand planes for a path that was chosen from the shape alone");
        if ((wide_nl_ && sym_) || pair_) {
            // the row side of the panels-inside-a-tile kernel: the planes once more, every 16 x 32 block stored as the A fragment a wave loads
            // (lssvm_tile_f32_wide.hip.hpp: the row fragments are re-loaded at every tile-panel -- whole cache lines instead of half lines);
            // the 256-row workgroups load a work item's row panel from it (1 KiB contiguous per load instruction instead of 64 bytes of each of 16 rows)
            const size_t plane_elems = static_cast<size_t>(X_.rows_alloc) * planes_.ldx16;
            planes_frag_.alloc_zero(static_cast<size_t>(planes_.nplanes) * plane_elems, st);
            const size_t pieces = static_cast<size_t>(planes_.nplanes) * X_.rows_alloc * (planes_.ldx16 / 8);
            hipLaunchKernelGGL(k_planes_fragment_major, dim3(static_cast<unsigned>((pieces + 255) / 256)), dim3(256), 0, st, planes_.buf.p, plane_elems, static_cast<int>(X_.rows_alloc), planes_.ldx16,
                               planes_.nplanes, planes_frag_.p);
            LSSVM_HIP_CHECK(hipGetLastError());
        }
        // rbf: folded records while the exponent terms stay small (rbf_r2_ = 2 max|c| in the exponent's unit)
        if (planes_.mode != 0) dc_folded_ = tile_params_.kernel_type == LSSVM_KERNEL_RBF && opt_.rbf_fold != 0 && rbf_r2_ <= FOLD_MAX_R2 && !rbf_grid_;
    }
    lap("operand planes, fragment-major copy");
    interleave_features<T>(X_, st);
    if constexpr (std::is_same_v<T, double>) {
        if (wide_nl_ && sym_) {
            // the row side of the fp64 panels-inside-a-sub-tile kernel: the (prepared) data once more, every 16 x 4 block stored as the A fragment a wave
            // loads (lssvm_tile_f64_wide.hip.hpp) -- after every in-place transformation of X_ above
            Xfrag_.alloc_zero(static_cast<size_t>(X_.rows_alloc) * X_.ldx, st);
            const size_t elems = static_cast<size_t>(X_.rows_alloc) * X_.ldx;
            hipLaunchKernelGGL(k_rows_fragment_major_f64, dim3(static_cast<unsigned>((elems + 255) / 256)), dim3(256), 0, st, X_.data.p, static_cast<int>(X_.rows_alloc), X_.ldx, Xfrag_.p);
            LSSVM_HIP_CHECK(hipGetLastError());
        }
    }
    if ((std::is_same_v<T, float> && (v2_eligible(opt_, X_.ldx, rbf_direct_) || wide_linear_ || wide_nl_)) || (std::is_same_v<T, double> && (v2_eligible_f64(opt_, X_.ldx) || wide_linear_ || wide_nl_))) {
        dc_.alloc_zero(static_cast<size_t>(std::max(num_tiles_, 1)) * 256, st);  // (d_j | c_j) records: 256 reals per 128 columns
    }
    lap("interleave, records");
    build_shard_lists(st);
    lap("work-item lists, slabs, events");
    ev_ready_.create(false);
    ev_consumed_.create(false);
    LSSVM_HIP_CHECK(hipStreamSynchronize(st));
    setup_ms_ = now_ms() - t0;
}

template <typename T>
Problem<T>::~Problem() {
    (void) hipSetDevice(device_);
    if (stream_.s != nullptr) (void) hipStreamSynchronize(stream_.s);
    // members release themselves (events, pinned words, device buffers, then the stream)
}

template <typename T>
TileArgs<T> Problem<T>::tile_args(const T *v_dev) const {
    TileArgs<T> a{};
    a.Xr = X_.data.p;
    a.Xrf = Xfrag_.p;
    a.frag_rows16 = static_cast<int>(X_.rows_alloc / 16);
    a.Xc = X_.data.p;
    a.cr = c_.p;
    a.cc = c_.p;
    a.dvec = v_dev;
    a.dc = dc_.p;
    a.items = sym_ ? items_.p : nullptr;  // (the symmetric variant launches band by band: enqueue_apply_K_local offsets these two)
    a.num_items = num_items_;
    a.colslab = colslab_.p;
    a.pair_origin = 0;
    a.partial = partial_.p;
    a.part_stride = static_cast<long>(part_blocks()) * TILE;
    a.ldx = X_.ldx;
    a.kchunks = X_.ldx / kchunk_of<T>();
    a.ib_begin = ib_begin_;
    a.num_ib = num_ib_;
    a.num_jt = num_tiles_;
    a.jc_tiles = jc_tiles_;
    a.jc_head_tiles = jc_head_tiles_;
    a.jc_head_count = jc_head_count_;
    a.ncols_valid = n_;
    set_kernel_scalars(a, tile_params_, rbf_direct_);
    if (rbf_grid_) {
        a.gamma = static_cast<T>(1.0 / (static_cast<double>(grid_sigma_) * static_cast<double>(grid_sigma_)));  // the chain carries sigma^2 t
        a.er = efac_.p;
        a.rbf_grid = 1;
    }
    if (poly_prescaled_) a.gamma = T(1);
    if constexpr (std::is_same_v<T, float>) {
        a.Xr16f = planes_frag_.p;
        if (planes_.mode != 0) set_plane_args(a, tile_params_, planes_, planes_, static_cast<size_t>(X_.rows_alloc), static_cast<size_t>(X_.rows_alloc));
    }
    a.wide_panels = wide_nl_ ? 1 : 0;
    set_launch_options(a, opt_);
    a.row_pair = pair_ ? 1 : 0;
    a.dc_folded = dc_folded_ ? 1 : 0;
    return a;
}

template <typename T>
void Problem<T>::drain_events() {
    for (EvPair &e : events_) {
        if (e.pending && hipEventQuery(e.b.e) == hipSuccess) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, e.a.e, e.b.e) == hipSuccess) {
                matvec_ms_ += ms;  // the band launches of one matvec add up; the matvec is counted once
                if (e.first_of_matvec) ++matvec_timed_;
            }
            e.pending = false;
        }
    }
}

template <typename T>
void Problem<T>::enqueue_apply_K_local(const T *v_dev, bool zero_first) {
    // this shard's part of the implicit K * v: tile kernel over its row blocks (band by band), slabs added in a fixed order
    hipStream_t st = stream_.s;
    // sampled matvecs: every `stride`-th, starting with the LAST of each run of `stride`, plus launches 1 ... 4 (a run of a few iterations still reports a
    // kernel time; no more than four: the first launches of a solve run a few per cent slow and would weigh on the average of a run of a hundred) -- never launch 0, the cold first matvec of cg_begin, which would otherwise carry `stride` times its weight in the average
    const uint64_t stride = static_cast<uint64_t>(event_stride());
    const bool timed = matvec_launches_ != 0 && (matvec_launches_ % stride == stride - 1 || matvec_launches_ <= 4);  // (short runs: every one of the first four after launch 0)
    ++matvec_launches_;
    auto free_event = [&]() -> EvPair * {
        if (!timed) return nullptr;
        for (int attempt = 0; attempt < 2; ++attempt) {
            for (EvPair &e : events_) {
                if (!e.pending) return &e;
            }
            drain_events();
        }
        return nullptr;
    };
    // symmetric variant: row sums of the device's blocks and the mirrored column sums of every band are ADDED into K*v (and, sharded, every rank
    // adds into all earlier rows): start from zero -- by the kernel that packs the records where there is one, by a memset otherwise
    const bool clear = zero_first || sym_;
    if constexpr (std::is_same_v<T, float>) {
        if (row_scaled_) {  // K v = D (Xs Xs^T) (D v): the tile kernels see D v (make_planes: a power-of-two scale per row)
            hipLaunchKernelGGL(k_scale_vector, dim3((nvec_ + 255) / 256), dim3(256), 0, st, v_dev, planes_.row_inv_scale.p, nvec_, vs_.p);
            v_dev = vs_.p;
        }
    }
    const bool prepacked = d_packed_ && v_dev == d_.p;  // k_update_d has left the records of d_ and cleared K*v (pack_for_d)
    d_packed_ = false;                                  // (either they are consumed now, or dc_ is about to hold another vector's)
    const bool pack = dc_.p != nullptr && num_ib_ > 0 && !prepacked;
    if (clear && !pack && !prepacked) LSSVM_HIP_CHECK(hipMemsetAsync(Kv_.p, 0, static_cast<size_t>(nvec_) * sizeof(T), st));
    if (num_ib_ <= 0) return;
    TileArgs<T> a = tile_args(v_dev);
    if (pack) {  // v2 kernels: pack (d_j | c_j) records for the LDS-DMA
        const int ncols = num_tiles_ * TILE;
        const int nzero = clear ? static_cast<int>(nvec_) : 0;
        const int nthreads = std::max(ncols, nzero);
        if constexpr (std::is_same_v<T, float>) {
            hipLaunchKernelGGL(k_pack_dc, dim3((nthreads + 255) / 256), dim3(256), 0, st, v_dev, c_.p, ncols, dc_.p, rbf_grid_ ? 2 : a.dc_folded, Kv_.p, nzero, rbf_grid_ ? efac_.p : static_cast<const float *>(nullptr));
        } else {
            hipLaunchKernelGGL(k_pack_dc_f64, dim3((nthreads + 255) / 256), dim3(256), 0, st, v_dev, c_.p, ncols, dc_.p, Kv_.p, nzero);
        }
    }
    const int nrows = num_ib_ * TILE;
    // wide linear problems: one pass per panel of 512 features, every pass ADDS its K_p * v (rows and mirrored columns) into K*v
    const int npanels = passes_per_matvec();
    const int panel_features = LINEAR_PANEL_FEATURES;
    if (sym_) {
        bool first = true;
        for (int panel = 0; panel < npanels; ++panel) {
            TileArgs<T> ap = a;
            if constexpr (std::is_same_v<T, float>) {
                if (wide_linear_) {
                    if (ap.Xr16f != nullptr) ap.Xr16f += static_cast<size_t>(panel) * (panel_features / 64) * (X_.rows_alloc / 16) * 1024;  // (the 64-feature chunk is the outermost index of a plane)
                    ap.Xr16 += static_cast<size_t>(panel) * panel_features;
                    ap.Xc16 += static_cast<size_t>(panel) * panel_features;
                    ap.nk64 = std::min(panel_features, planes_.ldx16 - panel * panel_features) / 64;
                }
            } else {
                if (wide_linear_) {  // (row-major fp64 data: a panel is a column range of X, the row stride stays)
                    ap.Xr += static_cast<size_t>(panel) * panel_features;
                    ap.Xc += static_cast<size_t>(panel) * panel_features;
                    ap.kchunks = std::min(panel_features, X_.ldx - panel * panel_features) / F64_KC;
                }
            }
            for (const Band &band : bands_) {
                TileArgs<T> ab = ap;
                ab.items = items_.p + band.item_begin;
                ab.num_items = band.item_count;
                ab.pair_origin = band.pair_origin;
                if (queue_.p != nullptr && band.item_count > queue_min_items_) {  // (a launch of no more items than CUs: one workgroup per item, no counters -- 2 % faster there)
                    ab.queue = queue_.p + 256 * queue_set_;
                    ab.queue_next = queue_.p + 256 * (1 - queue_set_);
                    ab.queue_grid = queue_min_items_;
                    queue_set_ ^= 1;
                }
                EvPair *ev = free_event();
                if (ev != nullptr) LSSVM_HIP_CHECK(hipEventRecord(ev->a.e, st));
                launch_tile_kernel<T>(ab, tile_params_.kernel_type, rbf_direct_, num_jc_, st);
                if (ev != nullptr) {
                    LSSVM_HIP_CHECK(hipEventRecord(ev->b.e, st));
                    ev->pending = true;
                    ev->first_of_matvec = first;
                }
                first = false;
                // fold the band's mirrored column sums into K*v before the next band re-uses the slab
                if (band.ib_end > 1) {
                    if constexpr (std::is_same_v<T, float>) {
                        // (block pairs: one record per pair and column tile, kept as the record of the pair's second -- odd -- block)
                        hipLaunchKernelGGL((k_reduce_colslab<T, 128>), dim3(band.ib_end - 1), dim3(1024), 0, st, colslab_.p, band.pair_origin, band.ib_begin, pair_ ? round_up(band.ib_end, 2) : band.ib_end, pair_ ? 2 : 1, Kv_.p, rbf_grid_ ? efac_.p : static_cast<const T *>(nullptr));
                    } else {  // fp64: records per 64-column sub-tile
                        hipLaunchKernelGGL((k_reduce_colslab<T, 64>), dim3(2 * (band.ib_end - 1)), dim3(1024), 0, st, colslab_.p, band.pair_origin, band.ib_begin, band.ib_end, 1, Kv_.p);
                    }
                }
            }
            // rows of this device's blocks: the slabs of the column chunks that exist for each block, added on top
            hipLaunchKernelGGL(k_reduce_partials_sym<T>, dim3((nrows + 255) / 256), dim3(256), 0, st, partial_.p, a.part_stride, jc_tiles_, jc_head_tiles_, jc_head_count_, ib_begin_, nrows, Kv_.p, 1);
        }
    } else {
        EvPair *ev = free_event();
        if (ev != nullptr) LSSVM_HIP_CHECK(hipEventRecord(ev->a.e, st));
        launch_tile_kernel<T>(a, tile_params_.kernel_type, rbf_direct_, num_jc_, st);
        if (ev != nullptr) {
            LSSVM_HIP_CHECK(hipEventRecord(ev->b.e, st));
            ev->pending = true;
            ev->first_of_matvec = true;
        }
        hipLaunchKernelGGL(k_reduce_partials<T>, dim3((nrows + 255) / 256), dim3(256), 0, st, partial_.p, a.part_stride, num_jc_, ib_begin_ * TILE, nrows, Kv_.p);
    }
    if constexpr (std::is_same_v<T, float>) {
        // (every entry this shard has added into: the symmetric variant mirrors into the rows of earlier blocks too; a sharded solve exchanges the scaled partial vectors)
        if (row_scaled_) hipLaunchKernelGGL(k_scale_vector, dim3((nvec_ + 255) / 256), dim3(256), 0, st, Kv_.p, planes_.row_inv_scale.p, nvec_, Kv_.p);
    }
    LSSVM_HIP_CHECK(hipGetLastError());
}

/* The records of the NEXT implicit matvec -- always K * d in the CG loop -- are packed by the kernel that updates d (k_update_d) instead of a k_pack_dc launch
 * in front of the tile kernel: what that kernel needs, and the note that it has been done.  `zero_first` as enqueue_apply_K_local's. */
template <typename T>
PackDc<T> Problem<T>::pack_for_d(bool zero_first) {
    PackDc<T> pk;
    if (dc_.p == nullptr || num_ib_ <= 0 || row_scaled_) return pk;  // (row-scaled planes: the records are those of D d, packed per matvec)
    pk.dc = dc_.p;
    pk.cc = c_.p;
    pk.ncols = num_tiles_ * TILE;
    pk.zero = Kv_.p;
    pk.nzero = (zero_first || sym_) ? static_cast<int>(nvec_) : 0;
    if constexpr (std::is_same_v<T, float>) {
        pk.folded = rbf_grid_ ? 2 : (dc_folded_ ? 1 : 0);
        pk.efac = rbf_grid_ ? efac_.p : nullptr;
    }
    d_packed_ = true;
    return pk;
}

template <typename T>
void Problem<T>::enqueue_sum_and_qdot(const T *v_dev, int slot_sum, int slot_q) {
    hipLaunchKernelGGL(k_sum_and_qdot<T>, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, stream_.s, v_dev, q_.p, n_, part(PART_SUMS));
    hipLaunchKernelGGL(k_finish2, dim3(1), dim3(RED_THREADS), 0, stream_.s, part(PART_SUMS), sc_.p, slot_sum, slot_q);
    LSSVM_HIP_CHECK(hipGetLastError());
}

template class Problem<float>;
template class Problem<double>;

/* ------------------------------------------------------------------ shared with lssvm_predict.hip ------------------------------------------------------------------ */
template float rbf_prescale<float>(const lssvm_params &, bool);
template double rbf_prescale<double>(const lssvm_params &, bool);
template void center_columns<float>(DeviceMatrix<float> &, DeviceMatrix<float> *, float, hipStream_t);
template void center_columns<double>(DeviceMatrix<double> &, DeviceMatrix<double> *, double, hipStream_t);
template void half_neg_norms<float>(const DeviceMatrix<float> &, DevBuf<float> &, hipStream_t);
template void half_neg_norms<double>(const DeviceMatrix<double> &, DevBuf<double> &, hipStream_t);
template void interleave_features<float>(DeviceMatrix<float> &, hipStream_t);
template void interleave_features<double>(DeviceMatrix<double> &, hipStream_t);
template void set_kernel_scalars<float>(TileArgs<float> &, const lssvm_params &, bool);
template void set_kernel_scalars<double>(TileArgs<double> &, const lssvm_params &, bool);
template void set_launch_options<float>(TileArgs<float> &, const Options &);
template void set_launch_options<double>(TileArgs<double> &, const Options &);
template void column_means<float>(const DeviceMatrix<float> &, DevBuf<float> &, hipStream_t);
template void column_means<double>(const DeviceMatrix<double> &, DevBuf<double> &, hipStream_t);
template double max_centred_sqnorm<float>(const DeviceMatrix<float> &, const DevBuf<float> &, hipStream_t);
template double max_centred_sqnorm<double>(const DeviceMatrix<double> &, const DevBuf<double> &, hipStream_t);
template bool rbf_wants_direct_form<float>(const Options &, const lssvm_params &, const DeviceMatrix<float> &, const DeviceMatrix<float> *, hipStream_t, double *);
template bool rbf_wants_direct_form<double>(const Options &, const lssvm_params &, const DeviceMatrix<double> &, const DeviceMatrix<double> *, hipStream_t, double *);

/* the set-up kernels that are no templates live in this translation unit (LSSVM_KERNELS_SETUP): the predict path enqueues them through these */
void enqueue_pack_records(const float *dvec, const float *cc, int ncols_padded, float *dc, int folded, const float *efac, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_dc, dim3((ncols_padded + 255) / 256), dim3(256), 0, s, dvec, cc, ncols_padded, dc, folded, static_cast<float *>(nullptr), 0, efac);
    LSSVM_HIP_CHECK(hipGetLastError());
}
void enqueue_pack_records(const double *dvec, const double *cc, int ncols_padded, double *dc, int, const double *, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_dc_f64, dim3((ncols_padded + 255) / 256), dim3(256), 0, s, dvec, cc, ncols_padded, dc, static_cast<double *>(nullptr), 0);
    LSSVM_HIP_CHECK(hipGetLastError());
}
void enqueue_planes_fragment_major(const uint16_t *planes, size_t plane_elems, int rows_alloc, int ldx16, int nplanes, uint16_t *frag, hipStream_t s) {
    const size_t pieces = static_cast<size_t>(nplanes) * rows_alloc * (ldx16 / 8);
    hipLaunchKernelGGL(k_planes_fragment_major, dim3(static_cast<unsigned>((pieces + 255) / 256)), dim3(256), 0, s, planes, plane_elems, rows_alloc, ldx16, nplanes, frag);
    LSSVM_HIP_CHECK(hipGetLastError());
}

}  // namespace lssvm
