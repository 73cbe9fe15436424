/*
 * lssvm_predict.hip -- csvm::predict_values behind the C ABI (reference: src/plssvm/backends/OpenMP/csvm.cpp:188-227,
 * include/plssvm/backends/HIP/predict_kernel.hip.hpp:34-117, gpu_csvm.hpp:656-730): the one-shot call, calculate_w, and the resident predictor.  A rectangular instance
 * of the tile kernels -- rows = the points to predict, columns = the support vectors -- prepared with the helpers of lssvm_problem.hip (declared in lssvm_problem.hip.hpp).
 */
#include "lssvm_problem.hip.hpp"

#include "lssvm_kernels.hip.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

namespace lssvm {

/* ------------------------------------------------------------------ predict path ------------------------------------------------------------------ */
/* What the rectangular 256-row kernel (tile_matvec_f32_pair_rect) needs besides the planes: the row side once more fragment-major (a wave's load instruction reads
 * 1 KiB in one piece), the item list of the whole rectangle in XCD-lane order (xcd_lane_order: the workgroups of an XCD share their column stream in its L2), and the
 * counters of a persistent launch.  Used by the one-shot predict_values and by the resident predictor. */
struct RectSetup {
    DevBuf<uint16_t> frag;
    DevBuf<int2> items;
    DevBuf<unsigned> queue;
};
static void setup_rect_launch(TileArgs<float> &ta, RectSetup &rs, const PlaneSet &planesP, int rows_alloc, int num_ib, int num_jc, hipStream_t s) {
    const size_t plane_elems = static_cast<size_t>(rows_alloc) * planesP.ldx16;
    rs.frag.alloc_zero(static_cast<size_t>(planesP.nplanes) * plane_elems, s);
    enqueue_planes_fragment_major(planesP.buf.p, plane_elems, rows_alloc, planesP.ldx16, planesP.nplanes, rs.frag.p, s);
    const int pairs = num_ib / 2;
    std::vector<std::vector<int2>> by_chunk(static_cast<size_t>(num_jc));
    for (int jc = 0; jc < num_jc; ++jc) {
        by_chunk[static_cast<size_t>(jc)].reserve(static_cast<size_t>(pairs));
        for (int pr = 0; pr < pairs; ++pr) by_chunk[static_cast<size_t>(jc)].push_back(make_int2(2 * pr, jc));
    }
    const std::vector<int2> items = xcd_lane_order(by_chunk);
    rs.items.alloc_zero(items.size(), s);
    LSSVM_HIP_CHECK(hipMemcpyAsync(rs.items.p, items.data(), items.size() * sizeof(int2), hipMemcpyHostToDevice, s));
    LSSVM_HIP_CHECK(hipStreamSynchronize(s));  // `items` goes out of scope
    int cus = 256;
    LSSVM_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    ta.items = rs.items.p;
    ta.num_items = static_cast<int>(items.size());
    if (ta.num_items > cus) {
        rs.queue.alloc_zero(512, s);
        ta.queue = rs.queue.p;
        ta.queue_next = rs.queue.p + 256;
        ta.queue_grid = cus;
    }
    ta.Xr16f = rs.frag.p;
    ta.row_pair = 1;
    ta.rect = 1;
}

/* out_p = w . x_p - rho: one pass over the points, HBM bound.  A group of L lanes (a power of two, at most 64) owns a point and reads its row in 16-byte pieces, so
 * that a wave's load instruction covers 1 KiB of consecutive memory wherever a row has at least 16 bytes x L. */
template <typename T>
static void launch_predict_linear(const DeviceMatrix<T> &P, const T *w, T rho, T *out, hipStream_t s) {
    constexpr int V = 16 / static_cast<int>(sizeof(T));  // elements per 16-byte piece
    const int pieces = P.ldx / V;                         // (ldx is a multiple of the k-chunk: 32 floats / 16 doubles)
    int L = 1;
    while (2 * L <= std::min(pieces, 64)) L *= 2;
    const int rows_per_block = 256 / L;
    const dim3 grid(static_cast<unsigned>((P.rows + rows_per_block - 1) / rows_per_block));
    switch (L) {
        case 4: hipLaunchKernelGGL((k_predict_linear_rows<T, 4>), grid, dim3(256), 0, s, P.data.p, P.ldx, P.rows, w, rho, out); break;
        case 8: hipLaunchKernelGGL((k_predict_linear_rows<T, 8>), grid, dim3(256), 0, s, P.data.p, P.ldx, P.rows, w, rho, out); break;
        case 16: hipLaunchKernelGGL((k_predict_linear_rows<T, 16>), grid, dim3(256), 0, s, P.data.p, P.ldx, P.rows, w, rho, out); break;
        case 32: hipLaunchKernelGGL((k_predict_linear_rows<T, 32>), grid, dim3(256), 0, s, P.data.p, P.ldx, P.rows, w, rho, out); break;
        default: hipLaunchKernelGGL((k_predict_linear_rows<T, 64>), grid, dim3(256), 0, s, P.data.p, P.ldx, P.rows, w, rho, out); break;
    }
}

template <typename T>
void calculate_w(const T *sv, size_t nsv, size_t nfeat, const T *alpha, T *w_out) {
    LSSVM_REQUIRE(sv != nullptr && nsv > 0, "The support vectors may not be empty!");                       // csvm.cpp:256
    LSSVM_REQUIRE(nfeat > 0, "Each support vector must at least contain one feature!");                     // csvm.cpp:257
    LSSVM_REQUIRE(alpha != nullptr && w_out != nullptr, "The alpha array may not be empty!");               // csvm.cpp:259
    select_device_checked(0);
    hipStream_t s = nullptr;
    DeviceMatrix<T> S;
    S.upload(sv, LSSVM_MEM_HOST, nsv, nfeat, 0, s);
    DevBuf<T> a, w;
    a.alloc_zero(nsv, s);
    w.alloc_zero(nfeat, s);
    LSSVM_HIP_CHECK(hipMemcpyAsync(a.p, alpha, nsv * sizeof(T), hipMemcpyHostToDevice, s));
    // w[f] = sum_i alpha_i sv[i][f]: partial sums over blocks of 256 support vectors (coalesced across the features), then the blocks in order -- the reference's chain
    // (csvm.cpp:255-280) is one sequential fma chain per feature; 128 threads walking 50 000 rows each took 10 ms where the matrix is read in 10 us
    const int rows_per_block = 256;
    const int nblocks = (S.rows + rows_per_block - 1) / rows_per_block;
    DevBuf<double> part;
    part.alloc_zero(static_cast<size_t>(nblocks) * S.ldx, s);
    hipLaunchKernelGGL(k_calculate_w_stage1<T>, dim3(nblocks, (S.ldx + 255) / 256), dim3(256), 0, s, S.data.p, S.ldx, S.rows, rows_per_block, a.p, part.p);
    hipLaunchKernelGGL(k_calculate_w_stage2<T>, dim3((S.dfeat + 255) / 256), dim3(256), 0, s, part.p, nblocks, S.ldx, S.dfeat, w.p);
    LSSVM_HIP_CHECK(hipGetLastError());
    LSSVM_HIP_CHECK(hipMemcpyAsync(w_out, w.p, nfeat * sizeof(T), hipMemcpyDeviceToHost, s));
    LSSVM_HIP_CHECK(hipStreamSynchronize(s));
}

template <typename T>
static void predict_values_impl(const Options &opt, const lssvm_params &params, const T *sv, size_t nsv, size_t nfeat, const T *alpha, T rho, T *w_inout, int *w_valid, const T *points,
                                size_t npoints, T *out, lssvm_predict_info &info) {
    check_params(&params);
    LSSVM_REQUIRE(sv != nullptr && nsv > 0, "The support vectors must not be empty!");                       // csvm.cpp:189
    LSSVM_REQUIRE(nfeat > 0, "The support vectors must contain at least one feature!");                      // csvm.cpp:190
    LSSVM_REQUIRE(alpha != nullptr, "The number of support vectors and number of weights must be the same!");  // csvm.cpp:192
    LSSVM_REQUIRE(points != nullptr && npoints > 0, "The data points to predict must not be empty!");        // csvm.cpp:194
    LSSVM_REQUIRE(out != nullptr && w_valid != nullptr, "out / w_valid must not be NULL");
    select_device_checked(0);
    hipStream_t s = nullptr;
    const double t0 = now_ms();
    Event ev_a, ev_b;  // around the kernel that does the product (what the reference times as "predict", gpu_csvm.hpp:656-730, is the whole call: total_ms)
    ev_a.create(true);
    ev_b.create(true);
    const auto finish_info = [&](double t_kernel_enqueued) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ev_a.e, ev_b.e) == hipSuccess) info.kernel_ms = ms;
        info.total_ms = now_ms() - t0;
        info.setup_ms = t_kernel_enqueued - t0;
    };

    if (params.kernel_type == LSSVM_KERNEL_LINEAR) {
        LSSVM_REQUIRE(w_inout != nullptr, "w must have num_features entries for the linear kernel");
        if (!*w_valid) {  // csvm.cpp:204-207
            calculate_w<T>(sv, nsv, nfeat, alpha, w_inout);
            *w_valid = 1;
        }
        DeviceMatrix<T> P;
        P.upload(points, LSSVM_MEM_HOST, npoints, nfeat, 0, s);
        DevBuf<T> w, o;
        w.alloc_zero(static_cast<size_t>(P.ldx), s);  // (zero padded like the points' rows: the kernel reads whole 16-byte pieces)
        o.alloc_zero(npoints, s);
        LSSVM_HIP_CHECK(hipMemcpyAsync(w.p, w_inout, nfeat * sizeof(T), hipMemcpyHostToDevice, s));
        LSSVM_HIP_CHECK(hipStreamSynchronize(s));
        const double t_kernel = now_ms();
        LSSVM_HIP_CHECK(hipEventRecord(ev_a.e, s));
        launch_predict_linear<T>(P, w.p, rho, o.p, s);
        LSSVM_HIP_CHECK(hipEventRecord(ev_b.e, s));
        LSSVM_HIP_CHECK(hipGetLastError());
        LSSVM_HIP_CHECK(hipMemcpyAsync(out, o.p, npoints * sizeof(T), hipMemcpyDeviceToHost, s));
        LSSVM_HIP_CHECK(hipStreamSynchronize(s));
        finish_info(t_kernel);
        return;
    }

    // polynomial / rbf: out_p = sum_i alpha_i k(sv_i, p) - rho : a rectangular instance of the tile kernel
    DeviceMatrix<T> S, P;
    S.upload(sv, LSSVM_MEM_HOST, nsv, nfeat, 0, s);
    // (the points padded to whole PAIRS of row blocks: the rectangular 256-row kernel below works on pairs; the padding is zero rows whose sums nobody reads)
    P.upload(points, LSSVM_MEM_HOST, npoints, nfeat, static_cast<size_t>(round_up(static_cast<long>(npoints), 2 * TILE)), s);
    DevBuf<T> cS, cP;
    double rbf_r2 = 0.0;
    bool rbf_direct = rbf_wants_direct_form<T>(opt, params, S, &P, s, &rbf_r2);  // same rule as the training matvec (Problem<T>)
    bool rbf_grid = false;
    if constexpr (std::is_same_v<T, float>) {
        if (rbf_wants_grid_planes(opt, params, nfeat, rbf_r2)) {
            rbf_grid = true;
            rbf_direct = false;
        }
    }
    DevBuf<T> eS, eP;  // grid planes: the folded factors E of the support vectors and of the points
    float grid_sigma = 1.0f;
    int dc_folded = 0;
    if (params.kernel_type == LSSVM_KERNEL_RBF && !rbf_direct) {
        center_columns<T>(S, &P, rbf_prescale<T>(params, v2_eligible_f64(opt, S.ldx) || wide_nonlinear_f64(opt, params, nfeat)), s);
        half_neg_norms<T>(S, cS, s);
        half_neg_norms<T>(P, cP, s);
    }
    bool wide = false;  // rbf / polynomial beyond the one-pass kernels: feature panels inside a tile (full-square instance)
    if constexpr (std::is_same_v<T, float>) {
        wide = wide_nonlinear(opt, params, rbf_direct, nfeat);
    } else {
        wide = wide_nonlinear_f64(opt, params, nfeat);
    }
    const bool v2 = wide || (std::is_same_v<T, float> ? v2_eligible(opt, S.ldx, rbf_direct) : v2_eligible_f64(opt, S.ldx));
    bool poly_prescaled = false;
    if constexpr (std::is_same_v<T, double>) {
        // the fp64 v2 kernel evaluates the polynomial on data that carries sqrt(gamma) (see Problem<T>'s constructor)
        if (v2 && params.kernel_type == LSSVM_KERNEL_POLYNOMIAL && params.gamma > 0.0) {
            const T sc = static_cast<T>(std::sqrt(params.gamma));
            hipLaunchKernelGGL(k_center<T>, dim3((S.dfeat + 255) / 256, S.rows), dim3(256), 0, s, S.data.p, S.ldx, S.dfeat, S.rows, static_cast<const T *>(nullptr), sc);
            hipLaunchKernelGGL(k_center<T>, dim3((P.dfeat + 255) / 256, P.rows), dim3(256), 0, s, P.data.p, P.ldx, P.dfeat, P.rows, static_cast<const T *>(nullptr), sc);
            LSSVM_HIP_CHECK(hipGetLastError());
            poly_prescaled = true;
        }
    }
    // fp32: both sides once more as operand planes (the split kernels, full-square instance: rows = points, columns = support vectors)
    PlaneSet planesS, planesP;
    if constexpr (std::is_same_v<T, float>) {
        if (v2 && rbf_grid) {
            grid_sigma = make_grid_planes(S, rbf_r2, planesS, cS.p, eS, s, wide);
            (void) make_grid_planes(P, rbf_r2, planesP, cP.p, eP, s, wide);  // (the same exponent scale: the same grid and the same sigma)
        } else if (v2) {
            make_planes(opt, params, rbf_direct, S, &P, planesS, &planesP, s, wide);
            if (wide && planesS.mode == 0) throw Error(LSSVM_ERR_INTERNAL, "no operand planes for the wide rbf / polynomial path");
            if (planesS.mode != 0) dc_folded = (params.kernel_type == LSSVM_KERNEL_RBF && opt.rbf_fold != 0 && rbf_r2 <= FOLD_MAX_R2 && !rbf_grid) ? 1 : 0;
        }
    }
    interleave_features<T>(S, s);
    interleave_features<T>(P, s);
    const int num_jt = S.rows_alloc / TILE;
    const int num_ib = P.rows_alloc / TILE;
    // Round 6: from 64 row blocks of points on, on at most 128 features, the RECTANGULAR 256-row kernel (tile_matvec_f32_pair_rect, lssvm_tile_f32_pair.hip.hpp) -- eight
    // waves of a pair of row blocks share one stream of the support vectors' planes, in persistent launches that draw their items from per-XCD counters, like the
    // training matvec's kernel; the 128-row full-square kernels ran this product at 0.49 of the 16-bit peak where the solve's kernel reaches 0.57 (200 000 points x
    // 50 000 support vectors x 128, gpurun_out/r06_bench_default_1.json).  Same conditions as Problem<float>'s pair_: a split mode, no run-time integer power, rbf
    // with both exponent terms folded (|c| <= PAIR_FOLD_MAX_C).
    bool rect = false;
    if constexpr (std::is_same_v<T, float>) {
        const bool poly_generic = params.kernel_type == LSSVM_KERNEL_POLYNOMIAL && params.degree != 2 && params.degree != 3;
        const bool rbf_ok = params.kernel_type != LSSVM_KERNEL_RBF || (dc_folded != 0 && rbf_r2 <= 2.0 * PAIR_FOLD_MAX_C);
        rect = v2 && !wide && !rbf_grid && !rbf_direct && planesS.mode != 0 && planesS.ldx16 <= 128 && !poly_generic && rbf_ok && opt.mfma_shape >= 3 && num_ib >= PAIR_MIN_TILES;
    }
    // column tiles per work item: the option, or automatically about 4096 work items (see Problem<T>'s constructor); 256-row items: about eight per CU, at most 64 tiles
    const long rect_tiles = std::min<long>(64, std::max<long>(4, (static_cast<long>(num_ib / 2) * num_jt + 1024) / 2048));
    const int jc_tiles = opt.j_chunk_tiles > 0
                             ? static_cast<int>(opt.j_chunk_tiles)
                             : (rect ? static_cast<int>(rect_tiles) : static_cast<int>(std::min<long>(16, std::max<long>(2, (static_cast<long>(num_ib) * num_jt + 2048) / 4096))));
    const int num_jc = (num_jt + jc_tiles - 1) / jc_tiles;
    DevBuf<T> a, partial, Kv, o;
    a.alloc_zero(S.rows_alloc, s);
    partial.alloc_zero(static_cast<size_t>(num_jc) * P.rows_alloc, s);
    Kv.alloc_zero(P.rows_alloc, s);
    o.alloc_zero(npoints, s);
    LSSVM_HIP_CHECK(hipMemcpyAsync(a.p, alpha, nsv * sizeof(T), hipMemcpyHostToDevice, s));

    TileArgs<T> ta{};
    ta.Xr = P.data.p;
    ta.Xc = S.data.p;
    ta.cr = cP.p;
    ta.cc = cS.p;
    ta.dvec = a.p;
    // rectangular instance of the tile kernel (full square variant): the v2 kernel when the feature count allows, with the
    // (alpha_j | c_j) records of the support vectors packed for its LDS-DMA
    DevBuf<T> dc;
    if (v2 && !(std::is_same_v<T, double> && params.kernel_type == LSSVM_KERNEL_POLYNOMIAL && !poly_prescaled)) {
        dc.alloc_zero(static_cast<size_t>(num_jt) * 256, s);
        const int ncols = num_jt * TILE;
        if constexpr (std::is_same_v<T, float>) {
            enqueue_pack_records(a.p, cS.p, ncols, dc.p, rbf_grid ? 2 : dc_folded, rbf_grid ? eS.p : static_cast<const float *>(nullptr), s);
        } else {
            enqueue_pack_records(a.p, cS.p, ncols, dc.p, 0, static_cast<const double *>(nullptr), s);
        }
    }
    ta.dc = dc.p;
    ta.dc_folded = dc_folded;
    ta.partial = partial.p;
    ta.part_stride = P.rows_alloc;
    ta.ldx = S.ldx;
    ta.kchunks = S.ldx / kchunk_of<T>();
    ta.ib_begin = 0;
    ta.num_ib = num_ib;
    ta.num_jt = num_jt;
    ta.jc_tiles = jc_tiles;
    ta.ncols_valid = S.rows;
    set_kernel_scalars(ta, params, rbf_direct);
    if (poly_prescaled) ta.gamma = T(1);
    if constexpr (std::is_same_v<T, float>) {
        if (planesS.mode != 0) set_plane_args(ta, params, planesS, planesP, static_cast<size_t>(S.rows_alloc), static_cast<size_t>(P.rows_alloc));
        if (rbf_grid) {
            ta.gamma = static_cast<T>(1.0 / (static_cast<double>(grid_sigma) * static_cast<double>(grid_sigma)));
            ta.er = eP.p;
            ta.rbf_grid = 1;
        }
    }
    ta.wide_panels = wide ? 1 : 0;
    set_launch_options(ta, opt);
    RectSetup rect_setup;
    if constexpr (std::is_same_v<T, float>) {
        if (rect) setup_rect_launch(ta, rect_setup, planesP, P.rows_alloc, num_ib, num_jc, s);
    }
    LSSVM_HIP_CHECK(hipStreamSynchronize(s));
    const double t_kernel = now_ms();
    // (measurement aid: LSSVM_MI355_PREDICT_REPEAT=k in the environment launches the product kernel k times -- it overwrites its slabs, the result is the same -- and times
    // the LAST launch: what the kernel takes once the chip's clocks have settled, beside the first launch after the set-up's idle gaps that a single call measures)
    int repeat = 1;
    if (const char *rep = std::getenv("LSSVM_MI355_PREDICT_REPEAT"); rep != nullptr) repeat = std::min(std::max(std::atoi(rep), 1), 64);
    for (int k = 0; k + 1 < repeat; ++k) {
        launch_tile_kernel<T>(ta, params.kernel_type, rbf_direct, num_jc, s);
        if (ta.queue != nullptr) std::swap(ta.queue, ta.queue_next);  // (a persistent launch zeroes the OTHER set of counters)
    }
    LSSVM_HIP_CHECK(hipEventRecord(ev_a.e, s));
    launch_tile_kernel<T>(ta, params.kernel_type, rbf_direct, num_jc, s);
    LSSVM_HIP_CHECK(hipEventRecord(ev_b.e, s));
    hipLaunchKernelGGL(k_reduce_partials<T>, dim3((P.rows_alloc + 255) / 256), dim3(256), 0, s, partial.p, ta.part_stride, num_jc, 0, P.rows_alloc, Kv.p);
    hipLaunchKernelGGL(k_sub_rho<T>, dim3((P.rows + 255) / 256), dim3(256), 0, s, Kv.p, P.rows, rho, o.p);
    LSSVM_HIP_CHECK(hipGetLastError());
    LSSVM_HIP_CHECK(hipMemcpyAsync(out, o.p, npoints * sizeof(T), hipMemcpyDeviceToHost, s));
    LSSVM_HIP_CHECK(hipStreamSynchronize(s));
    finish_info(t_kernel);
    info.gram_mode = (planesS.mode != 0 && dc.p != nullptr) ? (rbf_grid ? 3 : planesS.mode) : 0;
    info.rbf_direct = rbf_direct ? 1 : 0;
    info.rbf_exponent_scale = rbf_r2;
    info.f16_row_rel_error = planesS.f16_row_rel_error;
}

/* csvm::predict_values behind the C ABI.  fp32 rbf with rbf_form 0: where the grid planes chosen from the exponent scale do not represent the data, the call runs
 * again on the formula-exact kernel (as Solver's constructor does for the training problem). */
template <typename T>
void predict_values(const Options &opt, const lssvm_params &params, const T *sv, size_t nsv, size_t nfeat, const T *alpha, T rho, T *w_inout, int *w_valid, const T *points,
                    size_t npoints, T *out, lssvm_predict_info *info) {
    lssvm_predict_info local{};
    local.f16_row_rel_error = -1.0;
    try {
        predict_values_impl<T>(opt, params, sv, nsv, nfeat, alpha, rho, w_inout, w_valid, points, npoints, out, local);
    } catch (const GridPlanesUnfit &) {
        if (opt.rbf_form != 0) throw;
        Options direct = opt;
        direct.rbf_form = 1;
        local = lssvm_predict_info{};
        local.f16_row_rel_error = -1.0;
        predict_values_impl<T>(direct, params, sv, nsv, nfeat, alpha, rho, w_inout, w_valid, points, npoints, out, local);
    }
    if (info != nullptr) *info = local;
}

template void predict_values<float>(const Options &, const lssvm_params &, const float *, size_t, size_t, const float *, float, float *, int *, const float *, size_t, float *, lssvm_predict_info *);
template void predict_values<double>(const Options &, const lssvm_params &, const double *, size_t, size_t, const double *, double, double *, int *, const double *, size_t, double *, lssvm_predict_info *);
template void calculate_w<float>(const float *, size_t, size_t, const float *, float *);
template void calculate_w<double>(const double *, size_t, size_t, const double *, double *);

/* ------------------------------------------------------------------ the resident predictor ------------------------------------------------------------------ */
template <typename T>
class Predictor final : public PredictorBase {
  public:
    Predictor(const Options &opt, const lssvm_params &params, const T *sv, size_t nsv, size_t nfeat, const T *alpha, T rho) :
        opt_(opt), params_(params), nsv_(nsv), nfeat_(nfeat), rho_(rho) {
        dtype = std::is_same_v<T, float> ? LSSVM_DTYPE_F32 : LSSVM_DTYPE_F64;
        check_params(&params_);
        LSSVM_REQUIRE(sv != nullptr && nsv > 0, "The support vectors must not be empty!");   // csvm.cpp:189
        LSSVM_REQUIRE(nfeat > 0, "The support vectors must contain at least one feature!");  // csvm.cpp:190
        LSSVM_REQUIRE(alpha != nullptr, "The number of support vectors and number of weights must be the same!");
        select_device_checked(0);
        hipStream_t s = nullptr;
        if (params_.kernel_type == LSSVM_KERNEL_LINEAR) {
            // the linear kernel predicts through w = sum_i alpha_i sv_i (csvm.cpp:204-213): computed once, resident zero padded like a row of points
            w_host_.assign(nfeat, T(0));
            calculate_w<T>(sv, nsv, nfeat, alpha, w_host_.data());
            w_.alloc_zero(static_cast<size_t>(padded_features<T>(nfeat)), s);
            LSSVM_HIP_CHECK(hipMemcpyAsync(w_.p, w_host_.data(), nfeat * sizeof(T), hipMemcpyHostToDevice, s));
            LSSVM_HIP_CHECK(hipStreamSynchronize(s));
            return;
        }
        alpha_host_.assign(alpha, alpha + nsv);
        if constexpr (std::is_same_v<T, float>) prepare_resident(sv, s);
        // the one-shot path's input: a host copy where nothing is resident; a resident model keeps its support vectors as they came in HBM and fetches them if a batch ever asks
        if (!resident_) {
            sv_host_.assign(sv, sv + nsv * nfeat);
            S_.data.release();  // (whatever prepare_resident had built before it found the model outside the resident form)
            raw_.release();
            planesS_.buf.release();
            cS_.release();
            mean_.release();
        }
    }

    void predict(const void *points_v, int mem_kind, size_t npoints, void *out_v, lssvm_predict_info *info) override {
        const T *points = static_cast<const T *>(points_v);
        T *out = static_cast<T *>(out_v);
        LSSVM_REQUIRE(points != nullptr && npoints > 0, "The data points to predict must not be empty!");  // csvm.cpp:194
        LSSVM_REQUIRE(out != nullptr, "out must not be NULL");
        lssvm_predict_info local{};
        local.f16_row_rel_error = -1.0;
        bool done = false;
        if (params_.kernel_type == LSSVM_KERNEL_LINEAR) {
            predict_linear(points, mem_kind, npoints, out, local);
            done = true;
        } else if constexpr (std::is_same_v<T, float>) {
            if (resident_) done = predict_resident(points, mem_kind, npoints, out, local);
        }
        if (!done) {
            // what the resident form does not cover (fp64, more than 128 features, exponent scales beyond the norm expansion, a batch whose planes fail the f16 check or
            // that lies further from the support vectors' centre than the form chosen for them allows): the one-shot path, same result
            int w_valid = 0;
            std::vector<T> w_tmp(nfeat_);
            if (sv_host_.empty()) fetch_support_vectors();
            if (mem_kind == LSSVM_MEM_DEVICE) {  // (the one-shot entry point takes host buffers: a batch in HBM makes the round trip here -- the rare path)
                select_device_checked(0);
                std::vector<T> points_host(npoints * nfeat_), out_host(npoints);
                LSSVM_HIP_CHECK(hipMemcpy(points_host.data(), points, points_host.size() * sizeof(T), hipMemcpyDeviceToHost));
                predict_values<T>(opt_, params_, sv_host_.data(), nsv_, nfeat_, alpha_host_.data(), rho_, w_tmp.data(), &w_valid, points_host.data(), npoints, out_host.data(), &local);
                LSSVM_HIP_CHECK(hipMemcpy(out, out_host.data(), npoints * sizeof(T), hipMemcpyHostToDevice));
            } else {
                predict_values<T>(opt_, params_, sv_host_.data(), nsv_, nfeat_, alpha_host_.data(), rho_, w_tmp.data(), &w_valid, points, npoints, out, &local);
            }
            local.resident = 0;
        }
        if (info != nullptr) *info = local;
    }

  private:
    void predict_linear(const T *points, int mem_kind, size_t npoints, T *out, lssvm_predict_info &info) {
        select_device_checked(0);
        hipStream_t s = nullptr;
        const double t0 = now_ms();
        Event ev_a, ev_b;
        ev_a.create(true);
        ev_b.create(true);
        DeviceMatrix<T> P;
        P.upload(points, mem_kind, npoints, nfeat_, 0, s);
        DevBuf<T> o;
        o.alloc_zero(npoints, s);
        LSSVM_HIP_CHECK(hipStreamSynchronize(s));
        const double t_kernel = now_ms();
        LSSVM_HIP_CHECK(hipEventRecord(ev_a.e, s));
        launch_predict_linear<T>(P, w_.p, rho_, o.p, s);
        LSSVM_HIP_CHECK(hipEventRecord(ev_b.e, s));
        LSSVM_HIP_CHECK(hipGetLastError());
        LSSVM_HIP_CHECK(hipMemcpyAsync(out, o.p, npoints * sizeof(T), mem_kind == LSSVM_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
        LSSVM_HIP_CHECK(hipStreamSynchronize(s));
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ev_a.e, ev_b.e) == hipSuccess) info.kernel_ms = ms;
        info.total_ms = now_ms() - t0;
        info.setup_ms = t_kernel - t0;
        info.resident = 1;
    }

    /* fp32, rbf / polynomial, at most 128 features, a split Gram mode: the support vectors' side of the product, once */
    void prepare_resident(const float *sv, hipStream_t s) {
        if (round_up(static_cast<long>(nfeat_), 64) > 128 || opt_.gram_mode == 0 || opt_.tile_kernel == 1) return;
        if (params_.kernel_type == LSSVM_KERNEL_RBF && (opt_.rbf_form == 1 || opt_.rbf_form == 3)) return;  // (the direct kernel / the grid planes asked for: the one-shot path has them)
        S_.upload(sv, LSSVM_MEM_HOST, nsv_, nfeat_, 0, s);
        if (!v2_eligible(opt_, S_.ldx, false)) return;
        const bool rbf = params_.kernel_type == LSSVM_KERNEL_RBF;
        if (rbf) {
            column_means<float>(S_, mean_, s);
            const double sq = max_centred_sqnorm<float>(S_, mean_, s);
            r2_sv_ = 2.0 * static_cast<double>(static_cast<float>(params_.gamma)) * 1.4426950408889634 * sq;
            if (!(r2_sv_ <= RBF_DIRECT_ABOVE) && opt_.rbf_form != 2) return;  // (beyond the norm expansion's range: grid planes or the direct kernel, one-shot)
            scale_ = rbf_prescale<float>(params_, false);
            // S_ is centred and scaled in place below: the support vectors as they came stay beside it (the one-shot path's input, should a batch need it)
            raw_.alloc_zero(S_.data.count, s);
            LSSVM_HIP_CHECK(hipMemcpyAsync(raw_.p, S_.data.p, S_.data.count * sizeof(float), hipMemcpyDeviceToDevice, s));
            hipLaunchKernelGGL(k_center<float>, dim3((S_.dfeat + 255) / 256, S_.rows), dim3(256), 0, s, S_.data.p, S_.ldx, S_.dfeat, S_.rows, mean_.p, scale_);
            LSSVM_HIP_CHECK(hipGetLastError());
            half_neg_norms<float>(S_, cS_, s);
        }
        make_planes(opt_, params_, false, S_, nullptr, planesS_, nullptr, s);
        if (planesS_.mode == 0) return;
        num_jt_ = S_.rows_alloc / TILE;
        a_.alloc_zero(S_.rows_alloc, s);
        LSSVM_HIP_CHECK(hipMemcpyAsync(a_.p, alpha_host_.data(), nsv_ * sizeof(float), hipMemcpyHostToDevice, s));
        // the (alpha_j | c_j) records: folded for rbf while the exponent terms stay small -- decided per batch from ITS exponent scale too, so both forms are kept
        const int ncols = num_jt_ * TILE;
        dc_.alloc_zero(static_cast<size_t>(num_jt_) * 256, s);
        enqueue_pack_records(a_.p, cS_.p, ncols, dc_.p, 0, static_cast<const float *>(nullptr), s);
        if (rbf && opt_.rbf_fold != 0) {
            dc_folded_.alloc_zero(static_cast<size_t>(num_jt_) * 256, s);
            enqueue_pack_records(a_.p, cS_.p, ncols, dc_folded_.p, 1, static_cast<const float *>(nullptr), s);
        }
        LSSVM_HIP_CHECK(hipGetLastError());
        LSSVM_HIP_CHECK(hipStreamSynchronize(s));
        resident_ = true;
    }

    /* the support vectors back on the host, as they came (a resident model keeps no host copy until a batch needs the one-shot path) */
    void fetch_support_vectors() {
        if constexpr (std::is_same_v<T, float>) {
            select_device_checked(0);
            const float *src = raw_.p != nullptr ? raw_.p : S_.data.p;
            LSSVM_REQUIRE(src != nullptr, "the predictor holds no support vectors");
            sv_host_.resize(nsv_ * nfeat_);
            LSSVM_HIP_CHECK(hipMemcpy2D(sv_host_.data(), nfeat_ * sizeof(float), src, static_cast<size_t>(S_.ldx) * sizeof(float), nfeat_ * sizeof(float), nsv_, hipMemcpyDeviceToHost));
        }
    }

    /* a batch of points against the resident support vectors; false = this batch needs the one-shot path */
    bool predict_resident(const float *points, int mem_kind, size_t npoints, float *out, lssvm_predict_info &info) {
        select_device_checked(0);
        hipStream_t s = nullptr;
        const double t0 = now_ms();
        const char *dbg_env = std::getenv("LSSVM_MI355_DEBUG");
        const bool dbg = dbg_env != nullptr && dbg_env[0] == '1';
        double t_last = t0;
        auto lap = [&](const char *what) {  // LSSVM_MI355_DEBUG=1: where a call's time goes (the stream is drained at every lap, so the laps add up)
            if (!dbg) return;
            (void) hipStreamSynchronize(s);
            const double t = now_ms();
            std::fprintf(stderr, "[plssvm_amd] predictor: %-28s %8.3f ms\n", what, t - t_last);
            t_last = t;
        };
        const bool rbf = params_.kernel_type == LSSVM_KERNEL_RBF;
        DeviceMatrix<float> P;
        P.upload(points, mem_kind, npoints, nfeat_, static_cast<size_t>(round_up(static_cast<long>(npoints), 2 * TILE)), s);
        lap(mem_kind == LSSVM_MEM_DEVICE ? "points copied in HBM" : "points uploaded");
        DevBuf<float> cP;
        double r2 = r2_sv_;
        if (rbf) {
            const double sq = max_centred_sqnorm<float>(P, mean_, s);  // (against the SUPPORT VECTORS' means: the centre the resident side was prepared with)
            r2 = std::max(r2, 2.0 * static_cast<double>(static_cast<float>(params_.gamma)) * 1.4426950408889634 * sq);
            if (!(r2 <= RBF_DIRECT_ABOVE) && opt_.rbf_form != 2) return false;  // this batch reaches beyond the norm expansion's range
            hipLaunchKernelGGL(k_center<float>, dim3((P.dfeat + 255) / 256, P.rows), dim3(256), 0, s, P.data.p, P.ldx, P.dfeat, P.rows, mean_.p, scale_);
            LSSVM_HIP_CHECK(hipGetLastError());
            half_neg_norms<float>(P, cP, s);
        }
        lap("centred, norms");
        // the batch's planes: the kind and the scale of the support vectors' planes; two f16 planes must represent THIS batch too
        PlaneSet planesP;
        planesP.ldx16 = planesS_.ldx16;
        planesP.nplanes = planesS_.nplanes;
        planesP.mode = planesS_.mode;
        planesP.shift = planesS_.shift;
        planesP.buf.alloc_zero(static_cast<size_t>(planesP.nplanes) * P.rows_alloc * planesP.ldx16, s);
        if (planesS_.mode == 2) {
            DevBuf<unsigned> stats;
            stats.alloc_zero(4, s);
            split_f16_planes(P.data.p, P.ldx, P.dfeat, static_cast<size_t>(P.rows_alloc), planesP.ldx16, std::ldexp(1.0f, planesS_.shift), rbf ? F16_RBF_SHIFT : 0, planesP.buf.p,
                             static_cast<size_t>(P.rows_alloc) * planesP.ldx16, stats.p, s);
            unsigned host[4] = { 0, 0, 0, 0 };
            LSSVM_HIP_CHECK(hipMemcpyAsync(host, stats.p, sizeof(host), hipMemcpyDeviceToHost, s));
            LSSVM_HIP_CHECK(hipStreamSynchronize(s));
            float rel2 = 0.0f, rest2 = 0.0f, x2 = 0.0f;
            std::memcpy(&rel2, &host[0], sizeof(float));
            std::memcpy(&rest2, &host[1], sizeof(float));
            std::memcpy(&x2, &host[2], sizeof(float));
            bool ok = rel2 <= F16_REL2_MAX;
            if (!ok && rbf) ok = std::isfinite(rel2) && 2.0 * std::sqrt(static_cast<double>(rest2) * static_cast<double>(x2)) <= static_cast<double>(F16_ABS_MAX);
            info.f16_row_rel_error = std::max(planesS_.f16_row_rel_error, std::sqrt(static_cast<double>(rel2)));
            if (!ok && opt_.gram_mode != 2) return false;  // the support vectors' planes are f16, this batch needs bf16: one-shot (which splits both sides alike)
        } else {
            split_bf16_planes(P.data.p, P.ldx, P.dfeat, static_cast<size_t>(P.rows_alloc), planesP.ldx16, planesP.buf.p, static_cast<size_t>(P.rows_alloc) * planesP.ldx16, s);
        }
        lap("operand planes");
        const int num_ib = P.rows_alloc / TILE;
        const bool folded = rbf && dc_folded_.p != nullptr && r2 <= FOLD_MAX_R2;
        const bool poly_generic = params_.kernel_type == LSSVM_KERNEL_POLYNOMIAL && params_.degree != 2 && params_.degree != 3;
        const bool rbf_ok = !rbf || (folded && r2 <= 2.0 * PAIR_FOLD_MAX_C);
        const bool rect = !poly_generic && rbf_ok && opt_.mfma_shape >= 3 && num_ib >= PAIR_MIN_TILES;
        const long rect_tiles = std::min<long>(64, std::max<long>(4, (static_cast<long>(num_ib / 2) * num_jt_ + 1024) / 2048));
        const int jc_tiles = opt_.j_chunk_tiles > 0 ? static_cast<int>(opt_.j_chunk_tiles)
                                                   : (rect ? static_cast<int>(rect_tiles) : static_cast<int>(std::min<long>(16, std::max<long>(2, (static_cast<long>(num_ib) * num_jt_ + 2048) / 4096))));
        const int num_jc = (num_jt_ + jc_tiles - 1) / jc_tiles;
        DevBuf<float> partial, Kv, o;
        partial.alloc_zero(static_cast<size_t>(num_jc) * P.rows_alloc, s);
        Kv.alloc_zero(P.rows_alloc, s);
        o.alloc_zero(npoints, s);
        TileArgs<float> ta{};
        ta.Xr = P.data.p;
        ta.Xc = S_.data.p;
        ta.cr = cP.p;
        ta.cc = cS_.p;
        ta.dvec = a_.p;
        ta.dc = folded ? dc_folded_.p : dc_.p;
        ta.dc_folded = folded ? 1 : 0;
        ta.partial = partial.p;
        ta.part_stride = P.rows_alloc;
        ta.ldx = S_.ldx;
        ta.kchunks = S_.ldx / F32_KC;
        ta.num_ib = num_ib;
        ta.num_jt = num_jt_;
        ta.jc_tiles = jc_tiles;
        ta.ncols_valid = S_.rows;
        set_kernel_scalars(ta, params_, false);
        set_plane_args(ta, params_, planesS_, planesP, static_cast<size_t>(S_.rows_alloc), static_cast<size_t>(P.rows_alloc));
        set_launch_options(ta, opt_);
        RectSetup rect_setup;
        if (rect) setup_rect_launch(ta, rect_setup, planesP, P.rows_alloc, num_ib, num_jc, s);
        Event ev_a, ev_b;
        ev_a.create(true);
        ev_b.create(true);
        LSSVM_HIP_CHECK(hipStreamSynchronize(s));
        lap(rect ? "slabs, 256-row launch set up" : "slabs");
        const double t_kernel = now_ms();
        LSSVM_HIP_CHECK(hipEventRecord(ev_a.e, s));
        launch_tile_kernel<float>(ta, params_.kernel_type, false, num_jc, s);
        LSSVM_HIP_CHECK(hipEventRecord(ev_b.e, s));
        hipLaunchKernelGGL(k_reduce_partials<float>, dim3((P.rows_alloc + 255) / 256), dim3(256), 0, s, partial.p, ta.part_stride, num_jc, 0, P.rows_alloc, Kv.p);
        hipLaunchKernelGGL(k_sub_rho<float>, dim3((P.rows + 255) / 256), dim3(256), 0, s, Kv.p, P.rows, rho_, o.p);
        LSSVM_HIP_CHECK(hipGetLastError());
        lap("product, row sums");
        LSSVM_HIP_CHECK(hipMemcpyAsync(out, o.p, npoints * sizeof(float), mem_kind == LSSVM_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
        LSSVM_HIP_CHECK(hipStreamSynchronize(s));
        lap("values downloaded");
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ev_a.e, ev_b.e) == hipSuccess) info.kernel_ms = ms;
        info.total_ms = now_ms() - t0;
        info.setup_ms = t_kernel - t0;
        info.gram_mode = planesS_.mode;
        info.rbf_direct = 0;
        info.rbf_exponent_scale = r2;
        if (info.f16_row_rel_error < 0.0) info.f16_row_rel_error = planesS_.f16_row_rel_error;
        info.resident = 1;
        return true;
    }

    Options opt_;
    lssvm_params params_;
    size_t nsv_, nfeat_;
    T rho_;
    std::vector<T> sv_host_, alpha_host_, w_host_;  // the one-shot path's inputs
    DevBuf<T> w_;
    // fp32 resident form
    bool resident_ = false;
    DeviceMatrix<float> S_;
    DevBuf<float> mean_, cS_, a_, dc_, dc_folded_, raw_;
    PlaneSet planesS_;
    double r2_sv_ = 0.0;
    float scale_ = 1.0f;
    int num_jt_ = 0;
};

std::unique_ptr<PredictorBase> make_predictor(const Options &opt, const lssvm_params &params, int dtype, const void *sv, size_t nsv, size_t nfeat, const void *alpha, double rho) {
    if (dtype == LSSVM_DTYPE_F32) return std::make_unique<Predictor<float>>(opt, params, static_cast<const float *>(sv), nsv, nfeat, static_cast<const float *>(alpha), static_cast<float>(rho));
    return std::make_unique<Predictor<double>>(opt, params, static_cast<const double *>(sv), nsv, nfeat, static_cast<const double *>(alpha), rho);
}

}  // namespace lssvm
