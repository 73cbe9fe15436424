/*
 * lssvm_solver.hip -- Solver<T>: the CG driver over the shards of this process (see lssvm_problem.hip.hpp; the recipe is the reference's
 * src/plssvm/backends/OpenMP/csvm.cpp:71-183, the orchestration replaces include/plssvm/backends/gpu_csvm.hpp:477-654).  The shards themselves (data, planes,
 * work-item geometry, the tile-kernel launches) are lssvm_problem.hip, the exchange of the partial vectors between them lssvm_exchange.hip.  Compiled for gfx950 only.
 */
#include "lssvm_problem.hip.hpp"

#define LSSVM_KERNELS_CG
#include "lssvm_kernels.hip.hpp"

#include <algorithm>
#include <cstdlib>

namespace lssvm {

/* ------------------------------------------------------------------ Solver: CG over the shards of this process ------------------------------------------------------------------ */
template <typename T>
Solver<T>::Solver(const Options &opt, const lssvm_params &params, const void *X, int mem_kind, size_t num_points, size_t num_features, const std::vector<int> &devices, const lssvm_shard *shard) :
    opt_(opt) {
    dtype = std::is_same_v<T, float> ? LSSVM_DTYPE_F32 : LSSVM_DTYPE_F64;
    const double t0 = now_ms();
    LSSVM_REQUIRE(!devices.empty() && devices.size() <= static_cast<size_t>(MAX_LOCAL_DEVICES), "invalid number of devices");
    int rank0 = 0;
    world_ = static_cast<int>(devices.size());
    if (shard != nullptr && (shard->world != 1 || shard->rank != 0)) {
        // one process per GPU: this process holds rank `rank` of `world`
        LSSVM_REQUIRE(devices.size() == 1, "a shard descriptor (one process per GPU) and a device list (one process, several GPUs) exclude each other");
        LSSVM_REQUIRE(shard->world >= 1 && shard->rank >= 0 && shard->rank < shard->world, "invalid shard descriptor");
        rank0 = shard->rank;
        world_ = shard->world;
        if (opt_.skip_collective == 0) {
            // exchange 1: RCCL (lssvm_mi355_comm_init first); 2: HIP IPC + peer kernels (lssvm_mi355_problem_ipc_export / _connect after
            // the problem exists); 0: RCCL when this process has a communicator, else IPC
            const bool have_comm = comm().comm != nullptr;
            if (opt_.exchange == 1 || (opt_.exchange == 0 && have_comm)) {
                LSSVM_REQUIRE(have_comm && comm().world == world_ && comm().rank == rank0,
                              "row-block sharding over RCCL requested but lssvm_mi355_comm_init was not called with the same rank/world");
                LSSVM_REQUIRE(comm().device == devices[0], "the communicator was created for another device");
                exchange_ = Exchange::process_rccl;
            } else {
                LSSVM_REQUIRE(world_ <= MAX_LOCAL_DEVICES, "the peer exchange handles at most " + std::to_string(MAX_LOCAL_DEVICES) + " ranks");
                exchange_ = Exchange::process_peer;
            }
        }
    } else if (devices.size() == 1) {
        if (opt_.force_collective != 0 && comm().comm != nullptr && comm().world == 1) exchange_ = Exchange::process_rccl;
    } else {
        std::vector<int> sorted(devices);
        std::sort(sorted.begin(), sorted.end());
        const bool distinct = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
        // (exchange = 1 with a repeated ordinal: RCCL's own ncclCommInitAll refuses the list -- "duplicate GPU detected" -- and its message comes back)
        exchange_ = (opt_.exchange == 1 || (opt_.exchange == 0 && distinct)) ? Exchange::local_rccl : Exchange::peer;
        if (opt_.skip_collective != 0) exchange_ = Exchange::none;
    }
    const auto build_shards = [&] {
        shards_.clear();
        for (size_t r = 0; r < devices.size(); ++r) {
            shards_.push_back(std::make_unique<Problem<T>>(opt_, params, X, mem_kind, num_points, num_features, devices[r], rank0 + static_cast<int>(r), world_));
        }
    };
    try {
        build_shards();
    } catch (const GridPlanesUnfit &) {
        // rbf_form 0 chose the grid planes from the exponent scale, and they do not represent this data: the formula-exact kernel, as before the grid planes existed.
        // Every shard (and every rank: the data is replicated) sees the same planes and comes here alike.  An explicit rbf_form = 3 keeps the error.
        if (opt_.rbf_form != 0) throw;
        opt_.rbf_form = 1;
        build_shards();
    }
    if (exchange_ == Exchange::local_rccl) local_comms_ = local_comms_for(devices);
    if (exchange_ == Exchange::process_peer) {
        Problem<T> &p = *shards_[0];
        p.activate();
        ipc_ = std::make_unique<IpcPeers>(rank0, world_);
        p.Ksum_.alloc_zero(p.nvec_, p.stream());
        p.Kres_ = p.Ksum_.p;
    }
    if (exchange_ == Exchange::peer) {
        for (auto &p : shards_) {
            p->activate();
            for (auto &o : shards_) {
                if (o->device_ == p->device_) continue;
                int can = 0;
                LSSVM_HIP_CHECK(hipDeviceCanAccessPeer(&can, p->device_, o->device_));
                if (can == 0) throw Error(LSSVM_ERR_COMM, "device " + std::to_string(p->device_) + " cannot map the memory of device " + std::to_string(o->device_) + " (use option exchange = 1)");
                const hipError_t e = hipDeviceEnablePeerAccess(o->device_, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) LSSVM_HIP_CHECK(e);
                (void) hipGetLastError();
            }
            p->Ksum_.alloc_zero(p->nvec_, p->stream());
            p->Kres_ = p->Ksum_.p;
        }
    }
    shards_[0]->activate();
    ev_delta_.create(false);
    sync_all();
    setup_ms_ = now_ms() - t0;
}

template <typename T>
Solver<T>::~Solver() {
    for (auto &p : shards_) {
        (void) hipSetDevice(p->device_);
        (void) hipStreamSynchronize(p->stream());
    }
    if (ipc_ && ipc_->connected && xseq_ > 0 && ipc_->own->abort.load() == 0) {
        // the peers read this rank's partial vector through their IPC mapping: it may be freed only after their last read
        try {
            ipc_->wait_all(1, xseq_, 30.0);
        } catch (...) {
        }
    }
}

template <typename T>
void Solver<T>::sync_all() {
    for (auto &p : shards_) {
        p->activate();
        LSSVM_HIP_CHECK(hipStreamSynchronize(p->stream()));
        p->drain_events();
    }
}

template <typename T>
void Solver<T>::apply_K(Vec which) {
    const bool skip = world_ > 1 && opt_.skip_collective != 0;  // testing aid: this rank's share only
    const bool collective = exchange_ != Exchange::none;
    if (exchange_ == Exchange::process_peer && xseq_ > 0 && !skip) {
        // this rank's partial vector is about to be overwritten: every peer must have read the previous one
        ipc_->wait_all(1, xseq_, static_cast<double>(std::max<int64_t>(opt_.ipc_timeout_s, 1)));
    }
    for (auto &p : shards_) {
        p->activate();
        p->enqueue_apply_K_local(vec_of(*p, which), (p->sym_ && collective) || skip);
    }
    if (collective) exchange();
}

template <typename T>
void Solver<T>::get_q(void *q_out, double *QA_cost_out) {
    Problem<T> &p = *shards_[0];
    p.activate();
    if (q_out != nullptr) {
        LSSVM_HIP_CHECK(hipMemcpyAsync(q_out, p.q_.p, static_cast<size_t>(p.n_) * sizeof(T), hipMemcpyDeviceToHost, p.stream()));
        LSSVM_HIP_CHECK(hipStreamSynchronize(p.stream()));
    }
    if (QA_cost_out != nullptr) *QA_cost_out = p.QA_cost_;
}

template <typename T>
void Solver<T>::matvec(const void *d, void *ret_inout, double add) {
    LSSVM_REQUIRE(d != nullptr && ret_inout != nullptr, "The d / ret arrays may not be empty!");              // csvm.cpp:284-286
    LSSVM_REQUIRE(add == 1.0 || add == -1.0, "add must either be -1.0 or 1.0, but is " + std::to_string(add) + "!");  // svm_kernel.cpp:28
    const size_t bytes = static_cast<size_t>(shards_[0]->n_) * sizeof(T);
    for (auto &p : shards_) {
        p->activate();
        // tmp_ <- d (zero padded), Ad_ <- ret
        LSSVM_HIP_CHECK(hipMemcpyAsync(p->tmp_.p, d, bytes, hipMemcpyHostToDevice, p->stream()));
        LSSVM_HIP_CHECK(hipMemcpyAsync(p->Ad_.p, ret_inout, bytes, hipMemcpyHostToDevice, p->stream()));
        p->enqueue_sum_and_qdot(p->tmp_.p, SC_S, SC_QD);
    }
    apply_K(Vec::tmp);
    for (auto &p : shards_) {
        p->activate();
        hipLaunchKernelGGL(k_apply_ret<T>, dim3((p->n_ + 255) / 256), dim3(256), 0, p->stream(), p->Kres_, p->tmp_.p, p->q_.p, p->sc_.p, p->n_, p->inv_cost_, p->QA_cost_, add, p->Ad_.p);
        LSSVM_HIP_CHECK(hipGetLastError());
    }
    Problem<T> &p0 = *shards_[0];
    p0.activate();
    LSSVM_HIP_CHECK(hipMemcpyAsync(ret_inout, p0.Ad_.p, bytes, hipMemcpyDeviceToHost, p0.stream()));
    sync_all();
}

/* k_update_d packs the next matvec's records (Problem::pack_for_d) -- except over the IPC exchange: there a rank may clear its partial vector only once every peer
 * has read the previous one, which the host establishes in apply_K, after the direction update. */
template <typename T>
PackDc<T> Solver<T>::pack_with_direction(Problem<T> &p) {
    if (exchange_ == Exchange::process_peer) return PackDc<T>{};
    const bool skip = world_ > 1 && opt_.skip_collective != 0;
    return p.pack_for_d((p.sym_ && exchange_ != Exchange::none) || skip);
}

/* Shares by measured pace (lssvm_mi355_problem_rebalance).  weights == NULL: every shard's tile-kernel time per implicit matvec so far (HIP events) against the area
 * of its share -- one process driving all shards knows them all; one process per GPU gathers them over the library's RCCL communicator, so that every rank
 * computes the same weights (no such channel over HIP IPC: explicit weights only).  Nothing changes where the times lie within 2 % of each other.  Call between
 * cg_step calls. */
template <typename T>
int Solver<T>::rebalance(const double *weights, int count) {
    if (world_ <= 1 || !shards_[0]->sym_) return 0;
    sync_all();
    std::vector<double> w;
    if (weights != nullptr) {
        LSSVM_REQUIRE(count == world_, "one weight per rank");
        for (int k = 0; k < count; ++k) {
            LSSVM_REQUIRE(std::isfinite(weights[k]) && weights[k] > 0.0, "shard weights must be positive and finite");
            w.push_back(weights[k]);
        }
    } else {
        // pace of rank r = area of its share / kernel time per matvec
        std::vector<double> ms(static_cast<size_t>(world_), 0.0);
        for (auto &p : shards_) {
            p->activate();
            p->drain_events();
            // (since the last reshard, and never the cold first matvec of cg_begin: enqueue_apply_K_local does not time launch 0)
            const uint64_t timed = p->matvec_timed_ - p->pace_timed0_;
            ms[static_cast<size_t>(p->rank_)] = timed > 0 ? (p->matvec_ms_ - p->pace_ms0_) / static_cast<double>(timed) : 0.0;
        }
        if (static_cast<int>(shards_.size()) != world_) {  // one process per GPU: every rank contributes its own time
            LSSVM_REQUIRE(exchange_ == Exchange::process_rccl, "measured shares need the RCCL communicator between the processes (give explicit weights over HIP IPC)");
            Problem<T> &p = *shards_[0];
            p.activate();
            DevBuf<double> all;
            all.alloc_zero(static_cast<size_t>(world_), p.stream());
            LSSVM_HIP_CHECK(hipMemcpyAsync(all.p + p.rank_, &ms[static_cast<size_t>(p.rank_)], sizeof(double), hipMemcpyHostToDevice, p.stream()));
            Comm &c = comm();
            nccl_check(c.pAllGather(all.p + p.rank_, all.p, 1, ncclDouble, c.comm, p.stream()), "ncclAllGather");
            LSSVM_HIP_CHECK(hipMemcpyAsync(ms.data(), all.p, sizeof(double) * static_cast<size_t>(world_), hipMemcpyDeviceToHost, p.stream()));
            LSSVM_HIP_CHECK(hipStreamSynchronize(p.stream()));
        }
        double lo = ms[0], hi = ms[0];
        for (const double v : ms) {
            lo = std::min(lo, v);
            hi = std::max(hi, v);
        }
        if (!(lo > 0.0) || hi <= 1.02 * lo) return 0;  // (a rank without a timed matvec, or nothing to gain)
        const int tiles = shards_[0]->num_tiles_;
        double sum = 0.0;
        for (int r = 0; r < world_; ++r) {
            const double b0 = sym_block_boundary(tiles, r, world_, &shards_[0]->opt_.shard_weights), b1 = sym_block_boundary(tiles, r + 1, world_, &shards_[0]->opt_.shard_weights);
            const double area = 0.5 * (b1 * (b1 + 1.0) - b0 * (b0 + 1.0));
            w.push_back(std::max(area, 1.0) / ms[static_cast<size_t>(r)]);
            sum += w.back();
        }
        for (double &v : w) v *= static_cast<double>(world_) / sum;
    }
    for (auto &p : shards_) p->reshard(w);
    opt_.shard_weights = w;
    return 1;
}

template <typename T>
void Solver<T>::cg_begin(const void *y, double eps) {
    LSSVM_REQUIRE(y != nullptr, "The right hand side vector must not be empty!");
    LSSVM_REQUIRE(static_cast<T>(eps) > T(0), "The stopping criterion in the CG algorithm must be greater than 0.0, but is " + std::to_string(eps) + "!");  // csvm.cpp:77
    const double t0 = now_ms();
    eps_ = eps;
    info_shard_ = -1;
    iter_ = 0;
    converged_ = false;
    cg_wall_ms_ = 0.0;
    const size_t N = shards_[0]->N_;
    y_last_ = static_cast<double>(static_cast<const T *>(y)[N - 1]);
    for (auto &p : shards_) {
        p->activate();
        p->matvec_ms_ = p->pace_ms0_ = 0.0;
        p->matvec_launches_ = 0;
        p->matvec_timed_ = p->pace_timed0_ = 0;
        hipStream_t st = p->stream();
        LSSVM_HIP_CHECK(hipMemcpyAsync(p->ylast_.p, y, N * sizeof(T), hipMemcpyHostToDevice, st));
        const dim3 gn((p->n_ + 255) / 256), bn(256);
        hipLaunchKernelGGL(k_make_b<T>, gn, bn, 0, st, p->ylast_.p, p->n_, p->b_.p);        // csvm.cpp:89-91
        hipLaunchKernelGGL(k_fill<T>, gn, bn, 0, st, p->x_.p, p->n_, T(1));                // csvm.cpp:95
        p->enqueue_sum_and_qdot(p->x_.p, SC_SUMX, SC_QX);
    }
    // r = b - A x   (csvm.cpp:101-104)
    apply_K(Vec::x);
    for (auto &p : shards_) {
        p->activate();
        hipStream_t st = p->stream();
        hipLaunchKernelGGL(k_residual<T>, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, st, p->Kres_, p->x_.p, p->q_.p, p->b_.p, p->sc_.p, p->n_, p->inv_cost_, p->QA_cost_, p->r_.p, p->part(PART_RR));
        hipLaunchKernelGGL(k_finish_delta, dim3(1), dim3(RED_THREADS), 0, st, p->part(PART_RR), p->sc_.p, p->sc_.p + SC_COUNT - 1, 1);  // csvm.cpp:107-108
        // d = r   (csvm.cpp:111), and -- as partial sums that k_Ad_and_dAd finishes for itself -- the sums the next matvec's rank-1 terms need
        hipLaunchKernelGGL(k_update_d<T>, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, st, p->d_.p, p->r_.p, p->q_.p, p->sc_.p, p->n_, 1, p->part(PART_D), pack_with_direction(*p));
        LSSVM_HIP_CHECK(hipGetLastError());
    }
    Problem<T> &p0 = *shards_[0];
    p0.activate();
    LSSVM_HIP_CHECK(hipMemcpyAsync(p0.host_sc_.p, p0.sc_.p, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, p0.stream()));
    sync_all();
    delta0_ = static_cast<double>(static_cast<T>(p0.host_sc_.p[SC_DELTA0]));
    delta_ = delta0_;
    delta_before_ = 0.0;
    held_back_ = 0;
    begun_ = true;
    cg_wall_ms_ += now_ms() - t0;
}

template <typename T>
void Solver<T>::cg_step(uint64_t iterations, int *done_out) {
    LSSVM_REQUIRE(begun_, "cg_step called before cg_begin");
    const double t0 = now_ms();
    // target residuum in the real type, exactly as the reference evaluates "eps * eps * delta0" (csvm.cpp:155)
    const T target = static_cast<T>(eps_) * static_cast<T>(eps_) * static_cast<T>(delta0_);
    Problem<T> &p0 = *shards_[0];
    const auto enqueue_direction_update = [&] {  // d = beta d + r   (csvm.cpp:161-163), and the sums the next matvec needs
        for (auto &p : shards_) {
            p->activate();
            hipStream_t st = p->stream();
            hipLaunchKernelGGL(k_update_d<T>, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, st, p->d_.p, p->r_.p, p->q_.p, p->sc_.p, p->n_, 0, p->part(PART_D), pack_with_direction(*p));
            LSSVM_HIP_CHECK(hipGetLastError());
        }
    };
    bool matvec_enqueued = false;  // A d of the coming iteration is in the queue already (enqueued ahead of the previous stop test)
    for (uint64_t k = 0; k < iterations && !converged_; ++k) {
        // Ad = A d   (csvm.cpp:131-132)
        if (!matvec_enqueued) apply_K(Vec::d);
        matvec_enqueued = false;
        const bool refresh = iter_ % 50 == 49;
        for (auto &p : shards_) {
            p->activate();
            hipStream_t st = p->stream();
            // (every kernel of the chain reduces its predecessor's partial sums for itself -- finish2_in_block -- so no single-block kernel stands between them)
            hipLaunchKernelGGL(k_Ad_and_dAd<T>, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, st, p->Kres_, p->d_.p, p->q_.p, p->part(PART_D), p->sc_.p, p->n_, p->inv_cost_, p->QA_cost_, p->Ad_.p, p->part(PART_DAD));
            // alpha = delta / d.Ad (csvm.cpp:135) ; x += alpha d ; r -= alpha Ad   (csvm.cpp:138, :148) -- or, every 50th iteration, x only and r = b - A x below (csvm.cpp:140-145)
            hipLaunchKernelGGL(k_update_x_r<T>, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, st, p->x_.p, p->r_.p, p->d_.p, p->Ad_.p, p->part(PART_DAD), p->sc_.p, p->n_, refresh ? 0 : 1, p->part(PART_RR));
            if (refresh) p->enqueue_sum_and_qdot(p->x_.p, SC_SUMX, SC_QX);
            LSSVM_HIP_CHECK(hipGetLastError());
        }
        if (refresh) {
            apply_K(Vec::x);
            for (auto &p : shards_) {
                p->activate();
                hipLaunchKernelGGL(k_residual<T>, dim3(RED_BLOCKS), dim3(RED_THREADS), 0, p->stream(), p->Kres_, p->x_.p, p->q_.p, p->b_.p, p->sc_.p, p->n_, p->inv_cost_, p->QA_cost_, p->r_.p, p->part(PART_RR));
            }
        }
        for (auto &p : shards_) {
            p->activate();
            // the stop test needs delta on the host: shard 0's kernel stores its 8 bytes straight into mapped host memory (all shards hold the same
            // bits); the others publish into a spare device word
            hipLaunchKernelGGL(k_finish_delta, dim3(1), dim3(RED_THREADS), 0, p->stream(), p->part(PART_RR), p->sc_.p, p.get() == &p0 ? p0.host_delta_.dev : p->sc_.p + SC_COUNT - 1, 0);  // csvm.cpp:152-153
            LSSVM_HIP_CHECK(hipGetLastError());
        }
        p0.activate();
        LSSVM_HIP_CHECK(hipEventRecord(ev_delta_.e, p0.stream()));
        // Short matvecs (option enqueue_ahead_below_us): waiting for delta here would leave the device idle while the host wakes up and
        // launches the next kernels (about 4 % of a 50 000-point iteration).  So the direction update and the NEXT implicit matvec go
        // into the queue BEFORE the stop test is read: they touch d, K*d and the partial slabs only -- never x or r -- so a solve that
        // turns out to have converged is exactly where the reference stops (csvm.cpp:155-158), at the price of one discarded matvec.
        // The decision must be the same on every rank of a sharded solve (a rank that went ahead has one more collective in its queue than
        // one that did not: a hang at convergence, or every later collective paired one position off), so it is derived from the problem
        // alone -- size, feature count, real type, kernel path, world -- never from this rank's own event timings: an implicit matvec is priced at
        // the full-square rate of its path (one flat 500 TFLOP/s let the fp64 and generic kernels go ahead for matvecs ten times longer
        // than the option says, each converged solve then paying one discarded matvec of that length).  Every input is the same on all ranks.
        const double n_d = static_cast<double>(p0.n_);
        const double matvec_us = 2.0 * n_d * n_d * static_cast<double>(p0.X_.dfeat) / static_cast<double>(world_) / p0.nominal_full_square_rate() * 1e6;
        // Round 6: ... unless this iteration is forecast to be the last.  The discarded matvec is a third of a solve that converges after two iterations (the
        // reference's default epsilon on BASELINE configs[1]: 3.2 ms of CG where 2.4 are needed).  With the residuum falling by rho = delta / delta_before per
        // iteration, the stop test is about to succeed once delta rho^2 <= target -- this iteration or the next: for those (at most four in a row, should the
        // residuum stall there) the host waits for delta before it enqueues anything, as it does for long matvecs.  delta, its predecessor and the target are the
        // same bits on every rank, so every rank decides alike (see above); matvecs shorter than 100 us are not worth a host round trip and always go ahead.
        bool last_iterations = false;
        if (matvec_us >= 100.0 && delta_before_ > 0.0 && delta_ > 0.0 && held_back_ < 4) {
            const double rho = std::min(delta_ / delta_before_, 1.0);
            last_iterations = delta_ * rho * rho <= static_cast<double>(target);
        }
        held_back_ = last_iterations ? held_back_ + 1 : 0;
        const bool ahead = opt_.enqueue_ahead_below_us > 0 && matvec_us < static_cast<double>(opt_.enqueue_ahead_below_us) && k + 1 < iterations
                           && exchange_ != Exchange::process_peer && !last_iterations;
        if (ahead) {
            enqueue_direction_update();
            apply_K(Vec::d);
            p0.activate();
        }
        LSSVM_HIP_CHECK(hipEventSynchronize(ev_delta_.e));
        for (auto &p : shards_) p->drain_events();
        ++iter_;
        delta_before_ = delta_;
        delta_ = static_cast<double>(static_cast<T>(*static_cast<volatile double *>(p0.host_delta_.p)));
        if (static_cast<T>(delta_) <= target) {  // csvm.cpp:155-158: tested BEFORE the direction update
            converged_ = true;
            if (ahead) sync_all();  // let the discarded work drain
            break;
        }
        if (ahead) {
            matvec_enqueued = true;
        } else {
            enqueue_direction_update();
        }
    }
    cg_wall_ms_ += now_ms() - t0;
    if (done_out != nullptr) *done_out = converged_ ? 1 : 0;
}

template <typename T>
void Solver<T>::cg_finish(void *alpha_out, double *rho_out, lssvm_cg_info *info) {
    LSSVM_REQUIRE(begun_, "cg_finish called before cg_begin");
    LSSVM_REQUIRE(alpha_out != nullptr && rho_out != nullptr, "alpha_out / rho_out must not be NULL");
    const double t0 = now_ms();
    // bias = y_last + QA_cost * sum(x) - q^T x ; alpha_N = -sum(x) ; rho = -bias   (csvm.cpp:179-182)
    const bool check = shards_.size() > 1 && exchange_ != Exchange::none;
    for (auto &p : shards_) {
        if (p.get() != shards_[0].get() && !check) continue;
        p->activate();
        p->enqueue_sum_and_qdot(p->x_.p, SC_SUMX, SC_QX);
        LSSVM_HIP_CHECK(hipMemcpyAsync(p->host_sc_.p, p->sc_.p, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, p->stream()));
    }
    Problem<T> &p0 = *shards_[0];
    p0.activate();
    LSSVM_HIP_CHECK(hipMemcpyAsync(alpha_out, p0.x_.p, static_cast<size_t>(p0.n_) * sizeof(T), hipMemcpyDeviceToHost, p0.stream()));
    sync_all();
    if (check) {
        // every shard ran the same O(n) kernels on the same exchanged vectors: their scalars must agree to the bit
        for (auto &p : shards_) {
            for (int slot : { static_cast<int>(SC_DELTA), static_cast<int>(SC_SUMX), static_cast<int>(SC_QX) }) {
                if (std::memcmp(&p->host_sc_.p[slot], &p0.host_sc_.p[slot], sizeof(double)) != 0) {
                    throw Error(LSSVM_ERR_INTERNAL, "the shards of the solve diverged: device " + std::to_string(p->device_) + " holds another CG scalar than device "
                                                        + std::to_string(p0.device_));
                }
            }
        }
    }
    const T sum_x = static_cast<T>(p0.host_sc_.p[SC_SUMX]);
    const T bias = static_cast<T>(y_last_ + p0.QA_cost_ * p0.host_sc_.p[SC_SUMX] - p0.host_sc_.p[SC_QX]);
    static_cast<T *>(alpha_out)[p0.n_] = -sum_x;
    *rho_out = static_cast<double>(-bias);
    cg_wall_ms_ += now_ms() - t0;
    if (info != nullptr) fill_info(info);
}

template <typename T>
void Solver<T>::synchronize() {
    sync_all();
}

template <typename T>
void Solver<T>::fill_info(lssvm_cg_info *info) {
    std::memset(info, 0, sizeof(*info));
    const Problem<T> &p0 = *shards_[0];
    info->iterations = iter_;
    info->max_iterations = 0;
    info->residuum = delta_;
    info->initial_residuum = delta0_;
    info->target_residuum = static_cast<double>(static_cast<T>(eps_) * static_cast<T>(eps_) * static_cast<T>(delta0_));
    info->epsilon = eps_;
    info->avg_iteration_ms = iter_ > 0 ? cg_wall_ms_ / static_cast<double>(iter_) : 0.0;
    info->total_ms = cg_wall_ms_;
    info->setup_ms = setup_ms_;
    // the tile kernel of the slowest shard sets the pace of a sharded matvec.  WHICH shard is reported is decided once per cg_begin (the first call that finds timed
    // matvecs) and kept, so that the differences of (matvec_kernel_ms_total, matvec_timed) between two calls always belong to one shard
    if (info_shard_ < 0) {
        double slowest = 0.0;
        for (size_t r = 0; r < shards_.size(); ++r) {
            const double avg = shards_[r]->matvec_timed_ > 0 ? shards_[r]->matvec_ms_ / static_cast<double>(shards_[r]->matvec_timed_) : 0.0;
            if (avg > slowest) {
                slowest = avg;
                info_shard_ = static_cast<int>(r);
            }
        }
    }
    if (info_shard_ >= 0) {
        const Problem<T> &p = *shards_[static_cast<size_t>(info_shard_)];
        info->matvec_timed = p.matvec_timed_;
        info->matvec_kernel_ms_total = p.matvec_ms_;
        info->matvec_kernel_ms = p.matvec_timed_ > 0 ? p.matvec_ms_ / static_cast<double>(p.matvec_timed_) : 0.0;
    }
    info->f16_row_rel_error = p0.f16_row_rel_error_;
    info->matvec_launches = p0.matvec_launches_;
    info->devices_used = world_;
    info->converged = converged_ ? 1 : 0;
    info->symmetric = p0.sym_ ? 1 : 0;
    info->gram_mode = (p0.planes_.mode != 0 && p0.dc_.p != nullptr) ? (p0.rbf_grid_ ? 3 : p0.planes_.mode) : 0;  // (a split kernel is dispatched only where both exist; 3: rbf on grid planes)
    info->local_devices = static_cast<int32_t>(shards_.size());
    info->rbf_direct = p0.rbf_direct_ ? 1 : 0;
    info->rbf_exponent_scale = p0.rbf_r2_;
    info->tile_launches_per_matvec = static_cast<int32_t>(std::max<size_t>(p0.bands_.size(), 1)) * p0.passes_per_matvec();  // bands x feature panels
    info->persistent_launches = 0;
    if (p0.queue_.p != nullptr) {
        for (const auto &band : p0.bands_) info->persistent_launches += band.item_count > p0.queue_min_items_ ? p0.passes_per_matvec() : 0;
    }
    info->exchange = exchange_ == Exchange::none ? 0 : ((exchange_ == Exchange::peer || exchange_ == Exchange::process_peer) ? 2 : 1);
    // what RCCL itself says about the communicator the partial vectors travel over (a bench line can then prove that N ranks met)
    info->rccl_nranks = 0;
    info->rccl_rank = info->rccl_device = -1;
    ncclComm_t used = exchange_ == Exchange::process_rccl ? comm().comm : (exchange_ == Exchange::local_rccl && local_comms_ ? local_comms_->comms[0] : nullptr);
    if (used != nullptr) {
        int v = 0;
        if (comm().pCommCount(used, &v) == ncclSuccess) info->rccl_nranks = v;
        if (comm().pCommUserRank(used, &v) == ncclSuccess) info->rccl_rank = v;
        if (comm().pCommCuDevice(used, &v) == ncclSuccess) info->rccl_device = v;
    }
}

template class Solver<float>;
template class Solver<double>;

}  // namespace lssvm
