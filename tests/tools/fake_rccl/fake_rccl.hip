/*
 * fake_rccl.hip -- TEST-ONLY stand-in for librccl.so.1 (built as tests/tools/fake_rccl/librccl.so.1).  NOT part of the product:
 * libplssvm_amd.so never names, links or looks for this file; a test harness loads it into a CHILD process before the product library
 * (ctypes.CDLL(path) ahead of everything else, or LD_LIBRARY_PATH), so that the product's own `dlopen("librccl.so.1")` resolves to it by
 * SONAME.  (tests/test_capi_symbols.py checks that the product library carries no reference to it.)
 *
 * Why it exists ("Next round" item 1): the real RCCL refuses two ranks on one device, and the boxes of this pool have ONE
 * MI355X -- so the product's RCCL exchange (Solver<T>::exchange, Exchange::process_rccl / local_rccl in plssvm_amd/csrc/lssvm_problem.hip;
 * it replaces the reference's host-staged gpu_csvm::device_reduction, include/plssvm/backends/gpu_csvm.hpp:449-475) had never executed with
 * more than one rank.  This library implements exactly the entry points the product binds (lssvm_problem.hip, comm_load) with RCCL's
 * semantics -- stream-ordered, asynchronous to the calling host thread, in-place capable, group calls for several communicators of one
 * process -- for ANY placement of the ranks, several per device included:
 *
 *   ncclGetUniqueId / ncclCommInitRank   one communicator per PROCESS.  The unique id names a POSIX shared-memory page; every rank
 *                                        allocates a staging buffer, publishes its HIP-IPC handle there and maps those of its peers.
 *       ncclAllReduce / ncclAllGather    on the caller's stream:  host function (wait until every peer has consumed the previous
 *                                        collective) -> copy sendbuff into the own staging buffer -> host function (publish "ready",
 *                                        wait for every peer's "ready") -> kernel: recvbuff = fixed-order sum over the ranks' staging
 *                                        buffers (rank 0 + rank 1 + ...; all-gather: the slices in rank order) -> host function
 *                                        (publish "consumed").  Nothing spins on the device; a wait that exceeds FAKE_RCCL_TIMEOUT_S
 *                                        (default 120) raises the page's abort flag and every later call returns ncclSystemError.
 *                                        FAKE_RCCL_SYNC=1: the same steps from the calling thread with stream synchronisations instead
 *                                        of host functions.
 *   ncclCommInitAll                      the communicators of ONE process (devices may repeat).  Collectives must be issued inside
 *       ncclGroupStart / ncclGroupEnd    a group (outside one a multi-rank call would deadlock in RCCL too: ncclInvalidUsage here);
 *                                        at ncclGroupEnd the calls of the group's ranks are matched and scheduled with events only:
 *                                        every stream records "ready", waits for the others' "ready", reduces all send buffers in rank
 *                                        order into a private buffer, records "done", waits for the others' "done", and copies
 *                                        the result into its recvbuff.
 *   ncclCommCount / ncclCommCuDevice / ncclCommUserRank / ncclCommDestroy / ncclGetErrorString / ncclGetVersion
 *
 * The sum order (rank 0 first) is the order of the product's own peer kernels (k_peer_sum), so a test can compare the two exchanges
 * bit for bit.  float32 and float64 with ncclSum only -- what the path uses.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

constexpr int MAX_RANKS = 16;
constexpr uint32_t PAGE_MAGIC = 0x46524343u;  // "FRCC"

struct PeerPtrs {
    const void *p[MAX_RANKS];
};

template <typename T>
__global__ void k_fake_sum(PeerPtrs src, int nranks, size_t count, T *__restrict__ out) {
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        T s = static_cast<const T *>(src.p[0])[i];
        for (int r = 1; r < nranks; ++r) s += static_cast<const T *>(src.p[r])[i];
        out[i] = s;
    }
}
template <typename T>
__global__ void k_fake_gather(PeerPtrs src, int nranks, size_t sendcount, T *__restrict__ out, int skip_rank) {
    const size_t total = sendcount * static_cast<size_t>(nranks);
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const int r = static_cast<int>(i / sendcount);
        if (r == skip_rank) continue;  // in place: the own slice already sits where it belongs
        out[i] = static_cast<const T *>(src.p[r])[i - static_cast<size_t>(r) * sendcount];
    }
}

inline int grid_for(size_t count) { return static_cast<int>(std::min<size_t>((count + 255) / 256, 2048)); }

struct alignas(128) Slot {
    std::atomic<uint32_t> state;  // 0 empty, 1 handle published, 2 peers mapped, 3 destroyed
    int32_t pid;
    int32_t device;
    hipIpcMemHandle_t mem;
    alignas(64) std::atomic<uint64_t> ready;     // sequence number of the last collective whose input sits in this rank's staging buffer
    alignas(64) std::atomic<uint64_t> consumed;  // ... of the last collective whose inputs this rank has finished reading
};
struct Page {
    std::atomic<uint32_t> magic;
    std::atomic<int32_t> abort;
    int32_t nranks;
    Slot slots[MAX_RANKS];
};

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return static_cast<double>(ts.tv_sec) + 1e-9 * static_cast<double>(ts.tv_nsec);
}
double timeout_s() {
    static const double t = [] {
        const char *e = std::getenv("FAKE_RCCL_TIMEOUT_S");
        const double v = e != nullptr ? std::atof(e) : 120.0;
        return v > 0.0 ? v : 120.0;
    }();
    return t;
}
bool sync_mode() {
    static const bool s = [] {
        const char *e = std::getenv("FAKE_RCCL_SYNC");
        return e != nullptr && e[0] == '1';
    }();
    return s;
}
size_t staging_bytes() {
    static const size_t b = [] {
        const char *e = std::getenv("FAKE_RCCL_STAGING_MB");
        const long mb = e != nullptr ? std::atol(e) : 64;
        return static_cast<size_t>(mb > 0 ? mb : 64) << 20;
    }();
    return b;
}
void nap() {
    timespec ts{ 0, 20000 };
    nanosleep(&ts, nullptr);
}

struct LocalGroup;

}  // namespace

struct ncclComm {
    int rank = 0, nranks = 1, device = 0;
    // one communicator per process (ncclCommInitRank)
    Page *page = nullptr;
    void *staging = nullptr;
    void *peer_staging[MAX_RANKS] = {};
    uint64_t seq = 0;
    std::atomic<int> failed{ 0 };
    // communicators of one process (ncclCommInitAll)
    std::shared_ptr<LocalGroup> group;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    void *temp = nullptr;
    size_t temp_bytes = 0;
};

namespace {

struct LocalGroup {
    std::vector<ncclComm *> comms;
};

enum class Kind { all_reduce, all_gather };
struct Op {
    Kind kind;
    ncclComm *comm;
    const void *send;
    void *recv;
    size_t count;
    ncclDataType_t dt;
    hipStream_t stream;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

size_t dt_size(ncclDataType_t dt) { return dt == ncclFloat32 ? 4 : (dt == ncclFloat64 ? 8 : 0); }

#define FAKE_HIP(call)                                                                                                  \
    do {                                                                                                                \
        const hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) {                                                                                         \
            std::fprintf(stderr, "[fake_rccl] %s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            (void) hipGetLastError();                                                                                   \
            return ncclUnhandledCudaError;                                                                              \
        }                                                                                                               \
    } while (0)

/* ---- waits on the shared page (host functions of the stream, or the calling thread in sync mode) ---- */
struct HostStep {
    ncclComm *comm;
    uint64_t seq;
    int what;  // 0: wait until every peer has consumed seq; 1: publish ready = seq and wait for every peer's; 2: publish consumed = seq
};
bool wait_all(ncclComm *c, bool consumed, uint64_t seq) {
    const double t0 = now_s();
    for (int r = 0; r < c->nranks; ++r) {
        if (r == c->rank) continue;
        const Slot &s = c->page->slots[r];
        for (;;) {
            const uint64_t v = consumed ? s.consumed.load(std::memory_order_acquire) : s.ready.load(std::memory_order_acquire);
            if (v >= seq) break;
            if (c->page->abort.load(std::memory_order_relaxed) != 0) return false;
            if (now_s() - t0 > timeout_s()) {
                std::fprintf(stderr, "[fake_rccl] rank %d waited more than %.0f s for rank %d (%s %llu)\n", c->rank, timeout_s(), r, consumed ? "consumed" : "ready",
                             static_cast<unsigned long long>(seq));
                c->page->abort.store(1);
                return false;
            }
            nap();
        }
    }
    return true;
}
void run_step(const HostStep &h) {
    ncclComm *c = h.comm;
    Slot &own = c->page->slots[c->rank];
    bool ok = true;
    switch (h.what) {
        case 0: ok = wait_all(c, true, h.seq); break;
        case 1:
            own.ready.store(h.seq, std::memory_order_release);
            ok = wait_all(c, false, h.seq);
            break;
        default: own.consumed.store(h.seq, std::memory_order_release); break;
    }
    if (!ok) c->failed.store(1);
}
void host_fn(void *user) {
    HostStep *h = static_cast<HostStep *>(user);
    run_step(*h);
    delete h;
}
ncclResult_t step(ncclComm *c, hipStream_t st, uint64_t seq, int what) {
    if (sync_mode()) {
        FAKE_HIP(hipStreamSynchronize(st));
        run_step(HostStep{ c, seq, what });
        return c->failed.load() != 0 ? ncclSystemError : ncclSuccess;
    }
    FAKE_HIP(hipLaunchHostFunc(st, host_fn, new HostStep{ c, seq, what }));
    return ncclSuccess;
}

ncclResult_t run_process(const Op &op) {
    ncclComm *c = op.comm;
    const size_t es = dt_size(op.dt);
    const size_t bytes = op.count * es;
    if (c->failed.load() != 0 || (c->page != nullptr && c->page->abort.load() != 0)) return ncclSystemError;
    if (c->nranks == 1) {
        if (op.send != op.recv) FAKE_HIP(hipMemcpyAsync(op.recv, op.send, bytes, hipMemcpyDeviceToDevice, op.stream));
        return ncclSuccess;
    }
    if (bytes > staging_bytes()) {
        std::fprintf(stderr, "[fake_rccl] a collective of %zu bytes exceeds the staging buffer (FAKE_RCCL_STAGING_MB)\n", bytes);
        return ncclInvalidArgument;
    }
    int dev = 0;
    FAKE_HIP(hipGetDevice(&dev));
    if (dev != c->device) FAKE_HIP(hipSetDevice(c->device));
    const uint64_t seq = ++c->seq;
    ncclResult_t rc = step(c, op.stream, seq - 1, 0);
    if (rc != ncclSuccess) return rc;
    FAKE_HIP(hipMemcpyAsync(c->staging, op.send, bytes, hipMemcpyDeviceToDevice, op.stream));
    rc = step(c, op.stream, seq, 1);
    if (rc != ncclSuccess) return rc;
    PeerPtrs src{};
    for (int r = 0; r < c->nranks; ++r) src.p[r] = c->peer_staging[r];
    if (op.kind == Kind::all_reduce) {
        if (es == 4) hipLaunchKernelGGL(k_fake_sum<float>, dim3(grid_for(op.count)), dim3(256), 0, op.stream, src, c->nranks, op.count, static_cast<float *>(op.recv));
        else hipLaunchKernelGGL(k_fake_sum<double>, dim3(grid_for(op.count)), dim3(256), 0, op.stream, src, c->nranks, op.count, static_cast<double *>(op.recv));
    } else {
        // recvbuff holds nranks slices; an in-place call (sendbuff = recvbuff + rank * sendcount) keeps its own slice
        const bool in_place = op.send == static_cast<const char *>(op.recv) + static_cast<size_t>(c->rank) * bytes;
        const size_t total = op.count * static_cast<size_t>(c->nranks);
        if (es == 4) hipLaunchKernelGGL(k_fake_gather<float>, dim3(grid_for(total)), dim3(256), 0, op.stream, src, c->nranks, op.count, static_cast<float *>(op.recv), in_place ? c->rank : -1);
        else hipLaunchKernelGGL(k_fake_gather<double>, dim3(grid_for(total)), dim3(256), 0, op.stream, src, c->nranks, op.count, static_cast<double *>(op.recv), in_place ? c->rank : -1);
    }
    FAKE_HIP(hipGetLastError());
    rc = step(c, op.stream, seq, 2);
    if (dev != c->device) FAKE_HIP(hipSetDevice(dev));
    return rc;
}

/* the matched calls of ONE collective over the communicators of a process: events only, nothing blocks the host */
ncclResult_t run_local(const std::vector<Op> &ops) {
    const int n = static_cast<int>(ops.size());
    const Op &o0 = ops[0];
    const size_t es = dt_size(o0.dt);
    const size_t bytes = o0.count * es;
    int dev0 = 0;
    FAKE_HIP(hipGetDevice(&dev0));
    for (int r = 0; r < n; ++r) {
        ncclComm *c = ops[r].comm;
        FAKE_HIP(hipSetDevice(c->device));
        if (o0.kind == Kind::all_reduce && c->temp_bytes < bytes) {
            if (c->temp != nullptr) {
                FAKE_HIP(hipDeviceSynchronize());
                FAKE_HIP(hipFree(c->temp));
            }
            FAKE_HIP(hipMalloc(&c->temp, bytes));
            c->temp_bytes = bytes;
        }
        FAKE_HIP(hipEventRecord(c->ev_ready, ops[r].stream));
    }
    PeerPtrs src{};
    for (int r = 0; r < n; ++r) src.p[r] = ops[r].send;
    for (int r = 0; r < n; ++r) {
        ncclComm *c = ops[r].comm;
        hipStream_t st = ops[r].stream;
        FAKE_HIP(hipSetDevice(c->device));
        for (int o = 0; o < n; ++o) {
            if (o != r) FAKE_HIP(hipStreamWaitEvent(st, ops[o].comm->ev_ready, 0));
        }
        if (o0.kind == Kind::all_reduce) {
            if (es == 4) hipLaunchKernelGGL(k_fake_sum<float>, dim3(grid_for(o0.count)), dim3(256), 0, st, src, n, o0.count, static_cast<float *>(c->temp));
            else hipLaunchKernelGGL(k_fake_sum<double>, dim3(grid_for(o0.count)), dim3(256), 0, st, src, n, o0.count, static_cast<double *>(c->temp));
        } else {
            const bool in_place = ops[r].send == static_cast<const char *>(ops[r].recv) + static_cast<size_t>(r) * bytes;
            const size_t total = o0.count * static_cast<size_t>(n);
            if (es == 4) hipLaunchKernelGGL(k_fake_gather<float>, dim3(grid_for(total)), dim3(256), 0, st, src, n, o0.count, static_cast<float *>(ops[r].recv), in_place ? r : -1);
            else hipLaunchKernelGGL(k_fake_gather<double>, dim3(grid_for(total)), dim3(256), 0, st, src, n, o0.count, static_cast<double *>(ops[r].recv), in_place ? r : -1);
        }
        FAKE_HIP(hipGetLastError());
        FAKE_HIP(hipEventRecord(c->ev_done, st));
    }
    // a send buffer may be overwritten -- by the result itself when in place, by the caller's next kernel otherwise -- only after every rank has read it
    for (int r = 0; r < n; ++r) {
        ncclComm *c = ops[r].comm;
        hipStream_t st = ops[r].stream;
        FAKE_HIP(hipSetDevice(c->device));
        for (int o = 0; o < n; ++o) {
            if (o != r) FAKE_HIP(hipStreamWaitEvent(st, ops[o].comm->ev_done, 0));
        }
        if (o0.kind == Kind::all_reduce) FAKE_HIP(hipMemcpyAsync(ops[r].recv, c->temp, bytes, hipMemcpyDeviceToDevice, st));
    }
    FAKE_HIP(hipSetDevice(dev0));
    return ncclSuccess;
}

ncclResult_t flush_group() {
    std::vector<Op> ops;
    ops.swap(g_ops);
    std::vector<bool> used(ops.size(), false);
    for (size_t i = 0; i < ops.size(); ++i) {
        if (used[i]) continue;
        ncclComm *c = ops[i].comm;
        if (!c->group || c->group->comms.size() == 1) {
            used[i] = true;
            const ncclResult_t rc = run_process(ops[i]);
            if (rc != ncclSuccess) return rc;
            continue;
        }
        // the first pending call of every rank of this communicator's group forms one collective
        const LocalGroup &lg = *c->group;
        std::vector<Op> matched(lg.comms.size());
        for (size_t r = 0; r < lg.comms.size(); ++r) {
            bool found = false;
            for (size_t k = i; k < ops.size() && !found; ++k) {
                if (!used[k] && ops[k].comm == lg.comms[r]) {
                    matched[r] = ops[k];
                    used[k] = true;
                    found = true;
                }
            }
            if (!found) {
                std::fprintf(stderr, "[fake_rccl] group ended without a call for rank %zu of %zu (RCCL would hang here)\n", r, lg.comms.size());
                return ncclInvalidUsage;
            }
            if (matched[r].kind != matched[0].kind || matched[r].count != matched[0].count || matched[r].dt != matched[0].dt) {
                std::fprintf(stderr, "[fake_rccl] the ranks of a group disagree on the collective (kind / count / type)\n");
                return ncclInvalidUsage;
            }
        }
        const ncclResult_t rc = run_local(matched);
        if (rc != ncclSuccess) return rc;
    }
    return ncclSuccess;
}

ncclResult_t submit(const Op &op) {
    if (op.comm == nullptr || op.send == nullptr || op.recv == nullptr) return ncclInvalidArgument;
    if (dt_size(op.dt) == 0) return ncclInvalidArgument;
    if (g_depth > 0) {
        g_ops.push_back(op);
        return ncclSuccess;
    }
    if (op.comm->group && op.comm->group->comms.size() > 1) {
        std::fprintf(stderr, "[fake_rccl] a collective over several communicators of one process must be issued inside ncclGroupStart / ncclGroupEnd\n");
        return ncclInvalidUsage;
    }
    return run_process(op);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *version) {
    if (version == nullptr) return ncclInvalidArgument;
    *version = 0;  // no RCCL release has version 0: a stand-in
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t result) {
    switch (result) {
        case ncclSuccess: return "no error (fake_rccl)";
        case ncclUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
        case ncclSystemError: return "unhandled system error (fake_rccl)";
        case ncclInternalError: return "internal error (fake_rccl)";
        case ncclInvalidArgument: return "invalid argument (fake_rccl)";
        case ncclInvalidUsage: return "invalid usage (fake_rccl)";
        default: return "unknown result code (fake_rccl)";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *uniqueId) {
    if (uniqueId == nullptr) return ncclInvalidArgument;
    static std::atomic<unsigned> counter{ 0 };
    std::memset(uniqueId, 0, sizeof(*uniqueId));
    char name[96];
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    std::snprintf(name, sizeof(name), "/fake_rccl_%d_%u_%08lx", static_cast<int>(getpid()), counter.fetch_add(1), static_cast<unsigned long>(ts.tv_nsec));
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, sizeof(Page)) != 0) {
        close(fd);
        shm_unlink(name);
        return ncclSystemError;
    }
    void *m = mmap(nullptr, sizeof(Page), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        shm_unlink(name);
        return ncclSystemError;
    }
    Page *pg = new (m) Page;  // zero-filled by ftruncate; constructs the atomics
    pg->magic.store(PAGE_MAGIC, std::memory_order_release);
    munmap(m, sizeof(Page));
    std::memcpy(uniqueId->internal, name, std::strlen(name) + 1);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId commId, int rank) {
    if (comm == nullptr || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    *comm = nullptr;
    char name[sizeof(commId.internal) + 1];
    std::memcpy(name, commId.internal, sizeof(commId.internal));
    name[sizeof(commId.internal)] = 0;
    if (std::strncmp(name, "/fake_rccl_", 11) != 0) return ncclInvalidArgument;
    auto c = std::make_unique<ncclComm>();
    c->rank = rank;
    c->nranks = nranks;
    FAKE_HIP(hipGetDevice(&c->device));
    const int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    void *m = mmap(nullptr, sizeof(Page), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return ncclSystemError;
    c->page = static_cast<Page *>(m);
    if (c->page->magic.load(std::memory_order_acquire) != PAGE_MAGIC) return ncclSystemError;
    Slot &own = c->page->slots[rank];
    if (own.state.load() != 0) {
        std::fprintf(stderr, "[fake_rccl] rank %d of this unique id exists already\n", rank);
        return ncclInvalidUsage;
    }
    FAKE_HIP(hipMalloc(&c->staging, staging_bytes()));
    c->peer_staging[rank] = c->staging;
    own.pid = static_cast<int32_t>(getpid());
    own.device = c->device;
    if (nranks > 1) FAKE_HIP(hipIpcGetMemHandle(&own.mem, c->staging));
    own.state.store(1, std::memory_order_release);
    const double t0 = now_s();
    auto wait_state = [&](uint32_t want) {
        for (int r = 0; r < nranks; ++r) {
            while (c->page->slots[r].state.load(std::memory_order_acquire) < want) {
                if (now_s() - t0 > timeout_s() || c->page->abort.load() != 0) {
                    c->page->abort.store(1);
                    return false;
                }
                nap();
            }
        }
        return true;
    };
    if (!wait_state(1)) return ncclSystemError;
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) continue;
        const Slot &s = c->page->slots[r];
        if (s.pid == own.pid) {
            std::fprintf(stderr, "[fake_rccl] ranks %d and %d live in one process: use ncclCommInitAll\n", rank, r);
            c->page->abort.store(1);
            return ncclInvalidUsage;
        }
        hipIpcMemHandle_t h = s.mem;
        FAKE_HIP(hipIpcOpenMemHandle(&c->peer_staging[r], h, hipIpcMemLazyEnablePeerAccess));
    }
    own.state.store(2, std::memory_order_release);
    if (!wait_state(2)) return ncclSystemError;
    if (rank == 0) shm_unlink(name);  // every rank holds its mapping: the name can go
    *comm = c.release();
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist) {
    if (comm == nullptr || ndev < 1 || ndev > MAX_RANKS) return ncclInvalidArgument;
    int dev0 = 0, count = 0;
    FAKE_HIP(hipGetDevice(&dev0));
    FAKE_HIP(hipGetDeviceCount(&count));
    auto lg = std::make_shared<LocalGroup>();
    for (int r = 0; r < ndev; ++r) {
        auto *c = new ncclComm;
        c->rank = r;
        c->nranks = ndev;
        c->device = devlist != nullptr ? devlist[r] : r;
        if (c->device < 0 || c->device >= count) return ncclInvalidArgument;
        c->group = lg;
        FAKE_HIP(hipSetDevice(c->device));
        FAKE_HIP(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
        FAKE_HIP(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
        lg->comms.push_back(c);
        comm[r] = c;
    }
    // distinct devices read each other's buffers directly
    for (int r = 0; r < ndev; ++r) {
        FAKE_HIP(hipSetDevice(lg->comms[r]->device));
        for (int o = 0; o < ndev; ++o) {
            if (lg->comms[o]->device != lg->comms[r]->device) {
                (void) hipDeviceEnablePeerAccess(lg->comms[o]->device, 0);
                (void) hipGetLastError();
            }
        }
    }
    FAKE_HIP(hipSetDevice(dev0));
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (comm == nullptr) return ncclSuccess;
    int dev0 = 0;
    (void) hipGetDevice(&dev0);
    (void) hipSetDevice(comm->device);
    (void) hipDeviceSynchronize();
    if (comm->page != nullptr) {
        // the peers read this rank's staging buffer: it goes only after every rank has arrived here (or after a short wait)
        comm->page->slots[comm->rank].state.store(3, std::memory_order_release);
        const double t0 = now_s();
        for (int r = 0; r < comm->nranks; ++r) {
            while (comm->page->slots[r].state.load(std::memory_order_acquire) < 3 && comm->page->abort.load() == 0 && now_s() - t0 < 10.0) nap();
        }
        for (int r = 0; r < comm->nranks; ++r) {
            if (r != comm->rank && comm->peer_staging[r] != nullptr) (void) hipIpcCloseMemHandle(comm->peer_staging[r]);
        }
        if (comm->staging != nullptr) (void) hipFree(comm->staging);
        munmap(comm->page, sizeof(Page));
    }
    if (comm->ev_ready != nullptr) (void) hipEventDestroy(comm->ev_ready);
    if (comm->ev_done != nullptr) (void) hipEventDestroy(comm->ev_done);
    if (comm->temp != nullptr) (void) hipFree(comm->temp);
    if (comm->group) {
        for (auto &p : comm->group->comms) {
            if (p == comm) p = nullptr;
        }
    }
    (void) hipGetLastError();
    (void) hipSetDevice(dev0);
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
    if (comm == nullptr || count == nullptr) return ncclInvalidArgument;
    *count = comm->nranks;
    return ncclSuccess;
}
ncclResult_t ncclCommCuDevice(const ncclComm_t comm, int *device) {
    if (comm == nullptr || device == nullptr) return ncclInvalidArgument;
    *device = comm->device;
    return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) {
    if (comm == nullptr || rank == nullptr) return ncclInvalidArgument;
    *rank = comm->rank;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    if (op != ncclSum) return ncclInvalidArgument;
    return submit(Op{ Kind::all_reduce, comm, sendbuff, recvbuff, count, datatype, stream });
}
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    return submit(Op{ Kind::all_gather, comm, sendbuff, recvbuff, sendcount, datatype, stream });
}

ncclResult_t ncclGroupStart() {
    ++g_depth;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    return flush_group();
}

/* marker: lets a harness assert that THIS library (and not a real RCCL) is what a process loaded */
int fake_rccl_marker(void) { return 0x46524343; }

}  // extern "C"
