/*
 * lssvm_exchange.hip -- how the shards of a row-block sharded problem combine their partial K*v vectors once per implicit matvec (replaces the reference's
 * host-staged gpu_csvm::device_reduction, include/plssvm/backends/gpu_csvm.hpp:449-475): the lazily loaded RCCL entry points and communicators, the peer
 * kernels over xGMI, HIP IPC + host flags between the processes of a node, and Solver<T>::exchange itself.  Compiled for gfx950 only.
 */
#include "lssvm_problem.hip.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <thread>

namespace lssvm {

Comm &comm() {
    static Comm c;
    return c;
}

void comm_load() {
    static std::mutex m;
    const std::lock_guard<std::mutex> lock(m);
    Comm &c = comm();
    if (c.lib != nullptr) return;
    // if the process already carries an RCCL (e.g. PyTorch's), reuse it; otherwise load the system one
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void *lib = nullptr;
    for (const char *name : names) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib != nullptr) break;
    }
    if (lib == nullptr) throw Error(LSSVM_ERR_COMM, std::string("could not load RCCL: ") + dlerror());
    c.pGetUniqueId = reinterpret_cast<decltype(c.pGetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    c.pCommInitRank = reinterpret_cast<decltype(c.pCommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    c.pCommInitAll = reinterpret_cast<decltype(c.pCommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    c.pCommDestroy = reinterpret_cast<decltype(c.pCommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    c.pAllGather = reinterpret_cast<decltype(c.pAllGather)>(dlsym(lib, "ncclAllGather"));
    c.pAllReduce = reinterpret_cast<decltype(c.pAllReduce)>(dlsym(lib, "ncclAllReduce"));
    c.pGroupStart = reinterpret_cast<decltype(c.pGroupStart)>(dlsym(lib, "ncclGroupStart"));
    c.pGroupEnd = reinterpret_cast<decltype(c.pGroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    c.pGetErrorString = reinterpret_cast<decltype(c.pGetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    c.pCommCount = reinterpret_cast<decltype(c.pCommCount)>(dlsym(lib, "ncclCommCount"));
    c.pCommCuDevice = reinterpret_cast<decltype(c.pCommCuDevice)>(dlsym(lib, "ncclCommCuDevice"));
    c.pCommUserRank = reinterpret_cast<decltype(c.pCommUserRank)>(dlsym(lib, "ncclCommUserRank"));
    if (!c.pCommCount || !c.pCommCuDevice || !c.pCommUserRank || !c.pGetUniqueId || !c.pCommInitRank || !c.pCommInitAll || !c.pCommDestroy || !c.pAllGather || !c.pAllReduce || !c.pGroupStart || !c.pGroupEnd || !c.pGetErrorString) {
        throw Error(LSSVM_ERR_COMM, "the loaded RCCL library lacks a required symbol");
    }
    c.lib = lib;
}

void nccl_check(ncclResult_t rc, const char *what) {
    if (rc != ncclSuccess) throw Error(LSSVM_ERR_COMM, std::string(what) + " failed: " + comm().pGetErrorString(rc));
}

LocalComms::~LocalComms() {
    for (size_t r = 0; r < comms.size(); ++r) {
        if (comms[r] != nullptr) {
            (void) hipSetDevice(devices[r]);
            (void) comm().pCommDestroy(comms[r]);
        }
    }
}

/* the communicators of one device list are created once and re-used by later solves (ncclCommInitAll costs ~0.1 s per device);
 * the cache itself is never destroyed: tearing RCCL down from a static destructor races with the HIP runtime's own exit */
std::shared_ptr<LocalComms> local_comms_for(const std::vector<int> &devices) {
    static std::mutex m;
    static auto *cached = new std::shared_ptr<LocalComms>();
    const std::lock_guard<std::mutex> lock(m);
    if (*cached && (*cached)->devices == devices) return *cached;
    comm_load();
    cached->reset();  // communicators of another device list: destroyed here unless a live solver still holds them
    auto lc = std::make_shared<LocalComms>();
    lc->devices = devices;
    lc->comms.assign(devices.size(), nullptr);
    nccl_check(comm().pCommInitAll(lc->comms.data(), static_cast<int>(devices.size()), devices.data()), "ncclCommInitAll");
    *cached = lc;
    return lc;
}

/* ------------------------------------------------------------------ IpcPeers: one process per GPU without RCCL ------------------------------------------------------------------ */
constexpr uint32_t IPC_MAGIC = 0x4956534Cu;  // "LSVI"

IpcPeers::IpcPeers(int rank_in, int world_in) : rank(rank_in), world(world_in), flags(static_cast<size_t>(world_in), nullptr), vectors(static_cast<size_t>(world_in), nullptr) {
    static std::atomic<unsigned> serial{ 0 };
    own_name = "/plssvm_amd." + std::to_string(static_cast<long>(getpid())) + "." + std::to_string(serial.fetch_add(1)) + "." + std::to_string(rank);
    const int fd = shm_open(own_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) throw Error(LSSVM_ERR_COMM, "shm_open(" + own_name + ") failed: " + std::strerror(errno));
    if (ftruncate(fd, 4096) != 0) {
        (void) close(fd);
        (void) shm_unlink(own_name.c_str());
        throw Error(LSSVM_ERR_COMM, "ftruncate of the flag page failed");
    }
    void *m = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    (void) close(fd);
    if (m == MAP_FAILED) {
        (void) shm_unlink(own_name.c_str());
        throw Error(LSSVM_ERR_COMM, "mmap of the flag page failed");
    }
    own = new (m) IpcFlags{};  // a fresh segment is zero filled: ready = consumed = abort = 0
    flags[static_cast<size_t>(rank)] = own;
}

IpcPeers::~IpcPeers() {
    for (int r = 0; r < world; ++r) {
        if (r == rank) continue;
        if (vectors[static_cast<size_t>(r)] != nullptr) (void) hipIpcCloseMemHandle(vectors[static_cast<size_t>(r)]);
        if (flags[static_cast<size_t>(r)] != nullptr) (void) munmap(flags[static_cast<size_t>(r)], 4096);
    }
    if (own != nullptr) (void) munmap(own, 4096);
    if (!own_name.empty()) (void) shm_unlink(own_name.c_str());
}

void IpcPeers::connect(const IpcBlob *blobs, void *own_vector) {
    LSSVM_REQUIRE(!connected, "the peers of this problem are connected already");
    for (int r = 0; r < world; ++r) {
        const IpcBlob &b = blobs[r];
        LSSVM_REQUIRE(b.magic == IPC_MAGIC && b.rank == r && b.world == world, "blob " + std::to_string(r) + " is not the export of rank " + std::to_string(r) + " of this world");
        if (r == rank) {
            vectors[static_cast<size_t>(r)] = own_vector;
            continue;
        }
        char name[sizeof(b.shm_name) + 1] = {};
        std::memcpy(name, b.shm_name, sizeof(b.shm_name));
        const int fd = shm_open(name, O_RDONLY, 0);
        if (fd < 0) throw Error(LSSVM_ERR_COMM, std::string("shm_open(") + name + ") of rank " + std::to_string(r) + " failed: " + std::strerror(errno) + " (all ranks must run on one node)");
        void *m = mmap(nullptr, 4096, PROT_READ, MAP_SHARED, fd, 0);
        (void) close(fd);
        if (m == MAP_FAILED) throw Error(LSSVM_ERR_COMM, "mmap of the flag page of rank " + std::to_string(r) + " failed");
        flags[static_cast<size_t>(r)] = static_cast<IpcFlags *>(m);
        void *ptr = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&ptr, b.mem, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void) hipGetLastError();
            throw Error(LSSVM_ERR_COMM, "hipIpcOpenMemHandle of rank " + std::to_string(r) + "'s vector failed: " + hipGetErrorString(e) + " (HSA_ENABLE_IPC_MODE_LEGACY=0 set on every rank?)");
        }
        vectors[static_cast<size_t>(r)] = ptr;
    }
    connected = true;
}

void IpcPeers::wait_all(int which, uint64_t seq, double timeout_s) {
    const double t0 = now_ms();
    for (int r = 0; r < world; ++r) {
        if (r == rank) continue;
        const IpcFlags *f = flags[static_cast<size_t>(r)];
        const std::atomic<uint64_t> &counter = which == 0 ? f->ready : f->consumed;
        unsigned spins = 0;
        while (counter.load(std::memory_order_acquire) < seq) {
            if (f->abort.load(std::memory_order_acquire) != 0) {
                own->abort.store(1, std::memory_order_release);
                throw Error(LSSVM_ERR_COMM, "rank " + std::to_string(r) + " gave up (see its error)");
            }
            if (++spins > 2000) {
                std::this_thread::yield();
                if ((spins & 1023u) == 0 && now_ms() - t0 > timeout_s * 1e3) {
                    own->abort.store(1, std::memory_order_release);
                    throw Error(LSSVM_ERR_COMM, "rank " + std::to_string(r) + " did not reach implicit matvec " + std::to_string(seq) + " within " + std::to_string(timeout_s) + " s");
                }
            }
        }
    }
}

/* peer exchange: every device reads the partial vectors of ALL shards through its peer mappings (xGMI) and adds them in rank
 * order -- the same order on every device, so the exchanged vectors are bit-equal -- or, for the full-square variant, picks every
 * entry from the shard that owns its row */
struct PeerPtrs {
    const void *p[MAX_LOCAL_DEVICES];
};
template <typename T>
__global__ void k_peer_sum(PeerPtrs src, int world, int n, T *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T s = static_cast<const T *>(src.p[0])[i];
    for (int r = 1; r < world; ++r) s += static_cast<const T *>(src.p[r])[i];
    out[i] = s;
}
template <typename T>
__global__ void k_peer_gather(PeerPtrs src, int slice, int n, T *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = static_cast<const T *>(src.p[i / slice])[i];
}

template <typename T>
void Solver<T>::ipc_export(void *blob_out, size_t blob_bytes) {
    LSSVM_REQUIRE(exchange_ == Exchange::process_peer, "this problem does not use the IPC peer exchange (one process per GPU, option exchange = 2 or no RCCL communicator)");
    LSSVM_REQUIRE(blob_out != nullptr && blob_bytes >= IPC_BLOB_BYTES, "the blob buffer must hold LSSVM_IPC_BLOB_BYTES bytes");
    Problem<T> &p = *shards_[0];
    p.activate();
    LSSVM_HIP_CHECK(hipStreamSynchronize(p.stream()));
    IpcBlob b{};
    b.magic = IPC_MAGIC;
    b.rank = ipc_->rank;
    b.world = world_;
    b.device = p.device_;
    b.nvec = static_cast<uint64_t>(p.nvec_);
    b.real_size = sizeof(T);
    b.pid = static_cast<int32_t>(getpid());
    const hipError_t e = hipIpcGetMemHandle(&b.mem, p.Kv_.p);
    if (e != hipSuccess) {
        (void) hipGetLastError();
        throw Error(LSSVM_ERR_COMM, std::string("hipIpcGetMemHandle failed: ") + hipGetErrorString(e) + " (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
    }
    LSSVM_REQUIRE(ipc_->own_name.size() < sizeof(b.shm_name), "flag page name too long");
    std::memcpy(b.shm_name, ipc_->own_name.data(), ipc_->own_name.size());
    std::memset(blob_out, 0, IPC_BLOB_BYTES);
    std::memcpy(blob_out, &b, sizeof(b));
}

template <typename T>
void Solver<T>::ipc_connect(const void *blobs, size_t total_bytes) {
    LSSVM_REQUIRE(exchange_ == Exchange::process_peer, "this problem does not use the IPC peer exchange");
    LSSVM_REQUIRE(blobs != nullptr && total_bytes == static_cast<size_t>(world_) * IPC_BLOB_BYTES, "expected world x LSSVM_IPC_BLOB_BYTES bytes, the exports of all ranks in rank order");
    Problem<T> &p = *shards_[0];
    p.activate();
    std::vector<IpcBlob> all(static_cast<size_t>(world_));
    for (int r = 0; r < world_; ++r) {
        std::memcpy(&all[static_cast<size_t>(r)], static_cast<const unsigned char *>(blobs) + static_cast<size_t>(r) * IPC_BLOB_BYTES, sizeof(IpcBlob));
        LSSVM_REQUIRE(all[static_cast<size_t>(r)].nvec == static_cast<uint64_t>(p.nvec_) && all[static_cast<size_t>(r)].real_size == sizeof(T),
                      "rank " + std::to_string(r) + " holds a problem of another size or type");
    }
    ipc_->connect(all.data(), p.Kv_.p);
}

template <typename T>
void Solver<T>::exchange() {
    Comm &c = comm();
    const ncclDataType_t dt = std::is_same_v<T, float> ? ncclFloat32 : ncclFloat64;
    const bool sym = shards_[0]->sym_;
    const size_t count = static_cast<size_t>(shards_[0]->num_tiles_) * TILE;
    const size_t slice = static_cast<size_t>(shards_[0]->ib_per_rank_) * TILE;
    switch (exchange_) {
        case Exchange::none: break;
        case Exchange::process_rccl: {
            // one collective per implicit matvec.  Symmetric variant: every rank holds partial sums for all rows up to its last
            // block -> sum (in place); full square: every rank contributes its contiguous slice of K*v (in place)
            Problem<T> &p = *shards_[0];
            p.activate();
            if (sym) {
                nccl_check(c.pAllReduce(p.Kv_.p, p.Kv_.p, count, dt, ncclSum, c.comm, p.stream()), "ncclAllReduce");
            } else {
                nccl_check(c.pAllGather(p.Kv_.p + static_cast<size_t>(p.rank_) * slice, p.Kv_.p, slice, dt, c.comm, p.stream()), "ncclAllGather");
            }
            break;
        }
        case Exchange::local_rccl: {
            nccl_check(c.pGroupStart(), "ncclGroupStart");
            for (size_t r = 0; r < shards_.size(); ++r) {
                Problem<T> &p = *shards_[r];
                if (sym) {
                    nccl_check(c.pAllReduce(p.Kv_.p, p.Kv_.p, count, dt, ncclSum, local_comms_->comms[r], p.stream()), "ncclAllReduce");
                } else {
                    nccl_check(c.pAllGather(p.Kv_.p + r * slice, p.Kv_.p, slice, dt, local_comms_->comms[r], p.stream()), "ncclAllGather");
                }
            }
            nccl_check(c.pGroupEnd(), "ncclGroupEnd");
            break;
        }
        case Exchange::process_peer: {
            // one process per GPU without RCCL.  Kernel boundaries are the only device-side ordering used: a rank publishes "ready" on its
            // flag page (host memory shared by the node's ranks) AFTER its stream has drained, i.e. after the tile kernels that wrote the
            // partial vector have ended; the peers then read that vector through their IPC mapping in a kernel launched afterwards
            Problem<T> &p = *shards_[0];
            LSSVM_REQUIRE(ipc_ && ipc_->connected, "the peers of this problem were not connected (lssvm_mi355_problem_ipc_export on every rank, then lssvm_mi355_problem_ipc_connect)");
            p.activate();
            const double timeout_s = static_cast<double>(std::max<int64_t>(opt_.ipc_timeout_s, 1));
            ++xseq_;
            LSSVM_HIP_CHECK(hipStreamSynchronize(p.stream()));
            ipc_->own->ready.store(xseq_, std::memory_order_release);
            ipc_->wait_all(0, xseq_, timeout_s);
            PeerPtrs src{};
            for (int r = 0; r < world_; ++r) src.p[r] = ipc_->vectors[static_cast<size_t>(r)];
            const int n = static_cast<int>(count);
            if (sym) {
                hipLaunchKernelGGL(k_peer_sum<T>, dim3((n + 255) / 256), dim3(256), 0, p.stream(), src, world_, n, p.Ksum_.p);
            } else {
                hipLaunchKernelGGL(k_peer_gather<T>, dim3((n + 255) / 256), dim3(256), 0, p.stream(), src, static_cast<int>(slice), n, p.Ksum_.p);
            }
            LSSVM_HIP_CHECK(hipGetLastError());
            LSSVM_HIP_CHECK(hipStreamSynchronize(p.stream()));
            ipc_->own->consumed.store(xseq_, std::memory_order_release);
            break;
        }
        case Exchange::peer: {
            PeerPtrs src{};
            for (size_t r = 0; r < shards_.size(); ++r) src.p[r] = shards_[r]->Kv_.p;
            for (auto &p : shards_) {
                p->activate();
                LSSVM_HIP_CHECK(hipEventRecord(p->ev_ready_.e, p->stream()));
            }
            const int n = static_cast<int>(count);
            for (auto &p : shards_) {
                p->activate();
                for (auto &o : shards_) {
                    if (o.get() != p.get()) LSSVM_HIP_CHECK(hipStreamWaitEvent(p->stream(), o->ev_ready_.e, 0));
                }
                if (sym) {
                    hipLaunchKernelGGL(k_peer_sum<T>, dim3((n + 255) / 256), dim3(256), 0, p->stream(), src, static_cast<int>(shards_.size()), n, p->Ksum_.p);
                } else {
                    hipLaunchKernelGGL(k_peer_gather<T>, dim3((n + 255) / 256), dim3(256), 0, p->stream(), src, static_cast<int>(slice), n, p->Ksum_.p);
                }
                LSSVM_HIP_CHECK(hipGetLastError());
                LSSVM_HIP_CHECK(hipEventRecord(p->ev_consumed_.e, p->stream()));
            }
            // a shard's partial vector may be overwritten (next matvec) only after every device has read it
            for (auto &p : shards_) {
                p->activate();
                for (auto &o : shards_) {
                    if (o.get() != p.get()) LSSVM_HIP_CHECK(hipStreamWaitEvent(p->stream(), o->ev_consumed_.e, 0));
                }
            }
            break;
        }
    }
}

template void Solver<float>::exchange();
template void Solver<double>::exchange();
template void Solver<float>::ipc_export(void *, size_t);
template void Solver<double>::ipc_export(void *, size_t);
template void Solver<float>::ipc_connect(const void *, size_t);
template void Solver<double>::ipc_connect(const void *, size_t);

}  // namespace lssvm
