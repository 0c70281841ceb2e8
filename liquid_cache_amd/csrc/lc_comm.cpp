// The two exchange steps of a sharded scan behind the C ABI (include/liquid_cache_amd.h, "multi-GPU exchange"), so that a
// Rust host needs nothing but this library: one process per GPU, entries sharded by row range, and
//   lc_comm_allreduce_count   the 8-byte COUNT(*) all-reduce of `SELECT COUNT(*) ... WHERE <pushed-down predicate>`
//   lc_comm_allgather_mask    the per-rank hit-mask segments -> the single BooleanArray north_star names
// Reference counterpart: none in the hot path (the reference runs one process); the shape follows how its DataFusion
// layer merges partition results (a final AggregateExec / CoalescePartitionsExec over per-partition streams).
//
// Backends:
//   * RCCL over xGMI (device contexts).  librccl.so is opened lazily with dlopen, so a single-GPU process never needs it.
//     Variable-length mask segments are gathered with one ncclBroadcast per rank inside a group (all-gather-v).
//   * a shared-memory backend for HOST-ONLY contexts (lc_ctx_create with n_devices = 0): the ranks of one node meet in
//     a file under /dev/shm named by the unique id.  It exists so that the multi-rank logic (ids, offsets, ordering) is
//     exercised by the CPU test suite; pointers are host pointers there.  A DEVICE context takes it under
//     LC_OPT_COMM_SHARED_MEMORY = 1 (device pointers then travel through host copies): the dry run of a multi-rank job
//     whose ranks share ONE GPU, which RCCL refuses (tests, scripts/scale_dryrun.sh) — never a measurement.
#include <cerrno>
#include <cstring>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <string>
#include <thread>

#include "lc_internal.hpp"

using namespace lc;

namespace {

// ---- the handful of RCCL entry points, resolved at run time (signatures of rccl.h, NCCL 2.x ABI)
typedef struct { char internal[128]; } RcclUniqueId;
typedef void* RcclComm;
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
constexpr int kRcclUint64 = 5, kRcclSum = 0;  // ncclUint64, ncclSum

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy the process has loaded already comes first (PyTorch brings its own librccl: two instances of the library in
        // one process would each keep their own topology and proxy threads), then the system's
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (r.lib) break;
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (r.lib) break;
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.lib) return;
        auto sym = [&](const char* n) { return dlsym(r.lib, n); };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.Broadcast && r.GroupStart && r.GroupEnd;
    });
    return r;
}

lc_status rccl_fail(const char* what, int code) {
    const Rccl& r = rccl();
    return fail(LC_ERR_DEVICE, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(code) : "RCCL error"));
}

// ---- shared-memory backend (host-only contexts; one node)
struct ShmHeader {
    std::atomic<uint32_t> arrived;  // ranks that mapped the file
    std::atomic<uint32_t> barrier_count;
    std::atomic<uint32_t> barrier_sense;
    uint32_t world;
    uint64_t slots[64];             // one u64 per rank (counts, segment sizes)
};
constexpr size_t kShmData = size_t(64) << 20;  // mask exchange area
constexpr int kMaxShmRanks = 64;

}  // namespace

struct lc_comm {
    lc_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    RcclComm rc = nullptr;       // RCCL backend
    ShmHeader* shm = nullptr;    // shared-memory backend
    uint8_t* shm_data = nullptr;
    size_t shm_bytes = 0;
    std::string shm_name;
    uint32_t local_sense = 0;
    bool device_ptrs = false;    // shared-memory backend of a DEVICE context: the callers' pointers are device pointers
    bool poisoned = false;       // a rank did not arrive in time: the barrier state is no longer consistent — every later
                                 // collective fails fast instead of hanging or answering from half-written slots
};

namespace {

// false: a rank did not arrive within the time limit (it died: the others must not spin forever)
constexpr int kShmTimeoutSeconds = 120;
template <typename Cond>
bool spin_until(Cond&& done) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t i = 0; !done(); i++) {
        std::this_thread::yield();
        if ((i & 0xFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(kShmTimeoutSeconds)) return false;
    }
    return true;
}

bool shm_barrier(lc_comm* c) {
    if (c->poisoned) return false;
    ShmHeader* h = c->shm;
    c->local_sense ^= 1u;
    if (h->barrier_count.fetch_add(1, std::memory_order_acq_rel) + 1 == uint32_t(c->world)) {
        h->barrier_count.store(0, std::memory_order_relaxed);
        h->barrier_sense.store(c->local_sense, std::memory_order_release);
        return true;
    }
    if (spin_until([&] { return h->barrier_sense.load(std::memory_order_acquire) == c->local_sense; })) return true;
    c->poisoned = true;
    return false;
}

}  // namespace

extern "C" {

lc_status lc_comm_unique_id(lc_ctx* ctx, uint8_t* out_id) {
    return guarded([&]() -> lc_status {
    if (!ctx || !out_id) return fail(LC_ERR_INVALID, "null argument");
    std::memset(out_id, 0, LC_COMM_ID_BYTES);
    if (ctx->device < 0 || ctx->comm_shared_memory.load()) {
        // host-only context (or a device context told to use the test backend): the id names a file under /dev/shm
        static std::atomic<uint32_t> seq{0};
        std::snprintf(reinterpret_cast<char*>(out_id), LC_COMM_ID_BYTES, "/lc_comm_%d_%u_%llu", int(getpid()), seq.fetch_add(1),
                      (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
        return LC_OK;
    }
    Rccl& r = rccl();
    if (!r.ok) return fail(LC_ERR_DEVICE, "librccl.so could not be loaded");
    RcclUniqueId id;
    const int e = r.GetUniqueId(&id);
    if (e != 0) return rccl_fail("ncclGetUniqueId", e);
    static_assert(sizeof(id) == LC_COMM_ID_BYTES, "unique id size");
    std::memcpy(out_id, &id, sizeof(id));
    return LC_OK;
    });
}

lc_status lc_comm_init(lc_ctx* ctx, int32_t rank, int32_t world, const uint8_t* id, lc_comm** out) {
    return guarded([&]() -> lc_status {
    if (!ctx || !id || !out) return fail(LC_ERR_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(LC_ERR_INVALID, "rank / world out of range");
    std::unique_ptr<lc_comm> c(new lc_comm());
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    if (ctx->device < 0 || ctx->comm_shared_memory.load()) {
        if (world > kMaxShmRanks) return fail(LC_ERR_INVALID, "the shared-memory backend takes at most 64 ranks");
        c->device_ptrs = ctx->device >= 0;
        char name[LC_COMM_ID_BYTES + 1];
        std::memcpy(name, id, LC_COMM_ID_BYTES);
        name[LC_COMM_ID_BYTES] = 0;
        if (name[0] != '/') return fail(LC_ERR_INVALID, "not a shared-memory communicator id");
        c->shm_name = name;
        c->shm_bytes = sizeof(ShmHeader) + kShmData;
        const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
        if (fd < 0) return fail(LC_ERR_DEVICE, std::string("shm_open: ") + std::strerror(errno));
        if (ftruncate(fd, off_t(c->shm_bytes)) != 0) {
            close(fd);
            return fail(LC_ERR_DEVICE, std::string("ftruncate: ") + std::strerror(errno));
        }
        void* p = mmap(nullptr, c->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return fail(LC_ERR_DEVICE, std::string("mmap: ") + std::strerror(errno));
        c->shm = static_cast<ShmHeader*>(p);  // a fresh file is zero filled: valid initial state of the atomics
        c->shm_data = static_cast<uint8_t*>(p) + sizeof(ShmHeader);
        c->shm->world = uint32_t(world);
        c->shm->arrived.fetch_add(1, std::memory_order_acq_rel);
        if (!spin_until([&] { return c->shm->arrived.load(std::memory_order_acquire) >= uint32_t(world); })) {
            munmap(p, c->shm_bytes);
            shm_unlink(name);  // (whoever gives up removes the name: the ranks that did arrive hold their own mappings)
            return fail(LC_ERR_DEVICE, "shared-memory communicator: not every rank arrived");
        }
        *out = c.release();
        return LC_OK;
    }
    Rccl& r = rccl();
    if (!r.ok) return fail(LC_ERR_DEVICE, "librccl.so could not be loaded");
    LC_HIP(hipSetDevice(ctx->device));
    RcclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    const int e = r.CommInitRank(&c->rc, world, uid, rank);
    if (e != 0) return rccl_fail("ncclCommInitRank", e);
    *out = c.release();
    return LC_OK;
    });
}

void lc_comm_destroy(lc_comm* c) {
    if (!c) return;
    try {
        if (c->rc) (void)rccl().CommDestroy(c->rc);
        if (c->shm) {
            const bool clean = shm_barrier(c);  // (fails at once on a poisoned communicator: no 120 s wait for a dead peer)
            munmap(c->shm, c->shm_bytes);
            if (c->rank == 0 || !clean) shm_unlink(c->shm_name.c_str());
        }
        delete c;
    } catch (...) {
    }
}

int32_t lc_comm_rank(const lc_comm* c) { return c ? c->rank : -1; }
int32_t lc_comm_world(const lc_comm* c) { return c ? c->world : 0; }

lc_status lc_comm_allreduce_count(lc_comm* c, void* d_total, void* stream) {
    return guarded([&]() -> lc_status {
    if (!c || !d_total) return fail(LC_ERR_INVALID, "null argument");
    if (c->shm) {
        hipStream_t st = static_cast<hipStream_t>(stream);
        uint64_t v;
        if (c->device_ptrs) {
            LC_HIP(hipSetDevice(c->ctx->device));
            LC_HIP(hipMemcpyAsync(&v, d_total, 8, hipMemcpyDeviceToHost, st));
            LC_HIP(hipStreamSynchronize(st));
        } else {
            std::memcpy(&v, d_total, 8);
        }
        c->shm->slots[c->rank] = v;
        if (!shm_barrier(c)) return fail(LC_ERR_DEVICE, "shared-memory communicator: a rank did not arrive");
        uint64_t sum = 0;
        for (int r = 0; r < c->world; r++) sum += c->shm->slots[r];
        if (!shm_barrier(c)) return fail(LC_ERR_DEVICE, "shared-memory communicator: a rank did not arrive");  // (slots are reused)
        if (c->device_ptrs) {
            LC_HIP(hipMemcpyAsync(d_total, &sum, 8, hipMemcpyHostToDevice, st));
            LC_HIP(hipStreamSynchronize(st));
        } else {
            std::memcpy(d_total, &sum, 8);
        }
        return LC_OK;
    }
    if (c->world == 1) return LC_OK;
    LC_HIP(hipSetDevice(c->ctx->device));  // (the collective launches on the thread's current device)
    const int e = rccl().AllReduce(d_total, d_total, 1, kRcclUint64, kRcclSum, c->rc, static_cast<hipStream_t>(stream));
    if (e != 0) return rccl_fail("ncclAllReduce", e);
    return LC_OK;
    });
}

lc_status lc_comm_allgather_mask(lc_comm* c, const void* d_mask_local, uint64_t local_words, void* d_mask_all,
                                 const uint64_t* words_per_rank, void* stream) {
    return guarded([&]() -> lc_status {
    if (!c || !d_mask_all || !words_per_rank || (local_words && !d_mask_local)) return fail(LC_ERR_INVALID, "null argument");
    if (words_per_rank[c->rank] != local_words) return fail(LC_ERR_INVALID, "words_per_rank[rank] differs from local_words");
    uint64_t off = 0, my_off = 0, total = 0;
    for (int r = 0; r < c->world; r++) {
        if (r == c->rank) my_off = total;
        total += words_per_rank[r];
    }
    if (c->shm) {
        if (total * 8 > kShmData) return fail(LC_ERR_INVALID, "mask too large for the shared-memory test backend");
        hipStream_t sst = static_cast<hipStream_t>(stream);
        if (c->device_ptrs) {
            LC_HIP(hipSetDevice(c->ctx->device));
            if (local_words) LC_HIP(hipMemcpyAsync(c->shm_data + my_off * 8, d_mask_local, size_t(local_words) * 8, hipMemcpyDeviceToHost, sst));
            LC_HIP(hipStreamSynchronize(sst));
        } else {
            std::memcpy(c->shm_data + my_off * 8, d_mask_local, size_t(local_words) * 8);
        }
        if (!shm_barrier(c)) return fail(LC_ERR_DEVICE, "shared-memory communicator: a rank did not arrive");
        if (c->device_ptrs) {
            if (total) LC_HIP(hipMemcpyAsync(d_mask_all, c->shm_data, size_t(total) * 8, hipMemcpyHostToDevice, sst));
            LC_HIP(hipStreamSynchronize(sst));
        } else {
            std::memcpy(d_mask_all, c->shm_data, size_t(total) * 8);
        }
        if (!shm_barrier(c)) return fail(LC_ERR_DEVICE, "shared-memory communicator: a rank did not arrive");
        return LC_OK;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint64_t* all = static_cast<uint64_t*>(d_mask_all);
    LC_HIP(hipSetDevice(c->ctx->device));
    if (c->world == 1) {
        if (local_words && d_mask_all != d_mask_local)
            LC_HIP(hipMemcpyAsync(all, d_mask_local, size_t(local_words) * 8, hipMemcpyDeviceToDevice, st));
        return LC_OK;
    }
    // all-gather-v: rank r broadcasts its segment into everybody's buffer at r's offset (row-range shards are word
    // aligned per entry, so the segments concatenate without bit shifting)
    Rccl& r = rccl();
    int e = r.GroupStart();
    if (e != 0) return rccl_fail("ncclGroupStart", e);
    for (int root = 0; root < c->world && e == 0; root++) {
        const uint64_t n = words_per_rank[root];
        if (n) e = r.Broadcast(root == c->rank ? d_mask_local : all + off, all + off, size_t(n), kRcclUint64, root, c->rc, st);
        off += n;
    }
    const int e2 = r.GroupEnd();
    if (e != 0) return rccl_fail("ncclBroadcast", e);
    if (e2 != 0) return rccl_fail("ncclGroupEnd", e2);
    return LC_OK;
    });
}

}  // extern "C"
