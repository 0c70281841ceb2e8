// Runtime behind the C ABI (include/liquid_cache_amd.h): context, HBM arena, staging of Liquid IPC bytes,
// column scans and the per-entry drop-in calls.  Host logic only — all arithmetic on cached data happens in
// lc_kernels.hip.  There is deliberately NO CPU fallback: without a HIP device every call fails with LC_ERR_DEVICE.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include <hip/hip_runtime.h>

#include "lc_internal.hpp"

using namespace lc;

static void scan_destroy_now(lc_scan* s);  // lc_scan_destroy without the scan cache

namespace lc {

thread_local char g_last_error[512] = {0};

lc_status fail(lc_status st, const char* msg) noexcept {
    std::snprintf(g_last_error, sizeof(g_last_error), "%s", msg ? msg : "");
    return st;
}

// Caller holds s->mu.  See lc_scan::last_stream.
void scan_enter_stream(lc_scan* s, hipStream_t stream) {
    if (s->used && s->last_stream != stream) (void)hipStreamSynchronize(s->last_stream);
    s->last_stream = stream;
    s->used = true;
    if (std::find(s->streams_used.begin(), s->streams_used.end(), stream) == s->streams_used.end()) s->streams_used.push_back(stream);
}
void scan_note_stream(lc_scan* s, hipStream_t stream) {
    std::lock_guard<std::mutex> g(s->mu);
    if (std::find(s->streams_used.begin(), s->streams_used.end(), stream) == s->streams_used.end()) s->streams_used.push_back(stream);
}

}  // namespace lc

namespace lc {

// The calling thread works on the context's device from here on.  HIP's current device is per THREAD and defaults to device 0:
// a host that runs one context per GPU and calls from worker threads (the reference's tokio workers) would otherwise launch
// rank k's scans on device 0.  (A no-op for the device the thread is on already; null / host-only contexts are the callee's
// business.)
static inline void bind_device(const lc_ctx* ctx) {
    if (ctx && ctx->device >= 0) (void)hipSetDevice(ctx->device);
}

// ------------------------------------------------------------------ scratch pool
constexpr size_t kPoolMinClass = 4096, kPoolMaxClass = size_t(64) << 20, kPoolKeepPerClass = 64;
// blocks up to these classes are carved out of chunks (lc_ctx::pool_chunks); larger ones are allocations of their own
constexpr size_t kChunkBytes = size_t(128) << 20, kChunkMaxClass = size_t(16) << 20;
constexpr size_t kHostChunkBytes = size_t(8) << 20, kHostChunkMaxClass = size_t(2) << 20;
// pointers inside a chunk are never given back one by one
static bool in_chunks(const std::vector<void*>& chunks, size_t chunk_bytes, const void* p) {
    for (void* c : chunks)
        if (p >= c && p < static_cast<const uint8_t*>(c) + chunk_bytes) return true;
    return false;
}
// Caller holds ctx->pool_mu.  A block of `cls` bytes from the current chunk (a new chunk when it is full), or null.
static void* carve_device(lc_ctx* ctx, size_t cls) {
    if (cls > kChunkMaxClass) return nullptr;
    if (!ctx->pool_chunk_cur || size_t(ctx->pool_chunk_end - ctx->pool_chunk_cur) < cls) {
        void* c = ctx->pool_chunk_spare;  // allocated ahead, off every query's path
        ctx->pool_chunk_spare = nullptr;
        if (!c) {
            LC_PHASE("pool: new device chunk");
            if (hipMalloc(&c, kChunkBytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
        ctx->pool_chunks.push_back(c);
        ctx->pool_chunk_cur = static_cast<uint8_t*>(c);
        ctx->pool_chunk_end = ctx->pool_chunk_cur + kChunkBytes;
    }
    void* p = ctx->pool_chunk_cur;
    ctx->pool_chunk_cur += cls;
    // The chunk after this one is allocated by the builder thread as soon as a quarter of this one is left: a hipMalloc takes
    // 0.35 ms, and the allocation that crossed a chunk boundary used to be the records of a scan's first LIKE (0.09 -> 0.44 ms).
    if (size_t(ctx->pool_chunk_end - ctx->pool_chunk_cur) < kChunkBytes / 4 && !ctx->pool_chunk_spare && !ctx->pool_spare_requested &&
        ctx->device >= 0) {
        ctx->pool_spare_requested = true;
        try {
            (void)builder_submit(ctx, [ctx](hipStream_t) {
                void* c = nullptr;
                if (hipMalloc(&c, kChunkBytes) != hipSuccess) { (void)hipGetLastError(); c = nullptr; }
                std::lock_guard<std::mutex> g(ctx->pool_mu);
                ctx->pool_chunk_spare = c;
                ctx->pool_spare_requested = false;
            });
        } catch (...) {
            ctx->pool_spare_requested = false;
        }
    }
    return p;
}
static void* carve_host(lc_ctx* ctx, size_t cls) {
    if (cls > kHostChunkMaxClass) return nullptr;
    if (!ctx->hpool_chunk_cur || size_t(ctx->hpool_chunk_end - ctx->hpool_chunk_cur) < cls) {
        void* c = nullptr;
        LC_PHASE("pool: new pinned chunk");
        if (hipHostMalloc(&c, kHostChunkBytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        ctx->hpool_chunks.push_back(c);
        ctx->hpool_chunk_cur = static_cast<uint8_t*>(c);
        ctx->hpool_chunk_end = ctx->hpool_chunk_cur + kHostChunkBytes;
    }
    void* p = ctx->hpool_chunk_cur;
    ctx->hpool_chunk_cur += cls;
    return p;
}
void pool_prime(lc_ctx* ctx) {  // lc_ctx_create: the first chunks, off every query's path
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    void* d = carve_device(ctx, kPoolMinClass);
    void* h = carve_host(ctx, kPoolMinClass);
    if (d) ctx->pool_free[kPoolMinClass].push_back(d);
    if (h) ctx->hpool_free[kPoolMinClass].push_back(h);
}

// ---- per-thread streams
void ctx_register(lc_ctx* ctx);
void ctx_unregister(lc_ctx* ctx);
namespace {
std::mutex g_ctx_registry_mu;
std::unordered_map<uint64_t, lc_ctx*> g_ctx_registry;  // live contexts by uid
std::atomic<uint64_t> g_next_ctx_uid{0};
struct ThreadStreams {
    std::vector<std::pair<uint64_t, hipStream_t>> bound;  // (context uid, this thread's stream for it)
    ~ThreadStreams() {
        // the thread ends: its streams go back to their contexts' pools (a context that is gone destroyed them already)
        std::lock_guard<std::mutex> g(g_ctx_registry_mu);
        for (auto& b : bound) {
            auto it = g_ctx_registry.find(b.first);
            if (it == g_ctx_registry.end()) continue;
            std::lock_guard<std::mutex> g2(it->second->pool_mu);
            it->second->stream_pool.push_back(b.second);
        }
    }
};
thread_local ThreadStreams t_streams;
}  // namespace

void ctx_register(lc_ctx* ctx) {
    ctx->uid = ++g_next_ctx_uid;
    std::lock_guard<std::mutex> g(g_ctx_registry_mu);
    g_ctx_registry[ctx->uid] = ctx;
}
void ctx_unregister(lc_ctx* ctx) {
    std::lock_guard<std::mutex> g(g_ctx_registry_mu);
    g_ctx_registry.erase(ctx->uid);
}

// The calling thread's stream for this context (bound on its first call: from the pool of streams finished threads left
// behind, else created).  Never the null stream unless stream creation fails.
hipStream_t stream_acquire(lc_ctx* ctx) {
    for (auto& b : t_streams.bound)
        if (b.first == ctx->uid) return b.second;
    hipStream_t s = nullptr;
    {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        if (!ctx->stream_pool.empty()) {
            s = ctx->stream_pool.back();
            ctx->stream_pool.pop_back();
        }
    }
    if (!s) {
        // (the highest priority the device offers: the streams queries run on must not queue behind the builder's stream)
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) != hipSuccess) {
            (void)hipGetLastError();
            s = nullptr;
        }
        if (!s && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;  // null stream as a fallback
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        ctx->all_streams.push_back(s);
    }
    t_streams.bound.emplace_back(ctx->uid, s);
    return s;
}
void stream_release(lc_ctx*, hipStream_t) {}  // (the stream stays bound to the thread)

// ---- the builder: one worker thread per context with a stream of its own at the lowest priority the device offers.  It runs
// what a query should not wait for — the scan-level LIKE index of a scan (k_flat_build: ~4 ms and 2 GB per 100 M-row column)
// — the way the reference derives its prefilter outside the read path (at insert time, byte_view_array/conversions.rs:353-355).
// Started on first use, drained and joined by lc_ctx_destroy.
static void builder_main(lc_ctx* ctx) {
    (void)hipSetDevice(ctx->device);
    int lo = 0, hi = 0;
    hipStream_t st = nullptr;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess ||
        hipStreamCreateWithPriority(&st, hipStreamNonBlocking, lo) != hipSuccess) {
        (void)hipGetLastError();
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;
    }
    {
        std::lock_guard<std::mutex> g(ctx->builder_mu);
        ctx->builder_stream = st;
    }
    for (;;) {
        std::packaged_task<void()> job;
        {
            std::unique_lock<std::mutex> g(ctx->builder_mu);
            ctx->builder_cv.wait(g, [&] { return ctx->builder_stop || !ctx->builder_q.empty(); });
            if (ctx->builder_q.empty()) break;  // (stop is honoured once the queue is drained: waiters hold futures)
            job = std::move(ctx->builder_q.front());
            ctx->builder_q.pop_front();
        }
        job();
    }
    if (st) (void)hipStreamDestroy(st);
}
std::future<void> builder_submit(lc_ctx* ctx, std::function<void(hipStream_t)> fn) {
    std::packaged_task<void()> job([ctx, fn]() {
        try {
            fn(ctx->builder_stream);
        } catch (...) {  // a failed build leaves the scan on the entry-level index
        }
    });
    std::future<void> fut = job.get_future();
    {
        std::lock_guard<std::mutex> g(ctx->builder_mu);
        if (!ctx->builder_started) {
            ctx->builder_started = true;
            ctx->builder = std::thread(builder_main, ctx);
        }
        ctx->builder_q.push_back(std::move(job));
    }
    ctx->builder_cv.notify_one();
    return fut;
}
void builder_shutdown(lc_ctx* ctx) {
    {
        std::lock_guard<std::mutex> g(ctx->builder_mu);
        if (!ctx->builder_started) return;
        ctx->builder_stop = true;
    }
    ctx->builder_cv.notify_all();
    if (ctx->builder.joinable()) ctx->builder.join();
    std::lock_guard<std::mutex> g(ctx->builder_mu);
    ctx->builder_started = false;
}

void* pool_alloc(lc_ctx* ctx, size_t bytes) {
    size_t cls = kPoolMinClass;
    while (cls < bytes) cls <<= 1;
    {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        auto it = ctx->pool_free.find(cls);
        if (it != ctx->pool_free.end() && !it->second.empty()) {
            void* p = it->second.back();
            it->second.pop_back();
            ctx->pool_live[p] = cls;
            return p;
        }
        if (void* p = carve_device(ctx, cls)) {
            ctx->pool_live[p] = cls;
            return p;
        }
    }
    void* p = nullptr;
    LC_PHASE("pool_alloc: hipMalloc");
#ifdef LC_TRACE_PHASES
    std::fprintf(stderr, "[pool] hipMalloc of class %zu for %zu bytes\n", cls, bytes);
#endif
    if (hipMalloc(&p, cls) != hipSuccess) {
        // the scan-level LIKE indexes kept for the NEXT scan over the same entries are a cache, outside the entry accounting:
        // they go before an allocation fails
        (void)hipGetLastError();
        like_orphans_clear(ctx);
        if (hipMalloc(&p, cls) != hipSuccess) return nullptr;
    }
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    ctx->pool_live[p] = cls;
    return p;
}

// true: `p` is a block of the pool and has been released; false: it is not the pool's (the caller frees it its own way)
bool pool_release_if_owned(lc_ctx* ctx, void* p) {
    if (!p) return true;
    {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        if (ctx->pool_live.find(p) == ctx->pool_live.end()) return false;
    }
    pool_release(ctx, p);
    return true;
}

// The caller guarantees that no kernel or copy still uses `p` (the per-call API synchronises before it returns).
void pool_release(lc_ctx* ctx, void* p) {
    if (!p) return;
    size_t cls = 0;
    {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        auto it = ctx->pool_live.find(p);
        if (it == ctx->pool_live.end()) return;
        cls = it->second;
        ctx->pool_live.erase(it);
        std::vector<void*>& fl = ctx->pool_free[cls];
        if ((cls <= kPoolMaxClass && fl.size() < kPoolKeepPerClass) || in_chunks(ctx->pool_chunks, kChunkBytes, p)) {
            fl.push_back(p);
            return;
        }
    }
    LC_PHASE("pool_release: hipFree");
    (void)hipFree(p);
}

void* host_pool_alloc(lc_ctx* ctx, size_t bytes) {
    size_t cls = kPoolMinClass;
    while (cls < bytes) cls <<= 1;
    {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        auto it = ctx->hpool_free.find(cls);
        if (it != ctx->hpool_free.end() && !it->second.empty()) {
            void* p = it->second.back();
            it->second.pop_back();
            ctx->hpool_live[p] = cls;
            return p;
        }
        if (void* p = carve_host(ctx, cls)) {
            ctx->hpool_live[p] = cls;
            return p;
        }
    }
    void* p = nullptr;
    LC_PHASE("host_pool_alloc: hipHostMalloc");
    if (hipHostMalloc(&p, cls, hipHostMallocDefault) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    ctx->hpool_live[p] = cls;
    return p;
}

void host_pool_release(lc_ctx* ctx, void* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        auto it = ctx->hpool_live.find(p);
        if (it == ctx->hpool_live.end()) return;
        const size_t cls = it->second;
        ctx->hpool_live.erase(it);
        std::vector<void*>& fl = ctx->hpool_free[cls];
        if ((cls <= kPoolMaxClass && fl.size() < kPoolKeepPerClass) || in_chunks(ctx->hpool_chunks, kHostChunkBytes, p)) {
            fl.push_back(p);
            return;
        }
    }
    LC_PHASE("host_pool_release: hipHostFree");
    (void)hipHostFree(p);
}

void pool_destroy(lc_ctx* ctx) {
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    for (auto& kv : ctx->pool_free)
        for (void* p : kv.second)
            if (!in_chunks(ctx->pool_chunks, kChunkBytes, p)) (void)hipFree(p);
    for (auto& kv : ctx->pool_live)
        if (!in_chunks(ctx->pool_chunks, kChunkBytes, kv.first)) (void)hipFree(kv.first);
    for (auto& kv : ctx->hpool_free)
        for (void* p : kv.second)
            if (!in_chunks(ctx->hpool_chunks, kHostChunkBytes, p)) (void)hipHostFree(p);
    for (auto& kv : ctx->hpool_live)
        if (!in_chunks(ctx->hpool_chunks, kHostChunkBytes, kv.first)) (void)hipHostFree(kv.first);
    for (void* c : ctx->pool_chunks) (void)hipFree(c);
    if (ctx->pool_chunk_spare) (void)hipFree(ctx->pool_chunk_spare);
    ctx->pool_chunk_spare = nullptr;
    for (void* c : ctx->hpool_chunks) (void)hipHostFree(c);
    ctx->pool_chunks.clear();
    ctx->hpool_chunks.clear();
    ctx->pool_chunk_cur = ctx->pool_chunk_end = ctx->hpool_chunk_cur = ctx->hpool_chunk_end = nullptr;
    ctx->pool_free.clear();
    ctx->pool_live.clear();
    ctx->hpool_free.clear();
    ctx->hpool_live.clear();
}

// ------------------------------------------------------------------ arena
// Bump allocation inside 256 MiB slabs.  A slab is returned to the device as soon as nothing references it any more:
// `live` counts its staged entries PLUS the scans that pinned one of its entries (lc_scan_create), which is what lets
// lc_evict / re-staging run under a live scan — the scan keeps reading the old blob, like a cloned Arc in the reference,
// and the slab goes away with the last of them.  `max_hbm_bytes` bounds the slab capacity that is reserved.
lc_status arena_alloc(lc_ctx* ctx, size_t bytes, uint8_t** out, int* slab_idx) {
    bytes = align_up(bytes, kSectionAlign);
    if (!ctx->slabs.empty()) {
        Slab& s = ctx->slabs.back();
        if (s.base && s.used + bytes <= s.size) {
            *out = s.base + s.used;
            s.used += bytes;
            s.live++;
            *slab_idx = int(ctx->slabs.size()) - 1;
            return LC_OK;
        }
    }
    size_t want = std::max(bytes, kSlabBytes);
    if (ctx->max_hbm) {
        // the budget covers the slabs AND the scan-level LIKE indexes; the indexes kept for future scans are a cache and go
        // first (live scans keep theirs: staging then fails like the reference's CacheFull)
        auto left_now = [&]() {
            const uint64_t used = ctx->staged_bytes + ctx->index_bytes.load();
            return ctx->max_hbm > used ? ctx->max_hbm - used : uint64_t(0);
        };
        uint64_t left = left_now();
        if (bytes > left) {
            like_orphans_clear(ctx);
            left = left_now();
        }
        if (bytes > left) return fail(LC_ERR_OOM, "HBM budget exhausted (max_hbm_bytes)");
        want = std::max<size_t>(bytes, std::min<uint64_t>(want, left));  // small budgets get small slabs
    }
    Slab s;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&s.base), want);
    if (e != hipSuccess) {
        // the indexes kept for future scans (like_pipeline_orphan) are a cache: staged data comes first
        (void)hipGetLastError();
        like_orphans_clear(ctx);
        e = hipMalloc(reinterpret_cast<void**>(&s.base), want);
    }
    if (e != hipSuccess) return fail(LC_ERR_OOM, std::string("hipMalloc slab: ") + hipGetErrorString(e));
    s.size = want;
    s.used = bytes;
    s.live = 1;
    // reuse the slot of a slab that was freed (indices held by entries / scans stay valid)
    size_t idx = ctx->slabs.size();
    ctx->slabs.push_back(s);
    ctx->staged_bytes += want;
    *out = s.base;
    *slab_idx = int(idx);
    return LC_OK;
}

void arena_pin(lc_ctx* ctx, int slab_idx) {
    if (slab_idx >= 0 && size_t(slab_idx) < ctx->slabs.size()) ctx->slabs[size_t(slab_idx)].live++;
}
// a scan's pins, slab by slab: (slab, number of the scan's entries in it)
static void arena_pin_counts(lc_ctx* ctx, const std::vector<std::pair<int, uint32_t>>& pins) {
    for (const auto& p : pins)
        if (p.first >= 0 && size_t(p.first) < ctx->slabs.size()) ctx->slabs[size_t(p.first)].live += p.second;
}

// Caller holds ctx->mu exclusively and guarantees that no kernel still reads the slab's blobs.
static void arena_release_n(lc_ctx* ctx, int slab_idx, uint32_t n) {
    if (slab_idx < 0 || size_t(slab_idx) >= ctx->slabs.size() || n == 0) return;
    Slab& s = ctx->slabs[size_t(slab_idx)];
    s.live -= n;
    if (s.live == 0 && s.base) {
        (void)hipFree(s.base);
        ctx->staged_bytes -= s.size;
        s.base = nullptr;
        s.size = s.used = 0;  // a drained LAST slab is not bumped into again: the next entry opens a new slab
    }
}
void arena_release(lc_ctx* ctx, int slab_idx) { arena_release_n(ctx, slab_idx, 1); }
static void arena_release_counts(lc_ctx* ctx, const std::vector<std::pair<int, uint32_t>>& pins) {
    for (const auto& p : pins) arena_release_n(ctx, p.first, p.second);
}

// Live counts reserved for entries that are not published yet (lc_stage, the device encoders): if the call fails between
// the reservation and the publication the counts are given back, so the slab can still drain.  Declare it BEFORE any
// lock on ctx->mu that the function holds while it may fail (the destructor takes the lock itself).
struct ArenaReservation {
    lc_ctx* ctx;
    int slab = -1;
    int64_t count = 0;
    explicit ArenaReservation(lc_ctx* c) : ctx(c) {}
    void arm(int slab_idx, int64_t n) { slab = slab_idx; count = n; }
    void disarm() { count = 0; }
    ~ArenaReservation() {
        if (count <= 0) return;
        (void)hipDeviceSynchronize();  // nothing may still write into the blob when the slab is returned
        std::unique_lock<std::shared_mutex> g(ctx->mu);
        for (int64_t i = 0; i < count; i++) arena_release(ctx, slab);
    }
    ArenaReservation(const ArenaReservation&) = delete;
    ArenaReservation& operator=(const ArenaReservation&) = delete;
};

// ---- idle one-entry scans of the per-entry drop-in calls (lc_ctx::scan_cache)
// Caller holds ctx->mu exclusively: the entry `id` is being replaced or evicted — its idle scans (which pin the old blob)
// move to the graveyard; they are destroyed outside the lock (scan_cache_reap).
static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
constexpr uint32_t kIdBloomBits = 1u << 17;
static inline void id_bloom_add(std::vector<uint64_t>& b, uint64_t id) {
    const uint64_t h = mix64(id);
    const uint32_t a = uint32_t(h) & (kIdBloomBits - 1u), c = uint32_t(h >> 32) & (kIdBloomBits - 1u);
    b[a >> 6] |= uint64_t(1) << (a & 63u);
    b[c >> 6] |= uint64_t(1) << (c & 63u);
}
static inline bool id_bloom_has(const std::vector<uint64_t>& b, uint64_t id) {
    if (b.empty()) return true;
    const uint64_t h = mix64(id);
    const uint32_t a = uint32_t(h) & (kIdBloomBits - 1u), c = uint32_t(h >> 32) & (kIdBloomBits - 1u);
    return ((b[a >> 6] >> (a & 63u)) & (b[c >> 6] >> (c & 63u)) & 1u) != 0;
}
static void scan_cache_invalidate_locked(lc_ctx* ctx, uint64_t id) {
    ctx->evict_epoch++;  // (scans in callers' hands notice when they are given back: lc_scan_destroy)
    std::lock_guard<std::mutex> g(ctx->scan_cache_mu);
    // whole scans kept for the next lc_scan_create over their id list: the ones that (may) hold `id` pin its old blob
    for (size_t i = ctx->list_cache.size(); i-- > 0;) {
        if (!id_bloom_has(ctx->list_cache[i]->id_bloom, id)) continue;
        ctx->scan_graveyard.push_back(ctx->list_cache[i]);
        ctx->list_cache.erase(ctx->list_cache.begin() + long(i));
    }
    auto it = ctx->scan_cache.find(id);
    if (it == ctx->scan_cache.end()) return;
    for (lc_scan* s : it->second) ctx->scan_graveyard.push_back(s);
    ctx->scan_cache_size -= it->second.size();
    ctx->scan_cache.erase(it);
}
// Caller holds NO lock of the context.
static void scan_cache_reap(lc_ctx* ctx) {
    std::vector<lc_scan*> dead;
    {
        std::lock_guard<std::mutex> g(ctx->scan_cache_mu);
        dead.swap(ctx->scan_graveyard);
    }
    for (lc_scan* s : dead) scan_destroy_now(s);
}
// Caller holds ctx->mu exclusively.  Publishes `e` under `id` (replacing what was there: scans that pinned the old blob
// keep it alive) with a fresh uid.
static void publish_entry(lc_ctx* ctx, uint64_t id, Entry&& e) {
    auto old = ctx->entries.find(id);
    if (old != ctx->entries.end()) {
        ctx->entry_bytes -= old->second.device_bytes;
        arena_release(ctx, old->second.slab);
        ctx->entries.erase(old);
        scan_cache_invalidate_locked(ctx, id);
    }
    e.uid = ++ctx->next_uid;
    ctx->entry_bytes += e.device_bytes;
    ctx->entries.emplace(id, std::move(e));
}

// Keeps the slab of an entry alive while a call reads the entry's blob outside the cache lock (read-backs, squeezes).
struct SlabPin {
    lc_ctx* ctx;
    int slab;
    SlabPin(lc_ctx* c, int s) : ctx(c), slab(s) {}  // the caller holds ctx->mu (shared is enough for the lookup, the pin
                                                      // itself is taken under the unique lock: see pin_entry)
    ~SlabPin() {
        if (slab < 0) return;
        std::unique_lock<std::shared_mutex> g(ctx->mu);
        arena_release(ctx, slab);
    }
    SlabPin(const SlabPin&) = delete;
    SlabPin& operator=(const SlabPin&) = delete;
};

// Copy of a staged entry with its slab pinned (unique lock: `live` is a plain counter).  slab = -1 in `*pin_slab` when
// the entry is absent.
bool pin_entry(lc_ctx* ctx, uint64_t entry_id, Entry* out, int* pin_slab) {
    std::unique_lock<std::shared_mutex> g(ctx->mu);
    auto it = ctx->entries.find(entry_id);
    if (it == ctx->entries.end()) { *pin_slab = -1; return false; }
    *out = it->second;
    arena_pin(ctx, out->slab);
    *pin_slab = out->slab;
    return true;
}

// ------------------------------------------------------------------ symbol tables
struct CtxSymtabs : SymtabProvider {
    lc_ctx* ctx;
    explicit CtxSymtabs(lc_ctx* c) : ctx(c) {}
    const SymbolTable* find(uint64_t path_id) override {
        std::lock_guard<std::mutex> g(ctx->st_mu);
        auto it = ctx->symtab_slot.find(path_id);
        return it == ctx->symtab_slot.end() ? nullptr : ctx->symtabs[it->second].get();
    }
    const SymbolTable* insert(uint64_t path_id, const SymbolTable& st) override {
        std::lock_guard<std::mutex> g(ctx->st_mu);
        auto it = ctx->symtab_slot.find(path_id);
        if (it != ctx->symtab_slot.end()) return ctx->symtabs[it->second].get();
        ctx->symtab_slot[path_id] = uint32_t(ctx->symtabs.size());
        ctx->symtabs.emplace_back(new SymbolTable(st));
        return ctx->symtabs.back().get();
    }
};

// make sure every registered symbol table is resident on the device
lc_status sync_symtabs(lc_ctx* ctx) {
    std::lock_guard<std::mutex> g(ctx->st_mu);
    const size_t n = ctx->symtabs.size();
    if (n == ctx->d_symtabs_uploaded) return LC_OK;
    if (n > ctx->d_symtabs_cap) {
        const size_t cap = std::max<size_t>(1024, n * 2);
        DevSymtab* fresh = nullptr;
        LC_HIP(hipMalloc(reinterpret_cast<void**>(&fresh), cap * sizeof(DevSymtab)));
        if (ctx->d_symtabs) {
            // uploads are synchronous copies, so the old array is complete; it is retired, not freed (see lc_ctx)
            LC_HIP(hipMemcpy(fresh, ctx->d_symtabs, ctx->d_symtabs_uploaded * sizeof(DevSymtab),
                             hipMemcpyDeviceToDevice));
            ctx->d_symtabs_retired.push_back(ctx->d_symtabs);
        }
        ctx->d_symtabs = fresh;
        ctx->d_symtabs_cap = cap;
    }
    std::vector<DevSymtab> host(n - ctx->d_symtabs_uploaded);
    for (size_t i = ctx->d_symtabs_uploaded; i < n; i++) {
        DevSymtab& d = host[i - ctx->d_symtabs_uploaded];
        const SymbolTable& s = *ctx->symtabs[i];
        for (int c = 0; c < 256; c++) {
            d.sym[c] = c < s.n ? s.sym[c] : 0;
            d.len[c] = c < s.n ? s.len[c] : 0;
        }
    }
    LC_HIP(hipMemcpy(ctx->d_symtabs + ctx->d_symtabs_uploaded, host.data(), host.size() * sizeof(DevSymtab),
                     hipMemcpyHostToDevice));
    ctx->d_symtabs_uploaded = n;
    return LC_OK;
}

// ------------------------------------------------------------------ staging
struct Blob {
    std::vector<uint8_t> bytes;
    size_t add(const void* src, size_t n, size_t pad_to = kSectionAlign, size_t extra_zero = 0) {
        const size_t off = align_up(bytes.size(), pad_to);
        bytes.resize(off + n + extra_zero, 0);
        if (n && src) std::memcpy(bytes.data() + off, src, n);
        return off;
    }
};

// validity bitmap -> u64 words (zero padded)
size_t add_validity(Blob& b, const uint8_t* validity, size_t nbits) {
    const size_t words = (nbits + 63) / 64;
    const size_t off = b.add(nullptr, 0);
    b.bytes.resize(off + words * 8, 0);
    std::memcpy(b.bytes.data() + off, validity, bitmap_bytes(nbits));
    if (nbits & 7) b.bytes[off + bitmap_bytes(nbits) - 1] &= uint8_t((1u << (nbits & 7)) - 1);
    return off;
}

lc_status build_fixed(const uint8_t* bytes, size_t len, Entry* e, Blob* blob, size_t* off_packed, size_t* off_valid,
                      size_t* off_pidx, size_t* off_pval) {
    FixedView v;
    if (!parse_fixed(bytes, len, &v)) return fail(LC_ERR_CORRUPT, "malformed Liquid fixed-width array");
    e->is_str = false;
    e->logical = v.logical;
    e->phys = v.phys;
    e->len = v.bp.len;
    e->nullable = v.bp.has_nulls || v.bp.all_null;
    e->all_null = v.bp.all_null;
    e->W = v.bp.all_null ? 0 : v.bp.bit_width;
    e->dec_is256 = v.dec_is256;
    e->dec_precision = v.dec_precision;
    e->dec_scale = v.dec_scale;
    FixedDesc& d = e->fd;
    d = FixedDesc{};
    d.len = e->len;
    d.W = uint8_t(e->W);
    d.lane_log2 = uint8_t(v.lane_bits == 8 ? 3 : v.lane_bits == 16 ? 4 : v.lane_bits == 32 ? 5 : 6);
    d.value_width = uint8_t(v.value_width);
    if (v.logical == kInteger) {
        d.kind = kKindInt;
        d.is_signed = phys_unsigned(v.phys) ? 0 : 1;
        uint64_t r = v.reference;
        if (d.is_signed) {  // sign-extend the native-width reference
            const int w = phys_width(v.phys);
            if (w == 1) r = uint64_t(int64_t(int8_t(r)));
            else if (w == 2) r = uint64_t(int64_t(int16_t(r)));
            else if (w == 4) r = uint64_t(int64_t(int32_t(r)));
        }
        d.reference = r;
    } else if (v.logical == kDecimal) {
        d.kind = kKindDecimal;
        d.is_signed = 0;
        d.reference = v.reference;
    } else {
        d.kind = v.phys == kF32 ? kKindF32 : kKindF64;
        d.is_signed = 1;
        d.reference = v.phys == kF32 ? uint64_t(int64_t(int32_t(uint32_t(v.reference)))) : v.reference;
        d.alp_e = uint8_t(v.alp_e);
        d.alp_f = uint8_t(v.alp_f);
        d.patch_len = uint32_t(v.patch_len);
    }
    *off_packed = *off_valid = *off_pidx = *off_pval = size_t(-1);
    if (!e->all_null) {
        // +128 zero bytes so 16-byte loads of the last block never leave the blob
        *off_packed = blob->add(v.bp.values, packed_bytes(e->W, e->len), kSectionAlign, 128);
        if (v.bp.has_nulls) *off_valid = add_validity(*blob, v.bp.nulls, e->len);
        if (v.patch_len) {
            *off_pidx = blob->add(v.patch_indices, size_t(v.patch_len) * 8);
            *off_pval = blob->add(v.patch_values, size_t(v.patch_len) * size_t(v.value_width));
        }
    }
    return LC_OK;
}

// Serialised acceleration index of a byte-view entry ("LCIX", lc_entry_index_to_bytes): what lc_stage builds for
// substring-search columns — the bit-sliced bigram signatures and the inverted row lists — so that the disk tier can keep
// it beside the Liquid bytes and a re-stage does not rebuild it.
struct IndexHeader {
    uint32_t magic, version, d, n, sig_bits, flags;  // flags: 1 = signatures present, 2 = row lists present
    uint64_t sig_bytes, post_bytes;
    uint64_t content_hash;  // of the Liquid bytes the index was derived from (index_content_hash): a blob kept for another
                            // version of the entry — same dictionary size and row count, other strings — is not taken
};
constexpr uint32_t kIndexMagic = 0x5849434Cu;  // "LCIX"
constexpr uint32_t kIndexVersion = 2;
static_assert(sizeof(IndexHeader) == 48, "IndexHeader layout");

// FNV-1a over the sections an index depends on: the dictionary keys of the rows and their validity words (row lists), the
// offset residuals + line parameters and the FSST bytes (which dictionary value is which string) and the symbol table
// (what the FSST bytes decode to).
static uint64_t fnv1a(uint64_t h, const void* p, size_t n) {
    const uint8_t* b = static_cast<const uint8_t*>(p);
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
static uint64_t index_content_hash(const void* keys, size_t key_bytes, const void* validity_words, size_t validity_bytes,
                                   const void* residuals, size_t residual_bytes, const void* fsst, size_t fsst_bytes,
                                   int32_t slope, int32_t intercept, const SymbolTable& st) {
    uint64_t h = 1469598103934665603ull;
    h = fnv1a(h, keys, key_bytes);
    h = fnv1a(h, &validity_bytes, sizeof(validity_bytes));
    if (validity_bytes) h = fnv1a(h, validity_words, validity_bytes);
    h = fnv1a(h, residuals, residual_bytes);
    h = fnv1a(h, &slope, 4);
    h = fnv1a(h, &intercept, 4);
    h = fnv1a(h, fsst, fsst_bytes);
    h = fnv1a(h, &st.n, sizeof(st.n));
    h = fnv1a(h, st.len, sizeof(st.len));
    h = fnv1a(h, st.sym, sizeof(st.sym));
    return h ? h : 1;  // (0 = "not computed")
}

lc_status build_str(lc_ctx* ctx, const uint8_t* bytes, size_t len, uint64_t path_id, Entry* e, Blob* blob,
                    size_t offs[9], const uint8_t* index = nullptr, size_t index_len = 0) {
    ByteViewParsed v;
    if (!parse_byte_view(bytes, len, &v)) return fail(LC_ERR_CORRUPT, "malformed Liquid byte-view array");
    // (any number of rows: the dictionary keys are u16, so an entry has at most 65,536 distinct values, but the reference's
    // batch size is the caller's choice, builders.rs:68-71.  Entries of more than 65,535 rows carry no inverted row lists —
    // those address rows with 16 bits — and are evaluated by k_str_pred's key mapping.)
    uint32_t slot;
    const SymbolTable* host_st = nullptr;
    {
        std::lock_guard<std::mutex> g(ctx->st_mu);
        auto it = ctx->symtab_slot.find(path_id);
        if (it == ctx->symtab_slot.end())
            return fail(LC_ERR_NO_SYMTAB, "byte-view entry staged before lc_symtab_set for its path");
        slot = it->second;
        host_st = ctx->symtabs[slot].get();  // unique_ptr targets are stable: the checks below run WITHOUT the lock (they are
                                             // O(compressed bytes), and parallel staging threads used to queue on it)
    }
    {
        const SymbolTable& st = *host_st;
        // every code in the compressed bytes must exist in the table (bytes between the dictionary values included)
        for (uint32_t i = 0; i < v.fsst_len; i++) {
            const uint8_t c = v.fsst[i];
            if (c == kFsstEscape) { i++; continue; }
            if (c >= st.n) return fail(LC_ERR_CORRUPT, "FSST code outside the registered symbol table");
        }
        // The gather sizing passes and the `=` length test take a value's length from byte 7 of its prefix key
        // (PrefixKey::from_parts, raw/fsst_buffer.rs:162-188: the length behind the shared prefix, 255 = that or more) while
        // the decode writes whatever the codes decode to: a key that disagrees with its value would make the decode overrun the
        // slot sized for it.  The reference trusts its own serialisation; bytes from anywhere else are checked here, once.
        for (uint32_t k = 0; k < v.d; k++) {
            const uint32_t a = v.offset_at(k), b = v.offset_at(k + 1);
            if (a > b || b > v.fsst_len) return fail(LC_ERR_CORRUPT, "compact offsets outside the FSST buffer");
            uint64_t dl = 0;
            for (uint32_t i = a; i < b; i++) {
                const uint8_t c = v.fsst[i];
                if (c == kFsstEscape) { if (++i < b) dl++; }  // (a dangling escape marker decodes to nothing)
                else dl += st.len[c];
            }
            if (dl < v.shared_prefix_len) return fail(LC_ERR_CORRUPT, "dictionary value shorter than the shared prefix");
            const uint64_t rest = dl - v.shared_prefix_len;
            if (v.prefix_keys[size_t(k) * 8 + 7] != uint8_t(std::min<uint64_t>(rest, 255)))
                return fail(LC_ERR_CORRUPT, "prefix key length byte disagrees with the FSST-compressed value");
        }
    }
    e->is_str = true;
    e->logical = kByteView;
    e->phys = v.arrow_type;
    e->len = v.n;
    e->nullable = v.nullable;
    e->all_null = v.all_null;
    e->W = 16;
    e->path_id = path_id;
    e->dict_len = v.d;
    e->has_fp = v.fingerprints != nullptr;
    e->offsets_bytes = v.residual_count * uint32_t(v.offset_bytes);
    e->fsst_len = v.fsst_len;
    e->raw_bytes = v.uncompressed_bytes;
    StrDesc& d = e->sd;
    d = StrDesc{};
    d.n = v.n;
    d.d = v.d;
    d.slope = v.slope;
    d.intercept = v.intercept;
    d.offset_bytes = uint8_t(v.offset_bytes);
    d.fsst_len = v.fsst_len;
    d.shared_prefix_len = v.shared_prefix_len;
    d.symtab_slot = slot;
    {
        uint32_t empties = 0;
        for (uint32_t i = 0; i < v.d && empties < 2; i++) empties += v.offset_at(i + 1) == v.offset_at(i) ? 1u : 0u;
        d.multi_empty = empties >= 2 ? 1 : 0;
    }
    for (int i = 0; i < 9; i++) offs[i] = size_t(-1);
    // keys padded to a multiple of 8 (16-byte loads)
    offs[0] = blob->add(v.keys.data(), size_t(v.n) * 2, kSectionAlign, 16);
    if (v.nullable) {
        if (v.all_null) {
            std::vector<uint8_t> zeros(bitmap_bytes(v.n) + 8, 0);
            offs[1] = add_validity(*blob, zeros.data(), v.n);
        } else {
            offs[1] = add_validity(*blob, v.key_validity, v.n);
        }
    }
    offs[2] = blob->add(v.prefix_keys, size_t(v.d) * 8);
    if (v.fingerprints) offs[3] = blob->add(v.fingerprints, size_t(v.d) * 4);
    offs[4] = blob->add(v.residuals, size_t(v.residual_count) * size_t(v.offset_bytes), kSectionAlign, 8);
    offs[5] = blob->add(v.fsst, v.fsst_len, kSectionAlign, 16);
    offs[6] = blob->add(v.shared_prefix, v.shared_prefix_len, kSectionAlign, 8);
    // the content hash validates a prebuilt index blob: computed only when one is supplied (a byte-serial hash over every
    // staged byte view cost seconds of host time per 100 M-row column); lc_entry_index_to_bytes computes it on demand
    e->index_hash = 0;
    if (index && index_len >= sizeof(IndexHeader))
        e->index_hash = index_content_hash(blob->bytes.data() + offs[0], size_t(v.n) * 2,
                                           offs[1] == size_t(-1) ? nullptr : blob->bytes.data() + offs[1],
                                           offs[1] == size_t(-1) ? 0 : ((size_t(v.n) + 63) / 64) * 8,
                                           blob->bytes.data() + offs[4], size_t(v.residual_count) * size_t(v.offset_bytes),
                                           blob->bytes.data() + offs[5], v.fsst_len, v.slope, v.intercept, *host_st);
    // a prebuilt index is used when it describes exactly this entry — dictionary size, rows, signature width, section
    // sizes AND the hash of the bytes it was derived from; anything else is ignored and the index is rebuilt: a stale or
    // foreign blob can cost time, never a result
    const uint8_t* pre_sig = nullptr;
    const uint8_t* pre_post = nullptr;
    size_t pre_post_bytes = 0;
    if (index && index_len >= sizeof(IndexHeader)) {
        IndexHeader h;
        std::memcpy(&h, index, sizeof(h));
        const size_t nw = std::max<size_t>((size_t(v.d) + 63) / 64, 1);
        // (section sizes checked one by one: their sum may wrap)
        const size_t body = index_len - sizeof(h);
        const bool sizes_ok = h.sig_bytes <= body && h.post_bytes == body - h.sig_bytes &&
                              ((h.flags & 1u) || h.sig_bytes == 0) && ((h.flags & 2u) || h.post_bytes == 0) && (h.flags & ~3u) == 0;
        const bool head_ok = h.magic == kIndexMagic && h.version == kIndexVersion && h.d == v.d && h.n == v.n &&
                             h.sig_bits == uint32_t(kSigBits) && sizes_ok && h.content_hash == e->index_hash;
        if (head_ok && (h.flags & 1u) && h.sig_bytes == size_t(kSigBits) * nw * 8) {
            pre_sig = index + sizeof(h);
            // no slice may have a bit beyond the dictionary: the kernels turn set bits into dictionary keys
            if (v.d & 63u) {
                const uint64_t beyond = ~((uint64_t(1) << (v.d & 63u)) - 1);
                for (int b = 0; b < kSigBits && pre_sig; b++) {
                    uint64_t last;
                    std::memcpy(&last, pre_sig + (size_t(b) * nw + (nw - 1)) * 8, 8);
                    if (last & beyond) pre_sig = nullptr;
                }
            }
            if (v.d == 0) pre_sig = nullptr;
        }
        if (head_ok && (h.flags & 2u) && h.post_bytes == (size_t(v.d) + 1 + size_t(v.n) + 32) * 2) {
            pre_post = index + sizeof(h) + h.sig_bytes;
            pre_post_bytes = h.post_bytes;
            // the list bounds must be a monotone partition of at most n rows, the rows below n (the kernels index with them)
            const uint16_t* po = reinterpret_cast<const uint16_t*>(pre_post);
            bool ok = po[0] == 0 && po[v.d] <= v.n;
            for (uint32_t k = 0; k < v.d && ok; k++) ok = po[k] <= po[k + 1];
            for (uint32_t r = 0; ok && r < po[v.d]; r++) ok = po[v.d + 1 + r] < v.n;
            if (!ok) { pre_post = nullptr; pre_post_bytes = 0; }
        }
    }
    if (v.fingerprints && ctx->build_signatures && pre_sig) {
        const size_t nw = std::max<size_t>((size_t(v.d) + 63) / 64, 1);
        offs[7] = blob->add(pre_sig, size_t(kSigBits) * nw * 8);
    } else if (v.fingerprints && ctx->build_signatures && !ctx->signatures_on_host) {
        // substring-search columns: room for the bit-sliced bigram signatures (lc_kernels.hpp); k_str_build_signatures
        // fills it once the blob is in HBM, before the entry becomes visible
        const size_t nw = (size_t(v.d) + 63) / 64;
        offs[7] = blob->add(nullptr, 0);
        blob->bytes.resize(offs[7] + size_t(kSigBits) * std::max<size_t>(nw, 1) * 8, 0);
        e->sig_on_device = true;
    } else if (v.fingerprints && ctx->build_signatures) {
        // LC_OPT_HOST_BUILT_INDEX (tests): the same index built by the host
        const size_t nw = (size_t(v.d) + 63) / 64;
        std::vector<uint64_t> sig(size_t(kSigBits) * std::max<size_t>(nw, 1), 0);
        std::vector<uint8_t> tmp;
        for (uint32_t i = 0; i < v.d; i++) {
            const uint32_t a = v.offset_at(i), b = v.offset_at(i + 1);
            tmp.resize(size_t(b - a) * 8 + 16);
            const size_t dl = fsst_decode(*host_st, v.fsst + a, b - a, tmp.data());
            for (size_t k = 0; k + 1 < dl; k++)
                sig[size_t(bigram_bit(tmp[k], tmp[k + 1])) * nw + (i >> 6)] |= uint64_t(1) << (i & 63);
        }
        offs[7] = blob->add(sig.data(), sig.size() * 8);
    }
    if (v.fingerprints && ctx->build_signatures && ctx->build_postings && v.n <= kPostMaxRows && v.d > 0 && !v.all_null && pre_post) {
        offs[8] = blob->add(pre_post, pre_post_bytes, kSectionAlign, 16);
    } else if (v.fingerprints && ctx->build_signatures && ctx->build_postings && v.n <= kPostMaxRows && v.d > 0 && !v.all_null) {
        const std::vector<uint16_t> post = build_row_lists(v.keys.data(), v.nullable ? v.key_validity : nullptr, v.n, v.d);
        offs[8] = blob->add(post.data(), post.size() * 2, kSectionAlign, 16);
    }
    return LC_OK;
}

uint64_t fixed_alg_bytes(const Entry& e, bool with_sel) {
    // n*W/8 packed + n/8 selection + n/8 validity (nullable) + n/8 output
    const uint64_t n = e.len, m = (n + 7) / 8;
    return n * uint64_t(e.W) / 8 + (with_sel ? m : 0) + (e.nullable ? m : 0) + m;
}

// ------------------------------------------------------------------ predicate normalisation
lc_status make_fixed_pred(const Entry& e, const lc_predicate* p, FixedPred* out) {
    if (p->op < LC_OP_EQ || p->op > LC_OP_GE) return fail(LC_UNSUPPORTED, "operator not supported on numeric columns");
    *out = FixedPred{};
    out->op = p->op;
#ifdef LC_ABLATION
    if (const char* dbg = std::getenv("LC_DEBUG_FLAGS")) out->debug_flags = uint32_t(std::atoi(dbg));  // profiling builds only
#endif
    if (!p->lit) return fail(LC_ERR_INVALID, "literal is null");
    if (e.fd.kind == kKindF32 || e.fd.kind == kKindF64) {
        // the literal arrives in the column's own type (DataFusion casts it); its bits travel to the kernel, which
        // compares in Arrow's totalOrder
        if (e.fd.kind == kKindF32 && p->lit_tag == LC_LIT_F32 && p->lit_len == 4) {
            uint32_t b;
            std::memcpy(&b, p->lit, 4);
            out->lit = b;
        } else if (e.fd.kind == kKindF64 && p->lit_tag == LC_LIT_F64 && p->lit_len == 8) {
            std::memcpy(&out->lit, p->lit, 8);
        } else {
            return fail(LC_ERR_INVALID, "float predicate needs a literal of the column's float type");
        }
        return LC_OK;
    }
    if (e.fd.kind == kKindDecimal) {
        __int128 v;
        if (p->lit_tag == LC_LIT_I128 && p->lit_len == 16) std::memcpy(&v, p->lit, 16);
        else if (p->lit_tag == LC_LIT_I64 && p->lit_len == 8) { int64_t t; std::memcpy(&t, p->lit, 8); v = t; }
        else return fail(LC_ERR_INVALID, "decimal predicate needs an i128/i64 literal");
        if (v < 0) out->lit_class = -1;
        else if (v > __int128(UINT64_MAX)) out->lit_class = 1;
        else out->lit = uint64_t(v);
        return LC_OK;
    }
    if (p->lit_len != 8 || (p->lit_tag != LC_LIT_I64 && p->lit_tag != LC_LIT_U64))
        return fail(LC_ERR_INVALID, "integer predicate needs an i64/u64 literal");
    uint64_t raw;
    std::memcpy(&raw, p->lit, 8);
    if (e.fd.is_signed) {
        if (p->lit_tag == LC_LIT_U64 && raw > uint64_t(INT64_MAX)) out->lit_class = 1;
        else out->lit = raw;
    } else {
        if (p->lit_tag == LC_LIT_I64 && int64_t(raw) < 0) out->lit_class = -1;
        else out->lit = raw;
    }
    return LC_OK;
}

lc_status make_str_pred(const lc_predicate* p, StrPredHost* out) {
    out->p = StrPred{};
    out->p.op = p->op;
    if (p->lit_tag == LC_LIT_BOOL) {  // liquid_expr.rs:78-80 + helpers.rs:72-79
        if (!p->lit || p->lit_len < 1) return fail(LC_ERR_INVALID, "boolean literal missing");
        out->p.mode = 2;
        out->p.const_value = static_cast<const uint8_t*>(p->lit)[0] ? 1 : 0;
        return LC_OK;
    }
    if (p->lit_tag != LC_LIT_BYTES) return fail(LC_UNSUPPORTED, "byte-view predicate needs a bytes literal");
    const uint8_t* lit = static_cast<const uint8_t*>(p->lit);
    const size_t ll = size_t(p->lit_len);
    if (p->op >= LC_OP_EQ && p->op <= LC_OP_GE) {
        if (ll > size_t(kMaxNeedleBytes)) return fail(LC_UNSUPPORTED, "needle longer than 4096 bytes");
        out->p.mode = 0;
        out->needle.assign(lit, lit + ll);
    } else if (p->op == LC_OP_LIKE || p->op == LC_OP_NOT_LIKE) {
        const uint8_t* inner;
        size_t il;
        if (ll > size_t(kMaxNeedleBytes)) return fail(LC_UNSUPPORTED, "LIKE pattern longer than 4096 bytes");
        if (!substring_pattern(lit, ll, &inner, &il)) {
            // general pattern (prefix / suffix / `_` / several parts): Arrow `like` on every dictionary value.  The reference
            // only gets here for entries WITHOUT fingerprints (with them it `expect()`s a %needle% pattern,
            // comparisons.rs:150-166); the caller checks that.
            out->p.mode = 3;
            out->needle.assign(lit, lit + ll);
#ifdef LC_ABLATION
            if (const char* dbg = std::getenv("LC_DEBUG_FLAGS")) out->p.debug_flags = std::atoi(dbg);
#endif
            out->p.needle_len = uint32_t(out->needle.size());
            if (out->needle.size() <= size_t(kInlineNeedle))
                std::memcpy(out->p.needle_inline, out->needle.data(), out->needle.size());
            return LC_OK;
        }
        out->p.mode = 1;
        out->p.use_fingerprints = 1;
        // needles the folded automaton cannot hold: it runs over their first kMaxNeedleAutomaton bytes and the (very few)
        // dictionary values it accepts are matched exactly against the pattern; fingerprint and bigram bits are those of the
        // whole needle (comparisons.rs:598-651 computes the same: prefilter, then memmem of the needle)
        out->needle.assign(inner, inner + std::min(il, size_t(kMaxNeedleAutomaton)));
        if (il > size_t(kMaxNeedleAutomaton)) {
            out->verify.assign(lit, lit + ll);
            out->p.verify_len = uint32_t(ll);
        }
        for (size_t k = 0; k < il; k++) out->p.needle_fp |= 1u << (inner[k] & 31);
        // bigram positions in an order that covers the needle evenly at every prefix: both ends, then midpoints of the
        // remaining gaps, breadth first
        std::vector<size_t> order;
        if (il >= 2) {
            const size_t last = il - 2;
            std::vector<uint8_t> seen(last + 1, 0);
            std::vector<std::pair<size_t, size_t>> gaps;
            auto take = [&](size_t k) { if (!seen[k]) { seen[k] = 1; order.push_back(k); } };
            take(0);
            take(last);
            gaps.push_back({0, last});
            for (size_t g = 0; g < gaps.size(); g++) {
                const size_t lo = gaps[g].first, hi = gaps[g].second;
                if (hi - lo < 2) continue;
                const size_t mid = lo + (hi - lo) / 2;
                take(mid);
                gaps.push_back({lo, mid});
                gaps.push_back({mid, hi});
            }
        }
        for (size_t k : order) {
            if (out->p.n_sig_wide >= uint32_t(kMaxSigProbeWide)) break;
            const uint16_t bit = uint16_t(bigram_bit(inner[k], inner[k + 1]));
            bool dup = false;
            for (uint32_t q = 0; q < out->p.n_sig_wide; q++) dup |= out->p.sig_wide[q] == bit;
            if (!dup) out->p.sig_wide[out->p.n_sig_wide++] = bit;
        }
        out->p.n_sig_bits = std::min<uint32_t>(out->p.n_sig_wide, uint32_t(kMaxSigProbe));
        for (uint32_t q = 0; q < out->p.n_sig_bits; q++) out->p.sig_bits[q] = out->p.sig_wide[q];
        // k_str_pred always ANDs kMaxSigProbe slices (all loads in flight together): pad with repeats
        for (uint32_t q = out->p.n_sig_bits; q < uint32_t(kMaxSigProbe) && out->p.n_sig_bits > 0; q++)
            out->p.sig_bits[q] = out->p.sig_bits[q % out->p.n_sig_bits];
    } else {
        return fail(LC_UNSUPPORTED, "operator not supported on byte-view columns");
    }
#ifdef LC_ABLATION
    if (const char* dbg = std::getenv("LC_DEBUG_FLAGS")) out->p.debug_flags = std::atoi(dbg);
#endif
    out->p.needle_len = uint32_t(out->needle.size());
    if (out->needle.size() <= size_t(kInlineNeedle))
        std::memcpy(out->p.needle_inline, out->needle.data(), out->needle.size());
    return LC_OK;
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

const char* lc_version(void) { return "liquid_cache_amd 0.1 (gfx950)"; }

const char* lc_last_error(lc_ctx*) { return g_last_error; }

lc_status lc_ctx_create(const int32_t* device_ids, int32_t n_devices, uint64_t max_hbm_bytes, lc_ctx** out) {
    return guarded([&]() -> lc_status {
    if (!out) return fail(LC_ERR_INVALID, "out is null");
    *out = nullptr;
    if (n_devices == 0 && !device_ids) {
        // host-only context: Arrow->Liquid transcoding and symbol tables only; every device call fails loudly
        std::unique_ptr<lc_ctx> host(new lc_ctx());
        host->device = -1;
        ctx_register(host.get());
        *out = host.release();
        return LC_OK;
    }
    if (n_devices != 1)
        return fail(LC_ERR_INVALID, "one lc_ctx drives one device: run one process (or ctx) per GPU");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(LC_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
    int dev = 0;
    if (device_ids) dev = device_ids[0];
    else LC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= count) return fail(LC_ERR_INVALID, "device id out of range");
    LC_HIP(hipSetDevice(dev));
    std::unique_ptr<lc_ctx> ctx(new lc_ctx());
    ctx->device = dev;
    LC_HIP(hipGetDeviceProperties(&ctx->props, dev));
    ctx->max_hbm = max_hbm_bytes;
    // the code objects of the kernels, now (best effort): the first LIKE of a process paid 2.3 ms for the load of the index
    // builder's unit in front of its first launch
    (void)warm_code_object_kernels();
    (void)warm_code_object_like_pipeline();
    (void)warm_code_object_like_scanall();
    (void)warm_code_object_groupby();
    (void)warm_code_object_bv_encode();
    (void)hipGetLastError();
    // (no environment variable is read here: what the library stages and how it evaluates is decided by the caller
    // through lc_ctx_set_option, never by the process environment)
    ctx_register(ctx.get());
    // the builder thread and its stream, now: creating a stream takes the runtime ~7 ms during which other threads' stream waits
    // do not return — a first query that started the builder lazily waited 6.3 ms for its 40 us kernel
    (void)builder_submit(ctx.get(), [](hipStream_t) {});
    pool_prime(ctx.get());
    plan_slots_prime(ctx.get());
    *out = ctx.release();
    return LC_OK;
    });
}

lc_status lc_ctx_set_option(lc_ctx* ctx, int32_t option, int64_t value) {
    return guarded([&]() -> lc_status {
    if (!ctx) return fail(LC_ERR_INVALID, "null argument");
    std::unique_lock<std::shared_mutex> g(ctx->mu);
    switch (option) {
        case LC_OPT_SIGNATURE_INDEX: ctx->build_signatures = value != 0; return LC_OK;
        case LC_OPT_ROW_LISTS: ctx->build_postings = value != 0; return LC_OK;
        case LC_OPT_HOST_BUILT_INDEX: ctx->signatures_on_host = value != 0; return LC_OK;
        case LC_OPT_LIKE_MANY_HINT: ctx->like_many_hint = value != 0; return LC_OK;
        case LC_OPT_LIKE_INDEX_BUDGET_BYTES: ctx->like_index_budget = value < 0 ? 0 : uint64_t(value); return LC_OK;
        case LC_OPT_LIKE_INDEX_CACHE: ctx->like_index_cache = uint32_t(std::max<int64_t>(0, std::min<int64_t>(value, 1024))); return LC_OK;
        case LC_OPT_LIKE_INDEX_ASYNC: ctx->like_index_async = value != 0; return LC_OK;
        case LC_OPT_COMM_SHARED_MEMORY: ctx->comm_shared_memory = value != 0; return LC_OK;
        case LC_OPT_SCAN_CACHE: {
            ctx->scan_cache_max = uint32_t(std::max<int64_t>(0, std::min<int64_t>(value, 1024)));
            std::lock_guard<std::mutex> g2(ctx->scan_cache_mu);  // what no longer fits goes (destroyed by the next reap)
            while (ctx->list_cache.size() > size_t(ctx->scan_cache_max.load())) {
                ctx->scan_graveyard.push_back(ctx->list_cache.front());
                ctx->list_cache.erase(ctx->list_cache.begin());
            }
            return LC_OK;
        }
        case LC_OPT_LIKE_PATH:
            if (value < 0 || value > 5) return fail(LC_ERR_INVALID, "LC_OPT_LIKE_PATH takes 0 .. 5");
            ctx->like_path = int(value);
            return LC_OK;
        case LC_OPT_LIKE_PIPELINE_MIN_ENTRIES:
            ctx->like_pipeline_min_entries = value < 0 ? 0xFFFFFFFFu : uint32_t(std::min<int64_t>(value, 0xFFFFFFFFll));
            return LC_OK;
        default: return fail(LC_ERR_INVALID, "unknown context option");
    }
    });
}

void lc_ctx_destroy(lc_ctx* ctx) {
    if (!ctx) return;
    try {
    if (ctx->device < 0) { ctx_unregister(ctx); delete ctx; return; }
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    {
        std::vector<lc_scan*> idle;
        {
            std::lock_guard<std::mutex> g(ctx->scan_cache_mu);
            for (auto& kv : ctx->scan_cache)
                for (lc_scan* sc : kv.second) idle.push_back(sc);
            ctx->scan_cache.clear();
            ctx->scan_cache_size = 0;
            for (lc_scan* sc : ctx->scan_graveyard) idle.push_back(sc);
            ctx->scan_graveyard.clear();
            for (lc_scan* sc : ctx->list_cache) idle.push_back(sc);
            ctx->list_cache.clear();
        }
        for (lc_scan* sc : idle) scan_destroy_now(sc);
    }
    like_orphans_clear(ctx);
    builder_shutdown(ctx);
    plan_slots_destroy(ctx);
    for (Slab& s : ctx->slabs)
        if (s.base) (void)hipFree(s.base);
    pool_destroy(ctx);
    ctx_unregister(ctx);
    for (hipStream_t st : ctx->all_streams) (void)hipStreamDestroy(st);
    if (ctx->d_symtabs) (void)hipFree(ctx->d_symtabs);
    for (DevSymtab* p : ctx->d_symtabs_retired) (void)hipFree(p);
    delete ctx;
    } catch (...) {
    }
}

lc_status lc_device_info_get(lc_ctx* ctx, lc_device_info* out) {
    return guarded([&]() -> lc_status {
    if (!ctx || !out) return fail(LC_ERR_INVALID, "null argument");
    std::memset(out, 0, sizeof(*out));
    out->device_id = ctx->device;
    out->compute_units = ctx->props.multiProcessorCount;
    out->hbm_total_bytes = ctx->props.totalGlobalMem;
    std::shared_lock<std::shared_mutex> g(ctx->mu);
    out->hbm_staged_bytes = ctx->entry_bytes;
    out->staged_entries = ctx->entries.size();
    std::snprintf(out->name, sizeof(out->name), "%s", ctx->props.name);
    std::snprintf(out->gcn_arch, sizeof(out->gcn_arch), "%s", ctx->props.gcnArchName);
    return LC_OK;
    });
}

lc_status lc_symtab_set(lc_ctx* ctx, uint64_t path_id, const uint8_t* bytes, size_t len) {
    return guarded([&]() -> lc_status {
    if (!ctx || !bytes) return fail(LC_ERR_INVALID, "null argument");
    SymbolTable st;
    if (!st.load(bytes, len)) return fail(LC_ERR_CORRUPT, "malformed symbol table");
    std::lock_guard<std::mutex> g(ctx->st_mu);
    if (ctx->symtab_slot.count(path_id)) return fail(LC_ERR_INVALID, "symbol table of this path is already set");
    ctx->symtab_slot[path_id] = uint32_t(ctx->symtabs.size());
    ctx->symtabs.emplace_back(new SymbolTable(st));
    return LC_OK;
    });
}

lc_status lc_symtab_get(lc_ctx* ctx, uint64_t path_id, uint8_t** out_bytes, size_t* out_len) {
    return guarded([&]() -> lc_status {
    if (!ctx || !out_bytes || !out_len) return fail(LC_ERR_INVALID, "null argument");
    CtxSymtabs s(ctx);
    const SymbolTable* st = s.find(path_id);
    if (!st) return fail(LC_NOT_STAGED, "no symbol table for this path");
    const std::vector<uint8_t> b = st->save();
    *out_bytes = static_cast<uint8_t*>(std::malloc(b.size() ? b.size() : 1));
    std::memcpy(*out_bytes, b.data(), b.size());
    *out_len = b.size();
    return LC_OK;
    });
}

void lc_free(void* p) { std::free(p); }

static lc_status stage_impl(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, const uint8_t* const* bytes, const size_t* lens,
                            const uint64_t* path_ids, const uint8_t* const* index_bytes, const size_t* index_lens);

lc_status lc_stage(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, const uint8_t* const* bytes,
                   const size_t* lens, const uint64_t* path_ids) {
    return stage_impl(ctx, n, entry_ids, bytes, lens, path_ids, nullptr, nullptr);
}

lc_status lc_stage_indexed(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, const uint8_t* const* bytes, const size_t* lens,
                           const uint64_t* path_ids, const uint8_t* const* index_bytes, const size_t* index_lens) {
    if (index_bytes && !index_lens) return fail(LC_ERR_INVALID, "index_lens is null");
    return stage_impl(ctx, n, entry_ids, bytes, lens, path_ids, index_bytes, index_lens);
}

static lc_status stage_impl(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, const uint8_t* const* bytes, const size_t* lens,
                            const uint64_t* path_ids, const uint8_t* const* index_bytes, const size_t* index_lens) {
    return guarded([&]() -> lc_status {
    if (!ctx || (n && (!entry_ids || !bytes || !lens))) return fail(LC_ERR_INVALID, "null argument");
    if (ctx->device < 0) return fail(LC_ERR_DEVICE, "host-only context: no HIP device (there is no CPU fallback)");
    LC_HIP(hipSetDevice(ctx->device));
    // group entries into upload batches of <= 64 MiB
    uint64_t i = 0;
    while (i < n) {
        Blob blob;
        struct Pending {
            uint64_t id;
            Entry e;
            size_t off[9];
            size_t blob_begin;
        };
        std::vector<Pending> pend;
        while (i < n && blob.bytes.size() < (size_t(64) << 20)) {
            Pending p;
            p.id = entry_ids[i];
            int logical = 0, phys = 0;
            if (!bytes[i] || !read_ipc_header(bytes[i], lens[i], &logical, &phys))
                return fail(LC_ERR_CORRUPT, "bad Liquid IPC header");
            p.blob_begin = align_up(blob.bytes.size(), kSectionAlign);
            blob.bytes.resize(p.blob_begin, 0);
            for (auto& o : p.off) o = size_t(-1);
            lc_status st;
            if (logical == kByteView) {
                st = build_str(ctx, bytes[i], lens[i], path_ids ? path_ids[i] : 0, &p.e, &blob, p.off,
                               index_bytes ? index_bytes[i] : nullptr, index_bytes && index_bytes[i] ? index_lens[i] : 0);
            } else if (logical == kInteger || logical == kDecimal || logical == kFloat) {
                st = build_fixed(bytes[i], lens[i], &p.e, &blob, &p.off[0], &p.off[1], &p.off[2], &p.off[3]);
            } else {
                return fail(LC_UNSUPPORTED, "Liquid logical type not handled on the device (FixedLen/Linear)");
            }
            if (st != LC_OK) return st;
            p.e.device_bytes = align_up(blob.bytes.size(), kSectionAlign) - p.blob_begin;
            pend.push_back(std::move(p));
            i++;
        }
        {
            const lc_status ss = sync_symtabs(ctx);  // the signature builder reads the device copies
            if (ss != LC_OK) return ss;
        }
        // The cache lock is held only to reserve the blob and, later, to publish the entries: the copy and the signature
        // kernel of concurrent lc_stage calls (one host thread per column chunk is the usual staging pattern) overlap.
        uint8_t* dbase = nullptr;
        int slab = -1;
        const size_t total = align_up(blob.bytes.size(), kSectionAlign) + 256;
        blob.bytes.resize(total, 0);
        ArenaReservation reserved(ctx);  // gives the counts back if anything below fails before the entries are published
        {
            std::unique_lock<std::shared_mutex> g(ctx->mu);
            lc_status st = arena_alloc(ctx, total, &dbase, &slab);
            if (st != LC_OK) return st;
            ctx->slabs[size_t(slab)].live += int64_t(pend.size()) - 1;  // keeps the slab alive until the entries exist
            reserved.arm(slab, int64_t(pend.size()));
        }
        LC_HIP(hipMemcpy(dbase, blob.bytes.data(), total, hipMemcpyHostToDevice));
        std::vector<StrDesc> sig_descs;
        for (Pending& p : pend) {
            auto ptr = [&](size_t off) -> uint8_t* { return off == size_t(-1) ? nullptr : dbase + off; };
            p.e.slab = slab;
            if (p.e.is_str) {
                StrDesc& d = p.e.sd;
                d.keys = reinterpret_cast<const uint16_t*>(ptr(p.off[0]));
                d.validity = reinterpret_cast<const uint64_t*>(ptr(p.off[1]));
                d.prefix_keys = ptr(p.off[2]);
                d.fingerprints = reinterpret_cast<const uint32_t*>(ptr(p.off[3]));
                d.residuals = ptr(p.off[4]);
                d.fsst = ptr(p.off[5]);
                d.shared_prefix = ptr(p.off[6]);
                d.signatures = reinterpret_cast<const uint64_t*>(ptr(p.off[7]));
                d.postings = reinterpret_cast<const uint16_t*>(ptr(p.off[8]));
                if (p.e.sig_on_device) sig_descs.push_back(d);
            } else {
                FixedDesc& d = p.e.fd;
                d.packed = ptr(p.off[0]);
                d.validity = reinterpret_cast<const uint64_t*>(ptr(p.off[1]));
                d.patch_idx = reinterpret_cast<const uint64_t*>(ptr(p.off[2]));
                d.patch_val = ptr(p.off[3]);
            }
        }
        if (!sig_descs.empty()) {
            // the signature slices are built before any entry of this chunk becomes visible to evaluations
            const DevSymtab* d_st;
            {
                std::lock_guard<std::mutex> sg(ctx->st_mu);
                d_st = ctx->d_symtabs;  // uploaded by the sync_symtabs above; retired arrays are never freed
            }
            StrDesc* d_sd = static_cast<StrDesc*>(pool_alloc(ctx, sig_descs.size() * sizeof(StrDesc)));
            if (!d_sd) return fail(LC_ERR_OOM, "hipMalloc (signature builder descriptors)");
            hipStream_t side = stream_acquire(ctx);  // (the blob copy above was a synchronous hipMemcpy: it has landed)
            const hipError_t e1 = hipMemcpyAsync(d_sd, sig_descs.data(), sig_descs.size() * sizeof(StrDesc), hipMemcpyHostToDevice, side);
            uint32_t max_d = 1;
            for (const StrDesc& sd : sig_descs) max_d = std::max(max_d, sd.d);
            const hipError_t e2 = e1 == hipSuccess ? launch_str_build_signatures(d_sd, uint32_t(sig_descs.size()), max_d, d_st, side) : e1;
            const hipError_t e3 = hipStreamSynchronize(side);
            stream_release(ctx, side);
            pool_release(ctx, d_sd);
            if (e2 != hipSuccess || e3 != hipSuccess) return fail(LC_ERR_DEVICE, "k_str_build_signatures failed");
        }
        std::unique_lock<std::shared_mutex> g(ctx->mu);
        for (Pending& p : pend) {
            publish_entry(ctx, p.id, std::move(p.e));
        }
        reserved.disarm();
    }
    return sync_symtabs(ctx);
    });
}

lc_status lc_evict(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids) {
    return guarded([&]() -> lc_status {
    if (!ctx || (n && !entry_ids)) return fail(LC_ERR_INVALID, "null argument");
    if (ctx->device < 0) return fail(LC_ERR_DEVICE, "host-only context: no HIP device (there is no CPU fallback)");
    LC_HIP(hipSetDevice(ctx->device));
    // No device-wide synchronise: whatever still reads a blob holds a pin on its slab (scans, SlabPin), and a slab is only
    // returned to the driver when its last pin goes.  Other threads' evaluations keep running.
    {
        std::unique_lock<std::shared_mutex> g(ctx->mu);
        for (uint64_t i = 0; i < n; i++) {
            auto it = ctx->entries.find(entry_ids[i]);
            if (it == ctx->entries.end()) continue;
            ctx->entry_bytes -= it->second.device_bytes;
            arena_release(ctx, it->second.slab);
            ctx->entries.erase(it);
            scan_cache_invalidate_locked(ctx, entry_ids[i]);
        }
    }
    scan_cache_reap(ctx);  // idle one-entry scans of the evicted entries drop their pins now
    return LC_OK;
    });
}

lc_status lc_entry_info_get(lc_ctx* ctx, uint64_t entry_id, lc_entry_info* out) {
    return guarded([&]() -> lc_status {
    if (!ctx || !out) return fail(LC_ERR_INVALID, "null argument");
    std::shared_lock<std::shared_mutex> g(ctx->mu);
    auto it = ctx->entries.find(entry_id);
    if (it == ctx->entries.end()) return LC_NOT_STAGED;
    const Entry& e = it->second;
    std::memset(out, 0, sizeof(*out));
    out->logical_type = e.logical;
    out->physical_type = e.phys;
    out->len = e.len;
    out->nullable = e.nullable;
    out->all_null = e.all_null;
    out->bit_width = e.all_null ? 0 : e.W;
    out->dict_len = e.dict_len;
    out->has_fingerprints = e.has_fp;
    out->device_bytes = e.device_bytes;
    out->squeezed_date_field = e.squeezed_field;
    out->clamped_from_bit_width = e.clamped ? e.orig_W : 0;
    out->quantized_from_bit_width = e.quantized ? e.orig_W : 0;
    out->quantized_bucket_width = e.quantized ? (e.fq_shift > 0 ? (uint64_t(1) << e.fq_shift) : e.bucket_width) : 0;
    out->algorithmic_pred_bytes = e.is_str ? 0 : fixed_alg_bytes(e, false);
    return LC_OK;
    });
}

lc_status lc_transcode_arrow(lc_ctx* ctx, const struct ArrowArray* array, const struct ArrowSchema* schema,
                             int32_t hint, uint64_t path_id, uint8_t** out_bytes, size_t* out_len) {
    return guarded([&]() -> lc_status {
    if (!ctx || !array || !schema || !out_bytes || !out_len) return fail(LC_ERR_INVALID, "null argument");
    CtxSymtabs st(ctx);
    std::vector<uint8_t> out;
    const lc_status rc = transcode_arrow(array, schema, hint, st, path_id, out);
    if (rc != LC_OK) return fail(rc, "array type is not transcoded to a Liquid encoding");
    *out_bytes = static_cast<uint8_t*>(std::malloc(out.size() ? out.size() : 1));
    if (!*out_bytes) return fail(LC_ERR_OOM, "malloc");
    std::memcpy(*out_bytes, out.data(), out.size());
    *out_len = out.size();
    return LC_OK;
    });
}

lc_status lc_insert_arrow(lc_ctx* ctx, uint64_t entry_id, const struct ArrowArray* array,
                          const struct ArrowSchema* schema, int32_t hint, uint64_t path_id) {
    return guarded([&]() -> lc_status {
    uint8_t* b = nullptr;
    size_t l = 0;
    lc_status rc = lc_transcode_arrow(ctx, array, schema, hint, path_id, &b, &l);
    if (rc != LC_OK) return rc;
    const uint8_t* bp = b;
    rc = lc_stage(ctx, 1, &entry_id, &bp, &l, &path_id);
    std::free(b);
    return rc;
    });
}

lc_status lc_insert_arrow_batch(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, const struct ArrowArray* const* arrays,
                                const struct ArrowSchema* const* schemas, const int32_t* hints, const uint64_t* path_ids) {
    return guarded([&]() -> lc_status {
    if (!ctx || (n && (!entry_ids || !arrays || !schemas))) return fail(LC_ERR_INVALID, "null argument");
    // transcode every array (host), then ONE lc_stage: one upload, one signature-builder launch, one publication
    CtxSymtabs st(ctx);
    std::vector<std::vector<uint8_t>> blobs(n);
    std::vector<const uint8_t*> ptrs(n);
    std::vector<size_t> lens(n);
    std::vector<uint64_t> paths(n, 0);
    for (uint64_t i = 0; i < n; i++) {
        paths[i] = path_ids ? path_ids[i] : 0;
        const lc_status rc = transcode_arrow(arrays[i], schemas[i], hints ? hints[i] : LC_HINT_NONE, st, paths[i], blobs[i]);
        if (rc != LC_OK) return fail(rc, "array type is not transcoded to a Liquid encoding");
        ptrs[i] = blobs[i].data();
        lens[i] = blobs[i].size();
    }
    return lc_stage(ctx, n, entry_ids, ptrs.data(), lens.data(), paths.data());
    });
}

// ------------------------------------------------------------------ on-device transcoder (fixed-width integers)
static int int_phys_of_format(const char* f) {
    if (!f) return -1;
    const std::string fmt = f;
    if (fmt == "c") return kI8;
    if (fmt == "C") return kU8;
    if (fmt == "s") return kI16;
    if (fmt == "S") return kU16;
    if (fmt == "i") return kI32;
    if (fmt == "I") return kU32;
    if (fmt == "l") return kI64;
    if (fmt == "L") return kU64;
    if (fmt == "tdD") return kDate32;
    if (fmt == "tdm") return kDate64;
    if (fmt.rfind("ts", 0) == 0 && fmt.size() == 4 && fmt[3] == ':') {  // timestamps without a time zone
        switch (fmt[2]) {
            case 's': return kTsS;
            case 'm': return kTsMs;
            case 'u': return kTsUs;
            case 'n': return kTsNs;
            default: return -1;
        }
    }
    return -1;
}

// One array to encode on the device: where its values / validity words sit in the staging buffer, and what comes out.
namespace {
struct DevEncodeItem {
    uint64_t id = 0;
    int phys = 0, vw = 0;
    bool is_signed = false, has_validity = false;
    uint32_t n = 0;
    size_t in_values = 0, in_validity = 0;  // byte offsets in the staging buffer (insert path)
    const uint8_t* d_values = nullptr;      // device pointers the kernels read
    const uint64_t* d_validity = nullptr;
    int squeezed_field = -1;                // >= 0: the values are date components of an entry of type `phys`
    bool force_all_null = false;            // the source entry is all null (it carries no validity buffer to tell)
    // clamp squeeze: the values are the packed-domain offsets of an existing entry; width and reference are given
    bool forced = false;
    int forced_W = 0, orig_W = 0;
    uint64_t entry_reference = 0, clamp_max = 0;
    bool entry_signed = false;
    bool quantize = false;                  // with `forced`: bucket indices instead of clamped offsets
    uint64_t quant_width = 0;               // result: the bucket width
    bool is_float = false;                  // Float32 / Float64: the ALP path (device_encode_floats)
    uint32_t n_valid = 0;                   // floats: valid rows, counted on the host (all-null arrays skip the encoder)
    int in_stride = 0;                      // decimals from Arrow: bytes per value in the staging buffer (16 / 32)
    int logical = kInteger;                 // kDecimal: a decimal entry (u64 offsets of the unscaled values)
    int dec_precision = 0, dec_scale = 0, dec_is256 = 0, entry_value_width = 0;
    // results
    bool all_null = false;
    int W = 0;
    uint64_t reference = 0;
    size_t out_packed = size_t(-1), out_validity = size_t(-1), blob_begin = 0, blob_bytes = 0;
};

// min/max -> FoR reference + bit width, then pack into a fresh arena blob and register the entries.
lc_status device_encode_and_register(lc_ctx* ctx, std::vector<DevEncodeItem>& items) {
    const size_t n = items.size();
    if (n == 0) return LC_OK;
    std::vector<EncodeDesc> descs(n);
    uint32_t max_rows = 0;
    for (size_t i = 0; i < n; i++) {
        DevEncodeItem& it = items[i];
        EncodeDesc& d = descs[i];
        d = EncodeDesc{};
        d.values = it.d_values;
        d.validity = it.has_validity ? it.d_validity : nullptr;
        d.n = it.n;
        d.value_log2 = uint8_t(it.vw == 1 ? 0 : it.vw == 2 ? 1 : it.vw == 4 ? 2 : 3);
        d.is_signed = it.is_signed ? 1 : 0;
        d.stride_log2 = uint8_t(it.in_stride == 16 ? 4 : it.in_stride == 32 ? 5 : 0);
        max_rows = std::max(max_rows, it.n);
    }
    EncodeDesc* d_descs = static_cast<EncodeDesc*>(pool_alloc(ctx, n * sizeof(EncodeDesc)));
    EncodeMinMax* d_mm = static_cast<EncodeMinMax*>(pool_alloc(ctx, n * sizeof(EncodeMinMax)));
    struct Scratch {
        lc_ctx* c; void* a; void* b;
        ~Scratch() { (void)hipStreamSynchronize(nullptr); pool_release(c, a); pool_release(c, b); }
    } scratch{ctx, d_descs, d_mm};
    if (!d_descs || !d_mm) return fail(LC_ERR_OOM, "hipMalloc (encoder scratch)");
    LC_HIP(hipMemcpy(d_descs, descs.data(), n * sizeof(EncodeDesc), hipMemcpyHostToDevice));
    LC_HIP(launch_col_minmax(d_descs, uint32_t(n), d_mm, nullptr));
    std::vector<EncodeMinMax> mm(n);
    LC_HIP(hipMemcpy(mm.data(), d_mm, n * sizeof(EncodeMinMax), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++)
        if (items[i].in_stride && mm[i].n_wide)
            return fail(LC_UNSUPPORTED, "decimal values that do not fit a u64 stay on the reference's CPU path (fits_u64)");
    // layout of the batch's blob: per entry [packed + 128 slack][validity words]
    size_t total = 0;
    for (size_t i = 0; i < n; i++) {
        DevEncodeItem& it = items[i];
        it.all_null = (!it.forced && mm[i].n_valid == 0) || it.force_all_null;  // also empty arrays (primitive_array.rs:160-170)
        it.blob_begin = align_up(total, kSectionAlign);
        size_t cur = it.blob_begin;
        if (!it.all_null && it.forced && it.quantize) {
            // bucket width = ceil((max offset + 1) / 2^W') (primitive_array.rs:466-470); the last bucket absorbs the rest
            const uint64_t count = uint64_t(1) << it.forced_W;
            const uint64_t range_size = mm[i].mx == ~uint64_t(0) ? mm[i].mx : mm[i].mx + 1;  // saturating_add(1)
            it.quant_width = std::max<uint64_t>(range_size / count + (range_size % count ? 1 : 0), 1);
            it.clamp_max = count - 1;
        }
        if (!it.all_null && it.forced) {
            it.W = it.forced_W;
            it.reference = 0;  // the values already are offsets from the entry's reference
            it.out_packed = cur;
            cur = align_up(cur + packed_bytes(it.W, it.n) + 128, kSectionAlign);
            if (it.has_validity) {
                it.out_validity = cur;
                cur = align_up(cur + ((size_t(it.n) + 63) / 64) * 8, kSectionAlign);
            }
        } else if (!it.all_null) {
            uint64_t range;
            if (it.is_signed) range = uint64_t(int64_t(mm[i].mx)) - uint64_t(int64_t(mm[i].mn));
            else range = mm[i].mx - mm[i].mn;
            if (it.vw < 8) range &= (uint64_t(1) << (8 * it.vw)) - 1;  // wrapping subtraction in the native width (:173-180)
            it.W = bit_width_of(range);
            it.reference = mm[i].mn;
            it.out_packed = cur;
            cur = align_up(cur + packed_bytes(it.W, it.n) + 128, kSectionAlign);
            if (it.has_validity) {
                it.out_validity = cur;
                cur = align_up(cur + ((size_t(it.n) + 63) / 64) * 8, kSectionAlign);
            }
        }
        it.blob_bytes = cur - it.blob_begin;
        total = cur;
    }
    total = align_up(total, kSectionAlign) + 256;
    ArenaReservation reserved(ctx);  // declared before the lock: its destructor locks on its own (failure paths only)
    std::unique_lock<std::shared_mutex> g(ctx->mu);
    uint8_t* dbase = nullptr;
    int slab = -1;
    lc_status st = arena_alloc(ctx, total, &dbase, &slab);
    if (st != LC_OK) return st;
    ctx->slabs[size_t(slab)].live += int64_t(n) - 1;
    reserved.arm(slab, int64_t(n));
    LC_HIP(hipMemsetAsync(dbase, 0, total, nullptr));  // slack behind packed sections and tail words must be zero
    for (size_t i = 0; i < n; i++) {
        const DevEncodeItem& it = items[i];
        EncodeDesc& d = descs[i];
        d.W = uint8_t(it.all_null ? 0 : it.W);
        d.reference = it.reference;
        d.clamp_max = it.clamp_max;
        d.quant_width = it.quantize ? it.quant_width : 0;
        d.packed = it.all_null ? nullptr : dbase + it.out_packed;
        d.validity_out = it.out_validity == size_t(-1) ? nullptr : reinterpret_cast<uint64_t*>(dbase + it.out_validity);
    }
    LC_HIP(hipMemcpy(d_descs, descs.data(), n * sizeof(EncodeDesc), hipMemcpyHostToDevice));
    // one pack launch per lane width present in the batch (the kernel is templated on the lane type)
    for (int ll = 0; ll < 4; ll++) {
        // entries of other widths are skipped by giving the launch a filtered copy of the descriptors
        std::vector<EncodeDesc> sub;
        uint32_t sub_rows = 0;
        for (size_t i = 0; i < n; i++)
            if (descs[i].value_log2 == ll) { sub.push_back(descs[i]); sub_rows = std::max(sub_rows, descs[i].n); }
        if (sub.empty()) continue;
        if (sub.size() == n) {
            LC_HIP(launch_fl_pack(d_descs, uint32_t(n), sub_rows, ll + 3, nullptr));
        } else {
            EncodeDesc* d_sub = static_cast<EncodeDesc*>(pool_alloc(ctx, sub.size() * sizeof(EncodeDesc)));
            if (!d_sub) return fail(LC_ERR_OOM, "hipMalloc (encoder scratch)");
            const hipError_t e1 = hipMemcpy(d_sub, sub.data(), sub.size() * sizeof(EncodeDesc), hipMemcpyHostToDevice);
            const hipError_t e2 = e1 == hipSuccess ? launch_fl_pack(d_sub, uint32_t(sub.size()), sub_rows, ll + 3, nullptr) : e1;
            (void)hipStreamSynchronize(nullptr);
            pool_release(ctx, d_sub);
            if (e2 != hipSuccess) return fail(LC_ERR_DEVICE, "k_fl_pack launch failed");
        }
    }
    LC_HIP(hipStreamSynchronize(nullptr));
    for (size_t i = 0; i < n; i++) {
        const DevEncodeItem& it = items[i];
        Entry e;
        e.is_str = false;
        e.logical = it.logical;
        e.dec_precision = it.dec_precision;
        e.dec_scale = it.dec_scale;
        e.dec_is256 = it.dec_is256;
        e.phys = it.phys;
        e.len = it.n;
        e.all_null = it.all_null;
        e.nullable = it.has_validity || it.all_null;
        e.W = it.all_null ? 0 : it.W;
        e.slab = slab;
        e.device_bytes = it.blob_bytes;
        e.squeezed_field = it.squeezed_field;
        e.clamped = it.forced && !it.quantize;
        e.quantized = it.forced && it.quantize && !it.all_null;
        e.bucket_width = it.quant_width;
        e.orig_W = it.orig_W;
        FixedDesc& d = e.fd;
        d = FixedDesc{};
        d.len = it.n;
        d.W = uint8_t(e.W);
        d.lane_log2 = uint8_t(it.vw == 1 ? 3 : it.vw == 2 ? 4 : it.vw == 4 ? 5 : 6);
        d.value_width = uint8_t(it.vw);
        d.kind = kKindInt;
        d.is_signed = (it.forced ? it.entry_signed : it.is_signed) ? 1 : 0;
        d.reference = it.all_null ? 0 : (it.forced ? it.entry_reference : it.reference);  // sign-extended for signed types
        d.packed = it.all_null ? nullptr : dbase + it.out_packed;
        d.validity = it.out_validity == size_t(-1) ? nullptr : reinterpret_cast<const uint64_t*>(dbase + it.out_validity);
        if (e.quantized) {
            d.quantized = it.logical == kDecimal ? 2 : 1;
            d.patch_idx = reinterpret_cast<const uint64_t*>(uintptr_t(it.quant_width));  // see quant_bucket_width()
        }
        if (it.logical == kDecimal) {
            d.kind = kKindDecimal;
            d.value_width = uint8_t(it.entry_value_width);
        }
        publish_entry(ctx, it.id, std::move(e));
    }
    reserved.disarm();
    return LC_OK;
}
// Floats: ALP on the device (k_alp_search -> k_alp_encode -> k_fl_pack + k_alp_copy_patches), then the entries are
// registered exactly as lc_stage registers a LiquidFloatArray (build_fixed).
lc_status device_encode_floats(lc_ctx* ctx, std::vector<DevEncodeItem>& items) {
    if (items.empty()) return LC_OK;
    for (int vlog = 2; vlog <= 3; vlog++) {  // one pass per float width (the kernels are templated on it)
        std::vector<size_t> sel;
        for (size_t i = 0; i < items.size(); i++)
            if ((items[i].vw == 4) == (vlog == 2)) sel.push_back(i);
        if (sel.empty()) continue;
        const size_t m = sel.size();
        const size_t fw = size_t(1) << vlog;
        uint32_t stride = 1;
        for (size_t k : sel) stride = std::max(stride, items[k].n);
        std::vector<EncodeDesc> descs(m);
        for (size_t j = 0; j < m; j++) {
            const DevEncodeItem& it = items[sel[j]];
            EncodeDesc& d = descs[j];
            d = EncodeDesc{};
            d.values = it.d_values;
            d.validity = it.has_validity ? it.d_validity : nullptr;
            d.n = it.n_valid == 0 ? 0 : it.n;  // all-null arrays carry no encoded values (float_array.rs:620-631)
            d.value_log2 = uint8_t(vlog);
            d.is_signed = 1;
        }
        EncodeDesc* d_descs = static_cast<EncodeDesc*>(pool_alloc(ctx, m * sizeof(EncodeDesc)));
        AlpStatsHost* d_stats = static_cast<AlpStatsHost*>(pool_alloc(ctx, m * sizeof(AlpStatsHost)));
        uint8_t* d_enc = static_cast<uint8_t*>(pool_alloc(ctx, m * size_t(stride) * fw + 64));
        uint64_t* d_xi = static_cast<uint64_t*>(pool_alloc(ctx, m * size_t(stride) * 8 + 64));
        uint8_t* d_xv = static_cast<uint8_t*>(pool_alloc(ctx, m * size_t(stride) * fw + 64));
        void** d_ptrs = static_cast<void**>(pool_alloc(ctx, 2 * m * sizeof(void*)));
        struct Scratch {
            lc_ctx* c; void* p[6];
            ~Scratch() { (void)hipStreamSynchronize(nullptr); for (void* q : p) pool_release(c, q); }
        } scratch{ctx, {d_descs, d_stats, d_enc, d_xi, d_xv, d_ptrs}};
        if (!d_descs || !d_stats || !d_enc || !d_xi || !d_xv || !d_ptrs) return fail(LC_ERR_OOM, "hipMalloc (ALP encoder scratch)");
        LC_HIP(hipMemcpy(d_descs, descs.data(), m * sizeof(EncodeDesc), hipMemcpyHostToDevice));
        LC_HIP(hipMemsetAsync(d_stats, 0, m * sizeof(AlpStatsHost), nullptr));
        LC_HIP(launch_alp_search(d_descs, uint32_t(m), vlog, d_stats, nullptr));
        LC_HIP(launch_alp_encode(d_descs, uint32_t(m), vlog, d_stats, stride, d_enc, d_xi, d_xv, nullptr));
        std::vector<AlpStatsHost> stv(m);
        LC_HIP(hipMemcpy(stv.data(), d_stats, m * sizeof(AlpStatsHost), hipMemcpyDeviceToHost));
        // blob layout per entry: [packed + 128 slack][validity words][patch indices u64][patch values]
        struct Lay { size_t begin, packed, valid, pidx, pval, bytes; int W; bool all_null; };
        std::vector<Lay> lay(m);
        size_t total = 0;
        uint32_t max_exc = 0;
        for (size_t j = 0; j < m; j++) {
            const DevEncodeItem& it = items[sel[j]];
            Lay& L = lay[j];
            L.all_null = it.n_valid == 0;
            L.begin = align_up(total, kSectionAlign);
            L.packed = L.valid = L.pidx = L.pval = size_t(-1);
            size_t cur = L.begin;
            L.W = 0;
            if (!L.all_null) {
                const uint64_t range = vlog == 2 ? uint64_t(uint32_t(uint32_t(stv[j].mx) - uint32_t(stv[j].mn)))
                                                 : uint64_t(stv[j].mx) - uint64_t(stv[j].mn);
                L.W = bit_width_of(range);
                L.packed = cur;
                cur = align_up(cur + packed_bytes(L.W, it.n) + 128, kSectionAlign);
                if (it.has_validity) {
                    L.valid = cur;
                    cur = align_up(cur + ((size_t(it.n) + 63) / 64) * 8, kSectionAlign);
                }
                if (stv[j].n_exc) {
                    L.pidx = cur;
                    cur = align_up(cur + size_t(stv[j].n_exc) * 8, kSectionAlign);
                    L.pval = cur;
                    cur = align_up(cur + size_t(stv[j].n_exc) * fw, kSectionAlign);
                    max_exc = std::max(max_exc, stv[j].n_exc);
                }
            }
            L.bytes = cur - L.begin;
            total = cur;
        }
        total = align_up(total, kSectionAlign) + 256;
        ArenaReservation reserved(ctx);  // before the lock: see device_encode_and_register
        std::unique_lock<std::shared_mutex> g(ctx->mu);
        uint8_t* dbase = nullptr;
        int slab = -1;
        lc_status st = arena_alloc(ctx, total, &dbase, &slab);
        if (st != LC_OK) return st;
        ctx->slabs[size_t(slab)].live += int64_t(m) - 1;
        reserved.arm(slab, int64_t(m));
        LC_HIP(hipMemsetAsync(dbase, 0, total, nullptr));
        std::vector<void*> ptrs(2 * m, nullptr);
        for (size_t j = 0; j < m; j++) {
            const DevEncodeItem& it = items[sel[j]];
            const Lay& L = lay[j];
            EncodeDesc& d = descs[j];
            d.values = d_enc + j * size_t(stride) * fw;  // the encoded integers
            d.n = L.all_null ? 0 : it.n;
            d.W = uint8_t(L.W);
            d.reference = uint64_t(stv[j].mn);
            d.packed = L.all_null ? nullptr : dbase + L.packed;
            d.validity_out = L.valid == size_t(-1) ? nullptr : reinterpret_cast<uint64_t*>(dbase + L.valid);
            if (L.all_null) d.validity = nullptr;
            ptrs[j] = L.pidx == size_t(-1) ? nullptr : dbase + L.pidx;
            ptrs[m + j] = L.pval == size_t(-1) ? nullptr : dbase + L.pval;
        }
        LC_HIP(hipMemcpy(d_descs, descs.data(), m * sizeof(EncodeDesc), hipMemcpyHostToDevice));
        LC_HIP(hipMemcpy(d_ptrs, ptrs.data(), 2 * m * sizeof(void*), hipMemcpyHostToDevice));
        LC_HIP(launch_fl_pack(d_descs, uint32_t(m), stride, vlog + 3, nullptr));
        LC_HIP(launch_alp_copy_patches(d_stats, uint32_t(m), vlog, stride, d_xi, d_xv, d_ptrs, d_ptrs + m, max_exc, nullptr));
        LC_HIP(hipStreamSynchronize(nullptr));
        for (size_t j = 0; j < m; j++) {
            const DevEncodeItem& it = items[sel[j]];
            const Lay& L = lay[j];
            Entry e;
            e.is_str = false;
            e.logical = kFloat;
            e.phys = it.phys;
            e.len = it.n;
            e.all_null = L.all_null;
            e.nullable = it.has_validity || L.all_null;
            e.W = L.all_null ? 0 : L.W;
            e.slab = slab;
            e.device_bytes = L.bytes;
            FixedDesc& d = e.fd;
            d = FixedDesc{};
            d.len = it.n;
            d.W = uint8_t(e.W);
            d.lane_log2 = uint8_t(vlog + 3);
            d.value_width = uint8_t(fw);
            d.kind = vlog == 2 ? kKindF32 : kKindF64;
            d.is_signed = 1;
            if (!L.all_null) {
                d.reference = uint64_t(stv[j].mn);  // int64 bits, sign-extended for f32 by the kernel's int64 minimum
                d.alp_e = uint8_t(stv[j].e);
                d.alp_f = uint8_t(stv[j].f);
                d.patch_len = stv[j].n_exc;
                d.packed = dbase + L.packed;
                d.validity = L.valid == size_t(-1) ? nullptr : reinterpret_cast<const uint64_t*>(dbase + L.valid);
                d.patch_idx = L.pidx == size_t(-1) ? nullptr : reinterpret_cast<const uint64_t*>(dbase + L.pidx);
                d.patch_val = L.pval == size_t(-1) ? nullptr : dbase + L.pval;
            }
            publish_entry(ctx, it.id, std::move(e));
        }
        reserved.disarm();
    }
    return LC_OK;
}
}  // namespace

// ---- Utf8 / Binary arrays: dictionary, FSST, prefix keys, fingerprints, compact offsets and row lists on the device
// (lc_bv_encode.hip).  All-or-nothing: nothing is published unless every array of the call could be encoded.
struct BvItem {
    uint64_t id = 0;
    const struct ArrowArray* a = nullptr;
    int arrow_type = 0;
    bool want_fp = false;
    uint64_t path_id = 0;
    uint32_t slot = 0, encoder = 0;
    uint32_t n = 0;
    bool has_validity = false;
    bool views = false;  // Utf8View / BinaryView: the rows are gathered into offsets + bytes while they are staged
    size_t data_len = 0;
    uint32_t table_slots = 0;
    // row r of the array (no validity check)
    std::pair<const uint8_t*, size_t> row(uint32_t r) const {
        const size_t i = size_t(a->offset) + r;
        if (!views) {
            const int32_t* o = static_cast<const int32_t*>(a->buffers[1]);
            const uint8_t* d = static_cast<const uint8_t*>(a->buffers[2]);
            return {d ? d + o[i] : reinterpret_cast<const uint8_t*>(""), size_t(o[i + 1] - o[i])};
        }
        const uint8_t* view = static_cast<const uint8_t*>(a->buffers[1]) + 16 * i;
        const uint32_t len = rd<uint32_t>(view);
        if (len <= 12) return {view + 4, len};
        const int32_t buf = rd<int32_t>(view + 8), off = rd<int32_t>(view + 12);
        return {static_cast<const uint8_t*>(a->buffers[2 + buf]) + off, len};
    }
    // byte offsets inside the input staging buffer / the device scratch
    size_t in_offsets = 0, in_data = 0, in_validity = 0;
    size_t sc_table = 0, sc_row_slot = 0, sc_dict_row = 0, sc_dict_index = 0, sc_clen = 0, sc_offsets = 0, sc_fp = 0, sc_keys = 0, sc_comp = 0;
};

static lc_status device_encode_byte_views(lc_ctx* ctx, std::vector<BvItem>& items) {
    const size_t m = items.size();
    if (m == 0) return LC_OK;
    // symbol tables: train-once-per-path from the first array of the path (transcode.rs:16-33), as lc_insert_arrow does
    CtxSymtabs symtabs(ctx);
    std::vector<uint32_t> enc_slots;
    for (BvItem& it : items) {
        if (!symtabs.find(it.path_id)) {
            const uint8_t* validity = it.has_validity ? static_cast<const uint8_t*>(it.a->buffers[0]) : nullptr;
            std::vector<std::pair<const uint8_t*, size_t>> train;
            train.reserve(it.n);
            for (uint32_t r = 0; r < it.n; r++) {
                if (validity && !get_bit(validity, size_t(it.a->offset) + r)) continue;
                train.push_back(it.row(r));
            }
            symtabs.insert(it.path_id, fsst_train(train));
        }
        {
            std::lock_guard<std::mutex> g(ctx->st_mu);
            it.slot = ctx->symtab_slot.at(it.path_id);
        }
        size_t e = 0;
        while (e < enc_slots.size() && enc_slots[e] != it.slot) e++;
        if (e == enc_slots.size()) enc_slots.push_back(it.slot);
        it.encoder = uint32_t(e);
    }
    {
        const lc_status ss = sync_symtabs(ctx);  // the signature builder reads the device copies
        if (ss != LC_OK) return ss;
    }
    // input staging (pinned) and device scratch layouts
    size_t in_bytes = align_up(enc_slots.size() * sizeof(DevFsstEncoder), 256);
    size_t sc_bytes = 0, table_bytes = 0;
    for (BvItem& it : items) {
        it.in_offsets = in_bytes;
        in_bytes = align_up(in_bytes + (size_t(it.n) + 1) * 4, 16);
        it.in_data = in_bytes;
        in_bytes = align_up(in_bytes + it.data_len + 16, 16);
        if (it.has_validity) {
            it.in_validity = in_bytes;
            in_bytes = align_up(in_bytes + ((size_t(it.n) + 63) / 64) * 8, 16);
        }
        it.table_slots = 64;
        while (it.table_slots < it.n * 2u + 16u) it.table_slots <<= 1;
        it.sc_table = table_bytes;  // the tables sit together at the front of the scratch: one memset
        table_bytes += size_t(it.table_slots) * 4;
    }
    sc_bytes = align_up(table_bytes, 256);
    for (BvItem& it : items) {
        auto take = [&](size_t bytes) { const size_t at = sc_bytes; sc_bytes = align_up(sc_bytes + bytes, 16); return at; };
        it.sc_row_slot = take(size_t(it.n) * 4);
        it.sc_dict_row = take(size_t(it.n) * 4);
        it.sc_dict_index = take(size_t(it.n) * 4);
        it.sc_clen = take(size_t(it.n) * 4);
        it.sc_offsets = take((size_t(it.n) + 1) * 4);
        it.sc_fp = take(size_t(it.n) * 4);
        it.sc_keys = take(size_t(it.n) * 2);
        it.sc_comp = take(2 * it.data_len + 16);
    }
    const size_t sc_descs = sc_bytes;
    sc_bytes = align_up(sc_bytes + m * sizeof(BvEncodeDesc), 256);
    const size_t sc_packs = sc_bytes;
    sc_bytes = align_up(sc_bytes + m * sizeof(BvPackDesc), 256);
    const size_t sc_stats = sc_bytes;
    sc_bytes = align_up(sc_bytes + m * sizeof(BvEncodeStats), 256);
    in_bytes = align_up(in_bytes, 256);
    uint8_t* h_in = static_cast<uint8_t*>(host_pool_alloc(ctx, in_bytes));
    uint8_t* d_in = static_cast<uint8_t*>(pool_alloc(ctx, in_bytes));
    uint8_t* d_sc = static_cast<uint8_t*>(pool_alloc(ctx, sc_bytes));
    struct Bufs {
        lc_ctx* c; void* h; void* d; void* s;
        // (all device work of this call runs on `side`, which StreamGuard below has synchronised by the time this runs:
        // no device-wide synchronise, so concurrent calls of other host threads overlap)
        ~Bufs() { host_pool_release(c, h); pool_release(c, d); pool_release(c, s); }
    } bufs{ctx, h_in, d_in, d_sc};
    if (!h_in || !d_in || !d_sc) return fail(LC_ERR_OOM, "staging buffers of the on-device byte-view transcoder");
    for (size_t e = 0; e < enc_slots.size(); e++) {
        const SymbolTable* st;
        {
            std::lock_guard<std::mutex> g(ctx->st_mu);
            st = ctx->symtabs[enc_slots[e]].get();  // unique_ptr targets are stable
        }
        FsstEncoder enc(*st);
        enc.export_device(reinterpret_cast<DevFsstEncoder*>(h_in) + e, kDevEncShort2Slots, dev_enc_short2_hash);
    }
    for (const BvItem& it : items) {
        if (!it.views && !it.a->buffers[1]) {  // (an empty array may come without an offsets buffer)
            std::memset(h_in + it.in_offsets, 0, (size_t(it.n) + 1) * 4);
        } else if (!it.views) {
            const int32_t* o = static_cast<const int32_t*>(it.a->buffers[1]) + it.a->offset;
            const uint8_t* data = static_cast<const uint8_t*>(it.a->buffers[2]);
            std::memcpy(h_in + it.in_offsets, o, (size_t(it.n) + 1) * 4);
            if (it.data_len) std::memcpy(h_in + it.in_data, data + o[0], it.data_len);
        } else {
            // views: the same pass that copies the bytes into pinned memory lays them out as offsets + data
            int32_t* o = reinterpret_cast<int32_t*>(h_in + it.in_offsets);
            uint8_t* dst = h_in + it.in_data;
            const uint8_t* validity = it.has_validity ? static_cast<const uint8_t*>(it.a->buffers[0]) : nullptr;
            size_t pos = 0;
            for (uint32_t r = 0; r < it.n; r++) {
                o[r] = int32_t(pos);
                if (validity && !get_bit(validity, size_t(it.a->offset) + r)) continue;  // (null slots: no bytes)
                const auto v = it.row(r);
                if (v.second) std::memcpy(dst + pos, v.first, v.second);
                pos += v.second;
            }
            o[it.n] = int32_t(pos);
        }
        std::memset(h_in + it.in_data + it.data_len, 0, 16);
        if (it.has_validity) {
            const size_t words = (size_t(it.n) + 63) / 64;
            std::memset(h_in + it.in_validity, 0, words * 8);
            const uint8_t* src = static_cast<const uint8_t*>(it.a->buffers[0]);
            uint8_t* dst = h_in + it.in_validity;
            for (size_t k = 0; k < it.n; k++)
                if (get_bit(src, size_t(it.a->offset) + k)) set_bit(dst, k);
        }
    }
    hipStream_t side = stream_acquire(ctx);
    struct StreamGuard {
        lc_ctx* c; hipStream_t s;
        ~StreamGuard() { (void)hipStreamSynchronize(s); stream_release(c, s); }
    } sguard{ctx, side};
    LC_HIP(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, side));
    LC_HIP(hipMemsetAsync(d_sc, 0xFF, table_bytes, side));
    std::vector<BvEncodeDesc> descs(m);
    for (size_t i = 0; i < m; i++) {
        const BvItem& it = items[i];
        BvEncodeDesc& d = descs[i];
        d = BvEncodeDesc{};
        d.offsets = reinterpret_cast<const int32_t*>(d_in + it.in_offsets);
        d.data = d_in + it.in_data;
        d.validity = it.has_validity ? reinterpret_cast<const uint64_t*>(d_in + it.in_validity) : nullptr;
        d.table = reinterpret_cast<uint32_t*>(d_sc + it.sc_table);
        d.row_slot = reinterpret_cast<uint32_t*>(d_sc + it.sc_row_slot);
        d.dict_row = reinterpret_cast<uint32_t*>(d_sc + it.sc_dict_row);
        d.dict_index = reinterpret_cast<uint32_t*>(d_sc + it.sc_dict_index);
        d.clen = reinterpret_cast<uint32_t*>(d_sc + it.sc_clen);
        d.offsets_out = reinterpret_cast<uint32_t*>(d_sc + it.sc_offsets);
        d.fingerprints = reinterpret_cast<uint32_t*>(d_sc + it.sc_fp);
        d.keys = reinterpret_cast<uint16_t*>(d_sc + it.sc_keys);
        d.comp = d_sc + it.sc_comp;
        d.stats = reinterpret_cast<BvEncodeStats*>(d_sc + sc_stats) + i;
        d.n = it.n;
        d.table_mask = it.table_slots - 1;
        d.encoder = it.encoder;
    }
    BvEncodeDesc* d_descs = reinterpret_cast<BvEncodeDesc*>(d_sc + sc_descs);
    LC_HIP(hipMemcpyAsync(d_descs, descs.data(), m * sizeof(BvEncodeDesc), hipMemcpyHostToDevice, side));
    LC_HIP(launch_bv_build(d_descs, uint32_t(m), reinterpret_cast<const DevFsstEncoder*>(d_in), side));
    std::vector<BvEncodeStats> stats(m);
    LC_HIP(hipMemcpyAsync(stats.data(), d_sc + sc_stats, m * sizeof(BvEncodeStats), hipMemcpyDeviceToHost, side));
    LC_HIP(hipStreamSynchronize(side));
    for (size_t i = 0; i < m; i++) {
        if (stats[i].inexact)
            return fail(LC_UNSUPPORTED, "compact-offset fit of this array leaves the exact range of f64 sums: use lc_insert_arrow");
        if (stats[i].d > 65536) return fail(LC_UNSUPPORTED, "more than 65536 distinct values in one array");
    }
    // layout of the entries' blob: the sections build_str lays out, in the same order with the same slack
    struct Sections { size_t off[9]; size_t begin, bytes; };
    std::vector<Sections> lay(m);
    size_t total = 0;
    for (size_t i = 0; i < m; i++) {
        const BvItem& it = items[i];
        const BvEncodeStats& st = stats[i];
        Sections& L = lay[i];
        for (size_t& o : L.off) o = size_t(-1);
        L.begin = align_up(total, kSectionAlign);
        size_t cur = L.begin;
        auto add = [&](size_t bytes, size_t extra) { const size_t at = align_up(cur, kSectionAlign); cur = at + bytes + extra; return at; };
        const size_t nw = std::max<size_t>((size_t(st.d) + 63) / 64, 1);
        L.off[0] = add(size_t(it.n) * 2, 16);
        if (it.has_validity) L.off[1] = add(((size_t(it.n) + 63) / 64) * 8, 0);
        L.off[2] = add(size_t(st.d) * 8, 0);
        const bool has_fp = it.want_fp && st.d > 0;  // an empty fingerprint section reads back as "none" (serialization.rs:209-218)
        if (has_fp) L.off[3] = add(size_t(st.d) * 4, 0);
        L.off[4] = add((size_t(st.d) + 1) * st.offset_bytes, 8);
        L.off[5] = add(st.fsst_len, 16);
        L.off[6] = add(st.shared_prefix_len, 8);
        if (has_fp && ctx->build_signatures) {
            L.off[7] = add(size_t(kSigBits) * nw * 8, 0);
            if (ctx->build_postings && it.n <= kPostMaxRows && st.d > 0)
                L.off[8] = add((size_t(st.d) + 1 + size_t(it.n) + 32) * 2, 16);
        }
        L.bytes = align_up(cur, kSectionAlign) - L.begin;
        total = L.begin + L.bytes;
    }
    total = align_up(total, kSectionAlign) + 256;
    uint8_t* dbase = nullptr;
    int slab = -1;
    ArenaReservation reserved(ctx);
    {
        std::unique_lock<std::shared_mutex> g(ctx->mu);
        const lc_status st = arena_alloc(ctx, total, &dbase, &slab);
        if (st != LC_OK) return st;
        ctx->slabs[size_t(slab)].live += int64_t(m) - 1;
        reserved.arm(slab, int64_t(m));
    }
    LC_HIP(hipMemsetAsync(dbase, 0, total, side));  // slack behind sections, signature slices, spare list entries
    std::vector<BvPackDesc> packs(m);
    std::vector<Entry> entries(m);
    std::vector<StrDesc> sig_descs;
    for (size_t i = 0; i < m; i++) {
        const BvItem& it = items[i];
        const BvEncodeStats& st = stats[i];
        const Sections& L = lay[i];
        auto ptr = [&](int k) -> uint8_t* { return L.off[k] == size_t(-1) ? nullptr : dbase + L.off[k]; };
        BvPackDesc& p = packs[i];
        p = BvPackDesc{};
        p.keys = reinterpret_cast<uint16_t*>(ptr(0));
        p.validity = reinterpret_cast<uint64_t*>(ptr(1));
        p.prefix_keys = ptr(2);
        p.fingerprints = reinterpret_cast<uint32_t*>(ptr(3));
        p.residuals = ptr(4);
        p.fsst = ptr(5);
        p.shared_prefix = ptr(6);
        // (k_bv_pack sorts an entry's (key, row) pairs in LDS: up to kPostLdsRows rows; the lists of larger entries are made
        // on the host from the keys the kernels produced, below)
        p.postings = it.n <= kPostLdsRows ? reinterpret_cast<uint16_t*>(ptr(8)) : nullptr;
        p.d = st.d;
        p.shared_prefix_len = st.shared_prefix_len;
        p.offset_bytes = st.offset_bytes;
        p.slope = st.slope;
        p.intercept = st.intercept;
        Entry& e = entries[i];
        e.is_str = true;
        e.logical = kByteView;
        e.phys = it.arrow_type;
        e.len = it.n;
        e.nullable = it.has_validity;
        e.all_null = st.d == 0 && it.n > 0;  // (as build_str: an array of nulls only has no dictionary)
        e.W = 16;
        e.path_id = it.path_id;
        e.dict_len = st.d;
        e.has_fp = L.off[3] != size_t(-1);
        e.offsets_bytes = (st.d + 1) * st.offset_bytes;
        e.fsst_len = st.fsst_len;
        e.raw_bytes = st.raw_bytes;
        e.slab = slab;
        e.device_bytes = L.bytes;
        e.sig_on_device = L.off[7] != size_t(-1);
        StrDesc& d = e.sd;
        d = StrDesc{};
        d.n = it.n;
        d.d = st.d;
        d.slope = st.slope;
        d.intercept = st.intercept;
        d.offset_bytes = uint8_t(st.offset_bytes);
        d.fsst_len = st.fsst_len;
        d.shared_prefix_len = st.shared_prefix_len;
        d.symtab_slot = it.slot;
        d.keys = p.keys;
        d.validity = p.validity;
        d.prefix_keys = p.prefix_keys;
        d.fingerprints = p.fingerprints;
        d.residuals = p.residuals;
        d.fsst = p.fsst;
        d.shared_prefix = p.shared_prefix;
        d.signatures = reinterpret_cast<const uint64_t*>(ptr(7));
        d.postings = reinterpret_cast<const uint16_t*>(ptr(8));
        if (e.sig_on_device) sig_descs.push_back(d);
    }
    BvPackDesc* d_packs = reinterpret_cast<BvPackDesc*>(d_sc + sc_packs);
    LC_HIP(hipMemcpyAsync(d_packs, packs.data(), m * sizeof(BvPackDesc), hipMemcpyHostToDevice, side));
    LC_HIP(launch_bv_pack(d_descs, d_packs, uint32_t(m), side));
    if (!sig_descs.empty()) {
        const DevSymtab* d_st;
        {
            std::lock_guard<std::mutex> sg(ctx->st_mu);
            d_st = ctx->d_symtabs;
        }
        // the descriptors of the signature builder reuse the BvEncodeDesc area (k_bv_pack is ordered before it on the stream,
        // but still reads it): they go behind the stats instead
        StrDesc* d_sd = static_cast<StrDesc*>(pool_alloc(ctx, sig_descs.size() * sizeof(StrDesc)));
        if (!d_sd) return fail(LC_ERR_OOM, "hipMalloc (signature builder descriptors)");
        uint32_t max_d = 1;
        for (const StrDesc& sd : sig_descs) max_d = std::max(max_d, sd.d);
        const hipError_t e1 = hipMemcpyAsync(d_sd, sig_descs.data(), sig_descs.size() * sizeof(StrDesc), hipMemcpyHostToDevice, side);
        const hipError_t e2 = e1 == hipSuccess ? launch_str_build_signatures(d_sd, uint32_t(sig_descs.size()), max_d, d_st, side) : e1;
        const hipError_t e3 = hipStreamSynchronize(side);
        pool_release(ctx, d_sd);
        if (e2 != hipSuccess || e3 != hipSuccess) return fail(LC_ERR_DEVICE, "k_str_build_signatures failed");
    }
    LC_HIP(hipStreamSynchronize(side));
    for (size_t i = 0; i < m; i++) {
        // batch sizes over 8,192 rows (rare): the same lists the host transcoder attaches, from the entry's keys
        const BvItem& it = items[i];
        if (lay[i].off[8] == size_t(-1) || it.n <= kPostLdsRows) continue;
        std::vector<uint16_t> keys(it.n);
        std::vector<uint64_t> valid(it.has_validity ? (size_t(it.n) + 63) / 64 : 0);
        LC_HIP(hipMemcpyAsync(keys.data(), dbase + lay[i].off[0], size_t(it.n) * 2, hipMemcpyDeviceToHost, side));
        if (!valid.empty())
            LC_HIP(hipMemcpyAsync(valid.data(), dbase + lay[i].off[1], valid.size() * 8, hipMemcpyDeviceToHost, side));
        LC_HIP(hipStreamSynchronize(side));
        const std::vector<uint16_t> post = build_row_lists(keys.data(), valid.empty() ? nullptr : reinterpret_cast<const uint8_t*>(valid.data()),
                                                           it.n, stats[i].d);
        LC_HIP(hipMemcpyAsync(dbase + lay[i].off[8], post.data(), post.size() * 2, hipMemcpyHostToDevice, side));
        LC_HIP(hipStreamSynchronize(side));  // (`post` is a local)
    }
    std::unique_lock<std::shared_mutex> g(ctx->mu);
    for (size_t i = 0; i < m; i++) {
        publish_entry(ctx, items[i].id, std::move(entries[i]));
    }
    reserved.disarm();
    return LC_OK;
}

lc_status lc_insert_arrow_device(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, const struct ArrowArray* const* arrays,
                                 const struct ArrowSchema* const* schemas) {
    return lc_insert_arrow_batch_device(ctx, n, entry_ids, arrays, schemas, nullptr, nullptr);
}

lc_status lc_insert_arrow_batch_device(lc_ctx* ctx, uint64_t n_all, const uint64_t* ids_all, const struct ArrowArray* const* arrays_all,
                                       const struct ArrowSchema* const* schemas_all, const int32_t* hints, const uint64_t* path_ids) {
    return guarded([&]() -> lc_status {
    if (!ctx || (n_all && (!ids_all || !arrays_all || !schemas_all))) return fail(LC_ERR_INVALID, "null argument");
    if (ctx->device < 0) return fail(LC_ERR_DEVICE, "host-only context: no HIP device (there is no CPU fallback)");
    LC_HIP(hipSetDevice(ctx->device));
    // byte views (Utf8 / Binary) take their own encoder; everything else goes through the fixed-width one below
    std::vector<BvItem> views;
    std::vector<uint64_t> entry_ids_v;
    std::vector<const struct ArrowArray*> arrays_v;
    std::vector<const struct ArrowSchema*> schemas_v;
    for (uint64_t i = 0; i < n_all; i++) {
        const struct ArrowArray* a = arrays_all[i];
        const struct ArrowSchema* s = schemas_all[i];
        if (!a || !s) return fail(LC_ERR_INVALID, "null array");
        const std::string fmt = s->format ? s->format : "";
        if ((fmt == "u" || fmt == "z" || fmt == "vu" || fmt == "vz") && !s->dictionary) {
            if (a->length > 65536) return fail(LC_UNSUPPORTED, "byte-view entries of more than 65536 rows are not handled on the device");
            BvItem it;
            it.id = ids_all[i];
            it.a = a;
            it.views = fmt[0] == 'v';
            it.arrow_type = fmt == "u" ? kUtf8 : fmt == "z" ? kBinary : fmt == "vu" ? kUtf8View : kBinaryView;
            it.want_fp = hints && hints[i] == LC_HINT_SUBSTRING_SEARCH;
            it.path_id = path_ids ? path_ids[i] : 0;
            it.n = uint32_t(a->length);
            it.has_validity = a->n_buffers >= 1 && a->buffers[0] != nullptr;
            if (a->n_buffers < (it.views ? 2 : 3) || (it.n && !a->buffers[1]))
                return fail(LC_ERR_INVALID, "byte array without its offsets / views buffer");
            if (!it.views) {
                static const int32_t zero2[2] = {0, 0};
                const int32_t* o = a->buffers[1] ? static_cast<const int32_t*>(a->buffers[1]) + a->offset : zero2;
                // (the kernels take lengths as offset differences: a decreasing pair would send them out of the buffer)
                for (uint32_t r = 0; r < it.n; r++)
                    if (o[r + 1] < o[r] || o[r] < 0) return fail(LC_ERR_INVALID, "Utf8 / Binary offsets decrease");
                it.data_len = size_t(o[it.n] - o[0]);
                if (it.data_len && !a->buffers[2]) return fail(LC_ERR_INVALID, "Utf8 / Binary array without a data buffer");
            } else {
                const uint8_t* validity = it.has_validity ? static_cast<const uint8_t*>(a->buffers[0]) : nullptr;
                size_t total = 0;
                for (uint32_t r = 0; r < it.n; r++) {
                    if (validity && !get_bit(validity, size_t(a->offset) + r)) continue;
                    const uint8_t* view = static_cast<const uint8_t*>(a->buffers[1]) + 16 * (size_t(a->offset) + r);
                    const uint32_t len = rd<uint32_t>(view);
                    if (len > 12) {
                        // C Data Interface: buffers = [validity, views, data 0 .. data k-1, variadic sizes (i64 x k)] — the
                        // LAST buffer holds the data buffers' sizes and is not a data buffer itself
                        const int32_t buf = rd<int32_t>(view + 8);
                        const int32_t off = rd<int32_t>(view + 12);
                        if (buf < 0 || int64_t(buf) + 2 >= a->n_buffers - 1 || !a->buffers[2 + buf] || !a->buffers[a->n_buffers - 1])
                            return fail(LC_ERR_INVALID, "view refers to a data buffer the array does not have");
                        const int64_t size = static_cast<const int64_t*>(a->buffers[a->n_buffers - 1])[buf];
                        if (off < 0 || int64_t(off) + int64_t(len) > size)
                            return fail(LC_ERR_INVALID, "view reaches beyond its data buffer");
                    }
                    total += len;
                }
                if (total > size_t(INT32_MAX)) return fail(LC_UNSUPPORTED, "more than 2 GiB of bytes in one batch");
                it.data_len = total;
            }
            views.push_back(it);
        } else {
            entry_ids_v.push_back(ids_all[i]);
            arrays_v.push_back(a);
            schemas_v.push_back(s);
        }
    }
    const uint64_t n = entry_ids_v.size();
    const uint64_t* entry_ids = entry_ids_v.data();
    const struct ArrowArray* const* arrays = arrays_v.data();
    const struct ArrowSchema* const* schemas = schemas_v.data();
    std::vector<DevEncodeItem> items(n);
    size_t stage_bytes = 0;
    for (uint64_t i = 0; i < n; i++) {
        const struct ArrowArray* a = arrays[i];
        const struct ArrowSchema* s = schemas[i];
        if (!a || !s) return fail(LC_ERR_INVALID, "null array");
        int phys = int_phys_of_format(s->format);
        if (phys < 0 && s->format && (std::string(s->format) == "f" || std::string(s->format) == "g"))
            phys = s->format[0] == 'f' ? kF32 : kF64;
        int dec_p = 0, dec_s = 0, dec_bits = 128;
        const bool is_decimal = s->format && s->format[0] == 'd' && s->format[1] == ':' &&
                                std::sscanf(s->format, "d:%d,%d,%d", &dec_p, &dec_s, &dec_bits) >= 2 &&
                                (dec_bits == 128 || dec_bits == 256);
        if ((phys < 0 && !is_decimal) || s->dictionary || a->length > int64_t(UINT32_MAX))
            return fail(LC_UNSUPPORTED, "the on-device transcoder takes integer / date / timestamp / decimal / float / Utf8 / Binary arrays (use lc_insert_arrow)");
        DevEncodeItem& it = items[i];
        it.id = entry_ids[i];
        if (is_decimal) {  // LiquidDecimalArray: u64 offsets of the unscaled values (decimal_array.rs:127-177)
            phys = kU64;
            it.logical = kDecimal;
            it.dec_precision = dec_p;
            it.dec_scale = dec_s;
            it.dec_is256 = dec_bits == 256;
            it.entry_value_width = dec_bits / 8;
            it.in_stride = dec_bits / 8;
        }
        it.phys = phys;
        it.vw = phys_width(phys);
        it.is_signed = !phys_unsigned(phys);
        it.is_float = phys == kF32 || phys == kF64;
        if (it.is_float) { it.logical = kFloat; it.is_signed = true; }
        it.n = uint32_t(a->length);
        it.has_validity = a->n_buffers >= 1 && a->buffers[0] != nullptr;
        it.in_values = align_up(stage_bytes, 16);
        stage_bytes = it.in_values + size_t(it.n) * size_t(it.in_stride ? it.in_stride : it.vw) + 32;
        if (it.has_validity) {
            it.in_validity = align_up(stage_bytes, 16);
            stage_bytes = it.in_validity + ((size_t(it.n) + 63) / 64) * 8;
        }
    }
    {
        // every array of the call has been classified and validated by now: a call that is going to answer LC_UNSUPPORTED /
        // LC_ERR_INVALID for one of its arrays has published nothing (round-3 advisor finding)
        const lc_status rc = device_encode_byte_views(ctx, views);
        if (rc != LC_OK) return rc;
    }
    if (n == 0) return LC_OK;
    stage_bytes = align_up(stage_bytes, 256);
    uint8_t* h = static_cast<uint8_t*>(host_pool_alloc(ctx, stage_bytes));
    uint8_t* d_in = static_cast<uint8_t*>(pool_alloc(ctx, stage_bytes));
    struct Bufs {
        lc_ctx* c; void* h; void* d;
        ~Bufs() { (void)hipStreamSynchronize(nullptr); host_pool_release(c, h); pool_release(c, d); }
    } bufs{ctx, h, d_in};
    if (!h || !d_in) return fail(LC_ERR_OOM, "staging buffers of the on-device transcoder");
    for (uint64_t i = 0; i < n; i++) {
        const DevEncodeItem& it = items[i];
        const struct ArrowArray* a = arrays[i];
        const uint8_t* v = static_cast<const uint8_t*>(a->buffers[1]);
        const size_t vstride = size_t(it.in_stride ? it.in_stride : it.vw);
        if (v) std::memcpy(h + it.in_values, v + size_t(a->offset) * vstride, size_t(it.n) * vstride);
        else std::memset(h + it.in_values, 0, size_t(it.n) * vstride);
        if (it.has_validity) {
            const size_t words = (size_t(it.n) + 63) / 64;
            std::memset(h + it.in_validity, 0, words * 8);
            const std::vector<uint8_t> bm = [&]() {
                std::vector<uint8_t> out(bitmap_bytes(it.n) + 1, 0);
                const uint8_t* src = static_cast<const uint8_t*>(a->buffers[0]);
                if ((a->offset & 7) == 0) {
                    std::memcpy(out.data(), src + (a->offset >> 3), bitmap_bytes(it.n));
                    if (it.n & 7) out[bitmap_bytes(it.n) - 1] &= uint8_t((1u << (it.n & 7)) - 1);
                } else {
                    for (size_t k = 0; k < it.n; k++)
                        if (get_bit(src, size_t(a->offset) + k)) set_bit(out.data(), k);
                }
                return out;
            }();
            std::memcpy(h + it.in_validity, bm.data(), bitmap_bytes(it.n));
            items[i].n_valid = uint32_t(count_bits(bm.data(), it.n));
        } else {
            items[i].n_valid = it.n;
        }
    }
    LC_HIP(hipMemcpy(d_in, h, stage_bytes, hipMemcpyHostToDevice));
    std::vector<DevEncodeItem> ints, floats;
    for (DevEncodeItem& it : items) {
        it.d_values = d_in + it.in_values;
        it.d_validity = reinterpret_cast<const uint64_t*>(d_in + it.in_validity);
        (it.is_float ? floats : ints).push_back(it);
    }
    lc_status rc = device_encode_and_register(ctx, ints);
    if (rc != LC_OK) return rc;
    return device_encode_floats(ctx, floats);
    });
}

// LiquidArray::to_bytes() of a staged fixed-width entry (primitive_array.rs:603-679, decimal_array.rs:197-220,
// float_array.rs:397-519, bit_pack_array.rs:181-256): what the reference writes to its disk tier when it squeezes or
// evicts an entry, rebuilt from the HBM-resident form.
lc_status lc_entry_index_to_bytes(lc_ctx* ctx, uint64_t entry_id, uint8_t** out_bytes, size_t* out_len) {
    return guarded([&]() -> lc_status {
    if (!ctx || !out_bytes || !out_len) return fail(LC_ERR_INVALID, "null argument");
    if (ctx->device < 0) return fail(LC_ERR_DEVICE, "host-only context: no HIP device (there is no CPU fallback)");
    LC_HIP(hipSetDevice(ctx->device));
    *out_bytes = nullptr;
    *out_len = 0;
    Entry e;
    int pinned_slab = -1;
    if (!pin_entry(ctx, entry_id, &e, &pinned_slab)) return LC_NOT_STAGED;
    SlabPin pin(ctx, pinned_slab);
    if (!e.is_str || (!e.sd.signatures && !e.sd.postings)) return LC_OK;  // nothing to keep: *out_len == 0
    IndexHeader h{};
    h.magic = kIndexMagic;
    h.version = kIndexVersion;
    h.d = e.sd.d;
    h.n = e.sd.n;
    h.sig_bits = uint32_t(kSigBits);
    h.content_hash = e.index_hash;
    if (h.content_hash == 0) {
        // an entry encoded on the device (no host bytes at staging): hash the same sections read back from its blob
        const size_t kb = size_t(e.sd.n) * 2, vb = e.sd.validity ? ((size_t(e.sd.n) + 63) / 64) * 8 : 0;
        std::vector<uint8_t> tmp(kb + vb + e.offsets_bytes + e.fsst_len + 8);
        uint8_t* pk = tmp.data(), *pv = pk + kb, *pr = pv + vb, *pf = pr + e.offsets_bytes;
        const SymbolTable* st = nullptr;
        {
            std::lock_guard<std::mutex> g(ctx->st_mu);
            if (e.sd.symtab_slot < ctx->symtabs.size()) st = ctx->symtabs[e.sd.symtab_slot].get();
        }
        if (!st) return fail(LC_ERR_NO_SYMTAB, "entry refers to an unknown symbol table");
        if ((kb && hipMemcpy(pk, e.sd.keys, kb, hipMemcpyDeviceToHost) != hipSuccess) ||
            (vb && hipMemcpy(pv, e.sd.validity, vb, hipMemcpyDeviceToHost) != hipSuccess) ||
            (e.offsets_bytes && hipMemcpy(pr, e.sd.residuals, e.offsets_bytes, hipMemcpyDeviceToHost) != hipSuccess) ||
            (e.fsst_len && hipMemcpy(pf, e.sd.fsst, e.fsst_len, hipMemcpyDeviceToHost) != hipSuccess))
            return fail(LC_ERR_DEVICE, "hipMemcpy (entry index hash)");
        h.content_hash = index_content_hash(pk, kb, pv, vb, pr, e.offsets_bytes, pf, e.fsst_len, e.sd.slope, e.sd.intercept, *st);
    }
    const size_t nw = std::max<size_t>((size_t(e.sd.d) + 63) / 64, 1);
    if (e.sd.signatures) { h.flags |= 1u; h.sig_bytes = size_t(kSigBits) * nw * 8; }
    if (e.sd.postings) { h.flags |= 2u; h.post_bytes = (size_t(e.sd.d) + 1 + size_t(e.sd.n) + 32) * 2; }
    const size_t total = sizeof(h) + h.sig_bytes + h.post_bytes;
    uint8_t* buf = static_cast<uint8_t*>(std::malloc(total));
    if (!buf) return fail(LC_ERR_OOM, "malloc");
    std::memcpy(buf, &h, sizeof(h));
    hipError_t e1 = hipSuccess, e2 = hipSuccess;
    if (h.sig_bytes) e1 = hipMemcpy(buf + sizeof(h), e.sd.signatures, h.sig_bytes, hipMemcpyDeviceToHost);
    if (h.post_bytes) e2 = hipMemcpy(buf + sizeof(h) + h.sig_bytes, e.sd.postings, h.post_bytes, hipMemcpyDeviceToHost);
    if (e1 != hipSuccess || e2 != hipSuccess) { std::free(buf); return fail(LC_ERR_DEVICE, "hipMemcpy (entry index)"); }
    *out_bytes = buf;
    *out_len = total;
    return LC_OK;
    });
}

lc_status lc_entry_to_liquid_bytes(lc_ctx* ctx, uint64_t entry_id, uint8_t** out_bytes, size_t* out_len) {
    return guarded([&]() -> lc_status {
    if (!ctx || !out_bytes || !out_len) return fail(LC_ERR_INVALID, "null argument");
    if (ctx->device < 0) return fail(LC_ERR_DEVICE, "host-only context: no HIP device (there is no CPU fallback)");
    LC_HIP(hipSetDevice(ctx->device));
    // This is the evict-to-disk path, so an lc_evict / re-stage of the same entry on another thread is expected: the slab is
    // pinned for the duration of the read-back (a concurrent eviction then only drops the entry, not the bytes under us).
    Entry e;
    int pinned_slab = -1;
    if (!pin_entry(ctx, entry_id, &e, &pinned_slab)) return LC_NOT_STAGED;
    SlabPin pin(ctx, pinned_slab);
    if (e.is_str) {
        // LiquidByteViewArray::to_bytes (byte_view_array/serialization.rs:122-220): header, raw FSST buffer, keys as a
        // BitPackedArray<u16> at 16 bits, compact offsets, prefix keys, shared prefix, fingerprints — read back from HBM
        const StrDesc& d = e.sd;
        auto fetch = [&](const void* src, size_t n, std::vector<uint8_t>& dst) -> bool {
            dst.assign(n, 0);
            return n == 0 || hipMemcpy(dst.data(), src, n, hipMemcpyDeviceToHost) == hipSuccess;
        };
        std::vector<uint8_t> keys_b, valid_b, pk_b, fp_b, res_b, fsst_b, sp_b;
        if (!fetch(d.keys, size_t(d.n) * 2, keys_b) || !fetch(d.prefix_keys, size_t(d.d) * 8, pk_b) ||
            !fetch(d.residuals, e.offsets_bytes, res_b) || !fetch(d.fsst, d.fsst_len, fsst_b) ||
            !fetch(d.shared_prefix, d.shared_prefix_len, sp_b) ||
            !fetch(d.validity, d.validity ? ((size_t(d.n) + 63) / 64) * 8 : 0, valid_b) ||
            !fetch(d.fingerprints, d.fingerprints ? size_t(d.d) * 4 : 0, fp_b))
            return fail(LC_ERR_DEVICE, "hipMemcpy (entry sections)");
        std::vector<uint8_t> out(40, 0);
        auto pad8 = [&]() { while (out.size() & 7) out.push_back(0); };
        const size_t fsst_start = out.size();
        out.resize(fsst_start + 12 + fsst_b.size());
        wr<uint64_t>(out.data() + fsst_start, e.raw_bytes);
        wr<uint32_t>(out.data() + fsst_start + 8, d.fsst_len);
        if (!fsst_b.empty()) std::memcpy(out.data() + fsst_start + 12, fsst_b.data(), fsst_b.size());
        const uint32_t fsst_raw_size = uint32_t(out.size() - fsst_start);
        pad8();
        const size_t keys_start = out.size();
        // the reference always writes the keys at 16 bits (serialization.rs:141-150), all-null arrays included
        append_bitpacked<uint16_t>(out, 16, reinterpret_cast<const uint16_t*>(keys_b.data()),
                                   d.validity ? valid_b.data() : nullptr, d.n);
        const uint32_t keys_size = uint32_t(out.size() - keys_start);
        pad8();
        const size_t co_start = out.size();
        if (e.offsets_bytes) {
            out.resize(co_start + 9 + res_b.size());
            wr<int32_t>(out.data() + co_start, d.slope);
            wr<int32_t>(out.data() + co_start + 4, d.intercept);
            out[co_start + 8] = d.offset_bytes;
            std::memcpy(out.data() + co_start + 9, res_b.data(), res_b.size());
        }
        const uint32_t co_size = uint32_t(out.size() - co_start);
        pad8();
        out.insert(out.end(), pk_b.begin(), pk_b.end());
        pad8();
        out.insert(out.end(), sp_b.begin(), sp_b.end());
        pad8();
        out.insert(out.end(), fp_b.begin(), fp_b.end());
        write_ipc_header(out.data(), kByteView, e.phys);
        wr<uint32_t>(out.data() + 16, keys_size);
        wr<uint32_t>(out.data() + 20, co_size);
        wr<uint32_t>(out.data() + 24, d.shared_prefix_len);
        wr<uint32_t>(out.data() + 28, fsst_raw_size);
        wr<uint32_t>(out.data() + 32, uint32_t(fp_b.size()));
        uint8_t* buf = static_cast<uint8_t*>(std::malloc(std::max<size_t>(out.size(), 1)));
        if (!buf) return fail(LC_ERR_OOM, "malloc");
        std::memcpy(buf, out.data(), out.size());
        *out_bytes = buf;
        *out_len = out.size();
        return LC_OK;
    }
    if (e.squeezed_field >= 0) return fail(LC_NEEDS_BACKING, "a squeezed entry holds one date component only");
    if (e.clamped) return fail(LC_NEEDS_BACKING, "a clamp-squeezed entry holds half of its bits");
    if (e.quantized) return fail(LC_NEEDS_BACKING, "a quantize-squeezed entry holds bucket indices only");
    std::vector<uint8_t> out(16, 0);
    write_ipc_header(out.data(), e.logical, e.phys);
    const FixedDesc& d = e.fd;
    auto pad8 = [&]() { while (out.size() & 7) out.push_back(0); };
    if (e.logical == kInteger) {
        out.resize(24, 0);
        if (!e.all_null) std::memcpy(out.data() + 16, &d.reference, size_t(d.value_width));
    } else if (e.logical == kDecimal) {
        out.resize(32, 0);
        out[16] = uint8_t(e.dec_is256);
        out[17] = uint8_t(e.dec_precision);
        out[18] = uint8_t(int8_t(e.dec_scale));
        if (!e.all_null) std::memcpy(out.data() + 24, &d.reference, 8);
    } else {  // ALP float
        const size_t w = d.value_width;
        out.resize(16 + w, 0);
        if (!e.all_null) std::memcpy(out.data() + 16, &d.reference, w);
        pad8();
        out.push_back(d.alp_e);
        out.push_back(d.alp_f);
        out.resize(out.size() + 6, 0);
        const uint64_t pl = d.patch_len;
        const size_t o = out.size();
        out.resize(o + 8 + pl * 8 + pl * w, 0);
        std::memcpy(out.data() + o, &pl, 8);
        if (pl) {
            LC_HIP(hipMemcpy(out.data() + o + 8, d.patch_idx, pl * 8, hipMemcpyDeviceToHost));
            LC_HIP(hipMemcpy(out.data() + o + 8 + pl * 8, d.patch_val, pl * w, hipMemcpyDeviceToHost));
        }
        pad8();
    }
    // BitPackedArray section (:181-256)
    const size_t n = e.len, lane_bytes = size_t(1) << (d.lane_log2 - 3);
    const bool has_nulls = e.nullable;
    const size_t nulls_len = has_nulls ? bitmap_bytes(n) : 0;
    const size_t values_len = e.all_null ? n * lane_bytes : packed_bytes(e.W, n);
    const size_t start = out.size(), values_off = align8(16 + nulls_len);
    out.resize(start + values_off + values_len, 0);
    uint8_t* p = out.data() + start;
    wr<uint32_t>(p, uint32_t(n));
    p[4] = uint8_t(e.all_null ? 0 : e.W);
    p[5] = has_nulls ? 1 : 0;
    wr<uint32_t>(p + 6, uint32_t(nulls_len));
    wr<uint32_t>(p + 10, uint32_t(values_len));
    if (has_nulls && nulls_len && !e.all_null) {
        if (d.validity) LC_HIP(hipMemcpy(p + 16, d.validity, nulls_len, hipMemcpyDeviceToHost));
        else std::memset(p + 16, 0xFF, nulls_len);
        if (n & 7) p[16 + nulls_len - 1] &= uint8_t((1u << (n & 7)) - 1);
    }
    if (!e.all_null && values_len) LC_HIP(hipMemcpy(p + values_off, d.packed, values_len, hipMemcpyDeviceToHost));
    *out_bytes = static_cast<uint8_t*>(std::malloc(out.size() ? out.size() : 1));
    if (!*out_bytes) return fail(LC_ERR_OOM, "malloc");
    std::memcpy(*out_bytes, out.data(), out.size());
    *out_len = out.size();
    return LC_OK;
    });
}

// ------------------------------------------------------------------ scans

#ifdef LC_CALL_PROFILE
#include <chrono>
static std::atomic<uint64_t> g_prof_ns[16];
static std::atomic<uint64_t> g_prof_calls{0};
struct ProfDump { ~ProfDump() {
    const char* names[16] = {"checkout", "alloc", "eval launch", "compress/copy launch", "sync", "unpack", "release", "-",
                             "create:lock", "create:alloc", "create:copy+sync", "create:symtabs", "destroy", "stream_acquire", "-", "-"};
    std::fprintf(stderr, "lc_eval_predicate_batch profile over %llu calls:", (unsigned long long)g_prof_calls.load());
    for (int i = 0; i < 14; i++) std::fprintf(stderr, " %s %.1f us;", names[i], double(g_prof_ns[i].load()) / 1e3 / double(std::max<uint64_t>(g_prof_calls.load(), 1)));
    std::fprintf(stderr, "\n"); } } g_prof_dump;
#define LC_PROF_T0 auto _pt = std::chrono::steady_clock::now()
#define LC_PROF(i) do { auto _n = std::chrono::steady_clock::now(); g_prof_ns[i] += uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(_n - _pt).count()); _pt = _n; } while (0)
#else
#define LC_PROF_T0
#define LC_PROF(i)
#endif
static lc_status scan_create_impl(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, lc_scan** out, bool allow_squeezed,
                                  hipStream_t st = nullptr);

// lc_scan_create of a list seen before is O(1) on the device and a memcmp on the host: the scan that lc_scan_destroy gave back
// (descriptors, workgroup records, automata, the LIKE pipeline's records and plans — everything but the caller's handle) is
// handed out again as long as none of its entries has been replaced or evicted (publish_entry / lc_evict move the scans that
// hold the id to the graveyard).  The reference's reader names entries per query and keeps no scan objects
// (liquid_cache_reader.rs:264-339): a host that follows it pays the 0.3-1.4 ms of a cold creation once per list, not per query.
lc_status lc_scan_create(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, lc_scan** out) {
    LC_PHASE("lc_scan_create");
    if (ctx && out && entry_ids && n > 0 && ctx->scan_cache_max.load() > 0) {
        lc_scan* hit = nullptr;
        bool reap = false;
        {
            std::lock_guard<std::mutex> g(ctx->scan_cache_mu);
            reap = !ctx->scan_graveyard.empty();
            for (size_t i = ctx->list_cache.size(); i-- > 0;) {
                lc_scan* c = ctx->list_cache[i];
                if (c->ids.size() != n || c->ids[0] != entry_ids[0] || c->ids[n - 1] != entry_ids[n - 1] ||
                    std::memcmp(c->ids.data(), entry_ids, size_t(n) * 8) != 0)
                    continue;
                hit = c;
                ctx->list_cache.erase(ctx->list_cache.begin() + long(i));
                break;
            }
        }
        if (reap) scan_cache_reap(ctx);
        if (hit) {
            *out = hit;
            return LC_OK;
        }
    }
    const lc_status st = scan_create_impl(ctx, n, entry_ids, out, false);
    if (st == LC_OK && *out && n > 0) {
        lc_scan* s = *out;
        try {
            s->ids.assign(entry_ids, entry_ids + n);
            s->id_bloom.assign(kIdBloomBits / 64, 0);
            for (uint64_t i = 0; i < n; i++) id_bloom_add(s->id_bloom, entry_ids[i]);
            s->cacheable = true;
        } catch (...) {
            s->cacheable = false;
        }
    }
    return st;
}

static lc_status scan_create_impl(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, lc_scan** out, bool allow_squeezed,
                                  hipStream_t stream) {
    return guarded([&]() -> lc_status {
    if (!ctx || !out || (n && !entry_ids)) return fail(LC_ERR_INVALID, "null argument");
    *out = nullptr;
    if (n > 0xFFFFFFFFull) return fail(LC_ERR_INVALID, "too many entries in one scan");
    if (ctx->device < 0) return fail(LC_ERR_DEVICE, "host-only context: no HIP device (there is no CPU fallback)");
    LC_HIP(hipSetDevice(ctx->device));
    LC_PROF_T0;
    LC_PHASE("scan_create (all)");
    std::unique_ptr<lc_scan> s(new lc_scan());
    s->ctx = ctx;
    s->n = uint32_t(n);
    s->seg_offsets.assign(n + 1, 0);
    s->meta.reserve(n);
    s->lens.reserve(n);
    s->uids.reserve(n);
    {
        LC_PHASE("scan_create: capture entries");
        // ONE critical section captures the entries and pins their slabs: between a capture under one lock and a pin
        // under another an lc_evict / re-stage could drain the slab and the scan would keep dangling device pointers.
        // The pins are taken after every entry has validated, so the error returns below leave nothing pinned.
        std::unique_lock<std::shared_mutex> g(ctx->mu);
        s->evict_epoch = ctx->evict_epoch.load();
        uint32_t max_len = 0;
        lc_status bad = LC_OK;
        // (the slot of id i + 16 and the record of id i + 8 are prefetched while id i is copied: EntryMap::visit_many)
        ctx->entries.visit_many(entry_ids, n, [&](size_t i, EntryMap::value_type* node) -> bool {
            if (!node) { bad = fail(LC_NOT_STAGED, "entry is not staged"); return false; }
            s->meta.push_back(node->second);  // (one copy of the ~400-byte entry record, edited in place)
            Entry& e = s->meta.back();
            if (e.squeezed_field >= 0 && !allow_squeezed) {
                bad = fail(LC_NEEDS_BACKING, "entry is squeezed to one date component: predicates and plain reads need the "
                                             "full array from the disk tier");
                return false;
            }
            if (i == 0) {
                s->is_str = e.is_str;
                s->lane_log2 = e.is_str ? 4 : e.fd.lane_log2;
            } else if (e.is_str != s->is_str || (!e.is_str && e.fd.lane_log2 != s->lane_log2)) {
                bad = fail(LC_ERR_INVALID, "a scan covers entries of ONE column (same encoding and lane width)");
                return false;
            }
            const uint64_t off = s->seg_offsets[i];
            if (e.is_str) e.sd.mask_word_off = off;
            else e.fd.mask_word_off = off;
            s->seg_offsets[i + 1] = off + (uint64_t(e.len) + 63) / 64;
            s->lens.push_back(e.len);
            s->uids.push_back(e.uid);
            s->entry_bytes_total += e.device_bytes;
            if (s->slab_pins.empty() || s->slab_pins.back().first != e.slab) {  // (entries of a column sit in runs of one slab)
                size_t k = 0;
                while (k < s->slab_pins.size() && s->slab_pins[k].first != e.slab) k++;
                if (k == s->slab_pins.size()) s->slab_pins.emplace_back(e.slab, 0u);
                if (k + 1 != s->slab_pins.size()) std::swap(s->slab_pins[k], s->slab_pins.back());
            }
            s->slab_pins.back().second++;
            s->total_rows += e.len;
            max_len = std::max(max_len, e.len);
            if (!e.is_str) {
                s->max_w = std::max<uint32_t>(s->max_w, uint32_t(e.W));
                if (e.W > 0) s->min_w = s->min_w ? std::min<uint32_t>(s->min_w, uint32_t(e.W)) : uint32_t(e.W);
            }
            s->has_clamped |= e.clamped || e.quantized;
            s->has_fquant |= e.fq_shift > 0;
            s->fquant_patches |= e.fq_shift > 0 && e.fd.patch_len > 0;
            s->max_dict_len = std::max(s->max_dict_len, e.dict_len);
            s->any_fingerprints |= e.has_fp;
            if (e.is_str) {
                s->any_without_signatures |= e.sd.signatures == nullptr && e.sd.d > 0;  // (an all-null entry has no dictionary)
                s->any_multi_empty |= e.sd.multi_empty != 0;
                if (i == 0) s->uniform_slot = int32_t(e.sd.symtab_slot);
                else if (int32_t(e.sd.symtab_slot) != s->uniform_slot) s->uniform_slot = -1;
                s->symtab_slots.push_back(e.sd.symtab_slot);
                s->slot_lo = (i == 0) ? e.sd.symtab_slot : std::min(s->slot_lo, e.sd.symtab_slot);
                s->slot_hi = (i == 0) ? e.sd.symtab_slot : std::max(s->slot_hi, e.sd.symtab_slot);
                if (e.sd.d != 0 && (!e.sd.signatures || !e.sd.postings || !e.sd.fingerprints)) s->str_index_everywhere = false;
                s->max_str_rows = std::max(s->max_str_rows, e.sd.n);
                if (e.sd.d != 0) s->max_dict_rows = std::max(s->max_dict_rows, e.sd.n);
            } else {
                s->any_patch |= (e.fd.kind == kKindF32 || e.fd.kind == kKindF64) && e.fd.patch_len > 0;
                s->any_float |= e.fd.kind == kKindF32 || e.fd.kind == kKindF64;
            }
            return true;
        });
        if (bad != LC_OK) return bad;
        s->bpe = std::max<uint32_t>(1, (max_len + 1023) / 1024);
        // pin the slabs of the scan's entries: evicting or re-staging an entry under a live scan is then safe (the scan
        // keeps the blob it captured; lc_scan_destroy drops the pins)
        arena_pin_counts(ctx, s->slab_pins);
        s->pinned = true;
    }
    LC_PROF(8);
    const size_t desc_size = s->is_str ? sizeof(StrDesc) : sizeof(FixedDesc);
    // pinned staging: an asynchronous copy from pageable memory makes the runtime pin the pages for the duration of the
    // call — a process-wide lock that eight concurrent callers queue on (measured: 20 us alone, 1.2 ms with eight)
    const size_t desc_bytes = desc_size * std::max<uint64_t>(n, 1), seg_bytes = (n + 1) * 8;
    struct Pinned {
        lc_ctx* c; uint8_t* p;
        ~Pinned() { host_pool_release(c, p); }
    } host{ctx, static_cast<uint8_t*>(host_pool_alloc(ctx, desc_bytes + seg_bytes))};
    for (uint64_t i = 0; i < n && host.p; i++) {
        if (s->is_str) std::memcpy(host.p + i * desc_size, &s->meta[i].sd, desc_size);
        else std::memcpy(host.p + i * desc_size, &s->meta[i].fd, desc_size);
    }
    if (host.p) std::memcpy(host.p + desc_bytes, s->seg_offsets.data(), seg_bytes);
    s->d_descs = pool_alloc(ctx, desc_bytes);
    s->d_seg_offsets = static_cast<uint64_t*>(pool_alloc(ctx, seg_bytes));
    lc_status st = (!s->d_descs || !s->d_seg_offsets || !host.p) ? fail(LC_ERR_OOM, "hipMalloc (scan descriptors)") : LC_OK;
    // on the caller's stream (a pooled one for the per-entry calls): the null stream would order this call behind the
    // null-stream work of every other host thread
    LC_PROF(9);
    if (st == LC_OK && hipMemcpyAsync(s->d_descs, host.p, desc_bytes, hipMemcpyHostToDevice, stream) != hipSuccess)
        st = fail(LC_ERR_DEVICE, "hipMemcpy (scan descriptors)");
    if (st == LC_OK &&
        hipMemcpyAsync(s->d_seg_offsets, host.p + desc_bytes, seg_bytes, hipMemcpyHostToDevice, stream) != hipSuccess)
        st = fail(LC_ERR_DEVICE, "hipMemcpy (scan offsets)");
    if (st == LC_OK && hipStreamSynchronize(stream) != hipSuccess) st = fail(LC_ERR_DEVICE, "hipStreamSynchronize (scan descriptors)");
    if (st == LC_OK) s->streams_used.push_back(stream);
    LC_PROF(10);
    if (st == LC_OK) st = sync_symtabs(ctx);
    LC_PROF(11);
    if (st == LC_OK) {
        std::lock_guard<std::mutex> g(ctx->st_mu);
        s->d_symtabs = ctx->d_symtabs;
        s->n_symtabs = ctx->d_symtabs_uploaded;
    }
    if (st != LC_OK) {
        pool_release(ctx, s->d_descs);
        pool_release(ctx, s->d_seg_offsets);
        std::unique_lock<std::shared_mutex> g(ctx->mu);
        arena_release_counts(ctx, s->slab_pins);
        return st;
    }
    *out = s.release();
    return LC_OK;
    });
}

void lc_scan_destroy(lc_scan* s) {
    if (!s) return;
    LC_PHASE("lc_scan_destroy");
    lc_ctx* ctx = s->ctx;
    if (s->cacheable && ctx->scan_cache_max.load() > 0) {
        // kept for the next lc_scan_create over the same list.  What the caller may rely on stays true: nothing of this scan is
        // in flight on the caller's streams when the call returns (they are drained); the scan-level LIKE index — or the build
        // of it that the builder thread is still running — stays with the kept scan, where the index budget can reclaim it.
        try {
            (void)hipSetDevice(ctx->device);
            std::vector<hipStream_t> used;
            {
                // (the kept scan forgets the streams: the caller may destroy them once its scans are gone — "a stream must
                // outlive the scans it was used with" — and the next holder of this scan brings its own)
                std::lock_guard<std::mutex> g(s->mu);
                used.swap(s->streams_used);
                s->last_stream = nullptr;
                s->used = false;
            }
            for (hipStream_t st : used) (void)hipStreamSynchronize(st);
            // (an index build in flight for this scan goes on: the kept scan stays alive for the builder, which a scan that is
            // really destroyed waits for; index_reserve leaves pipelines with a job in flight alone)
            ctx->index_events++;  // (whatever index this scan holds is reclaimable from now on)
            std::vector<lc_scan*> out;
            bool kept = false;
            {
                // (under ctx->mu shared: an eviction that concerns this scan either ran before — then the entries' uids differ
                // and the scan is not kept — or finds it in the cache afterwards)
                std::shared_lock<std::shared_mutex> gc(ctx->mu);
                // nothing was replaced or evicted since the scan captured its entries: it is current.  Otherwise its entries'
                // publication ids decide (O(n), only after an eviction somewhere in the cache)
                bool current = true;
                const uint64_t epoch = ctx->evict_epoch.load();
                if (s->evict_epoch != epoch) {
                    for (size_t k = 0; current && k < s->ids.size(); k++) {
                        auto it = ctx->entries.find(s->ids[k]);
                        current = it != ctx->entries.end() && it->second.uid == s->meta[k].uid;
                    }
                    if (current) s->evict_epoch = epoch;
                }
                if (current) {
                    std::lock_guard<std::mutex> g(ctx->scan_cache_mu);
                    ctx->list_cache.push_back(s);
                    kept = true;
                    while (ctx->list_cache.size() > size_t(ctx->scan_cache_max.load())) {
                        out.push_back(ctx->list_cache.front());
                        ctx->list_cache.erase(ctx->list_cache.begin());
                    }
                }
            }
            for (lc_scan* q : out) scan_destroy_now(q);
            if (kept) return;
        } catch (...) {
        }
    }
    scan_destroy_now(s);
}

static void scan_destroy_now(lc_scan* s) {
    if (!s) return;
    LC_PROF_T0;
    try {
    (void)hipSetDevice(s->ctx->device);
    // nothing in flight may still read the scan's buffers when they are recycled: drain the streams its launches went to
    // (not the device — the calls of other host threads keep running)
    for (hipStream_t st : s->streams_used) (void)hipStreamSynchronize(st);
    if (s->streams_used.empty()) (void)hipStreamSynchronize(nullptr);
    pool_release(s->ctx, s->d_descs);
    pool_release(s->ctx, s->d_seg_offsets);
    pool_release(s->ctx, s->d_work);
    pool_release(s->ctx, s->d_gather);
    pool_release(s->ctx, s->d_wg_ranges);
    pool_release(s->ctx, s->d_total_acc);
    pool_release(s->ctx, s->d_mask_scratch);
    pool_release(s->ctx, s->d_or_tmp);
    pool_release(s->ctx, s->d_agg_partials);
    pool_release(s->ctx, s->d_wg_begins);
    host_pool_release(s->ctx, s->h_wg_begins);
    pool_release(s->ctx, s->d_group_ends);
    pool_release(s->ctx, s->d_group_entry_counts);
    like_pipeline_orphan(s->ctx, s->like);
    pool_release(s->ctx, s->d_automata);
    pool_release(s->ctx, s->d_needle);
    if (s->pinned) {
        std::unique_lock<std::shared_mutex> g(s->ctx->mu);
        arena_release_counts(s->ctx, s->slab_pins);
    }
    delete s;
    } catch (...) {
    }
    LC_PROF(12);
}

lc_status lc_scan_info_get(lc_scan* s, lc_scan_info* out) {
    return guarded([&]() -> lc_status {
    if (!s || !out) return fail(LC_ERR_INVALID, "null argument");
    std::memset(out, 0, sizeof(*out));
    out->entries = s->n;
    out->rows = s->total_rows;
    out->mask_words = s->seg_offsets.back();
    out->is_byte_view = s->is_str ? 1 : 0;
    out->max_bit_width = s->is_str ? 16 : int32_t(s->max_w);
    out->entry_bytes = s->entry_bytes_total;
    {  // (never nested inside the scan's lock: one order of locks everywhere)
        std::shared_lock<std::shared_mutex> gc(s->ctx->mu);
        out->ctx_slab_bytes = s->ctx->staged_bytes;
    }
    std::lock_guard<std::mutex> g(s->mu);
    out->ctx_index_bytes = s->ctx->index_bytes.load();
    uint32_t plans = 0;
    like_pipeline_info(s, &out->index_bytes, &out->unigram_index_bytes, &out->index_build_ms, &plans, &out->index_build_pending);
    out->like_plans = plans;
    out->last_like_kernel = s->last_like_kernel;
    return LC_OK;
    });
}

lc_status lc_scan_index_wait(lc_scan* s) {
    return guarded([&]() -> lc_status {
    if (!s) return fail(LC_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> g(s->mu);
    like_pipeline_wait(s);
    return LC_OK;
    });
}

uint64_t lc_scan_mask_words(const lc_scan* s) { return s ? s->seg_offsets.back() : 0; }
uint64_t lc_scan_rows(const lc_scan* s) { return s ? s->total_rows : 0; }
uint64_t lc_scan_entries(const lc_scan* s) { return s ? s->n : 0; }
const uint64_t* lc_scan_segment_offsets(const lc_scan* s) { return s ? s->seg_offsets.data() : nullptr; }

// LiquidPrimitiveClampedArray::try_eval_predicate_inner (hybrid_primitive_array.rs:199-222): can `op literal` be decided
// on rows that hold the sentinel (their real value is only known to be >= reference + sentinel)?
static bool clamp_resolves(const Entry& e, const FixedPred& fp) {
    const uint64_t sentinel = e.W >= 64 ? ~uint64_t(0) : ((uint64_t(1) << e.W) - 1);
    const bool strict = fp.op == LC_OP_EQ || fp.op == LC_OP_NE || fp.op == LC_OP_GT || fp.op == LC_OP_LE;
    if (fp.lit_class < 0) return true;    // literal below every value of the type
    if (fp.lit_class > 0) return false;   // above every value
    if (e.fd.is_signed) {
        const int64_t sent_abs = int64_t(e.fd.reference) + int64_t(sentinel);
        const int64_t k = int64_t(fp.lit);
        return strict ? k < sent_abs : k <= sent_abs;
    }
    const uint64_t sent_abs = e.fd.reference + sentinel;
    return strict ? fp.lit < sent_abs : fp.lit <= sent_abs;
}

// Which squeezed entries of the scan hold a valid, selected row that `preds` cannot decide: the sentinel rows of a
// clamp-squeezed entry, the rows in the literal's bucket of a quantize-squeezed one (the kernel knows which comparisons
// a bucket decides: packed_range_quantized).  `preds` null: a read — any sentinel row counts (to_arrow_known_only
// :129-146), and a quantized entry has no values at all (to_arrow_array hydrates, :688-690).  Synchronises.
static lc_status clamp_unresolved_entries(lc_ctx* ctx, lc_scan* s, const FixedPred* preds, int n_preds,
                                          const void* d_selection, hipStream_t stream, std::vector<uint32_t>* out) {
    out->clear();
    std::vector<uint32_t> suspects;
    bool any_quantized = false;
    for (uint32_t i = 0; i < s->n; i++) {
        const Entry& e = s->meta[i];
        if (e.all_null) continue;
        if (e.fq_shift > 0) {  // float-quantized: reads need the backing, predicates are decided by k_float_quant_pred
            if (!preds) out->push_back(i);
            continue;
        }
        if (e.quantized) {
            if (!preds) out->push_back(i);
            else { suspects.push_back(i); any_quantized = true; }
            continue;
        }
        if (!e.clamped) continue;
        bool resolves = preds != nullptr;
        for (int k = 0; k < n_preds && resolves; k++) resolves = clamp_resolves(e, preds[k]);
        if (!resolves) suspects.push_back(i);
    }
    if (suspects.empty()) return LC_OK;
    const uint64_t words = std::max<uint64_t>(s->seg_offsets.back(), 1);
    uint64_t* d_tmp = static_cast<uint64_t*>(pool_alloc(ctx, words * 8));
    uint32_t* d_cnt = static_cast<uint32_t*>(pool_alloc(ctx, size_t(s->n) * 4));
    struct Bufs {
        lc_ctx* c; void* a; void* b; hipStream_t st;
        ~Bufs() { (void)hipStreamSynchronize(st); pool_release(c, a); pool_release(c, b); }
    } bufs{ctx, d_tmp, d_cnt, stream};
    if (!d_tmp || !d_cnt) return fail(LC_ERR_OOM, "hipMalloc (sentinel pass)");
    ScanLaunch L{};
    L.n_entries = s->n;
    L.blocks_per_entry = s->bpe;
    L.d_selection = static_cast<const uint64_t*>(d_selection);
    L.d_hit = d_tmp;
    L.d_counts = d_cnt;
    // one probe pass per predicate when quantized entries are involved (their undecidable bucket depends on it); the
    // sentinel rows of clamped entries are the same in every pass
    std::vector<uint32_t> cnt(s->n, 0), total(s->n, 0);
    const int passes = any_quantized ? n_preds : 1;
    for (int k = 0; k < passes; k++) {
        FixedPred sp{};
        if (preds) sp = preds[k];
        sp.inner_op = sp.op;
        sp.op = LC_OP_INTERNAL_SENTINEL;
        LC_HIP(launch_fixed_pred(static_cast<const FixedDesc*>(s->d_descs), s->lane_log2, sp, nullptr, s->max_w, L, stream));
        LC_HIP(hipMemcpyAsync(cnt.data(), d_cnt, size_t(s->n) * 4, hipMemcpyDeviceToHost, stream));
        LC_HIP(hipStreamSynchronize(stream));
        for (uint32_t i : suspects) total[i] += cnt[i];
    }
    for (uint32_t i : suspects)
        if (total[i] > 0) out->push_back(i);
    std::sort(out->begin(), out->end());
    return LC_OK;
}

// sparse result of an evaluation (lc_scan_eval_hits)
struct HitsOut {
    void* d_hits = nullptr;
    uint64_t cap = 0;
    void* d_n_hits = nullptr;
    void* d_hit_first = nullptr;
    bool counters_zeroed = false;  // LC_HITS_COUNTERS_ZEROED: the caller zeroed *d_n_hits (one memset for a whole query)
    bool partitioned = false;      // LC_HITS_PARTITIONED
};

static lc_status scan_eval_impl(lc_ctx* ctx, lc_scan* s, const lc_predicate* pred, const void* d_selection,
                                void* d_mask_out, void* d_valid_out, void* d_counts_out, void* d_cand_bytes,
                                hipStream_t stream, const lc_predicate* pred2 = nullptr, void* d_total_out = nullptr,
                                bool tolerate_backing = false, const HitsOut* hits = nullptr) {
    return guarded([&]() -> lc_status {
    bind_device(ctx);
    LC_PHASE("scan_eval_impl");
    if (!ctx || !s || !pred) return fail(LC_ERR_INVALID, "null argument");
    // d_mask_out == NULL: the caller consumes COUNT(*), per-entry counts or the hit list and wants no mask
    if (!d_mask_out && !d_total_out && !d_counts_out && !(hits && hits->d_hits))
        return fail(LC_ERR_INVALID, "no output: d_mask_out is null and neither a count nor a hit list is asked for");
    if (!d_mask_out && d_valid_out) return fail(LC_ERR_INVALID, "a validity output needs the mask output");
    if (hits && hits->d_hits && !hits->d_n_hits) return fail(LC_ERR_INVALID, "d_n_hits is null");
    if (hits && hits->d_n_hits && !hits->counters_zeroed)
        LC_HIP(launch_zero_small(hits->d_n_hits, hits->partitioned ? kHitParts * kHitCounterStride * 8u : 8u, stream));
    if (s->n == 0) {
        if (d_total_out) LC_HIP(hipMemsetAsync(d_total_out, 0, 8, stream));
        return LC_OK;
    }
    {
        std::lock_guard<std::mutex> g(s->mu);
        scan_enter_stream(s, stream);
        s->last_native_hits = false;
        if (!d_mask_out && !s->d_mask_scratch) {
            s->d_mask_scratch = static_cast<uint64_t*>(pool_alloc(ctx, std::max<uint64_t>(s->seg_offsets.back(), 1) * 8));
            if (!s->d_mask_scratch) return fail(LC_ERR_OOM, "hipMalloc (mask scratch)");
        }
    }
    const bool want_hits = hits && hits->d_hits;
    ScanLaunch L{};
    if (!d_mask_out) {
        d_mask_out = s->d_mask_scratch;
        L.mask_optional = 1;
    }
    if (want_hits) {
        L.d_hits = static_cast<uint64_t*>(hits->d_hits);
        L.hits_cap = hits->cap;
        L.d_n_hits = static_cast<unsigned long long*>(hits->d_n_hits);
        L.d_hit_first = static_cast<uint32_t*>(hits->d_hit_first);
        L.hits_parts = hits->partitioned ? kHitParts : 1u;
    }
    // the evaluation proper; kernels that do not append the hit list themselves leave it to k_mask_to_hits below
    const lc_status est = [&]() -> lc_status {
    if (d_total_out) {
        std::lock_guard<std::mutex> g(s->mu);
        if (!s->d_total_acc) {
            s->d_total_acc = static_cast<unsigned long long*>(pool_alloc(ctx, size_t(kTotalWords) * 8));
            if (!s->d_total_acc) return fail(LC_ERR_OOM, "hipMalloc (count accumulator)");
            LC_HIP(hipMemsetAsync(s->d_total_acc, 0, size_t(kTotalWords) * 8, stream));  // once: launches leave it zero
        }
        L.d_total_acc = s->d_total_acc;
        L.d_total_out = static_cast<uint64_t*>(d_total_out);
    }
    L.n_entries = s->n;
    L.blocks_per_entry = s->bpe;
    L.d_selection = static_cast<const uint64_t*>(d_selection);
    L.d_hit = static_cast<uint64_t*>(d_mask_out);
    L.d_valid = static_cast<uint64_t*>(d_valid_out);
    L.d_counts = static_cast<uint32_t*>(d_counts_out);
    L.d_cand_bytes = static_cast<uint32_t*>(d_cand_bytes);  // instrumented pass: [candidate bytes x n | kernel bytes x n]
    L.d_own_bytes = d_cand_bytes ? static_cast<uint32_t*>(d_cand_bytes) + s->n : nullptr;
    L.uniform_slot = -1;
    L.d_work = s->d_work;
    // per-scan facts gathered once at lc_scan_create (walking the entries of a 600 M-row scan per evaluation cost more
    // host time than the kernel ran: ~150 us gaps between the passes of a predicate chain)
    L.max_dict_len = s->max_dict_len;
    if (s->is_str && !s->meta.empty()) {
        L.many_candidates = s->any_without_signatures ? 1u : 0u;
        L.uniform_slot = s->uniform_slot;
    }
    if (!s->is_str) {
        FixedPred fp, fp2;
        lc_status st = make_fixed_pred(s->meta[0], pred, &fp);
        if (st != LC_OK) return st;
        if (pred2) {
            // a second conjunct on the same column rides in the same pass: the two packed-domain ranges are intersected
            // per entry.  Ne is a range with a hole, not a range: the caller evaluates it as its own pass.
            if (pred->op == LC_OP_NE || pred2->op == LC_OP_NE)
                return fail(LC_UNSUPPORTED, "fused conjuncts must be Eq / Lt / LtEq / Gt / GtEq");
            st = make_fixed_pred(s->meta[0], pred2, &fp2);
            if (st != LC_OK) return st;
        }
        if (s->has_clamped) {
            // clamp-squeezed entries: a sentinel row the predicate cannot decide sends the caller to its disk tier
            // (Err(NeedsBacking), hybrid_primitive_array.rs:226-229); everything else evaluates on the clamped data as is
            const FixedPred both[2] = {fp, fp2};
            std::lock_guard<std::mutex> g(s->mu);
            st = clamp_unresolved_entries(ctx, s, both, pred2 ? 2 : 1, d_selection, stream, &s->needs_backing);
            if (st != LC_OK) return st;
            if (!s->needs_backing.empty() && !tolerate_backing)
                return fail(LC_NEEDS_BACKING, "a clamp-squeezed entry holds sentinel rows this predicate cannot decide");
        }
        // (every entry with packed data at one width, none of them squeezed: the kernel of that width — ALP floats included, their
        // packed-domain range is found the same way)
        L.uniform_w = (s->min_w == s->max_w && !s->has_clamped && !s->has_fquant) ? s->max_w : 0u;
        LC_HIP(launch_fixed_pred(static_cast<const FixedDesc*>(s->d_descs), s->lane_log2, fp, pred2 ? &fp2 : nullptr,
                                 s->max_w, L, stream));
        if (s->any_patch) {
            LC_HIP(launch_alp_patch_fix(static_cast<const FixedDesc*>(s->d_descs), s->lane_log2, fp, pred2 ? &fp2 : nullptr, L,
                                        stream));
        }
        if (s->has_fquant) {
            // float-quantized hybrids (the kernels above wrote zeros for them): the reference's bucket-bound decision;
            // an entry with an undecidable valid selected row answers Err(NeedsBacking) (float_array.rs:924-929)
            if (pred2) return fail(LC_UNSUPPORTED, "fused conjuncts are not evaluated on float-quantized entries");
            if (d_selection && s->fquant_patches)
                return fail(LC_UNSUPPORTED, "a selection over a float-quantized entry with ALP exceptions has no defined result in "
                                            "the reference (its patch indices are not re-based, float_array.rs:772-792)");
            uint32_t* d_und = static_cast<uint32_t*>(pool_alloc(ctx, size_t(s->n) * 4));
            if (!d_und) return fail(LC_ERR_OOM, "hipMalloc (float quantize flags)");
            struct Tmp {
                lc_ctx* c; void* p; hipStream_t st;
                ~Tmp() { (void)hipStreamSynchronize(st); pool_release(c, p); }
            } tmp{ctx, d_und, stream};
            LC_HIP(hipMemsetAsync(d_und, 0, size_t(s->n) * 4, stream));
            LC_HIP(launch_float_quant_pred(static_cast<const FixedDesc*>(s->d_descs), s->lane_log2, fp, L, d_und, stream));
            std::vector<uint32_t> und(s->n, 0);
            LC_HIP(hipMemcpyAsync(und.data(), d_und, size_t(s->n) * 4, hipMemcpyDeviceToHost, stream));
            LC_HIP(hipStreamSynchronize(stream));
            std::lock_guard<std::mutex> g(s->mu);
            for (uint32_t i = 0; i < s->n; i++)
                if (und[i]) s->needs_backing.push_back(i);
            std::sort(s->needs_backing.begin(), s->needs_backing.end());
            if (!s->needs_backing.empty() && !tolerate_backing)
                return fail(LC_NEEDS_BACKING, "a float-quantized entry holds a row whose bucket does not decide this predicate");
        }
        return LC_OK;
    }
    if (pred2) return fail(LC_UNSUPPORTED, "fused conjuncts are evaluated on fixed-width columns only");
    StrPredHost sp;
    const lc_status st = make_str_pred(pred, &sp);
    if (st != LC_OK) return st;
    if (sp.p.mode == 3)
        if (s->any_fingerprints)
                return fail(LC_UNSUPPORTED, "general LIKE patterns apply to byte views without fingerprints (the "
                                            "reference requires %needle% on SubstringSearch columns)");
    // `=` / `<>` with a literal of 2..63 bytes on a scan that can have the scan-level signature index: a value equals the
    // literal exactly when it CONTAINS it and has its length, so the evaluation is the substring search of k_like_flat (a
    // handful of candidates per entry instead of every prefix key) followed by a length test.  Anything the index path does
    // not take (unselective literals, entries without the index, validity outputs) falls through to k_str_pred below.
    StrPredHost sq;
    bool try_eq = false;
    if (sp.p.mode == 0 && (sp.p.op == LC_OP_EQ || sp.p.op == LC_OP_NE) && sp.needle.size() >= 2 &&
        (ctx->like_path == 0 || ctx->like_path == 4) && !d_valid_out &&
        !d_cand_bytes && s->any_fingerprints && !s->any_without_signatures && !pred2) {
        std::vector<uint8_t> pat;
        pat.push_back('%');
        pat.insert(pat.end(), sp.needle.begin(), sp.needle.end());
        pat.push_back('%');
        lc_predicate lp2 = *pred;
        lp2.op = sp.p.op == LC_OP_EQ ? LC_OP_LIKE : LC_OP_NOT_LIKE;
        lp2.lit = pat.data();
        lp2.lit_len = pat.size();
        // (a literal that holds `%`, `_` or a backslash is no plain substring pattern: make_str_pred says mode 3)
        // (a literal of more than 63 bytes: the automaton runs over its first 63 and k_like_flat compares the values of the
        // literal's length that contain those byte by byte — `verify` then carries the literal, not a pattern)
        if (make_str_pred(&lp2, &sq) == LC_OK && sq.p.mode == 1) {
            sq.p.eq_len = uint32_t(sp.needle.size());
            sq.p.verify_len = 0;
            sq.verify = sp.needle;
            try_eq = true;
        }
    }
    std::lock_guard<std::mutex> g(s->mu);
    LC_PHASE("eval: string path (all, host side)");
    if (!s->d_wg_ranges) {
        LC_PHASE("eval: workgroup records");
        // workgroup records: consecutive entries, at most four, never across a symbol-table change (row-group boundary).
        // The host says where each begins; k_str_wg_records copies the descriptors into them on the device.
        std::vector<uint32_t> begins;
        uint32_t begin = 0;
        for (uint32_t i = 1; i <= s->n; i++) {
            if (i == s->n || i - begin == 4 || s->symtab_slots[i] != s->symtab_slots[begin]) {
                begins.push_back(begin);
                begin = i;
            }
        }
        begins.push_back(s->n);
        const size_t n_recs = begins.size() - 1;
        s->n_wg_ranges = uint32_t(n_recs);
        s->d_wg_ranges = static_cast<StrWgRecord*>(pool_alloc(ctx, std::max<size_t>(n_recs, 1) * sizeof(StrWgRecord)));
        if (!s->d_wg_ranges) return fail(LC_ERR_OOM, "hipMalloc (scan workgroup records)");
        // (the staging blocks stay with the scan: nothing waits for the copy — the first LIKE of a 12,207-entry scan spent 70-120 us
        // here)
        s->d_wg_begins = static_cast<uint32_t*>(pool_alloc(ctx, begins.size() * 4));
        s->h_wg_begins = host_pool_alloc(ctx, begins.size() * 4);  // (through pinned staging: see scan_create_impl)
        if (!s->d_wg_begins || !s->h_wg_begins) return fail(LC_ERR_OOM, "scan workgroup records: staging");
        std::memcpy(s->h_wg_begins, begins.data(), begins.size() * 4);
        LC_HIP(hipMemcpyAsync(s->d_wg_begins, s->h_wg_begins, begins.size() * 4, hipMemcpyHostToDevice, stream));
        LC_HIP(launch_str_wg_records(static_cast<const StrDesc*>(s->d_descs), s->d_wg_begins, uint32_t(n_recs), s->d_wg_ranges, stream));
    }
    L.d_wg_ranges = s->d_wg_ranges;
    L.n_wg_ranges = s->n_wg_ranges;
    if (!s->d_work) {  // entry-draw counters of k_str_pred: zero once, the kernel leaves them zero
        // one 64-byte counter line per workgroup of the largest launch this scan can make
        const size_t groups = std::min<size_t>(kWorkGroupsMax, std::max<size_t>({s->n_wg_ranges, (size_t(s->n) + 3) / 4, 1}));
        s->d_work = static_cast<uint32_t*>(pool_alloc(ctx, groups * 64));
        if (!s->d_work) return fail(LC_ERR_OOM, "hipMalloc (scan work counters)");
        LC_HIP(hipMemsetAsync(s->d_work, 0, groups * 64, stream));
    }
    L.d_work = s->d_work;
    auto build_automata = [&](StrPredHost& q) -> lc_status {
        LC_PHASE("eval: automata");
        const uint32_t stride = automaton_stride(q.p.needle_len);
        // Folded over the symbol tables THIS scan's entries use — the slots [slot_lo, slot_hi] captured at scan creation — not
        // over every table of the context: a context that holds a few hundred row groups of a few dozen tables has thousands of
        // symbol tables, and a scan over one row group was allocating (70 MB: a hipMalloc, later a 3 ms hipFree) and folding
        // the automata of all of them.  The kernels index by the absolute slot: they get the base of slot 0, which lies in front
        // of the allocation.
        const size_t nst_ctx = s->n_symtabs;
        const bool sub = !s->symtab_slots.empty() && s->slot_hi < nst_ctx;
        const size_t lo = sub ? s->slot_lo : 0, nst = sub ? size_t(s->slot_hi - s->slot_lo) + 1 : nst_ctx;
        const size_t need = size_t(stride) * std::max<size_t>(nst, 1);
        if (need > s->automata_cap) {
            LC_HIP(hipStreamSynchronize(stream));
            pool_release(ctx, s->d_automata);  // (hipFree would synchronise the whole device)
            s->d_automata = static_cast<uint8_t*>(pool_alloc(ctx, need));
            if (!s->d_automata) { s->automata_cap = 0; return fail(LC_ERR_OOM, "hipMalloc (LIKE automata)"); }
            s->automata_cap = need;
            s->automata_symtabs = 0;
        }
        // a query evaluates one pattern over and over: the folded automata are rebuilt only when the needle or the
        // set of symbol tables changed (stream order keeps earlier launches valid)
        if (s->automata_symtabs != nst || s->automata_needle != q.needle) {
            LC_HIP(launch_str_automata(s->d_symtabs + lo, uint32_t(nst), q.needle.data(), q.p.needle_len, s->d_automata, stream));
            s->automata_needle = q.needle;
            s->automata_symtabs = nst;
        }
        q.p.automata = s->d_automata - lo * size_t(stride);
        q.p.automaton_stride = stride;
        return LC_OK;
    };
    if (try_eq) {
        const lc_status ba = build_automata(sq);
        if (ba != LC_OK) return ba;
        if (sq.p.eq_len > uint32_t(kMaxNeedleAutomaton)) {
            const size_t need = sq.verify.size() + 16;
            LC_HIP(hipStreamSynchronize(stream));  // a previous evaluation may still read the old literal
            if (need > s->needle_cap) {
                pool_release(ctx, s->d_needle);
                s->d_needle = static_cast<uint8_t*>(pool_alloc(ctx, need));
                if (!s->d_needle) { s->needle_cap = 0; return fail(LC_ERR_OOM, "hipMalloc (needle)"); }
                s->needle_cap = need;
            }
            LC_HIP(hipMemcpyAsync(s->d_needle, sq.verify.data(), sq.verify.size(), hipMemcpyHostToDevice, stream));
            LC_HIP(hipStreamSynchronize(stream));  // (`sq` is a local)
            sq.p.needle = s->d_needle;
        }
        bool handled = false, many = false;
        const lc_status ps = like_pipeline_eval(ctx, s, sq, L, stream, &handled, &many);
        if (ps != LC_OK) return ps;
        s->last_eq_flat = handled;
        if (handled) { s->last_like_scanall = false; return LC_OK; }
        s->last_like_kernel = LC_LIKE_KERNEL_STR_PRED;
    } else {
        s->last_eq_flat = false;
    }
    if (sp.p.mode == 1) {
        const lc_status ba = build_automata(sp);
        if (ba != LC_OK) return ba;
    }
    if (sp.p.mode == 1 && sp.p.verify_len != 0) {
        const size_t need = sp.verify.size() + 16;
        LC_HIP(hipStreamSynchronize(stream));  // previous evaluation may still read the old pattern
        if (need > s->needle_cap) {
            pool_release(ctx, s->d_needle);
            s->d_needle = static_cast<uint8_t*>(pool_alloc(ctx, need));
            if (!s->d_needle) { s->needle_cap = 0; return fail(LC_ERR_OOM, "hipMalloc (needle)"); }
            s->needle_cap = need;
        }
        LC_HIP(hipMemcpyAsync(s->d_needle, sp.verify.data(), sp.verify.size(), hipMemcpyHostToDevice, stream));
        LC_HIP(hipStreamSynchronize(stream));  // (`sp` is a local)
        sp.p.needle = s->d_needle;
    } else if ((sp.p.mode == 0 || sp.p.mode == 3) && sp.needle.size() > size_t(kInlineNeedle)) {
        const size_t need = sp.needle.size() + 16;
        LC_HIP(hipStreamSynchronize(stream));  // previous evaluation may still read the old needle
        if (need > s->needle_cap) {
            pool_release(ctx, s->d_needle);
            s->d_needle = static_cast<uint8_t*>(pool_alloc(ctx, need));
            if (!s->d_needle) { s->needle_cap = 0; return fail(LC_ERR_OOM, "hipMalloc (needle)"); }
            s->needle_cap = need;
        }
        LC_HIP(hipMemcpyAsync(s->d_needle, sp.needle.data(), sp.needle.size(), hipMemcpyHostToDevice, stream));
        LC_HIP(hipStreamSynchronize(stream));  // (`sp` is a local)
        sp.p.needle = s->d_needle;
    }
    if (sp.p.mode == 1) {
        // selective LIKE over an indexed column: k_like_lean (planned once per scan and needle); everything else: k_str_pred
        bool handled = false, many = false;
        const lc_status ps = like_pipeline_eval(ctx, s, sp, L, stream, &handled, &many);
        if (ps != LC_OK) return ps;
        if (handled) { s->last_like_scanall = false; return LC_OK; }
        if (many && ctx->like_many_hint) L.many_candidates = 1;
        // many candidates (no signature index, no fingerprints, a 1-byte or an unselective needle): the whole FSST buffer of
        // every entry streamed once, lane per word (k_like_scanall) instead of a chain per value
        const bool scanall_ok = (sp.p.op == LC_OP_LIKE || sp.p.op == LC_OP_NOT_LIKE) && sp.p.needle_len >= 1 && sp.p.verify_len == 0 &&
                                automaton_image_bytes(sp.p.needle_len) != 0 && !L.d_cand_bytes && !L.d_own_bytes &&
                                !s->any_multi_empty && ctx->like_path != 1 && s->n_wg_ranges > 0;
        // Where it is used (measured, 100 M-row URL column): every value walked — columns staged without fingerprints — 582 us
        // against 670 for k_str_pred's streaming walker; behind the fingerprint prefilter (40 % of the values) it loses (582
        // vs 480), and for unselective needles on indexed columns as well ('%ru/%' 980 vs 690, '%mail%' 830 vs 324): its ~9
        // instructions per compressed byte (value ends fall inside words) against ~4 for the value-aligned walkers.  So:
        // scans without fingerprints and without the signature index, or LC_OPT_LIKE_PATH = 5 (tests).
        const bool walk_all_scan = s->any_without_signatures && !s->any_fingerprints;
        s->last_like_scanall = scanall_ok && ((L.many_candidates && walk_all_scan) || ctx->like_path == 5);
        s->last_like_kernel = s->last_like_scanall ? LC_LIKE_KERNEL_SCANALL : LC_LIKE_KERNEL_STR_PRED;
        if (s->last_like_scanall) {
            LC_HIP(launch_like_scanall(s->d_wg_ranges, s->n_wg_ranges, sp.p, L, L.d_total_acc, stream));
            return LC_OK;
        }
    }
    LC_HIP(launch_str_pred(static_cast<const StrDesc*>(s->d_descs), s->d_symtabs, sp.p, L, stream));
    return LC_OK;
    }();
    if (est != LC_OK) return est;
    if (want_hits && !s->last_native_hits)
        LC_HIP(launch_mask_to_hits(s->d_descs, s->is_str, s->n, static_cast<const uint64_t*>(d_mask_out), L.d_hits, L.hits_cap,
                                   L.d_n_hits, L.d_hit_first, L.hits_parts, stream));
    return LC_OK;
    });
}

lc_status lc_scan_eval(lc_ctx* ctx, lc_scan* scan, const lc_predicate* pred, const void* d_selection,
                       void* d_mask_out, void* d_counts_out, void* stream) {
    return guarded([&]() -> lc_status {
    return scan_eval_impl(ctx, scan, pred, d_selection, d_mask_out, nullptr, d_counts_out, nullptr,
                          static_cast<hipStream_t>(stream));
    });
}

lc_status lc_scan_eval_and(lc_ctx* ctx, lc_scan* scan, const lc_predicate* preds, uint32_t n_preds, const void* d_selection,
                           void* d_mask_out, void* d_counts_out, void* stream) {
    if (!preds || n_preds == 0 || n_preds > 2) return fail(LC_ERR_INVALID, "lc_scan_eval_and takes one or two predicates");
    return scan_eval_impl(ctx, scan, &preds[0], d_selection, d_mask_out, nullptr, d_counts_out, nullptr,
                          static_cast<hipStream_t>(stream), n_preds == 2 ? &preds[1] : nullptr);
}

lc_status lc_scan_eval_count(lc_ctx* ctx, lc_scan* scan, const lc_predicate* preds, uint32_t n_preds,
                             const void* d_selection, void* d_mask_out, void* d_counts_out, void* d_total_out,
                             void* stream) {
    if (!preds || n_preds == 0 || n_preds > 2) return fail(LC_ERR_INVALID, "lc_scan_eval_count takes one or two predicates");
    if (!d_total_out) return fail(LC_ERR_INVALID, "d_total_out is null");
    return scan_eval_impl(ctx, scan, &preds[0], d_selection, d_mask_out, nullptr, d_counts_out, nullptr,
                          static_cast<hipStream_t>(stream), n_preds == 2 ? &preds[1] : nullptr, d_total_out);
}

lc_status lc_scan_eval_count_groups(lc_ctx* ctx, lc_scan* scan, const lc_predicate* preds, uint32_t n_preds,
                                    const void* d_selection, uint32_t n_groups, const uint32_t* group_ends,
                                    void* d_group_counts_out, void* d_mask_out, void* d_counts_out, void* d_total_out,
                                    void* stream) {
    return guarded([&]() -> lc_status {
    bind_device(ctx);
    if (!ctx || !scan || !preds || n_preds == 0 || n_preds > 2) return fail(LC_ERR_INVALID, "lc_scan_eval_count_groups takes one or two predicates");
    if (!d_group_counts_out || (n_groups && !group_ends)) return fail(LC_ERR_INVALID, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (uint32_t g = 0; g < n_groups; g++)
        if (group_ends[g] > scan->n || (g && group_ends[g] < group_ends[g - 1]))
            return fail(LC_ERR_INVALID, "group_ends must be non-decreasing entry bounds inside the scan");
    if (n_groups && group_ends[n_groups - 1] != scan->n) return fail(LC_ERR_INVALID, "the last group must end at the scan's last entry");
    if (n_groups == 0) return LC_OK;
    uint32_t* d_entry_counts = static_cast<uint32_t*>(d_counts_out);
    {
        // the groups' bounds on the device: uploaded when they differ from the last call's (a reader's row groups do not
        // change between queries)
        std::lock_guard<std::mutex> g(scan->mu);
        scan_enter_stream(scan, st);
        if (!d_entry_counts) {
            if (!scan->d_group_entry_counts) {
                scan->d_group_entry_counts = static_cast<uint32_t*>(pool_alloc(ctx, std::max<size_t>(scan->n, 1) * 4));
                if (!scan->d_group_entry_counts) return fail(LC_ERR_OOM, "hipMalloc (per-entry counts)");
            }
            d_entry_counts = scan->d_group_entry_counts;
        }
        if (scan->group_ends_host.size() != n_groups ||
            std::memcmp(scan->group_ends_host.data(), group_ends, size_t(n_groups) * 4) != 0) {
            LC_HIP(hipStreamSynchronize(st));  // (a launch in flight may still read the previous bounds)
            pool_release(ctx, scan->d_group_ends);
            scan->group_ends_host.clear();
            scan->d_group_ends = static_cast<uint32_t*>(pool_alloc(ctx, size_t(n_groups) * 4));
            void* h = host_pool_alloc(ctx, size_t(n_groups) * 4);
            if (!scan->d_group_ends || !h) {
                host_pool_release(ctx, h);
                return fail(LC_ERR_OOM, "row-group bounds: staging");
            }
            std::memcpy(h, group_ends, size_t(n_groups) * 4);
            const hipError_t ec = hipMemcpyAsync(scan->d_group_ends, h, size_t(n_groups) * 4, hipMemcpyHostToDevice, st);
            const hipError_t es = hipStreamSynchronize(st);
            host_pool_release(ctx, h);
            LC_HIP(ec);
            LC_HIP(es);
            scan->group_ends_host.assign(group_ends, group_ends + n_groups);
        }
    }
    if (scan->n == 0) {
        LC_HIP(hipMemsetAsync(d_group_counts_out, 0, size_t(n_groups) * 8, st));
        if (d_total_out) LC_HIP(hipMemsetAsync(d_total_out, 0, 8, st));
        return LC_OK;
    }
    const lc_status rc = scan_eval_impl(ctx, scan, &preds[0], d_selection, d_mask_out, nullptr, d_entry_counts, nullptr, st,
                                        n_preds == 2 ? &preds[1] : nullptr, d_total_out);
    if (rc != LC_OK) return rc;
    LC_HIP(launch_group_counts(d_entry_counts, scan->d_group_ends, n_groups, static_cast<uint64_t*>(d_group_counts_out), st));
    return LC_OK;
    });
}

lc_status lc_scan_aggregate(lc_ctx* ctx, lc_scan* scan, const void* d_selection, void* d_out, void* stream) {
    return guarded([&]() -> lc_status {
    bind_device(ctx);
    if (!ctx || !scan || !d_out) return fail(LC_ERR_INVALID, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (scan->n == 0) {
        LC_HIP(hipMemsetAsync(d_out, 0, sizeof(lc_aggregate), st));
        return LC_OK;
    }
    if (scan->is_str) return fail(LC_UNSUPPORTED, "aggregates apply to integer, date, timestamp and decimal columns");
    if (scan->any_float) return fail(LC_UNSUPPORTED, "aggregates apply to integer, date, timestamp and decimal columns");
    std::lock_guard<std::mutex> g(scan->mu);
    scan_enter_stream(scan, st);
    if (scan->has_clamped) {  // a selected row without its value in HBM (clamp sentinel, any quantized row)
        const lc_status cs = clamp_unresolved_entries(ctx, scan, nullptr, 0, d_selection, st, &scan->needs_backing);
        if (cs != LC_OK) return cs;
        if (!scan->needs_backing.empty())
            return fail(LC_NEEDS_BACKING, "a selected row of a squeezed entry has no value in HBM");
    }
    if (!scan->d_agg_partials) {
        scan->d_agg_partials = pool_alloc(ctx, size_t(fixed_agg_workgroups(scan->n, scan->lane_log2)) * kAggPartialBytes);
        if (!scan->d_agg_partials) return fail(LC_ERR_OOM, "hipMalloc (aggregate partials)");
    }
    ScanLaunch L{};
    L.n_entries = scan->n;
    L.blocks_per_entry = scan->bpe;
    L.d_selection = static_cast<const uint64_t*>(d_selection);
    LC_HIP(launch_fixed_agg(static_cast<const FixedDesc*>(scan->d_descs), scan->lane_log2, scan->meta[0].fd.is_signed, L,
                            scan->d_agg_partials, static_cast<uint64_t*>(d_out), st));
    return LC_OK;
    });
}

lc_status lc_scan_group_partials(lc_ctx* ctx, lc_scan* group_scan, lc_scan* value_scan, int32_t want_max, const void* d_selection,
                                 void* d_partials, uint64_t capacity, void* d_n_partials, void* stream) {
    return guarded([&]() -> lc_status {
    bind_device(ctx);
    if (!ctx || !group_scan || !d_n_partials || (capacity && !d_partials)) return fail(LC_ERR_INVALID, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    LC_HIP(hipMemsetAsync(d_n_partials, 0, 8, st));
    if (group_scan->n == 0) return LC_OK;
    if (!group_scan->is_str || (value_scan && !value_scan->is_str))
        return fail(LC_UNSUPPORTED, "group partials take byte-view (string / binary) columns");
    if (value_scan && (value_scan->ctx != group_scan->ctx || value_scan->lens != group_scan->lens))
        return fail(LC_ERR_INVALID, "the two scans must cover the same row ranges (same entry lengths)");
    for (lc_scan* s : {group_scan, value_scan}) {
        if (!s) continue;
        std::lock_guard<std::mutex> g(s->mu);
        scan_enter_stream(s, st);
    }
    LC_HIP(launch_group_partials(static_cast<const StrDesc*>(group_scan->d_descs),
                                 value_scan ? static_cast<const StrDesc*>(value_scan->d_descs) : nullptr,
                                 value_scan ? value_scan->d_symtabs : group_scan->d_symtabs, static_cast<const uint64_t*>(d_selection),
                                 group_scan->n, want_max ? 1 : 0, static_cast<lc_group_partial*>(d_partials), capacity,
                                 static_cast<unsigned long long*>(d_n_partials), st));
    return LC_OK;
    });
}

lc_status lc_scan_sum_product(lc_ctx* ctx, lc_scan* scan_a, lc_scan* scan_b, const void* d_selection, void* d_out,
                              void* stream) {
    return guarded([&]() -> lc_status {
    bind_device(ctx);
    if (!ctx || !scan_a || !scan_b || !d_out) return fail(LC_ERR_INVALID, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // entry LENGTHS, not word counts: entries of 60 and 64 rows share a mask layout but not a tail mask
    if (scan_a->ctx != scan_b->ctx || scan_a->lens != scan_b->lens)
        return fail(LC_ERR_INVALID, "the two scans must cover the same row ranges (same entry lengths)");
    if (scan_a->n == 0) {
        LC_HIP(hipMemsetAsync(d_out, 0, sizeof(lc_aggregate), st));
        return LC_OK;
    }
    if (scan_a->is_str || scan_b->is_str || scan_a->any_float || scan_b->any_float)
        return fail(LC_UNSUPPORTED, "aggregates apply to integer, date, timestamp and decimal columns");
    if (scan_a->lane_log2 != scan_b->lane_log2)
        return fail(LC_UNSUPPORTED, "SUM(a * b) takes two columns of the same lane width");
    lc_scan* both[2] = {scan_a, scan_b};
    for (lc_scan* s : both) {
        std::lock_guard<std::mutex> g(s->mu);
        // scan_a's scratch (aggregate partials) is used by the launch; scan_b's descriptors and blobs are read by it: both
        // must know the stream, or lc_scan_destroy(scan_b) could recycle them under the running kernel
        if (s == scan_a) scan_enter_stream(s, st);
        else if (std::find(s->streams_used.begin(), s->streams_used.end(), st) == s->streams_used.end()) s->streams_used.push_back(st);
        if (s->has_clamped) {
            const lc_status cs = clamp_unresolved_entries(ctx, s, nullptr, 0, d_selection, st, &s->needs_backing);
            if (cs != LC_OK) return cs;
            if (!s->needs_backing.empty()) return fail(LC_NEEDS_BACKING, "a selected row of a squeezed entry has no value in HBM");
        }
    }
    std::lock_guard<std::mutex> g(scan_a->mu);
    if (!scan_a->d_agg_partials) {
        scan_a->d_agg_partials = pool_alloc(ctx, size_t(fixed_agg_workgroups(scan_a->n, scan_a->lane_log2)) * kAggPartialBytes);
        if (!scan_a->d_agg_partials) return fail(LC_ERR_OOM, "hipMalloc (aggregate partials)");
    }
    ScanLaunch L{};
    L.n_entries = scan_a->n;
    L.blocks_per_entry = scan_a->bpe;
    L.d_selection = static_cast<const uint64_t*>(d_selection);
    LC_HIP(launch_fixed_sum_product(static_cast<const FixedDesc*>(scan_a->d_descs), static_cast<const FixedDesc*>(scan_b->d_descs),
                                    scan_a->lane_log2, L, scan_a->d_agg_partials, static_cast<uint64_t*>(d_out), st));
    return LC_OK;
    });
}

// Multi-column OR (CachedRowGroup::evaluate_selection_with_predicate, src/datafusion/src/cache/mod.rs:111-150): every
// (column scan, predicate) pair is evaluated over the SAME selection and the results are combined with Kleene OR.
static lc_status scan_eval_or_impl(lc_ctx* ctx, uint32_t n, lc_scan* const* scans, const lc_predicate* preds,
                                   const void* d_selection, void* d_mask_out, void* d_valid_out, void* d_counts_out,
                                   hipStream_t stream) {
    bind_device(ctx);
    if (!ctx || !scans || !preds || !d_mask_out || n == 0) return fail(LC_ERR_INVALID, "null argument");
    lc_scan* s0 = scans[0];
    if (!s0) return fail(LC_ERR_INVALID, "null scan");
    for (uint32_t i = 1; i < n; i++) {
        if (!scans[i] || scans[i]->ctx != s0->ctx || scans[i]->lens != s0->lens)
            return fail(LC_ERR_INVALID, "the scans of a multi-column OR must cover the same row ranges (same entry lengths)");
    }
    const uint64_t words = s0->seg_offsets.back();
    if (s0->n == 0) return LC_OK;
    uint64_t* tmp = nullptr;
    {
        std::lock_guard<std::mutex> g(s0->mu);
        scan_enter_stream(s0, stream);
        if (s0->or_tmp_words < 3 * words) {
            LC_HIP(hipStreamSynchronize(stream));  // earlier launches may still use the smaller scratch
            pool_release(ctx, s0->d_or_tmp);
            s0->d_or_tmp = static_cast<uint64_t*>(pool_alloc(ctx, std::max<uint64_t>(3 * words, 1) * 8));
            s0->or_tmp_words = s0->d_or_tmp ? 3 * words : 0;
            if (!s0->d_or_tmp) return fail(LC_ERR_OOM, "hipMalloc (OR scratch)");
        }
        tmp = s0->d_or_tmp;
    }
    uint64_t* hit = static_cast<uint64_t*>(d_mask_out);
    uint64_t* valid = d_valid_out ? static_cast<uint64_t*>(d_valid_out) : tmp + 2 * words;
    lc_status rc = scan_eval_impl(ctx, s0, &preds[0], d_selection, hit, valid, nullptr, nullptr, stream);
    for (uint32_t i = 1; i < n && rc == LC_OK; i++) {
        rc = scan_eval_impl(ctx, scans[i], &preds[i], d_selection, tmp, tmp + words, nullptr, nullptr, stream);
        if (rc == LC_OK && launch_mask_or_kleene(hit, valid, tmp, tmp + words, words, stream) != hipSuccess)
            rc = fail(LC_ERR_DEVICE, "k_mask_or_kleene launch failed");
    }
    if (rc == LC_OK && d_counts_out) {
        ScanLaunch L{};
        L.n_entries = s0->n;
        L.blocks_per_entry = s0->bpe;
        L.d_selection = hit;
        if (launch_mask_entry_counts(s0->d_descs, s0->is_str, L, static_cast<uint32_t*>(d_counts_out), stream) != hipSuccess)
            rc = fail(LC_ERR_DEVICE, "entry count launch failed");
    }
    return rc;
}

lc_status lc_scan_eval_or(lc_ctx* ctx, uint32_t n, lc_scan* const* scans, const lc_predicate* preds, const void* d_selection,
                          void* d_mask_out, void* d_valid_out, void* d_counts_out, void* stream) {
    return guarded([&]() -> lc_status {
        return scan_eval_or_impl(ctx, n, scans, preds, d_selection, d_mask_out, d_valid_out, d_counts_out,
                                 static_cast<hipStream_t>(stream));
    });
}

// A whole pushed-down filter in one call: the steps of LiquidRowFilter in evaluation order (row_filter.rs:481-515), every
// result the selection of the next step (boolean_buffer_and_then), masks alternating between two caller buffers.
lc_status lc_scan_eval_filter(lc_ctx* ctx, uint32_t n_steps, const lc_filter_step* steps, const void* d_selection,
                              void* d_mask_a, void* d_mask_b, void* d_counts_out, void* d_total_out, void** d_final_mask,
                              void* stream) {
    return guarded([&]() -> lc_status {
    bind_device(ctx);
    if (!ctx || (n_steps && !steps) || !d_mask_a || !d_mask_b || d_mask_a == d_mask_b)
        return fail(LC_ERR_INVALID, "null argument (two distinct mask buffers are needed)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const void* sel = d_selection;
    void* bufs[2] = {d_mask_a, d_mask_b};
    int next = 0;
    auto take = [&]() -> void* {  // an output buffer that is not the current selection
        void* o = bufs[next];
        if (o == sel) o = bufs[next ^ 1];
        next = (o == bufs[0]) ? 1 : 0;
        return o;
    };
    // a step that the fused chain kernel can take: a predicate (or fusable pair) on a fixed-width column of u16 / u32 / u64
    // lanes at <= 32 bits, no floats (their exception rows need a second kernel), no squeezed entries (probe passes)
    auto chainable = [&](const lc_filter_step& sp, const lc_scan* first) -> bool {
        if (sp.kind != LC_STEP_AND || !sp.scans || !sp.preds || sp.n_terms == 0 || sp.n_terms > 2) return false;
        const lc_scan* s = sp.scans[0];
        if (!s || s->n == 0 || s->is_str || s->lane_log2 < 4 || s->max_w > 32 || s->has_clamped || s->any_float) return false;
        if (first && (s->ctx != first->ctx || s->lens != first->lens)) return false;
        for (uint32_t t = 0; t < sp.n_terms; t++) {
            const int op = sp.preds[t].op;
            if (op < LC_OP_EQ || op > LC_OP_GE || !sp.preds[t].lit) return false;
            if (sp.n_terms == 2 && op == LC_OP_NE) return false;
        }
        return true;
    };
    for (uint32_t k = 0; k < n_steps; k++) {
        const lc_filter_step& sp = steps[k];
        if (!sp.scans || !sp.preds || sp.n_terms == 0) return fail(LC_ERR_INVALID, "empty filter step");
        // consecutive chainable steps run as ONE launch (k_fixed_chain): every wave takes its entry through all of them
        if (chainable(sp, nullptr)) {
            uint32_t run = 1;
            while (k + run < n_steps && run < uint32_t(kMaxChainSteps) && chainable(steps[k + run], sp.scans[0])) run++;
            if (run >= 2) {
                FixedChainArgs chain{};
                uint32_t max_w = 1;
                uint32_t col_w[kMaxChainSteps] = {};
                for (uint32_t j = 0; j < run; j++) {
                    const lc_filter_step& cj = steps[k + j];
                    lc_scan* s = cj.scans[0];
                    FixedChainStep& cs = chain.step[j];
                    cs.descs = static_cast<const FixedDesc*>(s->d_descs);
                    cs.lane_log2 = s->lane_log2;
                    lc_status rc = make_fixed_pred(s->meta[0], &cj.preds[0], &cs.pred);
                    if (rc != LC_OK) return rc;
                    cs.pred2 = FixedPred{};
                    cs.pred2.op = -1;
                    if (cj.n_terms == 2) {
                        rc = make_fixed_pred(s->meta[0], &cj.preds[1], &cs.pred2);
                        if (rc != LC_OK) return rc;
                    }
                    max_w = std::max(max_w, s->max_w);
                    col_w[j] = s->max_w;
                    // the chain kernel reads the descriptors and blobs of EVERY step's scan on `st`: each scan must drain that
                    // stream before its descriptors are recycled (lc_scan_destroy no longer synchronises the device)
                    scan_note_stream(s, st);
                }
                chain.n_steps = run;
                lc_scan* s_last = steps[k + run - 1].scans[0];
                const bool run_is_last = k + run == n_steps;
                ScanLaunch L{};
                L.n_entries = s_last->n;
                L.blocks_per_entry = s_last->bpe;
                L.d_selection = static_cast<const uint64_t*>(sel);
                void* out = take();
                L.d_hit = static_cast<uint64_t*>(out);
                L.d_counts = static_cast<uint32_t*>(d_counts_out);
                if (run_is_last && d_total_out) {
                    std::lock_guard<std::mutex> g(s_last->mu);
                    if (!s_last->d_total_acc) {
                        s_last->d_total_acc = static_cast<unsigned long long*>(pool_alloc(ctx, size_t(kTotalWords) * 8));
                        if (!s_last->d_total_acc) return fail(LC_ERR_OOM, "hipMalloc (count accumulator)");
                        LC_HIP(hipMemsetAsync(s_last->d_total_acc, 0, size_t(kTotalWords) * 8, st));
                    }
                    L.d_total_acc = s_last->d_total_acc;
                    L.d_total_out = static_cast<uint64_t*>(d_total_out);
                }
                LC_HIP(launch_fixed_chain(chain, max_w, col_w, L, st));
                sel = out;
                k += run - 1;
                continue;
            }
        }
        const bool last = k + 1 == n_steps;
        void* total = last ? d_total_out : nullptr;
        if (sp.kind == LC_STEP_OR) {
            void* out = take();
            const lc_status rc = scan_eval_or_impl(ctx, sp.n_terms, sp.scans, sp.preds, sel, out, nullptr, d_counts_out, st);
            if (rc != LC_OK) return rc;
            if (total) return fail(LC_UNSUPPORTED, "the fused COUNT(*) is produced by predicate steps, not by an OR step");
            sel = out;
            continue;
        }
        if (sp.kind != LC_STEP_AND || sp.n_terms > 2 || !sp.scans[0]) return fail(LC_ERR_INVALID, "bad filter step");
        void* out = take();
        lc_status rc = scan_eval_impl(ctx, sp.scans[0], &sp.preds[0], sel, out, nullptr, d_counts_out, nullptr, st,
                                      sp.n_terms == 2 ? &sp.preds[1] : nullptr, total);
        if (rc == LC_UNSUPPORTED && sp.n_terms == 2) {
            // not fusable after all (byte views, Ne): the two predicates as two chained passes
            rc = scan_eval_impl(ctx, sp.scans[0], &sp.preds[0], sel, out, nullptr, d_counts_out, nullptr, st);
            if (rc != LC_OK) return rc;
            sel = out;
            out = take();
            rc = scan_eval_impl(ctx, sp.scans[0], &sp.preds[1], sel, out, nullptr, d_counts_out, nullptr, st, nullptr, total);
        }
        if (rc != LC_OK) return rc;
        sel = out;
    }
    if (d_final_mask) *d_final_mask = const_cast<void*>(sel);
    return LC_OK;
    });
}

// Kernel time of ONE evaluation with the memory-side cache flushed before every launch: `flush_bytes` of scratch are
// READ between launches (the 256 MiB Infinity Cache of MI355X keeps a 150 MB working set resident across
// back-to-back identical passes, which a hot-cache QUERY over a 100-column table never enjoys).
// Host-side form of packed_range()'s constant test for integer / decimal entries: an entry whose FoR range excludes the
// literal is answered from its metadata and its packed data is never read.
static bool fixed_entry_is_constant(const Entry& e, const FixedPred& fp) {
    if (e.all_null || e.W == 0) return true;
    if (e.fd.kind == kKindF32 || e.fd.kind == kKindF64) return false;  // decided by the per-entry boundary search
    if (fp.lit_class != 0) return true;
    const uint64_t umax = e.W >= 64 ? ~uint64_t(0) : ((uint64_t(1) << e.W) - 1);
    const bool below = e.fd.is_signed ? (int64_t(fp.lit) < int64_t(e.fd.reference)) : (fp.lit < e.fd.reference);
    if (below) return true;
    const uint64_t dlit = fp.lit - e.fd.reference;
    if (dlit > umax) return true;
    return (fp.op == LC_OP_LT && dlit == 0) || (fp.op == LC_OP_GT && dlit == umax);
}

lc_status lc_scan_traffic_model(lc_scan* s, const lc_predicate* pred, int32_t with_selection, uint64_t* out_algorithmic,
                                uint64_t* out_kernel_bytes) {
    return guarded([&]() -> lc_status {
    if (s) bind_device(s->ctx);
    if (!s || !pred || !out_algorithmic || !out_kernel_bytes) return fail(LC_ERR_INVALID, "null argument");
    *out_algorithmic = *out_kernel_bytes = 0;
    uint64_t alg = 0, own = 0;
    const uint32_t sparse_flags = uint32_t(with_selection) & (LC_TRAFFIC_NO_MASK | LC_TRAFFIC_HIT_LIST);
    with_selection &= LC_TRAFFIC_WITH_SELECTION;
    if (!s->is_str) {
        if (s->n == 0) return LC_OK;
        FixedPred fp;
        const lc_status st = make_fixed_pred(s->meta[0], pred, &fp);
        if (st != LC_OK) return st;
        for (const Entry& e : s->meta) {
            const uint64_t full = fixed_alg_bytes(e, with_selection != 0);
            alg += full;
            own += sizeof(FixedDesc) + (fixed_entry_is_constant(e, fp) ? full - uint64_t(e.len) * uint64_t(e.W) / 8 : full);
        }
        *out_algorithmic = alg;
        *out_kernel_bytes = own;
        return LC_OK;
    }
    // byte views: 2n keys + n/8 out (+ n/8 selection, + n/8 validity) and
    //   LIKE        : 4D fingerprints + offsets + compressed bytes of the fingerprint candidates
    //   Eq/ordering : 8D prefix keys + compressed bytes of the ambiguous entries
    // Candidate bytes are data dependent: measured by ONE instrumented device pass (default stream, synchronised; the
    // serialisation rule of lc_scan_eval applies).  The same pass counts the bytes this library's kernel itself has to
    // move for the predicate (signature slices instead of fingerprints, only the candidates that survive them, keys
    // only for entries in which some dictionary value matched).
    lc_ctx* ctx = s->ctx;
    LC_HIP(hipSetDevice(ctx->device));
    uint32_t* d_cand = static_cast<uint32_t*>(pool_alloc(ctx, size_t(s->n) * 8 + 8));
    uint64_t* d_mask = static_cast<uint64_t*>(pool_alloc(ctx, std::max<uint64_t>(s->seg_offsets.back(), 1) * 8));
    std::vector<uint32_t> cand(size_t(s->n) * 2, 0);
    lc_status rc = (!d_cand || !d_mask) ? fail(LC_ERR_OOM, "hipMalloc (traffic model scratch)") : LC_OK;
    if (rc == LC_OK && hipMemset(d_cand, 0, size_t(s->n) * 8) != hipSuccess) rc = fail(LC_ERR_DEVICE, "hipMemset");
    const bool like = pred->op == LC_OP_LIKE || pred->op == LC_OP_NOT_LIKE;
    // a plain evaluation first: the path the predicate really takes (scan-level LIKE pipeline or k_str_pred) is then
    // planned, and the kernel bytes reported below describe THAT path
    bool scanall_plain = false;
    std::vector<uint32_t> counts_plain;
    if (rc == LC_OK && like) {
        rc = scan_eval_impl(ctx, s, pred, nullptr, d_mask, nullptr, d_cand, nullptr, nullptr);  // (d_cand doubles as the counts)
        scanall_plain = rc == LC_OK && s->last_like_scanall;
        if (scanall_plain) {
            counts_plain.assign(s->n, 0);
            if (hipDeviceSynchronize() != hipSuccess ||
                hipMemcpy(counts_plain.data(), d_cand, size_t(s->n) * 4, hipMemcpyDeviceToHost) != hipSuccess)
                rc = fail(LC_ERR_DEVICE, "traffic model: counts of the plain pass");
        }
        if (rc == LC_OK && hipMemset(d_cand, 0, size_t(s->n) * 8) != hipSuccess) rc = fail(LC_ERR_DEVICE, "hipMemset");
    }
    if (rc == LC_OK) rc = scan_eval_impl(ctx, s, pred, nullptr, d_mask, nullptr, nullptr, d_cand, nullptr);
    if (scanall_plain) s->last_like_scanall = true;  // (the instrumented pass above is k_str_pred's; EXPLAIN describes the plain path)
    if (rc == LC_OK && (hipDeviceSynchronize() != hipSuccess ||
                        hipMemcpy(cand.data(), d_cand, size_t(s->n) * 8, hipMemcpyDeviceToHost) != hipSuccess))
        rc = fail(LC_ERR_DEVICE, "traffic model: instrumented pass failed");
    (void)hipDeviceSynchronize();
    pool_release(ctx, d_cand);
    pool_release(ctx, d_mask);
    if (rc != LC_OK) return rc;
    for (uint32_t i = 0; i < s->n; i++) {
        const Entry& e = s->meta[i];
        const uint64_t n = e.len, m = (n + 7) / 8;
        alg += 2 * n + m + (with_selection ? m : 0) + (e.nullable ? m : 0) + cand[i];
        if (like) alg += (e.has_fp ? 4ull * e.dict_len : 0) + e.offsets_bytes;
        else if (pred->lit_tag == LC_LIT_BYTES) alg += 8ull * e.dict_len + (cand[i] ? e.offsets_bytes : 0);
        own += cand[size_t(s->n) + i];
    }
    if (like) {
        StrPredHost sp;
        if (make_str_pred(pred, &sp) == LC_OK && sp.p.mode == 1) {
            std::lock_guard<std::mutex> g(s->mu);
            const uint64_t pb = like_pipeline_bytes(s, sp, false, sparse_flags);
            if (pb) own = pb;  // the pipeline takes this needle: its two kernels' bytes, not k_str_pred's
            else if (scanall_plain) {
                // k_like_scanall: per entry its descriptor, the offset residuals, the whole FSST buffer, the keys of entries
                // in which some row hit (2 n; the kernel skips them when no dictionary value matched), validity words of
                // nullable entries with hits, mask words out
                own = 0;
                for (uint32_t i = 0; i < s->n; i++) {
                    const Entry& e = s->meta[i];
                    const uint64_t words = (uint64_t(e.len) + 63) / 64;
                    const bool hit = counts_plain[i] != 0 || pred->op == LC_OP_NOT_LIKE;
                    own += sizeof(StrDesc) + e.offsets_bytes + e.fsst_len + (hit ? 2ull * e.len : 0) +
                           words * 8 * (1 + ((hit && e.nullable) ? 1 : 0) + ((with_selection && hit) ? 1 : 0));
                }
            }
        }
    }
    if (!like && (pred->op == LC_OP_EQ || pred->op == LC_OP_NE) && pred->lit_tag == LC_LIT_BYTES) {
        // `=` / `<>` that the plain evaluation sends through the scan-level index: that kernel's bytes (the substring search
        // of the literal + 8 bytes of prefix key per value that contains it), not k_str_pred's
        uint64_t* d_m2 = static_cast<uint64_t*>(pool_alloc(ctx, std::max<uint64_t>(s->seg_offsets.back(), 1) * 8));
        if (d_m2) {
            const lc_status r2 = scan_eval_impl(ctx, s, pred, nullptr, d_m2, nullptr, nullptr, nullptr, nullptr);
            (void)hipDeviceSynchronize();
            pool_release(ctx, d_m2);
            if (r2 == LC_OK) {
                std::lock_guard<std::mutex> g(s->mu);
                if (s->last_eq_flat) {
                    StrPredHost sq;
                    sq.needle.assign(static_cast<const uint8_t*>(pred->lit),
                                     static_cast<const uint8_t*>(pred->lit) + std::min<size_t>(pred->lit_len, size_t(kMaxNeedleAutomaton)));
                    sq.p.mode = 1;
                    sq.p.op = LC_OP_LIKE;
                    sq.p.needle_len = uint32_t(sq.needle.size());
                    sq.p.eq_len = uint32_t(pred->lit_len);
                    const uint64_t pb = like_pipeline_bytes(s, sq, false, sparse_flags);
                    if (pb) own = pb;
                }
            }
        }
    }
    *out_algorithmic = alg;
    *out_kernel_bytes = own;
    return LC_OK;
    });
}

lc_status lc_scan_explain(lc_scan* s, const lc_predicate* pred, char* out, size_t cap) {
    return guarded([&]() -> lc_status {
    if (s) bind_device(s->ctx);
    if (!s || !pred || !out || cap == 0) return fail(LC_ERR_INVALID, "null argument");
    std::string text;
    if (!s->is_str) {
        text = s->max_w <= 32 && s->lane_log2 >= 4 ? "k_fixed_pred_reg (register resident, thread = FastLanes lane)"
                                                   : "k_fixed_pred (LDS staged)";
        if (s->any_patch) text += " + k_alp_patch_fix";
    } else {
        StrPredHost sp;
        const lc_status st = make_str_pred(pred, &sp);
        if (st != LC_OK) return st;
        std::lock_guard<std::mutex> g(s->mu);
        like_pipeline_wait(s);  // (an index build in flight is waited for: the line describes the steady state)
        if (sp.p.mode == 1) {
            text = like_pipeline_explain(s, sp);
            // (what the last evaluation of a LIKE on this scan really launched)
            if (s->last_like_scanall && text.compare(0, 10, "k_str_pred") == 0)
                text = "k_like_scanall (every dictionary value walked, lane per 8-byte word)" + text.substr(10);
        } else {
            text = "k_str_pred";
            if (s->last_eq_flat && sp.p.mode == 0 && (sp.p.op == LC_OP_EQ || sp.p.op == LC_OP_NE)) {
                // (what the last `=` / `<>` on this scan really launched: the substring search of the literal + a length test)
                StrPredHost sq;
                sq.needle.assign(sp.needle.begin(), sp.needle.begin() + long(std::min(sp.needle.size(), size_t(kMaxNeedleAutomaton))));
                sq.p.mode = 1;
                sq.p.needle_len = uint32_t(sq.needle.size());
                sq.p.eq_len = sp.p.needle_len;
                text = like_pipeline_explain(s, sq);
            }
        }
    }
    std::snprintf(out, cap, "%s", text.c_str());
    return LC_OK;
    });
}

uint64_t lc_scan_algorithmic_bytes(lc_scan* s, const lc_predicate* pred, int32_t with_selection) {
    uint64_t alg = 0, own = 0;
    return lc_scan_traffic_model(s, pred, with_selection, &alg, &own) == LC_OK ? alg : 0;
}

// ------------------------------------------------------------------ small device helpers
lc_status lc_device_alloc(lc_ctx* ctx, uint64_t bytes, void** out) {
    return guarded([&]() -> lc_status {
    if (!ctx || !out) return fail(LC_ERR_INVALID, "null argument");
    if (ctx->device < 0) return fail(LC_ERR_DEVICE, "host-only context: no HIP device (there is no CPU fallback)");
    LC_HIP(hipSetDevice(ctx->device));
    if (hipMalloc(out, bytes ? bytes : 8) != hipSuccess) {
        (void)hipGetLastError();
        like_orphans_clear(ctx);  // (cached scan-level indexes of destroyed scans go before an allocation fails)
        if (hipMalloc(out, bytes ? bytes : 8) != hipSuccess) return fail(LC_ERR_OOM, "hipMalloc");
    }
    return LC_OK;
    });
}
lc_status lc_device_free(lc_ctx* ctx, void* p) {
    return guarded([&]() -> lc_status {
    if (!ctx) return fail(LC_ERR_INVALID, "null argument");
    if (p) LC_HIP(hipFree(p));
    return LC_OK;
    });
}
lc_status lc_device_memset(lc_ctx* ctx, void* p, int v, uint64_t bytes, void* stream) {
    return guarded([&]() -> lc_status {
    if (!ctx || !p) return fail(LC_ERR_INVALID, "null argument");
    if (v == 0 && bytes <= 4096 && (bytes & 3u) == 0 && (reinterpret_cast<uintptr_t>(p) & 3u) == 0)
        LC_HIP(launch_zero_small(p, uint32_t(bytes), static_cast<hipStream_t>(stream)));  // (a query's counters)
    else
        LC_HIP(hipMemsetAsync(p, v, bytes, static_cast<hipStream_t>(stream)));
    return LC_OK;
    });
}
lc_status lc_device_to_host(lc_ctx* ctx, void* dst, const void* src, uint64_t bytes, void* stream) {
    return guarded([&]() -> lc_status {
    if (!ctx || !dst || !src) return fail(LC_ERR_INVALID, "null argument");
    LC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    LC_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return LC_OK;
    });
}
lc_status lc_host_to_device(lc_ctx* ctx, void* dst, const void* src, uint64_t bytes, void* stream) {
    return guarded([&]() -> lc_status {
    if (!ctx || !dst || !src) return fail(LC_ERR_INVALID, "null argument");
    LC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    LC_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return LC_OK;
    });
}
lc_status lc_stream_synchronize(lc_ctx* ctx, void* stream) {
    return guarded([&]() -> lc_status {
    if (!ctx) return fail(LC_ERR_INVALID, "null argument");
    LC_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return LC_OK;
    });
}

lc_status lc_stream_create(lc_ctx* ctx, void** out_stream) {
    return guarded([&]() -> lc_status {
    if (!ctx || !out_stream) return fail(LC_ERR_INVALID, "null argument");
    if (ctx->device < 0) return fail(LC_ERR_DEVICE, "host-only context: no HIP device (there is no CPU fallback)");
    LC_HIP(hipSetDevice(ctx->device));
    hipStream_t s = nullptr;
    int lo = 0, hi = 0;  // (query streams take the highest priority: see stream_acquire)
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) != hipSuccess) {
        (void)hipGetLastError();
        LC_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    }
    *out_stream = s;
    return LC_OK;
    });
}

lc_status lc_stream_destroy(lc_ctx* ctx, void* stream) {
    return guarded([&]() -> lc_status {
    if (!ctx) return fail(LC_ERR_INVALID, "null argument");
    if (stream) LC_HIP(hipStreamDestroy(static_cast<hipStream_t>(stream)));
    return LC_OK;
    });
}

// ------------------------------------------------------------------ per-entry drop-in calls
// One per-entry drop-in call: a stream from the context's pool and the call's scratch.  Everything the call launches or
// copies goes to that stream; the ONE host wait is on that stream (never the device: the reference's read path is driven
// from `target_partitions` worker threads at once, liquid_cache_reader.rs:297-391), and the scratch returns to the pools
// once the stream has drained.
struct CallStream {
    lc_ctx* ctx;
    hipStream_t st;
    std::vector<void*> dev, host;
    explicit CallStream(lc_ctx* c) : ctx(c), st(stream_acquire(c)) {}
    void* dalloc(size_t bytes) {
        void* p = pool_alloc(ctx, bytes ? bytes : 8);
        if (p) dev.push_back(p);
        return p;
    }
    void* halloc(size_t bytes) {  // pinned
        void* p = host_pool_alloc(ctx, bytes ? bytes : 8);
        if (p) host.push_back(p);
        return p;
    }
    bool drained = false;  // set by a call whose LAST enqueued work has been waited for: the destructor then waits no second time
    hipError_t sync() { return hipStreamSynchronize(st); }
    ~CallStream() {
        if (!drained) (void)hipStreamSynchronize(st);
        for (void* p : dev) pool_release(ctx, p);
        for (void* p : host) host_pool_release(ctx, p);
        stream_release(ctx, st);
    }
    CallStream(const CallStream&) = delete;
    CallStream& operator=(const CallStream&) = delete;
};

constexpr size_t kScanCacheMax = 8192, kScanCachePerEntry = 2;
// results of a per-entry call up to this size reach the host through pinned memory written by a kernel (no copy engine)
constexpr size_t kPinnedResultMax = size_t(1) << 20;

// An idle one-entry scan of `id` (exclusive use until scan_checkin), or a new one.  A cached scan is only handed out when
// the entry it captured is still the published one (uid).
static lc_status scan_checkout(lc_ctx* ctx, uint64_t id, bool allow_squeezed, hipStream_t st, lc_scan** out) {
    *out = nullptr;
    scan_cache_reap(ctx);
    lc_scan* cached = nullptr;
    {
        std::lock_guard<std::mutex> g(ctx->scan_cache_mu);
        auto it = ctx->scan_cache.find(id);
        if (it != ctx->scan_cache.end() && !it->second.empty()) {
            cached = it->second.back();
            it->second.pop_back();
            ctx->scan_cache_size--;
            if (it->second.empty()) ctx->scan_cache.erase(it);
        }
    }
    if (cached) {
        bool fresh = false, squeezed = false;
        {
            std::shared_lock<std::shared_mutex> g(ctx->mu);
            auto it = ctx->entries.find(id);
            fresh = it != ctx->entries.end() && it->second.uid == cached->meta[0].uid;
            squeezed = fresh && it->second.squeezed_field >= 0;
        }
        if (fresh && (!squeezed || allow_squeezed)) {
            *out = cached;
            return LC_OK;
        }
        lc_scan_destroy(cached);
    }
    return scan_create_impl(ctx, 1, &id, out, allow_squeezed, st);
}
static void scan_checkin(lc_ctx* ctx, uint64_t id, lc_scan* s) {
    if (!s) return;
    bool keep = false;
    {
        std::shared_lock<std::shared_mutex> g(ctx->mu);  // (an entry replaced meanwhile: its scan is not worth keeping)
        auto it = ctx->entries.find(id);
        keep = it != ctx->entries.end() && it->second.uid == s->meta[0].uid;
        if (keep) {
            std::lock_guard<std::mutex> g2(ctx->scan_cache_mu);
            std::vector<lc_scan*>& v = ctx->scan_cache[id];
            keep = v.size() < kScanCachePerEntry && ctx->scan_cache_size < kScanCacheMax;
            if (keep) { v.push_back(s); ctx->scan_cache_size++; }
            else if (v.empty()) ctx->scan_cache.erase(id);
        }
    }
    if (!keep) lc_scan_destroy(s);
}
struct ScanLease {  // check-in (or destruction) of a call's scan on every path out
    lc_ctx* ctx; uint64_t id; lc_scan* s; bool cached;
    ~ScanLease() { if (cached) scan_checkin(ctx, id, s); else if (s) lc_scan_destroy(s); }
};

lc_status lc_eval_predicate_batch(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, const lc_predicate* pred,
                                  const uint8_t* const* selections, uint8_t* const* out_values,
                                  uint8_t* const* out_validity, uint32_t* out_lens, int32_t* out_nullable,
                                  lc_status* statuses) {
    return guarded([&]() -> lc_status {
    if (!ctx || !pred || (n && (!entry_ids || !out_values || !out_lens)))
        return fail(LC_ERR_INVALID, "null argument");
    if (ctx->device < 0) return fail(LC_ERR_DEVICE, "host-only context: no HIP device (there is no CPU fallback)");
    LC_HIP(hipSetDevice(ctx->device));
    // entries that are not staged answer LC_NOT_STAGED individually (Option::None in the reference)
    std::vector<uint64_t> present_ids;
    std::vector<uint64_t> present_idx;
    {
        std::shared_lock<std::shared_mutex> g(ctx->mu);
        for (uint64_t i = 0; i < n; i++) {
            const bool ok = ctx->entries.count(entry_ids[i]) != 0;
            if (statuses) statuses[i] = ok ? LC_OK : LC_NOT_STAGED;
            if (ok) { present_ids.push_back(entry_ids[i]); present_idx.push_back(i); }
            else if (!statuses) return LC_NOT_STAGED;
        }
    }
    if (present_ids.empty()) return LC_OK;
    const bool one = present_ids.size() == 1;
    LC_PROF_T0;
    ScanLease lease{ctx, present_ids[0], nullptr, one};
    CallStream cs(ctx);  // (declared after the lease: the stream is drained before the scan is checked in / destroyed)
    LC_PROF(13);
    lc_scan* scan = nullptr;
    lc_status rc = one ? scan_checkout(ctx, present_ids[0], false, cs.st, &scan)
                       : scan_create_impl(ctx, present_ids.size(), present_ids.data(), &scan, false, cs.st);
    if (rc == LC_NOT_STAGED && statuses) {  // (evicted between the check above and the scan)
        for (uint64_t i : present_idx) statuses[i] = LC_NOT_STAGED;
        return LC_OK;
    }
    if (rc != LC_OK) return rc;
    lease.s = scan;
    LC_PROF(0);
    const uint64_t words = std::max<uint64_t>(scan->seg_offsets.back(), 1);
    const uint32_t m = scan->n;
    bool any_sel = false;
    if (selections)
        for (uint64_t i : present_idx) any_sel |= selections[i] != nullptr;
    // pinned staging: [selection | hit | valid] words + per-entry bit counts (entries without a selection get all ones)
    uint64_t* h_stage = static_cast<uint64_t*>(cs.halloc(words * 8 * 3 + size_t(m) * 4 + 64));
    if (!h_stage) return fail(LC_ERR_OOM, "hipHostMalloc (predicate staging)");
    uint64_t* h_sel = h_stage;
    uint64_t* h_hit = h_stage + words;
    uint64_t* h_valid = h_stage + 2 * words;
    uint32_t* h_bits = reinterpret_cast<uint32_t*>(h_stage + 3 * words);
    if (any_sel) {
        std::memset(h_sel, 0, words * 8);
        for (uint32_t k = 0; k < m; k++) {
            const Entry& e = scan->meta[k];
            uint8_t* dst = reinterpret_cast<uint8_t*>(h_sel + scan->seg_offsets[k]);
            const uint8_t* src = selections[present_idx[k]];
            const size_t nb = bitmap_bytes(e.len);
            if (src) std::memcpy(dst, src, nb);
            else std::memset(dst, 0xFF, nb);
            if (e.len & 7) dst[nb - 1] &= uint8_t((1u << (e.len & 7)) - 1);
        }
    }
    // device scratch in two blocks, so that the results leave in ONE copy: [hit | valid] and (under a selection)
    // [selection | compacted hit | compacted valid | bit counts] — the host block [hit | valid | bits] mirrors the tail
    // No selection and a small result: the predicate kernel stores its [hit | valid] words straight into the pinned block
    // (plain stores over the fabric; the stream wait below is the release) — no device scratch, no copy-engine hop.  Only
    // where every kernel of the evaluation WRITES the words: the ALP-patch / float-quantize / clamp passes re-read them.
    const bool direct = !any_sel && words * 16 <= kPinnedResultMax &&
                        (scan->is_str || (!scan->any_patch && !scan->has_fquant && !scan->has_clamped));
    uint64_t* d_hv = direct ? h_hit : static_cast<uint64_t*>(cs.dalloc(words * 8 * 2));
    uint64_t* d_cmp = any_sel ? static_cast<uint64_t*>(cs.dalloc(words * 8 * 3 + size_t(m) * 4 + 64)) : nullptr;
    if (!d_hv || (any_sel && !d_cmp)) return fail(LC_ERR_OOM, "hipMalloc (predicate scratch)");
    uint64_t *d_hit = d_hv, *d_valid = d_hv + words;
    uint64_t *d_sel = any_sel ? d_cmp : nullptr, *d_chit = any_sel ? d_cmp + words : nullptr,
             *d_cvalid = any_sel ? d_cmp + 2 * words : nullptr;
    uint32_t* d_bits = any_sel ? reinterpret_cast<uint32_t*>(d_cmp + 3 * words) : nullptr;
    LC_PROF(1);
    if (any_sel) LC_HIP(hipMemcpyAsync(d_sel, h_sel, words * 8, hipMemcpyHostToDevice, cs.st));
    rc = scan_eval_impl(ctx, scan, pred, d_sel, d_hit, d_valid, nullptr, nullptr, cs.st, nullptr, nullptr, true);
    if (rc != LC_OK) return rc;
    LC_PROF(2);
    std::vector<uint8_t> backing(m, 0);
    for (uint32_t k : scan->needs_backing) backing[k] = 1;
    if (!scan->needs_backing.empty() && !statuses) return fail(LC_NEEDS_BACKING, "a clamp-squeezed entry needs its backing bytes");
    if (any_sel) {
        // the reference returns a BooleanArray of popcount(selection) rows: compress hit/valid by the selection
        LC_HIP(launch_mask_compress(d_hit, d_sel, scan->d_seg_offsets, m, d_chit, d_bits, cs.st));
        LC_HIP(launch_mask_compress(d_valid, d_sel, scan->d_seg_offsets, m, d_cvalid, nullptr, cs.st));
        LC_HIP(hipMemcpyAsync(h_hit, d_chit, words * 8 * 2 + size_t(m) * 4, hipMemcpyDeviceToHost, cs.st));
    } else if (!direct) {
        LC_HIP(hipMemcpyAsync(h_hit, d_hv, words * 8 * 2, hipMemcpyDeviceToHost, cs.st));
    }
    LC_PROF(3);
    LC_HIP(cs.sync());  // the call's one wait
    cs.drained = true;  // (nothing is enqueued behind it)
    LC_PROF(4);
#ifdef LC_CALL_PROFILE
    g_prof_calls++;
#endif
    if (!any_sel)
        for (uint32_t k = 0; k < m; k++) h_bits[k] = scan->meta[k].len;
    for (uint32_t k = 0; k < m; k++) {
        const uint64_t i = present_idx[k];
        const Entry& e = scan->meta[k];
        if (backing[k]) {  // this entry's answer needs the full array (the others are complete)
            statuses[i] = LC_NEEDS_BACKING;
            out_lens[i] = 0;
            continue;
        }
        const size_t nb = bitmap_bytes(h_bits[k]);
        out_lens[i] = h_bits[k];
        if (out_values[i]) std::memcpy(out_values[i], h_hit + scan->seg_offsets[k], nb);
        if (out_validity && out_validity[i]) std::memcpy(out_validity[i], h_valid + scan->seg_offsets[k], nb);
        if (out_nullable) out_nullable[i] = e.nullable ? 1 : 0;
    }
    return LC_OK;
    });
}

// The predicate over MANY row groups of one column in one call, host in / host out: the reference's call shape (entries named
// per row group, no scan object in the caller's hands: liquid_stream.rs:358-430, liquid_cache_reader.rs:297-339) at a
// granularity a device can work with.  The scan behind it comes from the context's scan cache.
lc_status lc_eval_predicate_row_groups(lc_ctx* ctx, uint64_t n_entries, const uint64_t* entry_ids, uint32_t n_groups,
                                       const uint32_t* group_ends, const lc_predicate* preds, uint32_t n_preds,
                                       uint64_t* out_group_counts, uint64_t* out_mask, uint64_t out_mask_words,
                                       uint64_t* out_total) {
    return guarded([&]() -> lc_status {
    if (!ctx || !preds || n_preds == 0 || n_preds > 2 || (n_entries && !entry_ids) || (n_groups && (!group_ends || !out_group_counts)))
        return fail(LC_ERR_INVALID, "null argument");
    if (ctx->device < 0) return fail(LC_ERR_DEVICE, "host-only context: no HIP device (there is no CPU fallback)");
    LC_HIP(hipSetDevice(ctx->device));
    if (out_total) *out_total = 0;
    if (n_entries == 0) {
        if (n_groups && group_ends[n_groups - 1] != 0) return fail(LC_ERR_INVALID, "group_ends names entries the call does not hold");
        for (uint32_t g = 0; g < n_groups; g++) out_group_counts[g] = 0;
        return LC_OK;
    }
    struct Lease {  // (declared before the stream: the call's work is drained before the scan goes back to the cache)
        lc_scan* s = nullptr;
        ~Lease() { if (s) lc_scan_destroy(s); }
    } lease;
    LC_PHASE("row groups by id (all)");
    lc_status rc;
    {
        LC_PHASE("row groups by id: scan");
        rc = lc_scan_create(ctx, n_entries, entry_ids, &lease.s);
    }
    if (rc != LC_OK) return rc;
    lc_scan* scan = lease.s;
    CallStream cs(ctx);
    const uint64_t words = scan->seg_offsets.back();
    if (out_mask && out_mask_words < words)
        return fail(LC_ERR_INVALID, "out_mask is too small: " + std::to_string(words) + " words (sum of ceil(len / 64) over the entries) are needed");
    const size_t g_bytes = (size_t(n_groups) + 1) * 8, m_bytes = out_mask ? size_t(words) * 8 : 0;
    uint8_t* h = static_cast<uint8_t*>(cs.halloc(g_bytes + m_bytes));
    uint64_t* d_g = static_cast<uint64_t*>(cs.dalloc(g_bytes));
    uint64_t* d_m = out_mask ? static_cast<uint64_t*>(cs.dalloc(m_bytes)) : nullptr;
    if (!h || !d_g || (out_mask && !d_m)) return fail(LC_ERR_OOM, "row-group call: staging");
    uint64_t* d_total = d_g + n_groups;
    {
        LC_PHASE("row groups by id: evaluation calls");
        if (n_groups) {
            rc = lc_scan_eval_count_groups(ctx, scan, preds, n_preds, nullptr, n_groups, group_ends, d_g, d_m, nullptr, d_total, cs.st);
        } else {
            rc = lc_scan_eval_count(ctx, scan, preds, n_preds, nullptr, d_m, nullptr, d_total, cs.st);
        }
    }
    if (rc != LC_OK) return rc;
    {
        LC_PHASE("row groups by id: copies back + wait");
        LC_HIP(hipMemcpyAsync(h, d_g, g_bytes, hipMemcpyDeviceToHost, cs.st));
        if (out_mask) LC_HIP(hipMemcpyAsync(h + g_bytes, d_m, m_bytes, hipMemcpyDeviceToHost, cs.st));
        LC_HIP(cs.sync());
    }
    cs.drained = true;
    {
        // (the call's stream — the only one this lease of the scan ran on — has just been drained: lc_scan_destroy, which
        // synchronises every stream a scan was used on before it keeps the scan, finds nothing left to wait for; from sixteen
        // threads every HIP call counts)
        std::lock_guard<std::mutex> g(scan->mu);
        scan->streams_used.clear();
        scan->last_stream = nullptr;
    }
    if (n_groups) std::memcpy(out_group_counts, h, size_t(n_groups) * 8);
    if (out_total) std::memcpy(out_total, h + size_t(n_groups) * 8, 8);
    if (out_mask) std::memcpy(out_mask, h + g_bytes, m_bytes);
    return LC_OK;
    });
}

lc_status lc_eval_predicate(lc_ctx* ctx, uint64_t entry_id, const lc_predicate* pred, const uint8_t* selection,
                            uint8_t* out_values, uint8_t* out_validity, uint32_t* out_len, int32_t* out_nullable) {
    return guarded([&]() -> lc_status {
    if (!out_values || !out_len) return fail(LC_ERR_INVALID, "null output");
    const uint8_t* sels[1] = {selection};
    uint8_t* ov[1] = {out_values};
    uint8_t* ovalid[1] = {out_validity};
    lc_status st = LC_OK;
    int32_t nullable = 0;
    const lc_status rc = lc_eval_predicate_batch(ctx, 1, &entry_id, pred, sels, ov, ovalid, out_len, &nullable, &st);
    if (out_nullable) *out_nullable = nullable;
    return rc != LC_OK ? rc : st;
    });
}

lc_status lc_eval_predicate_or(lc_ctx* ctx, uint32_t n, const uint64_t* entry_ids, const lc_predicate* preds,
                               const uint8_t* selection, uint8_t* out_values, uint8_t* out_validity, uint32_t* out_len,
                               int32_t* out_nullable) {
    return guarded([&]() -> lc_status {
    if (!ctx || !entry_ids || !preds || !out_values || !out_validity || !out_len || n == 0)
        return fail(LC_ERR_INVALID, "null argument");
    if (ctx->device < 0) return fail(LC_ERR_DEVICE, "host-only context: no HIP device (there is no CPU fallback)");
    LC_HIP(hipSetDevice(ctx->device));
    std::vector<lc_scan*> scans(n, nullptr);
    struct Guard {  // (declared before the stream: it is drained before the scans are checked in)
        lc_ctx* c; const uint64_t* ids; std::vector<lc_scan*>& v;
        ~Guard() { for (size_t k = 0; k < v.size(); k++) scan_checkin(c, ids[k], v[k]); }
    } guard{ctx, entry_ids, scans};
    CallStream cs(ctx);
    for (uint32_t i = 0; i < n; i++) {
        const lc_status rc = scan_checkout(ctx, entry_ids[i], false, cs.st, &scans[i]);
        if (rc != LC_OK) return rc;  // LC_NOT_STAGED == the reference's early `None` (try_read_liquid, mod.rs:128-134)
    }
    const uint32_t len = scans[0]->meta[0].len;
    for (uint32_t i = 1; i < n; i++)
        if (scans[i]->meta[0].len != len) return fail(LC_ERR_INVALID, "the columns of a multi-column OR have different lengths");
    const uint64_t words = std::max<uint64_t>((uint64_t(len) + 63) / 64, 1);
    uint64_t* d_buf = static_cast<uint64_t*>(cs.dalloc(words * 8 * 5 + 64));
    uint64_t* h_buf = static_cast<uint64_t*>(cs.halloc(words * 8 * 3 + 64));
    if (!d_buf || !h_buf) return fail(LC_ERR_OOM, "hipMalloc (OR scratch)");
    uint64_t *d_sel = d_buf, *d_hit = d_buf + words, *d_valid = d_buf + 2 * words, *d_chit = d_buf + 3 * words,
             *d_cvalid = d_buf + 4 * words;
    uint32_t* d_bits = reinterpret_cast<uint32_t*>(d_buf + 5 * words);
    // selection (all rows when absent), tail masked
    std::memset(h_buf, 0, words * 8);
    const size_t nb = bitmap_bytes(len);
    if (selection) std::memcpy(h_buf, selection, nb);
    else std::memset(h_buf, 0xFF, nb);
    if (len & 7) reinterpret_cast<uint8_t*>(h_buf)[nb - 1] &= uint8_t((1u << (len & 7)) - 1);
    LC_HIP(hipMemcpyAsync(d_sel, h_buf, words * 8, hipMemcpyHostToDevice, cs.st));
    lc_status rc = scan_eval_or_impl(ctx, n, scans.data(), preds, d_sel, d_hit, d_valid, nullptr, cs.st);
    if (rc != LC_OK) return rc;
    LC_HIP(launch_mask_compress(d_hit, d_sel, scans[0]->d_seg_offsets, 1, d_chit, d_bits, cs.st));
    LC_HIP(launch_mask_compress(d_valid, d_sel, scans[0]->d_seg_offsets, 1, d_cvalid, nullptr, cs.st));
    uint32_t* h_bits = reinterpret_cast<uint32_t*>(h_buf + 2 * words);
    LC_HIP(hipMemcpyAsync(h_buf, d_chit, words * 8, hipMemcpyDeviceToHost, cs.st));
    LC_HIP(hipMemcpyAsync(h_buf + words, d_cvalid, words * 8, hipMemcpyDeviceToHost, cs.st));
    LC_HIP(hipMemcpyAsync(h_bits, d_bits, 4, hipMemcpyDeviceToHost, cs.st));
    LC_HIP(cs.sync());
    const uint32_t bits = *h_bits;
    std::memcpy(out_values, h_buf, bitmap_bytes(bits));
    std::memcpy(out_validity, h_buf + words, bitmap_bytes(bits));
    *out_len = bits;
    bool nullable = false;
    for (uint32_t i = 0; i < n; i++) nullable |= scans[i]->meta[0].nullable;
    if (out_nullable) *out_nullable = nullable ? 1 : 0;
    return LC_OK;
    });
}

lc_status lc_mask_and_then(lc_ctx* ctx, const uint8_t* left, uint64_t left_bits, const uint8_t* right,
                           uint64_t right_bits, uint8_t* out) {
    return guarded([&]() -> lc_status {
    if (!ctx || !left || !out || (right_bits && !right)) return fail(LC_ERR_INVALID, "null argument");
    if (left_bits == right_bits) {  // datafusion/src/utils.rs:69-72
        std::memcpy(out, right, bitmap_bytes(left_bits));
        return LC_OK;
    }
    if (ctx->device < 0) return fail(LC_ERR_DEVICE, "host-only context: no HIP device (there is no CPU fallback)");
    LC_HIP(hipSetDevice(ctx->device));
    const uint64_t lw = (left_bits + 63) / 64, rw = (right_bits + 63) / 64 + 1;
    CallStream cs(ctx);
    uint64_t* h = static_cast<uint64_t*>(cs.halloc((2 * lw + rw) * 8));
    uint64_t* d = static_cast<uint64_t*>(cs.dalloc((2 * lw + rw) * 8));
    if (!h || !d) return fail(LC_ERR_OOM, "hipMalloc");
    uint64_t *hl = h, *hr = h + lw, *ho = h + lw + rw;
    uint64_t *dl = d, *dr = d + lw, *dout = d + lw + rw;
    std::memset(h, 0, (lw + rw) * 8);
    std::memcpy(hl, left, bitmap_bytes(left_bits));
    if (left_bits & 63) hl[lw - 1] &= (uint64_t(1) << (left_bits & 63)) - 1;
    if (right_bits) std::memcpy(hr, right, bitmap_bytes(right_bits));
    LC_HIP(hipMemcpyAsync(d, h, (lw + rw) * 8, hipMemcpyHostToDevice, cs.st));
    LC_HIP(launch_mask_and_then(dl, left_bits, dr, dout, cs.st));
    LC_HIP(hipMemcpyAsync(ho, dout, lw * 8, hipMemcpyDeviceToHost, cs.st));
    LC_HIP(cs.sync());
    std::memcpy(out, ho, bitmap_bytes(left_bits));
    return LC_OK;
    });
}

// ---- Arrow C Data Interface export helpers ----
extern "C++" {
namespace {
struct ExportPriv {
    std::vector<void*> owned;
    const void* buffers[3] = {nullptr, nullptr, nullptr};
    std::string format;
};
void release_array(struct ArrowArray* a) {
    if (!a || !a->release) return;
    ExportPriv* p = static_cast<ExportPriv*>(a->private_data);
    for (void* q : p->owned) std::free(q);
    delete p;
    a->release = nullptr;
}
void release_schema(struct ArrowSchema* s) {
    if (!s || !s->release) return;
    delete static_cast<std::string*>(s->private_data);
    s->release = nullptr;
}
void fill_schema(struct ArrowSchema* s, const std::string& fmt) {
    std::memset(s, 0, sizeof(*s));
    std::string* keep = new std::string(fmt);
    s->format = keep->c_str();
    s->name = "";
    s->flags = 2;  // ARROW_FLAG_NULLABLE
    s->release = release_schema;
    s->private_data = keep;
}
std::string arrow_format(const Entry& e) {
    if (e.is_str) return (e.phys == kBinary || e.phys == kBinaryView || e.phys == kDict16Binary) ? "z" : "u";
    if (e.logical == kDecimal) {
        char buf[48];
        std::snprintf(buf, sizeof(buf), e.dec_is256 ? "d:%d,%d,256" : "d:%d,%d", e.dec_precision, e.dec_scale);
        return buf;
    }
    switch (e.phys) {
        case kI8: return "c"; case kI16: return "s"; case kI32: return "i"; case kI64: return "l";
        case kU8: return "C"; case kU16: return "S"; case kU32: return "I"; case kU64: return "L";
        case kF32: return "f"; case kF64: return "g"; case kDate32: return "tdD"; case kDate64: return "tdm";
        case kTsS: return "tss:"; case kTsMs: return "tsm:"; case kTsUs: return "tsu:"; case kTsNs: return "tsn:";
        default: return "l";
    }
}
}  // namespace
}  // extern "C++"

// ticks per day of a Date32 / Timestamp entry for the date-part path; 0: Date32; -1: not a date-like entry
static int64_t date_ticks_per_day(const Entry& e) {
    if (e.is_str || e.logical == kDecimal) return -1;
    switch (e.phys) {
        case kDate32: return 0;
        case kTsS: return 86400LL;
        case kTsMs: return 86400000LL;
        case kTsUs: return 86400000000LL;
        case kTsNs: return 86400000000000LL;
        default: return -1;
    }
}

static lc_status get_with_selection_impl(lc_ctx* ctx, uint64_t entry_id, const uint8_t* selection, int date_field,
                                         struct ArrowArray* out_array, struct ArrowSchema* out_schema);

// cache.get(&id).with_selection(&sel).read(): decode + compact on the device, export through the C Data Interface
lc_status lc_get_with_selection(lc_ctx* ctx, uint64_t entry_id, const uint8_t* selection,
                                struct ArrowArray* out_array, struct ArrowSchema* out_schema) {
    return guarded([&]() -> lc_status {
    return get_with_selection_impl(ctx, entry_id, selection, -1, out_array, out_schema);
    });
}

lc_status lc_get_date_part_with_selection(lc_ctx* ctx, uint64_t entry_id, const uint8_t* selection, int32_t field,
                                          struct ArrowArray* out_array, struct ArrowSchema* out_schema) {
    return guarded([&]() -> lc_status {
    if (field < LC_DATE_YEAR || field > LC_DATE_DAY_OF_WEEK) return fail(LC_ERR_INVALID, "unknown date field");
    return get_with_selection_impl(ctx, entry_id, selection, field, out_array, out_schema);
    });
}

static lc_status get_with_selection_impl(lc_ctx* ctx, uint64_t entry_id, const uint8_t* selection, int date_field,
                                         struct ArrowArray* out_array, struct ArrowSchema* out_schema) {
    return guarded([&]() -> lc_status {
    if (!ctx || !out_array || !out_schema) return fail(LC_ERR_INVALID, "null argument");
    if (ctx->device < 0) return fail(LC_ERR_DEVICE, "host-only context: no HIP device (there is no CPU fallback)");
    LC_HIP(hipSetDevice(ctx->device));
    ScanLease lease{ctx, entry_id, nullptr, true};
    CallStream cs(ctx);  // (after the lease: the stream is drained before the scan is checked in)
    lc_scan* scan = nullptr;
    lc_status rc = scan_checkout(ctx, entry_id, date_field >= 0, cs.st, &scan);
    if (rc != LC_OK) return rc;
    lease.s = scan;
    scan_note_stream(scan, cs.st);
    const Entry& e = scan->meta[0];
    if (date_field >= 0 && date_ticks_per_day(e) < 0)
        return fail(LC_UNSUPPORTED, "ExtractDate32 applies to Date32 / Timestamp entries");
    if (e.squeezed_field >= 0 && e.squeezed_field != date_field)
        return fail(LC_NEEDS_BACKING, "entry is squeezed to another date component");
    const bool squeezed = e.squeezed_field >= 0;
    const uint64_t words = std::max<uint64_t>((uint64_t(e.len) + 63) / 64, 1);
    // every launch and copy of this call goes to the call's stream; its scratch returns to the pools when `cs` goes
    auto dalloc = [&](size_t bytes) -> void* { return cs.dalloc(bytes); };
    auto dfree = [&]() {};
    hipStream_t st = cs.st;
#define LC_HIP_G(expr) LC_HIP(expr)
    // One pinned block [selection | compacted validity | bit count, totals] and its device mirror: every transfer of the call
    // is a copy between pinned and device memory on the call's stream.  (Measured and rejected: kernels reading the
    // selection from / writing the results to the pinned block in place — a kernel that touches host memory ends with a
    // system-scope release, ~80 us per launch.)
    uint64_t* h_pin = static_cast<uint64_t*>(cs.halloc(words * 8 * 2 + 64));
    uint64_t* d_pin = static_cast<uint64_t*>(dalloc(words * 8 * 3 + 64));
    if (!h_pin || !d_pin) return fail(LC_ERR_OOM, "hipHostMalloc (get staging)");
    uint64_t* h_sel = h_pin;
    uint64_t* h_valid = h_pin + words;
    uint64_t* h_small = h_pin + 2 * words;  // [0]: k bits (u32), [2..3]: string totals
    uint64_t *d_selc = d_pin, *d_vout = d_pin + words, *d_small = d_pin + 2 * words, *d_vsrc = d_pin + 2 * words + 8;
    h_small[0] = 0;
    const uint64_t* d_sel = nullptr;  // (what the gather kernels take: null = every row)
    // the selection the compaction uses: the caller's, or "all rows" with the tail masked
    if (selection) {
        std::memset(h_sel, 0, words * 8);
        std::memcpy(h_sel, selection, bitmap_bytes(e.len));
        if (e.len & 63) h_sel[words - 1] &= (uint64_t(1) << (e.len & 63)) - 1;
        d_sel = d_selc;
    } else {
        for (uint64_t w = 0; w < words; w++) h_sel[w] = ~uint64_t(0);
        if (e.len & 63) h_sel[words - 1] = (uint64_t(1) << (e.len & 63)) - 1;
        if (e.len == 0) h_sel[0] = 0;
    }
    // compacted validity (k bits)
    uint32_t k_bits = 0;
    {
        const uint64_t* vptr = e.is_str ? e.sd.validity : e.fd.validity;
        const uint64_t* vsrc = vptr;
        if (e.all_null) { LC_HIP_G(hipMemsetAsync(d_vsrc, 0, words * 8, st)); vsrc = d_vsrc; }
        else if (!vptr) { LC_HIP_G(hipMemsetAsync(d_vsrc, 0xFF, words * 8, st)); vsrc = d_vsrc; }
        LC_HIP_G(hipMemcpyAsync(d_selc, h_sel, words * 8, hipMemcpyHostToDevice, st));
        LC_HIP_G(launch_mask_compress(vsrc, d_selc, scan->d_seg_offsets, 1, d_vout, reinterpret_cast<uint32_t*>(d_small), st));
        LC_HIP_G(hipMemcpyAsync(h_valid, d_vout, words * 8 + 8, hipMemcpyDeviceToHost, st));  // (+ the bit count behind it)
        LC_HIP_G(hipStreamSynchronize(st));  // (k sizes the output)
        k_bits = *reinterpret_cast<const uint32_t*>(h_small);
    }
    const uint64_t k = k_bits;
    std::unique_ptr<ExportPriv> priv(new ExportPriv());
    auto host_alloc = [&](size_t bytes) -> uint8_t* {
        uint8_t* p = static_cast<uint8_t*>(std::calloc(bytes + 64, 1));
        priv->owned.push_back(p);
        return p;
    };
    int64_t null_count = 0;
    if (e.nullable) {
        uint8_t* vb = host_alloc(bitmap_bytes(k));
        std::memcpy(vb, h_valid, bitmap_bytes(k));
        null_count = int64_t(k) - int64_t(count_bits(vb, k));
        priv->buffers[0] = vb;
    }
    int64_t n_buffers = 2;
    if (!e.is_str && scan->has_clamped) {
        std::vector<uint32_t> needs;
        const lc_status cs = clamp_unresolved_entries(ctx, scan, nullptr, 0, d_sel, st, &needs);
        if (cs != LC_OK || !needs.empty()) {
            dfree();
            return cs != LC_OK ? cs : fail(LC_NEEDS_BACKING, "a selected row of the clamp-squeezed entry is at or above the sentinel");
        }
    }
    if (!e.is_str) {
        ScanLaunch L{};
        L.n_entries = 1;
        L.blocks_per_entry = scan->bpe;
        L.d_selection = d_sel;
        // a squeezed entry stores i32 components; the Arrow value keeps the width of the original type
        const size_t vw = squeezed ? size_t(phys_width(e.phys)) : size_t(e.fd.value_width);
        const size_t gw = e.fd.value_width;
        uint32_t* d_bc = static_cast<uint32_t*>(dalloc(size_t(scan->bpe) * 4));
        uint64_t* d_bo = static_cast<uint64_t*>(dalloc(fixed_gather_offsets_len(scan->bpe) * 8));
        uint64_t* d_eo = static_cast<uint64_t*>(dalloc(2 * 8));
        uint8_t* d_vals = static_cast<uint8_t*>(dalloc(std::max<size_t>(k, 1) * vw + 64));
        uint8_t* d_comp = squeezed ? static_cast<uint8_t*>(dalloc(std::max<size_t>(k, 1) * gw + 64)) : d_vals;
        if (!d_bc || !d_bo || !d_eo || !d_vals || !d_comp) { dfree(); return fail(LC_ERR_OOM, "hipMalloc"); }
        LC_HIP_G(hipMemsetAsync(d_vals, 0, std::max<size_t>(k, 1) * vw + 64, st));
        if (squeezed) LC_HIP_G(hipMemsetAsync(d_comp, 0, std::max<size_t>(k, 1) * gw + 64, st));
        LC_HIP_G(launch_fixed_gather(static_cast<const FixedDesc*>(scan->d_descs), scan->lane_log2, L, d_bc, d_bo, d_eo,
                                     d_comp, std::max<uint64_t>(k, 1), st));
        if (squeezed)  // SqueezedDate32Array::to_component_array (squeezed_date32_array.rs:267-359)
            LC_HIP_G(launch_component_lossy(reinterpret_cast<const int32_t*>(d_comp), k, int(vw), date_field,
                                            std::max<int64_t>(date_ticks_per_day(e), 1), d_vals, st));
        else if (date_field >= 0)
            LC_HIP_G(launch_date_lossy(d_vals, k, int(vw), date_field, date_ticks_per_day(e), st));
        uint8_t* vals = host_alloc(std::max<size_t>(k, 1) * vw);
        const size_t out_words = (k * vw + 7) / 8;  // (d_vals carries 64 zeroed bytes of slack)
        uint8_t* h_out = out_words * 8 <= kPinnedResultMax ? static_cast<uint8_t*>(cs.halloc(out_words * 8 + 8)) : nullptr;
        if (h_out) {
            LC_HIP_G(hipMemcpyAsync(h_out, d_vals, out_words * 8, hipMemcpyDeviceToHost, st));
            LC_HIP_G(hipStreamSynchronize(st));
            std::memcpy(vals, h_out, k * vw);
        } else {
            LC_HIP_G(hipMemcpyAsync(vals, d_vals, k * vw, hipMemcpyDeviceToHost, st));
            LC_HIP_G(hipStreamSynchronize(st));
        }
        priv->buffers[1] = vals;
    } else {
        n_buffers = 3;
        uint32_t* d_dlen = static_cast<uint32_t*>(dalloc((size_t(e.dict_len) + 1) * 4));
        int32_t* d_offs = static_cast<int32_t*>(dalloc((size_t(e.len) + 2) * 4));
        uint32_t* d_rows = static_cast<uint32_t*>(dalloc((size_t(e.len) + 1) * 4));
        uint64_t* d_tot = static_cast<uint64_t*>(dalloc(16));
        if (!d_dlen || !d_offs || !d_rows || !d_tot) { dfree(); return fail(LC_ERR_OOM, "hipMalloc"); }
        const StrDesc* descs = static_cast<const StrDesc*>(scan->d_descs);
        // passes 1+2 size the output, pass 3 decodes (two launches of the same helper keep the code in one place)
        LC_HIP_G(launch_str_gather(descs, scan->d_symtabs, 0, e.dict_len, 0, d_sel, d_dlen, d_offs, d_rows, d_tot, nullptr,
                                   st));
        LC_HIP_G(hipMemcpyAsync(h_small + 2, d_tot, 16, hipMemcpyDeviceToHost, st));
        LC_HIP_G(hipStreamSynchronize(st));
        const uint64_t tot[2] = {h_small[2], h_small[3]};
        if (tot[1] > uint64_t(INT32_MAX)) { dfree(); return fail(LC_UNSUPPORTED, "selected strings exceed 2 GiB (i32 offsets)"); }
        uint8_t* d_data = static_cast<uint8_t*>(dalloc(tot[1] + 64));
        if (!d_data) { dfree(); return fail(LC_ERR_OOM, "hipMalloc"); }
        LC_HIP_G(launch_str_gather(descs, scan->d_symtabs, 0, e.dict_len, uint32_t(tot[0]), d_sel, d_dlen, d_offs, d_rows,
                                   d_tot, d_data, st));
        int32_t* offs = reinterpret_cast<int32_t*>(host_alloc((k + 1) * 4));
        uint8_t* data = host_alloc(tot[1]);
        const size_t ow = ((k + 1) * 4 + 7) / 8, dw = (tot[1] + 7) / 8;  // (both device buffers are padded beyond that)
        uint64_t* h_out = (ow + dw) * 8 <= kPinnedResultMax ? static_cast<uint64_t*>(cs.halloc((ow + dw) * 8 + 8)) : nullptr;
        if (h_out) {
            LC_HIP_G(hipMemcpyAsync(h_out, d_offs, ow * 8, hipMemcpyDeviceToHost, st));
            LC_HIP_G(hipMemcpyAsync(h_out + ow, d_data, dw * 8, hipMemcpyDeviceToHost, st));
            LC_HIP_G(hipStreamSynchronize(st));
            std::memcpy(offs, h_out, (k + 1) * 4);
            std::memcpy(data, h_out + ow, tot[1]);
        } else {
            LC_HIP_G(hipMemcpyAsync(offs, d_offs, (k + 1) * 4, hipMemcpyDeviceToHost, st));
            LC_HIP_G(hipMemcpyAsync(data, d_data, tot[1], hipMemcpyDeviceToHost, st));
            LC_HIP_G(hipStreamSynchronize(st));
        }
        priv->buffers[1] = offs;
        priv->buffers[2] = data;
    }
    dfree();
#undef LC_HIP_G
    std::memset(out_array, 0, sizeof(*out_array));
    out_array->length = int64_t(k);
    out_array->null_count = null_count;
    out_array->offset = 0;
    out_array->n_buffers = n_buffers;
    out_array->buffers = priv->buffers;
    out_array->release = release_array;
    fill_schema(out_schema, arrow_format(e));
    out_array->private_data = priv.release();
    return LC_OK;
    });
}

lc_status lc_scan_gather_fixed(lc_ctx* ctx, lc_scan* scan, const void* d_selection, void* d_values_out,
                               uint64_t values_capacity_bytes, void* d_row_offsets, void* stream) {
    return guarded([&]() -> lc_status {
    bind_device(ctx);
    if (!ctx || !scan || !d_values_out || !d_row_offsets) return fail(LC_ERR_INVALID, "null argument");
    if (scan->is_str) return fail(LC_UNSUPPORTED, "scan-wide gather covers fixed-width columns");
    if (scan->n == 0) return LC_OK;
    scan_note_stream(scan, static_cast<hipStream_t>(stream));
    const uint64_t vw = scan->meta[0].fd.value_width;
    if (values_capacity_bytes < scan->total_rows * vw && !d_selection)
        return fail(LC_ERR_INVALID, "values buffer too small for an unselected gather");
    // with a selection the row count is only known on the device: rows beyond the capacity are not written, and
    // d_row_offsets[n] tells the caller how many there were
    const uint64_t capacity_rows = values_capacity_bytes / vw;
    hipStream_t st = static_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> g(scan->mu);
    if (scan->has_clamped) {  // a selected sentinel row has no value in HBM (to_arrow_known_only -> None -> hydrate from disk)
        const lc_status cs = clamp_unresolved_entries(ctx, scan, nullptr, 0, d_selection, st, &scan->needs_backing);
        if (cs != LC_OK) return cs;
        if (!scan->needs_backing.empty())
            return fail(LC_NEEDS_BACKING, "a selected row of a clamp-squeezed entry is at or above the clamp sentinel");
    }
    const size_t nblk = size_t(scan->n) * scan->bpe;
    const size_t need = nblk * 4 + fixed_gather_offsets_len(nblk) * 8 + 64;
    if (need > scan->needle_cap) {  // reuse the scan's scratch allocation
        LC_HIP(hipStreamSynchronize(st));
        if (scan->d_needle) LC_HIP(hipFree(scan->d_needle));
        LC_HIP(hipMalloc(reinterpret_cast<void**>(&scan->d_needle), need));
        scan->needle_cap = need;
    }
    uint64_t* d_bo = reinterpret_cast<uint64_t*>(scan->d_needle);
    uint32_t* d_bc = reinterpret_cast<uint32_t*>(scan->d_needle + fixed_gather_offsets_len(nblk) * 8);
    ScanLaunch L{};
    L.n_entries = scan->n;
    L.blocks_per_entry = scan->bpe;
    L.d_selection = static_cast<const uint64_t*>(d_selection);
    LC_HIP(launch_fixed_gather(static_cast<const FixedDesc*>(scan->d_descs), scan->lane_log2, L, d_bc, d_bo,
                               static_cast<uint64_t*>(d_row_offsets), static_cast<uint8_t*>(d_values_out), capacity_rows,
                               st));
    return LC_OK;
    });
}

lc_status lc_scan_gather_bytes_plan(lc_ctx* ctx, lc_scan* scan, const void* d_selection, void* d_row_offsets,
                                    void* d_row_refs, void* d_value_offsets, void* d_row_valid, uint64_t capacity_rows,
                                    uint64_t* out_rows, uint64_t* out_bytes, void* stream) {
    return guarded([&]() -> lc_status {
    bind_device(ctx);
    if (!ctx || !scan || !d_row_offsets || !d_row_refs || !d_value_offsets || !out_rows || !out_bytes)
        return fail(LC_ERR_INVALID, "null argument");
    if (!scan->is_str) return fail(LC_UNSUPPORTED, "lc_scan_gather_bytes covers byte-view columns");
    *out_rows = 0;
    *out_bytes = 0;
    scan_note_stream(scan, static_cast<hipStream_t>(stream));
    if (scan->n == 0) return LC_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    scan_note_stream(scan, st);
    const uint64_t n = scan->n;
    const uint64_t tiles_len = std::max<uint64_t>(n, capacity_rows) / 1024 + 4;
    uint32_t* d_counts = static_cast<uint32_t*>(pool_alloc(ctx, n * 4));
    uint64_t* d_tiles = static_cast<uint64_t*>(pool_alloc(ctx, tiles_len * 8));
    uint32_t* d_len = static_cast<uint32_t*>(pool_alloc(ctx, std::max<uint64_t>(capacity_rows, 1) * 4));
    auto release = [&]() {
        (void)hipStreamSynchronize(st);
        pool_release(ctx, d_counts);
        pool_release(ctx, d_tiles);
        pool_release(ctx, d_len);
    };
    if (!d_counts || !d_tiles || !d_len) { release(); return fail(LC_ERR_OOM, "hipMalloc (gather scratch)"); }
    ScanLaunch L{};
    L.n_entries = scan->n;
    L.blocks_per_entry = scan->bpe;
    L.d_selection = static_cast<const uint64_t*>(d_selection);
    const StrDesc* descs = static_cast<const StrDesc*>(scan->d_descs);
    uint64_t k = 0, bytes = 0;
    lc_status rc = LC_OK;
    if (launch_str_entry_offsets(descs, L, d_counts, d_tiles, static_cast<uint64_t*>(d_row_offsets), st) != hipSuccess ||
        hipMemcpyAsync(&k, static_cast<uint64_t*>(d_row_offsets) + n, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        rc = fail(LC_ERR_DEVICE, "gather plan: counting pass failed");
    *out_rows = k;
    if (rc == LC_OK && k > capacity_rows) rc = fail(LC_ERR_INVALID, "gather plan: capacity_rows is smaller than the selection");
    if (rc == LC_OK && k > 0) {
        if (launch_str_sel_rows(descs, scan->d_symtabs, L, static_cast<const uint64_t*>(d_row_offsets), capacity_rows, k,
                                static_cast<uint64_t*>(d_row_refs), d_len, static_cast<uint8_t*>(d_row_valid), d_tiles,
                                static_cast<uint64_t*>(d_value_offsets), st) != hipSuccess ||
            hipMemcpyAsync(&bytes, static_cast<uint64_t*>(d_value_offsets) + k, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            rc = fail(LC_ERR_DEVICE, "gather plan: length pass failed");
    } else if (rc == LC_OK) {
        if (hipMemsetAsync(d_value_offsets, 0, 8, st) != hipSuccess) rc = fail(LC_ERR_DEVICE, "gather plan: memset failed");
    }
    *out_bytes = bytes;
    release();
    return rc;
    });
}

lc_status lc_scan_gather_bytes(lc_ctx* ctx, lc_scan* scan, const void* d_row_refs, const void* d_value_offsets,
                               uint64_t rows, void* d_data, void* stream) {
    return guarded([&]() -> lc_status {
    bind_device(ctx);
    if (!ctx || !scan || (rows && (!d_row_refs || !d_value_offsets || !d_data))) return fail(LC_ERR_INVALID, "null argument");
    if (!scan->is_str) return fail(LC_UNSUPPORTED, "lc_scan_gather_bytes covers byte-view columns");
    scan_note_stream(scan, static_cast<hipStream_t>(stream));
    LC_HIP(launch_str_decode_sel(static_cast<const StrDesc*>(scan->d_descs), scan->d_symtabs,
                                 static_cast<const uint64_t*>(d_row_refs), static_cast<const uint64_t*>(d_value_offsets), rows,
                                 nullptr, rows, ~uint64_t(0), static_cast<uint8_t*>(d_data), static_cast<hipStream_t>(stream)));
    return LC_OK;
    });
}

lc_status lc_scan_gather_bytes_async(lc_ctx* ctx, lc_scan* scan, const void* d_selection, void* d_row_offsets,
                                     void* d_row_refs, void* d_value_offsets, void* d_row_valid, uint64_t capacity_rows,
                                     void* d_data, uint64_t capacity_bytes, void* stream) {
    return guarded([&]() -> lc_status {
    bind_device(ctx);
    if (!ctx || !scan || !d_row_offsets || !d_row_refs || !d_value_offsets || !d_data || capacity_rows == 0)
        return fail(LC_ERR_INVALID, "null argument");
    if (!scan->is_str) return fail(LC_UNSUPPORTED, "lc_scan_gather_bytes covers byte-view columns");
    if (scan->n == 0) return LC_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    scan_note_stream(scan, st);
    const uint64_t n = scan->n;
    // scratch owned by the scan (grow only): [entry counts u32 x n | row lengths u32 x capacity | tile sums u64]
    const uint64_t tiles_len = std::max<uint64_t>(n, capacity_rows) / 1024 + 4;
    const size_t need = align_up(n * 4, 256) + align_up(capacity_rows * 4, 256) + tiles_len * 8;
    std::lock_guard<std::mutex> g(scan->mu);
    if (need > scan->gather_cap) {
        LC_HIP(hipStreamSynchronize(st));
        pool_release(ctx, scan->d_gather);
        scan->d_gather = static_cast<uint8_t*>(pool_alloc(ctx, need));
        scan->gather_cap = scan->d_gather ? need : 0;
        if (!scan->d_gather) return fail(LC_ERR_OOM, "hipMalloc (gather scratch)");
    }
    uint32_t* d_counts = reinterpret_cast<uint32_t*>(scan->d_gather);
    uint32_t* d_len = reinterpret_cast<uint32_t*>(scan->d_gather + align_up(n * 4, 256));
    uint64_t* d_tiles = reinterpret_cast<uint64_t*>(scan->d_gather + align_up(n * 4, 256) + align_up(capacity_rows * 4, 256));
    ScanLaunch L{};
    L.n_entries = scan->n;
    L.blocks_per_entry = scan->bpe;
    L.d_selection = static_cast<const uint64_t*>(d_selection);
    const StrDesc* descs = static_cast<const StrDesc*>(scan->d_descs);
    LC_HIP(launch_str_entry_offsets(descs, L, d_counts, d_tiles, static_cast<uint64_t*>(d_row_offsets), st));
    // rows beyond the (device-side) count keep length 0, so the scan over the whole capacity yields their offsets too
    LC_HIP(hipMemsetAsync(d_len, 0, capacity_rows * 4, st));
    LC_HIP(launch_str_sel_rows(descs, scan->d_symtabs, L, static_cast<const uint64_t*>(d_row_offsets), capacity_rows,
                               capacity_rows, static_cast<uint64_t*>(d_row_refs), d_len, static_cast<uint8_t*>(d_row_valid),
                               d_tiles, static_cast<uint64_t*>(d_value_offsets), st));
    LC_HIP(launch_str_decode_sel(descs, scan->d_symtabs, static_cast<const uint64_t*>(d_row_refs),
                                 static_cast<const uint64_t*>(d_value_offsets), 0,
                                 static_cast<const uint64_t*>(d_row_offsets) + n, capacity_rows, capacity_bytes,
                                 static_cast<uint8_t*>(d_data), st));
    return LC_OK;
    });
}

lc_status lc_scan_eval_hits(lc_ctx* ctx, lc_scan* scan, const lc_predicate* preds, uint32_t n_preds, const void* d_selection,
                            void* d_hits_out, uint64_t capacity, void* d_n_hits, void* d_hit_first, void* d_counts_out,
                            void* d_total_out, uint32_t flags, void* stream) {
    bind_device(ctx);
    if (!preds || n_preds == 0 || n_preds > 2) return fail(LC_ERR_INVALID, "lc_scan_eval_hits takes one or two predicates");
    if (!d_hits_out || !d_n_hits) return fail(LC_ERR_INVALID, "d_hits_out / d_n_hits is null");
    HitsOut h;
    h.d_hits = d_hits_out;
    h.cap = capacity;
    h.d_n_hits = d_n_hits;
    h.d_hit_first = d_hit_first;
    h.counters_zeroed = (flags & LC_HITS_COUNTERS_ZEROED) != 0;
    h.partitioned = (flags & LC_HITS_PARTITIONED) != 0;
    if (h.partitioned && capacity < LC_HITS_PARTITIONS) return fail(LC_ERR_INVALID, "a partitioned list needs a capacity of at least LC_HITS_PARTITIONS records");
    return scan_eval_impl(ctx, scan, &preds[0], d_selection, nullptr, nullptr, d_counts_out, nullptr,
                          static_cast<hipStream_t>(stream), n_preds == 2 ? &preds[1] : nullptr, d_total_out, false, &h);
}

lc_status lc_scan_mask_to_hits(lc_ctx* ctx, lc_scan* scan, const void* d_mask, void* d_hits_out, uint64_t capacity,
                               void* d_n_hits, void* d_hit_first, uint32_t flags, void* stream) {
    return guarded([&]() -> lc_status {
    bind_device(ctx);
    if (!ctx || !scan || !d_mask || !d_hits_out || !d_n_hits) return fail(LC_ERR_INVALID, "null argument");
    const bool parts = (flags & LC_HITS_PARTITIONED) != 0;
    if (parts && capacity < LC_HITS_PARTITIONS) return fail(LC_ERR_INVALID, "a partitioned list needs a capacity of at least LC_HITS_PARTITIONS records");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!(flags & LC_HITS_COUNTERS_ZEROED)) LC_HIP(launch_zero_small(d_n_hits, parts ? kHitParts * kHitCounterStride * 8u : 8u, st));
    if (scan->n == 0) return LC_OK;
    scan_note_stream(scan, st);
    LC_HIP(launch_mask_to_hits(scan->d_descs, scan->is_str, scan->n, static_cast<const uint64_t*>(d_mask),
                               static_cast<uint64_t*>(d_hits_out), capacity, static_cast<unsigned long long*>(d_n_hits),
                               static_cast<uint32_t*>(d_hit_first), parts ? kHitParts : 1u, st));
    return LC_OK;
    });
}

lc_status lc_hits_compact(lc_ctx* ctx, const void* d_hits, const void* d_n_hits, uint64_t capacity, void* d_hits_out,
                          uint64_t capacity_out, void* d_n_hits_out, void* stream) {
    return guarded([&]() -> lc_status {
    bind_device(ctx);
    if (!ctx || !d_hits || !d_n_hits || !d_hits_out || !d_n_hits_out) return fail(LC_ERR_INVALID, "null argument");
    if (d_hits == d_hits_out) return fail(LC_ERR_INVALID, "lc_hits_compact does not compact in place");
    if (capacity < LC_HITS_PARTITIONS) return fail(LC_ERR_INVALID, "a partitioned list has a capacity of at least LC_HITS_PARTITIONS records");
    if (ctx->device < 0) return fail(LC_ERR_DEVICE, "host-only context: no HIP device (there is no CPU fallback)");
    LC_HIP(hipSetDevice(ctx->device));
    LC_HIP(launch_hits_compact(static_cast<const uint64_t*>(d_hits), static_cast<const unsigned long long*>(d_n_hits), capacity,
                               static_cast<uint64_t*>(d_hits_out), capacity_out, static_cast<unsigned long long*>(d_n_hits_out),
                               static_cast<hipStream_t>(stream)));
    return LC_OK;
    });
}

lc_status lc_scan_filter_hits(lc_ctx* ctx, lc_scan* scan, const lc_predicate* pred, const void* d_hits_in, const void* d_n_hits_in,
                              uint64_t capacity_in, void* d_hits_out, uint64_t capacity_out, void* d_n_hits_out, uint32_t flags,
                              void* stream) {
    return guarded([&]() -> lc_status {
    bind_device(ctx);
    if (!ctx || !scan || !pred || !d_hits_in || !d_n_hits_in || !d_hits_out || !d_n_hits_out) return fail(LC_ERR_INVALID, "null argument");
    if (d_hits_in == d_hits_out) return fail(LC_ERR_INVALID, "lc_scan_filter_hits does not filter in place");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool parts = (flags & LC_HITS_PARTITIONED) != 0;
    if (parts && (capacity_in < LC_HITS_PARTITIONS || capacity_out < LC_HITS_PARTITIONS))
        return fail(LC_ERR_INVALID, "a partitioned list needs a capacity of at least LC_HITS_PARTITIONS records");
    if (!(flags & LC_HITS_COUNTERS_ZEROED)) LC_HIP(launch_zero_small(d_n_hits_out, parts ? kHitParts * kHitCounterStride * 8u : 8u, st));
    if (scan->n == 0 || capacity_in == 0) return LC_OK;
    if (scan->has_clamped || scan->has_fquant)
        return fail(LC_UNSUPPORTED, "squeezed entries: the mask form decides which rows need the backing array");
    HitsPredLaunch h{};
    h.descs = scan->d_descs;
    h.symtabs = scan->d_symtabs;
    h.hits_in = static_cast<const uint64_t*>(d_hits_in);
    h.n_in = static_cast<const unsigned long long*>(d_n_hits_in);
    h.cap_in = capacity_in;
    h.hits_out = static_cast<uint64_t*>(d_hits_out);
    h.cap_out = capacity_out;
    h.n_out = static_cast<unsigned long long*>(d_n_hits_out);
    h.parts = parts ? kHitParts : 1u;
    h.const_value = -1;
    if (!scan->is_str) {
        const lc_status ps = make_fixed_pred(scan->meta[0], pred, &h.fp);
        if (ps != LC_OK) return ps;
        h.lane_log2 = scan->lane_log2;
        scan_note_stream(scan, st);
        LC_HIP(launch_pred_hits(h, st));
        return LC_OK;
    }
    StrPredHost sp;
    const lc_status ps = make_str_pred(pred, &sp);
    if (ps != LC_OK) return ps;
    if (sp.p.mode == 3 && scan->any_fingerprints)
        return fail(LC_UNSUPPORTED, "general LIKE patterns apply to byte views without fingerprints (the reference requires "
                                    "%needle% on SubstringSearch columns)");
    h.lane_log2 = 0;
    h.op = pred->op;
    if (sp.p.mode == 2) {
        h.const_value = sp.p.const_value ? 1 : 0;
    } else {
        // compare ops take the literal; [NOT] LIKE '%needle%' the needle (byte-wise contains, like the scan kernels' automaton);
        // any other pattern goes to the general matcher as written
        if (pred->lit_len > uint64_t(kMaxNeedleBytes)) return fail(LC_UNSUPPORTED, "literal over 4096 bytes");
        h.h_lit = static_cast<const uint8_t*>(pred->lit);
        h.lit_len = uint32_t(pred->lit_len);
        if (sp.p.mode == 1) {
            const uint8_t* inner = nullptr;
            size_t il = 0;
            if (substring_pattern(static_cast<const uint8_t*>(pred->lit), size_t(pred->lit_len), &inner, &il)) {
                h.substring = 1;  // (the whole needle, also beyond the 63 bytes the scan kernels' automaton holds)
                h.h_lit = inner;
                h.lit_len = uint32_t(il);
            }
        }
        if (h.lit_len > uint32_t(kInlineNeedle)) {
            std::lock_guard<std::mutex> g(scan->mu);
            scan_enter_stream(scan, st);
            const size_t need = size_t(h.lit_len) + 16;
            LC_HIP(hipStreamSynchronize(st));  // a previous evaluation may still read the old literal
            if (need > scan->needle_cap) {
                pool_release(ctx, scan->d_needle);
                scan->d_needle = static_cast<uint8_t*>(pool_alloc(ctx, need));
                if (!scan->d_needle) { scan->needle_cap = 0; return fail(LC_ERR_OOM, "hipMalloc (needle)"); }
                scan->needle_cap = need;
            }
            LC_HIP(hipMemcpyAsync(scan->d_needle, h.h_lit, h.lit_len, hipMemcpyHostToDevice, st));
            LC_HIP(hipStreamSynchronize(st));
            h.d_lit = scan->d_needle;
        }
    }
    scan_note_stream(scan, st);
    LC_HIP(launch_pred_hits(h, st));
    return LC_OK;
    });
}

lc_status lc_scan_gather_fixed_hits(lc_ctx* ctx, lc_scan* scan, const void* d_hits, const void* d_n_hits, uint64_t capacity_rows,
                                    void* d_values_out, void* d_row_valid, uint32_t flags, void* stream) {
    return guarded([&]() -> lc_status {
    bind_device(ctx);
    if (!ctx || !scan || !d_hits || !d_n_hits || !d_values_out) return fail(LC_ERR_INVALID, "null argument");
    if (scan->is_str) return fail(LC_UNSUPPORTED, "lc_scan_gather_fixed_hits covers fixed-width columns (byte views: lc_scan_gather_bytes_hits)");
    if (scan->has_clamped || scan->has_fquant)
        return fail(LC_UNSUPPORTED, "squeezed entries: lc_scan_gather_fixed decides which reads need the backing array");
    if (scan->n == 0 || capacity_rows == 0) return LC_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    scan_note_stream(scan, st);
    LC_HIP(launch_fixed_gather_hits(static_cast<const FixedDesc*>(scan->d_descs), scan->lane_log2, static_cast<const uint64_t*>(d_hits),
                                    static_cast<const unsigned long long*>(d_n_hits), capacity_rows,
                                    static_cast<uint8_t*>(d_values_out), static_cast<uint8_t*>(d_row_valid),
                                    (flags & LC_HITS_PARTITIONED) ? kHitParts : 1u, st));
    return LC_OK;
    });
}

lc_status lc_scan_gather_bytes_hits(lc_ctx* ctx, lc_scan* scan, const void* d_hits, const void* d_n_hits, uint64_t capacity_rows,
                                    void* d_views, void* d_row_valid, void* d_data, uint64_t capacity_bytes, void* d_n_bytes,
                                    uint32_t flags, void* stream) {
    return guarded([&]() -> lc_status {
    bind_device(ctx);
    if (!ctx || !scan || !d_hits || !d_n_hits || !d_views || !d_n_bytes || (capacity_bytes && !d_data))
        return fail(LC_ERR_INVALID, "null argument");
    if (!scan->is_str) return fail(LC_UNSUPPORTED, "lc_scan_gather_bytes_hits covers byte-view columns");
    if (capacity_bytes > 0x7FFFFFFFull) return fail(LC_ERR_INVALID, "a BinaryView offset is an i32: capacity_bytes must stay below 2 GiB");
    const bool slotted = (flags & LC_GATHER_SLOTTED) != 0;
    if (slotted && capacity_rows > capacity_bytes / LC_GATHER_SLOT_BYTES)
        return fail(LC_ERR_INVALID, "LC_GATHER_SLOTTED: capacity_bytes must hold capacity_rows slots of LC_GATHER_SLOT_BYTES");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!(flags & LC_HITS_COUNTERS_ZEROED)) LC_HIP(launch_zero_small(d_n_bytes, 8, st));
    if (scan->n == 0 || capacity_rows == 0) return LC_OK;
    scan_note_stream(scan, st);
    LC_HIP(launch_str_gather_hits(static_cast<const StrDesc*>(scan->d_descs), scan->d_symtabs, static_cast<const uint64_t*>(d_hits),
                                  static_cast<const unsigned long long*>(d_n_hits), capacity_rows, static_cast<uint32_t*>(d_views),
                                  static_cast<uint8_t*>(d_row_valid), static_cast<uint8_t*>(d_data), capacity_bytes,
                                  static_cast<unsigned long long*>(d_n_bytes), slotted, (flags & LC_HITS_PARTITIONED) ? kHitParts : 1u, st));
    return LC_OK;
    });
}

// LiquidPrimitiveArray::squeeze with an ExtractDate32 hint (primitive_array.rs:389-420 -> SqueezedDate32Array::
// from_liquid_date32 / from_liquid_timestamp): the entries are replaced IN HBM by the bit-packed component (e.g. years
// 1992..1998: 3 bits per row instead of 12).  Everything runs on the device: decode, component, min / max, pack.
lc_status lc_squeeze_date(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, int32_t field) {
    return guarded([&]() -> lc_status {
    if (!ctx || (n && !entry_ids)) return fail(LC_ERR_INVALID, "null argument");
    if (field < LC_DATE_YEAR || field > LC_DATE_DAY_OF_WEEK) return fail(LC_ERR_INVALID, "unknown date field");
    if (n == 0) return LC_OK;
    if (ctx->device < 0) return fail(LC_ERR_DEVICE, "host-only context: no HIP device (there is no CPU fallback)");
    LC_HIP(hipSetDevice(ctx->device));
    lc_scan* scan = nullptr;
    lc_status rc = scan_create_impl(ctx, n, entry_ids, &scan, false);
    if (rc != LC_OK) return rc;
    std::unique_ptr<lc_scan, void (*)(lc_scan*)> guard(scan, lc_scan_destroy);
    const int64_t tpd = scan->is_str ? -1 : date_ticks_per_day(scan->meta[0]);
    if (tpd < 0) return fail(LC_UNSUPPORTED, "ExtractDate32 squeezing applies to Date32 / Timestamp entries");
    for (const Entry& e : scan->meta)
        if (date_ticks_per_day(e) != tpd) return fail(LC_ERR_INVALID, "entries of different types in one squeeze call");
    const uint64_t rows = scan->total_rows;
    const size_t vw = scan->meta[0].fd.value_width;
    uint8_t* d_vals = static_cast<uint8_t*>(pool_alloc(ctx, std::max<uint64_t>(rows, 1) * vw + 64));
    int32_t* d_comp = static_cast<int32_t*>(pool_alloc(ctx, std::max<uint64_t>(rows, 1) * 4 + 64));
    uint64_t* d_offs = static_cast<uint64_t*>(pool_alloc(ctx, (n + 1) * 8));
    struct Bufs {
        lc_ctx* c; void* a; void* b; void* d;
        ~Bufs() { (void)hipStreamSynchronize(nullptr); pool_release(c, a); pool_release(c, b); pool_release(c, d); }
    } bufs{ctx, d_vals, d_comp, d_offs};
    if (!d_vals || !d_comp || !d_offs) return fail(LC_ERR_OOM, "hipMalloc (squeeze scratch)");
    rc = lc_scan_gather_fixed(ctx, scan, nullptr, d_vals, rows * vw, d_offs, nullptr);   // decode every row, in order
    if (rc != LC_OK) return rc;
    LC_HIP(launch_date_component(d_vals, rows, int(vw), field, tpd, d_comp, nullptr));
    std::vector<DevEncodeItem> items(n);
    uint64_t row0 = 0;
    for (uint64_t i = 0; i < n; i++) {
        const Entry& e = scan->meta[i];
        DevEncodeItem& it = items[i];
        it.id = entry_ids[i];
        it.phys = e.phys;   // the original Arrow type stays the entry's type (original_data_type)
        it.vw = 4;          // components are i32 on u32 lanes (BitPackedArray<UInt32Type>)
        it.is_signed = true;
        it.n = e.len;
        it.has_validity = e.fd.validity != nullptr;
        it.d_values = reinterpret_cast<const uint8_t*>(d_comp + row0);
        it.d_validity = e.fd.validity;
        it.squeezed_field = field;
        it.force_all_null = e.all_null;
        row0 += e.len;
    }
    rc = device_encode_and_register(ctx, items);
    if (rc != LC_OK) return rc;
    // all-null entries keep their nullability (they carry no validity buffer in either form)
    std::unique_lock<std::shared_mutex> g(ctx->mu);
    for (uint64_t i = 0; i < n; i++) {
        auto it = ctx->entries.find(entry_ids[i]);
        if (it != ctx->entries.end()) it->second.nullable = scan->meta[i].nullable;
    }
    return LC_OK;
    });
}

// LiquidPrimitiveArray::squeeze with IntegerSqueezePolicy::Clamp (primitive_array.rs:589-660): integer entries of at
// least 8 bits per value are re-packed IN HBM at half their width; offsets at or above the sentinel 2^(W/2) - 1 are
// stored as the sentinel.  Entries that do not qualify (narrower, all null, floats / decimals, already squeezed) are
// left as they are.  *out_squeezed (optional) receives how many entries were squeezed.
// LiquidFloatArray::squeeze, FloatSqueezePolicy::Quantize (float_array.rs:338-395): new width = W / 2, shift = W - W / 2, a row
// keeps ((reference + offset) >> shift) - (reference >> shift).  That bucket can be 2^(W/2) — one more than the new width
// holds — whenever (reference mod 2^shift) + max offset carries into bit W; what the reference's packer makes of such a
// value is the business of the fastlanes crate (not in the tree), so those entries are left unsqueezed here: results stay
// exact and nothing is claimed about undefined behaviour.  Gather of the offsets, bucket arithmetic and packing run on the
// device; validity and the ALP exceptions move to the new blob unchanged.
static lc_status squeeze_float_quantize(lc_ctx* ctx, const std::vector<uint64_t>& all_ids, uint64_t* out_done) {
    std::map<int, std::vector<uint64_t>> by_lane;
    {
        std::shared_lock<std::shared_mutex> g(ctx->mu);
        for (uint64_t id : all_ids) {
            auto it = ctx->entries.find(id);
            if (it != ctx->entries.end()) by_lane[it->second.fd.lane_log2].push_back(id);
        }
    }
    for (auto& kv : by_lane) {
        const std::vector<uint64_t>& ids = kv.second;
        lc_scan* scan = nullptr;
        lc_status rc = scan_create_impl(ctx, ids.size(), ids.data(), &scan, false);
        if (rc != LC_OK) return rc;
        std::unique_ptr<lc_scan, void (*)(lc_scan*)> guard(scan, lc_scan_destroy);
        const uint64_t rows = scan->total_rows, m = scan->n;
        const size_t vw = size_t(1) << (scan->lane_log2 - 3);
        const size_t nblk = size_t(m) * scan->bpe;
        std::vector<FixedDesc> zero_ref(m);
        for (uint64_t i = 0; i < m; i++) {  // plain unsigned offsets: no reference, no ALP decode, no patches
            zero_ref[i] = scan->meta[i].fd;
            zero_ref[i].reference = 0;
            zero_ref[i].kind = kKindInt;
            zero_ref[i].is_signed = 0;
            zero_ref[i].patch_len = 0;
            zero_ref[i].patch_idx = nullptr;
            zero_ref[i].patch_val = nullptr;
            zero_ref[i].value_width = uint8_t(vw);
        }
        FixedDesc* d_descs0 = static_cast<FixedDesc*>(pool_alloc(ctx, m * sizeof(FixedDesc)));
        uint8_t* d_vals = static_cast<uint8_t*>(pool_alloc(ctx, std::max<uint64_t>(rows, 1) * vw + 64));
        uint8_t* d_scr = static_cast<uint8_t*>(pool_alloc(ctx, nblk * 4 + fixed_gather_offsets_len(nblk) * 8 + (m + 1) * 8 + 64));
        EncodeDesc* d_enc = static_cast<EncodeDesc*>(pool_alloc(ctx, m * sizeof(EncodeDesc)));
        EncodeMinMax* d_mm = static_cast<EncodeMinMax*>(pool_alloc(ctx, m * sizeof(EncodeMinMax)));
        struct Bufs {
            lc_ctx* c; void* p[5];
            ~Bufs() { (void)hipStreamSynchronize(nullptr); for (void* q : p) pool_release(c, q); }
        } bufs{ctx, {d_descs0, d_vals, d_scr, d_enc, d_mm}};
        if (!d_descs0 || !d_vals || !d_scr || !d_enc || !d_mm) return fail(LC_ERR_OOM, "hipMalloc (float quantize scratch)");
        LC_HIP(hipMemcpy(d_descs0, zero_ref.data(), m * sizeof(FixedDesc), hipMemcpyHostToDevice));
        uint64_t* d_bo = reinterpret_cast<uint64_t*>(d_scr);
        uint64_t* d_eo = d_bo + fixed_gather_offsets_len(nblk);
        uint32_t* d_bc = reinterpret_cast<uint32_t*>(d_eo + m + 1);
        ScanLaunch L{};
        L.n_entries = uint32_t(m);
        L.blocks_per_entry = scan->bpe;
        LC_HIP(launch_fixed_gather(d_descs0, scan->lane_log2, L, d_bc, d_bo, d_eo, d_vals, std::max<uint64_t>(rows, 1), nullptr));
        // largest offset of every entry -> largest bucket
        std::vector<EncodeDesc> enc(m);
        uint64_t row0 = 0;
        for (uint64_t i = 0; i < m; i++) {
            const Entry& e = scan->meta[i];
            enc[i] = EncodeDesc{};
            enc[i].values = d_vals + row0 * vw;
            enc[i].validity = nullptr;  // null slots hold whatever offset the packer stored: they are bucketed like the rest
            enc[i].n = e.len;
            enc[i].value_log2 = uint8_t(scan->lane_log2 - 3);
            row0 += e.len;
        }
        LC_HIP(hipMemcpy(d_enc, enc.data(), m * sizeof(EncodeDesc), hipMemcpyHostToDevice));
        LC_HIP(launch_col_minmax(d_enc, uint32_t(m), d_mm, nullptr));
        std::vector<EncodeMinMax> mm(m);
        LC_HIP(hipMemcpy(mm.data(), d_mm, m * sizeof(EncodeMinMax), hipMemcpyDeviceToHost));
        struct Lay { bool take; int new_w, shift; size_t begin, packed, valid, pidx, pval, bytes; };
        std::vector<Lay> lay(m);
        size_t total = 0;
        uint32_t max_rows = 0;
        for (uint64_t i = 0; i < m; i++) {
            const Entry& e = scan->meta[i];
            Lay& y = lay[i];
            y = Lay{};
            y.new_w = e.W / 2;
            y.shift = e.W - y.new_w;
            const int lane_bits = int(vw) * 8;
            // signed arithmetic in the lane width, as the reference's
            auto sx = [&](uint64_t v) { return lane_bits == 32 ? int64_t(int32_t(uint32_t(v))) : int64_t(v); };
            const int64_t ref = sx(e.fd.reference);
            const int64_t top = lane_bits == 32 ? int64_t(int32_t(uint32_t(uint64_t(ref) + mm[i].mx))) : int64_t(uint64_t(ref) + mm[i].mx);
            const uint64_t max_bucket = uint64_t((top >> y.shift) - (ref >> y.shift));
            y.take = max_bucket < (uint64_t(1) << y.new_w);
            if (!y.take) continue;
            y.begin = align_up(total, kSectionAlign);
            size_t cur = y.begin;
            y.packed = cur;
            cur = align_up(cur + packed_bytes(y.new_w, e.len) + 128, kSectionAlign);
            y.valid = y.pidx = y.pval = size_t(-1);
            if (e.fd.validity) { y.valid = cur; cur = align_up(cur + ((size_t(e.len) + 63) / 64) * 8, kSectionAlign); }
            if (e.fd.patch_len) {
                y.pidx = cur;
                cur = align_up(cur + size_t(e.fd.patch_len) * 8, kSectionAlign);
                y.pval = cur;
                cur = align_up(cur + size_t(e.fd.patch_len) * vw, kSectionAlign);
            }
            y.bytes = cur - y.begin;
            total = cur;
            max_rows = std::max(max_rows, e.len);
        }
        uint64_t n_take = 0;
        for (const Lay& y : lay) n_take += y.take;
        if (n_take == 0) continue;
        total = align_up(total, kSectionAlign) + 256;
        ArenaReservation reserved(ctx);
        uint8_t* dbase = nullptr;
        int slab = -1;
        {
            // the cache lock is held for the allocation and (below) for the publication only — not across the copies, the
            // pack kernel and the wait for them, which would stall every concurrent scan creation, stage and evict
            std::unique_lock<std::shared_mutex> g(ctx->mu);
            rc = arena_alloc(ctx, total, &dbase, &slab);
            if (rc != LC_OK) return rc;
            ctx->slabs[size_t(slab)].live += int64_t(n_take) - 1;
            reserved.arm(slab, int64_t(n_take));
        }
        LC_HIP(hipMemsetAsync(dbase, 0, total, nullptr));
        std::vector<EncodeDesc> pack;
        for (uint64_t i = 0; i < m; i++) {
            if (!lay[i].take) continue;
            const Entry& e = scan->meta[i];
            EncodeDesc d = enc[i];
            d.W = uint8_t(lay[i].new_w);
            d.reference = 0;
            d.packed = dbase + lay[i].packed;
            d.fq_shift = uint8_t(lay[i].shift);
            d.fq_ref = e.fd.reference;
            pack.push_back(d);
            if (e.fd.validity)
                LC_HIP(hipMemcpyAsync(dbase + lay[i].valid, e.fd.validity, ((size_t(e.len) + 63) / 64) * 8, hipMemcpyDeviceToDevice, nullptr));
            if (e.fd.patch_len) {
                LC_HIP(hipMemcpyAsync(dbase + lay[i].pidx, e.fd.patch_idx, size_t(e.fd.patch_len) * 8, hipMemcpyDeviceToDevice, nullptr));
                LC_HIP(hipMemcpyAsync(dbase + lay[i].pval, e.fd.patch_val, size_t(e.fd.patch_len) * vw, hipMemcpyDeviceToDevice, nullptr));
            }
        }
        LC_HIP(hipMemcpy(d_enc, pack.data(), pack.size() * sizeof(EncodeDesc), hipMemcpyHostToDevice));
        LC_HIP(launch_fl_pack(d_enc, uint32_t(pack.size()), max_rows, scan->lane_log2, nullptr));
        LC_HIP(hipStreamSynchronize(nullptr));
        std::unique_lock<std::shared_mutex> g(ctx->mu);
        uint64_t n_published = 0;
        for (uint64_t i = 0; i < m; i++) {
            if (!lay[i].take) continue;
            {
                // an id that was re-staged or evicted since the scan captured it keeps what is there now: the quantized form
                // was built from the captured entry (its share of the slab reservation is given back)
                auto cur = ctx->entries.find(ids[i]);
                if (cur == ctx->entries.end() || cur->second.uid != scan->meta[i].uid) {
                    arena_release(ctx, slab);
                    continue;
                }
            }
            n_published++;
            Entry e = scan->meta[i];  // type, length, ALP exponents, reference stay
            e.fd.mask_word_off = 0;
            e.orig_W = e.W;
            e.W = lay[i].new_w;
            e.fd.W = uint8_t(lay[i].new_w);
            e.fd.quantized = uint8_t(0x80u | uint32_t(lay[i].shift));
            e.quantized = true;
            e.fq_shift = lay[i].shift;
            e.slab = slab;
            e.device_bytes = lay[i].bytes;
            e.fd.packed = dbase + lay[i].packed;
            e.fd.validity = lay[i].valid == size_t(-1) ? nullptr : reinterpret_cast<const uint64_t*>(dbase + lay[i].valid);
            e.fd.patch_idx = lay[i].pidx == size_t(-1) ? nullptr : reinterpret_cast<const uint64_t*>(dbase + lay[i].pidx);
            e.fd.patch_val = lay[i].pval == size_t(-1) ? nullptr : dbase + lay[i].pval;
            publish_entry(ctx, ids[i], std::move(e));
        }
        reserved.disarm();
        if (out_done) *out_done += n_published;
    }
    return LC_OK;
}

static lc_status squeeze_half_width(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, uint64_t* out_squeezed, bool quantize) {
    return guarded([&]() -> lc_status {
    if (!ctx || (n && !entry_ids)) return fail(LC_ERR_INVALID, "null argument");
    if (out_squeezed) *out_squeezed = 0;
    if (n == 0) return LC_OK;
    if (ctx->device < 0) return fail(LC_ERR_DEVICE, "host-only context: no HIP device (there is no CPU fallback)");
    LC_HIP(hipSetDevice(ctx->device));
    // entries that qualify, grouped by lane width (a scan covers one lane type)
    std::map<int, std::vector<uint64_t>> by_lane;
    std::vector<uint64_t> float_ids;
    {
        std::shared_lock<std::shared_mutex> g(ctx->mu);
        for (uint64_t i = 0; i < n; i++) {
            auto it = ctx->entries.find(entry_ids[i]);
            if (it == ctx->entries.end()) return fail(LC_NOT_STAGED, "entry is not staged");
            const Entry& e = it->second;
            // integers (both policies) and decimals (LiquidDecimalArray::squeeze always quantizes, decimal_array.rs:300-345)
            const bool is_float = e.fd.kind == kKindF32 || e.fd.kind == kKindF64;
            if (quantize && is_float && !e.is_str && !e.all_null && e.W >= 8 && !e.quantized) {
                float_ids.push_back(entry_ids[i]);  // FloatSqueezePolicy::Quantize, the only float policy (float_array.rs:61-65)
                continue;
            }
            const bool kind_ok = e.fd.kind == kKindInt || (quantize && e.fd.kind == kKindDecimal);
            if (e.is_str || !kind_ok || e.all_null || e.W < 8 || e.clamped || e.quantized || e.squeezed_field >= 0) continue;
            // Date32 / Timestamp arrays are never half-width squeezed by the reference: their squeeze() only knows the
            // date-component form and returns None without such a hint (primitive_array.rs:398-411)
            if (date_ticks_per_day(e) >= 0) continue;
            by_lane[e.fd.lane_log2].push_back(entry_ids[i]);
        }
    }
    for (auto& kv : by_lane) {
        const std::vector<uint64_t>& ids = kv.second;
        lc_scan* scan = nullptr;
        lc_status rc = scan_create_impl(ctx, ids.size(), ids.data(), &scan, false);
        if (rc != LC_OK) return rc;
        std::unique_ptr<lc_scan, void (*)(lc_scan*)> guard(scan, lc_scan_destroy);
        const uint64_t rows = scan->total_rows, m = scan->n;
        const bool decimal = scan->meta[0].fd.kind == kKindDecimal;
        const size_t vw = decimal ? 8 : scan->meta[0].fd.value_width;   // decimals: the u64 offsets, not i128 values
        const size_t nblk = size_t(m) * scan->bpe;
        // descriptors with reference 0: the gather then yields the packed-domain offsets themselves
        std::vector<FixedDesc> zero_ref(m);
        for (uint64_t i = 0; i < m; i++) {
            zero_ref[i] = scan->meta[i].fd;
            zero_ref[i].reference = 0;
            if (decimal) { zero_ref[i].kind = kKindInt; zero_ref[i].value_width = 8; zero_ref[i].is_signed = 0; }
        }
        FixedDesc* d_descs0 = static_cast<FixedDesc*>(pool_alloc(ctx, m * sizeof(FixedDesc)));
        uint8_t* d_vals = static_cast<uint8_t*>(pool_alloc(ctx, std::max<uint64_t>(rows, 1) * vw + 64));
        uint8_t* d_scr = static_cast<uint8_t*>(pool_alloc(ctx, nblk * 4 + fixed_gather_offsets_len(nblk) * 8 + (m + 1) * 8 + 64));
        struct Bufs {
            lc_ctx* c; void* a; void* b; void* d;
            ~Bufs() { (void)hipStreamSynchronize(nullptr); pool_release(c, a); pool_release(c, b); pool_release(c, d); }
        } bufs{ctx, d_descs0, d_vals, d_scr};
        if (!d_descs0 || !d_vals || !d_scr) return fail(LC_ERR_OOM, "hipMalloc (clamp squeeze scratch)");
        LC_HIP(hipMemcpy(d_descs0, zero_ref.data(), m * sizeof(FixedDesc), hipMemcpyHostToDevice));
        uint64_t* d_bo = reinterpret_cast<uint64_t*>(d_scr);
        uint64_t* d_eo = d_bo + fixed_gather_offsets_len(nblk);
        uint32_t* d_bc = reinterpret_cast<uint32_t*>(d_eo + m + 1);
        ScanLaunch L{};
        L.n_entries = uint32_t(m);
        L.blocks_per_entry = scan->bpe;
        LC_HIP(launch_fixed_gather(d_descs0, scan->lane_log2, L, d_bc, d_bo, d_eo, d_vals, std::max<uint64_t>(rows, 1), nullptr));
        std::vector<DevEncodeItem> items(m);
        uint64_t row0 = 0;
        for (uint64_t i = 0; i < m; i++) {
            const Entry& e = scan->meta[i];
            DevEncodeItem& it = items[i];
            it.id = ids[i];
            it.phys = e.phys;
            it.vw = int(vw);
            it.is_signed = false;  // the gathered values are unsigned offsets
            it.entry_signed = e.fd.is_signed != 0;
            it.n = e.len;
            it.has_validity = e.fd.validity != nullptr;
            it.d_values = d_vals + row0 * vw;
            it.d_validity = e.fd.validity;
            it.forced = true;
            it.orig_W = e.W;
            it.forced_W = std::max(e.W / 2, 1);   // "new squeezed bit width is half of the original" (:612)
            it.clamp_max = (uint64_t(1) << it.forced_W) - 1;
            it.quantize = quantize;
            it.logical = e.logical;
            it.dec_precision = e.dec_precision;
            it.dec_scale = e.dec_scale;
            it.dec_is256 = e.dec_is256;
            it.entry_value_width = e.fd.value_width;
            it.entry_reference = e.fd.reference;
            row0 += e.len;
        }
        rc = device_encode_and_register(ctx, items);
        if (rc != LC_OK) return rc;
        if (out_squeezed) *out_squeezed += m;
    }
    if (!float_ids.empty()) {
        uint64_t done = 0;
        const lc_status rc = squeeze_float_quantize(ctx, float_ids, &done);
        if (rc != LC_OK) return rc;
        if (out_squeezed) *out_squeezed += done;
    }
    return LC_OK;
    });
}

lc_status lc_squeeze_clamp(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, uint64_t* out_squeezed) {
    return squeeze_half_width(ctx, n, entry_ids, out_squeezed, false);
}
lc_status lc_squeeze_quantize(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, uint64_t* out_squeezed) {
    return squeeze_half_width(ctx, n, entry_ids, out_squeezed, true);
}

lc_status lc_scan_date_part(lc_ctx* ctx, lc_scan* scan, void* d_values, uint64_t n_values, int32_t field, void* stream) {
    return guarded([&]() -> lc_status {
    bind_device(ctx);
    if (!ctx || !scan || !d_values) return fail(LC_ERR_INVALID, "null argument");
    if (field < LC_DATE_YEAR || field > LC_DATE_DAY_OF_WEEK) return fail(LC_ERR_INVALID, "unknown date field");
    if (scan->n == 0 || n_values == 0) return LC_OK;
    const int64_t tpd = scan->is_str ? -1 : date_ticks_per_day(scan->meta[0]);
    if (tpd < 0) return fail(LC_UNSUPPORTED, "ExtractDate32 applies to Date32 / Timestamp columns");
    LC_HIP(launch_date_lossy(d_values, n_values, int(scan->meta[0].fd.value_width), field, tpd,
                             static_cast<hipStream_t>(stream)));
    return LC_OK;
    });
}

}  // extern "C"
