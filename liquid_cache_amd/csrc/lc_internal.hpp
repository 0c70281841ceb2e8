// Internals shared by the translation units behind the C ABI (lc_runtime.cpp, lc_like_pipeline.hip, lc_comm.cpp): the
// context / scan objects, the entry metadata, error plumbing and the device scratch pool.  Not installed, not part of
// the ABI.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <hip/hip_runtime.h>

#include "lc_host.hpp"
#include "lc_kernels.hpp"
#include "lc_transcode.hpp"

namespace lc {

// Message of the last failing call on this thread (a fixed buffer: recording an error can never throw).
lc_status fail(lc_status st, const char* msg) noexcept;
inline lc_status fail(lc_status st, const std::string& msg) noexcept { return fail(st, msg.c_str()); }

// Nothing may unwind across the C ABI (include/liquid_cache_amd.h): every entry point runs its body through this.
template <typename F>
lc_status guarded(F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(LC_ERR_OOM, "host allocation failed");
    } catch (const std::exception& e) {
        return fail(LC_ERR_INVALID, e.what());
    } catch (...) {
        return fail(LC_ERR_INVALID, "unexpected exception");
    }
}

// make VARIANT=trace EXTRA=-DLC_TRACE_PHASES: scoped wall-clock phases of the first evaluation of a scan on stderr (a
// profiling aid of scripts/first_eval_profile.py; compiled out of the shipped library)
#ifdef LC_TRACE_PHASES
struct TracePhase {
    const char* what;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit TracePhase(const char* w) : what(w) {}
    ~TracePhase() {
        static const std::chrono::steady_clock::time_point epoch = std::chrono::steady_clock::now();
        const auto t1 = std::chrono::steady_clock::now();
#ifdef LC_TRACE_MIN_US
        if (std::chrono::duration<double, std::micro>(t1 - t0).count() < double(LC_TRACE_MIN_US)) return;
#endif
        std::fprintf(stderr, "[phase] %-44s %9.1f us   (ends at %.1f us)\n", what, std::chrono::duration<double, std::micro>(t1 - t0).count(),
                     std::chrono::duration<double, std::micro>(t1 - epoch).count());
    }
};
#define LC_PHASE_CAT2(a, b) a##b
#define LC_PHASE_CAT(a, b) LC_PHASE_CAT2(a, b)
#define LC_PHASE(name) lc::TracePhase LC_PHASE_CAT(_phase_, __LINE__)(name)
#else
#define LC_PHASE(name)
#endif

#define LC_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return lc::fail(LC_ERR_DEVICE, (std::string(#expr) + ": " + hipGetErrorString(_e)).c_str()); \
    } while (0)

constexpr size_t kSlabBytes = size_t(256) << 20;
constexpr size_t kSectionAlign = 128;

struct Slab {
    uint8_t* base = nullptr;
    size_t size = 0, used = 0;
    int64_t live = 0;
};

struct Entry {
    bool is_str = false;
    int logical = 0, phys = 0;
    uint32_t len = 0;
    bool nullable = false, all_null = false;
    int W = 0;
    FixedDesc fd{};
    StrDesc sd{};
    uint64_t path_id = 0;
    uint32_t dict_len = 0;
    bool has_fp = false;
    int dec_precision = 0, dec_scale = 0, dec_is256 = 0;
    size_t device_bytes = 0;
    int slab = -1;
    uint32_t offsets_bytes = 0;  // compact offset residual bytes (byte views)
    uint32_t fsst_len = 0;
    // SqueezedDate32Array form (squeezed_date32_array.rs:46-53): the entry holds ONE calendar component of a Date32 /
    // Timestamp column, FoR + bit-packed on u32 lanes; `phys` keeps the original type.  -1: not squeezed.
    int squeezed_field = -1;
    // LiquidPrimitiveClampedArray form (hybrid_primitive_array.rs:73-80): packed at half the original width, offsets at or
    // above the sentinel 2^W - 1 are stored as the sentinel; orig_W is the width before the squeeze.
    bool clamped = false;
    int orig_W = 0;
    // LiquidPrimitiveQuantizedArray form (hybrid_primitive_array.rs:427-436): packed at half the original width, a row
    // holds the bucket (value - reference) / bucket_width.  Predicates only; every read needs the backing bytes.
    bool quantized = false;
    uint64_t bucket_width = 0;
    // LiquidFloatQuantizedArray form (float_array.rs:742-953): ALP-encoded values cut to half the bit width by a right shift
    // of the ABSOLUTE encoded value; fq_shift > 0 marks it (then `quantized` is set as well: every read needs the backing)
    int fq_shift = 0;
    bool sig_on_device = false;  // staging: the signature slices are still to be built by k_str_build_signatures
    uint64_t raw_bytes = 0;      // byte views: uncompressed size of the dictionary (RawFsstBuffer header)
    uint64_t index_hash = 0;     // byte views: hash of what the acceleration index was derived from (keys, validity, offsets,
                                 // FSST bytes, symbol table) — an "LCIX" blob is only taken back for exactly these bytes
    uint64_t uid = 0;            // unique per publication (publish_entry): a cached scan knows its entry was replaced
};

struct LikePipeline;  // lc_like_pipeline.hip: workgroup records + cached plans of k_like_lean

// The staged entries by id: open addressing over (id, node) slots, the 400-byte Entry records on the heap.  What it is for is
// find_many: lc_scan_create looks 12,207 ids up in a table of hundreds of thousands, and in a node-based std::unordered_map
// every look-up was two dependent cache misses (1.2-1.7 ms of a cold scan creation); here the slot of id i + 16 and the node
// of id i + 8 are prefetched while id i is copied.  The subset of the unordered_map interface the runtime uses; guarded by
// lc_ctx::mu like the map it replaces.
class EntryMap {
public:
    struct value_type {
        uint64_t first;
        Entry second;
    };
    class iterator {
    public:
        explicit iterator(value_type* p = nullptr) : p_(p) {}
        value_type* operator->() const { return p_; }
        value_type& operator*() const { return *p_; }
        bool operator==(const iterator& o) const { return p_ == o.p_; }
        bool operator!=(const iterator& o) const { return p_ != o.p_; }
    private:
        value_type* p_;
    };
    EntryMap() { rehash(1024); }
    ~EntryMap() {
        for (Slot& s : slots_)
            if (live(s)) delete s.node;
    }
    EntryMap(const EntryMap&) = delete;
    EntryMap& operator=(const EntryMap&) = delete;
    iterator end() const { return iterator(nullptr); }
    size_t size() const { return live_; }
    size_t count(uint64_t id) const { return probe(id) ? 1 : 0; }
    iterator find(uint64_t id) const { return iterator(probe(id)); }
    void erase(iterator it) {
        if (it == end()) return;
        size_t i = home(it->first);
        while (slots_[i].node != &*it) i = (i + 1) & mask_;
        delete slots_[i].node;
        slots_[i].node = tomb();
        live_--;
    }
    std::pair<iterator, bool> emplace(uint64_t id, Entry&& e) {
        if (value_type* p = probe(id)) return {iterator(p), false};
        if ((filled_ + 1) * 10 > slots_.size() * 7) rehash(live_ * 4 > slots_.size() ? slots_.size() * 2 : slots_.size());
        size_t i = home(id);
        while (slots_[i].node && slots_[i].node != tomb()) i = (i + 1) & mask_;
        if (!slots_[i].node) filled_++;
        slots_[i].key = id;
        slots_[i].node = new value_type{id, std::move(e)};
        live_++;
        return {iterator(slots_[i].node), true};
    }
    // visit(i, node of ids[i] or null) for i = 0 .. n - 1 in order, the slot of id i + 16 and the record of id i + 8 prefetched
    // while id i is visited: the visitor finds its record in L1 / L2 (looked up first and copied afterwards, the 4.9 MB of
    // 12,207 records had fallen back to L3 by the time they were copied — a cold lc_scan_create stayed at 1.3 ms).
    // Returns false as soon as the visitor does.
    template <typename F>
    bool visit_many(const uint64_t* ids, size_t n, F&& visit) const {
        constexpr size_t kAhead = 16, kNode = 8;
        value_type* ring[kNode];
        for (size_t i = 0; i < n + kAhead; i++) {
            if (i < n) __builtin_prefetch(&slots_[home(ids[i])]);
            if (i >= kAhead) {  // (the record probed kNode iterations ago)
                if (!visit(i - kAhead, ring[(i - kAhead) % kNode])) return false;
            }
            if (i >= kNode && i - kNode < n) {
                value_type* p = probe(ids[i - kNode]);
                ring[(i - kNode) % kNode] = p;
                if (p) {
                    const char* c = reinterpret_cast<const char*>(p);
                    for (size_t o = 0; o < sizeof(value_type); o += 64) __builtin_prefetch(c + o);
                }
            }
        }
        return true;
    }
    // out[i] = the node of ids[i] or null, with the slots and nodes of the ids ahead prefetched
    void find_many(const uint64_t* ids, size_t n, value_type** out) const {
        constexpr size_t kAhead = 16, kNode = 8;
        for (size_t i = 0; i < n + kAhead; i++) {
            if (i < n) __builtin_prefetch(&slots_[home(ids[i])]);
            if (i >= kNode && i - kNode < n) {
                value_type* p = probe(ids[i - kNode]);
                out[i - kNode] = p;
                if (p) {
                    const char* c = reinterpret_cast<const char*>(p);
                    __builtin_prefetch(c);
                    __builtin_prefetch(c + 64);
                    __builtin_prefetch(c + 128);
                    __builtin_prefetch(c + 192);
                    __builtin_prefetch(c + 256);
                    __builtin_prefetch(c + 320);
                    __builtin_prefetch(c + 384);
                }
            }
        }
    }

private:
    struct Slot {
        uint64_t key = 0;
        value_type* node = nullptr;
    };
    static value_type* tomb() { return reinterpret_cast<value_type*>(uintptr_t(1)); }
    static bool live(const Slot& s) { return s.node && s.node != tomb(); }
    size_t home(uint64_t id) const { return size_t((id * 0x9E3779B97F4A7C15ull) >> shift_); }
    value_type* probe(uint64_t id) const {
        for (size_t i = home(id);; i = (i + 1) & mask_) {
            const Slot& s = slots_[i];
            if (!s.node) return nullptr;
            if (s.node != tomb() && s.key == id) return s.node;
        }
    }
    void rehash(size_t n) {
        size_t cap = 1024;
        while (cap < n) cap <<= 1;
        std::vector<Slot> old;
        old.swap(slots_);
        slots_.assign(cap, Slot{});
        mask_ = cap - 1;
        shift_ = 64;
        for (size_t c = cap; c > 1; c >>= 1) shift_--;
        filled_ = 0;
        for (const Slot& s : old) {
            if (!live(s)) continue;
            size_t i = home(s.key);
            while (slots_[i].node) i = (i + 1) & mask_;
            slots_[i] = s;
            filled_++;
        }
    }
    std::vector<Slot> slots_;
    size_t mask_ = 0, live_ = 0, filled_ = 0;
    unsigned shift_ = 54;
};

}  // namespace lc

struct lc_ctx {
    int device = 0;
    hipDeviceProp_t props{};
    std::shared_mutex mu;
    lc::EntryMap entries;
    std::vector<lc::Slab> slabs;
    uint64_t max_hbm = 0;
    std::atomic<uint64_t> staged_bytes{0};  // slab capacity reserved on the device (what max_hbm bounds); written under `mu`,
                                            // read by the index builders without it
    uint64_t entry_bytes = 0;   // sum of the staged entries' blobs (what lc_device_info reports)
    std::atomic<uint64_t> index_bytes{0};  // scan-level LIKE indexes alive (live scans + the ones kept for the next scan): derived
                                           // data outside the slabs, charged to max_hbm when one is built
    // symbol tables
    std::mutex st_mu;
    std::unordered_map<uint64_t, uint32_t> symtab_slot;
    std::vector<std::unique_ptr<lc::SymbolTable>> symtabs;
    // options (lc_ctx_set_option): written under `mu`, read by evaluations and staging calls of other threads without it
    std::atomic<bool> build_signatures{true};   // LC_OPT_SIGNATURE_INDEX = 0 disables the bigram index (plain reference layout only)
    std::atomic<bool> signatures_on_host{false};  // LC_OPT_HOST_BUILT_INDEX = 1: build the index on the host (the device builder's oracle)
    std::atomic<uint64_t> like_index_budget{0};  // LC_OPT_LIKE_INDEX_BUDGET_BYTES: scan-level LIKE index bytes allowed alive (0: no
                                                 // bound of its own — max_hbm_bytes and free device memory still apply)
    std::atomic<uint32_t> like_index_cache{4};   // LC_OPT_LIKE_INDEX_CACHE: indexes of destroyed scans kept for the next scan over
                                                 // the same entries
    std::atomic<bool> like_index_async{true};    // LC_OPT_LIKE_INDEX_ASYNC: scan-level LIKE indexes are built by the context's builder
                                                 // thread on its own stream while the entry-level index answers (0: the first
                                                 // LIKE of a scan waits for the build, as before round 6)
    std::atomic<uint32_t> scan_cache_max{32};     // LC_OPT_SCAN_CACHE: destroyed scans kept for the next lc_scan_create over the same
                                                 // entry-id list (0: none)
    std::atomic<bool> comm_shared_memory{false};  // LC_OPT_COMM_SHARED_MEMORY: lc_comm_* of this DEVICE context run over the shared-memory
                                                  // test backend (ranks that share one GPU: dry runs, never a measurement)
    std::atomic<bool> like_many_hint{true};  // LC_OPT_LIKE_MANY_HINT (A/B aid): unselective planned needles take k_str_pred's sequential walker
    std::atomic<int> like_path{0};  // LC_OPT_LIKE_PATH: 0 auto, 1 k_str_pred, 2 auto without the scan-level index, 3 / 4 / 5 k_like_lean /
                                    // k_like_flat / k_like_scanall for every needle
    std::atomic<uint32_t> like_pipeline_min_entries{32};  // LC_OPT_LIKE_PIPELINE_MIN_ENTRIES: smaller scans are not planned (k_str_pred)
    std::atomic<bool> build_postings{true};       // LC_OPT_ROW_LISTS = 0: no inverted row lists (rows always mapped through the keys)
    lc::DevSymtab* d_symtabs = nullptr;
    size_t d_symtabs_cap = 0;
    size_t d_symtabs_uploaded = 0;
    // earlier (smaller) generations of the device array: launches that captured them may still be in flight on some
    // stream, so they are kept until the context goes away (a few hundred KB per doubling)
    std::vector<lc::DevSymtab*> d_symtabs_retired;
    // Recycled device scratch for the per-call drop-in API (descriptor arrays, masks, gather buffers): hipMalloc /
    // hipFree cost 0.1-1 ms each and hipFree synchronises the device, which would dominate an 8192-row call.
    std::mutex pool_mu;
    // Small and medium blocks are carved out of CHUNKS (64 MiB of device memory, 8 MiB pinned) that the context allocates when it
    // is created and when one runs full: in a process with some allocation history a hipMalloc takes 0.5-1 ms, a hipHostMalloc
    // 0.5-0.8 ms and a hipFree 3 ms — the first query of a fresh context made fifteen of them.  Carved blocks are recycled
    // through the size-class lists for ever and go back to the driver with their chunk (lc_ctx_destroy).
    std::vector<void*> pool_chunks, hpool_chunks;
    uint8_t *pool_chunk_cur = nullptr, *pool_chunk_end = nullptr, *hpool_chunk_cur = nullptr, *hpool_chunk_end = nullptr;
    void* pool_chunk_spare = nullptr;    // the next device chunk, allocated ahead by the builder thread when the current one runs low
    bool pool_spare_requested = false;  // (both under pool_mu)
    std::unordered_map<void*, size_t> pool_live;                // pointer -> size class (bytes)
    std::unordered_map<size_t, std::vector<void*>> pool_free;   // size class -> cached blocks
    // same for pinned host staging (pageable hipMemcpy runs at a fraction of the PCIe rate)
    std::unordered_map<void*, size_t> hpool_live;
    std::unordered_map<size_t, std::vector<void*>> hpool_free;
    // side streams for staging work (signature builder): concurrent lc_stage calls of different host threads do not wait
    // for each other's kernels the way they would on the null stream with a device-wide synchronise
    std::vector<hipStream_t> stream_pool;
    // Every host thread that calls into the context keeps ONE stream of its own for its calls (stream_acquire binds it on the
    // thread's first call; creating a stream costs ~10 ms, so a pool that hands streams around still created them inside
    // the callers' hot loops whenever more threads overlapped than before).  all_streams owns them: destroyed with the
    // context, whatever thread holds them.
    uint64_t uid = 0;                      // unique among the contexts of the process (thread-local bindings refer to it)
    std::vector<hipStream_t> all_streams;  // guarded by pool_mu
    // Idle ONE-ENTRY scans of the per-entry drop-in calls (lc_eval_predicate, lc_get_with_selection, ...), by entry id: a
    // partition thread evaluates the batches of its row range one entry at a time, and building a scan (descriptor upload,
    // pins, scratch) per call cost more than the kernel.  A call checks a scan out (exclusive use) and back in; scans of
    // evicted / replaced entries move to the graveyard under ctx->mu and are destroyed outside it (scan_cache_reap).
    std::atomic<uint64_t> next_uid{0};
    // scan-level LIKE indexes of destroyed scans (lc_like_pipeline.hip): a host that creates a scan per query over the same
    // entries gets the index (and the plans) of the previous one instead of rebuilding 2-3 GB in 13 ms
    std::mutex like_orphans_mu;
    std::vector<lc::LikePipeline*> like_orphans;  // most recently orphaned last
    // Slots for LIKE plans made on the fly (lc_like_pipeline.hip): device counters + their pinned mirror + an event each, carved
    // out of two blocks allocated once (releasing per-plan allocations back to pools that were full cost a hipHostFree —
    // 0.26 ms — on the query path of a host that walks hundreds of small scans)
    std::mutex plan_slots_mu;
    unsigned long long* plan_slots_d = nullptr;
    uint64_t* plan_slots_h = nullptr;
    std::vector<hipEvent_t> plan_slot_ev;   // created on first use, kept
    std::vector<uint32_t> plan_slots_free;
    bool plan_slots_tried = false;
    std::mutex index_reserve_mu;                  // one index reservation (budget check + eviction) at a time
    std::atomic<uint64_t> index_events{0};        // bumped whenever index memory is freed or becomes reclaimable (a scan that holds
                                                  // an index is given back): a scan that found no room tries again only after one
    std::mutex scan_cache_mu;
    std::unordered_map<uint64_t, std::vector<lc_scan*>> scan_cache;
    size_t scan_cache_size = 0;
    std::vector<lc_scan*> scan_graveyard;
    // Whole scans given back with lc_scan_destroy, kept for the next lc_scan_create over the SAME entry-id list while none of
    // their entries has been replaced or evicted (the reference's reader holds no scan objects: it names entries per query,
    // liquid_cache_reader.rs:264-339 — a host that follows it creates a scan per query).  Guarded by scan_cache_mu; most
    // recently destroyed last; any replacement / eviction of an entry moves all of them to the graveyard.
    std::vector<lc_scan*> list_cache;
    std::atomic<uint64_t> evict_epoch{0};  // bumped (under `mu`, exclusively) whenever an entry is replaced or evicted
    // The builder: ONE worker thread with its own (lowest-priority) stream runs the scan-level index builds off the query path.
    std::mutex builder_mu;
    std::condition_variable builder_cv;
    std::deque<std::packaged_task<void()>> builder_q;
    std::thread builder;
    bool builder_started = false, builder_stop = false;
    hipStream_t builder_stream = nullptr;  // created by the worker
};

struct lc_scan {
    lc_ctx* ctx = nullptr;
    bool is_str = false;
    int lane_log2 = 0;
    uint32_t n = 0, bpe = 0;
    uint32_t max_w = 0;  // widest entry (fixed width): <= 32 selects the register-resident predicate kernel
    uint32_t min_w = 0;  // narrowest entry that has packed data (W > 0); min_w == max_w: the kernel of that one width
    bool has_clamped = false;              // some entry is clamp-squeezed: evaluations first look for unresolved sentinels
    bool has_fquant = false;               // some entry is a float-quantized hybrid (k_float_quant_pred evaluates those)
    bool fquant_patches = false;           // ... and carries ALP exceptions: no selection can be applied (see lc_squeeze_quantize)
    std::vector<uint32_t> needs_backing;   // entries (scan order) whose last evaluation needs the full array
    std::vector<uint64_t> seg_offsets;  // n+1 word offsets
    std::vector<uint32_t> lens;         // rows per entry: two scans cover the same row ranges iff these are equal
    uint64_t total_rows = 0;
    void* d_descs = nullptr;
    uint64_t* d_seg_offsets = nullptr;
    std::vector<lc::Entry> meta;  // copies of the entries' metadata (descs have mask_word_off filled in)
    // What later passes over ALL entries need, in compact arrays captured while an entry's 400-byte record is in cache: the loops
    // of a scan's first LIKE walked `meta` four times (12,207 records: 0.13 + 0.17 ms when the records had left the cache again).
    std::vector<uint64_t> uids;         // Entry::uid, in scan order
    std::vector<uint32_t> symtab_slots; // byte views: StrDesc::symtab_slot, in scan order
    uint32_t slot_lo = 0, slot_hi = 0;  // ... their range: the LIKE automata are folded for these tables only
    bool str_index_everywhere = true;   // byte views: every entry with a dictionary carries signatures, row lists and fingerprints
    uint32_t max_str_rows = 0;          // byte views: the largest StrDesc::n
    uint32_t max_dict_rows = 0;         // ... among the entries that have a dictionary (an all-null entry has none)
    uint64_t entry_bytes_total = 0;     // sum of Entry::device_bytes
    std::vector<std::pair<int, uint32_t>> slab_pins;  // (slab, number of this scan's entries in it): what arena_pin_counts took
    std::vector<uint64_t> ids;    // the entry ids the scan was created over (list_cache: an identical list gets this scan back)
    std::vector<uint64_t> id_bloom;  // 2^17-bit Bloom filter of `ids` (two probes): does an evicted / replaced id concern this scan?
    bool cacheable = false;       // created by lc_scan_create (not a one-entry scan of the per-entry calls)
    uint64_t evict_epoch = 0;     // ctx->evict_epoch when the entries were captured (or last verified current)
    uint32_t* d_group_ends = nullptr;       // lc_scan_eval_count_groups: the row groups' entry bounds on the device ...
    std::vector<uint32_t> group_ends_host;  // ... as uploaded last
    uint32_t* d_group_entry_counts = nullptr;  // ... and the per-entry counts its evaluation writes (n x u32)
    // LIKE scratch
    uint8_t* d_automata = nullptr;
    size_t automata_cap = 0;
    std::vector<uint8_t> automata_needle;  // the automata in d_automata were built for this needle ...
    size_t automata_symtabs = 0;           // ... over this many symbol tables (0: nothing cached)
    uint8_t* d_needle = nullptr;
    size_t needle_cap = 0;
    lc::StrWgRecord* d_wg_ranges = nullptr;  // byte views: one record per workgroup (<= 4 entries of one symbol table)
    uint32_t n_wg_ranges = 0;
    uint32_t* d_wg_begins = nullptr;  // first entry of every record (device) and the pinned block it was uploaded from: kept
    void* h_wg_begins = nullptr;      // until the scan goes, so that nothing waits for the copy
    uint8_t* d_gather = nullptr;  // scratch of lc_scan_gather_bytes_async (grow only)
    size_t gather_cap = 0;
    uint32_t* d_work = nullptr;  // kWorkGroupsMax x {next entry, finished waves} (64-byte stride): dynamic entry
                                 // assignment of the persistent byte-view scan kernel
    // device symbol tables as of scan creation: every entry of the scan references a slot below n_symtabs, and an array
    // generation is never freed while the context lives, so launches need no lock against concurrent staging
    const lc::DevSymtab* d_symtabs = nullptr;
    size_t n_symtabs = 0;
    void* d_agg_partials = nullptr;  // lc_scan_aggregate: per-entry partials (allocated once)
    // facts about the entries, gathered once at creation
    uint32_t max_dict_len = 0;
    bool any_without_signatures = false, any_patch = false, any_fingerprints = false, any_float = false;
    bool any_multi_empty = false;          // byte views: some dictionary holds several empty values (k_like_scanall is not used)
    bool last_like_scanall = false;        // the last [NOT] LIKE evaluation went to k_like_scanall (EXPLAIN, byte model)
    bool last_eq_flat = false;             // the last `=` / `<>` evaluation went to k_like_flat (EXPLAIN, byte model)
    int32_t uniform_slot = -1;             // byte views: the symbol-table slot when every entry shares one, else -1
    uint64_t* d_or_tmp = nullptr;  // lc_scan_eval_or: [hit | valid | valid of the first column] scratch (grow only)
    size_t or_tmp_words = 0;
    unsigned long long* d_total_acc = nullptr;  // fused COUNT(*) accumulator (kTotalWords u64, zero between launches)
    uint64_t* d_mask_scratch = nullptr;    // mask words of evaluations whose caller wants none (d_mask_out == NULL) but whose
                                           // kernel cannot skip them (allocated on first need)
    int last_like_kernel = 0;              // LC_LIKE_KERNEL_*: what answered the last [NOT] LIKE / indexed `=` (lc_scan_info_get)
    bool last_native_hits = false;         // the last evaluation's kernel emitted the hit list itself (k_like_flat)
    lc::LikePipeline* like = nullptr;  // workgroup records + plans of k_like_lean (lc_like_pipeline.hip)
    bool pinned = false;  // the slabs of `meta` are pinned (arena_pin) until the scan is destroyed
    // The scan's scratch (automata, work counters, COUNT(*) accumulator, OR / aggregate temporaries) is used by
    // asynchronous launches after `mu` is released.  Calls on one scan are ordered on one stream; when a call arrives on
    // another stream, the previous one is drained first (scan_enter_stream) so the scratch is never shared in flight.
    hipStream_t last_stream = nullptr;
    bool used = false;
    std::vector<hipStream_t> streams_used;  // every stream a launch over this scan went to: lc_scan_destroy drains these
                                            // (not the device: other threads' calls keep running)
    std::mutex mu;
};


namespace lc {

hipStream_t stream_acquire(lc_ctx* ctx);
void stream_release(lc_ctx* ctx, hipStream_t s);
void* pool_alloc(lc_ctx* ctx, size_t bytes);
void pool_release(lc_ctx* ctx, void* p);  // the caller guarantees that nothing in flight still uses `p`
bool pool_release_if_owned(lc_ctx* ctx, void* p);  // false: not a block of the pool
void* host_pool_alloc(lc_ctx* ctx, size_t bytes);
void host_pool_release(lc_ctx* ctx, void* p);
void arena_pin(lc_ctx* ctx, int slab_idx);
void arena_release(lc_ctx* ctx, int slab_idx);  // caller holds ctx->mu exclusively
lc_status sync_symtabs(lc_ctx* ctx);
void scan_enter_stream(lc_scan* s, hipStream_t stream);  // caller holds s->mu
void scan_note_stream(lc_scan* s, hipStream_t stream);   // read-only use of the scan's descriptors on `stream`

struct StrPredHost {
    StrPred p{};
    std::vector<uint8_t> needle;
    std::vector<uint8_t> verify;  // p.verify_len != 0: the literal pattern the accepted values are matched against
};
lc_status make_str_pred(const lc_predicate* p, StrPredHost* out);

// lc_like_pipeline.hip.  Caller holds s->mu and has built the automata of `sp`.  *handled: the evaluation was launched
// by the pipeline; otherwise the caller launches k_str_pred.
lc_status like_pipeline_eval(lc_ctx* ctx, lc_scan* s, const StrPredHost& sp, const ScanLaunch& L, hipStream_t stream,
                             bool* handled, bool* many_candidates);
void like_pipeline_destroy(lc_ctx* ctx, LikePipeline* lp);
// blocks until the scan's index builds in flight have finished and their results are in place (lc_scan_index_wait,
// lc_scan_explain, lc_scan_info_get); caller holds s->mu
void like_pipeline_wait(lc_scan* s);
// the context's builder thread (lc_runtime.cpp): runs `fn` on the worker, the future completes when it returns
std::future<void> builder_submit(lc_ctx* ctx, std::function<void(hipStream_t)> fn);
void builder_shutdown(lc_ctx* ctx);
// lc_scan_destroy: keeps a pipeline that has a scan-level index for the next scan over the same entries (bounded), destroys
// the others.  like_orphans_clear: context teardown.
void like_pipeline_orphan(lc_ctx* ctx, LikePipeline* lp);
void like_orphans_clear(lc_ctx* ctx);
void plan_slots_destroy(lc_ctx* ctx);
void plan_slots_prime(lc_ctx* ctx);  // lc_ctx_create (device contexts)
std::string like_pipeline_explain(const lc_scan* s, const StrPredHost& sp);           // caller holds s->mu
uint64_t like_pipeline_bytes(const lc_scan* s, const StrPredHost& sp, bool with_counts, uint32_t sparse_flags = 0);  // caller holds s->mu
// what the scan's pipeline holds (lc_scan_info_get); caller holds s->mu
void like_pipeline_info(const lc_scan* s, uint64_t* bigram_bytes, uint64_t* unigram_bytes, double* build_ms, uint32_t* n_plans,
                        int32_t* build_pending);

}  // namespace lc
