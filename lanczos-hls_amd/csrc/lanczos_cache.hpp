// lanczos_cache.hpp -- host-side lifetime of what a context keeps on the device: retire lists, the workgroup-table cache of
// k_march, and the recency order of the plan cache.  No kernels here.
//
// Rule: a device block (or the page-locked source of an asynchronous upload) that launches still in flight may be reading is
// never freed on the spot and never with a device-wide sync: an event is recorded on every stream the block was used on, and
// the block is freed by a later call once those events have completed -- or at destruction, after the owner has drained the
// device.  If an event cannot be recorded (a stream the caller has destroyed in the meantime) the block is NOT treated as
// drained: it stays until the owner's final reap(true) behind a device-wide sync (lanczos_destroy).
//
// Why the page-locked source exists at all (DESIGN.md 9, round-3 harness crash): an intermediate state of round 3 uploaded the
// table with hipMemcpyAsync straight out of the std::vector that march_build_table had filled -- pageable memory whose copy the
// runtime may finish after the call has returned -- and the vector died at the end of the scope.  The copy then reads freed
// memory: mostly still mapped (silent), sometimes not (SIGSEGV with nothing on stdout, seen once in gpurun_out/r3g).  The
// committed code copies the table into a page-locked block owned by the cache item and frees it only through the retire list.
//
// The host logic is exercised without a GPU by tests/native/cache_lifetime_check.cpp (-fsanitize=address,undefined), which
// compiles this header against counting stand-ins for the few HIP calls it makes (LZ_CACHE_TEST_STUBS).
#pragma once
#include <algorithm>
#include <cstring>
#include <vector>

#ifndef LZ_CACHE_TEST_STUBS
#include <hip/hip_runtime.h>
#endif

namespace lz {

struct Retired {
    std::vector<void*> dev, host;   // hipFree / hipHostFree
    std::vector<hipEvent_t> ev;
    bool unrecorded = false;        // an event could not be recorded: only a reap behind a device-wide sync may free this
};
struct RetireList {
    std::vector<Retired> list;
    void retire(const std::vector<void*>& dev, const std::vector<void*>& host, const std::vector<hipStream_t>& streams) {
        Retired r;
        r.dev = dev;
        r.host = host;
        for (hipStream_t st : streams) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess && hipEventRecord(e, st) == hipSuccess) {
                r.ev.push_back(e);
            } else {
                if (e) (void)hipEventDestroy(e);
                r.unrecorded = true;
            }
        }
        list.push_back(r);
    }
    // wait == false: free what has provably drained.  wait == true: the caller has drained the device (or accepts waiting on
    // every recorded event); everything goes.
    void reap(bool wait) {
        for (size_t i = 0; i < list.size();) {
            bool done = !list[i].unrecorded || wait;
            for (hipEvent_t e : list[i].ev) {
                if (wait) (void)hipEventSynchronize(e);
                else if (hipEventQuery(e) != hipSuccess) done = false;
            }
            if (!done) {
                i++;
                continue;
            }
            for (hipEvent_t e : list[i].ev) (void)hipEventDestroy(e);
            for (void* p : list[i].dev) (void)hipFree(p);
            for (void* p : list[i].host) (void)hipHostFree(p);
            list.erase(list.begin() + i);
        }
    }
    ~RetireList() { reap(true); }
};
inline void note_stream(std::vector<hipStream_t>& v, hipStream_t s) {
    if (std::find(v.begin(), v.end(), s) == v.end()) v.push_back(s);
}

// recency order of a bounded cache: touch() on every hit and insert, oldest() names the entry to evict
template <typename Key>
struct LruOrder {
    std::vector<Key> order;   // least recently used first
    void touch(const Key& k) {
        for (size_t i = 0; i < order.size(); i++)
            if (memcmp(&order[i], &k, sizeof(Key)) == 0) {
                order.erase(order.begin() + i);
                break;
            }
        order.push_back(k);
    }
    bool empty() const { return order.empty(); }
    Key pop_oldest() {
        Key k = order.front();
        order.erase(order.begin());
        return k;
    }
};

// device copies of the workgroup tables a context has used (callers serialise per context).  A table is built once per launch
// shape into page-locked memory and uploaded asynchronously on the stream of its first launch; launches on other streams wait
// for that upload by event.  Bounded: the least recently used shape is retired when the 65th arrives.
template <typename Entry>
struct WgTabCacheT {
    static constexpr size_t kMaxItems = 64;
    struct Item {
        long long key[8];
        Entry* dev = nullptr;
        Entry* host = nullptr;             // page-locked source of the upload (stays valid while the copy is in flight)
        int n = 0, segs = 1;
        bool balanced = false;
        hipEvent_t uploaded = nullptr;     // recorded behind the upload
        hipStream_t upload_stream = nullptr;
        std::vector<hipStream_t> streams;  // streams this shape was uploaded / launched on
    };
    std::vector<Item> items;               // least recently used first
    RetireList retired;
    // set by every march_launch (query or launch): the batch size this kernel instance runs best at for this frame width -- the
    // largest one whose (strip, frame) pairs, cut in two chunks each, fill ONE resident round of workgroups with rank-aware
    // shares (config 2: 32 frames = 960 workgroups on 1 024 slots).  Larger batches are faster as several launches of this size
    // (64 frames: 2 x 207 us against 502 us in one launch, profiles/round3g_bench_default.json); 0: no preference
    int pref_frames = 0;
    // the largest batch that still is ONE resident round of two chunks per (strip, frame) pair (config 2: 34 frames): beyond it a
    // single launch falls back to one workgroup per slot with shares across pairs (48 frames: 399 us) and is slower than a launch of
    // pref_frames followed by the rest (32 + 16 frames: 207 + 113 us)
    int max_one_round_frames = 0;
    int wg_per_cu = 0;   // resident marching workgroups per CU of the instance the last query was about

    Item* find(const long long (&key)[8]) {
        for (size_t i = 0; i < items.size(); i++)
            if (memcmp(items[i].key, key, sizeof(key)) == 0) {
                if (i + 1 != items.size()) std::rotate(items.begin() + i, items.begin() + i + 1, items.end());  // most recent last
                return &items.back();
            }
        return nullptr;
    }
    // uploads `tab` on `stream` and returns the new item (nullptr + *err on failure).  Pointers to items are valid until the
    // next insert().
    Item* insert(const long long (&key)[8], const std::vector<Entry>& tab, int n, int segs, bool balanced, hipStream_t stream,
                 hipError_t* err) {
        retired.reap(false);
        Item it;
        memcpy(it.key, key, sizeof(key));
        it.n = n, it.segs = segs, it.balanced = balanced;
        const size_t bytes = sizeof(Entry) * tab.size();
        hipError_t e = hipMalloc((void**)&it.dev, bytes);
        if (e == hipSuccess) e = hipHostMalloc((void**)&it.host, bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&it.uploaded, hipEventDisableTiming);
        if (e == hipSuccess) {
            memcpy(it.host, tab.data(), bytes);
            e = hipMemcpyAsync(it.dev, it.host, bytes, hipMemcpyHostToDevice, stream);  // stream-ordered in front of the first launch
        }
        if (e == hipSuccess) e = hipEventRecord(it.uploaded, stream);
        if (e != hipSuccess) {
            // the copy may have been queued: nothing it touches is freed before that stream has drained
            if (it.uploaded) (void)hipEventDestroy(it.uploaded);
            retired.retire({it.dev}, {it.host}, {stream});
            *err = e;
            return nullptr;
        }
        it.upload_stream = stream;
        it.streams.push_back(stream);   // the upload itself reads `host` and writes `dev` on this stream
        if (items.size() >= kMaxItems) drop(0);  // bounded: the least recently used shape goes (freed once its launches have drained)
        items.push_back(it);
        *err = hipSuccess;
        return &items.back();
    }
    void drop(size_t i) {
        Item& it = items[i];
        retired.retire({it.dev}, {it.host}, it.streams);
        if (it.uploaded) (void)hipEventDestroy(it.uploaded);
        items.erase(items.begin() + i);
    }
    // the owner has made sure that nothing is in flight any more (lanczos_destroy: after the device has drained, before the
    // context's streams go): free everything now -- no events on streams that may be gone by the time a destructor runs
    void release_all() {
        retired.reap(true);
        for (Item& it : items) {
            (void)hipFree(it.dev);
            (void)hipHostFree(it.host);
            if (it.uploaded) (void)hipEventDestroy(it.uploaded);
        }
        items.clear();
    }
    ~WgTabCacheT() { release_all(); }
};

}  // namespace lz
