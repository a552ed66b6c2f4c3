// tests/native/cache_lifetime_check.cpp -- the host-side lifetime logic of lanczos_cache.hpp without a GPU, under
// -fsanitize=address,undefined: retire lists, the bounded workgroup-table cache, the recency order of the plan cache.
//
// The handful of HIP calls the header makes are replaced by a small model of streams: work queued on a stream (an asynchronous
// copy, an event record) completes only when the test drains that stream, and the model objects to
//   * freeing a block that a queued copy still reads or writes (the round-3 harness crash: DESIGN.md 9),
//   * freeing a block twice, leaking one, recording on a destroyed stream (reported as an error code, as HIP does).
// Round 3's verdict asked for exactly this: create -> many shapes -> destroy of the cache + RetireList with stubbed events.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <vector>

// ------------------------------------------------------------------------------------------------ the model
typedef int hipError_t;
static const hipError_t hipSuccess = 0, hipErrorNotReady = 600, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2;
struct StubStream {
    bool alive = true;
    std::vector<int> pending_events;                       // ids of events recorded and not yet reached
    std::vector<std::pair<const void*, const void*>> copies;  // (dst, src) of copies queued and not yet done
};
struct StubEvent {
    int id;
    bool recorded = false, done = false;
};
typedef StubStream* hipStream_t;
typedef StubEvent* hipEvent_t;
static const unsigned hipEventDisableTiming = 2, hipHostMallocDefault = 0;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1 };

static std::set<void*> g_dev, g_host;
static std::map<int, StubEvent*> g_events;
static std::vector<StubStream*> g_streams;
static int g_next_event = 1, g_fail_host_malloc_at = -1, g_host_mallocs = 0, g_fail_copy_at = -1, g_copies = 0, g_violations = 0;

static void violation(const char* what) {
    printf("VIOLATION: %s\n", what);
    g_violations++;
}
static bool in_flight(const void* p) {
    for (StubStream* s : g_streams)
        for (auto& c : s->copies)
            if (c.first == p || c.second == p) return true;
    return false;
}
static hipError_t hipMalloc(void** p, size_t n) {
    *p = malloc(n ? n : 1);
    g_dev.insert(*p);
    return hipSuccess;
}
static hipError_t hipHostMalloc(void** p, size_t n, unsigned) {
    if (g_host_mallocs++ == g_fail_host_malloc_at) {
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    *p = malloc(n ? n : 1);
    g_host.insert(*p);
    return hipSuccess;
}
static hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    if (!g_dev.count(p)) violation("hipFree of a block that is not a live device block");
    if (in_flight(p)) violation("hipFree of a device block a queued copy still writes");
    g_dev.erase(p);
    free(p);
    return hipSuccess;
}
static hipError_t hipHostFree(void* p) {
    if (!p) return hipSuccess;
    if (!g_host.count(p)) violation("hipHostFree of a block that is not a live page-locked block");
    if (in_flight(p)) violation("hipHostFree of the page-locked source of a queued copy");
    g_host.erase(p);
    free(p);
    return hipSuccess;
}
static hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
    *e = new StubEvent{g_next_event++};
    g_events[(*e)->id] = *e;
    return hipSuccess;
}
static hipError_t hipEventDestroy(hipEvent_t e) {
    if (!e || !g_events.count(e->id)) {
        violation("hipEventDestroy of a dead event");
        return hipErrorInvalidValue;
    }
    g_events.erase(e->id);
    delete e;
    return hipSuccess;
}
static hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    if (!s || !s->alive) return hipErrorInvalidValue;   // a stream the caller has destroyed
    e->recorded = true;
    e->done = false;
    s->pending_events.push_back(e->id);
    return hipSuccess;
}
static hipError_t hipEventQuery(hipEvent_t e) { return !e->recorded || e->done ? hipSuccess : hipErrorNotReady; }
static void drain(hipStream_t s);
static hipError_t hipEventSynchronize(hipEvent_t e) {
    if (e->recorded && !e->done)
        for (StubStream* s : g_streams)
            for (int id : s->pending_events)
                if (id == e->id) {
                    drain(s);
                    return hipSuccess;
                }
    return hipSuccess;
}
static hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t s) {
    if (g_copies++ == g_fail_copy_at) return hipErrorInvalidValue;
    if (!s || !s->alive) return hipErrorInvalidValue;
    (void)n;
    s->copies.push_back({dst, src});   // the bytes move when the stream is drained
    return hipSuccess;
}
static void drain(hipStream_t s) {
    for (auto& c : s->copies) {
        if (!g_dev.count((void*)c.first)) violation("a queued copy wrote a freed device block");
        if (!g_host.count((void*)c.second)) violation("a queued copy read a freed page-locked block");
    }
    s->copies.clear();
    for (int id : s->pending_events)
        if (g_events.count(id)) g_events[id]->done = true;
    s->pending_events.clear();
}
static hipStream_t new_stream() {
    g_streams.push_back(new StubStream());
    return g_streams.back();
}

#define LZ_CACHE_TEST_STUBS
#include "lanczos_cache.hpp"

static int fails = 0;
#define EXPECT(c)                                         \
    do {                                                  \
        if (!(c)) {                                       \
            printf("FAILED line %d: %s\n", __LINE__, #c); \
            fails++;                                      \
        }                                                 \
    } while (0)

struct Entry {
    int frame, tx, m_b, m_e;
};
typedef lz::WgTabCacheT<Entry> Cache;
static void make_key(long long (&k)[8], int shape) {
    for (int i = 0; i < 8; i++) k[i] = 1000 * i + shape;
}

int main() {
    hipStream_t s1 = new_stream(), s2 = new_stream();
    {   // 1. a retired block waits for its events; a block whose event could not be recorded waits for the final reap
        lz::RetireList rl;
        void *a, *b, *h;
        hipMalloc(&a, 16), hipMalloc(&b, 16), hipHostMalloc(&h, 16, 0);
        rl.retire({a}, {h}, {s1, s2});
        rl.reap(false);
        EXPECT(g_dev.count(a) && g_host.count(h));          // both events pending
        drain(s1);
        rl.reap(false);
        EXPECT(g_dev.count(a));                             // s2 still pending
        drain(s2);
        rl.reap(false);
        EXPECT(!g_dev.count(a) && !g_host.count(h) && rl.list.empty());
        hipStream_t dead = new_stream();
        dead->alive = false;
        rl.retire({b}, {}, {dead});
        rl.reap(false);
        EXPECT(g_dev.count(b) && rl.list.size() == 1);      // NOT treated as drained
        rl.reap(true);
        EXPECT(!g_dev.count(b) && rl.list.empty());
    }
    EXPECT(g_events.empty());
    {   // 2. the bounded table cache: 200 shapes on a stream that is never drained in between -- evicted tables (device block AND the
        // page-locked source of the upload) must outlive the queued copies; a shape that stays hot is never evicted
        Cache c;
        long long hot[8];
        make_key(hot, 7);
        std::vector<Entry> tab(33, Entry{1, 2, 3, 4});
        hipError_t e = hipSuccess;
        for (int shape = 0; shape < 200; shape++) {
            long long k[8];
            make_key(k, shape);
            Cache::Item* it = c.find(k);
            EXPECT(it == nullptr || shape == 7);
            if (!it) it = c.insert(k, tab, 33, 1, false, shape % 3 ? s1 : s2, &e);
            EXPECT(it && e == hipSuccess && it->n == 33 && it->upload_stream == (shape % 3 ? s1 : s2) && it->streams.size() == 1);
            if (shape >= 7) {
                Cache::Item* h = c.find(hot);     // a hit refreshes the entry
                EXPECT(h != nullptr);
                if (h) lz::note_stream(h->streams, s2);
            }
            EXPECT(c.items.size() <= Cache::kMaxItems);
        }
        EXPECT(c.items.size() == Cache::kMaxItems && c.find(hot) != nullptr);
        EXPECT(!c.retired.list.empty());                    // nothing was drained: the evicted tables are all still held
        const size_t held = c.retired.list.size();
        EXPECT(held == 200 - Cache::kMaxItems);
        drain(s1);
        c.retired.reap(false);
        EXPECT(c.retired.list.size() < held && !c.retired.list.empty());   // the ones uploaded on s2 wait on
        drain(s2);
        c.retired.reap(false);
        EXPECT(c.retired.list.empty());
        // 3. failures inside insert(): nothing leaks, a queued copy's blocks go through the retire list
        long long k[8];
        make_key(k, 5000);
        g_fail_host_malloc_at = g_host_mallocs;
        EXPECT(c.insert(k, tab, 33, 1, false, s1, &e) == nullptr && e == hipErrorOutOfMemory);
        g_fail_host_malloc_at = -1;
        g_fail_copy_at = g_copies;
        EXPECT(c.insert(k, tab, 33, 1, false, s1, &e) == nullptr && e == hipErrorInvalidValue);
        g_fail_copy_at = -1;
        EXPECT(c.find(k) == nullptr);
        // 4. a stream the caller destroys while its tables are cached: eviction cannot record on it and keeps the blocks
        hipStream_t gone = new_stream();
        make_key(k, 6000);
        EXPECT(c.insert(k, tab, 33, 1, false, gone, &e) != nullptr);
        drain(gone);
        gone->alive = false;
        for (int shape = 7000; shape < 7000 + (int)Cache::kMaxItems; shape++) {
            make_key(k, shape);
            EXPECT(c.insert(k, tab, 33, 1, false, s1, &e) != nullptr);
        }
        bool kept = false;
        for (auto& r : c.retired.list) kept = kept || r.unrecorded;
        EXPECT(kept);
        drain(s1);
        c.retired.reap(false);
        kept = false;
        for (auto& r : c.retired.list) kept = kept || r.unrecorded;
        EXPECT(kept);                                       // still there: only the owner's final reap may free it
        // 5. lanczos_destroy: the device has drained, everything goes, nothing is recorded any more
        drain(s1), drain(s2);
        c.release_all();
        EXPECT(c.items.empty() && c.retired.list.empty());
    }
    EXPECT(g_dev.empty() && g_host.empty() && g_events.empty());
    {   // 6. destruction with copies still queued (a context dropped without lanczos_destroy): the destructor waits, then frees
        Cache c;
        long long k[8];
        std::vector<Entry> tab(5, Entry{0, 0, 0, 1});
        hipError_t e;
        for (int shape = 0; shape < 3; shape++) {
            make_key(k, shape);
            EXPECT(c.insert(k, tab, 5, 1, true, s1, &e) != nullptr);
        }
        c.drop(0);
        drain(s1);   // (release_all's precondition: the owner has drained the device)
    }
    EXPECT(g_dev.empty() && g_host.empty() && g_events.empty());
    {   // 7. recency order of the plan cache
        struct K { int a, b; };
        lz::LruOrder<K> o;
        for (int i = 0; i < 5; i++) o.touch(K{i, i});
        o.touch(K{0, 0});                                   // a hit on the oldest
        K first = o.pop_oldest();
        EXPECT(first.a == 1);
        o.touch(K{2, 2});
        first = o.pop_oldest();
        EXPECT(first.a == 3 && o.order.size() == 3);
    }
    for (StubStream* s : g_streams) delete s;
    if (fails || g_violations) {
        printf("cache lifetime: %d failed expectation(s), %d violation(s)\n", fails, g_violations);
        return 1;
    }
    printf("cache lifetime: all cases ok\n");
    return 0;
}
