// CPU unit test of the device-table cache policy (pyfft_amd/csrc/table_cache.h); built and run by
// tests/test_host_cpu.py::test_table_cache_policy with g++ (no HIP).  "Device pointers" are malloc'ed bytes.
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include "table_cache.h"

using namespace sp;

static std::set<void *> live;
static void *alloc_tab() {
    void *p = malloc(16);
    live.insert(p);
    return p;
}
static void free_tab(void *p) {
    assert(live.count(p) == 1);            // never a double free
    live.erase(p);
    free(p);
}
// what spectral.hip's get_table_keyed does around the policy
static void *get(TableCache &c, uint64_t key, const void *const *pinned, int npinned) {
    if (TableEntry *e = c.find(key, 16)) return e->dev;
    uint64_t victim;
    while (c.full() && c.pick_victim(pinned, npinned, &victim)) free_tab(c.erase(victim));
    void *d = alloc_tab();
    void *old = c.insert(key, d, 16);
    assert(old == nullptr);
    return d;
}

int main() {
    TableCache c;
    c.cap = 8;
    // fill the cache over several calls
    for (uint64_t k = 0; k < 8; ++k) {
        c.begin_call();
        get(c, k, nullptr, 0);
    }
    assert(c.map.size() == 8 && live.size() == 8);
    // the round-1 bug: one call obtains table A (a hit, cached long ago), then misses on B with the cache full --
    // A must survive the eviction and stay valid until the call returns
    c.begin_call();
    void *A = get(c, 0, nullptr, 0);               // oldest entry, now touched by this call
    void *B = get(c, 100, nullptr, 0);             // miss -> evicts the LRU entry that is NOT A: key 1
    assert(live.count(A) == 1 && live.count(B) == 1);
    assert(c.map.count(0) == 1 && c.map.count(1) == 0 && c.map.size() == 8);
    void *C = get(c, 101, nullptr, 0);             // second miss in the same call: A and B both survive, key 2 goes
    assert(live.count(A) && live.count(B) && live.count(C) && c.map.count(2) == 0);
    // a pending accumulate pins its two tables across calls
    const void *pinned[2] = {c.map[3].dev, c.map[4].dev};
    for (uint64_t k = 200; k < 230; ++k) {
        c.begin_call();
        get(c, k, pinned, 2);
        assert(c.map.count(3) == 1 && c.map.count(4) == 1 && c.map.size() == 8);
        assert(live.count((void *)pinned[0]) && live.count((void *)pinned[1]));
    }
    // LRU order: touching an entry protects it from the next eviction
    c.begin_call();
    uint64_t oldest = 0, oldest_tick = ~0ull;
    for (auto &kv : c.map)
        if (kv.first != 3 && kv.first != 4 && kv.second.tick < oldest_tick) {
            oldest = kv.first;
            oldest_tick = kv.second.tick;
        }
    c.begin_call();
    get(c, oldest, pinned, 2);                     // refresh
    c.begin_call();
    get(c, 999, pinned, 2);
    assert(c.map.count(oldest) == 1);
    // a call that itself touches more tables than the cache holds grows the cache instead of freeing live tables
    c.begin_call();
    for (uint64_t k = 1000; k < 1020; ++k) get(c, k, nullptr, 0);
    for (uint64_t k = 1000; k < 1020; ++k) assert(c.map.count(k) == 1);
    assert(c.map.size() >= 20);
    // and shrinks back on later calls
    for (uint64_t k = 2000; k < 2040; ++k) {
        c.begin_call();
        get(c, k, nullptr, 0);
    }
    assert(c.map.size() == 8);
    for (auto &kv : c.map) assert(live.count(kv.second.dev) == 1);
    assert(live.size() == c.map.size());
    // what sp_shutdown's tables_release does: every table the cache still owns is freed exactly once (the sanitizer build's
    // leak check holds the test to it)
    for (auto &kv : c.map) free_tab(kv.second.dev);
    c.map.clear();
    assert(live.empty());
    printf("cache policy ok\n");
    return 0;
}
