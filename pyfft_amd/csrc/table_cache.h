// table_cache.h -- bookkeeping of the content-keyed device table cache (windows, FFT(window), filter spectra, response
// tables).  Pure host logic with no HIP calls, so that the eviction policy is unit-tested on the CPU
// (tests/cache_policy_test.cpp); spectral.hip supplies allocation and release.
//
// Policy: at most `cap` entries; a miss on a full cache evicts the LEAST RECENTLY USED entry that is neither
//   * obtained during the current API call (tick > call_start: the call may still launch kernels that read it), nor
//   * pinned by the caller (tables a pending sp_welch_accum keeps for sp_welch_finish).
// Nothing else is ever dropped, so a pointer handed out earlier in the same call stays valid until the call returns.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <map>

namespace sp {

struct TableEntry {
    void *dev;
    size_t bytes;
    uint64_t tick;      // last use
};

struct TableCache {
    std::map<uint64_t, TableEntry> map;
    uint64_t tick = 0, call_start = 0;
    size_t cap = 64;

    // every API entry point calls this once, under the library lock, before its first lookup
    void begin_call() { call_start = tick; }

    // hit: refreshes the entry's tick and returns it; miss: nullptr
    TableEntry *find(uint64_t key, size_t bytes) {
        auto it = map.find(key);
        if (it == map.end() || it->second.bytes != bytes) return nullptr;
        it->second.tick = ++tick;
        return &it->second;
    }
    bool full() const { return map.size() >= cap; }

    // the entry to drop before an insertion into a full cache; false if every entry is in use by the current call or pinned
    bool pick_victim(const void *const *pinned, int npinned, uint64_t *key_out) const {
        bool have = false;
        uint64_t best_tick = 0;
        for (const auto &kv : map) {
            if (kv.second.tick > call_start) continue;
            bool pin = false;
            for (int i = 0; i < npinned; ++i) pin = pin || (pinned[i] != nullptr && pinned[i] == kv.second.dev);
            if (pin) continue;
            if (!have || kv.second.tick < best_tick) {
                have = true;
                best_tick = kv.second.tick;
                *key_out = kv.first;
            }
        }
        return have;
    }
    // removes the entry and returns its device pointer (the caller frees it)
    void *erase(uint64_t key) {
        auto it = map.find(key);
        if (it == map.end()) return nullptr;
        void *d = it->second.dev;
        map.erase(it);
        return d;
    }
    // a stale entry under the same key (same content hash, other size) is returned for release
    void *insert(uint64_t key, void *dev, size_t bytes) {
        void *old = nullptr;
        auto it = map.find(key);
        if (it != map.end()) old = it->second.dev;
        map[key] = TableEntry{dev, bytes, ++tick};
        return old;
    }
};

}   // namespace sp
