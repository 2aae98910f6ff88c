// hs_rh8.h -- iteration order of robin_hood::unordered_flat_map<unsigned char, int> (3.11.1), the only
// hash-map order that is observable on the hot path (call_variants.cpp:477-501,:837-844; Partition.cpp:59-66;
// separate_reads.cpp:1086-1099). Fixed-size, allocation-free, usable from host and device code.
// Only what the path needs: insert-if-absent in a given sequence, then walk the slots in ascending order.
// Behaviour restated from the published algorithm of robin_hood.h (keyToIdx :1349-1361, insert :2330-2380,
// shiftUp :1377-1397, info-byte overflow :2383-2411, growth :2413-2443, rehash :2203-2237); pinned by
// tests/golden/robin_hood_order.json.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define HS_HD __host__ __device__
#else
#define HS_HD
#endif

namespace hs {

struct Rh8 {
    static constexpr int kCap = 784;   // 512 buckets + 255 overflow slots + sentinel/padding
    uint64_t mult;
    int mask;          // buckets - 1, 0 when nothing is allocated yet
    int nslots;        // buckets + min(80% of buckets, 255)
    int count, limit;  // elements, and the element count at which the table grows (0 = "must restructure")
    uint32_t inc, shift;
    uint8_t info[kCap];
    uint8_t key[kCap];

    HS_HD void clear() { mult = 0xc4ceb9fe1a85ec53ull; mask = 0; nslots = 0; count = 0; limit = 0; inc = 32; shift = 0; }

    HS_HD static int load_limit(int buckets) { return buckets * 80 / 100; }
    HS_HD static int slots_for(int buckets) { int m = load_limit(buckets); return buckets + (m < 255 ? m : 255); }

    HS_HD void home(uint8_t k, int& idx, uint32_t& inf) const {
        uint64_t h = (uint64_t)k;
        h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33;
        h *= mult; h ^= h >> 33;
        inf = inc + (uint32_t)((h & 31u) >> shift);
        idx = (int)((h >> 5) & (uint64_t)mask);
    }
    HS_HD void alloc(int buckets) {
        mask = buckets - 1; nslots = slots_for(buckets); count = 0; limit = load_limit(buckets);
        inc = 32; shift = 0;
        for (int i = 0; i < nslots + 8; ++i) info[i] = 0;
    }
    HS_HD bool widen_distance_bits() {
        if (inc <= 2) return false;
        inc >>= 1; shift++;
        for (int i = 0; i < nslots; ++i) info[i] = (uint8_t)((info[i] >> 1) & 0x7f);
        limit = load_limit(mask + 1);
        return true;
    }
    // place a key known to be absent, starting the probe at (idx, inf)
    HS_HD void place(uint8_t k, int idx, uint32_t inf) {
        const int ins = idx;
        const uint32_t ins_inf = inf;
        if (ins_inf + inc > 0xFF) limit = 0;
        while (info[idx] != 0) idx++;
        for (int i = idx; i != ins; --i) {
            key[i] = key[i - 1];
            info[i] = (uint8_t)(info[i - 1] + inc);
            if ((uint32_t)info[i] + inc > 0xFF) limit = 0;
        }
        key[ins] = k; info[ins] = (uint8_t)ins_inf; count++;
    }
    HS_HD void reinsert(uint8_t k) {
        if (limit == 0) widen_distance_bits();
        int idx; uint32_t inf;
        home(k, idx, inf);
        while (inf <= info[idx]) { idx++; inf += inc; }
        place(k, idx, inf);
    }
    HS_HD void rebuild(int buckets) {
        uint8_t old_key[kCap];
        int n_old = 0;
        for (int i = 0; i < nslots; ++i) if (info[i] != 0) old_key[n_old++] = key[i];
        alloc(buckets);
        for (int i = 0; i < n_old; ++i) reinsert(old_key[i]);   // ascending old slot order, as the rehash loop does
    }
    HS_HD void grow() {
        if (mask == 0) { alloc(8); return; }
        if (count < load_limit(mask + 1) && widen_distance_bits()) return;
        mult += 0xc4ceb9fe1a85ec54ull;
        if (count * 2 < load_limit(mask + 1)) rebuild(mask + 1);
        else rebuild((mask + 1) * 2);
    }
    // insert-if-absent
    HS_HD void insert(uint8_t k) {
        for (int attempt = 0; attempt < 256; ++attempt) {
            if (mask == 0) { grow(); continue; }
            int idx; uint32_t inf;
            home(k, idx, inf);
            while (inf < info[idx]) { idx++; inf += inc; }
            while (inf == info[idx]) {
                if (key[idx] == k) return;
                idx++; inf += inc;
            }
            if (count >= limit) { grow(); continue; }
            place(k, idx, inf);
            return;
        }
    }
    // writes the keys in iteration order, returns how many
    HS_HD int order(uint8_t* out) const {
        int n = 0;
        for (int i = 0; i < nslots; ++i) if (info[i] != 0) out[n++] = key[i];
        return n;
    }
};

}  // namespace hs
