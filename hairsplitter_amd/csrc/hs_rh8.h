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


// The same map over storage the caller provides (LDS on the device): info / key / tmp hold `cap` bytes each; a map that would
// need more slots than that sets `overflow` and stops growing (the kernels give it cap 512, which holds every key set of the path, and trap on
// the flag rather than go on with another order: tests/harness/rh8_selftest worstcase). cap 128 holds 64 buckets (51 keys),
// cap 512 holds 256 buckets (every set of byte keys the path can produce: 125 pileup codes + 3).
struct Rh8View {
    uint8_t* info; uint8_t* key; uint8_t* tmp;
    int cap;
    uint64_t mult;
    int mask, nslots, count, limit;
    uint32_t inc, shift;
    bool overflow;

    HS_HD void init(uint8_t* info_, uint8_t* key_, uint8_t* tmp_, int cap_) {
        info = info_; key = key_; tmp = tmp_; cap = cap_;
        mult = 0xc4ceb9fe1a85ec53ull; mask = 0; nslots = 0; count = 0; limit = 0; inc = 32; shift = 0; overflow = false;
    }
    HS_HD void home(uint8_t k, int& idx, uint32_t& inf) const {
        uint64_t h = (uint64_t)k;
        h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33;
        h *= mult; h ^= h >> 33;
        inf = inc + (uint32_t)((h & 31u) >> shift);
        idx = (int)((h >> 5) & (uint64_t)mask);
    }
    HS_HD bool alloc(int buckets) {
        if (Rh8::slots_for(buckets) + 8 > cap) { overflow = true; return false; }
        mask = buckets - 1; nslots = Rh8::slots_for(buckets); count = 0; limit = Rh8::load_limit(buckets);
        inc = 32; shift = 0;
        for (int i = 0; i < nslots + 8; ++i) info[i] = 0;
        return true;
    }
    HS_HD bool widen_distance_bits() {
        if (inc <= 2) return false;
        inc >>= 1; shift++;
        for (int i = 0; i < nslots; ++i) info[i] = (uint8_t)((info[i] >> 1) & 0x7f);
        limit = Rh8::load_limit(mask + 1);
        return true;
    }
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
        int n_old = 0;
        for (int i = 0; i < nslots; ++i) if (info[i] != 0) tmp[n_old++] = key[i];
        if (!alloc(buckets)) return;
        for (int i = 0; i < n_old; ++i) reinsert(tmp[i]);
    }
    HS_HD void grow() {
        if (mask == 0) { alloc(8); return; }
        if (count < Rh8::load_limit(mask + 1) && widen_distance_bits()) return;
        mult += 0xc4ceb9fe1a85ec54ull;
        if (count * 2 < Rh8::load_limit(mask + 1)) rebuild(mask + 1);
        else rebuild((mask + 1) * 2);
    }
    HS_HD void insert(uint8_t k) {
        for (int attempt = 0; attempt < 256 && !overflow; ++attempt) {
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
    HS_HD int order(uint8_t* out) const {
        int n = 0;
        for (int i = 0; i < nslots; ++i) if (info[i] != 0) out[n++] = key[i];
        return n;
    }
};

// std::sort of libstdc++ (GCC 11: introsort with a median-of-three pivot, threshold 16, heap sort at the depth limit, final
// insertion sort) on an array of packed (count << 8 | key) elements with the comparator "count greater" -- the call of
// call_variants.cpp:497-501. std::sort is not stable: where the reference sorts more than 16 (key, count) pairs with equal
// counts, which key ends up in front is decided by this very sequence of swaps, restated here so that the device can follow it.
// Checked against std::sort itself by tests/harness/rh8_selftest.cpp (random, tie-heavy and adversarial inputs).
struct CountSort {
    static HS_HD bool comp(uint32_t a, uint32_t b) { return (a >> 8) > (b >> 8); }
    static HS_HD void swp(uint32_t* a, int i, int j) { const uint32_t t = a[i]; a[i] = a[j]; a[j] = t; }
    static HS_HD void unguarded_linear_insert(uint32_t* a, int last) {
        const uint32_t val = a[last];
        int next = last - 1;
        while (comp(val, a[next])) { a[last] = a[next]; last = next; --next; }
        a[last] = val;
    }
    static HS_HD void insertion_sort(uint32_t* a, int first, int last) {
        if (first == last) return;
        for (int i = first + 1; i != last; ++i) {
            if (comp(a[i], a[first])) { const uint32_t val = a[i]; for (int j = i; j > first; --j) a[j] = a[j - 1]; a[first] = val; }
            else unguarded_linear_insert(a, i);
        }
    }
    static HS_HD void adjust_heap(uint32_t* a, int first, int hole, int len, uint32_t value) {
        const int top = hole;
        int child = hole;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            if (comp(a[first + child], a[first + (child - 1)])) child--;
            a[first + hole] = a[first + child];
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            a[first + hole] = a[first + (child - 1)];
            hole = child - 1;
        }
        int parent = (hole - 1) / 2;
        while (hole > top && comp(a[first + parent], value)) { a[first + hole] = a[first + parent]; hole = parent; parent = (hole - 1) / 2; }
        a[first + hole] = value;
    }
    static HS_HD void heap_sort(uint32_t* a, int first, int last) {      // __partial_sort(first, last, last)
        const int len = last - first;
        if (len >= 2) {
            for (int parent = (len - 2) / 2;; --parent) { adjust_heap(a, first, parent, len, a[first + parent]); if (parent == 0) break; }
        }
        while (last - first > 1) {
            --last;
            const uint32_t value = a[last];
            a[last] = a[first];
            adjust_heap(a, first, 0, last - first, value);
        }
    }
    static HS_HD int partition_pivot(uint32_t* a, int first, int last) {
        const int mid = first + (last - first) / 2;
        const int x = first + 1, y = mid, z = last - 1;      // median of the three to `first`
        if (comp(a[x], a[y])) {
            if (comp(a[y], a[z])) swp(a, first, y);
            else if (comp(a[x], a[z])) swp(a, first, z);
            else swp(a, first, x);
        } else if (comp(a[x], a[z])) swp(a, first, x);
        else if (comp(a[y], a[z])) swp(a, first, z);
        else swp(a, first, y);
        int lo = first + 1, hi = last;
        for (;;) {
            while (comp(a[lo], a[first])) ++lo;
            --hi;
            while (comp(a[first], a[hi])) --hi;
            if (!(lo < hi)) return lo;
            swp(a, lo, hi);
            ++lo;
        }
    }
    static HS_HD void sort(uint32_t* a, int n) { int stack[120]; sort_with_stack(a, n, stack); }
    // `stack`: 120 ints of scratch (LDS on the device, so that the kernel has no private segment)
    static HS_HD void sort_with_stack(uint32_t* a, int n, int* stack) {
        if (n <= 0) return;
        int lg = 0;
        for (int v = n; v > 1; v >>= 1) lg++;
        // __introsort_loop with its recursion on the right part unrolled into a stack of (first, last, depth)
        int* st_first = stack; int* st_last = stack + 40; int* st_depth = stack + 80;
        int sp = 0;
        st_first[0] = 0; st_last[0] = n; st_depth[0] = 2 * lg; sp = 1;
        while (sp > 0) {
            --sp;
            const int first = st_first[sp];
            int last = st_last[sp], depth = st_depth[sp];
            // the reference recurses into [cut, last) BEFORE it goes on with [first, cut): the order of the two does not change the
            // result (they touch disjoint ranges), so the right parts are simply stacked
            while (last - first > 16) {
                if (depth == 0) { heap_sort(a, first, last); break; }
                --depth;
                const int cut = partition_pivot(a, first, last);
                if (sp < 40) { st_first[sp] = cut; st_last[sp] = last; st_depth[sp] = depth; ++sp; }
                last = cut;
            }
        }
        if (n > 16) { insertion_sort(a, 0, 16); for (int i = 16; i < n; ++i) unguarded_linear_insert(a, i); }
        else insertion_sort(a, 0, n);
    }
};

}  // namespace hs
