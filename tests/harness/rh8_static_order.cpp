// TEST INFRASTRUCTURE: the rule k_robust_partitions uses to break ties between second alleles without replaying the
// insertions on the hash-map emulator -- "a robin_hood map of up to 12 char keys iterates them by (home bucket ascending, low
// five hash bits descending, insertion order), 8 buckets up to 6 keys, 16 buckets and the next multiplier from 7 to 12, unless
// some key sits 6 or more slots from its home bucket" -- against hs_rh8.h (itself pinned to the reference's header by
// tests/golden/robin_hood_order.json) on random key sets, half of them from small alphabets so that buckets collide.
// Prints "checked <n> bad <m> ..."; exit code 1 if any pair of keys with different ranks comes out in another order.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../hairsplitter_amd/csrc/hs_rh8.h"
static void home_low(uint8_t k, uint64_t mult, int buckets, int& home, int& low) {
    uint64_t h = k; h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= mult; h ^= h >> 33;
    home = (int)((h >> 5) & (uint64_t)(buckets - 1)); low = (int)(h & 31);
}
int main(int argc, char** argv) {
    const long trials = argc > 1 ? std::atol(argv[1]) : 2000000;
    long bad = 0, checked = 0, collisions = 0, displaced = 0;
    srand(3);
    const uint64_t m0 = 0xc4ceb9fe1a85ec53ull, m1 = m0 + 0xc4ceb9fe1a85ec54ull;
    for (long it = 0; it < trials; ++it) {
        int n = 1 + rand() % 12;
        const int alpha = (it & 1) ? 125 : 24;
        if (n > alpha) n = alpha;
        std::vector<uint8_t> keys;
        const int base = 33 + rand() % (126 - alpha);
        while ((int)keys.size() < n) { const uint8_t k = (uint8_t)(base + rand() % alpha); if (std::find(keys.begin(), keys.end(), k) == keys.end()) keys.push_back(k); }
        hs::Rh8 rh; rh.clear();
        for (uint8_t k : keys) rh.insert(k);
        uint8_t ord[300]; const int m = rh.order(ord);
        const bool big = n > 6;
        std::vector<int> rank((size_t)n);
        int cnt[16] = {0};
        for (int i = 0; i < n; ++i) { int home, low; home_low(keys[(size_t)i], big ? m1 : m0, big ? 16 : 8, home, low); rank[(size_t)i] = home * 32 + (31 - low); cnt[home]++; }
        bool far = false;
        if (big) { int carry = 0; for (int b = 0; b < 16; ++b) { if (cnt[b] > 0 && carry + cnt[b] - 1 >= 6) far = true; carry = std::max(0, carry + cnt[b] - 1); } }
        if (far) { displaced++; continue; }      // the kernel hands these to the emulator
        int where[256];
        for (int i = 0; i < m; ++i) where[ord[i]] = i;
        bool dup = false;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
            if (i != j && rank[(size_t)i] == rank[(size_t)j]) dup = true;
            if (rank[(size_t)i] < rank[(size_t)j] && where[keys[(size_t)i]] > where[keys[(size_t)j]]) bad++;
        }
        if (m != n) bad++;
        collisions += dup; checked++;
    }
    std::printf("checked %ld bad %ld sets-with-equal-ranks %ld sets-with-a-far-key %ld\n", checked, bad, collisions, displaced);
    return bad ? 1 : 0;
}
