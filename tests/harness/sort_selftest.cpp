// TEST INFRASTRUCTURE: the two restatements the device needs to break ties exactly as the reference does -- hs::Rh8View (the
// hash map over caller-provided storage) against hs::Rh8 (pinned to the reference's header by robin_hood_order.json), and
// hs::CountSort against std::sort of this machine's libstdc++ (the one oracle/_ref is built with) called the way
// call_variants.cpp:497-501 calls it -- on random, tie-heavy and adversarial inputs (McIlroy's "killer adversary" drives the
// quicksort part to its depth limit, so that the heap-sort branch is compared too).
// Prints "rh8view <n> bad <m>; sort <n> bad <m> heap-paths <k>"; exit code 1 on any difference.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>
#include "../../hairsplitter_amd/csrc/hs_rh8.h"

static long g_heap_paths = 0;

// adversary: values are decided lazily so that the pivot candidates always turn out small (M. D. McIlroy, "A Killer Adversary for Quicksort")
struct Adversary {
    std::vector<int> val; int gas, nsolid = 0, candidate = 0;
    explicit Adversary(int n) : val((size_t)n), gas(n - 1) { for (int& v : val) v = gas; }
    int freeze(int x) { return val[(size_t)x] = nsolid++; }
    bool less(int x, int y) {      // "x before y"
        if (val[(size_t)x] == gas && val[(size_t)y] == gas) { if (x == candidate) freeze(x); else freeze(y); }
        if (val[(size_t)x] == gas) candidate = x; else if (val[(size_t)y] == gas) candidate = y;
        return val[(size_t)x] < val[(size_t)y];
    }
};

int main(int argc, char** argv) {
    const long trials = argc > 1 ? std::atol(argv[1]) : 200000;
    srand(11);
    long bad_rh = 0, n_rh = 0;
    for (long it = 0; it < trials / 4; ++it) {
        const int n = 1 + rand() % ((it & 3) == 0 ? 128 : 40);
        std::vector<uint8_t> keys;
        const int alpha = (it & 1) ? 125 : 30, base = 33 + rand() % (126 - alpha);
        for (int i = 0; i < n; ++i) keys.push_back((uint8_t)(base + rand() % alpha));
        if (it & 4) { keys.push_back(0); keys.push_back(1); keys.push_back(2); }
        hs::Rh8 a; a.clear();
        for (uint8_t k : keys) a.insert(k);
        uint8_t oa[800], ob[800];
        const int na = a.order(oa);
        for (int cap : {128, 512}) {
            std::vector<uint8_t> info((size_t)cap), key((size_t)cap), tmp((size_t)cap);
            hs::Rh8View b; b.init(info.data(), key.data(), tmp.data(), cap);
            for (uint8_t k : keys) b.insert(k);
            if (b.overflow) { if (cap == 512 || na <= 51) bad_rh++; continue; }      // cap 128 must hold up to 51 keys, cap 512 everything
            const int nb = b.order(ob);
            if (nb != na || !std::equal(oa, oa + na, ob)) bad_rh++;
            n_rh++;
        }
    }
    long bad_sort = 0, n_sort = 0;
    auto check = [&](const std::vector<std::pair<uint8_t, int>>& in) {
        std::vector<std::pair<uint8_t, int>> v = in;
        std::sort(v.begin(), v.end(), [](const std::pair<uint8_t, int>& a, const std::pair<uint8_t, int>& b) { return a.second > b.second; });
        std::vector<uint32_t> p;
        for (auto& e : in) p.push_back(((uint32_t)e.second << 8) | e.first);
        hs::CountSort::sort(p.data(), (int)p.size());
        for (size_t i = 0; i < v.size(); ++i) if ((p[i] & 255u) != v[i].first || (int)(p[i] >> 8) != v[i].second) { bad_sort++; break; }
        n_sort++;
    };
    for (long it = 0; it < trials; ++it) {
        const int n = 1 + rand() % 131;
        const int spread = 1 + rand() % ((it & 1) ? 3 : 60);      // few distinct counts: long runs of equal keys
        std::vector<std::pair<uint8_t, int>> v;
        for (int i = 0; i < n; ++i) v.push_back(std::make_pair((uint8_t)i, rand() % spread));
        if (it % 7 == 0) std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.second < b.second; });
        if (it % 11 == 0) std::reverse(v.begin(), v.end());
        check(v);
    }
    // adversarial inputs: run std::sort against the adversary, read the values it was forced to fix, use them as counts
    for (int n = 17; n <= 131; ++n) {
        Adversary adv(n);
        std::vector<int> idx((size_t)n);
        for (int i = 0; i < n; ++i) idx[(size_t)i] = i;
        std::sort(idx.begin(), idx.end(), [&](int x, int y) { return adv.less(y, x); });      // "greater" order, as the path sorts
        std::vector<std::pair<uint8_t, int>> v;
        for (int i = 0; i < n; ++i) v.push_back(std::make_pair((uint8_t)i, adv.val[(size_t)i]));
        // does this input reach the depth limit? count quicksort levels of the restatement by a dry run on a copy
        {
            std::vector<uint32_t> p;
            for (auto& e : v) p.push_back(((uint32_t)e.second << 8) | e.first);
            int lg = 0; for (int x = n; x > 1; x >>= 1) lg++;
            int first = 0, last = n, depth = 2 * lg;      // leftmost chain only: enough to see the limit being hit
            while (last - first > 16) { if (depth == 0) { g_heap_paths++; break; } --depth; last = hs::CountSort::partition_pivot(p.data(), first, last); }
        }
        check(v);
        for (auto& e : v) e.second /= 3;      // the same shape with ties
        check(v);
    }
    std::printf("rh8view %ld bad %ld; sort %ld bad %ld heap-paths %ld\n", n_rh, bad_rh, n_sort, bad_sort, g_heap_paths);
    return (bad_rh || bad_sort) ? 1 : 0;
}
