// hs_host_cv.cpp -- sequential glue of stage 3 (HS_call_variants) over dense per-read arrays.
//
// The device produces the pileup, the per-position code statistics, the columns of the few positions that can matter
// (second allele seen >= 4 times) with their leading codes in the reference's order, the candidate SNPs (the greedy spacing
// scan of call_variants.cpp:525-536), and at the end loops C / D and the merge of the SNP lists (:721-764, :1335-1352).
// What is left here is the reference's inherently sequential logic in between: the evolving set of partitions of
// keep_only_robust_variants (loops A and B, :577-708), whose decisions go through libm (lgamma / exp / log).
//
// Representation (round 4: everything in RANK space): the reads of a contig are ranked by start position; a partition is three
// bit sets over the ranks (present / state +1 / state -1) plus two counters per rank (more, less) instead of the reference's
// sorted sparse lists, a candidate column arrives from the device as one bit set per distinct code (hs::CandBits). Comparing a
// column with a partition is popcounts of ANDs over the one or two words both occupy; augmenting is word-wise set algebra plus a
// counter bump per read of the column; no per-entry walk, no per-read state array. Results are identical because every rule
// of Partition.cpp is per shared read; only loops whose floating-point accumulation order is observable (compute_conf) are
// run in ascending read order.
//
// Loop D of the reference (:745-764) evaluates every position of the contig against every final partition; a
// rescue needs n10+n00 > 4 (:756), i.e. some non-reference code carried by >= 5 reads, so only positions whose
// second count is >= 5 can ever be rescued -- exactly the columns the device already extracted.
#include "hs_host.h"
#include "hs_rh8.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace hs {

static constexpr int8_t ABSENT = 2;

// Storage of the partitions of a contig: every partition is eight bytes per read of the contig plus its bit sets, a few
// thousand partitions are made and dropped per contig group and step -- as individual std::vectors that was a seventh of
// the host's CPU time in malloc / free / memset. Partitions are bump-allocated from 1-MiB blocks instead; the blocks come
// from and go back to a process-wide cache when the contig's state dies.
namespace {
constexpr size_t kArenaBlock = 1u << 20;
struct BlockCache {
    std::mutex mu;
    std::vector<char*> free_blocks;
    // blocks kept for the next contig: at most HS_ARENA_CACHE_MB (default 512) of them; what is freed beyond that goes back to malloc
    static size_t keep_max() { static const size_t v = []() { const char* e = std::getenv("HS_ARENA_CACHE_MB"); const long m = e ? std::atol(e) : 512; return (size_t)(m > 0 ? m : 0); }(); return v; }
    static char* must(void* p, size_t n) {
        if (!p) { std::fprintf(stderr, "hairsplitter: out of memory (a partition block of %zu bytes)\n", n); std::abort(); }
        return (char*)p;
    }
    char* get() {
        { std::lock_guard<std::mutex> g(mu); if (!free_blocks.empty()) { char* p = free_blocks.back(); free_blocks.pop_back(); return p; } }
        return must(std::malloc(kArenaBlock), kArenaBlock);
    }
    void put(char* p) { std::lock_guard<std::mutex> g(mu); if (free_blocks.size() < keep_max()) free_blocks.push_back(p); else std::free(p); }
};
BlockCache& block_cache() { static BlockCache* c = new BlockCache(); return *c; }
}  // namespace
struct PartitionArena {
    std::vector<char*> blocks, big;
    size_t used = kArenaBlock;
    void* alloc(size_t n) {
        n = (n + 63) & ~(size_t)63;
        if (n > kArenaBlock) { char* p = BlockCache::must(std::malloc(n), n); big.push_back(p); return p; }
        if (used + n > kArenaBlock) { blocks.push_back(block_cache().get()); used = 0; }
        void* p = blocks.back() + used;
        used += n;
        return p;
    }
    ~PartitionArena() { for (char* b : blocks) block_cache().put(b); for (char* b : big) std::free(b); }
};


// A partition in rank space. State of rank k: absent (present bit clear), +1 (plus), -1 (minus), 0 (present only).
// more / less are only meaningful where the present bit is set (never cleared beforehand: a partition is 8 bytes per read of
// the contig and a few thousand are made per contig group and step).
struct RankPartition {
    int left = -1, right = -1;
    int n_occ = 0;                 // numberOfOccurences
    int n_corr = 0;                // number_of_correlating_snps
    int lo = 0, hi = -1;           // present reads lie in [lo, hi] (read indices)
    int n_reads = 0, words = 0;
    uint64_t* present = nullptr;   // [words] each
    uint64_t* plus = nullptr;
    uint64_t* minus = nullptr;
    int32_t* more = nullptr;       // [words * 64] by rank
    int32_t* less = nullptr;
    const int32_t* rank_of = nullptr;
    const int32_t* orig_of = nullptr;   // rank -> read
    int wlo = 0, whi = -1;         // words that hold present reads
    int survivor = -1;             // loop B with pair distances from the device: ordinal among the partitions that pass its gate,
    bool pristine = true;          // and whether this (final) partition still is what was uploaded (no merge into it yet)
    int reach = -1;                // largest (exclusive) end position of a present read: no read of the partition covers a position >= reach
    void allocate(PartitionArena& arena, int n) {
        n_reads = n; words = (n + 63) >> 6;
        const size_t counters = (size_t)words * 64 * 4, bits = (size_t)words * 8;
        char* p = (char*)arena.alloc(3 * bits + 2 * counters);
        std::memset(p, 0, 3 * bits);
        present = (uint64_t*)p; plus = present + words; minus = plus + words;
        more = (int32_t*)(p + 3 * bits); less = more + (size_t)words * 64;
    }
    void copy_from(const RankPartition& o) {      // same contig: same sizes (this partition's own storage is kept)
        uint64_t* pr = present; int32_t* mo = more; int32_t* le = less;
        *this = o;
        present = pr; plus = pr + words; minus = plus + words; more = mo; less = le;
        std::memcpy(present, o.present, (size_t)words * 24);
        if (o.whi >= o.wlo) {
            const size_t a = (size_t)o.wlo * 64, n = (size_t)(o.whi - o.wlo + 1) * 64;
            std::memcpy(more + a, o.more + a, n * 4); std::memcpy(less + a, o.less + a, n * 4);
        }
    }
    void touch_word(int w) { if (whi < wlo) { wlo = whi = w; } else { if (w < wlo) wlo = w; if (w > whi) whi = w; } }
    int state_at(int k) const {      // 2 = absent
        const uint64_t b = 1ull << (k & 63); const size_t w = (size_t)k >> 6;
        if (!(present[w] & b)) return ABSENT;
        return (plus[w] & b) ? 1 : ((minus[w] & b) ? -1 : 0);
    }
    void set_state(int k, int s) {   // s in {-1, 0, 1}: present with that state
        const uint64_t b = 1ull << (k & 63); const size_t w = (size_t)k >> 6;
        present[w] |= b;
        if (s == 1) { plus[w] |= b; minus[w] &= ~b; } else if (s == -1) { minus[w] |= b; plus[w] &= ~b; } else { plus[w] &= ~b; minus[w] &= ~b; }
        touch_word((int)w);
    }
};

struct Contingency {
    int n00 = 0, n01 = 0, n10 = 0, n11 = 0;
    bool comparable = false;       // numberOfBases != 0 (call_variants.cpp:817)
    uint8_t most = 0, second = ' ';
    int second_slot = -2;          // the column's slot of `second` when the comparison knows it (-1: the column does not hold it; -2: not known)
};

float mean_distance_from_counts(int64_t n_err, int64_t n_len) {
    // totalDistance is a float that is incremented by one (saturates at 2^24); totalLength a double starting at 1
    float total_distance = n_err > 16777216 ? 16777216.0f : (float)n_err;
    double total_length = 1.0 + (double)n_len;
    return (float)(total_distance / total_length);
}


// most frequent non-reference code among `codes` restricted to the entries flagged in `take`
// (first in robin_hood iteration order on ties: call_variants.cpp:837-844, Partition.cpp:59-66).
// `signed_ref_quirk`: in distance() the reference compares a *signed* char with unsigned keys (:838), so a
// reference code >= 128 never equals any key and stays eligible.
// decision half of second_most_frequent(): `seen` = distinct codes in first-appearance order with their counts
static uint8_t second_from_seen(const uint8_t* seen, const int* cnt, int nseen, uint8_t ref, bool signed_ref_quirk, bool insert_ref_last,
                                uint8_t dflt) {
    if (nseen == 0) return dflt;
    const bool ref_eligible = signed_ref_quirk && ref >= 128;
    int best = -1, nbest = 0;
    uint8_t bestk = dflt;
    bool ref_seen = false;
    for (int i = 0; i < nseen; ++i) {
        const uint8_t k = seen[i];
        if (k == ref) { ref_seen = true; if (!ref_eligible) continue; }
        if (cnt[i] > best) { best = cnt[i]; nbest = 1; bestk = k; }
        else if (cnt[i] == best) nbest++;
    }
    if (ref_eligible && !ref_seen && insert_ref_last) {   // content2[ref_base] inserts a zero-count key
        if (0 > best) { best = 0; nbest = 1; bestk = ref; } else if (best == 0) nbest++;
    }
    if (best < 0) return dflt;
    if (nbest == 1) return bestk;
    // tie: the winner is the first of the tied keys in the hash map's iteration order
    Rh8 rh; rh.clear();
    for (int i = 0; i < nseen; ++i) rh.insert(seen[i]);
    if (insert_ref_last) rh.insert(ref);
    uint8_t ord[260];
    const int m = rh.order(ord);
    for (int i = 0; i < m; ++i) {
        const uint8_t k = ord[i];
        if (k == ref && !ref_eligible) continue;
        int c = 0;
        for (int j = 0; j < nseen; ++j) if (seen[j] == k) { c = cnt[j]; break; }
        if (c == best) return k;
    }
    return bestk;
}

static uint8_t second_most_frequent(const uint8_t* code, int n, const uint8_t* take, uint8_t ref, bool signed_ref_quirk,
                                    bool insert_ref_last, uint8_t dflt) {
    // distinct codes in first-appearance order with their counts (a column carries a dozen codes at most: linear search)
    uint8_t seen[256];
    int cnt[256];
    int nseen = 0;
    for (int i = 0; i < n; ++i) {
        if (take && !take[i]) continue;
        const uint8_t c = code[i];
        int k = 0;
        while (k < nseen && seen[k] != c) ++k;
        if (k == nseen) { seen[nseen] = c; cnt[nseen] = 0; nseen++; }
        cnt[k]++;
    }
    return second_from_seen(seen, cnt, nseen, ref, signed_ref_quirk, insert_ref_last, dflt);
}


// A candidate column as loop A sees it: views into the device's block (hs::CandBits)
struct ColView {
    int wlo = 0, whi = -1, W = 0, nslots = 0, n_entries = 0;
    const uint64_t* any = nullptr;     // [W]
    const uint64_t* slots = nullptr;   // [nslots][W]
    const uint8_t* codes = nullptr;    // [nslots]
    int idx_min = 0, idx_max = -1, reach = -1;
    ColView(const CandBits& h, const uint64_t* words) {
        wlo = h.wlo; W = h.n_words; whi = h.wlo + (int)h.n_words - 1; nslots = h.n_slots; n_entries = h.n_entries;
        any = words + h.word_off; slots = any + W; codes = reinterpret_cast<const uint8_t*>(slots + (size_t)nslots * W);
        idx_min = h.idx_min; idx_max = h.idx_max; reach = h.reach;
    }
    int slot_of(uint8_t code) const { for (int k = 0; k < nslots; ++k) if (codes[k] == code) return k; return -1; }
    const uint64_t* slot(int k) const { return slots + (size_t)k * W; }
    // once per column, for the comparisons with every live partition: where the reference code lies, which other code the column holds
    // most of (k1_slot, -1 when two share that count) and how many reads the next one has (c2): among ANY subset of the column's reads
    // no code other than those two has more than c2
    int ref_slot = -1, k1_slot = -1, c2 = 0;
    void prepare(uint8_t ref) {
        ref_slot = slot_of(ref); k1_slot = -1; c2 = 0;
        int best = 0;
        for (int k = 0; k < nslots; ++k) {
            if (k == ref_slot) continue;
            int c = 0; const uint64_t* bk = slot(k);
            for (int j = 0; j < W; ++j) c += __builtin_popcountll(bk[j]);
            if (c > best) { c2 = best; best = c; k1_slot = k; }
            else { if (c == best) k1_slot = -1; if (c > c2) c2 = c; }
        }
        if (k1_slot < 0) c2 = best;
    }
};

void cv_rank_reads(int n_reads, const int32_t* read_start, std::vector<int32_t>& rank_of, std::vector<int32_t>& orig_of) {
    rank_of.assign((size_t)n_reads, 0); orig_of.assign((size_t)((n_reads + 63) / 64) * 64, 0);
    std::vector<int32_t> order((size_t)n_reads);
    for (int r = 0; r < n_reads; ++r) order[(size_t)r] = r;
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return read_start[a] != read_start[b] ? read_start[a] < read_start[b] : a < b; });
    for (int k = 0; k < n_reads; ++k) { rank_of[(size_t)order[(size_t)k]] = k; orig_of[(size_t)k] = order[(size_t)k]; }
}

// what k_cand_bits computes, restated on the host (see hs_host.h)
void cv_build_cand_bits(int n_cols, const int64_t* off, const int32_t* idx, const uint8_t* code, const int32_t* rank_of, const int32_t* read_end,
                        CandBits* bits, std::vector<uint64_t>& words) {
    for (int c = 0; c < n_cols; ++c) {
        const int32_t* ci = idx + off[c]; const uint8_t* cc = code + off[c];
        const int n = (int)(off[c + 1] - off[c]);
        CandBits& h = bits[c];
        h.n_entries = n; h.word_off = (int64_t)words.size();
        if (n == 0) { h.wlo = 0; h.n_words = 0; h.n_slots = 0; h.idx_min = 0; h.idx_max = -1; h.reach = -1; continue; }
        int lo = 0x7fffffff, hi = -1, reach = -1;
        for (int i = 0; i < n; ++i) { const int w = rank_of[ci[i]] >> 6; lo = std::min(lo, w); hi = std::max(hi, w); reach = std::max(reach, read_end[ci[i]]); }
        const int W = hi - lo + 1;
        uint8_t codes[256]; int slot_of[256]; int nslots = 0;
        std::fill(slot_of, slot_of + 256, -1);
        for (int i = 0; i < n; ++i) if (slot_of[cc[i]] < 0) { slot_of[cc[i]] = nslots; codes[nslots++] = cc[i]; }
        h.wlo = lo; h.n_words = (uint16_t)W; h.n_slots = (uint16_t)nslots; h.idx_min = ci[0]; h.idx_max = ci[n - 1]; h.reach = reach;
        const size_t base = words.size();
        words.resize(base + (size_t)cand_bits_block_words(W, nslots), 0ull);
        uint64_t* any = words.data() + base; uint64_t* sl = any + W;
        for (int i = 0; i < n; ++i) {
            const int k = rank_of[ci[i]];
            const uint64_t b = 1ull << (k & 63);
            any[(k >> 6) - lo] |= b;
            sl[(size_t)slot_of[cc[i]] * W + ((k >> 6) - lo)] |= b;
        }
        std::memcpy(reinterpret_cast<uint8_t*>(sl + (size_t)nslots * W), codes, (size_t)nslots);
    }
}

#ifdef HS_SELFCHECK   // entry-walk form, kept as the cross-check of the bit-set form in the test harness build
// distance(Partition&, Column&, char): call_variants.cpp:778-967
static Contingency column_vs_partition(const RankPartition& p, const int32_t* idx, const uint8_t* code, int n, uint8_t ref) {
    Contingency r;
    std::vector<uint8_t> take((size_t)std::max(n, 1));
    int shared = 0;
    for (int i = 0; i < n; ++i) { take[i] = p.state_at(p.rank_of[idx[i]]) != ABSENT; shared += take[i]; }
    if (shared == 0) return r;
    r.comparable = true;
    r.most = ref;
    r.second = second_most_frequent(code, n, take.data(), ref, true, true, ' ');
    for (int i = 0; i < n; ++i) {
        if (!take[i]) continue;
        const int s = p.state_at(p.rank_of[idx[i]]);
        if (code[i] == r.most) { if (s == 1) r.n11++; else if (s == -1) r.n01++; }
        else if (code[i] == r.second) { if (s == 1) r.n10++; else if (s == -1) r.n00++; }
    }
    return r;
}
#endif

// distance(Partition&, Column&, char) (call_variants.cpp:778-967) on bit sets: popcounts of ANDs over the words both the column
// and the partition occupy. `orig_of`: rank -> read index (the order in which the reference's hash map meets the codes is the
// order of the READ INDICES, needed when the best count is tied)
#ifdef HS_LOOPA_STATS
static std::atomic<long> g_la_stat[8];
#define LA_STAT(i) g_la_stat[i].fetch_add(1, std::memory_order_relaxed)
#else
#define LA_STAT(i)
#endif
static Contingency column_vs_partition_bits(const RankPartition& p, const ColView& cb, uint8_t ref, const int32_t* orig_of) {
    Contingency r;
    const int W = cb.W;
    const int w0 = std::max(cb.wlo, p.wlo), w1 = std::min(cb.whi, p.whi);
    LA_STAT(0);
    if (w0 > w1) { LA_STAT(1); return r; }
    const uint64_t* any = cb.any - cb.wlo;      // (indexed by absolute word below)
    int shared = 0;
    for (int w = w0; w <= w1; ++w) shared += __builtin_popcountll(any[w] & p.present[(size_t)w]);
    if (shared == 0) { LA_STAT(2); return r; }
    r.comparable = true;
    r.most = ref;
    // Few shared reads: the table holds at most `shared` reads, the column can only fit the partition with at least half of its
    // own reads in the table (:624-627) and only correlate with chi-square > 15, which a 2x2 table of N reads cannot exceed N
    // for (14 leaves room for the float rounding) -- neither can happen, the counts are of no consequence
    const bool no_skip = false;
    if (!no_skip && shared <= 14 && (size_t)shared < (size_t)cb.n_entries / 2) { LA_STAT(3); return r; }
    if (!no_skip) {   // the same with the shared reads the partition has an opinion on (state +1 / -1): only those enter the table
        int decided = 0;
        for (int w = w0; w <= w1; ++w) decided += __builtin_popcountll(any[w] & (p.plus[(size_t)w] | p.minus[(size_t)w]));
        if (decided <= 14 && (size_t)decided < (size_t)cb.n_entries / 2) { LA_STAT(4); return r; }
    }
    auto slot_abs = [&](int k) { return cb.slots + (size_t)k * W - cb.wlo; };
    if (ref < 128) {
        // the usual case in one pass: the counts of the column's codes among the shared reads, the largest among the codes other
        // than the reference code; a tie of that largest count (the reference then takes the first of the tied codes in the
        // iteration order of its hash map) goes through the general form below
        int best = -1, nbest = 0, best_slot = -1;
        const int ref_slot = cb.ref_slot;
        bool settled = false;
        if (cb.k1_slot >= 0) {      // the column's own second code has more shared reads than any other code can have: it is the one, no tie
            const uint64_t* bk = slot_abs(cb.k1_slot);
            int ck = 0, cr = 0;
            for (int w = w0; w <= w1; ++w) ck += __builtin_popcountll(bk[w] & p.present[(size_t)w]);
            if (ref_slot >= 0) { const uint64_t* br = slot_abs(ref_slot); for (int w = w0; w <= w1; ++w) cr += __builtin_popcountll(br[w] & p.present[(size_t)w]); }
            const int rest = shared - cr - ck;
            if (ck > (rest < cb.c2 ? rest : cb.c2)) { best = ck; nbest = 1; best_slot = cb.k1_slot; settled = true; }
            else if (ck == 0 && rest == 0) { nbest = 0; settled = true; }      // (only the reference code among the shared reads)
        }
        for (int k = 0; k < cb.nslots && !settled; ++k) {
            if (k == ref_slot) continue;
            const uint64_t* bk = slot_abs(k);
            int c = 0;
            for (int w = w0; w <= w1; ++w) c += __builtin_popcountll(bk[w] & p.present[(size_t)w]);
            if (c == 0) continue;
            if (c > best) { best = c; nbest = 1; best_slot = k; } else if (c == best) nbest++;
        }
        if (nbest <= 1) {
            LA_STAT(5);
            r.second = best_slot >= 0 ? cb.codes[best_slot] : (uint8_t)' ';
            r.second_slot = best_slot;
            if (ref_slot >= 0) {
                const uint64_t* bm = slot_abs(ref_slot);
                for (int w = w0; w <= w1; ++w) { r.n11 += __builtin_popcountll(bm[w] & p.plus[(size_t)w]); r.n01 += __builtin_popcountll(bm[w] & p.minus[(size_t)w]); }
            }
            if (best_slot >= 0) {
                const uint64_t* bs = slot_abs(best_slot);
                for (int w = w0; w <= w1; ++w) { r.n10 += __builtin_popcountll(bs[w] & p.plus[(size_t)w]); r.n00 += __builtin_popcountll(bs[w] & p.minus[(size_t)w]); }
            }
            return r;
        }
    }
    // counts among the shared reads
    uint8_t seen[128]; int cnt[128]; int slot[128];
    int nseen = 0;
    for (int k = 0; k < cb.nslots && k < 128; ++k) {
        const uint64_t* bk = slot_abs(k);
        int c = 0;
        for (int w = w0; w <= w1; ++w) c += __builtin_popcountll(bk[w] & p.present[(size_t)w]);
        if (c) { seen[nseen] = cb.codes[k]; cnt[nseen] = c; slot[nseen] = k; nseen++; }
    }
    // second_from_seen() only looks at the order of `seen` when the best count is tied (the hash map is then filled in the
    // order the codes first appear among the shared reads = lowest shared read index, column entries being ascending):
    // that order is worked out only in that case
    {
        const bool ref_eligible = ref >= 128;
        int best = -1, nbest = 0;
        bool ref_seen = false;
        for (int i = 0; i < nseen; ++i) {
            if (seen[i] == ref) { ref_seen = true; if (!ref_eligible) continue; }
            if (cnt[i] > best) { best = cnt[i]; nbest = 1; } else if (cnt[i] == best) nbest++;
        }
        if (ref_eligible && !ref_seen) { if (0 > best) { best = 0; nbest = 1; } else if (best == 0) nbest++; }
        if (nbest > 1) {
            int first[128];
            for (int i = 0; i < nseen; ++i) {
                const uint64_t* bk = slot_abs(slot[i]);
                first[i] = 0x7fffffff;
                for (int w = w0; w <= w1; ++w) {
                    uint64_t x = bk[w] & p.present[(size_t)w];
                    while (x) { const int o = orig_of[w * 64 + __builtin_ctzll(x)]; if (o < first[i]) first[i] = o; x &= x - 1; }
                }
            }
            for (int i = 1; i < nseen; ++i)   // insertion sort by first shared appearance (a handful of codes)
                for (int j = i; j > 0 && first[j] < first[j - 1]; --j) { std::swap(first[j], first[j - 1]); std::swap(seen[j], seen[j - 1]); std::swap(cnt[j], cnt[j - 1]); }
        }
    }
    r.second = second_from_seen(seen, cnt, nseen, ref, true, true, ' ');
    const int sm = cb.slot_of(r.most), ss = cb.slot_of(r.second);
    if (sm >= 0) { const uint64_t* bm = slot_abs(sm); for (int w = w0; w <= w1; ++w) { r.n11 += __builtin_popcountll(bm[w] & p.plus[(size_t)w]); r.n01 += __builtin_popcountll(bm[w] & p.minus[(size_t)w]); } }
    if (ss >= 0 && r.second != r.most) { const uint64_t* bs = slot_abs(ss); for (int w = w0; w <= w1; ++w) { r.n10 += __builtin_popcountll(bs[w] & p.plus[(size_t)w]); r.n00 += __builtin_popcountll(bs[w] & p.minus[(size_t)w]); } }
    return r;
}

// computeChiSquare: call_variants.cpp:1135-1163 (float marginals, double squares, float result)
static float chi_square(const Contingency& d) {
    const int n = d.n00 + d.n01 + d.n10 + d.n11;
    if (n == 0) return 0;
    const float pmax1 = float(d.n10 + d.n11) / n;
    const float pmax2 = float(d.n01 + d.n11) / n;
    if (pmax1 * (1 - pmax1) == 0 && pmax2 * (1 - pmax2) == 0) return -1;
    if (pmax1 * pmax2 * (1 - pmax1) * (1 - pmax2) == 0) return 0;
    const float e00 = (1 - pmax1) * (1 - pmax2) * n, e01 = (1 - pmax1) * pmax2 * n;
    const float e10 = pmax1 * (1 - pmax2) * n, e11 = pmax1 * pmax2 * n;
    const double d00 = (double)(float)(d.n00 - e00), d01 = (double)(float)(d.n01 - e01);
    const double d10 = (double)(float)(d.n10 - e10), d11 = (double)(float)(d.n11 - e11);
    return (float)(d00 * d00 / (double)e00 + d01 * d01 / (double)e01 + d10 * d10 / (double)e10 + d11 * d11 / (double)e11);
}


// chi_square(d) > 15 for a table whose margins are known to be neither empty nor full: N (ad - bc)^2 / (r1 r2 c1 c2) in double decides unless
// it comes within 0.05 of the threshold, then the reference's own sequence of float and double operations does
static inline bool chi_square_above_15(const Contingency& d) {
    const double n = (double)(d.n00 + d.n01 + d.n10 + d.n11);
    const double r1 = (double)(d.n10 + d.n11), c1 = (double)(d.n01 + d.n11);
    const double den = r1 * (n - r1) * c1 * (n - c1);
    if (den > 0) {
        const double det = (double)d.n11 * d.n00 - (double)d.n10 * d.n01;
        const double chi = n * det * det / den;
        if (chi > 15.05) return true;
        if (chi < 14.95) return false;
    }
    return chi_square(d) > 15;
}

// Partition::Partition(Column&, pos, ref_base): Partition.cpp:32-83
static void partition_from_column(RankPartition& p, PartitionArena& arena, int n_reads, const ColView& cb, int pos, uint8_t ref,
                                  const int32_t* rank_of, const int32_t* orig_of) {
    p.left = p.right = pos; p.n_occ = 1; p.n_corr = 0;
    p.rank_of = rank_of; p.orig_of = orig_of; p.wlo = 0; p.whi = -1; p.reach = cb.reach;
    p.allocate(arena, n_reads);
    // the most frequent code other than the reference code (first in the hash map's order on ties; the map meets the codes in
    // the order of the column's entries = the order of the slots)
    int cnt[128];
    const int ns = std::min(cb.nslots, 128);
    for (int k = 0; k < ns; ++k) { int c = 0; const uint64_t* bk = cb.slot(k); for (int j = 0; j < cb.W; ++j) c += __builtin_popcountll(bk[j]); cnt[k] = c; }
    const uint8_t second = second_from_seen(cb.codes, cnt, ns, ref, false, false, 0);
    const int sr = cb.ref_slot, ss = second != ref ? cb.slot_of(second) : -1;
    for (int j = 0; j < cb.W; ++j) {
        const size_t w = (size_t)(cb.wlo + j);
        const uint64_t a = cb.any[j];
        if (!a) continue;
        p.present[w] = a;
        p.plus[w] = sr >= 0 ? cb.slot(sr)[j] : 0ull;
        p.minus[w] = ss >= 0 ? cb.slot(ss)[j] : 0ull;
        p.touch_word((int)w);
        for (uint64_t x = a; x; x &= x - 1) { const size_t k = w * 64 + (size_t)__builtin_ctzll(x); p.more[k] = 1; p.less[k] = 0; }
    }
    p.lo = cb.n_entries ? cb.idx_min : 0; p.hi = cb.n_entries ? cb.idx_max : -1;
}

// Partition::augmentPartition with the 'A'/'a'/' ' recoding of distance() folded in:
// Partition.cpp:243-397 + call_variants.cpp:856-872
static void augment(RankPartition& p, const ColView& cb, const Contingency& d, int pos) {
    if (pos != -1) {
        if (pos < p.left || p.left == -1) p.left = pos;
        if (pos > p.right) p.right = pos;
    }
    if (!d.comparable || cb.n_entries == 0) return;      // empty partition_to_augment (:251-253)
    // recoded column: 'A' where the read carries the column's reference code, 'a' for its second code, ' ' otherwise
    static const uint64_t zero_words[4] = {0, 0, 0, 0};
    std::vector<uint64_t> zheap;
    const uint64_t* zeros = zero_words;
    if (cb.W > 4) { zheap.assign((size_t)cb.W, 0ull); zeros = zheap.data(); }
    const int sA = cb.ref_slot /* (d.most is the reference code) */, sa_ = d.second != d.most ? (d.second_slot != -2 ? d.second_slot : cb.slot_of(d.second)) : -1;
    const uint64_t* A = sA >= 0 ? cb.slot(sA) : zeros;
    const uint64_t* a = sa_ >= 0 ? cb.slot(sa_) : zeros;
    int nA = 0, na = 0;
    for (int j = 0; j < cb.W; ++j) { nA += __builtin_popcountll(A[j]); na += __builtin_popcountll(a[j]); }
    // two most frequent characters over 0..254 except ' ', lowest character wins ties (:261-280): 'A' < 'a'.
    // vA / va = the sign an 'A' / 'a' entry votes with: +1 if it is the most frequent character, -1 if the second, 0 if neither
    int vA, va;
    if (nA == 0 && na == 0) { vA = 0; va = 0; }                       // mostc = 0, secondc = 1: no entry matches either
    else if (nA >= na) { vA = 1; va = na > 0 ? -1 : 0; }
    else { va = 1; vA = nA > 0 ? -1 : 0; }
    int swapped = 0;                           // phase vote over shared reads (:284-314): sum of vote x partition state
    for (int j = 0; j < cb.W; ++j) {
        const size_t w = (size_t)(cb.wlo + j);
        const uint64_t pl = p.plus[w], mi = p.minus[w];
        swapped += vA * (__builtin_popcountll(A[j] & pl) - __builtin_popcountll(A[j] & mi)) + va * (__builtin_popcountll(a[j] & pl) - __builtin_popcountll(a[j] & mi));
    }
    if (swapped < 0) { vA = -vA; va = -va; }   // std::swap(mostc, secondc)
    for (int j = 0; j < cb.W; ++j) {           // word-wise form of the sorted merge (:322-390)
        const size_t w = (size_t)(cb.wlo + j);
        const uint64_t any = cb.any[j];
        if (!any) continue;
        const uint64_t s_plus = (vA == 1 ? A[j] : 0ull) | (va == 1 ? a[j] : 0ull), s_minus = (vA == -1 ? A[j] : 0ull) | (va == -1 ? a[j] : 0ull);
        const uint64_t voting = s_plus | s_minus;
        const uint64_t P = p.present[w], PL = p.plus[w], MI = p.minus[w];
        const uint64_t fresh = any & ~P;                               // not in the partition yet: takes the vote as it is
        const uint64_t undecided = any & P & ~PL & ~MI & voting;       // state 0 meets a vote: takes it, counters restart
        const uint64_t agree = any & ((PL & s_plus) | (MI & s_minus));
        const uint64_t against = any & ((PL & s_minus) | (MI & s_plus));
        uint64_t npl = PL | ((fresh | undecided) & s_plus), nmi = MI | ((fresh | undecided) & s_minus);
        int32_t* more = p.more + w * 64; int32_t* less = p.less + w * 64;
        for (uint64_t x = fresh; x; x &= x - 1) { const int b = __builtin_ctzll(x); more[b] = (voting >> b) & 1; less[b] = 0; }
        for (uint64_t x = undecided; x; x &= x - 1) { const int b = __builtin_ctzll(x); more[b] = 1; less[b] = 0; }
        for (uint64_t x = agree; x; x &= x - 1) more[__builtin_ctzll(x)] += 1;
        for (uint64_t x = against; x; x &= x - 1) {
            const int b = __builtin_ctzll(x);
            if (less[b] + 1 > more[b]) { more[b] += 1; const uint64_t bit = 1ull << b; if (PL & bit) { npl &= ~bit; nmi |= bit; } else { nmi &= ~bit; npl |= bit; } }
            else less[b] += 1;
        }
        p.present[w] = P | any; p.plus[w] = npl; p.minus[w] = nmi;
        if (fresh) p.touch_word((int)w);
    }
    if (cb.reach > p.reach) p.reach = cb.reach;      // (a read that was present already ends at or before the old reach)
    if (p.hi < p.lo) { p.lo = cb.idx_min; p.hi = cb.idx_max; } else { p.lo = std::min(p.lo, cb.idx_min); p.hi = std::max(p.hi, cb.idx_max); }
    p.n_occ += 1;
}

// 0.5 n + 3 sqrt(n 0.5 (1 - 0.5)) as the reference forms it (double arithmetic, then float): Partition.cpp:154, call_variants.cpp:
// 1033-1034. A function of the integer n alone: tabulated once for the small n that occur (the same expression, the same bits).
static inline float three_sigma_threshold(int n) {
    static const std::vector<float> table = [] {
        std::vector<float> t(4096);
        for (int k = 0; k < 4096; ++k) t[(size_t)k] = (float)(0.5 * k + 3 * std::sqrt(k * 0.5 * (1 - 0.5)));
        return t;
    }();
    return n >= 0 && n < 4096 ? table[(size_t)n] : (float)(0.5 * n + 3 * std::sqrt(n * 0.5 * (1 - 0.5)));
}


// Partition::isInformative(false, meanError): Partition.cpp:141-179 (counts: any order of the reads)
static bool is_informative(const RankPartition& p, float mean_error) {
    int suspicious[2] = {0, 0};
    int number_of_reads = 0;
    for (int w = p.wlo; w <= p.whi; ++w)
        for (uint64_t x = p.plus[(size_t)w] | p.minus[(size_t)w]; x; x &= x - 1) {
            const int b = __builtin_ctzll(x);
            const size_t k = (size_t)w * 64 + (size_t)b;
            const int read_number = p.more[k] + p.less[k];
            float threshold = three_sigma_threshold(read_number);
            threshold = std::min(threshold, float(read_number) - 1);
            if ((float)p.more[k] > threshold) { suspicious[(p.plus[(size_t)w] >> b) & 1]++; number_of_reads++; }
        }
    const float min_reads = mean_error * number_of_reads / 2;
    return !(suspicious[0] < min_reads || suspicious[1] < min_reads);
}

static double lchoose(double n, double k) { return std::lgamma(n + 1) - std::lgamma(k + 1) - std::lgamma(n - k + 1); }

// Partition::isSignificant: Partition.cpp:197-233 (the "p != 0" test is on the ordinal of the read, :209: the partition's
// first read -- lowest read index -- is not counted in `reads`)
static float significance(const RankPartition& p, int total_columns) {
    int mutated = 0, reads = 0, columns = 0;
    int first_read = 0x7fffffff; bool first_counts = false;
    for (int w = p.wlo; w <= p.whi; ++w)
        for (uint64_t x = p.present[(size_t)w]; x; x &= x - 1) {
            const int b = __builtin_ctzll(x);
            const size_t k = (size_t)w * 64 + (size_t)b;
            const bool solid = p.more[k] > 1 && p.less[k] == 0;
            if (solid && ((p.minus[(size_t)w] >> b) & 1)) { mutated++; if (p.more[k] > columns) columns = p.more[k]; }
            if (solid) reads++;
            const int o = p.orig_of[k];
            if (o < first_read) { first_read = o; first_counts = solid; }
        }
    if (first_counts) reads--;
    const double pv = std::exp(::log((double)(float(mutated) / reads)) * columns * mutated + lchoose(reads, mutated) + lchoose(total_columns, columns));
    return (float)std::max(0.0, pv);
}

// Partition::compute_conf: Partition.cpp:716-732 with getConfidence :811-827 (ascending read order matters)
static float confidence_score(const RankPartition& p) {
    double conf = 1;
    int n = 0;
    for (int r = p.lo; r <= p.hi; ++r) {
        const size_t k = (size_t)p.rank_of[r];
        const int s = p.state_at((int)k);
        if (s == ABSENT) continue;
        if (p.more[k] > 1) {
            float c;
            if (s == 0) c = 0.5f;
            else if (p.more[k] + p.less[k] > 0) c = float(p.more[k]) / (p.more[k] + p.less[k]);
            else c = 1;
            conf *= c; n++;
        }
    }
    if (conf == 1) conf = 0.99;
    const double x = 1 / (1 - std::exp(std::log(conf) / n));
    return (float)(x * x * p.n_occ);
}

struct PartPartDistance { int n00 = 0, n01 = 0, n10 = 0, n11 = 0; short phased = 1; bool augmented = true; };

// distance(Partition&, Partition&, 2): call_variants.cpp:977-1127
static PartPartDistance partition_vs_partition(const RankPartition& a, const RankPartition& b, int threshold_p) {
    int comparable = 0;
    int scores[2] = {0, 0};
    short ndiv[2] = {0, 0}, nunsure[2] = {0, 0};
    int m00[2] = {0, 0}, m01[2] = {0, 0}, m10[2] = {0, 0}, m11[2] = {0, 0};
    // every count below is a sum over the reads both partitions hold: those are the common bits of the two `present` sets (any
    // order), not a walk over all reads between the partitions' first and last
    for (int w = std::max(a.wlo, b.wlo); w <= std::min(a.whi, b.whi); ++w)
    for (uint64_t x = a.present[(size_t)w] & b.present[(size_t)w]; x; x &= x - 1) {
        const int bit = __builtin_ctzll(x);
        const size_t r = (size_t)w * 64 + (size_t)bit;
        if (!(a.more[r] > 1 && b.more[r] > 1)) continue;
        comparable++;
        const float t1 = three_sigma_threshold(a.more[r] + a.less[r]);
        const float t2 = three_sigma_threshold(b.more[r] + b.less[r]);
        const bool both = (float)a.more[r] > t1 && (float)b.more[r] > t2;
        const bool either = (float)a.more[r] > t1 || (float)b.more[r] > t2;
        const int s1 = (int)((a.plus[(size_t)w] >> bit) & 1) - (int)((a.minus[(size_t)w] >> bit) & 1);
        const int s2 = (int)((b.plus[(size_t)w] >> bit) & 1) - (int)((b.minus[(size_t)w] >> bit) & 1);
        if (s2 == 1) {
            if (s1 == 1) { scores[0]++; scores[1]--; m11[0]++; m10[1]++; if (both) ndiv[1]++; if (either) nunsure[1]++; }
            else if (s1 == -1) { scores[0]--; scores[1]++; m01[0]++; m00[1]++; if (both) ndiv[0]++; if (either) nunsure[0]++; }
        } else if (s2 == -1) {
            if (s1 == 1) { scores[0]--; scores[1]++; m10[0]++; m11[1]++; if (both) ndiv[0]++; if (either) nunsure[0]++; }
            else if (s1 == -1) { scores[0]++; scores[1]--; m00[0]++; m01[1]++; if (both) ndiv[1]++; if (either) nunsure[1]++; }
        }
    }
    PartPartDistance d;
    if ((ndiv[0] >= threshold_p && ndiv[1] >= threshold_p) || (nunsure[0] >= 5 && nunsure[1] >= 5) || comparable == 0) d.augmented = false;
    const int k = scores[1] > scores[0] ? 1 : 0;
    d.n00 = m00[k]; d.n01 = m01[k]; d.n10 = m10[k]; d.n11 = m11[k];
    d.phased = (short)(-2 * k + 1);
    return d;
}

// Partition::mergePartition(p, phased): Partition.cpp:401-537, per read of b (every rule is per read)
static void merge_partitions(RankPartition& a, const RankPartition& b, short phased) {
    a.left = std::min(a.left, b.left);
    a.right = std::max(a.right, b.right);
    for (int w = b.wlo; w <= b.whi; ++w)
    for (uint64_t x = b.present[(size_t)w]; x; x &= x - 1) {
        const int bit = __builtin_ctzll(x);
        const size_t r = (size_t)w * 64 + (size_t)bit;
        const int ob = (int)((b.plus[(size_t)w] >> bit) & 1) - (int)((b.minus[(size_t)w] >> bit) & 1);
        const int sa = a.state_at((int)r);
        if (sa == ABSENT || sa == 0) { a.set_state((int)r, ob * phased); a.more[r] = b.more[r]; a.less[r] = b.less[r]; }
        else if (ob == 0) { /* keep a */ }
        else if (phased * ob == sa) {
            int which = 0;
            const double c1 = double(a.more[r]) / (a.more[r] + a.less[r]);
            const double c2 = double(b.more[r]) / (b.more[r] + b.less[r]);
            if (c1 < 0.9 && c2 > 0.9 && b.more[r] >= 10) which = 1;
            else if (c2 < 0.9 && c1 > 0.9 && a.more[r] >= 10) which = 2;
            int nm = 0, nl = 0;
            if (which != 1) { nm += a.more[r]; nl += a.less[r]; }
            if (which != 2) { nm += b.more[r]; nl += b.less[r]; }
            a.more[r] = nm; a.less[r] = nl;
        } else {   // phased * ob == -sa
            int which = 0;
            const double c1 = double(a.more[r]) / (a.more[r] + a.less[r]);
            const double c2 = double(b.more[r]) / (b.more[r] + b.less[r]);
            if (c1 < 0.8 && c2 > 0.8 && b.more[r] >= 10) which = 1;
            else if (c2 < 0.8 && c1 > 0.8 && a.more[r] >= 10) which = 2;
            int nm = 0, nl = 0;
            if (which != 1) { nm += a.more[r]; nl += a.less[r]; }
            if (which != 2) { nm += b.less[r]; nl += b.more[r]; }
            if (nl > nm) { a.set_state((int)r, -sa); std::swap(nm, nl); }
            a.more[r] = nm; a.less[r] = nl;
        }
    }
    if (b.hi >= b.lo) { if (a.hi < a.lo) { a.lo = b.lo; a.hi = b.hi; } else { a.lo = std::min(a.lo, b.lo); a.hi = std::max(a.hi, b.hi); } }
    a.reach = std::max(a.reach, b.reach);
    a.n_occ += b.n_occ;
}

// Per-contig state of the stage-3 glue between its steps (loop A -> loop B -> export of the final partitions)
struct CvContigState {
    int n_reads = 0, n_candidates = 0;
    float mean_distance = 0;
    PartitionArena arena;                // storage of every partition below
    std::vector<RankPartition> parts;    // what loop A leaves (host loop or imported from the device)
    std::vector<int32_t> rank_of, orig_of;   // reads ranked by start position (ties by index): the bit order of the bit sets
    std::vector<RankPartition> finals;
    std::vector<int32_t> survivors;      // (cv_loop_b_survivors) indices in `parts` of the partitions that pass loop B's gate
};

CvContigState* cv_state_new() { return new CvContigState(); }
void cv_state_free(CvContigState* st) { delete st; }

void cv_phase_begin(CvContigState& st, int n_reads, int n_candidates, float mean_distance, ContigCvResult& out) {
    st.n_reads = n_reads; st.n_candidates = n_candidates; st.mean_distance = mean_distance;
    out.n_candidates = n_candidates;
}

// loop A (:590-638) on the host: sequential over the candidate columns of the contig
void cv_phase_a_host(CvContigState& st, const CandidateSet& cs, const int32_t* read_start, const int32_t* rank_pre, const int32_t* orig_pre) {
    const int n_reads = st.n_reads;
    static const bool tim = []() { const char* e = std::getenv("HS_TIMING"); return e && std::string(e) == "loop_a"; }();      // (HS_TIMING=loop_a: a line per contig)
    auto nowus = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a0 = tim ? nowus() : 0;
    long n_cmp = 0, n_aug = 0;
    double t_aug = 0;
    std::vector<RankPartition>& parts = st.parts;
    parts.clear();
    if (rank_pre && orig_pre) {      // (the batch ranked the reads when it was created: no sort per step)
        st.rank_of.assign(rank_pre, rank_pre + n_reads);
        st.orig_of.assign((size_t)((n_reads + 63) / 64) * 64, 0);
        std::copy(orig_pre, orig_pre + n_reads, st.orig_of.begin());
    } else cv_rank_reads(n_reads, read_start, st.rank_of, st.orig_of);
    const std::vector<int32_t>& rank_of = st.rank_of; const std::vector<int32_t>& orig_of = st.orig_of;
    int last_position = -5;
    // the partitions a column can still meet, in creation order: one that is more than 50 kb behind (:595) or none of whose reads
    // reaches the position stays so for every later column (positions ascend; `right` and `reach` only move when the partition
    // is augmented, which takes a comparison), so it leaves the list for good
    std::vector<int> active;
    for (int ci = 0; ci < cs.n; ++ci) {
        const int pos = cs.rec[ci].pos;
        const uint8_t k0 = cs.rec[ci].k0;
        if (pos - last_position <= 5) continue;
        ColView colbits(cs.bits[ci], cs.words);
        colbits.prepare(k0);
        const int n = colbits.n_entries;
        bool found = false;
        int n_corr = 0;
        size_t kept = 0;
        for (size_t a = 0; a < active.size(); ++a) {
            const size_t p = (size_t)active[a];
            if (found) { active[kept++] = (int)p; continue; }      // (behind the partition that took the column: not looked at, :630)
            if (std::abs(pos - parts[p].right) > 50000) continue;
            // no read of the partition reaches this position: nothing is shared, the comparison yields "not comparable", which
            // neither correlates nor matches (:817-828) -- skipped without looking at the bit sets
            if (pos >= parts[p].reach) continue;
            active[kept++] = (int)p;
            const Contingency d = column_vs_partition_bits(parts[p], colbits, k0, orig_of.data());
            n_cmp++;
#ifdef HS_SELFCHECK
            if (cs.idx) {
                const int32_t* idx = cs.idx + cs.off[ci]; const uint8_t* code = cs.code + cs.off[ci];
                if ((int)(cs.off[ci + 1] - cs.off[ci]) != n) { std::fprintf(stderr, "HS_SELFCHECK: bit sets of another column\n"); std::abort(); }
                const Contingency e = column_vs_partition(parts[p], idx, code, n, k0);
                if (e.n00 != d.n00 || e.n01 != d.n01 || e.n10 != d.n10 || e.n11 != d.n11 || e.comparable != d.comparable || e.second != d.second) {
                    // the bit-set form leaves the counts at zero where they cannot matter (few shared reads): the entry walk must
                    // then say "no correlation, no fit" as well
                    const int ec = e.n00 + e.n11 + e.n01 + e.n10;
                    const bool e_corr = e.n00 + e.n01 > 0.1 * ec && e.n00 + e.n01 < 0.9 * ec && e.n01 + e.n11 > 0.1 * ec && e.n01 + e.n11 < 0.9 * ec && chi_square(e) > 15;
                    const bool e_enough = (size_t)ec >= (size_t)n / 2;
                    const bool e_fit = (e.n01 <= std::max(0.1 * (e.n00 + e.n01), 1.0) && e.n10 < std::max(0.1 * (e.n11 + e.n10), 1.0) && e_enough)
                                       || (e.n00 <= std::max(0.1 * (e.n00 + e.n01), 1.0) && e.n11 < std::max(0.1 * (e.n11 + e.n10), 1.0) && e_enough);
                    const bool skipped = d.comparable && d.n00 + d.n01 + d.n10 + d.n11 == 0 && e.comparable;
                    if (!skipped || e_corr || e_fit) { std::fprintf(stderr, "HS_SELFCHECK: bit-set column_vs_partition differs\n"); std::abort(); }
                }
            }
#endif
            const int comparable = d.n00 + d.n11 + d.n01 + d.n10;
            if (d.n00 + d.n01 > 0.1 * comparable && d.n00 + d.n01 < 0.9 * comparable && d.n01 + d.n11 > 0.1 * comparable
                && d.n01 + d.n11 < 0.9 * comparable && chi_square_above_15(d)) {
                n_corr += 1; parts[p].n_corr += 1;
            }
            const bool enough = (size_t)comparable >= (size_t)n / 2;
            if ((d.n01 <= std::max(0.1 * (d.n00 + d.n01), 1.0) && d.n10 < std::max(0.1 * (d.n11 + d.n10), 1.0) && enough)
                || (d.n00 <= std::max(0.1 * (d.n00 + d.n01), 1.0) && d.n11 < std::max(0.1 * (d.n11 + d.n10), 1.0) && enough)) {
                found = true; n_aug++;
                const double ta0 = tim ? nowus() : 0;
#ifdef HS_SELFCHECK
                std::vector<int> before, want;
                if (cs.idx) {      // the entry-wise augmentPartition on a copy of the states, compared below
                    const int32_t* idx = cs.idx + cs.off[ci]; const uint8_t* code = cs.code + cs.off[ci];
                    int nA = 0, na = 0;
                    for (int i = 0; i < n; ++i) { const int isA = code[i] == d.most, isa = (code[i] == d.second) & !isA; nA += isA; na += isa; }
                    int vA, va;
                    if (nA == 0 && na == 0) { vA = 0; va = 0; } else if (nA >= na) { vA = 1; va = na > 0 ? -1 : 0; } else { va = 1; vA = nA > 0 ? -1 : 0; }
                    int swapped = 0;
                    for (int i = 0; i < n; ++i) {
                        const int s = parts[p].state_at(rank_of[(size_t)idx[i]]);
                        const int v = code[i] == d.most ? vA : (code[i] == d.second ? va : 0);
                        swapped += s == ABSENT ? 0 : v * s;
                    }
                    if (swapped < 0) { vA = -vA; va = -va; }
                    for (int i = 0; i < n; ++i) {      // (state, more, less) after the element-wise merge of the reference
                        const int k = rank_of[(size_t)idx[i]];
                        int st_ = parts[p].state_at(k), mo = st_ == ABSENT ? 0 : parts[p].more[k], le = st_ == ABSENT ? 0 : parts[p].less[k];
                        const int s = code[i] == d.most ? vA : (code[i] == d.second ? va : 0);
                        if (st_ == ABSENT) { st_ = s; mo = std::abs(s); le = 0; }
                        else if (s == 0) {}
                        else if (st_ == 0) { st_ = s; mo = 1; le = 0; }
                        else if (s == st_) mo += 1;
                        else { if (le + 1 > mo) { st_ = -st_; mo += 1; } else le += 1; }
                        want.push_back(st_); want.push_back(mo); want.push_back(le);
                    }
                }
#endif
                augment(parts[p], colbits, d, pos);
#ifdef HS_SELFCHECK
                if (cs.idx) {
                    const int32_t* idx = cs.idx + cs.off[ci];
                    for (int i = 0; i < n; ++i) {
                        const int k = rank_of[(size_t)idx[i]];
                        if (parts[p].state_at(k) != want[(size_t)3 * i] || parts[p].more[k] != want[(size_t)3 * i + 1] || parts[p].less[k] != want[(size_t)3 * i + 2]) {
                            std::fprintf(stderr, "HS_SELFCHECK: word-wise augmentPartition differs from the entry walk\n"); std::abort();
                        }
                    }
                }
#endif
                if (tim) t_aug += nowus() - ta0;
            }
        }
        active.resize(kept);
        if (!found) {
            active.push_back((int)parts.size());
            parts.emplace_back();
            partition_from_column(parts.back(), st.arena, n_reads, colbits, pos, k0, rank_of.data(), orig_of.data());
            parts.back().n_corr = n_corr;
        } else last_position = pos;
    }
#ifdef HS_LOOPA_STATS
    if (tim) std::fprintf(stderr, "[hs loopa stats] calls %ld no-common-words %ld shared0 %ld few-shared %ld few-decided %ld fast-table %ld\n", g_la_stat[0].load(), g_la_stat[1].load(), g_la_stat[2].load(), g_la_stat[3].load(), g_la_stat[4].load(), g_la_stat[5].load());
#endif
    if (tim) std::fprintf(stderr, "[hs timing] loop A: %d candidates, %zu partitions, %ld comparisons, %ld augmentations; %.0f us (augment %.0f)\n",
                          cs.n, parts.size(), n_cmp, n_aug, nowus() - t_a0, t_aug);
}

// loop A ran on the device (k_loop_a): its partitions become the host's. The device ranks the reads exactly as
// cv_rank_reads() does (the batch carries that order), so its bit sets and per-rank counters are taken as they are -- the
// words [w0, w1] of every partition, which hold all of its reads.
void cv_phase_a_import(CvContigState& st, const int32_t* read_start, int n_parts, const CvPartRecord* rec, const uint64_t* bits, const int32_t* cnt,
                       const int32_t* rank_pre, const int32_t* orig_pre) {
    const int N = st.n_reads;
    if (rank_pre && orig_pre) {
        st.rank_of.assign(rank_pre, rank_pre + N);
        st.orig_of.assign((size_t)((N + 63) / 64) * 64, 0);
        std::copy(orig_pre, orig_pre + N, st.orig_of.begin());
    } else cv_rank_reads(N, read_start, st.rank_of, st.orig_of);
    std::vector<RankPartition>& parts = st.parts;
    parts.clear();
    parts.resize((size_t)n_parts);
    for (int p = 0; p < n_parts; ++p) {
        RankPartition& d = parts[(size_t)p];
        const CvPartRecord& r = rec[p];
        d.left = r.left; d.right = r.right; d.n_occ = r.n_occ; d.n_corr = r.n_corr; d.lo = r.lo; d.hi = r.hi; d.reach = r.reach;
        d.rank_of = st.rank_of.data(); d.orig_of = st.orig_of.data(); d.wlo = 0; d.whi = -1;
        d.allocate(st.arena, N);
        const int span = r.w1 - r.w0 + 1;
        const uint64_t* pb = bits + 3 * r.word_off;
        const int32_t* pc = cnt + 64 * r.word_off;
        for (int j = 0; j < span; ++j) {
            const size_t w = (size_t)(r.w0 + j);
            const uint64_t pres = pb[j];
            if (!pres) continue;
            d.present[w] = pres; d.plus[w] = pb[span + j]; d.minus[w] = pb[2 * span + j];
            d.touch_word((int)w);
            for (uint64_t x = pres; x; x &= x - 1) {
                const int bit = __builtin_ctzll(x);
                const int32_t v = pc[j * 64 + bit];
                d.more[w * 64 + (size_t)bit] = v & 0xffff; d.less[w * 64 + (size_t)bit] = (v >> 16) & 0xffff;
            }
        }
    }
}

// loop B's gate (:650-653) for every partition of loop A: it depends on the partition alone, not on the finals
int cv_loop_b_survivors(CvContigState& st) {
    st.survivors.clear();
    for (size_t p1 = 0; p1 < st.parts.size(); ++p1) {
        const double p_value = significance(st.parts[p1], st.n_candidates);
        if ((p_value < 0.001 || st.parts[p1].n_corr > 1) && is_informative(st.parts[p1], st.mean_distance)) {
            st.parts[p1].survivor = (int)st.survivors.size();
            st.survivors.push_back((int32_t)p1);
        }
    }
    return (int)st.survivors.size();
}
// a partition as dense arrays over the READ INDICES (what the device kernels index): state 2 = absent
static void export_dense(const RankPartition& p, int8_t* state, int32_t* more, int32_t* less) {
    const size_t N = (size_t)p.n_reads;
    std::memset(state, ABSENT, N);
    if (more) { std::memset(more, 0, N * 4); std::memset(less, 0, N * 4); }
    for (int w = p.wlo; w <= p.whi; ++w)
        for (uint64_t x = p.present[(size_t)w]; x; x &= x - 1) {
            const int bit = __builtin_ctzll(x);
            const size_t k = (size_t)w * 64 + (size_t)bit;
            const size_t r = (size_t)p.orig_of[k];
            state[r] = (int8_t)((int)((p.plus[(size_t)w] >> bit) & 1) - (int)((p.minus[(size_t)w] >> bit) & 1));
            if (more) { more[r] = p.more[k]; less[r] = p.less[k]; }
        }
}
// the survivors' dense arrays (n_reads entries each, one after the other) for k_partition_pair_distance
void cv_export_survivors(const CvContigState& st, int8_t* state, int32_t* more, int32_t* less) {
    const size_t N = (size_t)st.n_reads;
    for (size_t k = 0; k < st.survivors.size(); ++k) export_dense(st.parts[(size_t)st.survivors[k]], state + k * N, more + k * N, less + k * N);
}
const std::vector<float>& cv_three_sigma_table() {
    static const std::vector<float> t = [] { std::vector<float> v(4096); for (int k = 0; k < 4096; ++k) v[(size_t)k] = three_sigma_threshold(k); return v; }();
    return t;
}

// loop B (:646-708). pair_table (optional): distance(survivor i, survivor j, 2) for i < j at 8 * (j (j - 1) / 2 + i), as
// k_partition_pair_distance leaves it: used while the final partition still is survivor i as uploaded (no merge into it yet)
void cv_phase_b(CvContigState& st, ContigCvResult& out, const int32_t* pair_table) {
    std::vector<RankPartition>& parts = st.parts;
    const float mean_distance = st.mean_distance;
    out.n_partitions = (int)parts.size();
    if (parts.empty()) return;
    std::vector<RankPartition>& finals = st.finals;
    RankPartition scratch;
    for (size_t p1 = 0; p1 < parts.size(); ++p1) {
        if (pair_table) { if (parts[p1].survivor < 0) continue; }
        else {
            const double p_value = significance(parts[p1], st.n_candidates);
            if (!((p_value < 0.001 || parts[p1].n_corr > 1) && is_informative(parts[p1], mean_distance))) continue;
        }
        bool different = true;
        for (size_t p2 = 0; p2 < finals.size(); ++p2) {
            PartPartDistance d;
            const int32_t* row = nullptr;
            if (pair_table && finals[p2].pristine && finals[p2].survivor >= 0) {
                const int64_t i = finals[p2].survivor, j = parts[p1].survivor;
                row = pair_table + 8 * (j * (j - 1) / 2 + i);
                if (!row[6]) row = nullptr;      // (a vote count beyond the threshold table: the host's own walk)
            }
            if (row) {
                if (row[7] == 0) continue;       // no comparable read: augmented = false
                d.n00 = row[0]; d.n01 = row[1]; d.n10 = row[2]; d.n11 = row[3]; d.phased = (short)row[4]; d.augmented = row[5] != 0;
            } else {
                {   // partitions without a common read are "not comparable" (comparable == 0 -> augmented = false, :1107-1111): one AND
                    // over the few words both occupy instead of a walk over all reads of the contig
                    const RankPartition& fa = finals[p2]; const RankPartition& fb = parts[p1];
                    bool any = false;
                    for (int w = std::max(fa.wlo, fb.wlo); w <= std::min(fa.whi, fb.whi) && !any; ++w) any = (fa.present[(size_t)w] & fb.present[(size_t)w]) != 0;
                    if (!any) continue;
                }
                d = partition_vs_partition(finals[p2], parts[p1], 2);
            }
            if (d.augmented && (d.n00 + d.n11 > 5 * (d.n01 + d.n10) || d.n10 + d.n01 > 5 * (d.n00 + d.n11))
                && d.n10 < std::max(2, 2 * d.n01) && d.n01 < std::max(2, 2 * d.n10)) {
                bool do_merge = d.n01 + d.n10 < 0.1 * (d.n00 + d.n11);
                if (!do_merge) {
                    if (!scratch.present) scratch.allocate(st.arena, st.n_reads);      // one trial copy per contig, reused
                    RankPartition& merged = scratch;
                    merged.copy_from(finals[p2]);
                    merge_partitions(merged, parts[p1], d.phased);
                    do_merge = confidence_score(merged) > confidence_score(finals[p2]);
                }
                if (do_merge) { merge_partitions(finals[p2], parts[p1], d.phased); finals[p2].pristine = false; different = false; break; }
            }
        }
        if (different) finals.push_back(parts[p1]);      // (a partition of loop A is looked at once: its storage goes along)
    }
    out.n_final_partitions = (int)finals.size();
    std::vector<RankPartition>().swap(parts);
}

// Loops C (:721-738) and D (:745-764) and the merge of the two SNP lists (:1335-1352) run on the device: the final partitions
// leave as dense state arrays over the read indices.
int cv_final_partitions(const CvContigState& st) { return (int)st.finals.size(); }
void cv_export_partitions(const CvContigState& st, int8_t* state, int64_t state_base, int64_t* state_off) {
    int64_t o = 0;
    for (size_t k = 0; k < st.finals.size(); ++k) {
        const RankPartition& p = st.finals[k];
        state_off[k] = state_base + o;
        export_dense(p, state + o, nullptr, nullptr);
        o += (int64_t)p.n_reads;
    }
}
}  // namespace hs
