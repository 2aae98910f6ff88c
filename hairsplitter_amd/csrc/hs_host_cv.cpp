// hs_host_cv.cpp -- sequential glue of stage 3 (HS_call_variants) over dense per-read arrays.
//
// The device produces the pileup, the per-position code statistics, the columns of the few positions that can matter
// (second allele seen >= 4 times) with their leading codes in the reference's order, the candidate SNPs (the greedy spacing
// scan of call_variants.cpp:525-536), and at the end loops C / D and the merge of the SNP lists (:721-764, :1335-1352).
// What is left here is the reference's inherently sequential logic in between: the evolving set of partitions of
// keep_only_robust_variants (loops A and B, :577-708), whose decisions go through libm (lgamma / exp / log).
//
// Representation: a partition is three dense arrays over the contig's N reads (state, more, less) instead of
// the reference's sorted sparse lists, so comparing a column with a partition costs O(column depth) instead of
// O(|partition| + depth), and augmenting is element-wise. Results are identical because every rule of
// Partition.cpp is per shared read; only loops whose floating-point accumulation order is observable
// (compute_conf) are run in ascending read order.
//
// Loop D of the reference (:745-764) evaluates every position of the contig against every final partition; a
// rescue needs n10+n00 > 4 (:756), i.e. some non-reference code carried by >= 5 reads, so only positions whose
// second count is >= 5 can ever be rescued -- exactly the columns the device already extracted.
#include "hs_host.h"
#include "hs_rh8.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace hs {

static constexpr int8_t ABSENT = 2;

// Storage of the partitions of a contig: every partition is nine bytes per read of the contig plus its bit sets, a few
// thousand partitions are made and dropped per contig group and step -- as individual std::vectors that was a seventh of
// the host's CPU time in malloc / free / memset. Partitions are bump-allocated from 1-MiB blocks instead; the blocks come
// from and go back to a process-wide cache when the contig's state dies.
namespace {
constexpr size_t kArenaBlock = 1u << 20;
struct BlockCache {
    std::mutex mu;
    std::vector<char*> free_blocks;
    char* get() {
        { std::lock_guard<std::mutex> g(mu); if (!free_blocks.empty()) { char* p = free_blocks.back(); free_blocks.pop_back(); return p; } }
        return (char*)std::malloc(kArenaBlock);
    }
    void put(char* p) { std::lock_guard<std::mutex> g(mu); if (free_blocks.size() < 4096) free_blocks.push_back(p); else std::free(p); }
};
BlockCache& block_cache() { static BlockCache* c = new BlockCache(); return *c; }
}  // namespace
struct PartitionArena {
    std::vector<char*> blocks, big;
    size_t used = kArenaBlock;
    void* alloc(size_t n) {
        n = (n + 63) & ~(size_t)63;
        if (n > kArenaBlock) { char* p = (char*)std::malloc(n); big.push_back(p); return p; }
        if (used + n > kArenaBlock) { blocks.push_back(block_cache().get()); used = 0; }
        void* p = blocks.back() + used;
        used += n;
        return p;
    }
    ~PartitionArena() { for (char* b : blocks) block_cache().put(b); for (char* b : big) std::free(b); }
};

struct DensePartition {
    int left = -1, right = -1;
    int n_occ = 0;                 // numberOfOccurences
    int n_corr = 0;                // number_of_correlating_snps
    int lo = 0, hi = -1;           // present reads lie in [lo, hi]
    int8_t* state = nullptr;       // [n_reads] ABSENT, or mostFrequentBases in {-1,0,1}
    int32_t* more = nullptr;       // [n_reads]
    int32_t* less = nullptr;
    int n_reads = 0, words = 0;
    // the same states as bit sets over the reads (loop A compares every candidate column with every live partition:
    // popcounts of ANDs instead of a walk over the column entries); maintained by partition_from_column() and augment().
    // Bit k = the read of RANK k in the order of the reads' start positions: the reads of a column (all of them cover its
    // position) and of a partition (they cover SNPs a few kb apart) then sit in a few neighbouring words, [wlo, whi], whatever
    // the order of the records in the SAM file; only those words are looked at.
    uint64_t* present = nullptr;   // [words] each
    uint64_t* plus = nullptr;
    uint64_t* minus = nullptr;
    const int32_t* rank_of = nullptr;
    const int32_t* orig_of = nullptr;   // rank -> read
    int wlo = 0, whi = -1;         // words that hold present reads
    int survivor = -1;             // loop B with pair distances from the device: ordinal among the partitions that pass its gate,
    bool pristine = true;          // and whether this (final) partition still is what was uploaded (no merge into it yet)
    int reach = -1;                // largest (exclusive) end position of a present read: no read of the partition covers a position >= reach
    // storage from the contig's arena: every read absent, counters and bit sets zero
    void allocate(PartitionArena& arena, int n) {
        n_reads = n; words = (n + 63) >> 6;
        const size_t counters = ((size_t)n * 4 + 63) & ~(size_t)63, states = ((size_t)n + 63) & ~(size_t)63, bits = (size_t)words * 8;
        char* p = (char*)arena.alloc(2 * counters + states + 3 * bits);
        std::memset(p, 0, 2 * counters + states + 3 * bits);
        more = (int32_t*)p; less = (int32_t*)(p + counters); state = (int8_t*)(p + 2 * counters);
        present = (uint64_t*)(p + 2 * counters + states); plus = present + words; minus = plus + words;
        std::memset(state, ABSENT, (size_t)n);
    }
    void copy_from(const DensePartition& o) {      // same contig: same sizes (this partition's own storage is kept)
        int8_t* s = state; int32_t* mo = more; int32_t* le = less; uint64_t* pr = present; uint64_t* pl = plus; uint64_t* mi = minus;
        *this = o;
        state = s; more = mo; less = le; present = pr; plus = pl; minus = mi;
        std::memcpy(state, o.state, (size_t)n_reads); std::memcpy(more, o.more, (size_t)n_reads * 4); std::memcpy(less, o.less, (size_t)n_reads * 4);
        std::memcpy(present, o.present, (size_t)words * 8); std::memcpy(plus, o.plus, (size_t)words * 8); std::memcpy(minus, o.minus, (size_t)words * 8);
    }
    void sync_bits(int r) {
        const int k = rank_of[r];
        const uint64_t b = 1ull << (k & 63);
        const size_t w = (size_t)k >> 6;
        const int8_t s = state[(size_t)r];
        if (s == 2) present[w] &= ~b; else { present[w] |= b; if (whi < wlo) { wlo = whi = (int)w; } else { if ((int)w < wlo) wlo = (int)w; if ((int)w > whi) whi = (int)w; } }
        if (s == 1) plus[w] |= b; else plus[w] &= ~b;
        if (s == -1) minus[w] |= b; else minus[w] &= ~b;
    }
};

struct Contingency {
    int n00 = 0, n01 = 0, n10 = 0, n11 = 0;
    bool comparable = false;       // numberOfBases != 0 (call_variants.cpp:817)
    uint8_t most = 0, second = ' ';
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

#ifdef HS_SELFCHECK   // entry-walk form, kept as the cross-check of the bit-set form in the test harness build
// distance(Partition&, Column&, char): call_variants.cpp:778-967
static Contingency column_vs_partition(const DensePartition& p, const int32_t* idx, const uint8_t* code, int n, uint8_t ref) {
    Contingency r;
    uint8_t take_stack[512];
    std::vector<uint8_t> take_heap;
    uint8_t* take = take_stack;
    if (n > 512) { take_heap.resize(n); take = take_heap.data(); }
    int shared = 0;
    for (int i = 0; i < n; ++i) { take[i] = p.state[idx[i]] != ABSENT; shared += take[i]; }
    if (shared == 0) return r;
    r.comparable = true;
    r.most = ref;
    r.second = second_most_frequent(code, n, take, ref, true, true, ' ');
    for (int i = 0; i < n; ++i) {
        if (!take[i]) continue;
        const int8_t s = p.state[idx[i]];
        if (code[i] == r.most) { if (s == 1) r.n11++; else if (s == -1) r.n01++; }
        else if (code[i] == r.second) { if (s == 1) r.n10++; else if (s == -1) r.n00++; }
    }
    return r;
}
#endif

// A candidate column as one bit set per distinct code (the reads that carry it), in first-appearance order.
struct ColumnBits {
    int words = 0, nslots = 0, n_entries = 0;
    uint8_t code_of[128];
    uint8_t slot_of[256];         // code -> slot, 0xFF = none yet; reset for the used codes at the next build
    std::vector<uint64_t> bits;   // [nslots][words]
    std::vector<uint64_t> any;    // [words]
    int wlo = 0, whi = -1;        // words that hold reads of the column (bit = rank of the read by start position)
    ColumnBits() { std::memset(slot_of, 0xFF, sizeof(slot_of)); }
    void build(const int32_t* idx, const uint8_t* code, int n, int n_reads, const int32_t* rank_of) {
        for (int k = 0; k < nslots; ++k) slot_of[code_of[k]] = 0xFF;
        words = (n_reads + 63) >> 6;
        nslots = 0; n_entries = n;
        // the reads of a column sit in a few neighbouring words (bit = rank by start position): only those words are kept valid
        int32_t rk_stack[512];
        std::vector<int32_t> rk_heap;
        int32_t* rk = rk_stack;
        if (n > 512) { rk_heap.resize((size_t)n); rk = rk_heap.data(); }
        int lo = words, hi = -1;
        for (int i = 0; i < n; ++i) { const int r = rank_of[idx[i]]; rk[i] = r; const int w = r >> 6; if (w < lo) lo = w; if (w > hi) hi = w; }
        wlo = lo; whi = hi;
        if (any.size() < (size_t)words) any.resize((size_t)words);
        if (bits.size() < (size_t)8 * words) bits.resize((size_t)8 * words);
        for (int w = lo; w <= hi; ++w) any[(size_t)w] = 0ull;
        for (int i = 0; i < n; ++i) {
            int k = slot_of[code[i]];
            if (k == 0xFF) {
                if (nslots == 128) continue;   // cannot happen: 125 pileup codes
                k = nslots;
                slot_of[code[i]] = (uint8_t)k;
                code_of[nslots++] = code[i];
                if (bits.size() < (size_t)nslots * words) bits.resize((size_t)nslots * 2 * words);
                for (int w = lo; w <= hi; ++w) bits[(size_t)k * words + (size_t)w] = 0ull;
            }
            const int r = rk[i];
            const uint64_t b = 1ull << (r & 63);
            bits[(size_t)k * words + ((size_t)r >> 6)] |= b;
            any[(size_t)r >> 6] |= b;
        }
    }
};

// column_vs_partition() on bit sets: same result, popcounts of ANDs instead of a walk over the column entries; only the
// words both the column and the partition occupy are visited. `orig_of`: rank -> read index (the order in which the
// reference's hash map meets the codes is the order of the READ INDICES, needed when the best count is tied)
static Contingency column_vs_partition_bits(const DensePartition& p, const ColumnBits& cb, uint8_t ref, const int32_t* orig_of) {
    Contingency r;
    const int W = cb.words;
    const int w0 = std::max(cb.wlo, p.wlo), w1 = std::min(cb.whi, p.whi);
    int shared = 0;
    for (int w = w0; w <= w1; ++w) shared += __builtin_popcountll(cb.any[(size_t)w] & p.present[(size_t)w]);
    if (shared == 0) return r;
    r.comparable = true;
    r.most = ref;
    // Few shared reads: the table holds at most `shared` reads, the column can only fit the partition with at least half of its
    // own reads in the table (:624-627) and only correlate with chi-square > 15, which a 2x2 table of N reads cannot exceed N
    // for (14 leaves room for the float rounding) -- neither can happen, the counts are of no consequence
    static const bool no_skip = std::getenv("HS_LOOP_A_NO_SKIP") != nullptr;      // (diagnostic: form every table)
    if (!no_skip && shared <= 14 && (size_t)shared < (size_t)cb.n_entries / 2) return r;
    if (!no_skip) {   // the same with the shared reads the partition has an opinion on (state +1 / -1): only those enter the table
        int decided = 0;
        for (int w = w0; w <= w1; ++w) decided += __builtin_popcountll(cb.any[(size_t)w] & (p.plus[(size_t)w] | p.minus[(size_t)w]));
        if (decided <= 14 && (size_t)decided < (size_t)cb.n_entries / 2) return r;
    }
    if (ref < 128) {
        // the usual case in one pass: the counts of the column's codes among the shared reads, the largest among the codes other
        // than the reference code; a tie of that largest count (the reference then takes the first of the tied codes in the
        // iteration order of its hash map) goes through the general form below
        int best = -1, nbest = 0, best_slot = -1;
        const int ref_slot = cb.slot_of[ref] == 0xFF ? -1 : (int)cb.slot_of[ref];
        for (int k = 0; k < cb.nslots; ++k) {
            if (k == ref_slot) continue;
            const uint64_t* bk = cb.bits.data() + (size_t)k * W;
            int c = 0;
            for (int w = w0; w <= w1; ++w) c += __builtin_popcountll(bk[w] & p.present[(size_t)w]);
            if (c == 0) continue;
            if (c > best) { best = c; nbest = 1; best_slot = k; } else if (c == best) nbest++;
        }
        if (nbest <= 1) {
            r.second = best_slot >= 0 ? cb.code_of[best_slot] : (uint8_t)' ';
            if (ref_slot >= 0) {
                const uint64_t* bm = cb.bits.data() + (size_t)ref_slot * W;
                for (int w = w0; w <= w1; ++w) { r.n11 += __builtin_popcountll(bm[w] & p.plus[(size_t)w]); r.n01 += __builtin_popcountll(bm[w] & p.minus[(size_t)w]); }
            }
            if (best_slot >= 0) {
                const uint64_t* bs = cb.bits.data() + (size_t)best_slot * W;
                for (int w = w0; w <= w1; ++w) { r.n10 += __builtin_popcountll(bs[w] & p.plus[(size_t)w]); r.n00 += __builtin_popcountll(bs[w] & p.minus[(size_t)w]); }
            }
            return r;
        }
    }
    // counts among the shared reads
    uint8_t seen[128]; int cnt[128]; int slot[128];
    int nseen = 0;
    for (int k = 0; k < cb.nslots; ++k) {
        const uint64_t* bk = cb.bits.data() + (size_t)k * W;
        int c = 0;
        for (int w = w0; w <= w1; ++w) c += __builtin_popcountll(bk[w] & p.present[(size_t)w]);
        if (c) { seen[nseen] = cb.code_of[k]; cnt[nseen] = c; slot[nseen] = k; nseen++; }
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
                const uint64_t* bk = cb.bits.data() + (size_t)slot[i] * W;
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
    const uint64_t* bm = nullptr; const uint64_t* bs = nullptr;
    for (int k = 0; k < cb.nslots; ++k) {
        if (cb.code_of[k] == r.most) bm = cb.bits.data() + (size_t)k * W;
        if (cb.code_of[k] == r.second) bs = cb.bits.data() + (size_t)k * W;
    }
    if (bm) for (int w = w0; w <= w1; ++w) { r.n11 += __builtin_popcountll(bm[w] & p.plus[(size_t)w]); r.n01 += __builtin_popcountll(bm[w] & p.minus[(size_t)w]); }
    if (bs && r.second != r.most) for (int w = w0; w <= w1; ++w) { r.n10 += __builtin_popcountll(bs[w] & p.plus[(size_t)w]); r.n00 += __builtin_popcountll(bs[w] & p.minus[(size_t)w]); }
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

// Partition::Partition(Column&, pos, ref_base): Partition.cpp:32-83
static void partition_from_column(DensePartition& p, PartitionArena& arena, int n_reads, const int32_t* idx, const uint8_t* code, int n, int pos, uint8_t ref,
                                  const int32_t* rank_of, const int32_t* orig_of, const int32_t* read_end) {
    p.left = p.right = pos; p.n_occ = 1; p.n_corr = 0;
    p.rank_of = rank_of; p.orig_of = orig_of; p.wlo = 0; p.whi = -1; p.reach = -1;
    for (int i = 0; i < n; ++i) p.reach = std::max(p.reach, read_end[idx[i]]);
    p.allocate(arena, n_reads);
    const uint8_t second = second_most_frequent(code, n, nullptr, ref, false, false, 0);
    for (int i = 0; i < n; ++i) {
        const int r = idx[i];
        p.state[r] = code[i] == ref ? 1 : (code[i] == second ? -1 : 0);
        p.more[r] = 1; p.less[r] = 0;
    }
    for (int i = 0; i < n; ++i) p.sync_bits(idx[i]);
    p.lo = n ? idx[0] : 0; p.hi = n ? idx[n - 1] : -1;
}

// Partition::augmentPartition with the 'A'/'a'/' ' recoding of distance() folded in:
// Partition.cpp:243-397 + call_variants.cpp:856-872
static void augment(DensePartition& p, const int32_t* idx, const uint8_t* code, int n, const Contingency& d, int pos, const int32_t* read_end) {
    if (pos != -1) {
        if (pos < p.left || p.left == -1) p.left = pos;
        if (pos > p.right) p.right = pos;
    }
    if (!d.comparable || n == 0) return;      // empty partition_to_augment (:251-253)
    // recoded column: +1 where the read carries the column's reference code ('A'), -1 for its second code ('a'), 0 otherwise
    int8_t cls_stack[512];
    std::vector<int8_t> cls_heap;
    int8_t* cls = cls_stack;
    if (n > 512) { cls_heap.resize((size_t)n); cls = cls_heap.data(); }
    int nA = 0, na = 0;
    for (int i = 0; i < n; ++i) {
        const int isA = code[i] == d.most, isa = (code[i] == d.second) & !isA;
        nA += isA; na += isa;
        cls[i] = (int8_t)(isA - isa);
    }
    // two most frequent characters over 0..254 except ' ', lowest character wins ties (:261-280): 'A' < 'a'.
    // vA / va = the sign an 'A' / 'a' entry votes with: +1 if it is the most frequent character, -1 if the second, 0 if neither
    int vA, va;
    if (nA == 0 && na == 0) { vA = 0; va = 0; }                       // mostc = 0, secondc = 1: no entry matches either
    else if (nA >= na) { vA = 1; va = na > 0 ? -1 : 0; }
    else { va = 1; vA = nA > 0 ? -1 : 0; }
    int swapped = 0;                           // phase vote over shared reads (:284-314): sum of vote x partition state
    for (int i = 0; i < n; ++i) {
        const int8_t s = p.state[idx[i]];
        const int v = cls[i] > 0 ? vA : (cls[i] < 0 ? va : 0);
        swapped += s == ABSENT ? 0 : v * s;
    }
    if (swapped < 0) { vA = -vA; va = -va; }   // std::swap(mostc, secondc)
    for (int i = 0; i < n; ++i) {              // element-wise form of the sorted merge (:322-390)
        const int r = idx[i];
        const int s = cls[i] > 0 ? vA : (cls[i] < 0 ? va : 0);
        int8_t& st = p.state[r];
        if (st == ABSENT) { st = (int8_t)s; p.more[r] = std::abs(s); p.less[r] = 0; if (read_end[r] > p.reach) p.reach = read_end[r]; }
        else if (s == 0) { continue; /* nothing new */ }
        else if (st == 0) { st = (int8_t)s; p.more[r] = 1; p.less[r] = 0; }
        else if (s == st) { p.more[r] += 1; continue; }
        else {                                 // s == -st
            if (p.less[r] + 1 > p.more[r]) { st = (int8_t)-st; p.more[r] += 1; }
            else { p.less[r] += 1; continue; }
        }
        p.sync_bits(r);                        // only when the state changed
    }
    if (n) { if (p.hi < p.lo) { p.lo = idx[0]; p.hi = idx[n - 1]; } else { p.lo = std::min(p.lo, idx[0]); p.hi = std::max(p.hi, idx[n - 1]); } }
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

// Partition::isInformative(false, meanError): Partition.cpp:141-179
static bool is_informative(const DensePartition& p, float mean_error) {
    int suspicious[2] = {0, 0};
    int number_of_reads = 0;
    for (int r = p.lo; r <= p.hi; ++r) {
        if (p.state[r] == ABSENT) continue;
        const int read_number = p.more[r] + p.less[r];
        float threshold = three_sigma_threshold(read_number);
        threshold = std::min(threshold, float(read_number) - 1);
        if ((float)p.more[r] > threshold) {
            if (p.state[r] == -1) { suspicious[0]++; number_of_reads++; }
            else if (p.state[r] == 1) { suspicious[1]++; number_of_reads++; }
        }
    }
    const float min_reads = mean_error * number_of_reads / 2;
    return !(suspicious[0] < min_reads || suspicious[1] < min_reads);
}

static double lchoose(double n, double k) { return std::lgamma(n + 1) - std::lgamma(k + 1) - std::lgamma(n - k + 1); }

// Partition::isSignificant: Partition.cpp:197-233 (the "p != 0" test is on the ordinal of the read, :209)
static float significance(const DensePartition& p, int total_columns) {
    int mutated = 0, reads = 0, columns = 0, ordinal = 0;
    for (int r = p.lo; r <= p.hi; ++r) {
        if (p.state[r] == ABSENT) continue;
        if (p.state[r] == -1 && p.more[r] > 1 && p.less[r] == 0) { mutated++; if (p.more[r] > columns) columns = p.more[r]; }
        if (ordinal != 0 && p.more[r] > 1 && p.less[r] == 0) reads++;
        ordinal++;
    }
    const double pv = std::exp(::log((double)(float(mutated) / reads)) * columns * mutated + lchoose(reads, mutated) + lchoose(total_columns, columns));
    return (float)std::max(0.0, pv);
}

// Partition::compute_conf: Partition.cpp:716-732 with getConfidence :811-827 (ascending read order matters)
static float confidence_score(const DensePartition& p) {
    double conf = 1;
    int n = 0;
    for (int r = p.lo; r <= p.hi; ++r) {
        if (p.state[r] == ABSENT) continue;
        if (p.more[r] > 1) {
            float c;
            if (p.state[r] == 0) c = 0.5f;
            else if (p.more[r] + p.less[r] > 0) c = float(p.more[r]) / (p.more[r] + p.less[r]);
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
static PartPartDistance partition_vs_partition(const DensePartition& a, const DensePartition& b, int threshold_p) {
    int comparable = 0;
    int scores[2] = {0, 0};
    short ndiv[2] = {0, 0}, nunsure[2] = {0, 0};
    int m00[2] = {0, 0}, m01[2] = {0, 0}, m10[2] = {0, 0}, m11[2] = {0, 0};
    // every count below is a sum over the reads both partitions hold: those are the common bits of the two `present` sets (any
    // order), not a walk over all reads between the partitions' first and last
    const int32_t* orig_of = a.orig_of;
    for (int w = std::max(a.wlo, b.wlo); w <= std::min(a.whi, b.whi); ++w)
    for (uint64_t x = a.present[(size_t)w] & b.present[(size_t)w]; x; x &= x - 1) {
        const int r = orig_of[w * 64 + __builtin_ctzll(x)];
        if (!(a.more[r] > 1 && b.more[r] > 1)) continue;
        comparable++;
        const float t1 = three_sigma_threshold(a.more[r] + a.less[r]);
        const float t2 = three_sigma_threshold(b.more[r] + b.less[r]);
        const bool both = (float)a.more[r] > t1 && (float)b.more[r] > t2;
        const bool either = (float)a.more[r] > t1 || (float)b.more[r] > t2;
        const int s1 = a.state[r], s2 = b.state[r];
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

// Partition::mergePartition(p, phased): Partition.cpp:401-537, element-wise
static void merge_partitions(DensePartition& a, const DensePartition& b, short phased) {
    a.left = std::min(a.left, b.left);
    a.right = std::max(a.right, b.right);
    for (int r = b.lo; r <= b.hi; ++r) {
        const int8_t ob = b.state[r];
        if (ob == ABSENT) continue;
        int8_t& sa = a.state[r];
        if (sa == ABSENT || sa == 0) { sa = (int8_t)(ob * phased); a.more[r] = b.more[r]; a.less[r] = b.less[r]; }
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
            if (nl > nm) { sa = (int8_t)-sa; std::swap(nm, nl); }
            a.more[r] = nm; a.less[r] = nl;
        }
    }
    if (b.hi >= b.lo) { if (a.hi < a.lo) { a.lo = b.lo; a.hi = b.hi; } else { a.lo = std::min(a.lo, b.lo); a.hi = std::max(a.hi, b.hi); } }
    for (int r = b.lo; r <= b.hi; ++r) if (b.state[r] != ABSENT) a.sync_bits(r);   // the bit sets follow (loop B asks them whether two partitions share a read)
    a.reach = std::max(a.reach, b.reach);
    a.n_occ += b.n_occ;
}

// Per-contig state of the stage-3 glue between its steps (loop A -> loop B -> export of the final partitions)
struct CvContigState {
    int n_reads = 0, n_candidates = 0;
    float mean_distance = 0;
    PartitionArena arena;                // storage of every partition below
    std::vector<DensePartition> parts;   // what loop A leaves (host loop or imported from the device)
    std::vector<int32_t> rank_of, orig_of;   // reads ranked by start position (ties by index): the bit order of the bit sets
    std::vector<DensePartition> finals;
    std::vector<int32_t> survivors;      // (cv_loop_b_survivors) indices in `parts` of the partitions that pass loop B's gate
};

CvContigState* cv_state_new() { return new CvContigState(); }
void cv_state_free(CvContigState* st) { delete st; }

void cv_phase_begin(CvContigState& st, int n_reads, int n_candidates, float mean_distance, ContigCvResult& out) {
    st.n_reads = n_reads; st.n_candidates = n_candidates; st.mean_distance = mean_distance;
    out.n_candidates = n_candidates;
}

static void rank_reads(CvContigState& st, const int32_t* read_start) {
    const int n_reads = st.n_reads;
    st.rank_of.assign((size_t)n_reads, 0); st.orig_of.assign((size_t)((n_reads + 63) / 64) * 64, 0);
    std::vector<int32_t> order((size_t)n_reads);
    for (int r = 0; r < n_reads; ++r) order[(size_t)r] = r;
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return read_start[a] != read_start[b] ? read_start[a] < read_start[b] : a < b; });
    for (int k = 0; k < n_reads; ++k) { st.rank_of[(size_t)order[(size_t)k]] = k; st.orig_of[(size_t)k] = order[(size_t)k]; }
}

// loop A (:590-638) on the host: sequential over the candidate columns of the contig
void cv_phase_a_host(CvContigState& st, const CandidateSet& cs, const int32_t* read_start, const int32_t* read_end) {
    const int n_reads = st.n_reads;
    auto col_idx = [&](int i) { return cs.idx + cs.off[i]; };
    auto col_code = [&](int i) { return cs.code + cs.off[i]; };
    auto col_n = [&](int i) { return (int)(cs.off[i + 1] - cs.off[i]); };
    const bool tim = std::getenv("HS_TIMING_AB") != nullptr;
    auto nowus = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a0 = tim ? nowus() : 0;
    long n_cmp = 0, n_aug = 0;
    double t_build = 0, t_aug = 0;
    std::vector<DensePartition>& parts = st.parts;
    parts.clear();
    ColumnBits colbits;
    rank_reads(st, read_start);
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
        const int32_t* idx = col_idx(ci); const uint8_t* code = col_code(ci); const int n = col_n(ci);
        bool found = false;
        int n_corr = 0;
        const double tb0 = tim ? nowus() : 0;
        if (!parts.empty()) colbits.build(idx, code, n, n_reads, rank_of.data());
        if (tim) t_build += nowus() - tb0;
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
            {
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
                && d.n01 + d.n11 < 0.9 * comparable && chi_square(d) > 15) {
                n_corr += 1; parts[p].n_corr += 1;
            }
            const bool enough = (size_t)comparable >= (size_t)n / 2;
            if ((d.n01 <= std::max(0.1 * (d.n00 + d.n01), 1.0) && d.n10 < std::max(0.1 * (d.n11 + d.n10), 1.0) && enough)
                || (d.n00 <= std::max(0.1 * (d.n00 + d.n01), 1.0) && d.n11 < std::max(0.1 * (d.n11 + d.n10), 1.0) && enough)) {
                found = true; n_aug++;
                const double ta0 = tim ? nowus() : 0;
                augment(parts[p], idx, code, n, d, pos, read_end);
                if (tim) t_aug += nowus() - ta0;
            }
        }
        active.resize(kept);
        if (!found) {
            active.push_back((int)parts.size());
            parts.emplace_back();
            partition_from_column(parts.back(), st.arena, n_reads, idx, code, n, pos, k0, rank_of.data(), orig_of.data(), read_end);
            parts.back().n_corr = n_corr;
        } else last_position = pos;
    }
    if (tim) std::fprintf(stderr, "[hs timing] loop A: %d candidates, %zu partitions, %ld comparisons, %ld augmentations; %.0f us (build %.0f, augment %.0f)\n",
                          cs.n, parts.size(), n_cmp, n_aug, nowus() - t_a0, t_build, t_aug);
}

// loop A ran on the device (k_loop_a): its partitions become the host's dense form. The device ranks the reads exactly as
// rank_reads() does (the batch carries that order), so its bit sets are taken as they are.
void cv_phase_a_import(CvContigState& st, const int32_t* read_start, int n_parts, const CvPartRecord* rec, const uint64_t* bits, const int32_t* cnt) {
    const int N = st.n_reads;
    const int W = (N + 63) >> 6;
    rank_reads(st, read_start);
    std::vector<DensePartition>& parts = st.parts;
    parts.clear();
    parts.resize((size_t)n_parts);
    for (int p = 0; p < n_parts; ++p) {
        DensePartition& d = parts[(size_t)p];
        const CvPartRecord& r = rec[p];
        d.left = r.left; d.right = r.right; d.n_occ = r.n_occ; d.n_corr = r.n_corr; d.lo = r.lo; d.hi = r.hi; d.reach = r.reach;
        d.rank_of = st.rank_of.data(); d.orig_of = st.orig_of.data(); d.wlo = W; d.whi = -1;
        d.allocate(st.arena, N);
        const uint64_t* pb = bits + (size_t)p * 3 * W;
        const int32_t* pc = cnt + (size_t)p * N;
        std::memcpy(d.present, pb, (size_t)W * 8); std::memcpy(d.plus, pb + W, (size_t)W * 8); std::memcpy(d.minus, pb + 2 * W, (size_t)W * 8);
        for (int w = 0; w < W; ++w) {
            if (!d.present[w]) continue;
            if (w < d.wlo) d.wlo = w;
            if (w > d.whi) d.whi = w;
            for (uint64_t x = d.present[w]; x; x &= x - 1) {
                const int k = w * 64 + __builtin_ctzll(x);
                const int q = st.orig_of[(size_t)k];
                const uint64_t b = 1ull << (k & 63);
                d.state[q] = (d.plus[w] & b) ? 1 : ((d.minus[w] & b) ? -1 : 0);
                d.more[q] = pc[k] & 0xffff; d.less[q] = (pc[k] >> 16) & 0xffff;      // (the device keeps the counters by rank)
            }
        }
        if (d.whi < d.wlo) d.wlo = 0;
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
// the survivors' dense arrays (n_reads entries each, one after the other) for k_partition_pair_distance
void cv_export_survivors(const CvContigState& st, int8_t* state, int32_t* more, int32_t* less) {
    const size_t N = (size_t)st.n_reads;
    for (size_t k = 0; k < st.survivors.size(); ++k) {
        const DensePartition& p = st.parts[(size_t)st.survivors[k]];
        std::memcpy(state + k * N, p.state, N); std::memcpy(more + k * N, p.more, N * 4); std::memcpy(less + k * N, p.less, N * 4);
    }
}
const std::vector<float>& cv_three_sigma_table() {
    static const std::vector<float> t = [] { std::vector<float> v(4096); for (int k = 0; k < 4096; ++k) v[(size_t)k] = three_sigma_threshold(k); return v; }();
    return t;
}

// loop B (:646-708). pair_table (optional): distance(survivor i, survivor j, 2) for i < j at 8 * (j (j - 1) / 2 + i), as
// k_partition_pair_distance leaves it: used while the final partition still is survivor i as uploaded (no merge into it yet)
void cv_phase_b(CvContigState& st, ContigCvResult& out, const int32_t* pair_table) {
    std::vector<DensePartition>& parts = st.parts;
    const float mean_distance = st.mean_distance;
    out.n_partitions = (int)parts.size();
    if (parts.empty()) return;
    std::vector<DensePartition>& finals = st.finals;
    DensePartition scratch;
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
                    const DensePartition& fa = finals[p2]; const DensePartition& fb = parts[p1];
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
                    if (!scratch.state) scratch.allocate(st.arena, st.n_reads);      // one trial copy per contig, reused
                    DensePartition& merged = scratch;
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
    std::vector<DensePartition>().swap(parts);
}

// Loops C (:721-738) and D (:745-764) and the merge of the two SNP lists (:1335-1352) run on the device: the final partitions
// leave as dense state arrays.
int cv_final_partitions(const CvContigState& st) { return (int)st.finals.size(); }
void cv_export_partitions(const CvContigState& st, int8_t* state, int64_t state_base, int64_t* state_off) {
    int64_t o = 0;
    for (size_t k = 0; k < st.finals.size(); ++k) {
        const DensePartition& p = st.finals[k];
        state_off[k] = state_base + o;
        std::memcpy(state + o, p.state, (size_t)p.n_reads);
        o += (int64_t)p.n_reads;
    }
}
}  // namespace hs
