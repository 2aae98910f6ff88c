// hs_kernels_loopa.hip -- loop A of keep_only_robust_variants on the device (call_variants.cpp:590-638): the candidate columns
// of a contig, in position order, meet the partitions found so far; a column that fits one augments it
// (Partition::augmentPartition, Partition.cpp:243-397), the others start a partition of their own (Partition.cpp:32-83). The
// chain over the columns of ONE contig is sequential by nature; contigs are independent: one wavefront per contig, and what
// counts is the LATENCY of a candidate (a lone wavefront issues one instruction every four to five cycles): nothing a candidate
// needs may sit behind a global-memory round trip.
//
// Same formulation as the host's (hs_host_cv.cpp): a partition is three bit sets over the contig's reads -- present / state +1 /
// state -1 -- with bit k = the read of rank k by start position, so that the reads of a column sit in at most four neighbouring
// 64-bit words; distance(Partition&, Column&) (call_variants.cpp:778-967) is popcounts of ANDs of those words with the bit sets
// of the column's codes.
//   * k_loop_a_prepare (one wavefront per candidate, all candidates at once): everything about a column that does not depend on
//     the partitions -- a 128-byte record (position, words, codes with their counts and their place in the iteration order of the
//     reference's hash map, the second allele a partition started by the column would get) + 64 words (any, 15 code slots x 4).
//     k_loop_a reads one int and one 64-bit word per lane and column, two columns ahead.
//   * comparison: LANES = live partitions (64 slots of bit sets in LDS, the slot's record in registers). Every lane forms the 2x2
//     table of its partition from its own (at most four) words against the column's words as scalars, chi-square and the two
//     verdicts (correlates / fits); the first fit in creation order wins, the partitions created before it count a correlation
//     -- the reference's `break`.
//   * augmenting the partition that took the column: LANES = the 64 READS of one word, everything about partition and column is
//     a scalar mask (s_and / s_bcnt; a lane tests its bit with the mask as the condition of a v_cndmask). What the decision
//     of a read needs of its two counters is more - less (>= 0 by construction; an opposing vote flips the state exactly when it
//     is 0): a byte per (slot, read) in LDS for the reads that can still meet a column -- a ring of eight words. The counters
//     themselves (more | less << 16 per read, what loop B reads) only ever receive additions: no-return atomics on a zeroed row
//     in global memory, nothing waits for them.
//   * Equal counts among the second alleles are broken by the reference in the iteration order of its hash map. With up to six
//     keys (8 buckets, no growth) that order is (home bucket, low five hash bits descending, insertion) -- tests/harness/
//     rh8_static_order.cpp --, a byte per code made by k_loop_a_prepare; two tied keys with the same byte, more than six keys,
//     and reference codes >= 128 (the reference's signed / unsigned comparison, :838) go through the emulator, one lane at a time.
// A contig that does not fit (more than 2048 reads, 64 live partitions, a column with more than 15 codes / 128 reads / 4 words,
// the partition pool, a counter beyond its width) is reported and done by the host (cv_phase_a_host).
// Included by hs_capi.hip after hs_kernels_cols.hip.
#pragma once

namespace hsdev {

#define HS_LA_SLOTS 64
#define HS_LA_MAXW 32          // words per bit set: contigs of up to 2048 reads
#define HS_LA_CODES 15         // distinct codes of a column
#define HS_LA_FAST_W 4         // words a column may spread over
#define HS_LA_RING 8           // words of reads whose (more - less) bytes are kept in LDS

struct LoopAPartition {        // == hs::CvPartRecord (what the host imports per partition)
    int32_t left, right, n_occ, n_corr, lo, hi, reach;
    int32_t w0, w1;            // the words its reads lie in (w1 < w0: none)
    int32_t pad;
    long long word_off;        // pool: first counter of the partition; packed: first word of its span in the range's packed arrays
};
static_assert(sizeof(LoopAPartition) == 48, "LoopAPartition layout");

// What k_loop_a reads of a column: 32 ints (lane l < 32 loads word l)
struct LoopAColumn {           // 128 bytes
    int32_t pos, wlo, n, idx_min, idx_max, reach;      // words 0..5: position, first word, entries, first / last read index, largest alignment end
    int16_t ww, nslots;                                // word 6: words (-1: not for the device), distinct codes
    int16_t ref_slot, new_second;                      // word 7: slot of the reference code, slot of the second allele of a partition this column starts (-1: none)
    int32_t ref;                                       // word 8: the reference code k0
    int32_t pad[7];
    uint32_t slot[16];                                 // words 16..31: code | order byte << 8 | count << 16 per slot
};
static_assert(sizeof(LoopAColumn) == 128, "LoopAColumn layout");

struct LoopAShared {           // scratch of the emulator path (one wavefront per workgroup)
    unsigned long long cb[HS_LA_CODES + 1][HS_LA_FAST_W];      // bit sets of the column's codes over the column's words
    int code_of[16];
    uint8_t rh_info[128], rh_key[128], rh_tmp[128];
    int x_seen[16], x_cnt[16], x_first[16];
};

// computeChiSquare(...) > 15 (call_variants.cpp:1135-1163). The reference's own sequence of float / double operations (chi_square_dev)
// decides only where a single-precision form of the same statistic, n (ad - bc)^2 / (row and column sums), comes within 2 of
// the threshold; its relative error is a few 1e-7 on tables of at most a few hundred reads, the margin is 13 %
static __device__ __forceinline__ bool chi_square_gt15(int n00, int n01, int n10, int n11) {
    const int r0 = n00 + n01, r1 = n10 + n11, c0 = n00 + n10, c1 = n01 + n11;
    if (r0 == 0 || r1 == 0 || c0 == 0 || c1 == 0) return false;      // (the reference returns 0 or -1 for a degenerate table)
    const float det = (float)(n00 * n11 - n01 * n10);
    const float est = (float)(r0 + r1) * det * det / ((float)r0 * (float)r1 * (float)c0 * (float)c1);
    if (est < 13.0f) return false;
    if (est > 17.0f) return true;
    Table2x2 t; t.n00 = n00; t.n01 = n01; t.n10 = n10; t.n11 = n11;
    return chi_square_dev(t) > 15;
}

// place of a key in the iteration order of an 8-bucket robin_hood map (up to six keys): home bucket << 5 | 31 - low five hash bits
static __device__ __forceinline__ int rh8_order_byte(int k) {
    unsigned long long h = (unsigned long long)(k & 255);
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
    return (int)((((h >> 5) & 7ull) << 5) | (31ull - (h & 31ull)));
}

// second_from_seen() of the host (hs_host_cv.cpp): the most frequent eligible code among `seen` (first-appearance order) with
// the reference's tie order; run by ONE lane (the tables in LDS belong to the wavefront)
template <class Sh>
static __device__ int second_from_seen_dev(Sh& S, int nseen, int ref, bool quirk, bool insert_ref_last, int dflt) {
    if (nseen == 0) return dflt;
    const bool ref_eligible = quirk && ref >= 128;
    int best = -1, nbest = 0, bestk = dflt;
    bool ref_seen = false;
    for (int i = 0; i < nseen; ++i) {
        const int k = S.x_seen[i];
        if (k == ref) { ref_seen = true; if (!ref_eligible) continue; }
        if (S.x_cnt[i] > best) { best = S.x_cnt[i]; nbest = 1; bestk = k; } else if (S.x_cnt[i] == best) nbest++;
    }
    if (ref_eligible && !ref_seen && insert_ref_last) { if (0 > best) { best = 0; nbest = 1; bestk = ref; } else if (best == 0) nbest++; }
    if (best < 0) return dflt;
    if (nbest == 1) return bestk;
    hs::Rh8View rh; rh.init(S.rh_info, S.rh_key, S.rh_tmp, 128);
    for (int i = 0; i < nseen; ++i) rh.insert((uint8_t)S.x_seen[i]);
    if (insert_ref_last) rh.insert((uint8_t)ref);
    const int m = rh.order(S.rh_tmp);
    for (int i = 0; i < m; ++i) {
        const int k = S.rh_tmp[i];
        if (k == ref && !ref_eligible) continue;
        int c = 0;
        for (int j = 0; j < nseen; ++j) if (S.x_seen[j] == k) { c = S.x_cnt[j]; break; }
        if (c == best) return k;
    }
    return bestk;
}

// One wavefront per candidate column of the contigs the device walks (on_dev[contig - c_first] != 0), the others are left alone.
struct LoopAPrepShared { uint8_t rh_info[128], rh_key[128], rh_tmp[128]; int x_seen[16], x_cnt[16], x_first[16]; unsigned long long cb[HS_LA_CODES + 1][HS_LA_FAST_W]; };
__global__ __launch_bounds__(256) void k_loop_a_prepare(
    const hs_colrec_dev* __restrict__ cand_rec, const int64_t* __restrict__ cand_off, const int32_t* __restrict__ cand_idx, const uint8_t* __restrict__ cand_code,
    const int32_t* __restrict__ cand_len /* non-NULL: cand_off[k] is column k's place in cand_idx / cand_code and cand_len[k] its length */,
    const ColumnsHeader* __restrict__ header, long long cap_cand, const int32_t* __restrict__ contig_rec_off, const int2* __restrict__ rank_end,
    int c_first, const uint8_t* __restrict__ on_dev, LoopAColumn* __restrict__ col_hdr, unsigned long long* __restrict__ col_words) {
    __shared__ LoopAPrepShared s_all[4];
    const int lane = lane_id();
    const int wv = wave_id();
    LoopAPrepShared& S = s_all[wv];
    long long n_cand = header->n_flagged;
    if (n_cand > cap_cand) n_cand = 0;      // (the packed block did not hold the candidates: the pass is run again)
    const long long k = (long long)blockIdx.x * 4 + wv;
    if (k >= n_cand) return;
    const hs_colrec_dev rec = cand_rec[k];
    if (!on_dev[rec.contig - c_first]) return;
    const int r0 = contig_rec_off[rec.contig];
    const int ref = (int)rec.k0;
    const int64_t e0 = cand_off[k];
    const int n = cand_len ? cand_len[k] : (int)(cand_off[k + 1] - e0);
    int rk[2], cd[2];      // rank and code of entries lane, 64 + lane (a column deeper than 128 reads goes to the host)
    int l_lo = 0x7fffffff, l_hi = -1, l_reach = -1, l_imin = 0x7fffffff, l_imax = -1;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int e = c * 64 + lane;
        rk[c] = -1; cd[c] = -1;
        if (e < n && e < 128) {
            const int ri = cand_idx[e0 + e];
            const int2 re = rank_end[r0 + ri];
            rk[c] = re.x; cd[c] = (int)cand_code[e0 + e];
            const int w = re.x >> 6;
            l_lo = w < l_lo ? w : l_lo; l_hi = w > l_hi ? w : l_hi; l_reach = re.y > l_reach ? re.y : l_reach;
            l_imin = ri < l_imin ? ri : l_imin; l_imax = ri > l_imax ? ri : l_imax;
        }
    }
    const int wlo = -wave_max_i32(-l_lo), whi = wave_max_i32(l_hi), reach = wave_max_i32(l_reach);
    const int imin = -wave_max_i32(-l_imin), imax = wave_max_i32(l_imax);
    int ww = n > 0 ? whi - wlo + 1 : 0;
    const bool narrow = n > 0 && n <= 128 && ww <= HS_LA_FAST_W;
    for (int x = lane; x < (HS_LA_CODES + 1) * HS_LA_FAST_W; x += 64) (&S.cb[0][0])[x] = 0ull;
    wave_lds_sync();
    // the distinct codes in first-appearance order (entries ascend by read index): lane q keeps slot q
    int slot_code = -1, slot_cnt = 0, nslots = 0, ref_slot = -1;
    bool many = false;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        unsigned long long rem = __ballot(cd[c] >= 0);
        while (rem) {
            const int X = __builtin_amdgcn_readlane(cd[c], __builtin_ctzll(rem));
            const unsigned long long mX = __ballot(cd[c] == X);
            rem &= ~mX;
            const unsigned long long hit = __ballot(slot_code == X);
            int ks;
            if (hit) ks = __builtin_ctzll(hit);
            else {
                if (nslots == HS_LA_CODES) { many = true; break; }
                ks = nslots++;
                if (lane == ks) { slot_code = X; slot_cnt = 0; }
                if (X == ref) ref_slot = ks;
            }
            if (lane == ks) slot_cnt += __popcll(mX);
            if (cd[c] == X && narrow) {
                const unsigned long long bit = 1ull << (rk[c] & 63);
                atomicOr(&S.cb[ks + 1][(rk[c] >> 6) - wlo], bit);
                atomicOr(&S.cb[0][(rk[c] >> 6) - wlo], bit);
            }
        }
    }
    wave_lds_sync();
    const bool usable = narrow && !many;
    // the second allele of a partition that this column starts (Partition.cpp:32-83): the most frequent code other than the
    // reference code over ALL entries, the first of equal ones in the hash map's order (keys inserted in slot order)
    int new_second = -1;
    if (usable) {
        const bool elig = lane < nslots && lane != ref_slot;
        const int best = wave_max_i32(elig ? slot_cnt : -1);
        const unsigned long long tied = __ballot(elig && slot_cnt == best);
        if (best >= 0) {
            if (__popcll(tied) == 1) new_second = __builtin_ctzll(tied);
            else {
                if (lane < nslots) { S.x_seen[lane] = slot_code; S.x_cnt[lane] = slot_cnt; }
                wave_lds_sync();
                if (lane == 0) {
                    const int sc = second_from_seen_dev(S, nslots, ref, false, false, 0);
                    int q = -1;
                    for (int i = 0; i < nslots; ++i) if (S.x_seen[i] == sc) q = i;
                    S.x_first[0] = q;
                }
                wave_lds_sync();
                new_second = S.x_first[0];
            }
        }
    }
    // the order of the slots as k_loop_a walks them: the reference code first, the others by descending count (its loop over the second
    // alleles ends as soon as no remaining code can reach the best count so far); the order of first appearance has done its duty above
    int perm = 0;      // lane q < nslots: the new index of slot q
    for (int j = 0; j < nslots; ++j) {
        const int cj = __builtin_amdgcn_readlane(slot_cnt, j);
        const bool before = j == ref_slot ? lane != j : (lane != ref_slot && (cj > slot_cnt || (cj == slot_cnt && j < lane)));
        perm += before ? 1 : 0;
    }
    if (lane < 16) S.x_first[lane] = lane < nslots ? perm : 0;
    wave_lds_sync();
    const int new_ref = ref_slot >= 0 ? S.x_first[ref_slot] : -1, new_sec = new_second >= 0 ? S.x_first[new_second] : -1;
    LoopAColumn* h = col_hdr + k;
    int* hw = reinterpret_cast<int*>(h);
    if (lane == 0) {
        h->pos = rec.pos; h->wlo = wlo; h->n = n; h->idx_min = n > 0 ? imin : 0; h->idx_max = n > 0 ? imax : -1; h->reach = reach;
        h->ww = (int16_t)(usable ? ww : -1); h->nslots = (int16_t)nslots; h->ref_slot = (int16_t)new_ref; h->new_second = (int16_t)new_sec; h->ref = ref;
    }
    if (lane >= 9 && lane < 16) hw[lane] = 0;
    if (lane < 16) S.x_cnt[lane] = 0;
    wave_lds_sync();
    if (lane < nslots) S.x_cnt[perm] = (int)((unsigned)slot_code | ((unsigned)rh8_order_byte(slot_code) << 8) | ((unsigned)slot_cnt << 16));
    wave_lds_sync();
    if (lane < 16) h->slot[lane] = (unsigned)S.x_cnt[lane];
    unsigned long long word = 0ull;
    if (usable) {
        if (lane < 4) word = S.cb[0][lane];
        else if ((lane >> 2) - 1 < nslots) {
            // the lane that holds word w of new slot r: find the old slot q with perm(q) = r
            const int r = (lane >> 2) - 1;
            int q = 0;
            for (int j = 0; j < nslots; ++j) if (S.x_first[j] == r) q = j;
            word = S.cb[q + 1][lane & 3];
        }
    }
    col_words[k * 64 + lane] = word;
}

// first candidate of every contig of the range (prefix sums of the per-contig counts the candidates' scan left in the info block)
__global__ __launch_bounds__(64) void k_loop_a_offsets(const int32_t* __restrict__ ctg_n, int c_count, int64_t* __restrict__ cand_off /* [C+1] */) {
    if (blockIdx.x != 0) return;
    const int lane = lane_id();
    long long base = 0;
    for (int c0 = 0; c0 < c_count; c0 += 64) {
        const int c = c0 + lane;
        const int v = c < c_count ? ctg_n[c] : 0;
        const int incl = wave_scan_incl(v);
        if (c < c_count) cand_off[c] = base + incl - v;
        base += __builtin_amdgcn_readlane(incl, 63);
    }
    if (lane == 0) cand_off[c_count] = base;
}

static __device__ __forceinline__ unsigned long long la_rl64(unsigned long long v, int l) {      // v_readlane of a 64-bit value (l wave-uniform)
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffull), l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

#ifdef HS_LA_DIAG      // cycles of the sections of k_loop_a, summed over the wavefronts (diag[0..7]) + candidates (diag[8]) + emulator lanes (diag[9])
#define HS_LA_T(i) do { const long long t__ = (long long)__builtin_amdgcn_s_memtime(); if (lane == 0) la_acc[i] += t__ - la_t; la_t = t__; } while (0)
#else
#define HS_LA_T(i) do { } while (0)
#endif

// What a lane (= a live partition) needs to decide "x > 0.1 c", "x < 0.9 c", "x <= max(0.1 s, 1.0)", "x < max(0.1 s, 1.0)" (:611-627, double
// arithmetic in the reference) with integers: floor / ceiling of the reference's own double products, tabulated once per wavefront
struct LoopATables { uint8_t floor01[256], ceil01[256], ceil09[256]; };

// One wavefront per contig of dev_list (heaviest first). LDS (static, 82 KB): the slot tables [3][32][64] u64 (present, plus, minus), the
// (more - less) bytes of the reads that can still meet a column, the tables above, the emulator's scratch.
__global__ __launch_bounds__(64) void k_loop_a(
    const LoopAColumn* __restrict__ col_hdr, const unsigned long long* __restrict__ col_words,
    const int64_t* __restrict__ cand_off /* [C+1] */, const int32_t* __restrict__ dev_list, int n_dev, int c_first, const int32_t* __restrict__ contig_rec_off,
    const int32_t* __restrict__ orig_of /* per record: rank -> read of its contig */,
    const int64_t* __restrict__ part_cap_off /* [C+1] partitions the pool holds per contig */, const int64_t* __restrict__ bits_off /* [C+1] words */,
    const int64_t* __restrict__ cnt_off /* [C+1] counters */, LoopAPartition* __restrict__ parts, unsigned long long* __restrict__ g_bits, int32_t* g_cnt,
    int32_t* __restrict__ n_parts /* [C] */, int32_t* __restrict__ n_span /* [C] words of all spans */, int32_t* __restrict__ failed /* [C] */,
    unsigned long long* __restrict__ diag) {
    __shared__ unsigned long long la_tab[3][HS_LA_MAXW][HS_LA_SLOTS];
    __shared__ uint8_t s_d[HS_LA_SLOTS][HS_LA_RING * 64];
    __shared__ LoopATables T;
    __shared__ LoopAShared S;
    if ((int)blockIdx.x >= n_dev) return;
    __builtin_amdgcn_s_setprio(3);      // a chain of dependent instructions beside other kernels' wavefronts on the same SIMD: first in line at issue
    const int ci = dev_list[blockIdx.x];
    const int lane = lane_id();
    const int r0 = contig_rec_off[c_first + ci];
    const int N = contig_rec_off[c_first + ci + 1] - r0;
    const int W = (N + 63) >> 6;
    const int RS = W * 64;             // counters per partition row
    const long long k0c = cand_off[ci], k1c = cand_off[ci + 1];
    const long long p_base = part_cap_off[ci];
    const int p_cap = (int)(part_cap_off[ci + 1] - p_base);
    unsigned long long* __restrict__ gb = g_bits + bits_off[ci];      // partition p: words [p * 3 W, (p + 1) * 3 W): present, plus, minus
    int32_t* gc = g_cnt + cnt_off[ci];                                 // partition p: counters [p * RS, (p + 1) * RS), indexed by RANK
    if (W > HS_LA_MAXW) { if (lane == 0) { failed[ci] = 1; n_parts[ci] = 0; n_span[ci] = 0; } return; }
    for (int x = lane; x < 3 * HS_LA_MAXW * HS_LA_SLOTS; x += 64) (&la_tab[0][0][0])[x] = 0ull;      // a free slot's words are zero
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 64 + lane;
        const double a = 0.1 * (double)c, z = 0.9 * (double)c;
        const int fa = (int)a, fz = (int)z;      // (non-negative: truncation is the floor)
        T.floor01[c] = (uint8_t)fa; T.ceil01[c] = (uint8_t)((double)fa < a ? fa + 1 : fa); T.ceil09[c] = (uint8_t)((double)fz < z ? fz + 1 : fz);
    }
    // lane = slot: the partition it holds (-1: free) and its record
    int s_pid = -1, s_right = 0, s_reach = 0, s_left = 0, s_nocc = 0, s_ncorr = 0, s_lo = 0, s_hi = -1, s_w0 = 0, s_w1 = -1;
    int P = 0, last_position = -5, span_total = 0, hi_water = 0;
    bool fail = false;
    wave_lds_sync();
#ifdef HS_LA_DIAG
    long long la_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long la_t = (long long)__builtin_amdgcn_s_memtime();
#endif
    // writes slot `s` (wave-uniform) to the pool and frees it; all lanes take part
    auto evict = [&](int s) {
        const int pid = __builtin_amdgcn_readlane(s_pid, s);
        const int w0 = __builtin_amdgcn_readlane(s_w0, s), w1 = __builtin_amdgcn_readlane(s_w1, s);
        const int span = w1 - w0 + 1;
        for (int x = lane; x < 3 * span; x += 64) {
            const int t = x / span, w = w0 + x % span;
            gb[(long long)pid * 3 * W + t * W + w] = la_tab[t][w][s];
            la_tab[t][w][s] = 0ull;
        }
        if (lane == s) {
            LoopAPartition r;
            r.left = s_left; r.right = s_right; r.n_occ = s_nocc; r.n_corr = s_ncorr; r.lo = s_lo; r.hi = s_hi; r.reach = s_reach; r.w0 = w0; r.w1 = w1; r.pad = 0;
            r.word_off = cnt_off[ci] + (long long)pid * RS;
            parts[p_base + pid] = r;
            s_pid = -1;
        }
        span_total += span;
    };
    // the column in flight and the four behind it (vector loads through per-lane pointers that move on by one column)
    const int* hp = reinterpret_cast<const int*>(col_hdr) + k0c * 32 + (lane & 31);
    const unsigned long long* wp = col_words + k0c * 64 + lane;
    struct Col { int hw; unsigned long long dw; };
    Col n1, n2, n3, n4;
    n1.hw = 0; n1.dw = 0ull; n2 = n1; n3 = n1; n4 = n1;
    const long long n_cand = k1c - k0c;
    if (n_cand > 0) { n1.hw = hp[0]; n1.dw = wp[0]; }
    if (n_cand > 1) { n2.hw = hp[32]; n2.dw = wp[64]; }
    if (n_cand > 2) { n3.hw = hp[64]; n3.dw = wp[128]; }
    if (n_cand > 3) { n4.hw = hp[96]; n4.dw = wp[192]; }
    hp += 128; wp += 256;
    Col cur;
    int pos = 0, wlo = 0, n = 0, nslots = 0, ref_slot = 0, new_second = 0, ref = 0;

    // ---- one candidate column whose reads lie in WW words ----
    auto step = [&](auto wwc) -> bool {      // false: the contig is the host's after all
        constexpr int WW = decltype(wwc)::value;
        const unsigned long long dw = cur.dw;      // word w of `any` in lane w, of code slot q in lane 4 (q + 1) + w
        // ---- the live partitions against the column: lanes = slots ----
        const bool used = s_pid >= 0;
        const int dist = pos - s_right;
        const bool elig = used & ((dist < 0 ? -dist : dist) <= 50000) & (pos < s_reach);
        const bool dead = used & !elig;      // (final: positions ascend, right / reach only move when the partition is augmented)
        unsigned long long pr[WW], pl[WW], mi[WW], an[WW];
        const unsigned long long* tb = &la_tab[0][wlo][lane];
#pragma unroll
        for (int w = 0; w < WW; ++w) {
            pr[w] = tb[w * HS_LA_SLOTS]; pl[w] = tb[(HS_LA_MAXW + w) * HS_LA_SLOTS]; mi[w] = tb[(2 * HS_LA_MAXW + w) * HS_LA_SLOTS];
            an[w] = la_rl64(dw, w);
        }
        int shared = 0, decided = 0;
#pragma unroll
        for (int w = 0; w < WW; ++w) { shared += __popcll(an[w] & pr[w]); decided += __popcll(an[w] & (pl[w] | mi[w])); }
        // few shared (or decided) reads: the table can neither fit nor correlate, its counts are of no consequence (hs_host_cv.cpp)
        const unsigned half_n = (unsigned)n / 2u;
        const bool skip = !elig | (shared == 0) | ((shared <= 14) & ((unsigned)shared < half_n)) | ((decided <= 14) & ((unsigned)decided < half_n));
        int n00 = 0, n01 = 0, n10 = 0, n11 = 0, best_slot = -1;
        bool need_exact = false, corr = false, fit = false;
        const unsigned long long livem = __ballot(!skip);
        if (livem != 0ull) {
            // per code slot other than the reference code (slot 0 when the column has it; the others by descending count): its count among
            // the shared reads; the largest wins, of equal ones the first in the hash map's order -- key = count << 8 | 255 - order byte
            int best_key = 0, second_key = 0, nkeys = 1;      // (the reference code is a key of the map whether a shared read carries it or not)
            int q = ref_slot == 0 ? 1 : 0;
            for (; q < nslots; ++q) {
                const int info = __builtin_amdgcn_readlane(cur.hw, 16 + q);
                int c = 0;
#pragma unroll
                for (int w = 0; w < WW; ++w) c += __popcll(la_rl64(dw, 4 * (q + 1) + w) & pr[w]);
                const int key = c ? ((c << 8) | (255 - ((info >> 8) & 255))) : 0;
                nkeys += c ? 1 : 0;
                const int lo_k = key < best_key ? key : best_key;
                second_key = lo_k > second_key ? lo_k : second_key;
                best_slot = key > best_key ? q : best_slot;
                best_key = key > best_key ? key : best_key;
                // no code behind this one has more reads in the whole column than the best count so far: nothing can reach it
                const int next_total = q + 1 < nslots ? (int)((unsigned)__builtin_amdgcn_readlane(cur.hw, 16 + (q + 1 < 16 ? q + 1 : 15)) >> 16) : 0;
                if (__ballot(!skip & ((best_key >> 8) <= next_total)) == 0ull) break;
            }
            nkeys += q < nslots ? nslots - 1 - q : 0;      // (the loop ended early: the codes it did not look at count as keys, on the safe side)
            const bool tie = ((second_key >> 8) == (best_key >> 8)) & (best_key != 0);
            const bool amb = second_key == best_key;
            // (a lane that left the loop early has a best count no other code can reach: no tie, whatever codes were not looked at)
            need_exact = !skip & ((ref >= 128) | (tie & (amb | (nkeys > 6))));
            best_slot = skip ? -1 : best_slot;
            const bool want = !skip & !need_exact;
            if (ref_slot == 0) {
#pragma unroll
                for (int w = 0; w < WW; ++w) { const unsigned long long x = la_rl64(dw, 4 + w); n11 += __popcll(x & pl[w]); n01 += __popcll(x & mi[w]); }
            }
            n11 = want ? n11 : 0; n01 = want ? n01 : 0;
            // the words of the lane's own second allele: a gather across the lanes of `dw`
            const bool want2 = want & (best_slot >= 0);
            const int src = want2 ? 4 * (best_slot + 1) : 0;
#pragma unroll
            for (int w = 0; w < WW; ++w) {
                const unsigned lo = (unsigned)__shfl((int)(unsigned)(dw & 0xffffffffull), src + w, 64), hi = (unsigned)__shfl((int)(unsigned)(dw >> 32), src + w, 64);
                const unsigned long long x = want2 ? (((unsigned long long)hi << 32) | lo) : 0ull;
                n10 += __popcll(x & pl[w]); n00 += __popcll(x & mi[w]);
            }
        }
        HS_LA_T(1);
        // the lanes whose second allele needs the emulator, one at a time (column_vs_partition_bits, general form)
        unsigned long long X = __ballot(need_exact);
        if (X) {
            const int32_t* __restrict__ og_of = orig_of + r0;
            (&S.cb[0][0])[lane] = dw;
            { const int v = __shfl(cur.hw, 16 + (lane & 15), 64); if (lane < 16) S.code_of[lane] = v & 255; }
            wave_lds_sync();
#ifdef HS_LA_DIAG
            if (lane == 0) la_acc[9] += __popcll(X);
#endif
            while (X) {
                const int l = __builtin_ctzll(X);
                X &= X - 1ull;
                if (lane == l) {
                    int nseen = 0;
                    for (int q = 0; q < nslots; ++q) {
                        int c = 0;
                        for (int w = 0; w < WW; ++w) c += __popcll(S.cb[q + 1][w] & la_tab[0][wlo + w][lane]);
                        if (c) { S.x_seen[nseen] = S.code_of[q]; S.x_cnt[nseen] = c; S.x_first[nseen] = q; nseen++; }      // (x_first: the slot for now)
                    }
                    // the hash map meets the codes in the order of the READ INDICES of the shared reads that carry them
                    for (int i = 0; i < nseen; ++i) {
                        const int q = S.x_first[i];
                        int f = 0x7fffffff;
                        for (int w = 0; w < WW; ++w) {
                            unsigned long long x = S.cb[q + 1][w] & la_tab[0][wlo + w][lane];
                            while (x) { const int o = og_of[(wlo + w) * 64 + __builtin_ctzll(x)]; if (o < f) f = o; x &= x - 1ull; }
                        }
                        S.x_first[i] = f;
                    }
                    for (int i = 1; i < nseen; ++i)
                        for (int j = i; j > 0 && S.x_first[j] < S.x_first[j - 1]; --j) {
                            int t;
                            t = S.x_first[j]; S.x_first[j] = S.x_first[j - 1]; S.x_first[j - 1] = t;
                            t = S.x_seen[j]; S.x_seen[j] = S.x_seen[j - 1]; S.x_seen[j - 1] = t;
                            t = S.x_cnt[j]; S.x_cnt[j] = S.x_cnt[j - 1]; S.x_cnt[j - 1] = t;
                        }
                    const int second = second_from_seen_dev(S, nseen, ref, true, true, ' ');
                    int sm = -1, ss = -1;
                    for (int q = 0; q < nslots; ++q) { if (S.code_of[q] == ref) sm = q; if (S.code_of[q] == second) ss = q; }
                    n11 = 0; n01 = 0; n10 = 0; n00 = 0;
                    if (sm >= 0) for (int w = 0; w < WW; ++w) { n11 += __popcll(S.cb[sm + 1][w] & la_tab[1][wlo + w][lane]); n01 += __popcll(S.cb[sm + 1][w] & la_tab[2][wlo + w][lane]); }
                    if (ss >= 0 && second != ref) for (int w = 0; w < WW; ++w) { n10 += __popcll(S.cb[ss + 1][w] & la_tab[1][wlo + w][lane]); n00 += __popcll(S.cb[ss + 1][w] & la_tab[2][wlo + w][lane]); }
                    best_slot = second != ref ? ss : -1;
                }
                wave_lds_sync();
            }
        }
        HS_LA_T(2);
        // verdicts (:611-627), per lane (a lane without a table has none: `enough` needs half of the column's reads in it)
        if (livem != 0ull || n <= 1) {
            const int s0 = n00 + n01, s1 = n11 + n10, a2 = n01 + n11;
            const int comparable = s0 + s1;
            const int cc = comparable < 255 ? comparable : 255;      // (a column of the device has at most 128 reads)
            const int f01c = T.floor01[cc], c09c = T.ceil09[cc], f01s = T.floor01[s0 < 255 ? s0 : 255], c01s = T.ceil01[s1 < 255 ? s1 : 255];
            const bool mid = elig & (s0 > f01c) & (s0 < c09c) & (a2 > f01c) & (a2 < c09c);
            if (__ballot(mid) != 0ull) corr = mid && chi_square_gt15(n00, n01, n10, n11);
            const int t0 = f01s > 1 ? f01s : 1, t1 = c01s > 1 ? c01s : 1;      // x <= max(0.1 s0, 1.0): x <= floor; x < max(0.1 s1, 1.0): x < ceiling
            fit = elig & ((unsigned)comparable >= half_n) & (((n01 <= t0) & (n10 < t1)) | ((n00 <= t0) & (n11 < t1)));
        }
        // the first fit in creation order takes the column; the partitions created before it (and it) count their correlation
        const unsigned long long fitm = __ballot(fit);
        const bool found = fitm != 0ull;
        int f = 0;
        if (found) {
            if ((fitm & (fitm - 1ull)) == 0ull) f = __builtin_ctzll(fitm);
            else {
                const int fit_pid = -wave_max_i32(fit ? -s_pid : -0x7fffffff);
                f = __builtin_ctzll(__ballot(fit & (s_pid == fit_pid)));
            }
        }
        const int f_pid = found ? __builtin_amdgcn_readlane(s_pid, f) : 0x7fffffff;
        const bool counts = corr & (s_pid <= f_pid);
        s_ncorr += counts ? 1 : 0;
        const int n_corr_col = __popcll(__ballot(counts));
        // the dead slots go to the pool now (they are complete): a freed slot is one comparison less for every later column
        unsigned long long D = __ballot(dead);
        while (D) { const int s = __builtin_ctzll(D); D &= D - 1ull; evict(s); }
        HS_LA_T(3);
        const int creach = __builtin_amdgcn_readlane(cur.hw, 5), i0 = __builtin_amdgcn_readlane(cur.hw, 3), i1 = __builtin_amdgcn_readlane(cur.hw, 4);
        if (found) {
            // ---- Partition::augmentPartition with the 'A' / 'a' / ' ' recoding of distance() folded in: lanes = the reads of a word ----
            if (lane == f) { s_left = (pos < s_left || s_left == -1) ? pos : s_left; s_right = pos > s_right ? pos : s_right; }
            const int f_shared = __builtin_amdgcn_readlane(shared, f);
            if (f_shared != 0) {
                const int sa = __builtin_amdgcn_readlane(best_slot, f);      // the slot of 'a' (-1: no read carries it)
                const int nA = ref_slot == 0 ? (int)((unsigned)__builtin_amdgcn_readlane(cur.hw, 16) >> 16) : 0;
                const int na = sa >= 0 ? (int)((unsigned)__builtin_amdgcn_readlane(cur.hw, 16 + (sa < 0 ? 0 : sa)) >> 16) : 0;
                int vA, va;                                   // the two most frequent characters of the recoded column, the lowest wins ties (:261-280)
                if (nA == 0 && na == 0) { vA = 0; va = 0; }
                else if (nA >= na) { vA = 1; va = na > 0 ? -1 : 0; }
                else { va = 1; vA = nA > 0 ? -1 : 0; }
                // phase vote over the shared reads (:284-314): the 2x2 table of the partition that took the column holds the four sums
                const int f11 = __builtin_amdgcn_readlane(n11, f), f01 = __builtin_amdgcn_readlane(n01, f), f10 = __builtin_amdgcn_readlane(n10, f), f00 = __builtin_amdgcn_readlane(n00, f);
                const int swapped = vA * (f11 - f01) + va * (f10 - f00);
                if (swapped < 0) { vA = -vA; va = -va; }
                const int pid = f_pid;
                bool over = false;
                uint8_t* drow = &s_d[f][lane];
                int d_old[WW];
#pragma unroll
                for (int w = 0; w < WW; ++w) d_old[w] = (int)drow[((wlo + w) & (HS_LA_RING - 1)) * 64];      // more - less of the lane's read in word w (meaningless where it is fresh)
                int32_t* crow = gc + (long long)pid * RS + wlo * 64 + lane;
                unsigned long long* tf = &la_tab[0][wlo][f];
#pragma unroll
                for (int w = 0; w < WW; ++w) {
                    const unsigned long long any = an[w];      // (uniform: made by v_readlane)
                    const unsigned long long A = ref_slot == 0 ? la_rl64(dw, 4 + w) : 0ull;
                    const unsigned long long a = sa >= 0 ? la_rl64(dw, 4 * (sa + 1) + w) : 0ull;
                    const unsigned long long PR = la_rl64(pr[w], f), PL = la_rl64(pl[w], f), MI = la_rl64(mi[w], f);
                    const unsigned long long s_plus = (vA == 1 ? A : 0ull) | (va == 1 ? a : 0ull), s_minus = (vA == -1 ? A : 0ull) | (va == -1 ? a : 0ull);
                    const unsigned long long voting = s_plus | s_minus;
                    const unsigned long long fresh = any & ~PR;                              // not in the partition yet: takes the vote as it is
                    const unsigned long long undecided = any & PR & ~PL & ~MI & voting;      // state 0 meets a vote: takes it
                    const unsigned long long agree = any & ((PL & s_plus) | (MI & s_minus));
                    const unsigned long long against = any & ((PL & s_minus) | (MI & s_plus));
                    const unsigned long long touched = fresh | undecided | agree | against;
                    const bool b_fresh = __builtin_amdgcn_inverse_ballot_w64(fresh), b_und = __builtin_amdgcn_inverse_ballot_w64(undecided);
                    const bool b_agree = __builtin_amdgcn_inverse_ballot_w64(agree), b_against = __builtin_amdgcn_inverse_ballot_w64(against);
                    const bool b_voting = __builtin_amdgcn_inverse_ballot_w64(voting), b_touched = __builtin_amdgcn_inverse_ballot_w64(touched);
                    const int dold = d_old[w];
                    const bool zero = dold == 0;
                    const unsigned long long flip = __ballot(b_against & zero);             // less + 1 > more: the state turns over (:372-379)
                    // fresh: more = 1 if it votes; state 0 meets a vote: more = 1, less = 0 (more was 0 or 1); same vote: more += 1; opposing vote: less += 1,
                    // or, turning over, more += 1
                    const int d_against = zero ? 1 : dold - 1, i_against = zero ? 1 : 65536;
                    const int d_new = b_fresh ? (b_voting ? 1 : 0) : (b_und ? 1 : (b_agree ? dold + 1 : d_against));
                    const int inc = b_fresh ? (b_voting ? 1 : 0) : (b_und ? 1 - dold : (b_agree ? 1 : i_against));
                    over = over | (b_touched & (d_new > 255));
                    if (b_touched) {
                        drow[((wlo + w) & (HS_LA_RING - 1)) * 64] = (uint8_t)d_new;
                        if (inc) __hip_atomic_fetch_add(crow + w * 64, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const unsigned long long take = fresh | undecided;
                    unsigned long long npl = PL | (take & s_plus), nmi = MI | (take & s_minus);
                    npl = (npl & ~(flip & PL)) | (flip & MI); nmi = (nmi & ~(flip & MI)) | (flip & PL);
                    if (lane == 0) { tf[w * HS_LA_SLOTS] = PR | any; tf[(HS_LA_MAXW + w) * HS_LA_SLOTS] = npl; tf[(2 * HS_LA_MAXW + w) * HS_LA_SLOTS] = nmi; }
                }
                if (lane == f) {
                    s_reach = creach > s_reach ? creach : s_reach;
                    const bool none = s_hi < s_lo;
                    s_lo = (none || i0 < s_lo) ? i0 : s_lo; s_hi = (none || i1 > s_hi) ? i1 : s_hi;
                    s_w0 = wlo < s_w0 ? wlo : s_w0;
                    s_w1 = wlo + WW - 1 > s_w1 ? wlo + WW - 1 : s_w1;
                    s_nocc += 1;
                }
                if (__ballot(over) != 0ull || __builtin_amdgcn_readlane(s_nocc, f) >= 65535) return false;      // (a counter beyond its width)
            }
            last_position = pos;
            HS_LA_T(4);
        } else {
            // ---- Partition::Partition(Column&, pos, ref_base): Partition.cpp:32-83 ----
            if (P >= p_cap) return false;
            const unsigned long long freem = __ballot(s_pid < 0);
            if (freem == 0ull) return false;      // 64 live partitions
            const int s = __builtin_ctzll(freem);
            const int pid = P++;
            int32_t* row = gc + (long long)pid * RS;
            for (int x = wlo * 64 + lane; x < RS; x += 64) row[x] = 0;      // (a read of this partition has a rank in or behind the column's first word; rank r is lane r % 64's from here on)
#pragma unroll
            for (int w = 0; w < WW; ++w) {
                const unsigned long long any = an[w];
                const unsigned long long A = ref_slot == 0 ? la_rl64(dw, 4 + w) : 0ull;
                const unsigned long long a = new_second >= 0 ? la_rl64(dw, 4 * (new_second + 1) + w) : 0ull;
                if (lane == 0) { la_tab[0][wlo + w][s] = any; la_tab[1][wlo + w][s] = A; la_tab[2][wlo + w][s] = a; }
                if (__builtin_amdgcn_inverse_ballot_w64(any)) { s_d[s][((wlo + w) & (HS_LA_RING - 1)) * 64 + lane] = 1; row[(wlo + w) * 64 + lane] = 1; }      // more = 1, less = 0
            }
            if (lane == s) {
                s_pid = pid; s_left = pos; s_right = pos; s_nocc = 1; s_ncorr = n_corr_col; s_reach = creach; s_lo = i0; s_hi = i1; s_w0 = wlo; s_w1 = wlo + WW - 1;
            }
            HS_LA_T(5);
        }
        return true;
    };

    for (long long k = 0; k < n_cand; ++k) {
        cur = n1;
        n1 = n2; n2 = n3; n3 = n4;
        if (k + 4 < n_cand) { n4.hw = hp[0]; n4.dw = wp[0]; }
        hp += 32; wp += 64;
        pos = __builtin_amdgcn_readlane(cur.hw, 0);
        if (pos - last_position <= 5) continue;                  // (:592)
        wlo = __builtin_amdgcn_readlane(cur.hw, 1); n = __builtin_amdgcn_readlane(cur.hw, 2);
        const int h6 = __builtin_amdgcn_readlane(cur.hw, 6), h7 = __builtin_amdgcn_readlane(cur.hw, 7);
        const int ww = (int)(short)(h6 & 0xffff);
        nslots = h6 >> 16; ref_slot = (int)(short)(h7 & 0xffff); new_second = h7 >> 16;
        ref = __builtin_amdgcn_readlane(cur.hw, 8);
#ifdef HS_LA_DIAG
        if (lane == 0) la_acc[8] += 1;
#endif
        if (ww <= 0 || n <= 0 || ref_slot > 0) { fail = true; break; }      // (too many codes / too deep / spread too wide: the host's)
        hi_water = wlo + ww > hi_water ? wlo + ww : hi_water;
        if (hi_water - wlo > HS_LA_RING || wlo + ww > W) { fail = true; break; }      // (the reads that can still meet a column span more words than the ring holds)
        HS_LA_T(0);
        bool ok;
        switch (ww) {
            case 1: ok = step(std::integral_constant<int, 1>()); break;
            case 2: ok = step(std::integral_constant<int, 2>()); break;
            case 3: ok = step(std::integral_constant<int, 3>()); break;
            default: ok = step(std::integral_constant<int, 4>()); break;
        }
        if (!ok) { fail = true; break; }
    }
    // the live partitions join the others in the pool
    unsigned long long U = __ballot(s_pid >= 0);
    while (U) { const int s = __builtin_ctzll(U); U &= U - 1ull; evict(s); }
    if (lane == 0) { n_parts[ci] = fail ? 0 : P; n_span[ci] = fail ? 0 : span_total; failed[ci] = fail ? 1 : 0; }
#ifdef HS_LA_DIAG
    HS_LA_T(7);
    if (lane == 0 && diag) for (int i = 0; i < 10; ++i) atomicAdd(&diag[i], (unsigned long long)la_acc[i]);
#else
    (void)diag;
#endif
}

// part_base[c] / span_base[c] = partitions / span words of the contigs before c (contig order); totals[0..1] = the two sums
__global__ __launch_bounds__(64) void k_loop_a_scan(const int32_t* __restrict__ n_parts, const int32_t* __restrict__ n_span, int c_count,
                                                    int64_t* __restrict__ part_base /* [C+1] */, int64_t* __restrict__ span_base /* [C+1] */, long long* __restrict__ totals) {
    if (blockIdx.x != 0) return;
    const int lane = lane_id();
    long long pb = 0, sb = 0;
    for (int c0 = 0; c0 < c_count; c0 += 64) {
        const int c = c0 + lane;
        const int p = c < c_count ? n_parts[c] : 0, s = c < c_count ? n_span[c] : 0;
        const int ip = wave_scan_incl(p), is = wave_scan_incl(s);
        if (c < c_count) { part_base[c] = pb + ip - p; span_base[c] = sb + is - s; }
        pb += __builtin_amdgcn_readlane(ip, 63); sb += __builtin_amdgcn_readlane(is, 63);
    }
    if (lane == 0) { part_base[c_count] = pb; span_base[c_count] = sb; totals[0] = pb; totals[1] = sb; }
}

// The partitions of every contig back to back, in the form the host imports: records, and per partition the words of its span
// [w0, w1] only -- present, plus, minus (3 x span words at 3 x word_off) and the counters of those words' reads (64 x span at
// 64 x word_off). One workgroup per contig the device walked.
__global__ __launch_bounds__(256) void k_loop_a_pack(
    const int32_t* __restrict__ dev_list, int n_dev, const int32_t* __restrict__ n_parts, int c_first, const int32_t* __restrict__ contig_rec_off,
    const int64_t* __restrict__ part_cap_off, const int64_t* __restrict__ bits_off, const LoopAPartition* __restrict__ parts,
    const unsigned long long* __restrict__ g_bits, const int32_t* __restrict__ g_cnt, const int64_t* __restrict__ part_base, const int64_t* __restrict__ span_base,
    long long cap_parts, long long cap_span, LoopAPartition* __restrict__ out_rec, unsigned long long* __restrict__ out_bits, int32_t* __restrict__ out_cnt) {
    __shared__ long long s_off[1024];
    if ((int)blockIdx.x >= n_dev) return;
    const int c = dev_list[blockIdx.x];
    const int P = n_parts[c];
    const int N = contig_rec_off[c_first + c + 1] - contig_rec_off[c_first + c];
    const int W = (N + 63) >> 6;
    const long long pb = part_base[c], sb = span_base[c];
    if (pb + P > cap_parts || span_base[c + 1] > cap_span) return;      // (the caller sees the totals and runs the pass again with room)
    const LoopAPartition* __restrict__ src = parts + part_cap_off[c];
    const unsigned long long* __restrict__ gb = g_bits + bits_off[c];
    for (int p0 = 0; p0 < P; p0 += 1024) {
        const int np = P - p0 < 1024 ? P - p0 : 1024;
        __syncthreads();
        if (threadIdx.x == 0) {      // (a contig ends with a few dozen partitions)
            long long o = p0 == 0 ? sb : s_off[1023] + (src[p0 - 1].w1 - src[p0 - 1].w0 + 1);
            for (int i = 0; i < np; ++i) { s_off[i] = o; o += src[p0 + i].w1 - src[p0 + i].w0 + 1; }
        }
        __syncthreads();
        for (int i = (int)(threadIdx.x >> 6); i < np; i += 4) {
            const int p = p0 + i;
            LoopAPartition r = src[p];
            const int span = r.w1 - r.w0 + 1;
            const long long wo = s_off[i];
            const int32_t* __restrict__ row = g_cnt + r.word_off;
            const int lane = (int)(threadIdx.x & 63u);
            for (int x = lane; x < 3 * span; x += 64) { const int t = x / span, w = x % span; out_bits[3 * wo + x] = gb[(long long)p * 3 * W + t * W + r.w0 + w]; }
            for (int x = lane; x < 64 * span; x += 64) out_cnt[64 * wo + x] = row[r.w0 * 64 + x];
            if (lane == 0) { r.word_off = wo; out_rec[pb + p] = r; }
        }
    }
}

}  // namespace hsdev
