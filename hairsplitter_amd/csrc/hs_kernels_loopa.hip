// hs_kernels_loopa.hip -- loop A of keep_only_robust_variants on the device (call_variants.cpp:590-638): the candidate columns
// of a contig, in position order, meet the partitions found so far; a column that fits one augments it
// (Partition::augmentPartition, Partition.cpp:243-397), the others start a partition of their own (Partition.cpp:32-83). The
// chain over the columns of ONE contig is sequential by nature; contigs are independent: one wavefront per contig.
//
// Same formulation as the host's (hs_host_cv.cpp): a partition is three bit sets over the contig's reads -- present / state +1 /
// state -1 -- with bit k = the read of rank k by start position, so that the reads of a column sit in a few neighbouring 64-bit
// words whatever the order of the SAM file; distance(Partition&, Column&) (call_variants.cpp:778-967) is popcounts of ANDs of
// those words with the bit sets of the column's codes.
//   * The partitions a column can still meet (at most 50 kb behind, some read reaching the position: both conditions are final
//     once they fail, positions ascend) live in 64 LDS slots, LANES = SLOTS for the comparison: every lane forms the 2x2 table
//     of its partition, chi-square and the two verdicts (correlates / fits); the first fit in creation order wins (two wave
//     reductions), the partitions created before it count a correlation -- the reference's `break`.
//   * Building the column's bit sets, augmenting the partition that took it and starting a new one are LANES = ENTRIES of the
//     column (its reads), with LDS atomics on the slot's words; the per-read counters (more | less << 16) of a partition are a
//     row of a global table.
//   * Equal counts among the second alleles are broken by the reference in the iteration order of its hash map: the lane that
//     meets one replays it (hs::Rh8View on LDS tables), also for reference codes >= 128 (the reference's signed / unsigned
//     comparison, :838).
// A contig that does not fit the tables (more than 2048 reads, 64 live partitions, 16 codes in a column, a column spread over
// more than 16 words, the partition pool) is reported and done by the host (cv_phase_a_host).
// Included by hs_capi.hip after hs_kernels_cols.hip.
#pragma once

namespace hsdev {

#define HS_LA_SLOTS 64
#define HS_LA_MAXW 32          // words per bit set: contigs of up to 2048 reads
#define HS_LA_CODES 16         // distinct codes of a column
#define HS_LA_WIN 16           // words a column may spread over
#define HS_LA_FAST_W 4         // ... and the usual case, with the bit sets of its codes made ahead (k_loop_a_prepare)

struct LoopAPartition {        // == hs::CvPartRecord (what the host imports per partition)
    int32_t left, right, n_occ, n_corr, lo, hi, reach, pad;
    long long elem;            // first counter of the partition in the counter pool
};

#ifdef HS_LA_DIAG      // cycles of the sections of k_loop_a, summed over the wavefronts (stat[0..7]) + candidates (stat[8]) + exact-path lanes (stat[9])
#define HS_LA_T(i) do { const long long t__ = (long long)__builtin_amdgcn_s_memtime(); if (lane == 0) la_acc[i] += t__ - la_t; la_t = t__; } while (0)
#else
#define HS_LA_T(i) do { } while (0)
#endif

struct LoopAShared {           // fixed-size part of the LDS of a wavefront
    unsigned long long cb[HS_LA_CODES][HS_LA_WIN];      // bit sets of the column's codes over the column's words
    unsigned long long any[HS_LA_WIN];
    int code_of[HS_LA_CODES], cnt_of[HS_LA_CODES];
    int right[HS_LA_SLOTS], reach[HS_LA_SLOTS], birth[HS_LA_SLOTS], n_corr[HS_LA_SLOTS], n_occ[HS_LA_SLOTS], left[HS_LA_SLOTS], lo[HS_LA_SLOTS], hi[HS_LA_SLOTS];
    uint8_t rh_info[128], rh_key[128], rh_tmp[128];
    int x_seen[HS_LA_CODES], x_cnt[HS_LA_CODES], x_first[HS_LA_CODES];
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

// second_from_seen() of the host (hs_host_cv.cpp): the most frequent eligible code among `seen` (first-appearance order) with
// the reference's tie order; run by ONE lane (the tables in LDS belong to the wavefront)
static __device__ int second_from_seen_dev(LoopAShared& S, int nseen, int ref, bool quirk, bool insert_ref_last, int dflt) {
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

// Everything about a candidate column that does not depend on the partitions, made for all candidates at once (one wavefront
// per column) before the sequential kernel walks them:
//   cand_rc[e]   = rank of the entry's read by start position on its contig << 8 | its code
//   header       = first word the column's reads lie in, number of words, its distinct codes in first-appearance order with
//                  their counts, the slot of the reference code
//   64 words     = the bit sets of the codes over the column's words, when it has at most 15 codes in at most 4 words (nearly
//                  always): word w of `any` at [w], of code slot q at [4 (q + 1) + w] -- the sequential kernel keeps them one
//                  per lane and reads them with v_readlane
struct LoopAColumn {           // 64 bytes
    int32_t wlo;
    int16_t ww, nslots, ref_slot, fast;      // fast: the 64 words hold the bit sets; else the sequential kernel builds them (or gives up: ww = -1)
    uint8_t codes[HS_LA_CODES];
    uint16_t cnts[HS_LA_CODES];
    int16_t n, pad;                          // entries of the column (its first 128 sit in the column's row of cand_row)
};
static_assert(sizeof(LoopAColumn) == 64, "LoopAColumn layout");

__global__ __launch_bounds__(256) void k_loop_a_prepare(const hs_colrec_dev* __restrict__ cand_rec, const int64_t* __restrict__ cand_ent_off, const int32_t* __restrict__ cand_idx,
                                                        const uint8_t* __restrict__ cand_code, int64_t n_cand, const int32_t* __restrict__ contig_rec_off,
                                                        const int32_t* __restrict__ rank_of, int32_t* __restrict__ cand_row /* [n_cand][128] */, LoopAColumn* __restrict__ col_hdr,
                                                        unsigned long long* __restrict__ col_words, int32_t* __restrict__ col_ends /* [n_cand][2]: first and last read */) {
    __shared__ unsigned long long s_cb[4][HS_LA_CODES + 1][HS_LA_FAST_W];
    const int lane = lane_id();
    const int wv = wave_id();
    const int64_t k = (int64_t)blockIdx.x * 4 + wv;
    if (k >= n_cand) return;
    const hs_colrec_dev rec = cand_rec[k];
    const int r0 = contig_rec_off[rec.contig];
    const int ref = (int)rec.k0;
    const int64_t e0 = cand_ent_off[k];
    const int n = (int)(cand_ent_off[k + 1] - e0);
    int rc[2];      // (a column deeper than 128 reads goes to the host)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int e = c * 64 + lane;
        rc[c] = e < n ? ((rank_of[r0 + cand_idx[e0 + e]] << 8) | (int)cand_code[e0 + e]) : -1;
        cand_row[k * 128 + e] = rc[c];
    }
    if (lane == 0) { col_ends[2 * k] = n > 0 ? cand_idx[e0] : 0; col_ends[2 * k + 1] = n > 0 ? cand_idx[e0 + n - 1] : -1; }
    int wlo, whi;
    {
        const int w0 = rc[0] >= 0 ? (rc[0] >> 14) : -1, w1 = rc[1] >= 0 ? (rc[1] >> 14) : -1;
        const int hi_l = w0 > w1 ? w0 : w1;
        const int lo0 = rc[0] >= 0 ? w0 : 0x7fffffff, lo1 = rc[1] >= 0 ? w1 : 0x7fffffff;
        const int lo_l = lo0 < lo1 ? lo0 : lo1;
        whi = wave_max_i32(hi_l); wlo = -wave_max_i32(-lo_l);
    }
    const int ww = n > 0 ? whi - wlo + 1 : 0;
    const bool narrow = ww <= HS_LA_FAST_W && n > 0 && n <= 128;
    for (int x = lane; x < (HS_LA_CODES + 1) * HS_LA_FAST_W; x += 64) (&s_cb[wv][0][0])[x] = 0ull;
    wave_lds_sync();
    int slot_code = -1, slot_cnt = 0, nslots = 0, ref_slot = -1;
    bool many = false;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int cd = rc[c] >= 0 ? (rc[c] & 255) : -1;
        const int rk = rc[c] >> 8;
        unsigned long long rem = __ballot(rc[c] >= 0);
        while (rem) {
            const int X = __builtin_amdgcn_readlane(cd, __builtin_ctzll(rem));
            const unsigned long long mX = __ballot(cd == X);
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
            if (cd == X && narrow) {
                const unsigned long long bit = 1ull << (rk & 63);
                atomicOr(&s_cb[wv][ks + 1][(rk >> 6) - wlo], bit);
                atomicOr(&s_cb[wv][0][(rk >> 6) - wlo], bit);
            }
        }
    }
    wave_lds_sync();
    const bool fast = narrow && !many && nslots <= HS_LA_CODES - 1;
    LoopAColumn* h = col_hdr + k;
    if (lane == 0) { h->wlo = wlo; h->ww = (int16_t)((many || n == 0 || n > 128) ? -1 : ww); h->nslots = (int16_t)nslots; h->ref_slot = (int16_t)ref_slot; h->fast = fast ? 1 : 0; h->n = (int16_t)(n > 32767 ? 32767 : n); h->pad = 0; }
    if (lane < HS_LA_CODES) { h->codes[lane] = (uint8_t)(slot_code < 0 ? 0 : slot_code); h->cnts[lane] = (uint16_t)slot_cnt; }
    col_words[k * 64 + lane] = fast ? (&s_cb[wv][0][0])[lane] : 0ull;
}

// One wavefront per contig of the range. Dynamic LDS: the slot tables [3][W][64] u64 (present, plus, minus).
// The column in flight sits in registers (two chunks of 64 entries: rank << 8 | code per lane; a deeper column sends the contig to
// the host), the next one is loaded while this one is decided. Columns that lie in at most HS_LA_FAST words (nearly all) are
// compared with the table words of the lanes' partitions in registers.
#define HS_LA_FAST HS_LA_FAST_W
__global__ __launch_bounds__(64) void k_loop_a(
    const hs_colrec_dev* __restrict__ cand_rec, const int32_t* __restrict__ cand_row, const int32_t* __restrict__ col_ends,
    const LoopAColumn* __restrict__ col_hdr, const unsigned long long* __restrict__ col_words,
    const int64_t* __restrict__ cand_off /* [C+1] */, int c_first, int c_count, const int32_t* __restrict__ contig_rec_off,
    const int32_t* __restrict__ orig_of /* per record: rank -> read of its contig */, const int32_t* __restrict__ read_end_by_rank /* per record, rank order */,
    const int32_t* __restrict__ ctg_order /* heaviest contig first */,
    const int64_t* __restrict__ part_cap_off /* [C+1] partitions the pool holds per contig */, const int64_t* __restrict__ bits_off /* [C+1] words */,
    const int64_t* __restrict__ cnt_off /* [C+1] counters */, LoopAPartition* __restrict__ parts, unsigned long long* __restrict__ g_bits, int32_t* g_cnt,
    int32_t* __restrict__ n_parts /* [C] */, int32_t* __restrict__ failed /* [C] */, int w_cap, unsigned long long* __restrict__ diag) {
    extern __shared__ unsigned long long la_tab[];      // [3][w_cap][64]
    __shared__ LoopAShared S;
    if ((int)blockIdx.x >= c_count) return;
    const int ci = ctg_order[blockIdx.x];
    const int lane = lane_id();
    const int r0 = contig_rec_off[c_first + ci];
    const int N = contig_rec_off[c_first + ci + 1] - r0;
    const int W = (N + 63) >> 6;
    const long long k0c = cand_off[ci], k1c = cand_off[ci + 1];
    const long long p_base = part_cap_off[ci];
    const int p_cap = (int)(part_cap_off[ci + 1] - p_base);
    unsigned long long* __restrict__ gb = g_bits + bits_off[ci];      // partition p: words [p * 3 W, (p + 1) * 3 W)
    int32_t* gc = g_cnt + cnt_off[ci];                                 // partition p: counters [p * N, (p + 1) * N), indexed by RANK
    const int32_t* __restrict__ og_of = orig_of + r0;
    const int32_t* __restrict__ rend = read_end_by_rank + r0;
    auto tab = [&](int t, int w, int slot) -> unsigned long long& { return la_tab[((size_t)t * w_cap + w) * HS_LA_SLOTS + slot]; };
    if (W > w_cap || W > HS_LA_MAXW) { if (lane == 0) { failed[ci] = 1; n_parts[ci] = 0; } return; }
    int slot_pid = -1;                 // lane = slot: the partition it holds (-1: free)
    S.birth[lane] = 0x7fffffff;
    wave_lds_sync();
    int P = 0, last_position = -5;
    bool fail = false;
#ifdef HS_LA_DIAG
    long long la_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long la_t = (long long)__builtin_amdgcn_s_memtime();
#endif
    // writes slot `s` (wave-uniform) to the pool and frees it; all lanes take part
    auto evict = [&](int s, int pid) {
        for (int x = lane; x < 3 * W; x += 64) gb[(long long)pid * 3 * W + x] = tab(x / W, x % W, s);
        if (lane == 0) {
            LoopAPartition r;
            r.left = S.left[s]; r.right = S.right[s]; r.n_occ = S.n_occ[s]; r.n_corr = S.n_corr[s]; r.lo = S.lo[s]; r.hi = S.hi[s]; r.reach = S.reach[s]; r.pad = 0;
            r.elem = cnt_off[ci] + (long long)pid * N;
            parts[p_base + pid] = r;
            S.birth[s] = 0x7fffffff;
        }
    };
    // the column in flight and the next one (loaded one column ahead)
    // Everything of a column comes in with VECTOR loads (lane-dependent addresses) one column ahead: scalar loads share their
    // counter with the LDS, and the first LDS access after them would wait for the whole prefetch.
    struct Col { int recw, hdrw, endw, rc0, rc1; unsigned long long dw; };
    auto load_col = [&](long long k) {
        Col c;
        c.recw = reinterpret_cast<const int*>(cand_rec)[k * 4 + (lane & 3)];      // lane 0: position, lane 3: k0 | k1 << 8 | ...
        c.hdrw = reinterpret_cast<const int*>(col_hdr)[k * 16 + (lane & 15)];    // the 16 words of the LoopAColumn
        c.endw = col_ends[2 * k + (lane & 1)];
        c.rc0 = cand_row[k * 128 + lane];
        c.rc1 = cand_row[k * 128 + 64 + lane];
        c.dw = col_words[k * 64 + lane];
        return c;
    };
    int last_pid = -1;         // the partition that took the previous column: its counters for this column's reads are fetched ahead
    Col nxt;
    nxt.recw = 0; nxt.hdrw = 0; nxt.endw = 0; nxt.rc0 = -1; nxt.rc1 = -1; nxt.dw = 0ull;
    if (k0c < k1c) nxt = load_col(k0c);
    for (long long k = k0c; k < k1c && !fail; ++k) {
        const Col cur = nxt;
        if (k + 1 < k1c) nxt = load_col(k + 1);
        const int pos = __builtin_amdgcn_readlane(cur.recw, 0);
        if (pos - last_position <= 5) continue;                  // (:592)
        const int ref = __builtin_amdgcn_readlane(cur.recw, 3) & 255;
        const int n = __builtin_amdgcn_readlane(cur.hdrw, 15) & 0xffff;
        const int cur_i0 = __builtin_amdgcn_readlane(cur.endw, 0), cur_i1 = __builtin_amdgcn_readlane(cur.endw, 1);
        if (n == 0 || n > 128) { fail = true; break; }
        const int rc[2] = {cur.rc0, cur.rc1};
#ifdef HS_LA_DIAG
        if (lane == 0) la_acc[8] += 1;
#endif
        int spec[2] = {0, 0};      // counters of `last_pid` for this column's reads: nearly always the partition that takes this column too
        if (last_pid >= 0) {
#pragma unroll
            for (int c = 0; c < 2; ++c) if (rc[c] >= 0) spec[c] = __hip_atomic_load(gc + (long long)last_pid * N + (rc[c] >> 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- the column's words and codes: made ahead by k_loop_a_prepare ----
        const int wlo = __builtin_amdgcn_readlane(cur.hdrw, 0);
        const int h1 = __builtin_amdgcn_readlane(cur.hdrw, 1), h2 = __builtin_amdgcn_readlane(cur.hdrw, 2);
        const int ww = (int)(short)(h1 & 0xffff), nslots = (int)(short)(h1 >> 16), ref_slot = (int)(short)(h2 & 0xffff);
        const bool fast_col = (h2 >> 16) != 0;
        const unsigned long long dw = cur.dw;      // fast column: word w of `any` in lane w, of code slot q in lane 4 (q + 1) + w
        // lane q: the code of slot q (first-appearance order) and its count, out of the header words 3..6 / 7..14
        const int codes_w = __shfl(cur.hdrw, 3 + ((lane & 15) >> 2), 64), cnts_w = __shfl(cur.hdrw, 7 + ((lane & 15) >> 1), 64);
        const int cur_scode = (codes_w >> (8 * (lane & 3))) & 255, cur_scnt = (cnts_w >> (16 * (lane & 1))) & 0xffff;
        const int slot_code = lane < nslots ? cur_scode : -1;
        HS_LA_T(0);
        if (ww < 0 || ww > HS_LA_WIN) { fail = true; break; }     // (too many codes / too deep / spread too wide: the host's)
        if (lane < HS_LA_CODES) { S.code_of[lane] = slot_code; S.cnt_of[lane] = cur_scnt; }
        if (!fast_col) {
            // the unusual column (more than four words): its bit sets into LDS here
            for (int x = lane; x < HS_LA_CODES * HS_LA_WIN; x += 64) (&S.cb[0][0])[x] = 0ull;
            if (lane < HS_LA_WIN) S.any[lane] = 0ull;
            wave_lds_sync();
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (rc[c] >= 0) {
                    const int cd = rc[c] & 255, rk = rc[c] >> 8;
                    int ks = -1;
                    for (int q = 0; q < nslots; ++q) if (__builtin_amdgcn_readlane(cur_scode, q) == cd) ks = q;
                    const unsigned long long bit = 1ull << (rk & 63);
                    atomicOr(&S.cb[ks][(rk >> 6) - wlo], bit);
                    atomicOr(&S.any[(rk >> 6) - wlo], bit);
                }
            }
        }
        wave_lds_sync();
        HS_LA_T(1);
        // ---- the live partitions against the column: lanes = slots ----
        const bool used = slot_pid >= 0;
        bool elig = false;
        if (used) {
            const int dist = pos - S.right[lane];
            elig = (dist < 0 ? -dist : dist) <= 50000 && pos < S.reach[lane];
        }
        const bool dead = used && !elig;      // (final: positions ascend, right / reach only move when the partition is augmented)
        int n00 = 0, n01 = 0, n10 = 0, n11 = 0;
        bool need_exact = false;
        int best = -1, nbest = 0, best_slot = -1, shared = 0;
        auto rl64 = [&](unsigned long long v, int l) -> unsigned long long {      // v_readlane of a 64-bit value (l wave-uniform)
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffull), l);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
            return ((unsigned long long)hi << 32) | lo;
        };
        if (fast_col) {
            // the usual column: its (at most four) words of the lane's partition in registers, the words of the column's codes out of
            // the lanes of `dw` as scalars: every count is ANDs and popcounts on registers
            unsigned long long pr[HS_LA_FAST], pl[HS_LA_FAST], mi[HS_LA_FAST], an[HS_LA_FAST];
#pragma unroll
            for (int w = 0; w < HS_LA_FAST; ++w) {
                const bool in = w < ww && elig;
                const int ws = in ? wlo + w : 0;
                pr[w] = tab(0, ws, lane); pl[w] = tab(1, ws, lane); mi[w] = tab(2, ws, lane); an[w] = rl64(dw, w);
                if (!in) { pr[w] = 0ull; pl[w] = 0ull; mi[w] = 0ull; }
            }
            int decided = 0;
#pragma unroll
            for (int w = 0; w < HS_LA_FAST; ++w) { shared += __popcll(an[w] & pr[w]); decided += __popcll(an[w] & (pl[w] | mi[w])); }
            const bool skip = !elig || shared == 0 || (shared <= 14 && (unsigned)shared < (unsigned)n / 2u) || (decided <= 14 && (unsigned)decided < (unsigned)n / 2u);
            if (__ballot(!skip) != 0ull) {
                for (int q = 0; q < nslots; ++q) {
                    if (q == ref_slot) continue;
                    int c = 0;
#pragma unroll
                    for (int w = 0; w < HS_LA_FAST; ++w) c += __popcll(rl64(dw, 4 * (q + 1) + w) & pr[w]);
                    if (c > 0) { if (c > best) { best = c; nbest = 1; best_slot = q; } else if (c == best) nbest++; }
                }
                if (skip) { best_slot = -1; nbest = 0; }
                else if (ref >= 128 || nbest > 1) need_exact = true;
                else {
                    if (ref_slot >= 0) {
#pragma unroll
                        for (int w = 0; w < HS_LA_FAST; ++w) { const unsigned long long x = rl64(dw, 4 * (ref_slot + 1) + w); n11 += __popcll(x & pl[w]); n01 += __popcll(x & mi[w]); }
                    }
                }
                // the words of the lane's own second allele: a gather across the lanes of `dw`
                const bool want = !skip && !need_exact && best_slot >= 0;
#pragma unroll
                for (int w = 0; w < HS_LA_FAST; ++w) {
                    const int src = want ? 4 * (best_slot + 1) + w : 0;
                    const unsigned lo = (unsigned)__shfl((int)(unsigned)(dw & 0xffffffffull), src, 64), hi = (unsigned)__shfl((int)(unsigned)(dw >> 32), src, 64);
                    const unsigned long long x = want ? (((unsigned long long)hi << 32) | lo) : 0ull;
                    n10 += __popcll(x & pl[w]); n00 += __popcll(x & mi[w]);
                }
            }
            if (__ballot(need_exact) != 0ull) {      // the tie order needs the column's bit sets where the general form reads them
                S.any[lane & 15] = 0ull;
                if (lane < 4) S.any[lane] = dw;
                else (&S.cb[0][0])[((lane >> 2) - 1) * HS_LA_WIN + (lane & 3)] = dw;
                wave_lds_sync();
            }
        } else if (elig) {
            int decided = 0;
            for (int w = 0; w < ww; ++w) {
                const unsigned long long a = S.any[w];
                shared += __popcll(a & tab(0, wlo + w, lane));
                decided += __popcll(a & (tab(1, wlo + w, lane) | tab(2, wlo + w, lane)));
            }
            // few shared (or decided) reads: the table can neither fit nor correlate, its counts are of no consequence (hs_host_cv.cpp)
            const bool skip = shared == 0 || (shared <= 14 && (unsigned)shared < (unsigned)n / 2u) || (decided <= 14 && (unsigned)decided < (unsigned)n / 2u);
            if (!skip) {
                if (ref >= 128) need_exact = true;
                else {
                    for (int q = 0; q < nslots; ++q) {
                        if (q == ref_slot) continue;
                        int c = 0;
                        for (int w = 0; w < ww; ++w) c += __popcll(S.cb[q][w] & tab(0, wlo + w, lane));
                        if (c == 0) continue;
                        if (c > best) { best = c; nbest = 1; best_slot = q; } else if (c == best) nbest++;
                    }
                    if (nbest > 1) need_exact = true;
                    else {
                        if (ref_slot >= 0) for (int w = 0; w < ww; ++w) { n11 += __popcll(S.cb[ref_slot][w] & tab(1, wlo + w, lane)); n01 += __popcll(S.cb[ref_slot][w] & tab(2, wlo + w, lane)); }
                        if (best_slot >= 0) for (int w = 0; w < ww; ++w) { n10 += __popcll(S.cb[best_slot][w] & tab(1, wlo + w, lane)); n00 += __popcll(S.cb[best_slot][w] & tab(2, wlo + w, lane)); }
                    }
                }
            }
        }
        HS_LA_T(2);
        // the lanes whose second allele needs the reference's tie order, one at a time (column_vs_partition_bits, general form)
        unsigned long long X = __ballot(need_exact);
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
                    for (int w = 0; w < ww; ++w) c += __popcll(S.cb[q][w] & tab(0, wlo + w, lane));
                    if (c) { S.x_seen[nseen] = S.code_of[q]; S.x_cnt[nseen] = c; S.x_first[nseen] = q; nseen++; }      // (x_first: the slot for now)
                }
                const bool ref_eligible = ref >= 128;
                int bst = -1, nb = 0;
                bool ref_seen = false;
                for (int i = 0; i < nseen; ++i) {
                    if (S.x_seen[i] == ref) { ref_seen = true; if (!ref_eligible) continue; }
                    if (S.x_cnt[i] > bst) { bst = S.x_cnt[i]; nb = 1; } else if (S.x_cnt[i] == bst) nb++;
                }
                if (ref_eligible && !ref_seen) { if (0 > bst) { bst = 0; nb = 1; } else if (bst == 0) nb++; }
                if (nb > 1) {
                    // the hash map meets the codes in the order of the READ INDICES of the shared reads that carry them
                    for (int i = 0; i < nseen; ++i) {
                        const int q = S.x_first[i];
                        int f = 0x7fffffff;
                        for (int w = 0; w < ww; ++w) {
                            unsigned long long x = S.cb[q][w] & tab(0, wlo + w, lane);
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
                }
                const int second = second_from_seen_dev(S, nseen, ref, true, true, ' ');
                int sm = -1, ss = -1;
                for (int q = 0; q < nslots; ++q) { if (S.code_of[q] == ref) sm = q; if (S.code_of[q] == second) ss = q; }
                if (sm >= 0) for (int w = 0; w < ww; ++w) { n11 += __popcll(S.cb[sm][w] & tab(1, wlo + w, lane)); n01 += __popcll(S.cb[sm][w] & tab(2, wlo + w, lane)); }
                if (ss >= 0 && second != ref) for (int w = 0; w < ww; ++w) { n10 += __popcll(S.cb[ss][w] & tab(1, wlo + w, lane)); n00 += __popcll(S.cb[ss][w] & tab(2, wlo + w, lane)); }
                best_slot = ss;
            }
            wave_lds_sync();
        }
        HS_LA_T(3);
        // verdicts (:611-627), per lane
        const int comparable = n00 + n11 + n01 + n10;
        const double dc = (double)comparable;
        bool corr = false;
        if (elig && (double)(n00 + n01) > 0.1 * dc && (double)(n00 + n01) < 0.9 * dc && (double)(n01 + n11) > 0.1 * dc && (double)(n01 + n11) < 0.9 * dc)
            corr = chi_square_gt15(n00, n01, n10, n11);
        const bool enough = (unsigned long long)comparable >= (unsigned long long)n / 2ull;
        const double m0 = 0.1 * (double)(n00 + n01), m1 = 0.1 * (double)(n11 + n10);
        const double t0 = m0 > 1.0 ? m0 : 1.0, t1 = m1 > 1.0 ? m1 : 1.0;      // std::max(x, 1.0)
        const bool fit = elig && enough && (((double)n01 <= t0 && (double)n10 < t1) || ((double)n00 <= t0 && (double)n11 < t1));
        // the first fit in creation order takes the column; the partitions created before it (and it) count their correlation
        const int my_birth = used ? S.birth[lane] : 0x7fffffff;
        const int fit_birth = -wave_max_i32(fit ? -my_birth : -0x7fffffff);
        const bool found = fit_birth != 0x7fffffff;
        const bool counts = corr && (!found || my_birth <= fit_birth);
        if (counts) S.n_corr[lane] += 1;
        const int n_corr_col = __popcll(__ballot(counts));
        const unsigned long long Fm = __ballot(fit && my_birth == fit_birth);
        // the dead slots go to the pool now (they are complete): a freed slot is one comparison less for every later column
        unsigned long long D = __ballot(dead);
        while (D) { const int s = __builtin_ctzll(D); D &= D - 1ull; const int pid = __builtin_amdgcn_readlane(slot_pid, s); evict(s, pid); if (lane == s) slot_pid = -1; }
        wave_lds_sync();
        HS_LA_T(4);
        if (found) {
            // ---- Partition::augmentPartition with the 'A' / 'a' / ' ' recoding of distance() folded in: lanes = entries ----
            const int f = __builtin_ctzll(Fm);
            const int pid = __builtin_amdgcn_readlane(slot_pid, f);
            const int f_shared = __builtin_amdgcn_readlane(shared, f);
            const int f_second_slot = __builtin_amdgcn_readlane(best_slot, f);
            const int second = f_second_slot >= 0 ? S.code_of[f_second_slot] : (int)' ';
            if (lane == 0) { if (pos < S.left[f] || S.left[f] == -1) S.left[f] = pos; if (pos > S.right[f]) S.right[f] = pos; }
            if (f_shared != 0) {
                int nA = 0, na = 0;
                bool isA[2], isa[2], pr_[2];
                int st_[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const bool v = rc[c] >= 0;
                    const int cd = rc[c] & 255, rk = rc[c] >> 8;
                    isA[c] = v && cd == ref; isa[c] = v && cd == second && !isA[c];
                    nA += __popcll(__ballot(isA[c])); na += __popcll(__ballot(isa[c]));
                    const int w = v ? rk >> 6 : 0;
                    const unsigned long long bit = 1ull << (rk & 63);
                    pr_[c] = v && (tab(0, w, f) & bit) != 0ull;
                    st_[c] = (tab(1, w, f) & bit) ? 1 : ((tab(2, w, f) & bit) ? -1 : 0);
                }
                int vA, va;                                   // the two most frequent characters of the recoded column, the lowest wins ties (:261-280)
                if (nA == 0 && na == 0) { vA = 0; va = 0; }
                else if (nA >= na) { vA = 1; va = na > 0 ? -1 : 0; }
                else { va = 1; vA = nA > 0 ? -1 : 0; }
                int swapped = 0;                              // phase vote over the shared reads (:284-314)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int v = isA[c] ? vA : (isa[c] ? va : 0);
                    const int t = pr_[c] ? v * st_[c] : 0;
                    swapped += __popcll(__ballot(t == 1)) - __popcll(__ballot(t == -1));
                }
                if (swapped < 0) { vA = -vA; va = -va; }
                int reach_l = -1;
#pragma unroll
                for (int c = 0; c < 2; ++c) {                 // element-wise form of the sorted merge (:322-390)
                    if (rc[c] >= 0) {
                        const int s = isA[c] ? vA : (isa[c] ? va : 0);
                        const int rk = rc[c] >> 8;
                        const int w = rk >> 6;
                        const unsigned long long bit = 1ull << (rk & 63);
                        const bool pr = pr_[c];
                        const int st = st_[c];
                        int32_t* cp = gc + (long long)pid * N + rk;      // (agent-scope accesses: served by L2, another lane's earlier store is seen)
                        int new_st = st;
                        bool change = false;
                        if (!pr) {
                            new_st = s; change = true;
                            __hip_atomic_store(cp, s < 0 ? -s : s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // more = |s|, less = 0
                            reach_l = rend[rk] > reach_l ? rend[rk] : reach_l;
                        } else if (s == 0) {
                        } else if (st == 0) { new_st = s; change = true; __hip_atomic_store(cp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                        else if (s == st) { atomicAdd(cp, 1); }
                        else {
                            const int v = pid == last_pid ? spec[c] : __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const int mo = v & 0xffff, le = (v >> 16) & 0xffff;
                            if (le + 1 > mo) { new_st = -st; change = true; __hip_atomic_store(cp, (mo + 1) | (le << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                            else __hip_atomic_store(cp, mo | ((le + 1) << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if (change) {
                            atomicOr(&tab(0, w, f), bit);
                            if (new_st == 1) { atomicOr(&tab(1, w, f), bit); atomicAnd(&tab(2, w, f), ~bit); }
                            else if (new_st == -1) { atomicOr(&tab(2, w, f), bit); atomicAnd(&tab(1, w, f), ~bit); }
                            else { atomicAnd(&tab(1, w, f), ~bit); atomicAnd(&tab(2, w, f), ~bit); }
                        }
                    }
                }
                reach_l = wave_max_i32(reach_l);
                if (lane == 0) {
                    if (reach_l > S.reach[f]) S.reach[f] = reach_l;
                    if (S.hi[f] < S.lo[f]) { S.lo[f] = cur_i0; S.hi[f] = cur_i1; } else { if (cur_i0 < S.lo[f]) S.lo[f] = cur_i0; if (cur_i1 > S.hi[f]) S.hi[f] = cur_i1; }
                    S.n_occ[f] += 1;
                    if (S.n_occ[f] >= 65535) S.n_occ[f] = -1;      // (the 16-bit counters would wrap: reported below)
                }
            }
            last_position = pos;
            last_pid = pid;
            wave_lds_sync();
            HS_LA_T(5);
            if (S.n_occ[f] < 0) { fail = true; break; }
        } else {
            // ---- Partition::Partition(Column&, pos, ref_base): Partition.cpp:32-83 ----
            if (P >= p_cap) { fail = true; break; }
            const unsigned long long freem = __ballot(slot_pid < 0);
            if (freem == 0ull) { fail = true; break; }      // 64 live partitions
            const int s = __builtin_ctzll(freem);
            // the second allele over ALL entries (second_most_frequent, no quirk, the reference code not inserted, default 0)
            int second = 0;
            {
                int bst = -1, nb = 0, bq = -1;
                for (int q = 0; q < nslots; ++q) {
                    if (q == ref_slot) continue;
                    const int c = S.cnt_of[q];
                    if (c > bst) { bst = c; nb = 1; bq = q; } else if (c == bst) nb++;
                }
                if (nb == 1) second = S.code_of[bq];
                else if (nb > 1) {
                    if (lane == 0) {
                        for (int q = 0; q < nslots; ++q) { S.x_seen[q] = S.code_of[q]; S.x_cnt[q] = S.cnt_of[q]; }      // slots are in first-appearance order
                        S.x_first[0] = second_from_seen_dev(S, nslots, ref, false, false, 0);
                    }
                    wave_lds_sync();
                    second = S.x_first[0];
                }
            }
            const int pid = P++;
            for (int x = lane; x < 3 * W; x += 64) tab(x / W, x % W, s) = 0ull;
            for (long long x = lane; x < N; x += 64) __hip_atomic_store(gc + (long long)pid * N + x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wave_lds_sync();
            int reach_l = -1;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (rc[c] >= 0) {
                    const int cd = rc[c] & 255, rk = rc[c] >> 8;
                    const unsigned long long bit = 1ull << (rk & 63);
                    atomicOr(&tab(0, rk >> 6, s), bit);
                    if (cd == ref) atomicOr(&tab(1, rk >> 6, s), bit);
                    else if (cd == second) atomicOr(&tab(2, rk >> 6, s), bit);
                    __hip_atomic_store(gc + (long long)pid * N + rk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    reach_l = rend[rk] > reach_l ? rend[rk] : reach_l;
                }
            }
            reach_l = wave_max_i32(reach_l);
            if (lane == 0) {
                S.left[s] = pos; S.right[s] = pos; S.n_occ[s] = 1; S.n_corr[s] = n_corr_col; S.reach[s] = reach_l;
                S.lo[s] = cur_i0; S.hi[s] = cur_i1; S.birth[s] = pid;
            }
            if (lane == s) slot_pid = pid;
            last_pid = pid;
            wave_lds_sync();
            HS_LA_T(6);
        }
    }
    // the live partitions join the others in the pool
    wave_lds_sync();
    unsigned long long U = __ballot(slot_pid >= 0);
    while (U) { const int s = __builtin_ctzll(U); U &= U - 1ull; const int pid = __builtin_amdgcn_readlane(slot_pid, s); evict(s, pid); }
    if (lane == 0) { n_parts[ci] = fail ? 0 : P; failed[ci] = fail ? 1 : 0; }
#ifdef HS_LA_DIAG
    HS_LA_T(7);
    if (lane == 0 && diag) for (int i = 0; i < 10; ++i) atomicAdd(&diag[i], (unsigned long long)la_acc[i]);
#else
    (void)diag;
#endif
}

// part_base[c] = partitions of the contigs before c (contig order)
__global__ __launch_bounds__(64) void k_loop_a_scan(const int32_t* __restrict__ n_parts, int c_count, int64_t* __restrict__ part_base /* [C+1] */) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long long p = 0;
    for (int c = 0; c < c_count; ++c) { part_base[c] = p; p += n_parts[c]; }
    part_base[c_count] = p;
}

// the partitions of every contig back to back: records, bit sets (3 W words each) and counters (N each); one workgroup per contig.
// out_bits_base / out_cnt_base: [C] first word / counter of the contig in the packed arrays (prefixes of P x 3 W and P x N)
__global__ __launch_bounds__(256) void k_loop_a_pack(
    const int32_t* __restrict__ n_parts, int c_first, const int32_t* __restrict__ contig_rec_off, const int64_t* __restrict__ part_cap_off, const int64_t* __restrict__ bits_off,
    const int64_t* __restrict__ cnt_off, const LoopAPartition* __restrict__ parts, const unsigned long long* __restrict__ g_bits, const int32_t* __restrict__ g_cnt,
    const int64_t* __restrict__ part_base, const int64_t* __restrict__ out_bits_base, const int64_t* __restrict__ out_cnt_base,
    LoopAPartition* __restrict__ out_rec, unsigned long long* __restrict__ out_bits, int32_t* __restrict__ out_cnt) {
    const int c = (int)blockIdx.x;
    const int P = n_parts[c];
    const int N = contig_rec_off[c_first + c + 1] - contig_rec_off[c_first + c];
    const int W = (N + 63) >> 6;
    const long long pb = part_base[c];
    for (int p = (int)threadIdx.x; p < P; p += 256) { LoopAPartition r = parts[part_cap_off[c] + p]; r.elem = out_cnt_base[c] + (long long)p * N; out_rec[pb + p] = r; }
    const long long nb = (long long)P * 3 * W, nc = (long long)P * N;
    for (long long x = threadIdx.x; x < nb; x += 256) out_bits[out_bits_base[c] + x] = g_bits[bits_off[c] + x];
    for (long long x = threadIdx.x; x < nc; x += 256) out_cnt[out_cnt_base[c] + x] = g_cnt[cnt_off[c] + x];
}

}  // namespace hsdev
