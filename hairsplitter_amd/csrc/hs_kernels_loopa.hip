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

static __device__ __forceinline__ float chi_square_la(int n00, int n01, int n10, int n11) {      // computeChiSquare, call_variants.cpp:1135-1163
    Table2x2 t; t.n00 = n00; t.n01 = n01; t.n10 = n10; t.n11 = n11;
    return chi_square_dev(t);
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

// One wavefront per contig of the range. Dynamic LDS: the slot tables [3][W][64] u64 (present, plus, minus).
__global__ __launch_bounds__(64) void k_loop_a(
    const hs_colrec_dev* __restrict__ cand_rec, const int64_t* __restrict__ cand_ent_off, const int32_t* __restrict__ cand_idx, const uint8_t* __restrict__ cand_code,
    const int64_t* __restrict__ cand_off /* [C+1] */, int c_first, int c_count, const int32_t* __restrict__ contig_rec_off,
    const int32_t* __restrict__ rank_of /* per record of the batch */, const int32_t* __restrict__ orig_of /* per record: rank -> read of its contig */,
    const int32_t* __restrict__ read_end /* per record */, const int32_t* __restrict__ ctg_order /* heaviest contig first */,
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
    int32_t* gc = g_cnt + cnt_off[ci];                                 // partition p: counters [p * N, (p + 1) * N)
    const int32_t* __restrict__ rk_of = rank_of + r0;
    const int32_t* __restrict__ og_of = orig_of + r0;
    const int32_t* __restrict__ rend = read_end + r0;
    auto tab = [&](int t, int w, int slot) -> unsigned long long& { return la_tab[((size_t)t * w_cap + w) * HS_LA_SLOTS + slot]; };
    if (W > w_cap || W > HS_LA_MAXW) { if (lane == 0) { failed[ci] = 1; n_parts[ci] = 0; } return; }
    // slots: all free
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
    for (long long k = k0c; k < k1c && !fail; ++k) {
        const hs_colrec_dev rec = cand_rec[k];
        const int pos = rec.pos;
        if (pos - last_position <= 5) continue;                  // (:592)
        const int ref = (int)rec.k0;
        const long long e0 = cand_ent_off[k];
        const int n = (int)(cand_ent_off[k + 1] - e0);
        const int32_t* __restrict__ idx = cand_idx + e0;
        const uint8_t* __restrict__ code = cand_code + e0;
#ifdef HS_LA_DIAG
        if (lane == 0) la_acc[8] += 1;
#endif
        // ---- the column's words and codes ----
        int wlo = 0x7fffffff, whi = -1;
        for (int base = 0; base < n; base += 64) {
            const int e = base + lane;
            const int w = e < n ? (rk_of[idx[e]] >> 6) : -1;
            const int mn = -wave_max_i32(e < n ? -w : -0x7fffffff), mx = wave_max_i32(w);
            wlo = mn < wlo ? mn : wlo; whi = mx > whi ? mx : whi;
        }
        const int ww = whi - wlo + 1;
        HS_LA_T(0);
        if (n == 0 || ww > HS_LA_WIN) { fail = true; break; }
        for (int x = lane; x < HS_LA_CODES * HS_LA_WIN; x += 64) (&S.cb[0][0])[x] = 0ull;
        if (lane < HS_LA_WIN) S.any[lane] = 0ull;
        wave_lds_sync();
        int nslots = 0, ref_slot = -1;
        for (int base = 0; base < n && !fail; base += 64) {
            const int e = base + lane;
            const bool valid = e < n;
            const int cd = valid ? (int)code[e] : -1;
            const int rk = valid ? rk_of[idx[e]] : 0;
            unsigned long long rem = __ballot(valid);
            while (rem) {
                const int X = __builtin_amdgcn_readlane(cd, __builtin_ctzll(rem));
                const unsigned long long mX = __ballot(cd == X);
                rem &= ~mX;
                int ks = -1;
                for (int q = 0; q < nslots; ++q) if (S.code_of[q] == X) ks = q;      // (uniform: a handful of codes)
                if (ks < 0) {
                    if (nslots == HS_LA_CODES) { fail = true; break; }
                    ks = nslots++;
                    if (lane == 0) { S.code_of[ks] = X; S.cnt_of[ks] = 0; }
                    if (X == ref) ref_slot = ks;
                }
                if (lane == 0) S.cnt_of[ks] += __popcll(mX);
                if (cd == X) {
                    const unsigned long long bit = 1ull << (rk & 63);
                    atomicOr(&S.cb[ks][(rk >> 6) - wlo], bit);
                    atomicOr(&S.any[(rk >> 6) - wlo], bit);
                }
                wave_lds_sync();
            }
        }
        if (fail) break;
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
        if (elig) {
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
            corr = chi_square_la(n00, n01, n10, n11) > 15;
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
        // the dead slots go to the pool now (they are complete) -- only when room is needed or at the end would do as well, but a
        // freed slot is one comparison less for every later column
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
                for (int base = 0; base < n; base += 64) {
                    const int e = base + lane;
                    const int cd = e < n ? (int)code[e] : -1;
                    const bool isA = e < n && cd == ref, isa = e < n && cd == second && !isA;
                    nA += __popcll(__ballot(isA)); na += __popcll(__ballot(isa));
                }
                int vA, va;                                   // the two most frequent characters of the recoded column, the lowest wins ties (:261-280)
                if (nA == 0 && na == 0) { vA = 0; va = 0; }
                else if (nA >= na) { vA = 1; va = na > 0 ? -1 : 0; }
                else { va = 1; vA = nA > 0 ? -1 : 0; }
                int swapped = 0;                              // phase vote over the shared reads (:284-314)
                for (int base = 0; base < n; base += 64) {
                    const int e = base + lane;
                    int t = 0;
                    if (e < n) {
                        const int cd = (int)code[e];
                        const bool isA = cd == ref, isa = cd == second && !isA;
                        const int rk = rk_of[idx[e]];
                        const unsigned long long bit = 1ull << (rk & 63);
                        const bool pr = (tab(0, rk >> 6, f) & bit) != 0ull;
                        const int st = (tab(1, rk >> 6, f) & bit) ? 1 : ((tab(2, rk >> 6, f) & bit) ? -1 : 0);
                        const int v = isA ? vA : (isa ? va : 0);
                        t = pr ? v * st : 0;
                    }
                    swapped += __popcll(__ballot(t == 1)) - __popcll(__ballot(t == -1));
                }
                if (swapped < 0) { vA = -vA; va = -va; }
                int reach_l = -1;
                for (int base = 0; base < n; base += 64) {   // element-wise form of the sorted merge (:322-390)
                    const int e = base + lane;
                    if (e < n) {
                        const int r = idx[e];
                        const int cd = (int)code[e];
                        const bool isA = cd == ref, isa = cd == second && !isA;
                        const int s = isA ? vA : (isa ? va : 0);
                        const int rk = rk_of[r];
                        const int w = rk >> 6;
                        const unsigned long long bit = 1ull << (rk & 63);
                        const bool pr = (tab(0, w, f) & bit) != 0ull;
                        const int st = (tab(1, w, f) & bit) ? 1 : ((tab(2, w, f) & bit) ? -1 : 0);
                        int32_t* cp = gc + (long long)pid * N + r;      // (agent-scope accesses: served by L2, another lane's earlier store is seen)
                        int new_st = st;
                        bool change = false;
                        if (!pr) {
                            new_st = s; change = true;
                            __hip_atomic_store(cp, s < 0 ? -s : s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // more = |s|, less = 0
                            reach_l = rend[r] > reach_l ? rend[r] : reach_l;
                        } else if (s == 0) {
                        } else if (st == 0) { new_st = s; change = true; __hip_atomic_store(cp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                        else if (s == st) { atomicAdd(cp, 1); }
                        else {
                            const int v = __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                    const int i0 = idx[0], i1 = idx[n - 1];
                    if (S.hi[f] < S.lo[f]) { S.lo[f] = i0; S.hi[f] = i1; } else { if (i0 < S.lo[f]) S.lo[f] = i0; if (i1 > S.hi[f]) S.hi[f] = i1; }
                    S.n_occ[f] += 1;
                    if (S.n_occ[f] >= 65535) S.n_occ[f] = -1;      // (the 16-bit counters would wrap: reported below)
                }
            }
            last_position = pos;
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
            for (int base = 0; base < n; base += 64) {
                const int e = base + lane;
                if (e < n) {
                    const int r = idx[e];
                    const int cd = (int)code[e];
                    const int rk = rk_of[r];
                    const unsigned long long bit = 1ull << (rk & 63);
                    atomicOr(&tab(0, rk >> 6, s), bit);
                    if (cd == ref) atomicOr(&tab(1, rk >> 6, s), bit);
                    else if (cd == second) atomicOr(&tab(2, rk >> 6, s), bit);
                    __hip_atomic_store(gc + (long long)pid * N + r, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    reach_l = rend[r] > reach_l ? rend[r] : reach_l;
                }
            }
            reach_l = wave_max_i32(reach_l);
            if (lane == 0) {
                S.left[s] = pos; S.right[s] = pos; S.n_occ[s] = 1; S.n_corr[s] = n_corr_col; S.reach[s] = reach_l;
                S.lo[s] = idx[0]; S.hi[s] = idx[n - 1]; S.birth[s] = pid;
            }
            if (lane == s) slot_pid = pid;
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
