// hs_kernels_parts.hip -- loop A of keep_only_robust_variants on the device (call_variants.cpp:590-638): the candidate
// columns of a contig, in position order, are compared with the partitions found so far; a column that fits one augments
// it (Partition::augmentPartition, Partition.cpp:243-397), the others start a partition of their own (Partition.cpp:32-83).
// The chain over the columns of ONE contig is sequential by nature (each comparison sees the partitions as the previous
// column left them); contigs are independent.
//
//   k_robust_partitions   one workgroup (4 wavefronts) per contig. Per candidate column: the partitions it may touch
//                         (|pos - right| <= 50 kb, some read of the partition reaches pos) are taken four at a time, one
//                         comparison per wavefront -- distance(Partition&, Column&) + computeChiSquare exactly as
//                         k_column_partition_test does them (lanes = reads of the column, the 2x2 table out of ballots)
//                         -- and the verdicts are applied in partition order, so that the first fitting partition wins and
//                         only the partitions before it count a correlation (the reference's `break`).
//                         A partition = a row of N bytes (state of every read: 2 absent, -1 / 0 / +1) and two rows of N
//                         ints (more, less) in a pool shared by the launch, handed out by an atomic bump pointer.
//   k_partitions_pack     scalars of the partitions of every contig, packed in (contig, partition) order for the download.
//
// What is left to the host: the greedy spacing scan that names the candidates (:525-536, it needs the exact tie order of
// the column's top-3) and loop B (:646-708), whose merge decisions go through lgamma / exp / log in double precision
// (Partition.cpp:197-233, :716-732) -- the C library's values, not restated here.
#pragma once

namespace hsdev {

#define HS_PART_ABSENT 2

struct ColumnTable {
    int n00, n01, n10, n11;
    int shared;        // reads of the column the partition holds (0: "not comparable", call_variants.cpp:817-828)
    int second;        // the column's second allele among those reads
};

// distance(Partition&, Column&): call_variants.cpp:778-967 (QUIRK = true, INSERT_REF = true, dflt ' '), and the second
// allele of Partition::Partition(Column&): Partition.cpp:59-66 (state = nullptr: every entry counts; QUIRK = INSERT_REF =
// false, dflt 0). Same construction as column_vs_partition_dev (hs_kernels.hip); wave-uniform result.
template <bool QUIRK, bool INSERT_REF>
static __device__ ColumnTable column_table_dev(const int32_t* __restrict__ idx, const uint8_t* __restrict__ code, int n,
                                               const int8_t* state, int ref, int dflt, uint8_t* s_seen /* [128] */,
                                               uint8_t* s_ord /* [264] */, int* s_ord_n /* [1] */) {
    const int lane = lane_id();
    int sc[2] = {-1, -1}, st_tot[2] = {0, 0}, st_pos[2] = {0, 0}, st_neg[2] = {0, 0};
    int nseen = 0, shared = 0;
    for (int base = 0; base < n; base += 64) {
        const int e = base + lane;
        const bool valid = e < n;
        const int cd = valid ? (int)code[e] : -1;
        const int stv = valid ? (state ? (int)state[idx[e]] : 0) : HS_PART_ABSENT;
        const bool take = valid && stv != HS_PART_ABSENT;
        const unsigned long long plus = __ballot(take && stv == 1), minus = __ballot(take && stv == -1);
        unsigned long long rem = __ballot(take);
        shared += __popcll(rem);
        while (rem) {
            const int leader = __builtin_ctzll(rem);
            const int c = __builtin_amdgcn_readlane(cd, leader);
            const unsigned long long m = __ballot(take && cd == c);
            rem &= ~m;
            const int kt = __popcll(m), kp = __popcll(m & plus), kn = __popcll(m & minus);
            const unsigned long long hit0 = __ballot(sc[0] == c), hit1 = __ballot(sc[1] == c);
            if (hit0) { if (sc[0] == c) { st_tot[0] += kt; st_pos[0] += kp; st_neg[0] += kn; } }
            else if (hit1) { if (sc[1] == c) { st_tot[1] += kt; st_pos[1] += kp; st_neg[1] += kn; } }
            else {
                const int slot = nseen & 63;
                if (nseen < 64) { if (lane == slot) { sc[0] = c; st_tot[0] = kt; st_pos[0] = kp; st_neg[0] = kn; } }
                else { if (lane == slot) { sc[1] = c; st_tot[1] = kt; st_pos[1] = kp; st_neg[1] = kn; } }
                nseen++;
            }
        }
    }
    ColumnTable r; r.n00 = r.n01 = r.n10 = r.n11 = 0; r.shared = shared; r.second = dflt;
    if (shared == 0) return r;
    const unsigned long long ref0 = __ballot(sc[0] == ref), ref1 = __ballot(sc[1] == ref);
    const bool ref_seen = (ref0 | ref1) != 0ull;
    if (ref0) { const int l = __builtin_ctzll(ref0); r.n11 = __builtin_amdgcn_readlane(st_pos[0], l); r.n01 = __builtin_amdgcn_readlane(st_neg[0], l); }
    else if (ref1) { const int l = __builtin_ctzll(ref1); r.n11 = __builtin_amdgcn_readlane(st_pos[1], l); r.n01 = __builtin_amdgcn_readlane(st_neg[1], l); }
    const bool ref_eligible = QUIRK && ref >= 128;
    int key0 = (lane < nseen && (sc[0] != ref || ref_eligible)) ? st_tot[0] : -1;
    int key1 = (lane + 64 < nseen && (sc[1] != ref || ref_eligible)) ? st_tot[1] : -1;
    int best = key0 > key1 ? key0 : key1;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(best, d, 64); best = o > best ? o : best; }
    int second = dflt;
    bool second_is_unseen_ref = false;
    if (INSERT_REF && ref_eligible && !ref_seen && best < 0) { second = ref; second_is_unseen_ref = true; best = 0; }
    if (best >= 0 && !second_is_unseen_ref) {
        const unsigned long long b0 = __ballot(key0 == best), b1 = __ballot(key1 == best);
        const int nbest = __popcll(b0) + __popcll(b1) + ((INSERT_REF && ref_eligible && !ref_seen && best == 0) ? 1 : 0);
        if (nbest == 1) {
            second = b0 ? __builtin_amdgcn_readlane(sc[0], __builtin_ctzll(b0)) : __builtin_amdgcn_readlane(sc[1], __builtin_ctzll(b1));
        } else {
            // tie: first of the tied keys in the hash map's iteration order (keys inserted in first-appearance order, then ref)
            if (lane < nseen) s_seen[lane] = (uint8_t)sc[0];
            if (lane + 64 < nseen) s_seen[lane + 64] = (uint8_t)sc[1];
            wave_lds_sync();
            if (lane == 0) {
                hs::Rh8 rh; rh.clear();
                for (int i = 0; i < nseen; ++i) rh.insert(s_seen[i]);
                if (INSERT_REF) rh.insert((uint8_t)ref);
                s_ord_n[0] = rh.order(s_ord);
            }
            wave_lds_sync();
            const int m = s_ord_n[0];
            second = -1;
            for (int i = 0; i < m && second < 0; ++i) {
                const int k = s_ord[i];
                if (k == ref && !ref_eligible) continue;
                const unsigned long long h0 = __ballot(lane < nseen && sc[0] == k), h1 = __ballot(lane + 64 < nseen && sc[1] == k);
                int cnt = 0;
                if (h0) cnt = __builtin_amdgcn_readlane(st_tot[0], __builtin_ctzll(h0));
                else if (h1) cnt = __builtin_amdgcn_readlane(st_tot[1], __builtin_ctzll(h1));
                if (cnt == best) second = k;
            }
            if (second < 0) second = dflt;
            wave_lds_sync();
        }
    }
    r.second = second;
    if (second != ref) {
        const unsigned long long s0 = __ballot(lane < nseen && sc[0] == second), s1 = __ballot(lane + 64 < nseen && sc[1] == second);
        if (s0) { const int l = __builtin_ctzll(s0); r.n10 = __builtin_amdgcn_readlane(st_pos[0], l); r.n00 = __builtin_amdgcn_readlane(st_neg[0], l); }
        else if (s1) { const int l = __builtin_ctzll(s1); r.n10 = __builtin_amdgcn_readlane(st_pos[1], l); r.n00 = __builtin_amdgcn_readlane(st_neg[1], l); }
    }
    return r;
}

// scalars of the partitions, one slot per candidate column of the launch (a contig cannot have more partitions than
// candidates): partition p of contig c at slot cand_off[c] + p
struct PartitionScalars {
    int32_t* left; int32_t* right; int32_t* n_occ; int32_t* n_corr; int32_t* lo; int32_t* hi; int32_t* reach;
    long long* elem;       // first element of the partition's rows in the pool
};

struct PartitionRecord { int32_t left, right, n_occ, n_corr, lo, hi, reach, pad; long long elem; };   // == hs::CvPartRecord

struct PartitionVerdict { int p, corr, found, shared, second; };

static __device__ __forceinline__ void workgroup_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }

// Partition::augmentPartition with the 'A'/'a'/' ' recoding of distance() folded in (Partition.cpp:243-397 +
// call_variants.cpp:856-872), element-wise over the reads of the column: one wavefront, lanes = entries.
static __device__ void augment_partition_dev(const PartitionScalars& ps, long long slot, int8_t* state, int32_t* more, int32_t* less,
                                             const int32_t* __restrict__ idx, const uint8_t* __restrict__ code, int n, int most, int second,
                                             int shared, int pos, const int32_t* __restrict__ read_end) {
    const int lane = lane_id();
    if (lane == 0 && pos != -1) {
        const int l = ps.left[slot];
        if (pos < l || l == -1) ps.left[slot] = pos;
        if (pos > ps.right[slot]) ps.right[slot] = pos;
    }
    if (shared == 0 || n == 0) return;                       // empty partition_to_augment (:251-253)
    int nA = 0, na = 0;
    for (int base = 0; base < n; base += 64) {
        const int e = base + lane;
        const int cd = e < n ? (int)code[e] : -1;
        const bool isA = cd == most && e < n, isa = cd == second && !isA && e < n;
        nA += __popcll(__ballot(isA)); na += __popcll(__ballot(isa));
    }
    // the two most frequent characters of the recoded column, the lowest wins ties (:261-280): 'A' < 'a'
    int vA, va;
    if (nA == 0 && na == 0) { vA = 0; va = 0; }
    else if (nA >= na) { vA = 1; va = na > 0 ? -1 : 0; }
    else { va = 1; vA = nA > 0 ? -1 : 0; }
    int swapped = 0;                                         // phase vote over the shared reads (:284-314)
    for (int base = 0; base < n; base += 64) {
        const int e = base + lane;
        int t = 0;
        if (e < n) {
            const int cd = (int)code[e];
            const bool isA = cd == most, isa = cd == second && !isA;
            const int s = (int)state[idx[e]];
            const int v = isA ? vA : (isa ? va : 0);
            t = s == HS_PART_ABSENT ? 0 : v * s;
        }
        swapped += __popcll(__ballot(t == 1)) - __popcll(__ballot(t == -1));
    }
    if (swapped < 0) { vA = -vA; va = -va; }
    int reach_l = -1;
    for (int base = 0; base < n; base += 64) {              // element-wise form of the sorted merge (:322-390)
        const int e = base + lane;
        if (e < n) {
            const int r = idx[e];
            const int cd = (int)code[e];
            const bool isA = cd == most, isa = cd == second && !isA;
            const int s = isA ? vA : (isa ? va : 0);
            const int st = (int)state[r];
            if (st == HS_PART_ABSENT) { state[r] = (int8_t)s; more[r] = s < 0 ? -s : s; less[r] = 0; reach_l = read_end[r] > reach_l ? read_end[r] : reach_l; }
            else if (s == 0) { }
            else if (st == 0) { state[r] = (int8_t)s; more[r] = 1; less[r] = 0; }
            else if (s == st) more[r] += 1;
            else {
                const int mo = more[r], le = less[r];
                if (le + 1 > mo) { state[r] = (int8_t)-st; more[r] = mo + 1; }
                else less[r] = le + 1;
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(reach_l, d, 64); reach_l = o > reach_l ? o : reach_l; }
    if (lane == 0) {
        if (reach_l > ps.reach[slot]) ps.reach[slot] = reach_l;
        const int first = idx[0], last = idx[n - 1];
        if (ps.hi[slot] < ps.lo[slot]) { ps.lo[slot] = first; ps.hi[slot] = last; }
        else { if (first < ps.lo[slot]) ps.lo[slot] = first; if (last > ps.hi[slot]) ps.hi[slot] = last; }
        ps.n_occ[slot] += 1;
    }
}

__global__ __launch_bounds__(256) void k_robust_partitions(
    const int64_t* __restrict__ col_off, const int32_t* __restrict__ col_idx, const uint8_t* __restrict__ col_code,
    const int64_t* __restrict__ cand_off /* [C+1] */, const int32_t* __restrict__ cand_col, const int32_t* __restrict__ cand_pos,
    const uint8_t* __restrict__ cand_ref, const int32_t* __restrict__ ctg_n, const int64_t* __restrict__ read_off,
    const int32_t* __restrict__ read_end, const int32_t* __restrict__ ctg_order /* heaviest contig first */, int n_contigs,
    int8_t* pool_state, int32_t* pool_more, int32_t* pool_less, long long pool_cap, unsigned long long* pool_used,
    PartitionScalars ps, int32_t* __restrict__ n_parts, int32_t* __restrict__ failed) {
    __shared__ uint8_t s_seen[4][128];
    __shared__ uint8_t s_ord[4][264];
    __shared__ int s_ord_n[4];
    __shared__ PartitionVerdict s_res[4];
    __shared__ long long s_elem;
    if ((int)blockIdx.x >= n_contigs) return;
    const int c = ctg_order[blockIdx.x];
    const int lane = lane_id(), wv = wave_id(), tid = (int)threadIdx.x;
    const long long k0 = cand_off[c], k1 = cand_off[c + 1];
    const int N = ctg_n[c];
    const int32_t* __restrict__ rend = read_end + read_off[c];
    int P = 0, last_position = -5;
    for (long long k = k0; k < k1; ++k) {
        const int pos = cand_pos[k];
        if (pos - last_position <= 5) continue;              // (:592; every thread of the workgroup carries the same values)
        const int col = cand_col[k];
        const int ref = (int)cand_ref[k];
        const int64_t e0 = col_off[col];
        const int n = (int)(col_off[col + 1] - e0);
        const int32_t* __restrict__ idx = col_idx + e0;
        const uint8_t* __restrict__ code = col_code + e0;
        int found_p = -1, n_corr = 0, f_shared = 0, f_second = ' ';
        for (int chunk = 0; chunk < P && found_p < 0; chunk += 64) {
            const int pl = chunk + lane;
            bool elig = false;
            if (pl < P) {
                const int right = ps.right[k0 + pl], reach = ps.reach[k0 + pl];
                const int dist = pos - right;
                elig = (dist < 0 ? -dist : dist) <= 50000 && pos < reach;     // (:595) and: no read of the partition reaches pos -> nothing shared
            }
            unsigned long long M = __ballot(elig);           // the same in every wavefront
            while (M != 0ull && found_p < 0) {
                unsigned long long mine = M;
                for (int t = 0; t < wv; ++t) mine &= mine - 1ull;
                const int my_p = mine ? chunk + __builtin_ctzll(mine) : -1;
                PartitionVerdict v; v.p = my_p; v.corr = 0; v.found = 0; v.shared = 0; v.second = ' ';
                if (my_p >= 0) {
                    const ColumnTable d = column_table_dev<true, true>(idx, code, n, pool_state + ps.elem[k0 + my_p], ref, ' ', s_seen[wv], s_ord[wv], &s_ord_n[wv]);
                    const int comparable = d.n00 + d.n11 + d.n01 + d.n10;
                    Table2x2 t2; t2.n00 = d.n00; t2.n01 = d.n01; t2.n10 = d.n10; t2.n11 = d.n11;
                    const double dc = (double)comparable;
                    if ((double)(d.n00 + d.n01) > 0.1 * dc && (double)(d.n00 + d.n01) < 0.9 * dc && (double)(d.n01 + d.n11) > 0.1 * dc
                        && (double)(d.n01 + d.n11) < 0.9 * dc && chi_square_dev(t2) > 15) v.corr = 1;
                    const bool enough = (unsigned long long)comparable >= (unsigned long long)n / 2ull;
                    const double m0 = 0.1 * (double)(d.n00 + d.n01), m1 = 0.1 * (double)(d.n11 + d.n10);
                    const double t0 = m0 > 1.0 ? m0 : 1.0, t1 = m1 > 1.0 ? m1 : 1.0;      // std::max(x, 1.0)
                    if (((double)d.n01 <= t0 && (double)d.n10 < t1 && enough) || ((double)d.n00 <= t0 && (double)d.n11 < t1 && enough)) v.found = 1;
                    v.shared = d.shared; v.second = d.second;
                }
                if (lane == 0) s_res[wv] = v;
                __syncthreads();
                for (int t = 0; t < 4; ++t) {                // the verdicts in partition order: stop at the first fit (:630)
                    const PartitionVerdict r = s_res[t];
                    if (r.p < 0) break;
                    if (r.corr) { n_corr++; if (tid == 0) ps.n_corr[k0 + r.p] += 1; }
                    if (r.found) { found_p = r.p; f_shared = r.shared; f_second = r.second; break; }
                }
                for (int t = 0; t < 4 && M != 0ull; ++t) M &= M - 1ull;
                __syncthreads();
            }
        }
        if (found_p >= 0) {
            if (wv == 0) {
                const long long el = ps.elem[k0 + found_p];
                augment_partition_dev(ps, k0 + found_p, pool_state + el, pool_more + el, pool_less + el, idx, code, n, ref, f_second, f_shared, pos, rend);
            }
            last_position = pos;
        } else {
            // Partition::Partition(Column&, pos, ref_base): Partition.cpp:32-83
            if (tid == 0) s_elem = (long long)atomicAdd(pool_used, (unsigned long long)N);
            __syncthreads();
            const long long el = s_elem;
            if (el + (long long)N > pool_cap) { if (tid == 0) { failed[c] = 1; n_parts[c] = 0; } return; }      // pool exhausted: the host redoes this contig
            for (int j = tid; j < N; j += 256) pool_state[el + j] = (int8_t)HS_PART_ABSENT;
            workgroup_fence();
            __syncthreads();
            if (wv == 0) {
                const ColumnTable d = column_table_dev<false, false>(idx, code, n, nullptr, ref, 0, s_seen[0], s_ord[0], &s_ord_n[0]);
                int reach_l = -1;
                for (int base = 0; base < n; base += 64) {
                    const int e = base + lane;
                    if (e < n) {
                        const int r = idx[e];
                        const int cd = (int)code[e];
                        pool_state[el + r] = (int8_t)(cd == ref ? 1 : (cd == d.second ? -1 : 0));
                        pool_more[el + r] = 1; pool_less[el + r] = 0;
                        reach_l = rend[r] > reach_l ? rend[r] : reach_l;
                    }
                }
#pragma unroll
                for (int dd = 32; dd >= 1; dd >>= 1) { const int o = __shfl_xor(reach_l, dd, 64); reach_l = o > reach_l ? o : reach_l; }
                if (lane == 0) {
                    const long long slot = k0 + P;
                    ps.left[slot] = pos; ps.right[slot] = pos; ps.n_occ[slot] = 1; ps.n_corr[slot] = n_corr; ps.reach[slot] = reach_l;
                    ps.lo[slot] = n ? idx[0] : 0; ps.hi[slot] = n ? idx[n - 1] : -1; ps.elem[slot] = el;
                }
            }
            P++;
        }
        workgroup_fence();
        __syncthreads();
    }
    if (tid == 0) { n_parts[c] = P; failed[c] = 0; }
}

// part_base[c] = partitions of the contigs before c; scalars of partition p of contig c at part_base[c] + p
__global__ __launch_bounds__(256) void k_partitions_pack(
    const int64_t* __restrict__ cand_off, const int32_t* __restrict__ n_parts, int n_contigs, PartitionScalars ps,
    int64_t* __restrict__ part_base /* [C+1] */, PartitionRecord* __restrict__ out) {
    if (threadIdx.x == 0) {
        long long s = 0;
        for (int c = 0; c < n_contigs; ++c) { part_base[c] = s; s += n_parts[c]; }
        part_base[n_contigs] = s;
    }
    __threadfence_block();
    __syncthreads();
    for (int c = 0; c < n_contigs; ++c) {
        const long long b = part_base[c], k0 = cand_off[c];
        for (int p = (int)threadIdx.x; p < n_parts[c]; p += 256) {
            const long long s = k0 + p;
            PartitionRecord r;
            r.left = ps.left[s]; r.right = ps.right[s]; r.n_occ = ps.n_occ[s]; r.n_corr = ps.n_corr[s]; r.lo = ps.lo[s]; r.hi = ps.hi[s];
            r.reach = ps.reach[s]; r.pad = 0; r.elem = ps.elem[s];
            out[b + p] = r;
        }
    }
}

}  // namespace hsdev
