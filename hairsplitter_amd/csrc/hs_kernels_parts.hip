// hs_kernels_parts.hip -- loop A of keep_only_robust_variants on the device (call_variants.cpp:590-638): the candidate
// columns of a contig, in position order, are compared with the partitions found so far; a column that fits one augments
// it (Partition::augmentPartition, Partition.cpp:243-397), the others start a partition of their own (Partition.cpp:32-83).
// The chain over the columns of ONE contig is sequential by nature (each comparison sees the partitions as the previous
// column left them); contigs are independent.
//
//   k_robust_partitions   one wavefront per contig, LANES = PARTITIONS (blocks of 64). The states of a block of partitions
//                         are a byte table [read][64] (bit 0 present, bit 1 state +1, bit 2 state -1; all zero = absent),
//                         in LDS as far as it fits, the rest in global memory. Per candidate column (<= 64 reads) every
//                         lane gathers, from one 64-byte row per read of the column, three 64-bit masks over the
//                         column's entries -- the reads its partition holds, those with state +1, those with -1 -- and the
//                         whole of distance(Partition&, Column&) (call_variants.cpp:778-967) becomes popcounts of those
//                         masks against the (wave-uniform) entry masks of the column's codes. The verdicts of the 64
//                         partitions come out together; they are applied in partition order with two ballots (the
//                         first fit wins, only the partitions before it count a correlation: the reference's `break`).
//                         Second alleles that are tied among the shared reads (broken by hash-map order in the
//                         reference), reference codes >= 128 and columns deeper than 64 go through the exact
//                         wavefront-wide form (column_table_dev). Augmenting / creating a partition is lanes = reads.
//   k_partitions_scan     partitions per contig -> offsets of the packed download
//   k_partitions_unpack   the tables back to one row of N states / more / less per partition, for loop B on the host
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
// `tab` = the partition's column of a state table (stride 64 bytes per read), nullptr: every entry counts.
static __device__ __forceinline__ int state_of_byte(int b) { return b == 0 ? HS_PART_ABSENT : ((b & 2) ? 1 : ((b & 4) ? -1 : 0)); }
template <bool QUIRK, bool INSERT_REF>
static __device__ ColumnTable column_table_dev(const int32_t* __restrict__ idx, const uint8_t* __restrict__ code, int n,
                                               const uint8_t* tab, int ref, int dflt, uint8_t* s_seen /* [128] */,
                                               uint8_t* s_ord /* [264] */, int* s_ord_n /* [1] */) {
    const int lane = lane_id();
    int sc[2] = {-1, -1}, st_tot[2] = {0, 0}, st_pos[2] = {0, 0}, st_neg[2] = {0, 0};
    int nseen = 0, shared = 0;
    for (int base = 0; base < n; base += 64) {
        const int e = base + lane;
        const bool valid = e < n;
        const int cd = valid ? (int)code[e] : -1;
        const int stv = valid ? (tab ? state_of_byte((int)__hip_atomic_load(tab + (long long)idx[e] * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0) : HS_PART_ABSENT;
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

struct PartitionRecord { int32_t left, right, n_occ, n_corr, lo, hi, reach, pad; long long elem; };   // == hs::CvPartRecord

// scalars of the partitions, one slot per candidate column of the launch (a contig cannot have more partitions than
// candidates): partition p of contig c at slot cand_off[c] + p
struct PartitionScalars {
    int32_t* left; int32_t* right; int32_t* n_occ; int32_t* n_corr; int32_t* lo; int32_t* hi; int32_t* reach;
};

#define HS_PART_LDS_SCALARS 1024      // partitions whose `right` / `reach` (what decides whether a column looks at them) live in LDS

// One wavefront per contig. Dynamic LDS: right[1024] | reach[1024] | state tables of the first lds_blocks blocks.
__global__ __launch_bounds__(64) void k_robust_partitions(
    const int64_t* __restrict__ col_off, const int32_t* __restrict__ col_idx, const uint8_t* __restrict__ col_code,
    const int64_t* __restrict__ cand_off /* [C+1] */, const int32_t* __restrict__ cand_col, const int32_t* __restrict__ cand_pos,
    const uint8_t* __restrict__ cand_ref, const int32_t* __restrict__ ctg_n, const int64_t* __restrict__ read_off,
    const int32_t* __restrict__ read_end, const int32_t* __restrict__ ctg_order /* heaviest contig first */, int n_contigs,
    const int64_t* __restrict__ tab_off /* [C+1] elements: contig c owns blocks of N * 64 from tab_off[c] */, uint8_t* tab /* zeroed */,
    int32_t* tab_more, int32_t* tab_less, PartitionScalars ps, int32_t* __restrict__ n_parts, int lds_bytes) {
    extern __shared__ unsigned char parts_lds[];
    __shared__ uint8_t s_seen[128];
    __shared__ uint8_t s_ord[264];
    __shared__ int s_ord_n[1];
    __shared__ uint16_t s_rank[256];     // place of a code in the iteration order of a robin_hood map of <= 6 keys (see below)
    __shared__ uint16_t s_rank16[256];   // ... of 7..12 keys
    if ((int)blockIdx.x >= n_contigs) return;
    const int c = ctg_order[blockIdx.x];
    const int lane = lane_id();
    // robin_hood::unordered_flat_map<char, int> iterates its keys by (home bucket ascending, low five hash bits descending,
    // insertion order) as long as no key sits 6 or more slots from its home bucket (hs_rh8.h: home(), insert(), place()). The
    // first two are a function of the key and of the table's size: 8 buckets up to 6 keys, then 16 buckets and the next
    // multiplier up to 12 keys (grow()). Checked against the emulator by tests/test_cpu_oracle.py (rh8_static_order).
    for (int k = lane; k < 256; k += 64) {
        unsigned long long h = (unsigned long long)k;
        h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33;
        unsigned long long h0 = h * 0xc4ceb9fe1a85ec53ull; h0 ^= h0 >> 33;
        unsigned long long h1 = h * (0xc4ceb9fe1a85ec53ull + 0xc4ceb9fe1a85ec54ull); h1 ^= h1 >> 33;
        s_rank[k] = (uint16_t)((((h0 >> 5) & 7ull) << 5) | (31ull - (h0 & 31ull)));
        s_rank16[k] = (uint16_t)((((h1 >> 5) & 15ull) << 5) | (31ull - (h1 & 31ull)));
    }
    const long long k0 = cand_off[c], k1 = cand_off[c + 1];
    const int N = ctg_n[c];
    const int32_t* __restrict__ rend = read_end + read_off[c];
    int32_t* s_right = reinterpret_cast<int32_t*>(parts_lds);
    int32_t* s_reach = s_right + HS_PART_LDS_SCALARS;
    uint8_t* s_tab = parts_lds + 2 * HS_PART_LDS_SCALARS * 4;
    const long long blk = (long long)N * 64;                                            // bytes (elements) of one block of 64 partitions
    const int n_blocks_cap = (int)((tab_off[c + 1] - tab_off[c]) / (blk > 0 ? blk : 1));
    int lds_blocks = blk > 0 ? (int)(((long long)lds_bytes - 2 * HS_PART_LDS_SCALARS * 4) / blk) : 0;
    if (lds_blocks > n_blocks_cap) lds_blocks = n_blocks_cap;
    if (lds_blocks < 0) lds_blocks = 0;
    uint8_t* g_tab = tab + tab_off[c];
    int32_t* g_more = tab_more + tab_off[c];
    int32_t* g_less = tab_less + tab_off[c];
    for (long long x = lane; x < (long long)lds_blocks * blk / 4; x += 64) reinterpret_cast<uint32_t*>(s_tab)[x] = 0u;
    wave_lds_sync();
    int P = 0, last_position = -5;
    for (long long k = k0; k < k1; ++k) {
        const int pos = cand_pos[k];
        if (pos - last_position <= 5) continue;              // (:592)
        const int col = cand_col[k];
        const int ref = (int)cand_ref[k];
        const int64_t e0 = col_off[col];
        const int n = (int)(col_off[col + 1] - e0);
        const int32_t* __restrict__ idx = col_idx + e0;
        const uint8_t* __restrict__ code = col_code + e0;
        // the column across the lanes (lane i = entry i) and the entry masks of its codes
        const bool fast = n <= 64;
        const bool ref_elig = ref >= 128;      // the reference's signed / unsigned comparison (:838): such a reference code competes as second allele too
        const int my_idx = lane < n && fast ? idx[lane] : 0;
        const int my_code = lane < n && fast ? (int)code[lane] : -1;
        const unsigned long long m_ref = __ballot(my_code == ref);
        int found_p = -1, n_corr = 0, f_shared = 0, f_second = ' ';
        const int n_blocks = (P + 63) >> 6;
        for (int b = 0; b < n_blocks && found_p < 0; ++b) {
            const int pl = b * 64 + lane;
            bool elig = false;
            if (pl < P) {
                const int right = pl < HS_PART_LDS_SCALARS ? s_right[pl] : __hip_atomic_load(&ps.right[k0 + pl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int reach = pl < HS_PART_LDS_SCALARS ? s_reach[pl] : __hip_atomic_load(&ps.reach[k0 + pl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int dist = pos - right;
                elig = (dist < 0 ? -dist : dist) <= 50000 && pos < reach;     // (:595) and: no read of the partition reaches pos -> nothing shared
            }
            unsigned long long E = __ballot(elig);
            if (E == 0ull) continue;
            const bool in_lds = b < lds_blocks;
            const uint8_t* tb = in_lds ? s_tab + (long long)b * blk : g_tab + (long long)b * blk;      // (generic pointer: the exact path)
            int n00 = 0, n01 = 0, n10 = 0, n11 = 0, shared = 0, second = ' ';
            bool exact = !fast;
            if (fast) {
                // pass 1: per lane, which entries of the column its partition holds, and with which state
                unsigned plo = 0, phi = 0, slo = 0, shi = 0, mlo = 0, mhi = 0;      // present / state +1 / state -1, bit i = entry i
                const long long boff = (long long)b * blk + lane;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g * 16 < n) {                          // (uniform) sixteen rows in flight at a time; entries past n read row of entry 0's lane value 0 and are masked
                        unsigned by[16];
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const int r = __builtin_amdgcn_readlane(my_idx, g * 16 + j);
                            by[j] = in_lds ? (unsigned)s_tab[boff + (long long)r * 64]
                                           : (unsigned)__hip_atomic_load(g_tab + boff + (long long)r * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const int i = g * 16 + j;
                            const unsigned v = i < n ? by[j] : 0u;
                            if (i < 32) { plo |= (v & 1u) << i; slo |= ((v >> 1) & 1u) << i; mlo |= ((v >> 2) & 1u) << i; }
                            else { phi |= (v & 1u) << (i - 32); shi |= ((v >> 1) & 1u) << (i - 32); mhi |= ((v >> 2) & 1u) << (i - 32); }
                        }
                    }
                }
                const unsigned long long pres = ((unsigned long long)phi << 32) | plo, plus = ((unsigned long long)shi << 32) | slo, minus = ((unsigned long long)mhi << 32) | mlo;
                shared = __popcll(pres);
                n11 = __popcll(plus & m_ref); n01 = __popcll(minus & m_ref);
                // pass 2: the most frequent other code among the shared reads (:832-844). Equal counts: the reference takes the first of them in the iteration order of its hash map, which holds the
                // codes seen among the shared reads and the reference code -- for up to 6 keys that order is s_rank, then
                // insertion order (only then, or with more keys, the exact form below is needed)
                int best = -1, n_seen = 0, r8 = 0, r16 = 0, sec8 = ' ', sec16 = ' ';
                bool amb8 = false, amb16 = false;
                unsigned long long mb8 = 0ull, mb16 = 0ull, homes = 0ull;      // homes: keys per bucket of the 16-bucket table, 4 bits each
                unsigned long long rem = __ballot(my_code >= 0 && (ref_elig || my_code != ref));
                while (rem) {
                    const int X = __builtin_amdgcn_readlane(my_code, __builtin_ctzll(rem));
                    const unsigned long long mX = __ballot(my_code == X);
                    rem &= ~mX;
                    const int k8 = (int)s_rank[X], k16 = (int)s_rank16[X];
                    const int t = __popcll(pres & mX);
                    if (t > 0) {
                        n_seen++;
                        homes += 1ull << ((k16 >> 5) * 4);
                        if (t > best) { best = t; r8 = k8; r16 = k16; amb8 = amb16 = false; mb8 = mb16 = mX; sec8 = sec16 = X; }
                        else if (t == best) {
                            if (k8 < r8) { r8 = k8; amb8 = false; mb8 = mX; sec8 = X; } else if (k8 == r8) amb8 = true;
                            if (k16 < r16) { r16 = k16; amb16 = false; mb16 = mX; sec16 = X; } else if (k16 == r16) amb16 = true;
                        }
                    }
                }
                const bool ref_counted = ref_elig && (pres & m_ref) != 0ull;      // (then the loop above met it)
                const int n_keys = n_seen + (ref_counted ? 0 : 1);               // the map also holds the reference code
                if (!ref_counted) homes += 1ull << (((int)s_rank16[ref] >> 5) * 4);
                unsigned long long m_best;
                if (n_keys <= 6) { second = sec8; m_best = mb8; exact = amb8; }
                else {
                    second = sec16; m_best = mb16; exact = amb16 || n_keys > 12;
                    int carry = 0;                             // a key 6 or more slots from home: the table restructures itself differently
#pragma unroll
                    for (int bk = 0; bk < 16; ++bk) {
                        const int cb = (int)((homes >> (4 * bk)) & 15ull);
                        if (cb > 0 && carry + cb - 1 >= 6) exact = true;
                        carry = carry + cb - 1 > 0 ? carry + cb - 1 : 0;
                    }
                }
                if (best < 0) second = ' ';
                if (second == ref) m_best = 0ull;              // c == mostFrequent is tested first (:899-936): nothing is left for an equal second
                n10 = __popcll(plus & m_best); n00 = __popcll(minus & m_best);
            }
            // the few partitions that need the reference's tie order (or a deep column): wavefront-wide, one at a time
            unsigned long long X = __ballot(exact && elig);
            while (X) {
                const int l = __builtin_ctzll(X);
                X &= X - 1ull;
                const uint8_t* tcol = tb + l;
                const ColumnTable d = column_table_dev<true, true>(idx, code, n, tcol, ref, ' ', s_seen, s_ord, s_ord_n);
                if (lane == l) { n00 = d.n00; n01 = d.n01; n10 = d.n10; n11 = d.n11; shared = d.shared; second = d.second; }
            }
            const int comparable = n00 + n11 + n01 + n10;
            const double dc = (double)comparable;
            bool corr = false;
            if (elig && (double)(n00 + n01) > 0.1 * dc && (double)(n00 + n01) < 0.9 * dc && (double)(n01 + n11) > 0.1 * dc && (double)(n01 + n11) < 0.9 * dc) {
                Table2x2 t2; t2.n00 = n00; t2.n01 = n01; t2.n10 = n10; t2.n11 = n11;
                corr = chi_square_dev(t2) > 15;
            }
            const bool enough = (unsigned long long)comparable >= (unsigned long long)n / 2ull;
            const double m0 = 0.1 * (double)(n00 + n01), m1 = 0.1 * (double)(n11 + n10);
            const double t0 = m0 > 1.0 ? m0 : 1.0, t1 = m1 > 1.0 ? m1 : 1.0;      // std::max(x, 1.0)
            const bool fit = elig && enough && (((double)n01 <= t0 && (double)n10 < t1) || ((double)n00 <= t0 && (double)n11 < t1));
            const unsigned long long F = __ballot(fit);
            unsigned long long Cm = __ballot(corr);
            if (F) {                                           // the first fit wins; the partitions after it are not looked at (:630)
                const int f = __builtin_ctzll(F);
                Cm &= (f == 63 ? ~0ull : ((2ull << f) - 1ull));
                found_p = b * 64 + f;
                f_shared = __builtin_amdgcn_readlane(shared, f); f_second = __builtin_amdgcn_readlane(second, f);
            }
            if ((Cm >> lane) & 1ull) atomicAdd(&ps.n_corr[k0 + pl], 1);
            n_corr += __popcll(Cm);
        }
        if (found_p >= 0) {
            // Partition::augmentPartition with the 'A'/'a'/' ' recoding of distance() folded in (Partition.cpp:243-397 +
            // call_variants.cpp:856-872), element-wise over the reads of the column: lanes = entries
            const int b = found_p >> 6, fl = found_p & 63;
            const bool in_lds = b < lds_blocks;
            uint8_t* tcol = (in_lds ? s_tab + (long long)b * blk : g_tab + (long long)b * blk) + fl;
            int32_t* mcol = g_more + (long long)b * blk + fl;
            int32_t* lcol = g_less + (long long)b * blk + fl;
            const long long slot = k0 + found_p;
            if (lane == 0) { atomicMin(&ps.left[slot], pos); atomicMax(&ps.right[slot], pos); }      // (left is never -1 here: set at creation)
            if (found_p < HS_PART_LDS_SCALARS && lane == 0 && pos > s_right[found_p]) s_right[found_p] = pos;
            if (f_shared != 0 && n != 0) {                    // (else: empty partition_to_augment, :251-253)
                const int most = ref, second = f_second;
                int nA = 0, na = 0;
                for (int base = 0; base < n; base += 64) {
                    const int e = base + lane;
                    const int cd = e < n ? (int)code[e] : -1;
                    const bool isA = cd == most && e < n, isa = cd == second && !isA && e < n;
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
                        const bool isA = cd == most, isa = cd == second && !isA;
                        const int by = in_lds ? (int)tcol[(long long)idx[e] * 64] : (int)__hip_atomic_load(tcol + (long long)idx[e] * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const int s = state_of_byte(by);
                        const int v = isA ? vA : (isa ? va : 0);
                        t = s == HS_PART_ABSENT ? 0 : v * s;
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
                        const bool isA = cd == most, isa = cd == second && !isA;
                        const int s = isA ? vA : (isa ? va : 0);
                        uint8_t* tp = tcol + (long long)r * 64;
                        const int by = in_lds ? (int)*tp : (int)__hip_atomic_load(tp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const int st = state_of_byte(by);
                        const long long o = (long long)r * 64;
                        const uint8_t nb = (uint8_t)(s == 1 ? 3 : (s == -1 ? 5 : 1));
                        if (st == HS_PART_ABSENT) {
                            __hip_atomic_store(tp, nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(mcol + o, s < 0 ? -s : s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(lcol + o, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            reach_l = rend[r] > reach_l ? rend[r] : reach_l;
                        } else if (s == 0) {
                        } else if (st == 0) {
                            __hip_atomic_store(tp, nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(mcol + o, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(lcol + o, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else if (s == st) {
                            atomicAdd(mcol + o, 1);
                        } else {
                            const int mo = __hip_atomic_load(mcol + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const int le = __hip_atomic_load(lcol + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (le + 1 > mo) {
                                __hip_atomic_store(tp, (uint8_t)(st == 1 ? 5 : 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(mcol + o, mo + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            } else __hip_atomic_store(lcol + o, le + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(reach_l, d, 64); reach_l = o > reach_l ? o : reach_l; }
                if (lane == 0) {
                    atomicMax(&ps.reach[slot], reach_l);
                    if (found_p < HS_PART_LDS_SCALARS && reach_l > s_reach[found_p]) s_reach[found_p] = reach_l;
                    atomicMin(&ps.lo[slot], idx[0]); atomicMax(&ps.hi[slot], idx[n - 1]);
                    atomicAdd(&ps.n_occ[slot], 1);
                }
            }
            last_position = pos;
        } else {
            // Partition::Partition(Column&, pos, ref_base): Partition.cpp:32-83 (the table is all "absent" where nothing was written)
            const int b = P >> 6, fl = P & 63;
            const bool in_lds = b < lds_blocks;
            uint8_t* tcol = (in_lds ? s_tab + (long long)b * blk : g_tab + (long long)b * blk) + fl;
            int32_t* mcol = g_more + (long long)b * blk + fl;
            int32_t* lcol = g_less + (long long)b * blk + fl;
            const ColumnTable d = column_table_dev<false, false>(idx, code, n, nullptr, ref, 0, s_seen, s_ord, s_ord_n);
            int reach_l = -1;
            for (int base = 0; base < n; base += 64) {
                const int e = base + lane;
                if (e < n) {
                    const int r = idx[e];
                    const int cd = (int)code[e];
                    const long long o = (long long)r * 64;
                    __hip_atomic_store(tcol + o, (uint8_t)(cd == ref ? 3 : (cd == d.second ? 5 : 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(mcol + o, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(lcol + o, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    reach_l = rend[r] > reach_l ? rend[r] : reach_l;
                }
            }
#pragma unroll
            for (int dd = 32; dd >= 1; dd >>= 1) { const int o = __shfl_xor(reach_l, dd, 64); reach_l = o > reach_l ? o : reach_l; }
            if (lane == 0) {
                const long long slot = k0 + P;
                __hip_atomic_store(&ps.left[slot], pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ps.right[slot], pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ps.n_occ[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ps.n_corr[slot], n_corr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ps.reach[slot], reach_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ps.lo[slot], n ? idx[0] : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ps.hi[slot], n ? idx[n - 1] : -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (P < HS_PART_LDS_SCALARS) { s_right[P] = pos; s_reach[P] = reach_l; }
            }
            P++;
        }
        wave_lds_sync();      // (global memory is only touched with agent-scope atomics: nothing of this wavefront's own writes can be stale)
    }
    // the LDS part of the tables joins the rest in global memory
    wave_lds_sync();
    for (long long x = lane; x < (long long)lds_blocks * blk / 4; x += 64) reinterpret_cast<uint32_t*>(g_tab)[x] = reinterpret_cast<const uint32_t*>(s_tab)[x];
    if (lane == 0) n_parts[c] = P;
}

// part_base[c] = partitions of the contigs before c; elem_base[c] = elements (N per partition) of the contigs before c
__global__ __launch_bounds__(64) void k_partitions_scan(const int32_t* __restrict__ n_parts, const int32_t* __restrict__ ctg_n, int n_contigs,
                                                        int64_t* __restrict__ part_base /* [C+1] */, int64_t* __restrict__ elem_base /* [C+1] */) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long long p = 0, e = 0;
    for (int c = 0; c < n_contigs; ++c) { part_base[c] = p; elem_base[c] = e; p += n_parts[c]; e += (long long)n_parts[c] * ctg_n[c]; }
    part_base[n_contigs] = p; elem_base[n_contigs] = e;
}

// one workgroup per contig: records + one row of N states / more / less per partition
__global__ __launch_bounds__(256) void k_partitions_unpack(
    const int64_t* __restrict__ cand_off, const int32_t* __restrict__ n_parts, const int32_t* __restrict__ ctg_n, const int64_t* __restrict__ tab_off,
    const uint8_t* __restrict__ tab, const int32_t* __restrict__ tab_more, const int32_t* __restrict__ tab_less, PartitionScalars ps,
    const int64_t* __restrict__ part_base, const int64_t* __restrict__ elem_base, PartitionRecord* __restrict__ rec,
    int8_t* __restrict__ state, int32_t* __restrict__ more, int32_t* __restrict__ less) {
    const int c = (int)blockIdx.x;
    const int P = n_parts[c], N = ctg_n[c];
    const long long k0 = cand_off[c], pb = part_base[c], eb = elem_base[c];
    for (int p = (int)threadIdx.x; p < P; p += 256) {
        const long long s = k0 + p;
        PartitionRecord r;
        r.left = ps.left[s]; r.right = ps.right[s]; r.n_occ = ps.n_occ[s]; r.n_corr = ps.n_corr[s]; r.lo = ps.lo[s]; r.hi = ps.hi[s];
        r.reach = ps.reach[s]; r.pad = 0; r.elem = eb + (long long)p * N;
        rec[pb + p] = r;
    }
    const long long blk = (long long)N * 64;
    const long long total = (long long)P * N;
    for (long long x = threadIdx.x; x < total; x += 256) {      // x = p * N + r: rows are written contiguously
        const int p = (int)(x / N), r = (int)(x % N);
        const long long t = tab_off[c] + (long long)(p >> 6) * blk + (long long)r * 64 + (p & 63);
        const int by = (int)tab[t];
        state[eb + x] = (int8_t)state_of_byte(by);
        more[eb + x] = by ? tab_more[t] : 0;
        less[eb + x] = by ? tab_less[t] : 0;
    }
}

}  // namespace hsdev
