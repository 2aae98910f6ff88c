// hs_kernels_myers.hip -- A1: banded Myers bit-vector alignment with the results of the reference's
// bundled edlib. What stage 5 asks of it is edlibAlign(query, target, k = -1, EDLIB_MODE_HW, EDLIB_TASK_PATH) (create_new_contigs.cpp:558-629,
// tools.cpp:515-534) -- edit distance, first end location, its start location and one optimal alignment -- for a 200-300 bp
// query against a target of a few hundred to a few thousand bases there; any query up to 2^20 bases here. One wavefront per
// pair, Myers sweeps (lane b owns query block b, 64 blocks per pass; at step s lane b is at target column s - b and takes the
// horizontal carry of the block above from the lane before it; the bottom row of a pass is handed to the next through memory):
//   1. HW (infix): minimum of the bottom row, FIRST column that attains it            (edlib.cpp:560-700; the k-doubling of
//      edlibAlign :194-214 only bounds the band of the reference's own search: the optimum it returns is the exact one)
//   2. SHW of the reversed query on the reversed target prefix that ends there, LAST best column = the start location
//      (:226-258: "taking last location as start ensures that alignment will not start with insertions")
//   3. the alignment on target[start .. end] as obtainAlignment forms it (:1166-1219). A matrix edlib would keep whole (below
//      1 MB of its own bookkeeping: 300 x 9700, 1800 x 1800) -> NW with the vertical delta words (P, M) and the bottom score of
//      every (column, block) kept (:737-930 with findAlignment), then the traceback of :947-1130 by lane 0: up (insertion)
//      before left (deletion) before the diagonal, on exact cell scores recomputed from the stored words. A larger one ->
//      Hirschberg as edlib does it (:1236-1404): the target cut in halves, NW of the left half forwards and of the right half
//      backwards, the FIRST query row where the two last columns add up to the optimum, recursion on the upper-left and the
//      lower-right part -- which of the equally good alignments comes out depends on these cuts, so they are edlib's. The
//      recursion is a stack of frames in LDS, walked depth first (upper left first: the moves come out in order); the
//      scratch of a pair is one leaf matrix (<= 1.26 MB) + two columns of ints, whatever the lengths.
//      edlib computes these matrices inside a band around the diagonal; every score it READS there is the exact one, so any
//      computation that is exact on the cells of alignments within the bound gives the same decisions: whole columns
//      (oracle/edlib_path_oracle.py, numpy, pinned against the reference's edlib up to 60 kb) or the static band of the sweeps
//      here (MyersBand below).
// Kernels: k_myers_hw_path (a wavefront per pair, any length), k_myers_hw_path_grouped<8|16|32> (short queries, 64 / G pairs per
// wavefront), k_myers_distance / k_myers_distance_grouped<G> (hs_edit_distance: NW / SHW / HW distance + first end location).
// Alignment ops as edlib's: 0 match, 1 insertion (query base without target base), 2 deletion, 3 mismatch.
// Sequences are 2-bit base codes (A C G T), as everywhere on this path. Included by hs_capi.hip after hs_kernels.hip.
#pragma once
#define MY_TCHUNK 2048      /* target columns staged in LDS at a time */

namespace hsdev {

struct MyersSeq {              // a sequence seen forwards or backwards
    const uint8_t* p; int n; bool rev;
    __device__ __forceinline__ int at(int i) const { return (int)(p[rev ? n - 1 - i : i] & 3); }
    // rows r0 .. r0 + 63 (those below n) as two bit planes: bit k of m0 / m1 = bit 0 / 1 of the code of row r0 + k. Sixteen
    // (unaligned) dword loads in flight; bit b of the four bytes of a dword lands in one nibble through a multiplication whose
    // partial products do not meet (byte i, bit b -> bit 28 + i forwards, 31 - i backwards).
    __device__ __forceinline__ void planes(int r0, uint64_t& m0, uint64_t& m1, uint64_t& valid) const {
        const int nrow = (n - r0) < 64 ? (n - r0) : 64;
        valid = nrow >= 64 ? ~0ull : ((1ull << nrow) - 1ull);
        uint32_t x[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int r = r0 + 4 * w;
            x[w] = 0u;
            if (r + 3 < n) x[w] = rev ? *reinterpret_cast<const u32_unaligned*>(p + (n - 4 - r)) : *reinterpret_cast<const u32_unaligned*>(p + r);
            else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (r + b < n) x[w] |= (uint32_t)p[rev ? n - 1 - (r + b) : r + b] << (rev ? 8 * (3 - b) : 8 * b);
            }
        }
        const uint32_t mul = rev ? ((1u << 31) | (1u << 22) | (1u << 13) | (1u << 4)) : ((1u << 28) | (1u << 21) | (1u << 14) | (1u << 7));
        uint32_t lo0 = 0, lo1 = 0, hi0 = 0, hi1 = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t n0 = ((x[w] & 0x01010101u) * mul) >> 28, n1 = (((x[w] >> 1) & 0x01010101u) * mul) >> 28;
            if (w < 8) { lo0 |= n0 << (4 * w); lo1 |= n1 << (4 * w); } else { hi0 |= n0 << (4 * (w - 8)); hi1 |= n1 << (4 * (w - 8)); }
        }
        m0 = ((uint64_t)hi0 << 32) | lo0; m1 = ((uint64_t)hi1 << 32) | lo1;
    }
};
// lane i takes the value of lane i - 1 (lane 0: zero): one DPP move across the whole wavefront
static __device__ __forceinline__ int wave_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }

// The band of a sweep: block b (query rows 64 b .. 64 b + 63) is computed for the columns 64 b + lo .. 64 b + 63 + hi only
// (Ukkonen: a cell on an alignment of at most k errors lies within k diagonals of where the alignment starts and of where it
// ends). A block that is not computed hands +1 per column to the block below it and a block that enters the band starts from
// +1 per row -- what edlib assumes at the edges of its own band (edlib.cpp:773-817): every score inside the band is then an upper
// bound, and exact on every cell an alignment of at most k errors passes through; those are the only ones the results read.
struct MyersBand {
    int lo, hi;
    static __device__ __forceinline__ MyersBand whole() { return MyersBand{-(1 << 29), 1 << 29}; }
    // start and end fixed (NW, also a half of Hirschberg's cut of a qn x tn problem of score k): diagonals [-k, k] and [D - k, D + k]
    static __device__ __forceinline__ MyersBand global(int qn, int tn, int k) {
        const int D = tn - qn;
        MyersBand b{max(-k, D - k), min(k, D + k)};
        if (b.lo > 0) b.lo = 0;
        if (b.hi < b.lo + 1) b.hi = b.lo + 1;
        return b;
    }
    static __device__ __forceinline__ MyersBand prefix(int k) { return MyersBand{-k, k < 1 ? 1 : k}; }                 // SHW: start fixed, any end column
    static __device__ __forceinline__ MyersBand infix(int qn, int tn, int k) {                                        // HW: any start column, any end column
        MyersBand b{-k, tn - qn + k};
        if (b.hi < b.lo + 1) b.hi = b.lo + 1;
        return b;
    }
    __device__ __forceinline__ bool holds(int blk, int col) const { return col >= (blk << 6) + lo && col <= (blk << 6) + 63 + hi; }
};
#define MY_INF (1 << 28)

// One sweep. mode 0 NW, 1 SHW, 2 HW. Outputs through references (valid in every lane): final bottom-row score of the
// last column, best bottom-row score over the columns, first and last column attaining it (-1: before the target).
// hb: [tn] the horizontal deltas of a pass's last block for the next pass, hbot: [tn] its bottom scores (a block of the next pass
// that enters the band starts from them).
// store != nullptr: P, M, bottom score of every (column, block) of the band at store[(col * nblocks + blk) * 3 ...] as three
// 64-bit words {P, M, score}. col_scores != nullptr: the scores of the LAST column, one int per query row (what Hirschberg's
// split reads), MY_INF outside the band.
static __device__ void myers_sweep(const MyersSeq& q, const MyersSeq& t, int mode, MyersBand band, int8_t* __restrict__ hb, int32_t* __restrict__ hbot,
                                   uint8_t* tbuf /* LDS [MY_TCHUNK + 64] */, unsigned long long* __restrict__ store, int32_t* __restrict__ col_scores,
                                   int& out_score, int& out_best, int& out_first, int& out_last) {
    const int lane = lane_id();
    const int qn = q.n, tn = t.n;
    const int nblocks = (qn + 63) >> 6;
    const int last_row = (qn - 1) & 63;
    // The reference's edlib pads the query to a multiple of 64 rows and reads the score of column c off column c + W (W = padding
    // rows, edlib.cpp:664-690): with W > 0 the columns "before the target" (score = query length) take part and win ties, which is
    // what best = qn, first = -1 reproduces; with W == 0 there are none, and the first real column that reaches the best score --
    // query length included -- is the answer (64 x 'A' in 'CCC...': end location 0, path 1X63I).
    int score = qn, best = (qn & 63) == 0 ? qn + 1 : qn, best_first = -1, best_last = -1;
    bool reached_end = false;
    for (int pb = 0; pb < nblocks; pb += 64) {
        const int blk = pb + lane;
        const bool bvalid = blk < nblocks;
        const bool is_last_blk = blk == nblocks - 1;
        const int nb_pass = (nblocks - pb) < 64 ? (nblocks - pb) : 64;
        // this block's columns, the block above's last column
        const int jlo_u = (blk << 6) + band.lo, jhi_u = (blk << 6) + 63 + band.hi;
        const int jlo = jlo_u > 0 ? jlo_u : 0, jhi = jhi_u < tn - 1 ? jhi_u : tn - 1;
        const int jhi_up = (jhi_u - 64) < tn - 1 ? (jhi_u - 64) : tn - 1;
        const bool some = bvalid && jlo <= jhi;
        int s_begin = some ? jlo + lane : 0x7fffffff, s_last = some ? jhi + lane : -1;
        for (int o = 32; o > 0; o >>= 1) { s_begin = min(s_begin, __shfl_xor(s_begin, o, 64)); s_last = max(s_last, __shfl_xor(s_last, o, 64)); }
        uint64_t peq[4] = {0, 0, 0, 0};
        if (some) {
            uint64_t m0, m1, valid;
            q.planes(blk << 6, m0, m1, valid);
            peq[0] = ~m1 & ~m0 & valid; peq[1] = ~m1 & m0 & valid; peq[2] = m1 & ~m0 & valid; peq[3] = m1 & m0 & valid;
        }
        uint64_t Pv = ~0ull, Mv = 0ull;
        int h_prev = 0;                       // horizontal delta out of the last column done: bit 0 = +1, bit 1 = -1
        int bottom = (blk + 1) << 6;          // D[64 blk + 63][-1]
        const int s_lo = jlo + lane, s_hi = some ? jhi + lane : -1, s_up = jhi_up + lane;
        const bool top = lane == 0 && pb == 0;
        const bool feeds = lane == nb_pass - 1 && !is_last_blk;
        for (int s0 = s_begin; s0 <= s_last; s0 += MY_TCHUNK) {
            __builtin_amdgcn_wave_barrier();
            for (int x = lane; x < MY_TCHUNK + 64; x += 64) {
                const int col = s0 - 63 + x;
                tbuf[x] = (col >= 0 && col < tn) ? (uint8_t)t.at(col) : (uint8_t)0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const int s_end = (s0 + MY_TCHUNK) <= s_last ? (s0 + MY_TCHUNK) : s_last + 1;
            const uint8_t* trow = tbuf + 63 - lane - s0;
            for (int s = s0; s < s_end; ++s) {
                int h_up = wave_shr1(h_prev);
                int bot_up = wave_shr1(bottom);
                if (s >= s_lo && s <= s_hi) {
                    const int j = s - lane;
                    const bool up_in = s <= s_up;      // the block above was computed in this column (it never starts later than this one)
                    if (lane == 0) {
                        if (pb == 0) h_up = mode == 2 ? 0 : 1;
                        else if (up_in) { h_up = (int)hb[j]; bot_up = hbot[j]; }
                    }
                    const int h = (up_in || top) ? h_up : 1;
                    if (s == s_lo && jlo_u > 0) {          // the block enters the band: +1 per row below the block above's bottom of the column before
                        Pv = ~0ull; Mv = 0ull;
                        bottom = bot_up - (h_up & 1) + (h_up >> 1) + 64;
                        if (is_last_blk) score = bottom - (63 - last_row);
                    }
                    const int sym = trow[s] & 3;
                    uint64_t Eq = peq[sym];
                    const uint64_t Xv = Eq | Mv;
                    Eq |= (uint64_t)(uint32_t)(h >> 1);
                    const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                    uint64_t Ph = Mv | ~(Xh | Pv);
                    uint64_t Mh = Pv & Xh;
                    const int hout = (int)(Ph >> 63) | ((int)(Mh >> 62) & 2);
                    bottom += (hout & 1) - (hout >> 1);
                    if (is_last_blk) {
                        score += (int)((Ph >> last_row) & 1ull) - (int)((Mh >> last_row) & 1ull);
                        if (mode != 0) {
                            if (score < best) { best = score; best_first = j; best_last = j; }
                            else if (score == best) best_last = j;
                        }
                        if (j == tn - 1) reached_end = true;
                    }
                    Ph = (Ph << 1) | (uint64_t)(uint32_t)(h & 1);
                    Mh = (Mh << 1) | (uint64_t)(uint32_t)(h >> 1);
                    Pv = Mh | ~(Xv | Ph);
                    Mv = Ph & Xv;
                    h_prev = hout;
                    if (feeds) { hb[j] = (int8_t)hout; hbot[j] = bottom; }
                    if (store) {
                        unsigned long long* o = store + ((int64_t)j * nblocks + blk) * 3;
                        o[0] = Pv; o[1] = Mv; o[2] = (unsigned long long)(long long)bottom;
                    }
                }
            }
        }
        if (col_scores && bvalid) {      // rows 64 blk + 63 .. 64 blk of the last column, downwards differences undone
            const bool at_end = some && jhi == tn - 1;
            int sc_row = bottom;
            const int rbase = blk << 6;
            for (int k = 63; k >= 0; --k) {
                if (rbase + k < qn) col_scores[rbase + k] = at_end ? sc_row : MY_INF;
                sc_row += (int)((Mv >> k) & 1ull) - (int)((Pv >> k) & 1ull);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        __builtin_amdgcn_wave_barrier();
    }
    const int owner = (nblocks - 1) & 63;
    out_score = __shfl(reached_end ? score : MY_INF, owner, 64); out_best = __shfl(best, owner, 64);
    out_first = __shfl(best_first, owner, 64); out_last = __shfl(best_last, owner, 64);
}

// exact score of cell (row, col) of the NW matrix from the stored words; boundaries as edlib's (:980-984)
static __device__ __forceinline__ int myers_cell(const unsigned long long* __restrict__ store, int nblocks, MyersBand band, int row, int col) {
    if (row < 0) return col + 1;
    if (col < 0) return row + 1;
    if (!band.holds(row >> 6, col)) return MY_INF;      // (not computed: no alignment of the score asked for passes here)
    const unsigned long long* o = store + ((int64_t)col * nblocks + (row >> 6)) * 3;
    const int k = row & 63;
    const unsigned long long above = k == 63 ? 0ull : (~0ull << (k + 1));      // the rows of the block below this one
    return (int)(long long)o[2] - __popcll(o[0] & above) + __popcll(o[1] & above);
}

// lane 0: the traceback of edlib.cpp:947-1140 over the stored words of one NW matrix (query rows x an columns, final score sc):
// up (insertion) before left (deletion) before the diagonal, on exact cell scores; the moves are written in alignment order.
static __device__ int myers_traceback(const unsigned long long* __restrict__ sto, int nblocks, MyersBand band, int qn, int an, int sc, uint8_t* __restrict__ op) {
    int row = qn - 1, col = an - 1, cur = sc, n = 0;
    while (true) {
        const int u = myers_cell(sto, nblocks, band, row - 1, col);           // (the three neighbours are requested together)
        const int l = myers_cell(sto, nblocks, band, row, col - 1);
        const int ul = (row == 0 && col == 0) ? 0 : myers_cell(sto, nblocks, band, row - 1, col - 1);
        if (u + 1 == cur) {                                   // up: insertion (:1022-1055)
            op[n++] = 1; cur = u; row--;
            if (row < 0) { for (int i = 0; i < col + 1; ++i) op[n++] = 2; break; }
            continue;
        }
        if (l + 1 == cur) {                                   // left: deletion (:1057-1087)
            op[n++] = 2; cur = l; col--;
            if (col < 0) { for (int i = 0; i < row + 1; ++i) op[n++] = 1; break; }
            continue;
        }
        op[n++] = ul == cur ? 0 : 3;                          // diagonal: match / mismatch (:1089-1134)
        cur = ul; row--; col--;
        if (col < 0) { for (int i = 0; i < row + 1; ++i) op[n++] = 1; break; }
        if (row < 0) { for (int i = 0; i < col + 1; ++i) op[n++] = 2; break; }
    }
    for (int i = 0; i < n / 2; ++i) { const uint8_t x = op[i]; op[i] = op[n - 1 - i]; op[n - 1 - i] = x; }
    return n;
}

// edlib keeps the whole matrix of an alignment only below this size and cuts it in halves otherwise (edlib.cpp:1192-1196:
// (2 words + 1 int) per (block, column) + 2 ints per column < 1 MB): which of the optimal alignments comes out depends on it.
static __device__ __forceinline__ bool myers_leaf(int qn, int tn) {
    const long long nb = (qn + 63) >> 6;
    return 20ll * nb * tn + 8ll * tn < 1024ll * 1024ll;
}
#define MY_LEAF_CELLS 52428      /* blocks x columns of the largest matrix that passes myers_leaf */
#define MY_MAX_QUERY (1 << 20)   /* (a matrix that does not pass has >= 4 columns up to this query length) */

#ifndef MY_WAVES_PER_EU
#define MY_WAVES_PER_EU 1
#endif
__global__ __launch_bounds__(64, MY_WAVES_PER_EU) void k_myers_hw_path(
    const uint8_t* __restrict__ query, const int64_t* __restrict__ query_off, const uint8_t* __restrict__ target,
    const int64_t* __restrict__ target_off, const int32_t* __restrict__ pair_ids, int n_list, int8_t* __restrict__ hscratch,
    const int64_t* __restrict__ hscratch_off, unsigned long long* __restrict__ store, const int64_t* __restrict__ store_off,
    int32_t* __restrict__ col_scratch, int want_path,
    int32_t* __restrict__ dist, int32_t* __restrict__ start_loc, int32_t* __restrict__ end_loc,
    uint8_t* __restrict__ ops, const int64_t* __restrict__ ops_off, int32_t* __restrict__ ops_len) {
    __shared__ uint8_t tbuf[MY_TCHUNK + 64];
    __shared__ int s_stack[40][5];
    const int lane = lane_id();
    if ((int)blockIdx.x >= n_list) return;
    const int pr = pair_ids[blockIdx.x];      // (the pairs the grouped kernel below does not take)
    const uint8_t* qp = query + query_off[pr];
    const int qn = (int)(query_off[pr + 1] - query_off[pr]);
    const uint8_t* tp = target + target_off[pr];
    const int tn = (int)(target_off[pr + 1] - target_off[pr]);
    int8_t* hb = hscratch + hscratch_off[pr];                                   // [tn + 64] bytes, then [tn + 64] ints
    int32_t* hbot = reinterpret_cast<int32_t*>(hb + ((tn + 64 + 3) & ~3));
    uint8_t* op = ops ? ops + ops_off[pr] : nullptr;
    if (qn == 0 || tn == 0) {      // edlib.cpp:174-191: distance = query length, end location -1, no start location / path
        if (lane == 0) { dist[pr] = qn; end_loc[pr] = -1; start_loc[pr] = -1; if (ops_len) ops_len[pr] = 0; }
        return;
    }
    int sc, best, first, last;
    // 1. HW: distance and first end location. The band needs a bound on the distance: 1/16 of the query length first; when the
    //    best score found lies above the bound it is the score of an alignment that exists (every score in the band is one), so
    //    the optimum is at most that and the second sweep, with that bound, holds it (edlibAlign doubles its bound from 64
    //    instead, :194-214; the answer does not depend on the bounds tried: the first one that holds the optimum returns it
    //    exactly).
    for (int k = max(64, qn >> 4);;) {
        const bool all = k >= qn;
        myers_sweep(MyersSeq{qp, qn, false}, MyersSeq{tp, tn, false}, 2, all ? MyersBand::whole() : MyersBand::infix(qn, tn, k), hb, hbot, tbuf, nullptr, nullptr,
                    sc, best, first, last);
        if (all || best <= k) break;
        k = best;
    }
    const int d = best, e = first;
    if (e < 0) {   // the whole query before the target (:233-246): start location 0, the alignment over an empty target is all insertions (:1171-1178)
        if (lane == 0) { dist[pr] = d; end_loc[pr] = -1; start_loc[pr] = 0; }
        if (want_path && op) { for (int i = lane; i < qn; i += 64) op[i] = 1; if (lane == 0) ops_len[pr] = qn; }
        return;
    }
    // 2. start location: reversed query against the reversed target prefix [0, e], last best column
    myers_sweep(MyersSeq{qp, qn, true}, MyersSeq{tp, e + 1, true}, 1, MyersBand::prefix(d), hb, hbot, tbuf, nullptr, nullptr, sc, best, first, last);
    const int st = e - last;
    if (lane == 0) { dist[pr] = d; end_loc[pr] = e; start_loc[pr] = st; }
    if (!want_path || !ops) return;
    // 3. the alignment of the query with target[st .. e] (obtainAlignment, :1166-1219), depth first with the upper-left part before
    //    the lower-right one, so that the moves come out in order. A frame: {query begin, query length, target begin, target length, score}.
    if (qn > MY_MAX_QUERY) { if (lane == 0) ops_len[pr] = -1; return; }
    unsigned long long* sto = store + store_off[pr];
    int32_t* left = col_scratch + 2 * (query_off[pr] - query_off[0]);      // [qn] last column of the left half, [qn] of the reversed right half
    int32_t* right_rev = left + qn;
    int sp = 0, n_out = 0;
    if (lane == 0) { s_stack[0][0] = 0; s_stack[0][1] = qn; s_stack[0][2] = st; s_stack[0][3] = e - st + 1; s_stack[0][4] = d; }
    sp = 1;
    while (sp > 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        --sp;
        const int fqa = s_stack[sp][0], fqn = s_stack[sp][1], fta = s_stack[sp][2], ftn = s_stack[sp][3], fbest = s_stack[sp][4];
        __builtin_amdgcn_wave_barrier();
        if (fqn == 0 || ftn == 0) {                            // :1173-1180: all deletions / all insertions
            const uint8_t mv = fqn == 0 ? 2 : 1;
            for (int i = lane; i < fqn + ftn; i += 64) op[n_out + i] = mv;
            n_out += fqn + ftn;
            continue;
        }
        if (myers_leaf(fqn, ftn)) {                            // :1196-1209: the whole matrix and the traceback
            const int nblocks = (fqn + 63) >> 6;
            const MyersBand band = MyersBand::global(fqn, ftn, fbest);
            myers_sweep(MyersSeq{qp + fqa, fqn, false}, MyersSeq{tp + fta, ftn, false}, 0, band, hb, hbot, tbuf, sto, nullptr, sc, best, first, last);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            __builtin_amdgcn_wave_barrier();
            int n = 0;
            if (lane == 0) n = myers_traceback(sto, nblocks, band, fqn, ftn, sc, op + n_out);
            n_out += __shfl(n, 0, 64);
            continue;
        }
        // :1236-1404 Hirschberg: the target in halves, the left one forwards and the right one backwards up to the cut, then the
        // FIRST query row whose two scores add up to the optimum (:1322-1333), the two boundary rows after it (:1335-1353)
        const int lw = ftn / 2, rw = ftn - lw;
        const MyersBand band = MyersBand::global(fqn, ftn, fbest);      // (of the whole frame: both halves see the same diagonals, the right one mirrored)
        myers_sweep(MyersSeq{qp + fqa, fqn, false}, MyersSeq{tp + fta, lw, false}, 0, band, hb, hbot, tbuf, nullptr, left, sc, best, first, last);
        myers_sweep(MyersSeq{qp + fqa, fqn, true}, MyersSeq{tp + fta + lw, rw, true}, 0, band, hb, hbot, tbuf, nullptr, right_rev, sc, best, first, last);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        __builtin_amdgcn_wave_barrier();
        // left[i]: query[0..i] against the left half; right[i] = right_rev[fqn - 1 - i]: query[i..] against the right half
        int cut = -2, ls = 0, rs = 0;
        for (int base = 0; base < fqn - 1; base += 64) {
            const int i = base + lane;
            int a = 0, b = 0;
            const bool in = i < fqn - 1;
            if (in) { a = left[i]; b = right_rev[fqn - 2 - i]; }
            const unsigned long long hit = __ballot(in && a + b == fbest);
            if (hit) {
                const int l0 = __builtin_ctzll(hit);
                cut = base + l0; ls = __shfl(a, l0, 64); rs = __shfl(b, l0, 64);
                break;
            }
        }
        if (cut == -2) {
            const int r0 = right_rev[fqn - 1], l_last = left[fqn - 1];
            if (lw + r0 == fbest) { cut = -1; ls = lw; rs = r0; }
            else if (l_last + rw == fbest) { cut = fqn - 1; ls = l_last; rs = rw; }
            else { if (lane == 0) ops_len[pr] = -1; return; }      // (edlib: EDLIB_STATUS_ERROR -- the score handed down was not the optimum)
        }
        const int ulh = cut + 1;
        if (lane == 0) {
            s_stack[sp][0] = fqa + ulh; s_stack[sp][1] = fqn - ulh; s_stack[sp][2] = fta + lw; s_stack[sp][3] = rw; s_stack[sp][4] = rs;
            s_stack[sp + 1][0] = fqa; s_stack[sp + 1][1] = ulh; s_stack[sp + 1][2] = fta; s_stack[sp + 1][3] = lw; s_stack[sp + 1][4] = ls;
        }
        sp += 2;
    }
    if (lane == 0) ops_len[pr] = n_out;
}

// ---- short queries: G lanes per pair, 64 / G pairs per wavefront ----------------------------------------------------------
// The stage-5 call sites align 200-300 bases (five 64-row blocks): with a wavefront per pair five lanes of 64 work. Here a pair
// takes G = 8 / 16 / 32 lanes (queries up to 64 G bases whose matrix edlib keeps whole: one pass, one leaf, no cuts), every
// per-pair quantity lives in the lanes of its group, the loops run to the longest pair of the wavefront and a group that is
// done idles under the exec mask. Same sweeps, same band, same traceback as above.
#define MY_GCHUNK 1024
template <int G>
static __device__ void myers_sweep_grouped(bool active, const MyersSeq& q, const MyersSeq& t, int mode, MyersBand band, uint8_t* __restrict__ tb /* this group's LDS [MY_GCHUNK + 64] */,
                                           unsigned long long* __restrict__ store, int& out_score, int& out_best, int& out_first, int& out_last) {
    const int lane = lane_id(), gl = lane & (G - 1);
    const int qn = q.n, tn = t.n;
    const int nblocks = (qn + 63) >> 6;
    const int last_row = (qn - 1) & 63;
    int score = qn, best = (qn & 63) == 0 ? qn + 1 : qn, best_first = -1, best_last = -1;
    bool reached_end = false;
    const int blk = gl;
    const bool is_last_blk = blk == nblocks - 1;
    const int jlo_u = (blk << 6) + band.lo, jhi_u = (blk << 6) + 63 + band.hi;
    const int jlo = jlo_u > 0 ? jlo_u : 0, jhi = jhi_u < tn - 1 ? jhi_u : tn - 1;
    const int jhi_up = (jhi_u - 64) < tn - 1 ? (jhi_u - 64) : tn - 1;
    const bool some = active && blk < nblocks && jlo <= jhi;
    const int s_lo = jlo + gl, s_hi = some ? jhi + gl : -1, s_up = jhi_up + gl;
    int s_begin = some ? s_lo : 0x3fffffff, s_last = s_hi;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) { s_begin = min(s_begin, __shfl_xor(s_begin, o, 64)); s_last = max(s_last, __shfl_xor(s_last, o, 64)); }
    int n_steps = s_last - s_begin + 1;          // of this group; the loop runs to the longest of the wavefront
    if (n_steps < 0) n_steps = 0;
    int n_max = n_steps;
#pragma unroll
    for (int o = 32; o >= G; o >>= 1) n_max = max(n_max, __shfl_xor(n_max, o, 64));
    uint64_t peq[4] = {0, 0, 0, 0};
    if (some) {
        uint64_t m0, m1, valid;
        q.planes(blk << 6, m0, m1, valid);
        peq[0] = ~m1 & ~m0 & valid; peq[1] = ~m1 & m0 & valid; peq[2] = m1 & ~m0 & valid; peq[3] = m1 & m0 & valid;
    }
    uint64_t Pv = ~0ull, Mv = 0ull;
    int h_prev = 0, bottom = (blk + 1) << 6;
    const bool top = gl == 0;
    for (int i0 = 0; i0 < n_max; i0 += MY_GCHUNK) {
        __builtin_amdgcn_wave_barrier();
        // columns s_begin + i0 - (G - 1) ... of this group's target, four per lane and load
        if (n_steps > i0) {
            const int c0 = s_begin + i0 - (G - 1);
            for (int x = 4 * gl; x < MY_GCHUNK + G; x += 4 * G) {
                const int c = c0 + x;
                uint32_t w = 0u;
                if (c >= 0 && c + 3 < tn) {
                    w = t.rev ? __builtin_bswap32(*reinterpret_cast<const u32_unaligned*>(t.p + (tn - 4 - c))) : *reinterpret_cast<const u32_unaligned*>(t.p + c);
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b) if (c + b >= 0 && c + b < tn) w |= (uint32_t)t.at(c + b) << (8 * b);
                }
                *reinterpret_cast<uint32_t*>(tb + x) = w;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const int i_end = (i0 + MY_GCHUNK) < n_max ? (i0 + MY_GCHUNK) : n_max;
        const uint8_t* trow = tb + (G - 1) - gl - i0;
        for (int i = i0; i < i_end; ++i) {
            const int s = s_begin + i;
            int h_up = wave_shr1(h_prev);
            const int bot_up = wave_shr1(bottom);
            if (s >= s_lo && s <= s_hi) {
                const int j = s - gl;
                const bool up_in = s <= s_up;
                if (top) h_up = mode == 2 ? 0 : 1;
                const int h = (up_in || top) ? h_up : 1;
                if (s == s_lo && jlo_u > 0) {
                    Pv = ~0ull; Mv = 0ull;
                    bottom = bot_up - (h_up & 1) + (h_up >> 1) + 64;
                    if (is_last_blk) score = bottom - (63 - last_row);
                }
                const int sym = trow[i] & 3;
                uint64_t Eq = peq[sym];
                const uint64_t Xv = Eq | Mv;
                Eq |= (uint64_t)(uint32_t)(h >> 1);
                const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                uint64_t Ph = Mv | ~(Xh | Pv);
                uint64_t Mh = Pv & Xh;
                const int hout = (int)(Ph >> 63) | ((int)(Mh >> 62) & 2);
                bottom += (hout & 1) - (hout >> 1);
                if (is_last_blk) {
                    score += (int)((Ph >> last_row) & 1ull) - (int)((Mh >> last_row) & 1ull);
                    if (mode != 0) {
                        if (score < best) { best = score; best_first = j; best_last = j; }
                        else if (score == best) best_last = j;
                    }
                    if (j == tn - 1) reached_end = true;
                }
                Ph = (Ph << 1) | (uint64_t)(uint32_t)(h & 1);
                Mh = (Mh << 1) | (uint64_t)(uint32_t)(h >> 1);
                Pv = Mh | ~(Xv | Ph);
                Mv = Ph & Xv;
                h_prev = hout;
                if (store) {
                    unsigned long long* o = store + ((int64_t)j * nblocks + blk) * 3;
                    o[0] = Pv; o[1] = Mv; o[2] = (unsigned long long)(long long)bottom;
                }
            }
        }
    }
    const int owner = (lane & ~(G - 1)) + ((nblocks - 1) & (G - 1));
    out_score = __shfl(reached_end ? score : MY_INF, owner, 64); out_best = __shfl(best, owner, 64);
    out_first = __shfl(best_first, owner, 64); out_last = __shfl(best_last, owner, 64);
}

template <int G>
__global__ __launch_bounds__(64) void k_myers_hw_path_grouped(
    const uint8_t* __restrict__ query, const int64_t* __restrict__ query_off, const uint8_t* __restrict__ target,
    const int64_t* __restrict__ target_off, const int32_t* __restrict__ pair_ids, int n_list,
    unsigned long long* __restrict__ store, const int64_t* __restrict__ store_off, int want_path,
    int32_t* __restrict__ dist, int32_t* __restrict__ start_loc, int32_t* __restrict__ end_loc,
    uint8_t* __restrict__ ops, const int64_t* __restrict__ ops_off, int32_t* __restrict__ ops_len) {
    constexpr int NG = 64 / G;
    __shared__ __attribute__((aligned(16))) uint8_t tbuf[NG][MY_GCHUNK + 64];
    const int lane = lane_id(), gl = lane & (G - 1), grp = lane / G;
    const int slot = (int)blockIdx.x * NG + grp;
    const bool live = slot < n_list;
    const int pr = live ? pair_ids[slot] : 0;
    const uint8_t* qp = query; const uint8_t* tp = target;
    int qn = 0, tn = 0;
    if (live) { qp += query_off[pr]; qn = (int)(query_off[pr + 1] - query_off[pr]); tp += target_off[pr]; tn = (int)(target_off[pr + 1] - target_off[pr]); }
    uint8_t* op = (ops && live) ? ops + ops_off[pr] : nullptr;
    uint8_t* tb = tbuf[grp];
    bool act = live && qn > 0 && tn > 0;
    if (live && !act && gl == 0) { dist[pr] = qn; end_loc[pr] = -1; start_loc[pr] = -1; if (ops_len) ops_len[pr] = 0; }      // edlib.cpp:174-191
    int sc, best, first, last;
    // 1. HW with a doubled bound (every group at its own)
    int k = max(64, qn >> 4), d = 0, e = -1;
    bool pending = act;
    while (__ballot(pending) != 0ull) {
        const bool all = k >= qn;
        myers_sweep_grouped<G>(pending, MyersSeq{qp, qn, false}, MyersSeq{tp, tn, false}, 2, all ? MyersBand::whole() : MyersBand::infix(qn, tn, k), tb, nullptr, sc, best, first, last);
        if (pending) { if (all || best <= k) { pending = false; d = best; e = first; } else k = best; }      // (see k_myers_hw_path)
    }
    if (act && e < 0) {      // the whole query before the target (:233-246)
        if (gl == 0) { dist[pr] = d; end_loc[pr] = -1; start_loc[pr] = 0; }
        if (want_path && op) { for (int i = gl; i < qn; i += G) op[i] = 1; if (gl == 0) ops_len[pr] = qn; }
        act = false;
    }
    // 2. start location
    myers_sweep_grouped<G>(act, MyersSeq{qp, qn, true}, MyersSeq{tp, e + 1, true}, 1, MyersBand::prefix(d), tb, nullptr, sc, best, first, last);
    const int st = e - last;
    if (act && gl == 0) { dist[pr] = d; end_loc[pr] = e; start_loc[pr] = st; }
    if (!want_path || !ops) return;
    // 3. one leaf: the matrix whole, the traceback by the group's first lane
    const int an = e - st + 1;
    const MyersBand band = MyersBand::global(qn, an, d);
    unsigned long long* sto = store + (live ? store_off[pr] : 0);
    myers_sweep_grouped<G>(act, MyersSeq{qp, qn, false}, MyersSeq{tp + st, an, false}, 0, band, tb, sto, sc, best, first, last);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    __builtin_amdgcn_wave_barrier();
    if (act && gl == 0) ops_len[pr] = myers_traceback(sto, (qn + 63) >> 6, band, qn, an, sc, op);
}


// ---- distance and end location only (hs_edit_distance: edlib's TASK_DISTANCE / TASK_LOC end, modes NW / SHW / HW) ------------
// The same banded sweeps. The bound: 1/16 of the query length (+ the length difference for NW) first; a score above the bound
// is the score of an alignment that exists, so the sweep with that score as its bound holds the optimum.
static __device__ __forceinline__ MyersBand myers_mode_band(int mode, int qn, int tn, int k, bool all) {
    if (all) return MyersBand::whole();
    return mode == 0 ? MyersBand::global(qn, tn, k) : mode == 1 ? MyersBand::prefix(k) : MyersBand::infix(qn, tn, k);
}
static __device__ __forceinline__ int myers_first_bound(int mode, int qn, int tn) {
    const int D = tn > qn ? tn - qn : qn - tn;
    return max(64, qn >> 4) + (mode == 0 ? D : 0);
}

__global__ __launch_bounds__(64) void k_myers_distance(
    const uint8_t* __restrict__ query, const int64_t* __restrict__ query_off, const uint8_t* __restrict__ target,
    const int64_t* __restrict__ target_off, const int32_t* __restrict__ pair_ids, int n_list, int mode, int8_t* __restrict__ hscratch,
    const int64_t* __restrict__ hscratch_off, int32_t* __restrict__ dist, int32_t* __restrict__ end_loc) {
    __shared__ uint8_t tbuf[MY_TCHUNK + 64];
    const int lane = lane_id();
    if ((int)blockIdx.x >= n_list) return;
    const int pr = pair_ids[blockIdx.x];
    const uint8_t* qp = query + query_off[pr];
    const int qn = (int)(query_off[pr + 1] - query_off[pr]);
    const uint8_t* tp = target + target_off[pr];
    const int tn = (int)(target_off[pr + 1] - target_off[pr]);
    if (qn == 0 || tn == 0) {      // nothing to sweep: all deletions / all insertions (NW), an empty placement otherwise
        if (lane == 0) { dist[pr] = mode == 0 ? (qn ? qn : tn) : qn; end_loc[pr] = (mode == 0 && qn == 0) ? tn - 1 : -1; }
        return;
    }
    int8_t* hb = hscratch + hscratch_off[pr];
    int32_t* hbot = reinterpret_cast<int32_t*>(hb + ((tn + 64 + 3) & ~3));
    const int kmax = mode == 0 ? max(qn, tn) : qn;
    int sc, best, first, last;
    for (int k = myers_first_bound(mode, qn, tn);;) {
        const bool all = k >= kmax;
        myers_sweep(MyersSeq{qp, qn, false}, MyersSeq{tp, tn, false}, mode, myers_mode_band(mode, qn, tn, k, all), hb, hbot, tbuf, nullptr, nullptr, sc, best, first, last);
        const int got = mode == 0 ? sc : best;
        if (all || got <= k) break;
        k = got >= MY_INF ? 2 * k : got;
    }
    if (lane == 0) {
        if (mode == 0) { dist[pr] = sc; end_loc[pr] = tn - 1; }
        else { dist[pr] = best; end_loc[pr] = first; }
    }
}

template <int G>
__global__ __launch_bounds__(64) void k_myers_distance_grouped(
    const uint8_t* __restrict__ query, const int64_t* __restrict__ query_off, const uint8_t* __restrict__ target,
    const int64_t* __restrict__ target_off, const int32_t* __restrict__ pair_ids, int n_list, int mode,
    int32_t* __restrict__ dist, int32_t* __restrict__ end_loc) {
    constexpr int NG = 64 / G;
    __shared__ __attribute__((aligned(16))) uint8_t tbuf[NG][MY_GCHUNK + 64];
    const int lane = lane_id(), gl = lane & (G - 1), grp = lane / G;
    const int slot = (int)blockIdx.x * NG + grp;
    const bool live = slot < n_list;
    const int pr = live ? pair_ids[slot] : 0;
    const uint8_t* qp = query; const uint8_t* tp = target;
    int qn = 0, tn = 0;
    if (live) { qp += query_off[pr]; qn = (int)(query_off[pr + 1] - query_off[pr]); tp += target_off[pr]; tn = (int)(target_off[pr + 1] - target_off[pr]); }
    const bool act = live && qn > 0 && tn > 0;
    if (live && !act && gl == 0) { dist[pr] = mode == 0 ? (qn ? qn : tn) : qn; end_loc[pr] = (mode == 0 && qn == 0) ? tn - 1 : -1; }
    const int kmax = mode == 0 ? max(qn, tn) : qn;
    int sc = 0, best = 0, first = -1, last = -1, r_sc = 0, r_best = 0, r_first = -1;
    int k = myers_first_bound(mode, qn, tn);
    bool pending = act;
    while (__ballot(pending) != 0ull) {
        const bool all = k >= kmax;
        myers_sweep_grouped<G>(pending, MyersSeq{qp, qn, false}, MyersSeq{tp, tn, false}, mode, myers_mode_band(mode, qn, tn, k, all), tbuf[grp], nullptr, sc, best, first, last);
        if (pending) {
            const int got = mode == 0 ? sc : best;
            if (all || got <= k) { pending = false; r_sc = sc; r_best = best; r_first = first; }
            else k = got >= MY_INF ? 2 * k : got;
        }
    }
    if (act && gl == 0) {
        if (mode == 0) { dist[pr] = r_sc; end_loc[pr] = tn - 1; }
        else { dist[pr] = r_best; end_loc[pr] = r_first; }
    }
}


}  // namespace hsdev
