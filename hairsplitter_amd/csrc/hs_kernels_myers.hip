// hs_kernels_myers.hip -- A1 for the stage-5 call sites: what the reference asks of its bundled edlib
// there is edlibAlign(query, target, k = -1, EDLIB_MODE_HW, EDLIB_TASK_PATH) (create_new_contigs.cpp:558-629,
// tools.cpp:515-534) -- edit distance, first end location, its start location and one optimal alignment -- for a 200-300 bp
// query against a target of a few hundred to a few thousand bases there; any query up to 2^20 bases here. One wavefront per
// pair, Myers sweeps (lane b owns query block b, 64 blocks per pass, anti-diagonal schedule as in k_myers):
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
//      edlib computes these matrices inside a band around the diagonal; every score it reads there is the exact one, so
//      whole exact columns give the same decisions (oracle/edlib_path_oracle.py makes the same argument on numpy columns and
//      is pinned against the reference's edlib up to 60 kb).
// Alignment ops as edlib's: 0 match, 1 insertion (query base without target base), 2 deletion, 3 mismatch.
// Sequences are 2-bit base codes (A C G T), as everywhere on this path. Included by hs_capi.hip after hs_kernels.hip.
#pragma once

namespace hsdev {

struct MyersSeq {              // a sequence seen forwards or backwards
    const uint8_t* p; int n; bool rev;
    __device__ __forceinline__ int at(int i) const { return (int)(p[rev ? n - 1 - i : i] & 3); }
};

// One sweep. mode 0 NW, 1 SHW, 2 HW. Outputs through references (valid in every lane): final bottom-row score of the
// last column, best bottom-row score over the columns, first and last column attaining it (-1: before the target).
// store != nullptr: P, M, bottom score of every (column, block) at store[(col * nblocks + blk) * 3 ...] as three 64-bit words
// {P, M, score}. col_scores != nullptr: the scores of the LAST column, one int per query row (what Hirschberg's split reads).
static __device__ void myers_sweep(const MyersSeq& q, const MyersSeq& t, int mode, int8_t* __restrict__ hb, uint8_t* tbuf /* LDS [MY_TCHUNK + 64] */,
                                   unsigned long long* __restrict__ store, int32_t* __restrict__ col_scores, int& out_score, int& out_best, int& out_first,
                                   int& out_last) {
    const int lane = lane_id();
    const int qn = q.n, tn = t.n;
    const int nblocks = (qn + 63) >> 6;
    const int last_row = (qn - 1) & 63;
    // The reference's edlib pads the query to a multiple of 64 rows and reads the score of column c off column c + W (W = padding
    // rows, edlib.cpp:664-690): with W > 0 the columns "before the target" (score = query length) take part and win ties, which is
    // what best = qn, first = -1 reproduces; with W == 0 there are none, and the first real column that reaches the best score --
    // query length included -- is the answer (64 x 'A' in 'CCC...': end location 0, path 1X63I).
    int score = qn, best = (qn & 63) == 0 ? qn + 1 : qn, best_first = -1, best_last = -1;
    for (int pb = 0; pb < nblocks; pb += 64) {
        const int blk = pb + lane;
        const bool bvalid = blk < nblocks;
        const bool is_last_blk = blk == nblocks - 1;
        const int nb_pass = (nblocks - pb) < 64 ? (nblocks - pb) : 64;
        uint64_t peq[4] = {0, 0, 0, 0};
        if (bvalid) {
            const int rbase = blk << 6;
            for (int k = 0; k < 64; ++k) {
                const int row = rbase + k;
                if (row < qn) peq[q.at(row)] |= 1ull << k;
            }
        }
        uint64_t Pv = ~0ull, Mv = 0ull;
        int hout_prev = 0;
        int bottom = (blk + 1) << 6;          // D[64 blk + 63][-1]
        const int nsteps = tn + nb_pass - 1;
        for (int s0 = 0; s0 < nsteps; s0 += MY_TCHUNK) {
            __builtin_amdgcn_wave_barrier();
            for (int x = lane; x < MY_TCHUNK + 64; x += 64) {
                const int col = s0 - 63 + x;
                tbuf[x] = (col >= 0 && col < tn) ? (uint8_t)t.at(col) : (uint8_t)0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const int s_end = (s0 + MY_TCHUNK) < nsteps ? (s0 + MY_TCHUNK) : nsteps;
            for (int s = s0; s < s_end; ++s) {
                const int j = s - lane;
                const int hin_up = __shfl_up(hout_prev, 1, 64);
                const bool work = bvalid && j >= 0 && j < tn;
                int hin;
                if (lane == 0) hin = pb == 0 ? (mode == 2 ? 0 : 1) : (work ? (int)hb[j] : 0);
                else hin = hin_up;
                if (work) {
                    const int sym = tbuf[(s - s0) + 63 - lane] & 3;
                    uint64_t Eq = peq[sym];
                    const uint64_t Xv = Eq | Mv;
                    if (hin < 0) Eq |= 1ull;
                    const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                    uint64_t Ph = Mv | ~(Xh | Pv);
                    uint64_t Mh = Pv & Xh;
                    int hout = 0;
                    if (Ph >> 63) hout = 1; else if (Mh >> 63) hout = -1;
                    bottom += hout;
                    if (is_last_blk) {
                        score += (int)((Ph >> last_row) & 1ull) - (int)((Mh >> last_row) & 1ull);
                        if (mode != 0) {
                            if (score < best) { best = score; best_first = j; best_last = j; }
                            else if (score == best) best_last = j;
                        }
                    }
                    Ph <<= 1; Mh <<= 1;
                    if (hin < 0) Mh |= 1ull; else if (hin > 0) Ph |= 1ull;
                    Pv = Mh | ~(Xv | Ph);
                    Mv = Ph & Xv;
                    hout_prev = hout;
                    if (lane == nb_pass - 1 && !is_last_blk) hb[j] = (int8_t)hout;
                    if (store) {
                        unsigned long long* o = store + ((int64_t)j * nblocks + blk) * 3;
                        o[0] = Pv; o[1] = Mv; o[2] = (unsigned long long)(long long)bottom;
                    }
                }
            }
        }
        if (col_scores && bvalid) {      // rows 64 blk + 63 .. 64 blk of the last column, downwards differences undone
            int sc_row = bottom;
            const int rbase = blk << 6;
            for (int k = 63; k >= 0; --k) {
                if (rbase + k < qn) col_scores[rbase + k] = sc_row;
                sc_row += (int)((Mv >> k) & 1ull) - (int)((Pv >> k) & 1ull);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        __builtin_amdgcn_wave_barrier();
    }
    const int owner = (nblocks - 1) & 63;
    out_score = __shfl(score, owner, 64); out_best = __shfl(best, owner, 64);
    out_first = __shfl(best_first, owner, 64); out_last = __shfl(best_last, owner, 64);
}

// exact score of cell (row, col) of the NW matrix from the stored words; boundaries as edlib's (:980-984)
static __device__ __forceinline__ int myers_cell(const unsigned long long* __restrict__ store, int nblocks, int row, int col) {
    if (row < 0) return col + 1;
    if (col < 0) return row + 1;
    const unsigned long long* o = store + ((int64_t)col * nblocks + (row >> 6)) * 3;
    const int k = row & 63;
    const unsigned long long above = k == 63 ? 0ull : (~0ull << (k + 1));      // the rows of the block below this one
    return (int)(long long)o[2] - __popcll(o[0] & above) + __popcll(o[1] & above);
}

// lane 0: the traceback of edlib.cpp:947-1140 over the stored words of one NW matrix (query rows x an columns, final score sc):
// up (insertion) before left (deletion) before the diagonal, on exact cell scores; the moves are written in alignment order.
static __device__ int myers_traceback(const unsigned long long* __restrict__ sto, int nblocks, int qn, int an, int sc, uint8_t* __restrict__ op) {
    int row = qn - 1, col = an - 1, cur = sc, n = 0;
    while (true) {
        const int u = myers_cell(sto, nblocks, row - 1, col);           // (the three neighbours are requested together)
        const int l = myers_cell(sto, nblocks, row, col - 1);
        const int ul = (row == 0 && col == 0) ? 0 : myers_cell(sto, nblocks, row - 1, col - 1);
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

__global__ __launch_bounds__(64) void k_myers_hw_path(
    const uint8_t* __restrict__ query, const int64_t* __restrict__ query_off, const uint8_t* __restrict__ target,
    const int64_t* __restrict__ target_off, int n_pairs, int8_t* __restrict__ hscratch, const int64_t* __restrict__ hscratch_off,
    unsigned long long* __restrict__ store, const int64_t* __restrict__ store_off, int32_t* __restrict__ col_scratch, int want_path,
    int32_t* __restrict__ dist, int32_t* __restrict__ start_loc, int32_t* __restrict__ end_loc,
    uint8_t* __restrict__ ops, const int64_t* __restrict__ ops_off, int32_t* __restrict__ ops_len) {
    __shared__ uint8_t tbuf[MY_TCHUNK + 64];
    __shared__ int s_stack[40][5];
    const int lane = lane_id();
    const int pr = (int)blockIdx.x;
    if (pr >= n_pairs) return;
    const uint8_t* qp = query + query_off[pr];
    const int qn = (int)(query_off[pr + 1] - query_off[pr]);
    const uint8_t* tp = target + target_off[pr];
    const int tn = (int)(target_off[pr + 1] - target_off[pr]);
    int8_t* hb = hscratch + hscratch_off[pr];
    uint8_t* op = ops ? ops + ops_off[pr] : nullptr;
    if (qn == 0 || tn == 0) {      // edlib.cpp:174-191: distance = query length, end location -1, no start location / path
        if (lane == 0) { dist[pr] = qn; end_loc[pr] = -1; start_loc[pr] = -1; if (ops_len) ops_len[pr] = 0; }
        return;
    }
    int sc, best, first, last;
    // 1. HW: distance and first end location
    myers_sweep(MyersSeq{qp, qn, false}, MyersSeq{tp, tn, false}, 2, hb, tbuf, nullptr, nullptr, sc, best, first, last);
    const int d = best, e = first;
    if (e < 0) {   // the whole query before the target (:233-246): start location 0, the alignment over an empty target is all insertions (:1171-1178)
        if (lane == 0) { dist[pr] = d; end_loc[pr] = -1; start_loc[pr] = 0; }
        if (want_path && op) { for (int i = lane; i < qn; i += 64) op[i] = 1; if (lane == 0) ops_len[pr] = qn; }
        return;
    }
    // 2. start location: reversed query against the reversed target prefix [0, e], last best column
    myers_sweep(MyersSeq{qp, qn, true}, MyersSeq{tp, e + 1, true}, 1, hb, tbuf, nullptr, nullptr, sc, best, first, last);
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
            myers_sweep(MyersSeq{qp + fqa, fqn, false}, MyersSeq{tp + fta, ftn, false}, 0, hb, tbuf, sto, nullptr, sc, best, first, last);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            __builtin_amdgcn_wave_barrier();
            int n = 0;
            if (lane == 0) n = myers_traceback(sto, nblocks, fqn, ftn, sc, op + n_out);
            n_out += __shfl(n, 0, 64);
            continue;
        }
        // :1236-1404 Hirschberg: the target in halves, the left one forwards and the right one backwards up to the cut, then the
        // FIRST query row whose two scores add up to the optimum (:1322-1333), the two boundary rows after it (:1335-1353)
        const int lw = ftn / 2, rw = ftn - lw;
        myers_sweep(MyersSeq{qp + fqa, fqn, false}, MyersSeq{tp + fta, lw, false}, 0, hb, tbuf, nullptr, left, sc, best, first, last);
        myers_sweep(MyersSeq{qp + fqa, fqn, true}, MyersSeq{tp + fta + lw, rw, true}, 0, hb, tbuf, nullptr, right_rev, sc, best, first, last);
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

}  // namespace hsdev
