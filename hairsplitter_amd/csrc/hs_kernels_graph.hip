// hs_kernels_graph.hip -- K6: the read graph of a clustering window (create_read_graph_matrix,
// separate_reads.cpp:706-828) built on the device from the resident sim/diff matrices of K5, so that the two N x N
// int matrices never cross PCIe. Included by hs_capi.hip after hs_kernels.hip (shares its wave helpers).
//
// One wavefront per row (window, masked read r1). The reference sorts the N distances of the row with std::sort and
// walks the result; the set it links only depends on order statistics of the row (largest two values, number of exact
// ones, k-th largest value below one) unless a run of equal distances at the five-neighbour cut-off is only partly
// taken. Those rows -- a few per ten thousand -- are reported back and resolved on the host with std::sort itself.
// Rows live in the window's local index space (position in its ascending list of masked reads): only masked reads
// can be linked (:806-815), reads outside the mask only contribute zeros to the statistics.
#pragma once

namespace hsdev {

// distances live in [0, 1]: their bit patterns order like the values
static __device__ __forceinline__ float wave_max_f01(float v) { return __int_as_float(wave_max_i32(__float_as_int(v))); }

// largest value of the row strictly below `cur` that satisfies the filter, and how often it occurs; -1 if none.
// FILTER 0: value != 1 (every entry of the row, the implicit zeros of unmasked reads included)
// FILTER 1: below < value < above and value != 1 (the cut-off candidates)
template <int FILTER>
static __device__ __forceinline__ void next_level(const float* __restrict__ dv, int m, int lane, float cur, float below, float above,
                                                  int extra_zeros, float* v_out, int* c_out) {
    float best = -1.f;
    for (int j = lane; j < m; j += 64) {
        const float d = dv[j];
        const bool ok = FILTER == 0 ? (d != 1.f) : (d > below && d != 1.f && d < above);
        if (ok && d < cur && d > best) best = d;
    }
    if (FILTER == 0 && extra_zeros > 0 && 0.f < cur && best < 0.f) best = 0.f;
    // "none" (-1) does not order like its bits: reduce on a monotone integer key
    const int key = best < 0.f ? -1 : __float_as_int(best);
    const int kmax = wave_max_i32(key);
    const float vv = kmax < 0 ? -1.f : __int_as_float(kmax);
    int c = 0;
    if (kmax >= 0) {
        for (int j = lane; j < m; j += 64) {
            const float d = dv[j];
            const bool ok = FILTER == 0 ? (d != 1.f) : (d > below && d != 1.f && d < above);
            if (ok && d == vv) c++;
        }
        c = wave_sum_i32(c);
        if (FILTER == 0 && vv == 0.f) c += extra_zeros;
    }
    *v_out = vv; *c_out = c;
}

// LM: the low-memory path (create_read_graph_low_memory, separate_reads.cpp:538-693). sim / diff are then the window-local m x m
// matrices of k_simdiff_windows (window w at win_mat_off[w], indexed by local ids), and the distance is that path's own: every
// other masked read takes part whether the two share a similar SNP or not (no `sim > 0` guard, :618), so 0 / 0 occurs; a row with
// a NaN goes to the host (std::sort with NaNs is the reference's behaviour, not ours to restate with order statistics).
template <bool LM>
__global__ __launch_bounds__(256) void k_read_graph_rows(
    const int32_t* __restrict__ sim, const int32_t* __restrict__ diff, const int64_t* __restrict__ ctg_out_off,
    const int32_t* __restrict__ ctg_n, const int32_t* __restrict__ win_contig, const int64_t* __restrict__ win_mask_off,
    const int32_t* __restrict__ mask_ids, const int32_t* __restrict__ row_win, const int64_t* __restrict__ win_bits_off,
    int row_base, int n_rows, float below, int cap, unsigned long long* __restrict__ bits, unsigned long long* __restrict__ amb_head /* [0] rows left to the host, [1] staged entries */,
    int32_t* __restrict__ amb_rows, int amb_cap, const int64_t* __restrict__ win_mat_off, int es /* element stride of sim / diff: 2 = (sim, diff) pairs */,
    long long* __restrict__ amb_stage_off /* [amb_cap] where the row's entries were staged, -1: no room */, int32_t* __restrict__ stage_sim, int32_t* __restrict__ stage_diff,
    long long stage_cap, const int32_t* __restrict__ pos_rank /* row / column of the contig's matrices that holds a read (the matrices are in the order of
    the reads' start positions: k_simdiff), per contig at rank_off[c]; NULL: the read index itself */, const int64_t* __restrict__ rank_off) {
    extern __shared__ unsigned char s_dyn[];
    const int lane = lane_id();
    const int wv = wave_id(), waves = (int)(blockDim.x >> 6);
    const int row = row_base + (int)blockIdx.x * waves + wv;
    if (row >= row_base + n_rows) return;         // wave-uniform
    float* __restrict__ dv = reinterpret_cast<float*>(s_dyn) + (size_t)wv * 2 * cap;
    int* __restrict__ tv = reinterpret_cast<int*>(dv + cap);
    const int w = row_win[row];
    const int64_t m0 = win_mask_off[w];
    const int m = (int)(win_mask_off[w + 1] - m0);
    const int i = row - (int)m0;
    const int c = win_contig[w];
    const int N = ctg_n[c];
    const int32_t* __restrict__ ids = mask_ids + m0;
    const int r1 = ids[i];
    const int32_t* __restrict__ prk = (!LM && pos_rank) ? pos_rank + rank_off[c] : nullptr;
    const int r1k = prk ? prk[r1] : r1;
    const int32_t* __restrict__ srow = LM ? sim + (win_mat_off[w] + (int64_t)i * m) * es : sim + (ctg_out_off[c] + (int64_t)r1k * N) * es;
    const int32_t* __restrict__ drow = LM ? diff + (win_mat_off[w] + (int64_t)i * m) * es : diff + (ctg_out_off[c] + (int64_t)r1k * N) * es;
    // a row the order statistics cannot decide goes to the host -- with its sim / diff entries (the N of the contig, or the m of a
    // window-local matrix) staged here, so that the host needs ONE transfer to resolve every such row of the call
    auto give_up = [&]() {
        const int len = LM ? m : N;
        long long so = -1;
        if (lane == 0) {
            const long long k = (long long)atomicAdd(&amb_head[0], 1ull);
            if (k < amb_cap) {
                amb_rows[k] = row;
                if (stage_sim) { so = (long long)atomicAdd(&amb_head[1], (unsigned long long)len); if (so + len > stage_cap) so = -1; }
                amb_stage_off[k] = so;
            }
        }
        so = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(so & 0xffffffffll), 0)) |
                         ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)so >> 32), 0) << 32));
        // (staged in the order of the READ indices, whatever the order of the matrix: the host walks the row by read)
        if (so >= 0) for (int j = lane; j < len; j += 64) { const int64_t at = (int64_t)(prk ? prk[j] : j) * es; stage_sim[so + j] = srow[at]; stage_diff[so + j] = drow[at]; }
    };
    if (m > cap || N < 2 || !(below >= 0.f)) { give_up(); return; }

    // distances of the masked reads (:752-759 / :611-624); every other read of the contig has distance 0
    int max_compat = 0;
    bool nan_l = false;
    for (int j = lane; j < m; j += 64) {
        const int r = ids[j];
        const int64_t at = (int64_t)(LM ? j : (prk ? prk[r] : r)) * es;
        const int s = srow[at], dd = drow[at];
        float d = 0.f;
        if (LM) {
            if (r != r1) {
                const float df = (float)(dd - 1 > 0 ? dd - 1 : 0);
                d = 1.f - df / (float)(s + dd);
                if (s > max_compat) max_compat = s;
            }
        } else if (r != r1 && s > 0) {
            const float df = (float)(dd - 1 > 0 ? dd - 1 : 0);
            d = 1.f - df / (float)(s + dd);
            if (s > max_compat) max_compat = s;
        }
        dv[j] = d; tv[j] = r != r1 ? s + dd : 0x7fffffff;
    }
    max_compat = wave_max_i32(max_compat);
    const double thr = 0.7 * (double)max_compat;   // :762-766
    for (int j = lane; j < m; j += 64) {
        if ((double)tv[j] < thr) dv[j] = 0.f;
        if (LM && dv[j] != dv[j]) nan_l = true;
    }
    wave_lds_sync();
    if (LM && __ballot(nan_l) != 0ull) { give_up(); return; }

    const int extra_zeros = N - m;
    // two largest values with multiplicity, number of exact ones
    float s0, s1; int c0, ones = 0;
    {
        float best = 0.f;
        for (int j = lane; j < m; j += 64) { const float d = dv[j]; if (d > best) best = d; if (d == 1.f) ones++; }
        s0 = wave_max_f01(best);
        ones = wave_sum_i32(ones);
        int cnt = 0; float second = -1.f;
        for (int j = lane; j < m; j += 64) { const float d = dv[j]; if (d == s0) cnt++; else if (d > second) second = d; }
        c0 = wave_sum_i32(cnt) + (s0 == 0.f ? extra_zeros : 0);
        if (extra_zeros > 0 && s0 > 0.f && second < 0.f) second = 0.f;
        const int key = second < 0.f ? -1 : __float_as_int(second);
        const int kmax = wave_max_i32(key);
        s1 = c0 >= 2 ? s0 : (kmax < 0 ? -1.f : __int_as_float(kmax));
    }
    float above = s0 - (s0 - s1) * 3;              // :779
    if (above == 1.f && ones < N) {                // :780-794: the value four places after the last exact one
        const int idx = (ones + 4) < (N - 1) ? (ones + 4) : (N - 1);
        int k = idx - ones;
        float cur = 2.f, v = 0.f; int cc = 0;
        for (int it = 0; it < 5; ++it) {
            next_level<0>(dv, m, lane, cur, below, above, extra_zeros, &v, &cc);
            if (cc == 0) { v = 0.f; break; }       // cannot happen (k < N - ones); the minimum of the row is 0
            if (k < cc) break;
            k -= cc; cur = v;
        }
        above = v;
    }
    // entries taken whatever the neighbour count: exact ones and everything at or above `above` (:806-815)
    const int mw64 = (m + 63) >> 6;
    unsigned long long* __restrict__ wb = bits + win_bits_off[w];
    int nA = 0;
    for (int j = lane; j < m; j += 64) { const float d = dv[j]; if (d > below && (d == 1.f || d >= above)) nA++; }
    nA = wave_sum_i32(nA);
    float cut = 3.f;   // entries below `above` are taken from `cut` upwards
    if (nA < 5) {
        const int need = 5 - nA;
        int nB = 0;
        for (int j = lane; j < m; j += 64) { const float d = dv[j]; if (d > below && d != 1.f && d < above) nB++; }
        nB = wave_sum_i32(nB);
        if (nB > 0 && nB <= need) cut = -1.f;
        else if (nB > need) {
            int remaining = need; float cur = 2.f, v = 0.f; int cc = 0;
            bool ambiguous = false;
            for (int it = 0; it < 5; ++it) {
                next_level<1>(dv, m, lane, cur, below, above, 0, &v, &cc);
                if (remaining <= cc) { ambiguous = remaining != cc; break; }
                remaining -= cc; cur = v;
            }
            if (ambiguous) { give_up(); return; }   // std::sort's arrangement of the equal run decides
            cut = v;
        }
    }
    for (int j = lane; j < m; j += 64) {
        const float d = dv[j];
        const bool take = d > below && ((d == 1.f || d >= above) || (d != 1.f && d < above && d >= cut));
        if (take) {
            atomicOr(&wb[(int64_t)i * mw64 + (j >> 6)], 1ull << (j & 63));
            atomicOr(&wb[(int64_t)j * mw64 + (i >> 6)], 1ull << (i & 63));
        }
    }
}

// rows the device could not decide: their sim / diff rows, compacted for one copy to the host
__global__ __launch_bounds__(256) void k_read_graph_fetch_rows(
    const int32_t* __restrict__ sim, const int32_t* __restrict__ diff, const int64_t* __restrict__ src_off,
    const int32_t* __restrict__ len, const int64_t* __restrict__ dst_off, int32_t* __restrict__ out_sim, int32_t* __restrict__ out_diff, int es) {
    const int k = (int)blockIdx.x;
    const int n = len[k];
    for (int j = (int)threadIdx.x; j < n; j += 256) { out_sim[dst_off[k] + j] = sim[(src_off[k] + j) * es]; out_diff[dst_off[k] + j] = diff[(src_off[k] + j) * es]; }
}

// links decided on the host: (window-local row, window-local column) pairs with the bit-matrix base of their window
__global__ __launch_bounds__(256) void k_read_graph_patch(const int64_t* __restrict__ base, const int32_t* __restrict__ mw64,
                                                         const int32_t* __restrict__ pi, const int32_t* __restrict__ pj, int n,
                                                         unsigned long long* __restrict__ bits) {
    const int k = (int)(blockIdx.x * 256 + threadIdx.x);
    if (k >= n) return;
    unsigned long long* wb = bits + base[k];
    atomicOr(&wb[(int64_t)pi[k] * mw64[k] + (pj[k] >> 6)], 1ull << (pj[k] & 63));
    atomicOr(&wb[(int64_t)pj[k] * mw64[k] + (pi[k] >> 6)], 1ull << (pi[k] & 63));
}

// degree of every row of every window
__global__ __launch_bounds__(256) void k_read_graph_degrees(const unsigned long long* __restrict__ bits, const int32_t* __restrict__ row_win,
                                                           const int64_t* __restrict__ win_mask_off, const int64_t* __restrict__ win_bits_off,
                                                           int n_rows, int32_t* __restrict__ deg) {
    const int row = (int)(blockIdx.x * 256 + threadIdx.x);
    if (row >= n_rows) return;
    const int w = row_win[row];
    const int m = (int)(win_mask_off[w + 1] - win_mask_off[w]);
    const int mw64 = (m + 63) >> 6;
    const unsigned long long* p = bits + win_bits_off[w] + (int64_t)(row - win_mask_off[w]) * mw64;
    int d = 0;
    for (int k = 0; k < mw64; ++k) d += __popcll(p[k]);
    deg[row] = d;
}

// exclusive prefix sum of n non-negative ints (degrees of graph rows, selection counts of tiles); out has n + 1 entries.
// Three launches: per-tile sums (4096 ints per workgroup, coalesced), scan of the tile sums by one workgroup, and the scan
// inside every tile on top of its offset. A tile sum fits 32 bits (4096 x at most 20 000); offsets are 64-bit.
#define HS_SCAN_TILE 4096
__global__ __launch_bounds__(256) void k_scan_tile_sums(const int32_t* __restrict__ in, int n, long long* __restrict__ tile_sum) {
    __shared__ int s_w[4];
    const int t = (int)threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * HS_SCAN_TILE;
    int s = 0;
#pragma unroll
    for (int k = 0; k < HS_SCAN_TILE / 256; ++k) { const int64_t i = base + k * 256 + t; if (i < n) s += in[i]; }
    s = wave_sum_i32(s);
    if ((t & 63) == 0) s_w[t >> 6] = s;
    __syncthreads();
    if (t == 0) tile_sum[blockIdx.x] = (long long)s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(1024) void k_scan_tile_offsets(const long long* __restrict__ tile_sum, int n_tiles, long long* __restrict__ tile_off,
                                                           int64_t* __restrict__ total_out) {
    __shared__ long long s_part[1024];
    const int t = (int)threadIdx.x;
    long long carry = 0;
    for (int base = 0; base < n_tiles; base += 1024) {
        const long long v = base + t < n_tiles ? tile_sum[base + t] : 0;
        s_part[t] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const long long a = t >= d ? s_part[t - d] : 0;
            __syncthreads();
            s_part[t] += a;
            __syncthreads();
        }
        if (base + t < n_tiles) tile_off[base + t] = carry + s_part[t] - v;
        carry += s_part[1023];
        __syncthreads();
    }
    if (t == 0) *total_out = carry;
}

__global__ __launch_bounds__(256) void k_scan_apply(const int32_t* __restrict__ in, int n, const long long* __restrict__ tile_off,
                                                    int64_t* __restrict__ out) {
    __shared__ int s_w[2][4];
    const int t = (int)threadIdx.x, lane = t & 63, wv = t >> 6;
    const int64_t base = (int64_t)blockIdx.x * HS_SCAN_TILE;
    long long run = tile_off[blockIdx.x];
#pragma unroll 1
    for (int k = 0; k < HS_SCAN_TILE / 256; ++k) {
        const int64_t i = base + k * 256 + t;
        const int v = i < n ? in[i] : 0;
        const int incl = wave_scan_incl(v);
        if (lane == 63) s_w[k & 1][wv] = incl;
        __syncthreads();   // one barrier per chunk: the wave totals alternate between two slots
        const int w0 = s_w[k & 1][0], w1 = s_w[k & 1][1], w2 = s_w[k & 1][2], w3 = s_w[k & 1][3];
        const int before = (wv > 0 ? w0 : 0) + (wv > 1 ? w1 : 0) + (wv > 2 ? w2 : 0);
        if (i < n) out[i] = run + before + incl - v;
        run += (long long)w0 + w1 + w2 + w3;
    }
}

// neighbour lists as window-local indices, ascending (= ascending read id: the window's mask list is ascending)
__global__ __launch_bounds__(256) void k_read_graph_fill(const unsigned long long* __restrict__ bits, const int32_t* __restrict__ row_win,
                                                        const int64_t* __restrict__ win_mask_off, const int64_t* __restrict__ win_bits_off,
                                                        const int64_t* __restrict__ nbr_off, int n_rows, int32_t* __restrict__ nbr) {
    const int row = (int)(blockIdx.x * 256 + threadIdx.x);
    if (row >= n_rows) return;
    const int w = row_win[row];
    const int64_t m0 = win_mask_off[w];
    const int m = (int)(win_mask_off[w + 1] - m0);
    const int mw64 = (m + 63) >> 6;
    const unsigned long long* p = bits + win_bits_off[w] + (int64_t)(row - m0) * mw64;
    int32_t* o = nbr + nbr_off[row];
    for (int k = 0; k < mw64; ++k) {
        unsigned long long x = p[k];
        while (x) { const int b = __builtin_ctzll(x); x &= x - 1; *o++ = (k << 6) + b; }
    }
}

}  // namespace hsdev
