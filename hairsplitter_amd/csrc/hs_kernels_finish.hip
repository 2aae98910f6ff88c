// hs_kernels_finish.hip -- K8: the tail of finalize_clustering for one clustering window per wavefront:
// first-seen renumbering (separate_reads.cpp:973-984), merge_close_clusters (cluster_graph.cpp:402-501) and
// merge_wrongly_split_haplotypes (separate_reads.cpp:1007-1327), on the labels the third Chinese-Whispers wave left on the
// device. Small sequential logic (a handful of clusters, a few dozen reads, ~20 SNPs per window) that cost the host ~9 us
// per window; here it runs next to the data, one window per wavefront, tens of thousands of windows per launch.
// Windows outside the kernel's fixed-size tables (more than 16 cluster labels, more than 8 clusters left, more than 16
// cluster links -- where std::sort stops being an insertion sort) are reported back (ok = 0) and finished by the host code.
// Included by hs_capi.hip after hs_kernels.hip.
#pragma once

namespace hsdev {

#define HS_FIN_KCAP 16      // cluster labels entering merge_close_clusters
#define HS_FIN_GCAP 8       // clusters entering merge_wrongly_split
#define HS_FIN_LCAP 16      // cluster links (std::sort is a plain insertion sort up to 16 elements)
#define HS_FIN_MCAP (HS_FIN_KCAP + 2)

static __device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(64) void k_finish_window(
    const int32_t* __restrict__ labels_in, const int64_t* __restrict__ win_label_base, const int32_t* __restrict__ win_n,
    const int32_t* __restrict__ win_graph, const int32_t* __restrict__ adj_off, const int32_t* __restrict__ adj,
    const int64_t* __restrict__ graph_off_base, const int64_t* __restrict__ graph_adj_base, const uint8_t* __restrict__ mask,
    const int32_t* __restrict__ visit, const int32_t* __restrict__ visit_n, const int64_t* __restrict__ col_off,
    const int32_t* __restrict__ col_idx, const uint8_t* __restrict__ col_code, const int32_t* __restrict__ col_pos,
    const int64_t* __restrict__ win_snp_first, const int64_t* __restrict__ win_snp_last, const int32_t* __restrict__ win_pos_lo,
    const int32_t* __restrict__ win_pos_hi, int n_windows, int32_t* __restrict__ labels_out, uint8_t* __restrict__ ok_out) {
    extern __shared__ int32_t fin_lds[];
    __shared__ int s_votes[HS_FIN_KCAP], s_count[HS_FIN_KCAP], s_initial[HS_FIN_KCAP], s_tested[HS_FIN_KCAP];
    __shared__ int s_index_of[HS_FIN_KCAP], s_slot_of[HS_FIN_KCAP];
    __shared__ int s_cnts[HS_FIN_GCAP][256];
    __shared__ int s_nb[HS_FIN_GCAP], s_major[HS_FIN_GCAP], s_glist[HS_FIN_GCAP], s_gidx[HS_FIN_GCAP];
    __shared__ int s_incompat[HS_FIN_GCAP * HS_FIN_GCAP], s_pos_last[HS_FIN_GCAP * HS_FIN_GCAP];
    __shared__ int s_link_cnt[HS_FIN_MCAP * HS_FIN_MCAP], s_links_in[HS_FIN_MCAP], s_o2n[HS_FIN_MCAP], s_new_index[HS_FIN_MCAP];
    __shared__ int s_scalar[8];   // K, G, flags ... written by lane 0, read by all
    const int lane = lane_id();
    const int w = (int)blockIdx.x;
    if (w >= n_windows) return;
    const int N = win_n[w];
    const int g = win_graph[w];
    const int32_t* __restrict__ aoff = adj_off + graph_off_base[g];
    const int32_t* __restrict__ anb = adj + graph_adj_base[g];
    const uint8_t* __restrict__ msk = mask + graph_off_base[g] - g;
    const int32_t* __restrict__ vis = visit + graph_off_base[g] - g;
    const int n_visit = visit_n[g];
    const int32_t* __restrict__ lin = labels_in + win_label_base[w];
    int32_t* __restrict__ lout = labels_out + win_label_base[w];
    int32_t* lab = fin_lds;            // [N] current labels ("clusters")
    int32_t* nc = fin_lds + N;         // [N] trial labels / scratch map
    int32_t* mlist = fin_lds + 2 * N;  // [N] masked reads, ascending
    auto bail = [&]() { if (lane == 0) ok_out[w] = 0; };

    // ---- labels in, masked list, first-seen renumbering (:973-984) ----
    int m = 0;
    for (int r0 = 0; r0 < N; r0 += 64) {
        const int r = r0 + lane;
        const bool mk = r < N && msk[r] != 0;
        if (r < N) { lab[r] = lin[r]; nc[r] = -1; }
        const unsigned long long b = __ballot(mk);
        if (mk) mlist[m + __popcll(b & ((1ull << lane) - 1ull))] = r;
        m += __popcll(b);
    }
    wave_sync_lds();
    if (lane == 0) {
        int K = 0;
        bool bad = false;
        for (int j = 0; j < m; ++j) {
            const int r = mlist[j];
            const int l = lab[r];
            if (l >= 0) {
                if (l >= N) { bad = true; break; }
                if (nc[l] < 0) nc[l] = K++;      // nc doubles as the old -> new map here
                lab[r] = nc[l];
            }
        }
        s_scalar[0] = K; s_scalar[1] = bad ? 1 : 0;
    }
    wave_sync_lds();
    const int K = s_scalar[0];
    if (s_scalar[1] || K > HS_FIN_KCAP) { bail(); return; }

    // ---- merge_close_clusters (cluster_graph.cpp:402-501) ----
    if (lane < HS_FIN_KCAP) { s_initial[lane] = 0; s_votes[lane] = 0; s_tested[lane] = 0; }
    wave_sync_lds();
    for (int j = lane; j < m; j += 64) { const int l = lab[mlist[j]]; if (l >= 0) atomicAdd(&s_initial[l], 1); }
    for (int r = lane; r < N; r += 64) nc[r] = lab[r];
    wave_sync_lds();
    for (int j = 0; j < m; ++j) {
        const int target = lab[mlist[j]];                      // uniform
        if (target < 0 || s_tested[target]) continue;
        if (lane < K) s_count[lane] = s_initial[lane];
        wave_sync_lds();
        int changes = 3, iters = 0;
        while (changes > 0 && iters < 10) {
            changes = 0;
            for (int k0 = 0; k0 < n_visit; k0 += 64) {
                const int kk = k0 + lane;
                int i_l = -1, o0_l = 0, o1_l = 0;
                if (kk < n_visit) { i_l = vis[kk]; o0_l = aoff[i_l]; o1_l = aoff[i_l + 1]; }
                unsigned long long act = __ballot(kk < n_visit && nc[i_l < 0 ? 0 : i_l] == target);
                while (act) {
                    const int l = __builtin_ctzll(act);
                    act &= act - 1ull;
                    const int i = __builtin_amdgcn_readlane(i_l, l);
                    if (nc[i] != target) continue;            // changed earlier in this batch of 64 (cannot: only i itself changes)
                    const int o0 = __builtin_amdgcn_readlane(o0_l, l), o1 = __builtin_amdgcn_readlane(o1_l, l);
                    for (int o = o0 + lane; o < o1; o += 64) { const int lb = nc[anb[o]]; if (lb >= 0) atomicAdd(&s_votes[lb], 1); }
                    wave_sync_lds();
                    // largest and runner-up in ascending label order with strict '>' (:455-470): (count desc, label asc)
                    const int v = lane < K ? s_votes[lane] : 0;
                    const int key = v > 0 ? ((v << 8) | (255 - lane)) : 0;
                    const int best = wave_max_i32(key);
                    const int max_value = best >> 8, max_index = best ? 255 - (best & 255) : 0;
                    const int best2 = wave_max_i32((best && lane == max_index) ? 0 : key);
                    const int second_value = best2 >> 8, second_index = best2 ? 255 - (best2 & 255) : 0;
                    if (lane < K) s_votes[lane] = 0;
                    if (max_value > 0 && max_index != target) {
                        if (lane == 0) { s_count[target]--; s_count[max_index]++; nc[i] = max_index; }
                        changes++;
                    } else if (max_value > 0 && max_value <= 2 * second_value) {
                        if (lane == 0) { s_count[target]--; s_count[second_index]++; nc[i] = second_index; }
                        changes++;
                    }
                    wave_sync_lds();
                }
            }
            iters++;
        }
        const bool dissolved = s_count[target] == 0;
        wave_sync_lds();
        if (lane == 0) s_tested[target] = 1;
        if (dissolved) {
            for (int q = lane; q < m; q += 64) { const int r = mlist[q]; lab[r] = nc[r]; }
            if (lane < K) s_initial[lane] = s_count[lane];
        } else {
            for (int q = lane; q < m; q += 64) { const int r = mlist[q]; nc[r] = lab[r]; }
        }
        wave_sync_lds();
    }

    // ---- merge_wrongly_split_haplotypes (separate_reads.cpp:1007-1327) ----
    if (lane < HS_FIN_KCAP) { s_index_of[lane] = -1; s_slot_of[lane] = -1; }
    wave_sync_lds();
    if (lane == 0) {
        int index = 0;
        for (int j = 0; j < m; ++j) { const int cl = lab[mlist[j]]; if (cl > -1 && s_index_of[cl] < 0) s_index_of[cl] = index++; }
        int G = 0;
        for (int l = 0; l < K; ++l) if (s_index_of[l] >= 0) { if (G < HS_FIN_GCAP) { s_slot_of[l] = G; s_glist[G] = l; s_gidx[G] = s_index_of[l]; } G++; }
        s_scalar[2] = G;
    }
    wave_sync_lds();
    const int G = s_scalar[2];
    if (G <= 1) {
        for (int r = lane; r < N; r += 64) lout[r] = lab[r] == -2 ? -2 : 0;
        if (lane == 0) ok_out[w] = 1;
        return;
    }
    if (G > HS_FIN_GCAP) { bail(); return; }
    for (int x = lane; x < G * 256; x += 64) (&s_cnts[0][0])[x] = 0;
    for (int x = lane; x < G * G; x += 64) { s_incompat[x] = 0; s_pos_last[x] = -10; }
    wave_sync_lds();
    const int pos_lo = win_pos_lo[w], pos_hi = win_pos_hi[w];
    for (int64_t s = win_snp_first[w]; s < win_snp_last[w]; ++s) {
        const int p = col_pos[s];
        if (!(p >= pos_lo && p < pos_hi)) continue;
        if (lane < G) { s_nb[lane] = 0; s_major[lane] = 0; }   // 0 == the operator[] default for clusters absent at this SNP
        wave_sync_lds();
        for (int64_t e = col_off[s] + lane; e < col_off[s + 1]; e += 64) {
            const int cl = lab[col_idx[e]];
            if (cl > -1) { const int sl = s_slot_of[cl]; atomicAdd(&s_cnts[sl][col_code[e]], 1); atomicAdd(&s_nb[sl], 1); }
        }
        wave_sync_lds();
        for (int i = 0; i < G; ++i) {
            // (largest count, runner-up count) of the cluster's bases; a tied maximum yields runner-up == maximum (:1090-1099)
            int t1 = 0, c1 = -1, t2 = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int code = lane + 64 * q;
                const int v = s_cnts[i][code];
                if (v > t1) { t2 = t1; t1 = v; c1 = code; } else if (v > t2) t2 = v;
                s_cnts[i][code] = 0;
            }
            const int mx = wave_max_i32(t1);
            if (mx == 0) continue;                              // cluster absent at this SNP: majority stays 0
            const int n_at = wave_sum_i32((t1 == mx ? 1 : 0) + (t2 == mx ? 1 : 0));
            const int below = wave_max_i32(t1 == mx ? t2 : t1);  // best count strictly below the maximum when it is unique
            const int second_max = n_at >= 2 ? mx : below;
            int max_base = wave_max_i32(t1 == mx ? c1 : -1);
            if (second_max * 2 > mx || s_nb[i] * 0.5 > mx) max_base = ' ';
            if (lane == 0) s_major[i] = max_base & 255;
        }
        wave_sync_lds();
        int first_max = -1; bool several = false;
        for (int i = 0; i < G; ++i) {
            const int mb = s_major[i];
            if (mb == 0 || mb == ' ') { if (mb == 0) continue; continue; }
            if (first_max < 0) first_max = mb; else if (mb != first_max) several = true;
        }
        // clusters absent at the SNP keep majority 0, which is not ' ': they take part in the comparison below (sic), but they
        // do not count as a "max base" for the `several` test (:1100-1112 only inserts bases of clusters that carry reads)
        if (several && lane < G * G) {
            const int a = lane / G, b = lane % G;
            const int ma = s_major[a], mb = s_major[b];
            if (ma != ' ' && mb != ' ' && s_glist[a] > s_glist[b]) {
                const int i1 = s_gidx[a], i2 = s_gidx[b];
                if (ma != mb && p - s_pos_last[i1 * G + i2] > 10) {
                    s_incompat[i1 * G + i2] += 1; s_incompat[i2 * G + i1] += 1;
                    s_pos_last[i1 * G + i2] = p; s_pos_last[i2 * G + i1] = p;
                }
            }
        }
        wave_sync_lds();
    }
    // link ratios (:1189-1250): a dense (label + 2) x (label + 2) count matrix walked in ascending key order
    const int M = K + 2;
    for (int x = lane; x < M * M; x += 64) s_link_cnt[x] = 0;
    if (lane < M) s_links_in[lane] = 0;
    wave_sync_lds();
    for (int q = lane; q < m; q += 64) {
        const int k = mlist[q];
        const int c2 = lab[k] + 2;
        for (int o = aoff[k]; o < aoff[k + 1]; ++o) {
            const int c1 = lab[anb[o]] + 2;
            if (c1 != c2) atomicAdd(&s_link_cnt[c1 * M + c2], 1);
            atomicAdd(&s_links_in[c1], 1);
        }
    }
    wave_sync_lds();
    if (lane == 0) {
        int lc1[HS_FIN_LCAP], lc2[HS_FIN_LCAP];
        double lr[HS_FIN_LCAP];
        int nl = 0;
        bool over = false;
        for (int c1 = 0; c1 < M && !over; ++c1)
            for (int c2 = 0; c2 < M; ++c2)
                if (s_link_cnt[c1 * M + c2] > 0) {
                    if (nl == HS_FIN_LCAP) { over = true; break; }
                    lc1[nl] = c1 - 2; lc2[nl] = c2 - 2; lr[nl] = (double)s_link_cnt[c1 * M + c2] / s_links_in[c1]; nl++;
                }
        if (over) { s_scalar[3] = 1; }
        else {
            s_scalar[3] = 0;
            // std::sort with `a.second > b.second` on <= 16 elements == libstdc++'s insertion sort (stl_algo.h __insertion_sort)
            for (int i = 1; i < nl; ++i) {
                const int a1 = lc1[i], a2 = lc2[i]; const double ar = lr[i];
                int j = i;
                while (j > 0 && ar > lr[j - 1]) { lc1[j] = lc1[j - 1]; lc2[j] = lc2[j - 1]; lr[j] = lr[j - 1]; --j; }
                lc1[j] = a1; lc2[j] = a2; lr[j] = ar;
            }
            for (int x = 0; x < M; ++x) s_o2n[x] = 0;
            for (int i = 0; i < G; ++i) s_o2n[s_glist[i] + 2] = s_glist[i];
            s_o2n[1] = -1; s_o2n[0] = -2;
            for (int q = 0; q < nl; ++q) {
                if (!(lr[q] > 0.01)) continue;
                const int c1 = lc1[q], c2 = lc2[q];
                if (s_o2n[c1 + 2] == s_o2n[c2 + 2]) continue;
                bool bad = false;
                for (int i = 0; i < G; ++i) {
                    if (s_o2n[s_glist[i] + 2] != s_o2n[c1 + 2]) continue;
                    for (int k = 0; k < G; ++k)
                        if (s_o2n[s_glist[k] + 2] == s_o2n[c2 + 2] && s_incompat[s_gidx[i] * G + s_gidx[k]] > 1) bad = true;
                }
                if (!bad) { const int to = s_o2n[c1 + 2], from = s_o2n[c2 + 2]; for (int k = 0; k < G; ++k) if (s_o2n[s_glist[k] + 2] == from) s_o2n[s_glist[k] + 2] = to; }
            }
            for (int x = 0; x < M; ++x) s_new_index[x] = -1;
            int ni = 0;
            for (int i = 0; i < G; ++i) { const int v = s_o2n[s_glist[i] + 2]; if (s_new_index[v + 2] < 0) s_new_index[v + 2] = ni++; }
            for (int i = 0; i < G; ++i) s_o2n[s_glist[i] + 2] = s_new_index[s_o2n[s_glist[i] + 2] + 2];
        }
    }
    wave_sync_lds();
    if (s_scalar[3]) { bail(); return; }
    for (int r = lane; r < N; r += 64) lout[r] = s_o2n[lab[r] + 2];
    if (lane == 0) ok_out[w] = 1;
}

}  // namespace hsdev
