// hs_kernels_cw.hip -- the clustering chain of a window in the window's LOCAL index space.
//
// A clustering window only ever touches its masked reads (the reads present at its first and last SNP,
// separate_reads.cpp:1590-1622): create_read_graph_matrix links masked reads only (:806-815), Chinese Whispers only
// updates masked nodes and a masked node only has masked neighbours (cluster_graph.cpp:240-310), every later step
// (:840-994, :1007-1327) skips the label -2 the other reads carry. Node j of a window is therefore the j-th masked read
// (ascending read id, so "lowest label wins" and "first appearance" mean the same as on read ids); a graph is the window's
// row range of the K6 CSR; labels, vote counters and every table are m-sized (m ~ 40 of the N ~ 500..20 000 reads of a
// contig). The one place where the VALUE of a label matters is merge_clusterings' double key (:848-853): it is formed from
// the read ids (mask_ids[label]).
//
//   k_cw_visit_lists      per window: nodes with neighbours in the order of the contig's shuffled permutation
//   k_cw_seeded_rows      per-SNP runs (:1674-1705), FOUR instances per wavefront: one 16-lane DPP row per instance
//                         (mean degree ~ 16), windows with m <= 256
//   k_cw_seeded_wave      the same for any m, one wavefront per instance (labels in LDS, or in global scratch beyond its size)
//   k_window_tail         one wavefront per window: merge_clusterings ids -> CW -> small clusters dropped -> CW
//                         (finalize_clustering :897-971) -> first-seen renumbering, merge_close_clusters
//                         (cluster_graph.cpp:402-501), merge_wrongly_split_haplotypes (:1007-1327); labels never leave LDS
//   k_cw_local            one wavefront per (window, initial labels): the optional ploidy cap (:1341-1396)
// Included by hs_capi.hip after hs_kernels.hip.
#pragma once

namespace hsdev {

#define HS_FIN_KCAP 16      // cluster labels entering merge_close_clusters
#define HS_FIN_GCAP 8       // clusters entering merge_wrongly_split
#define HS_FIN_LCAP 16      // cluster links (std::sort is a plain insertion sort up to 16 elements)
#define HS_FIN_MCAP (HS_FIN_KCAP + 2)
#define HS_TAIL_MAP_EXTRA 1024      // bytes of dynamic LDS behind the tail's arrays: the read id -> cluster map of narrow windows
#define HS_TAIL_BATCH 8     // SNP columns of a window in flight at once in merge_wrongly_split
#define HS_CWR_CAP 255      // nodes per run in the row-packed kernel (local ids and labels are bytes, 255 = none)
#ifndef HS_CW_REG_LABELS
#define HS_CW_REG_LABELS 8
#endif

static __device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// position of read id r in the ascending list ids[0..m), or -1
static __device__ __forceinline__ int local_index(const int32_t* __restrict__ ids, int m, int r) {
    int lo = 0, hi = m;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (ids[mid] < r) lo = mid + 1; else hi = mid; }
    return (lo < m && ids[lo] == r) ? lo : -1;
}

// ------------------------------------------------------------------------------------------------
// Visiting order of a window: its nodes with at least one neighbour, in the order of the contig's permutation
// (std::shuffle(mt19937(seed)) of 0..N-1 on the host: cluster_graph.cpp:254-258; with the pinned seed the same order in
// every sweep). rank[r] = position of read r in that permutation; the nodes are sorted by rank by counting, per node, the
// nodes that come before it (m ~ 40: one pass of the wave over an LDS copy of the ranks).
// One workgroup (256 threads) per window.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cw_visit_lists(
    const int64_t* __restrict__ off, const int32_t* __restrict__ nbr, const int64_t* __restrict__ win_row0, const int32_t* __restrict__ mask_ids,
    const int32_t* __restrict__ win_contig, const int64_t* __restrict__ ctg_rank_off, const int32_t* __restrict__ rank,
    int n_windows, int cap, int32_t* __restrict__ visit, int32_t* __restrict__ visit_n,
    // the "visit program" of windows with m <= HS_CWR_CAP for k_cw_seeded_rows (nullptr: not wanted): per visit one dword
    // {node, chunks << 8, first chunk << 16} at prog_info[row0 + v]; the neighbour lists cut in 16-byte chunks of local ids
    // (255 = none) at prog_bytes[off[row0] + 15 * row0 ...] (room for nnz + 15 m bytes per window); chunks in all at prog_steps[w]
    uint32_t* __restrict__ prog_info, uint8_t* __restrict__ prog_bytes, int32_t* __restrict__ prog_steps,
    // windows with m <= 64 (k_cw_seeded_lanes): the neighbours of the v-th visited node as one bit mask over local ids
    unsigned long long* __restrict__ prog_adj) {
    extern __shared__ int32_t s_rank[];   // [cap] rank of every node with neighbours, -1 otherwise
    const int w = (int)blockIdx.x;
    if (w >= n_windows) return;
    const int tid = (int)threadIdx.x;
    const int64_t r0 = win_row0[w];
    const int m = (int)(win_row0[w + 1] - r0);
    const int32_t* __restrict__ rk = rank + ctg_rank_off[win_contig[w]];
    int32_t* __restrict__ out = visit + r0;
    __shared__ int s_total;
    __shared__ int s_scan[256];
    if (tid == 0) s_total = 0;
    __syncthreads();
    // windows wider than the LDS copy are processed against global memory (rank gathers straight from `rank`)
    const bool in_lds = m <= cap;
    int mine = 0;
    for (int j = tid; j < m; j += 256) {
        const bool has = off[r0 + j + 1] > off[r0 + j];
        const int v = has ? rk[mask_ids[r0 + j]] : -1;
        if (in_lds) s_rank[j] = v;
        mine += has ? 1 : 0;
    }
    atomicAdd(&s_total, mine);
    __syncthreads();
    for (int j = tid; j < m; j += 256) {
        const int v = in_lds ? s_rank[j] : (off[r0 + j + 1] > off[r0 + j] ? rk[mask_ids[r0 + j]] : -1);
        if (v < 0) continue;
        int before = 0;
        if (in_lds) { for (int k = 0; k < m; ++k) { const int u = s_rank[k]; before += (u >= 0 && u < v) ? 1 : 0; } }
        else { for (int k = 0; k < m; ++k) { if (off[r0 + k + 1] > off[r0 + k]) before += rk[mask_ids[r0 + k]] < v ? 1 : 0; } }
        out[before] = j;
    }
    if (tid == 0) visit_n[w] = s_total;
    if (!prog_info || m > HS_CWR_CAP) return;              // (block-uniform)
    __threadfence_block();
    __syncthreads();
    const int n_visit = s_total;                           // <= 255
    const int64_t base = off[r0];
    int i = 0, deg = 0, nc = 0;
    if (tid < n_visit) { i = out[tid]; deg = (int)(off[r0 + i + 1] - off[r0 + i]); nc = (deg + 15) >> 4; }
    s_scan[tid] = nc;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {                    // inclusive scan of the chunk counts
        const int a = tid >= d ? s_scan[tid - d] : 0;
        __syncthreads();
        s_scan[tid] += a;
        __syncthreads();
    }
    const int first = s_scan[tid] - nc;
    if (tid == 255) prog_steps[w] = s_scan[255];
    if (tid < n_visit) {
        prog_info[r0 + tid] = (uint32_t)i | ((uint32_t)nc << 8) | ((uint32_t)first << 16);
        uint8_t* pb = prog_bytes + base + 15 * r0 + (int64_t)first * 16;
        const int32_t* an = nbr + off[r0 + i];
        for (int k = 0; k < nc * 16; ++k) pb[k] = k < deg ? (uint8_t)an[k] : (uint8_t)255;
        if (prog_adj && m <= 64) {
            unsigned long long a = 0ull;
            for (int k = 0; k < deg; ++k) a |= 1ull << an[k];
            prog_adj[r0 + tid] = a;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// One Chinese-Whispers run by one wavefront on a local graph (cluster_graph.cpp:240-310): nodes sequential in the visiting
// order, the neighbours of the current node across the lanes; <= 15 sweeps, stop when a sweep changes <= 2 nodes (:167).
// Vote: lowest label among the most frequent ones (:272-279), labels < 0 do not vote. With at most 64 neighbours the vote is
// taken in registers (one ballot per distinct label; nodes that see more than HS_CW_REG_LABELS distinct labels go through
// the counters `cnt`, which are all zero between visits); the neighbour ids of the next node are loaded one visit ahead.
// `lab` / `cnt`: m ints each, LDS or global. Returns the number of sweeps.
// ------------------------------------------------------------------------------------------------
static __device__ int cw_local_wave(const int64_t* __restrict__ off_w /* at the window's first row */, const int32_t* __restrict__ nbr,
                                    const int32_t* __restrict__ vis, int n_visit, int m, int32_t* lab, int32_t* cnt, int lane) {
    const int64_t base = off_w[0];
    const int32_t* __restrict__ anb = nbr + base;
    const bool small_labels = m <= 65535;
    int changes = 3, iters = 0;
    while (changes > 2 && iters < 15) {
        changes = 0;
        for (int k0 = 0; k0 < n_visit; k0 += 64) {
            const int kk = k0 + lane;
            int i_l = -1, o0_l = 0, o1_l = 0;
            if (kk < n_visit) { i_l = vis[kk]; o0_l = (int)(off_w[i_l] - base); o1_l = (int)(off_w[i_l + 1] - base); }
            unsigned long long act = __ballot(o1_l > o0_l);
            int nb_next = -1;
            if (act) {
                const int ln = __builtin_ctzll(act);
                const int p0 = __builtin_amdgcn_readlane(o0_l, ln), p1 = __builtin_amdgcn_readlane(o1_l, ln);
                nb_next = (p1 - p0 <= 64 && p0 + lane < p1) ? anb[p0 + lane] : -1;
            }
            while (act) {
                const int l = __builtin_ctzll(act);
                act &= act - 1ull;
                const int i = __builtin_amdgcn_readlane(i_l, l);
                const int o0 = __builtin_amdgcn_readlane(o0_l, l), o1 = __builtin_amdgcn_readlane(o1_l, l);
                const int nb = nb_next;
                if (act) {
                    const int ln = __builtin_ctzll(act);
                    const int p0 = __builtin_amdgcn_readlane(o0_l, ln), p1 = __builtin_amdgcn_readlane(o1_l, ln);
                    nb_next = (p1 - p0 <= 64 && p0 + lane < p1) ? anb[p0 + lane] : -1;
                }
                int best_cnt = 0, best_lab = -1;
                if (o1 - o0 <= 64) {
                    const int lb = nb >= 0 ? lab[nb] : -1;
                    unsigned long long rem = __ballot(lb >= 0);
                    unsigned best_key = 0u;   // count << 16 | (65535 - label): largest count, lowest label among equals
                    if (small_labels) {
#pragma unroll
                        for (int tries = 0; tries < HS_CW_REG_LABELS; ++tries) {
                            if (!rem) break;
                            const int v = __builtin_amdgcn_readlane(lb, __builtin_ctzll(rem));
                            const unsigned long long mm = __ballot(lb == v);
                            const unsigned key = ((unsigned)__popcll(mm) << 16) | (unsigned)(65535 - v);
                            best_key = key > best_key ? key : best_key;
                            rem &= ~mm;
                        }
                    }
                    best_cnt = (int)(best_key >> 16);
                    best_lab = best_cnt ? 65535 - (int)(best_key & 0xffffu) : -1;
                    if (rem) {
                        if (lb >= 0) atomicAdd(&cnt[lb], 1);
                        wave_sync_lds();
                        const int c = lb >= 0 ? cnt[lb] : 0;
                        best_cnt = wave_max_i32(c);
                        best_lab = 0x7fffffff - wave_max_i32((lb >= 0 && c == best_cnt) ? 0x7fffffff - lb : 0);
                        wave_sync_lds();
                        if (lb >= 0) cnt[lb] = 0;
                    }
                } else {
                    for (int o = o0 + lane; o < o1; o += 64) { const int lb = lab[anb[o]]; if (lb >= 0) atomicAdd(&cnt[lb], 1); }
                    wave_sync_lds();
                    unsigned long long best = 0ull;
                    for (int o = o0 + lane; o < o1; o += 64) {
                        const int lb = lab[anb[o]];
                        if (lb >= 0) {
                            const unsigned long long key = ((unsigned long long)(unsigned)cnt[lb] << 32) | (unsigned)(0x7fffffff - lb);
                            best = key > best ? key : best;
                        }
                    }
                    best = wave_max_u64(best);
                    wave_sync_lds();
                    for (int o = o0 + lane; o < o1; o += 64) { const int lb = lab[anb[o]]; if (lb >= 0) cnt[lb] = 0; }
                    best_cnt = (int)(best >> 32);
                    best_lab = 0x7fffffff - (int)(best & 0xffffffffull);
                }
                if (best_cnt > 0) {
                    if (lab[i] != best_lab) changes++;
                    wave_sync_lds();
                    if (lane == 0) lab[i] = best_lab;
                }
                wave_sync_lds();
            }
        }
        iters++;
    }
    return iters;
}

// Seeding of a per-SNP run (separate_reads.cpp:1678-1691): every node starts alone; the nodes that carry the same code at
// the seeding SNP start in the cluster of the first node (lowest read id) carrying it. `first`: 256 ints of scratch.
template <int LANES>
static __device__ __forceinline__ void cw_seed_labels(const int32_t* __restrict__ ids, int m, int64_t c0, int64_t c1,
                                                      const int32_t* __restrict__ col_idx, const uint8_t* __restrict__ col_code,
                                                      int32_t* first, int l) {
    for (int64_t e = c0 + l; e < c1; e += LANES) {
        const int j = local_index(ids, m, col_idx[e]);
        if (j >= 0) atomicMin(&first[col_code[e]], j);
    }
}

// ------------------------------------------------------------------------------------------------
// Per-SNP runs, row-packed: FOUR runs per wavefront, one 16-lane DPP row each, with its own slice of LDS (labels as
// bytes + vote counters). The lanes of a row hold the neighbours of the row's current node (mean degree ~ 16): LDS atomic
// votes, every lane reads its label's total, the row maximum of (count << 16 | 65535 - label) comes from four row_ror DPP
// steps. Rows of a wavefront run in lock step on different runs (EXEC masks off the finished ones).
// Windows with m > HS_CWR_CAP are left to k_cw_seeded_wave.
// Output: slab[inst_slab_off + j] = label (local) of node j.
// ------------------------------------------------------------------------------------------------
static __device__ __forceinline__ unsigned row_max_u32(unsigned v) {
    unsigned o;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false); v = o > v ? o : v;   // row_ror:8
    o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false); v = o > v ? o : v;   // row_ror:4
    o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false); v = o > v ? o : v;   // row_ror:2
    o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false); v = o > v ? o : v;   // row_ror:1
    return v;
}

// One workgroup (128 threads = 8 DPP rows) per UNIT = up to eight consecutive per-SNP runs of ONE window, so that the
// window's graph is staged once in LDS and shared by the rows: a "visit program" -- per visited node (in visiting order) its
// neighbour list cut in chunks of 16 bytes (local ids fit a byte: m <= 256), one dword of info per visit {node, chunks,
// first chunk}. Inside the sweeps nothing but LDS is touched: per visit one read of the info dword (prefetched a visit
// ahead), the neighbour bytes, the label gather, the atomic votes and the read-back, i.e. four LDS round trips on the
// dependent chain whatever the degree (up to 64 neighbours; beyond, a loop). A window whose program does not fit
// `prog_cap` runs the same sweeps against global memory.
__global__ __launch_bounds__(128) void k_cw_seeded_rows(
    const int64_t* __restrict__ off, const int32_t* __restrict__ nbr, const int64_t* __restrict__ win_row0,
    const int32_t* __restrict__ mask_ids, const int32_t* __restrict__ visit, const int32_t* __restrict__ visit_n,
    const uint32_t* __restrict__ prog_info, const uint8_t* __restrict__ prog_bytes, const int32_t* __restrict__ prog_steps,
    const int32_t* __restrict__ unit_win, const int32_t* __restrict__ unit_inst0, const int32_t* __restrict__ unit_n, int n_units,
    const int64_t* __restrict__ inst_seed_col, const int64_t* __restrict__ inst_slab_off, const int64_t* __restrict__ col_off,
    const int32_t* __restrict__ col_idx, const uint8_t* __restrict__ col_code, int m_cap, int prog_cap,
    int32_t* __restrict__ slab, unsigned long long* __restrict__ stat /* [2]: sweeps, bytes */) {
    extern __shared__ int32_t cwr_dyn[];
    const int tid = (int)threadIdx.x, row = tid >> 4, l = tid & 15;
    const int u = (int)blockIdx.x;
    if (u >= n_units) return;
    const int w = unit_win[u];
    const bool live = row < unit_n[u];
    const int inst = unit_inst0[u] + (live ? row : 0);
    const int64_t seed = inst_seed_col[inst];              // the seeding column: issued first, needed last
    const int64_t r0 = win_row0[w];
    const int m = (int)(win_row0[w + 1] - r0);
    const int n_visit = visit_n[w];
    const int steps = prog_steps[w];
    const int64_t c0 = col_off[seed], c1 = col_off[seed + 1];
    const int32_t* __restrict__ ids = mask_ids + r0;
    const int32_t* __restrict__ vis = visit + r0;
    const int64_t* __restrict__ off_w = off + r0;
    const int64_t base = off_w[0];
    const int32_t* __restrict__ anb = nbr + base;
    // up to four entries of the seeding column per lane (columns deeper than 64 reads re-read the rest below)
    int e_r[4]; int e_c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int64_t e = c0 + l + 16 * k; e_r[k] = e < c1 ? col_idx[e] : -1; e_c[k] = e < c1 ? (int)col_code[e] : 0; }
    // LDS: info[m_cap] | ids[m_cap] | per row cnt[cnt_cap] x 8 | prog bytes[prog_cap] | per row labels bytes[m_cap] x 8
    // (cnt_cap >= 256: the counters double as the first-node-per-code table while seeding)
    const int cnt_cap = m_cap > 256 ? m_cap : 256;
    uint32_t* s_info = reinterpret_cast<uint32_t*>(cwr_dyn);
    int32_t* s_ids = cwr_dyn + m_cap;
    int32_t* cnt = cwr_dyn + 2 * m_cap + row * cnt_cap;
    uint8_t* s_prog = reinterpret_cast<uint8_t*>(cwr_dyn + 2 * m_cap + 8 * cnt_cap);
    uint8_t* lab = s_prog + prog_cap + row * m_cap;
    // ---- the window's visit program (built once per window by k_cw_visit_lists) and read ids: three coalesced copies ----
    const bool staged = steps * 16 <= prog_cap;
    for (int v = tid; v < n_visit; v += 128) s_info[v] = prog_info[r0 + v];
    for (int j = tid; j < m; j += 128) s_ids[j] = ids[j];
    if (staged) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(prog_bytes + base + 15 * r0);      // (16-byte chunks; the base may be unaligned: byte-wise below if so)
        if ((((uintptr_t)src) & 3u) == 0) { for (int x = tid; x < steps * 4; x += 128) reinterpret_cast<uint32_t*>(s_prog)[x] = src[x]; }
        else { const uint8_t* sb = prog_bytes + base + 15 * r0; for (int x = tid; x < steps * 16; x += 128) s_prog[x] = sb[x]; }
    }
    // ---- seeding (:1678-1691) ----
    for (int j = l; j < cnt_cap; j += 16) cnt[j] = 0x7fffffff;
    for (int j = l; j < m_cap; j += 16) lab[j] = (uint8_t)j;
    __syncthreads();
    int e_j[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { e_j[k] = (live && e_r[k] >= 0) ? local_index(s_ids, m, e_r[k]) : -1; if (e_j[k] >= 0) atomicMin(&cnt[e_c[k]], e_j[k]); }
    if (live) for (int64_t e = c0 + 64 + l; e < c1; e += 16) { const int j = local_index(s_ids, m, col_idx[e]); if (j >= 0) atomicMin(&cnt[col_code[e]], j); }
    wave_sync_lds();
#pragma unroll
    for (int k = 0; k < 4; ++k) if (e_j[k] >= 0) lab[e_j[k]] = (uint8_t)cnt[e_c[k]];
    if (live) for (int64_t e = c0 + 64 + l; e < c1; e += 16) { const int j = local_index(s_ids, m, col_idx[e]); if (j >= 0) lab[j] = (uint8_t)cnt[col_code[e]]; }
    wave_sync_lds();
    for (int j = l; j < cnt_cap; j += 16) cnt[j] = 0;
    wave_sync_lds();
#ifdef HS_CW_DIAG
    if (live && l == 0) { int alive = 0; for (int j = 0; j < m; ++j) alive += lab[j] == j ? 1 : 0; atomicAdd(&stat[20 + (alive > 15 ? 15 : alive)], 1ull); atomicAdd(&stat[36 + (m >> 4 > 15 ? 15 : m >> 4)], 1ull); }
#endif

    // ---- the sweeps. The rows of a workgroup run the same window, so the visits, the node and its chunk count are the same
    // for all of them: control flow is wave-uniform (scalar branches, no EXEC juggling), a row that has converged just stops
    // storing. ----
    int iters = 0;
    bool active = live;
    if (staged) {
        while (__ballot(active) != 0ull) {
            int changes = 0;
            uint32_t inf = n_visit > 0 ? s_info[0] : 0u;
            for (int v = 0; v < n_visit; ++v) {
                const uint32_t inf_next = s_info[v + 1 < n_visit ? v + 1 : v];        // off the dependent chain
                const int i = (int)(inf & 255u);
                const int nc = __builtin_amdgcn_readfirstlane((int)((inf >> 8) & 255u));
                const uint8_t* pg = s_prog + (inf >> 16) * 16 + l;
                unsigned best = 0u;
                const int old = (int)lab[i];
                if (nc == 1) {
                    // up to 16 neighbours, one per lane: how many lanes of the row hold the same label = 15 row rotations
                    const int n0 = pg[0];
                    const int lb = n0 != 255 ? (int)lab[n0] : -1;
                    int cnt = 1;
#define HS_ROT(K) { const int o = __builtin_amdgcn_update_dpp(-2, lb, 0x120 + (K), 0xf, 0xf, false); cnt += o == lb ? 1 : 0; }
                    HS_ROT(1) HS_ROT(2) HS_ROT(3) HS_ROT(4) HS_ROT(5) HS_ROT(6) HS_ROT(7) HS_ROT(8)
                    HS_ROT(9) HS_ROT(10) HS_ROT(11) HS_ROT(12) HS_ROT(13) HS_ROT(14) HS_ROT(15)
#undef HS_ROT
                    best = row_max_u32(lb >= 0 ? (((unsigned)cnt << 16) | (unsigned)(65535 - lb)) : 0u);
                } else if (nc <= 4) {
                    // up to 64 neighbours, up to four labels per lane in registers; one round per distinct label of the row:
                    // the row maximum picks an uncounted label X, its count = set bits of the row's 16-bit slice of the
                    // ballots "label == X"
                    const int n0 = pg[0], n1 = pg[16], n2 = nc > 2 ? pg[32] : 255, n3 = nc > 3 ? pg[48] : 255;
                    int lb0 = n0 != 255 ? (int)lab[n0] : -1, lb1 = n1 != 255 ? (int)lab[n1] : -1;
                    int lb2 = n2 != 255 ? (int)lab[n2 & 255] : -1, lb3 = n3 != 255 ? (int)lab[n3 & 255] : -1;
                    const int sh = tid & 48;                  // first lane of this row inside the wavefront
                    bool left = true;
                    for (int round = 0; round < HS_CW_REG_LABELS; ++round) {
                        const int mine = lb0 >= 0 ? lb0 : (lb1 >= 0 ? lb1 : (lb2 >= 0 ? lb2 : lb3));
                        const unsigned key = row_max_u32((unsigned)(mine + 1));
                        if (__ballot(key != 0u) == 0ull) { left = false; break; }       // every row of the wavefront is done
                        const int X = (int)key - 1;                                      // -1 in a row that is done: matches nothing below
                        const unsigned long long b0 = __ballot(lb0 == X && X >= 0), b1 = __ballot(lb1 == X && X >= 0);
                        const unsigned long long b2 = __ballot(lb2 == X && X >= 0), b3 = __ballot(lb3 == X && X >= 0);
                        const int c = __popc((unsigned)(b0 >> sh) & 0xffffu) + __popc((unsigned)(b1 >> sh) & 0xffffu) + __popc((unsigned)(b2 >> sh) & 0xffffu)
                                    + __popc((unsigned)(b3 >> sh) & 0xffffu);
                        const unsigned k = X >= 0 ? (((unsigned)c << 16) | (unsigned)(65535 - X)) : 0u;
                        best = k > best ? k : best;
                        lb0 = lb0 == X ? -1 : lb0; lb1 = lb1 == X ? -1 : lb1; lb2 = lb2 == X ? -1 : lb2; lb3 = lb3 == X ? -1 : lb3;
                    }
                    if (left && __ballot((lb0 & lb1 & lb2 & lb3) >= 0 || lb0 >= 0 || lb1 >= 0 || lb2 >= 0 || lb3 >= 0) != 0ull) {
                        // some row saw more distinct labels than rounds: everything of this visit again through the LDS counters
                        const int m0 = n0 != 255 ? (int)lab[n0] : -1, m1 = n1 != 255 ? (int)lab[n1] : -1, m2 = n2 != 255 ? (int)lab[n2 & 255] : -1, m3 = n3 != 255 ? (int)lab[n3 & 255] : -1;
                        best = 0u;
                        if (m0 >= 0) atomicAdd(&cnt[m0], 1);
                        if (m1 >= 0) atomicAdd(&cnt[m1], 1);
                        if (m2 >= 0) atomicAdd(&cnt[m2], 1);
                        if (m3 >= 0) atomicAdd(&cnt[m3], 1);
                        wave_sync_lds();
                        if (m0 >= 0) { const unsigned k = ((unsigned)cnt[m0] << 16) | (unsigned)(65535 - m0); best = k > best ? k : best; }
                        if (m1 >= 0) { const unsigned k = ((unsigned)cnt[m1] << 16) | (unsigned)(65535 - m1); best = k > best ? k : best; }
                        if (m2 >= 0) { const unsigned k = ((unsigned)cnt[m2] << 16) | (unsigned)(65535 - m2); best = k > best ? k : best; }
                        if (m3 >= 0) { const unsigned k = ((unsigned)cnt[m3] << 16) | (unsigned)(65535 - m3); best = k > best ? k : best; }
                        best = row_max_u32(best);
                        wave_sync_lds();
                        if (m0 >= 0) cnt[m0] = 0;
                        if (m1 >= 0) cnt[m1] = 0;
                        if (m2 >= 0) cnt[m2] = 0;
                        if (m3 >= 0) cnt[m3] = 0;
                    }
                } else {
                    for (int c = 0; c < nc; ++c) { const int nb = pg[c * 16]; if (nb != 255) atomicAdd(&cnt[lab[nb]], 1); }
                    wave_sync_lds();
                    for (int c = 0; c < nc; ++c) { const int nb = pg[c * 16]; if (nb != 255) { const int lb = lab[nb]; const unsigned k = ((unsigned)cnt[lb] << 16) | (unsigned)(65535 - lb); best = k > best ? k : best; } }
                    best = row_max_u32(best);
                    wave_sync_lds();
                    for (int c = 0; c < nc; ++c) { const int nb = pg[c * 16]; if (nb != 255) cnt[lab[nb]] = 0; }
                }
                const int best_lab = 65535 - (int)(best & 0xffffu);      // a visited node has neighbours: the count is > 0
                changes += old != best_lab ? 1 : 0;
                wave_sync_lds();
                if (l == 0 && active) lab[i] = (uint8_t)best_lab;
                wave_sync_lds();
                inf = inf_next;
            }
            if (active) { iters++; active = changes > 2 && iters < 15; }
        }
    } else {
        int changes = 3;
        while (live && changes > 2 && iters < 15) {
            changes = 0;
            for (int v = 0; v < n_visit; ++v) {
                const int i = vis[v];
                const int o0 = (int)(off_w[i] - base), o1 = (int)(off_w[i + 1] - base);
                unsigned best = 0u;
                for (int o = o0 + l; o < o1; o += 16) atomicAdd(&cnt[lab[anb[o]]], 1);
                wave_sync_lds();
                for (int o = o0 + l; o < o1; o += 16) { const int lb = lab[anb[o]]; const unsigned key = ((unsigned)cnt[lb] << 16) | (unsigned)(65535 - lb); best = key > best ? key : best; }
                best = row_max_u32(best);
                wave_sync_lds();
                for (int o = o0 + l; o < o1; o += 16) cnt[lab[anb[o]]] = 0;
                const int best_lab = 65535 - (int)(best & 0xffffu);
                if ((int)lab[i] != best_lab) changes++;
                wave_sync_lds();
                if (l == 0) lab[i] = (uint8_t)best_lab;
                wave_sync_lds();
            }
            iters++;
        }
    }
    if (live) {
        int32_t* __restrict__ out = slab + inst_slab_off[inst];
        for (int j = l; j < m; j += 16) out[j] = (int32_t)lab[j];
        if (l == 0 && stat) {
            atomicAdd(&stat[0], (unsigned long long)iters);
            atomicAdd(&stat[1], (unsigned long long)iters * (4ull * (unsigned long long)(off_w[m] - base) + 8ull * (unsigned long long)m));
            atomicAdd(&stat[4 + (iters > 15 ? 15 : iters)], 1ull);      // histogram of the sweeps per run (diagnostic, HS_TIMING)
        }
    }
}

// The same for any window: one wavefront per instance. lds_cap = nodes whose labels + counters fit the dynamic LDS of the
// launch; wider windows keep them in `gscratch` (2 * m ints per instance at gscratch_off[k]).
__global__ __launch_bounds__(64) void k_cw_seeded_wave(
    const int64_t* __restrict__ off, const int32_t* __restrict__ nbr, const int64_t* __restrict__ win_row0,
    const int32_t* __restrict__ mask_ids, const int32_t* __restrict__ visit, const int32_t* __restrict__ visit_n,
    const int32_t* __restrict__ inst_list, int n_list, const int32_t* __restrict__ n_list_dev /* when the list was made on the device */,
    const int32_t* __restrict__ inst_win, const int64_t* __restrict__ inst_seed_col,
    const int64_t* __restrict__ inst_slab_off, const int64_t* __restrict__ col_off, const int32_t* __restrict__ col_idx,
    const uint8_t* __restrict__ col_code, int lds_cap, int32_t* __restrict__ gscratch, const int64_t* __restrict__ gscratch_off,
    int32_t* __restrict__ slab, unsigned long long* __restrict__ stat) {
    extern __shared__ int32_t cw_dyn[];     // [2 * lds_cap]
    __shared__ int32_t s_first[256];
    const int lane = lane_id();
    const int n_runs = n_list_dev ? *n_list_dev : n_list;
    for (int k = (int)blockIdx.x; k < n_runs; k += (int)gridDim.x) {
    wave_sync_lds();
    const int inst = inst_list[k];
    const int w = inst_win[inst];
    const int64_t r0 = win_row0[w];
    const int m = (int)(win_row0[w + 1] - r0);
    const int32_t* __restrict__ ids = mask_ids + r0;
    int32_t* lab = m <= lds_cap ? cw_dyn : gscratch + gscratch_off[k];
    int32_t* cnt = lab + (m <= lds_cap ? lds_cap : m);
    for (int j = lane; j < m; j += 64) { lab[j] = j; cnt[j] = 0; }
    for (int j = lane; j < 256; j += 64) s_first[j] = 0x7fffffff;
    wave_sync_lds();
    const int64_t s = inst_seed_col[inst];
    cw_seed_labels<64>(ids, m, col_off[s], col_off[s + 1], col_idx, col_code, s_first, lane);
    wave_sync_lds();
    for (int64_t e = col_off[s] + lane; e < col_off[s + 1]; e += 64) {
        const int j = local_index(ids, m, col_idx[e]);
        if (j >= 0) lab[j] = s_first[col_code[e]];
    }
    wave_sync_lds();
    const int iters = cw_local_wave(off + r0, nbr, visit + r0, visit_n[w], m, lab, cnt, lane);
    int32_t* __restrict__ out = slab + inst_slab_off[inst];
    for (int j = lane; j < m; j += 64) out[j] = lab[j];
    if (lane == 0 && stat) {
        atomicAdd(&stat[0], (unsigned long long)iters);
        atomicAdd(&stat[1], (unsigned long long)iters * (4ull * (unsigned long long)(off[r0 + m] - off[r0]) + 8ull * (unsigned long long)m));
        atomicAdd(&stat[4 + (iters > 15 ? 15 : iters)], 1ull);
    }
    }
}

// ------------------------------------------------------------------------------------------------
// Per-SNP runs of windows with m <= 64, ONE RUN PER LANE. The labels of a run are kept as sets: at most HS_CWL_SLOTS labels
// are alive after seeding (one per base seen in the column plus one per read that does not cover it; 4-5 typically), slot s
// = the s-th lowest label, set[s] = the nodes that carry it as a 64-bit mask in two VGPRs. A visit is then
//     votes of label s = popcount(neighbours(i) & set[s]),  winner = max over s of (votes << 8 | 255 - s)
// -- the lowest label among the most frequent ones (cluster_graph.cpp:272-279) -- with no memory on the dependent chain
// but one LDS byte (the slot node i is in now, to count the change). The neighbour masks come from k_cw_visit_lists, in
// visiting order, and are fetched one visit ahead.
//   k_cw_seed_sets     one wavefront per run: labels from the seeding column (:1678-1691) -> slot of every node, the sets,
//                      the label each slot stands for; runs with more labels alive than slots go to `ovf_list`
//   k_cw_seeded_lanes  64 runs per wavefront
// ------------------------------------------------------------------------------------------------
#define HS_CWL_SLOTS 16

__global__ __launch_bounds__(256) void k_cw_seed_sets(
    const int64_t* __restrict__ win_row0, const int32_t* __restrict__ mask_ids, const int32_t* __restrict__ inst_list, int n_list,
    const int32_t* __restrict__ inst_win, const int64_t* __restrict__ inst_seed_col, const int64_t* __restrict__ col_off,
    const int32_t* __restrict__ col_idx, const uint8_t* __restrict__ col_code,
    unsigned long long* __restrict__ seed_sets /* [n_list][16] */, uint8_t* __restrict__ seed_names /* [n_list][16] */,
    uint8_t* __restrict__ seed_slots /* [n_list][64] */, uint8_t* __restrict__ seed_n /* [n_list]: labels alive, 255 = too many */,
    int32_t* __restrict__ ovf_list, int32_t* __restrict__ ovf_n) {
    __shared__ int32_t s_ids[4][64];
    __shared__ int32_t s_lab[4][64];
    __shared__ int32_t s_first[4][256];
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6;
    const int q = (int)blockIdx.x * 4 + wv;
    if (q >= n_list) return;                      // (no workgroup barrier below: the wavefronts are independent)
    int32_t* ids = s_ids[wv]; int32_t* lab = s_lab[wv]; int32_t* first = s_first[wv];
    const int inst = inst_list[q];
    const int w = inst_win[inst];
    const int64_t seed = inst_seed_col[inst];
    const int64_t r0 = win_row0[w];
    const int m = (int)(win_row0[w + 1] - r0);    // <= 64
    const int64_t c0 = col_off[seed], c1 = col_off[seed + 1];
    const int64_t e0 = c0 + lane;
    const int R0 = e0 < c1 ? col_idx[e0] : -1;
    const int code0 = e0 < c1 ? (int)col_code[e0] : 0;
    ids[lane] = lane < m ? mask_ids[r0 + lane] : 0x7fffffff;
    lab[lane] = lane;
    for (int c = lane; c < 256; c += 64) first[c] = 0x7fffffff;
    wave_sync_lds();
    const int j0 = R0 >= 0 ? local_index(ids, m, R0) : -1;
    if (j0 >= 0) atomicMin(&first[code0], j0);
    for (int64_t e = e0 + 64; e < c1; e += 64) { const int j = local_index(ids, m, col_idx[e]); if (j >= 0) atomicMin(&first[col_code[e]], j); }
    wave_sync_lds();
    if (j0 >= 0) lab[j0] = first[code0];
    for (int64_t e = e0 + 64; e < c1; e += 64) { const int j = local_index(ids, m, col_idx[e]); if (j >= 0) lab[j] = first[col_code[e]]; }
    wave_sync_lds();
    const int L = lab[lane];                                            // < 64
    const bool node = lane < m;
    const unsigned long long leaders = __ballot(node && L == lane);
    const int alive = __popcll(leaders);
    const int slot = __popcll(leaders & ((1ull << L) - 1ull));         // labels in ascending order
    seed_slots[(int64_t)q * 64 + lane] = (uint8_t)(node ? slot : 0);
    if (alive > HS_CWL_SLOTS) {
        if (lane == 0) { seed_n[q] = 255; ovf_list[atomicAdd(ovf_n, 1)] = inst; }
        return;
    }
    unsigned long long mine = 0ull;
#pragma unroll
    for (int k = 0; k < HS_CWL_SLOTS; ++k) { const unsigned long long S = __ballot(node && slot == k); if (lane == k) mine = S; }
    if (lane < HS_CWL_SLOTS) seed_sets[(int64_t)q * HS_CWL_SLOTS + lane] = mine;
    if (node && L == lane) ids[slot] = lane;                            // (the read ids are not needed any more)
    wave_sync_lds();
    if (lane < HS_CWL_SLOTS) seed_names[(int64_t)q * HS_CWL_SLOTS + lane] = (uint8_t)(lane < alive ? ids[lane] : 0);
    if (lane == 0) seed_n[q] = (uint8_t)alive;
}

// The sweeps of 64 runs, one per lane, with NS slots in use (the slots above stay empty). Returns the lane's number of sweeps.
template <int NS>
static __device__ __forceinline__ int cwl_sweeps(unsigned (&set_lo)[HS_CWL_SLOTS], unsigned (&set_hi)[HS_CWL_SLOTS], uint8_t* my_slot,
                                                 const unsigned long long* __restrict__ adj_w, const uint32_t* __restrict__ inf_w, int nv, bool ok) {
    int iters = 0;
    bool active = ok;
    while (__ballot(active) != 0ull) {
        int changes = 0;
        unsigned long long adj_n = (active && nv > 0) ? adj_w[0] : 0ull;
        uint32_t inf_n = (active && nv > 0) ? inf_w[0] : 0u;
        for (int v = 0;; ++v) {
            const bool on = active && v < nv;
            if (__ballot(on) == 0ull) break;
            const unsigned long long adj = adj_n;
            const int i = (int)(inf_n & 255u);
            if (active && v + 1 < nv) { adj_n = adj_w[v + 1]; inf_n = inf_w[v + 1]; }
            const int old = (int)my_slot[i];
            const unsigned lo = (unsigned)adj, hi = (unsigned)(adj >> 32);
            unsigned best = 0u;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const unsigned c = (unsigned)__popc(lo & set_lo[s]) + (unsigned)__popc(hi & set_hi[s]);
                const unsigned key = (c << 8) | (unsigned)(255 - s);
                best = key > best ? key : best;
            }
            const int b = 255 - (int)(best & 255u);
            const bool chg = on && b != old;
            changes += chg ? 1 : 0;
            if (__ballot(chg) != 0ull) {
                const unsigned long long bit = chg ? 1ull << i : 0ull;
                const unsigned bl = (unsigned)bit, bh = (unsigned)(bit >> 32);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    set_lo[s] = (set_lo[s] & ~bl) | (s == b ? bl : 0u);
                    set_hi[s] = (set_hi[s] & ~bh) | (s == b ? bh : 0u);
                }
                if (chg) my_slot[i] = (uint8_t)b;
            }
        }
        if (active) { iters++; active = changes > 2 && iters < 15; }
    }
    return iters;
}

__global__ __launch_bounds__(64) void k_cw_seeded_lanes(
    const int64_t* __restrict__ off, const int64_t* __restrict__ win_row0, const int32_t* __restrict__ visit_n,
    const uint32_t* __restrict__ prog_info, const unsigned long long* __restrict__ prog_adj,
    const int32_t* __restrict__ inst_list, int n_list, const int32_t* __restrict__ inst_win, const int64_t* __restrict__ inst_slab_off,
    const unsigned long long* __restrict__ seed_sets, const uint8_t* __restrict__ seed_names, const uint8_t* __restrict__ seed_slots /* padded to 64 runs */,
    const uint8_t* __restrict__ seed_n, int32_t* __restrict__ slab, unsigned long long* __restrict__ stat) {
    __shared__ uint32_t s_slot[64 * 17];           // [run][68 bytes]: slot of every node (17 dwords apart: no bank conflicts)
    __shared__ uint32_t s_name[64 * 4];            // [run][16 bytes]: the label a slot stands for
    __shared__ unsigned int s_hist[16];
    const int lane = (int)threadIdx.x;
    const int q0 = (int)blockIdx.x * 64, q = q0 + lane;
    const bool valid = q < n_list;
    const int inst = valid ? inst_list[q] : 0;
    const int na = valid ? (int)seed_n[q] : 255;
    const bool ok = valid && na != 255;
    const int w = ok ? inst_win[inst] : 0;
    const int64_t r0 = ok ? win_row0[w] : 0;
    const int m = ok ? (int)(win_row0[w + 1] - r0) : 0;
    const int nv = ok ? visit_n[w] : 0;
    const int64_t out0 = ok ? inst_slab_off[inst] : 0;
    unsigned set_lo[HS_CWL_SLOTS], set_hi[HS_CWL_SLOTS];
    {
        const ulonglong2* p = reinterpret_cast<const ulonglong2*>(seed_sets + (int64_t)(valid ? q : 0) * HS_CWL_SLOTS);
#pragma unroll
        for (int k = 0; k < HS_CWL_SLOTS / 2; ++k) {
            const ulonglong2 v = p[k];
            set_lo[2 * k] = ok ? (unsigned)v.x : 0u; set_hi[2 * k] = ok ? (unsigned)(v.x >> 32) : 0u;
            set_lo[2 * k + 1] = ok ? (unsigned)v.y : 0u; set_hi[2 * k + 1] = ok ? (unsigned)(v.y >> 32) : 0u;
        }
        const uint4 nm = reinterpret_cast<const uint4*>(seed_names)[valid ? q : 0];
        s_name[lane * 4 + 0] = nm.x; s_name[lane * 4 + 1] = nm.y; s_name[lane * 4 + 2] = nm.z; s_name[lane * 4 + 3] = nm.w;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(seed_slots + (int64_t)q0 * 64);
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int x = lane + 64 * k; s_slot[(x >> 4) * 17 + (x & 15)] = src[x]; }
        if (lane < 16) s_hist[lane] = 0u;
    }
    wave_sync_lds();
    int nmax = 0;                                  // slots in use by any run of this wavefront
#pragma unroll
    for (int k = 1; k <= HS_CWL_SLOTS; ++k) if (__ballot(ok && na >= k) != 0ull) nmax = k;
    uint8_t* my_slot = reinterpret_cast<uint8_t*>(s_slot) + lane * 68;
    int iters;
    if (nmax <= 4) iters = cwl_sweeps<4>(set_lo, set_hi, my_slot, prog_adj + r0, prog_info + r0, nv, ok);
    else if (nmax <= 6) iters = cwl_sweeps<6>(set_lo, set_hi, my_slot, prog_adj + r0, prog_info + r0, nv, ok);
    else if (nmax <= 8) iters = cwl_sweeps<8>(set_lo, set_hi, my_slot, prog_adj + r0, prog_info + r0, nv, ok);
    else if (nmax <= 10) iters = cwl_sweeps<10>(set_lo, set_hi, my_slot, prog_adj + r0, prog_info + r0, nv, ok);
    else if (nmax <= 12) iters = cwl_sweeps<12>(set_lo, set_hi, my_slot, prog_adj + r0, prog_info + r0, nv, ok);
    else iters = cwl_sweeps<HS_CWL_SLOTS>(set_lo, set_hi, my_slot, prog_adj + r0, prog_info + r0, nv, ok);
    wave_sync_lds();
    // labels out, run by run: lane j writes node j (coalesced)
    const unsigned long long okm = __ballot(ok);
    const uint8_t* names = reinterpret_cast<const uint8_t*>(s_name);
    const uint8_t* slots = reinterpret_cast<const uint8_t*>(s_slot);
    const int out_lo = (int)(unsigned)out0, out_hi = (int)(out0 >> 32);
    for (int r = 0; r < 64; ++r) {
        if (!((okm >> r) & 1ull)) continue;
        const int m_r = __builtin_amdgcn_readlane(m, r);
        const int64_t o_r = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(out_hi, r) << 32) | (unsigned)__builtin_amdgcn_readlane(out_lo, r));
        if (lane < m_r) slab[o_r + lane] = (int32_t)names[r * 16 + slots[r * 68 + lane]];
    }
    if (stat) {
        unsigned long long sw = ok ? (unsigned long long)iters : 0ull;
        unsigned long long by = ok ? (unsigned long long)iters * (4ull * (unsigned long long)(off[r0 + m] - off[r0]) + 8ull * (unsigned long long)m) : 0ull;
        if (ok) atomicAdd(&s_hist[iters > 15 ? 15 : iters], 1u);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { sw += __shfl_xor(sw, d, 64); by += __shfl_xor(by, d, 64); }
        wave_sync_lds();
        if (lane == 0) { atomicAdd(&stat[0], sw); atomicAdd(&stat[1], by); }
        if (lane < 16 && s_hist[lane]) atomicAdd(&stat[4 + lane], (unsigned long long)s_hist[lane]);
    }
}

// One run per (window, initial local labels): labels_io[inst_label_off[i] .. + m). Used by the optional ploidy cap.
__global__ __launch_bounds__(64) void k_cw_local(
    const int64_t* __restrict__ off, const int32_t* __restrict__ nbr, const int64_t* __restrict__ win_row0,
    const int32_t* __restrict__ visit, const int32_t* __restrict__ visit_n, const uint8_t* __restrict__ win_final_empty,
    const int32_t* __restrict__ inst_win, const int64_t* __restrict__ inst_label_off, int n_inst, int lds_cap,
    int32_t* __restrict__ gscratch, const int64_t* __restrict__ gscratch_off, int32_t* __restrict__ labels_io) {
    extern __shared__ int32_t cw_dyn[];
    const int lane = lane_id();
    const int k = (int)blockIdx.x;
    if (k >= n_inst) return;
    const int w = inst_win[k];
    const int64_t r0 = win_row0[w];
    const int m = (int)(win_row0[w + 1] - r0);
    int32_t* lab = m <= lds_cap ? cw_dyn : gscratch + gscratch_off[k];
    int32_t* cnt = lab + (m <= lds_cap ? lds_cap : m);
    int32_t* __restrict__ io = labels_io + inst_label_off[k];
    for (int j = lane; j < m; j += 64) { lab[j] = io[j]; cnt[j] = 0; }
    wave_sync_lds();
    cw_local_wave(off + r0, nbr, visit + r0, win_final_empty[w] ? 0 : visit_n[w], m, lab, cnt, lane);
    for (int j = lane; j < m; j += 64) io[j] = lab[j];
}

// ------------------------------------------------------------------------------------------------
// A window of at most 64 nodes with its labels kept TWICE in registers: lane j = node j holds its label L (-1: none), lane l = label l
// holds the set S of its nodes as a bit mask (labels are node indices renumbered by first appearance: < m <= 64). A vote of a node's
// neighbours (one bit mask `adj`, k_cw_visit_lists) is then popcount(adj & S) in every lane at once and one wave maximum, instead of a
// leader loop over the distinct labels among the neighbours.
// ------------------------------------------------------------------------------------------------
static __device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#define HS_OR_STEP(ctrl, rm) lo |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, ctrl, rm, 0xf, false); hi |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, ctrl, rm, 0xf, false);
    HS_OR_STEP(0x111, 0xf) HS_OR_STEP(0x112, 0xf) HS_OR_STEP(0x114, 0xf) HS_OR_STEP(0x118, 0xf) HS_OR_STEP(0x142, 0xa) HS_OR_STEP(0x143, 0xc)
#undef HS_OR_STEP
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, 63) << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
}
static __device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffull), l);
}
// S from L: one 64-bit LDS atomic per node (s_sets: 64 words)
static __device__ __forceinline__ unsigned long long sets_from_labels(int L, unsigned long long* s_sets, int lane) {
    s_sets[lane] = 0ull;
    wave_sync_lds();
    if (L >= 0) atomicOr(&s_sets[L], 1ull << lane);
    wave_sync_lds();
    const unsigned long long S = s_sets[lane];
    wave_sync_lds();
    return S;
}
// labels renumbered by first appearance (the label whose first node comes first becomes 0, ...): returns the number of labels
static __device__ __forceinline__ int sets_renumber(unsigned long long& S, int& L, unsigned long long* s_sets, int* s_relabel, int lane) {
    const int f = S ? __builtin_ctzll(S) : 64;
    const unsigned long long F = wave_or_u64(S ? 1ull << f : 0ull);      // the first nodes of all labels
    const int nl = S ? __popcll(F & ((1ull << f) - 1ull)) : -1;
    s_sets[lane] = 0ull; s_relabel[lane] = nl;
    wave_sync_lds();
    if (nl >= 0) s_sets[nl] = S;
    const int Ln = L >= 0 ? s_relabel[L] : -1;
    wave_sync_lds();
    S = s_sets[lane]; L = Ln;
    wave_sync_lds();
    return __popcll(F);
}
// one Chinese-Whispers run (cluster_graph.cpp:240-310; what cw_local_wave does on lists): lane v holds the v-th visit's node and
// neighbour mask (adjv, nodev), lane j node j's neighbour mask (adjn). A sweep visits the nodes in order and a node takes the most frequent
// label among its neighbours, the lowest of equals (:272-279). While many labels are alive a sweep is walked visit by visit (one popcount
// per lane = label and a wave maximum per visit); with few labels alive, the votes of ALL nodes under the current labels are formed at
// once (lane = node, a loop over the alive labels) and the sweep jumps to the first node in visiting order that would change -- the
// nodes before it are stable under these labels and stay so until something changes --, applies that change and votes again.
#define HS_CW_SPEC_LABELS 8
static __device__ int cw_run_sets(unsigned long long adjv, int nodev, unsigned long long adjn, int n_visit, unsigned long long& S, int& L, int lane) {
    int changes = 3, iters = 0;
    while (changes > 2 && iters < 15) {
        changes = 0;
        unsigned long long alive = __ballot(S != 0ull);
        if (__popcll(alive) > HS_CW_SPEC_LABELS) {
            for (int v = 0; v < n_visit; ++v) {
                const unsigned long long adj = readlane_u64(adjv, v);
                const int i = __builtin_amdgcn_readlane(nodev, v);
                const int c = __popcll(adj & S);
                const int best = wave_max_i32(c ? ((c << 6) | (63 - lane)) : 0);
                if (best) {
                    const int bl = 63 - (best & 63);
                    const int old = __builtin_amdgcn_readlane(L, i);
                    if (old != bl) {
                        changes++;
                        const unsigned long long bit = 1ull << i;
                        if (lane == old) S &= ~bit;
                        if (lane == bl) S |= bit;
                        if (lane == i) L = bl;
                    }
                }
            }
        } else {
            int p = 0;      // visits before p are done in this sweep
            while (p < n_visit) {
                int best = 0;
                for (unsigned long long am = alive; am; am &= am - 1ull) {
                    const int l = __builtin_ctzll(am);
                    const int c = __popcll(adjn & readlane_u64(S, l));
                    const int key = c ? ((c << 6) | (63 - l)) : 0;
                    best = key > best ? key : best;
                }
                const int wants = 63 - (best & 63);
                const unsigned long long U = __ballot(best != 0 && wants != L);                                          // by node
                const unsigned long long Uv = __ballot(lane >= p && lane < n_visit && ((U >> nodev) & 1ull));          // by visit, from p on
                if (!Uv) break;
                const int v0 = __builtin_ctzll(Uv);
                const int i = __builtin_amdgcn_readlane(nodev, v0);
                const int bl = __builtin_amdgcn_readlane(wants, i), old = __builtin_amdgcn_readlane(L, i);
                const unsigned long long bit = 1ull << i;
                if (lane == old) S &= ~bit;
                if (lane == bl) S |= bit;
                if (lane == i) L = bl;
                changes++;
                p = v0 + 1;
                alive = __ballot(S != 0ull);
            }
        }
        iters++;
    }
    return iters;
}

// ------------------------------------------------------------------------------------------------
// id[j] = number of distinct first appearances before first[j], where first[j] = index of the first element equal to
// element j (or -1: no label). One wavefront, chunks of 64 with a ballot prefix; `pre` is m ints of scratch.
// ------------------------------------------------------------------------------------------------
static __device__ void first_seen_ids_wave(const int32_t* first, int m, int32_t* pre, int32_t* out_id, int lane) {
    int carry = 0;
    for (int b0 = 0; b0 < m; b0 += 64) {
        const int j = b0 + lane;
        const bool f = j < m && first[j] == j;
        const unsigned long long b = __ballot(f);
        if (j < m) pre[j] = carry + __popcll(b & ((1ull << lane) - 1ull));
        carry += __popcll(b);
    }
    wave_sync_lds();
    for (int j = lane; j < m; j += 64) { const int f = first[j]; out_id[j] = f >= 0 ? pre[f] : -1; }
    wave_sync_lds();
}

// ------------------------------------------------------------------------------------------------
// The tail of a clustering window, one wavefront per window of the chain (see the file header). Arrays of m ints in LDS
// (or in global scratch for windows wider than lds_cap): lab, nc, cnt, t0, t1 and m doubles agg.
// labels3_out: what the third Chinese-Whispers run leaves (for the windows the host has to finish: ok_out = 0 when the
// window exceeds the fixed tables of the cluster-merging steps or finish_on_device is off); final_out: finished labels.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_window_tail(
    const int64_t* __restrict__ off, const int32_t* __restrict__ nbr, const int64_t* __restrict__ win_row0,
    const int32_t* __restrict__ mask_ids, const int32_t* __restrict__ visit, const int32_t* __restrict__ visit_n,
    const uint8_t* __restrict__ win_final_empty, const uint32_t* __restrict__ prog_info, const unsigned long long* __restrict__ prog_adj,
    const int32_t* __restrict__ chain_win, const int64_t* __restrict__ chain_row0, const int64_t* __restrict__ chain_seed_begin,
    const int64_t* __restrict__ chain_slab0, const int32_t* __restrict__ slab, const int32_t* __restrict__ chain_list /* the chain windows of this launch */, int n_chain,
    const int64_t* __restrict__ col_off, const int32_t* __restrict__ col_idx, const uint8_t* __restrict__ col_code,
    const int32_t* __restrict__ col_pos, const int64_t* __restrict__ win_snp_first, const int64_t* __restrict__ win_snp_last,
    const int32_t* __restrict__ win_pos_lo, const int32_t* __restrict__ win_pos_hi, int finish_on_device,
    int lds_cap, int32_t* __restrict__ gscratch, const int64_t* __restrict__ gscratch_off,
    int32_t* __restrict__ labels3_out, int32_t* __restrict__ final_out, uint8_t* __restrict__ ok_out, unsigned long long* __restrict__ stat) {
    extern __shared__ int32_t tail_dyn[];    // [7 * lds_cap] (the doubles first: 8-byte aligned), then HS_TAIL_MAP_EXTRA bytes
    __shared__ int s_count[HS_FIN_KCAP], s_initial[HS_FIN_KCAP], s_tested[HS_FIN_KCAP];
    __shared__ int s_index_of[HS_FIN_KCAP], s_slot_of[HS_FIN_KCAP];
    __shared__ int s_glist[HS_FIN_GCAP], s_gidx[HS_FIN_GCAP];
    __shared__ int s_incompat[HS_FIN_GCAP * HS_FIN_GCAP];
    __shared__ int s_link_cnt[HS_FIN_MCAP * HS_FIN_MCAP], s_links_in[HS_FIN_MCAP], s_o2n[HS_FIN_MCAP], s_new_index[HS_FIN_MCAP];
    static_assert(HS_FIN_MCAP * HS_FIN_MCAP >= 256, "the deep-column histogram shares the link matrix's words");
    int* const s_cnts = s_link_cnt;      // (256-bin histogram of snp_deep: done with before the links are counted)
    __shared__ int s_scalar[8];
    __shared__ unsigned long long s_sets[64];
    __shared__ int s_relabel[64];
    __shared__ int s_lc1[HS_FIN_LCAP], s_lc2[HS_FIN_LCAP];
    __shared__ double s_lr[HS_FIN_LCAP];
    const int lane = lane_id();
    if ((int)blockIdx.x >= n_chain) return;
    const int c = chain_list[blockIdx.x];
    const int w = chain_win[c];
    const int64_t r0 = win_row0[w];
    const int m = (int)(win_row0[w + 1] - r0);
    const int32_t* __restrict__ ids = mask_ids + r0;
    const int64_t* __restrict__ off_w = off + r0;
    const int64_t abase = off_w[0];
    const int32_t* __restrict__ anb = nbr + abase;
    const int32_t* __restrict__ vis = visit + r0;
    const int n_visit = win_final_empty[w] ? 0 : visit_n[w];   // finalize_clustering may be handed an empty graph (:1708)
    const int K = (int)(chain_seed_begin[c + 1] - chain_seed_begin[c]);
    const int32_t* __restrict__ sl = slab + chain_slab0[c];
    const bool in_lds = m <= lds_cap;
    const int stride = in_lds ? lds_cap : m;
    int32_t* basep = in_lds ? tail_dyn : gscratch + gscratch_off[c];
    double* agg = reinterpret_cast<double*>(basep);          // [stride]
    int32_t* lab = basep + 2 * stride;
    int32_t* nc = lab + stride;
    int32_t* cnt = nc + stride;
    int32_t* t0 = cnt + stride;
    int32_t* t1 = t0 + stride;
    int32_t* __restrict__ l3 = labels3_out + chain_row0[c];
    int32_t* __restrict__ lout = final_out + chain_row0[c];
    auto bail = [&]() { if (lane == 0) ok_out[c] = 0; };
    unsigned long long sweeps = 0;

#ifdef HS_TAIL_DIAG
    unsigned long long tq0 = __builtin_readcyclecounter();
#define HS_TQ(k) { const unsigned long long tq1 = __builtin_readcyclecounter(); if (lane == 0 && stat) atomicAdd(&stat[18 + (k)], tq1 - tq0); tq0 = tq1; }
#else
#define HS_TQ(k)
#endif
    // ---- merge_clusterings ids (:840-874): the reference's double key sum_i label_i * 2^i, labels = read ids ----
    for (int j = lane; j < m; j += 64) t1[j] = ids[j];      // (the read ids out of LDS: the labels of the runs index them)
    wave_sync_lds();
    for (int j = lane; j < m; j += 64) {
        double a = 0.0, f = 1.0;
        int i = 0;
        for (; i + 16 <= K; i += 16) {      // sixteen labels in flight, added in the reference's order
            int lv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) lv[u] = sl[(int64_t)(i + u) * m + j];
#pragma unroll
            for (int u = 0; u < 16; ++u) { a += (double)t1[lv[u]] * f; f *= 2.0; }      // exact powers of two
        }
        for (; i + 4 <= K; i += 4) {      // four labels in flight, added in the reference's order
            const int l0 = sl[(int64_t)i * m + j], l1 = sl[(int64_t)(i + 1) * m + j], l2 = sl[(int64_t)(i + 2) * m + j], l3v = sl[(int64_t)(i + 3) * m + j];
            a += (double)t1[l0] * f; f *= 2.0; a += (double)t1[l1] * f; f *= 2.0; a += (double)t1[l2] * f; f *= 2.0; a += (double)t1[l3v] * f; f *= 2.0;      // exact powers of two
        }
        for (; i < K; ++i) { a += (double)t1[sl[(int64_t)i * m + j]] * f; f *= 2.0; }
        agg[j] = a; cnt[j] = 0;
    }
    wave_sync_lds();
    {
        // t0[j] = the first node with node j's key: an open-addressing table over nc|cnt (2 * stride ints) holds, per distinct key, the
        // smallest index seen so far -- any index in a slot carries the slot's key, so a probe compares against whichever is there
        int T = 1;
        while (T <= m) T <<= 1;      // m < T <= 2 m: an empty slot always exists
        int32_t* tab = nc;
        for (int x = lane; x < T; x += 64) tab[x] = -1;
        wave_sync_lds();
        for (int j = lane; j < m; j += 64) {
            const double a = agg[j];
            const unsigned long long bits = (unsigned long long)__double_as_longlong(a);
            unsigned h = ((unsigned)(bits >> 32) * 0x9E3779B1u ^ (unsigned)bits * 0x85EBCA6Bu);
            h = (h ^ (h >> 15)) & (unsigned)(T - 1);
            for (;;) {
                const int cur = atomicCAS(&tab[h], -1, j);
                if (cur == -1) break;
                if (agg[cur] == a) { atomicMin(&tab[h], j); break; }
                h = (h + 1u) & (unsigned)(T - 1);
            }
            t0[j] = (int)h;
        }
        wave_sync_lds();
        for (int j = lane; j < m; j += 64) t0[j] = tab[t0[j]];
        wave_sync_lds();
        for (int j = lane; j < m; j += 64) cnt[j] = 0;
        wave_sync_lds();
    }
    first_seen_ids_wave(t0, m, t1, lab, lane);
    HS_TQ(0)
    const bool masks = m <= 64 && prog_adj != nullptr;      // the window's neighbour masks exist (k_cw_visit_lists)
    wave_sync_lds();
    int Kc = 0;
    unsigned long long fin_S = 0ull, fin_adjn = 0ull;      // (narrow windows: the finished label sets and every node's neighbours, for the steps below)
    int fin_L = -1;
    if (masks) {
        // ---- at most 64 nodes: the labels stay in registers, as node -> label and as label -> set of nodes ----
        const unsigned long long adjv = lane < n_visit ? prog_adj[r0 + lane] : 0ull;
        const int nodev = lane < n_visit ? (int)(prog_info[r0 + lane] & 255u) : 0;
        int L = lane < m ? lab[lane] : -1;
        s_sets[lane] = 0ull;
        wave_sync_lds();
        if (lane < n_visit) s_sets[nodev] = adjv;
        wave_sync_lds();
        const unsigned long long adjn = s_sets[lane];      // node j's neighbours (none: not in the visiting list)
        wave_sync_lds();
        unsigned long long S = sets_from_labels(L, s_sets, lane);
        // run on the finalize graph (:881)
        sweeps += (unsigned long long)cw_run_sets(adjv, nodev, adjn, n_visit, S, L, lane);
        // clusters with fewer than 5 reads become -1, the others are renumbered by first appearance (:924-955)
        {
            const bool small = S != 0ull && __popcll(S) < 5;
            const unsigned long long gone = wave_or_u64(small ? S : 0ull);
            if (small) S = 0ull;
            if ((gone >> lane) & 1ull) L = -1;
            sets_renumber(S, L, s_sets, s_relabel, lane);
        }
        // run (:970)
        sweeps += (unsigned long long)cw_run_sets(adjv, nodev, adjn, n_visit, S, L, lane);
        if (lane < m) l3[lane] = L;
        if (lane == 0 && stat) {
            atomicAdd(&stat[0], sweeps);
            atomicAdd(&stat[1], sweeps * (4ull * (unsigned long long)(off_w[m] - abase) + 8ull * (unsigned long long)m));
        }
        if (!finish_on_device) { bail(); return; }
        HS_TQ(1)
        // first-seen renumbering (:973-984)
        Kc = sets_renumber(S, L, s_sets, s_relabel, lane);
        if (Kc > HS_FIN_KCAP) { bail(); return; }
        HS_TQ(2)
        // ---- merge_close_clusters (cluster_graph.cpp:402-501): every cluster in turn (in the order its first node appears) tries to
        // give its nodes away -- a node goes to the label most of its neighbours carry, or to the runner-up when it is at least half
        // as strong --; only a cluster that dissolves completely stays dissolved ----
        {
            unsigned long long N = S;      // the working copy of the sets (nc); S stays what is committed (lab)
            int Lw = L;
            unsigned long long tested = 0ull;
            for (int j = 0; j < m; ++j) {
                const int target = __builtin_amdgcn_readlane(L, j);
                if (target < 0 || ((tested >> target) & 1ull)) continue;
                int changes = 3, iters = 0;
                while (changes > 0 && iters < 10) {
                    changes = 0;
                    // where every node of the cluster would go under the current labels (lane = node): largest and runner-up among its
                    // neighbours' labels in ascending label order with strict '>' (:455-470: count desc, label asc); the sweep jumps to
                    // the first such node in visiting order, moves it and votes again
                    int p = 0;
                    while (p < n_visit) {
                        const unsigned long long alive = __ballot(N != 0ull);
                        int best = 0, second = 0;
                        for (unsigned long long am = alive; am; am &= am - 1ull) {
                            const int l = __builtin_ctzll(am);
                            const int c = __popcll(adjn & readlane_u64(N, l));
                            const int key = c > 0 ? ((c << 8) | (255 - l)) : 0;
                            if (key > best) { second = best; best = key; } else if (key > second) second = key;
                        }
                        const int max_value = best >> 8, max_index = best ? 255 - (best & 255) : 0;
                        const int second_value = second >> 8, second_index = second ? 255 - (second & 255) : 0;
                        int to = -1;
                        if (max_value > 0 && max_index != target) to = max_index;
                        else if (max_value > 0 && max_value <= 2 * second_value) to = second_index;
                        const unsigned long long members = readlane_u64(N, target);
                        const unsigned long long U = __ballot(((members >> lane) & 1ull) && to >= 0);                         // by node
                        const unsigned long long Uv = __ballot(lane >= p && lane < n_visit && ((U >> nodev) & 1ull));        // by visit, from p on
                        if (!Uv) break;
                        const int v0 = __builtin_ctzll(Uv);
                        const int i = __builtin_amdgcn_readlane(nodev, v0);
                        const int dst = __builtin_amdgcn_readlane(to, i);
                        const unsigned long long bit = 1ull << i;
                        if (lane == target) N &= ~bit;
                        if (lane == dst) N |= bit;
                        if (lane == i) Lw = dst;
                        changes++;
                        p = v0 + 1;
                    }
                    iters++;
                }
                const bool dissolved = readlane_u64(N, target) == 0ull;
                tested |= 1ull << target;
                if (dissolved) { S = N; L = Lw; } else { N = S; Lw = L; }
            }
        }
        if (lane < m) { lab[lane] = L; nc[lane] = L; }
        fin_S = S; fin_L = L; fin_adjn = adjn;
        wave_sync_lds();
    } else {
    // ---- wider windows: labels in LDS (or global scratch), neighbour lists ----
    // ---- run on the finalize graph (:881) ----
    sweeps += (unsigned long long)cw_local_wave(off_w, nbr, vis, n_visit, m, lab, cnt, lane);
    // ---- clusters with fewer than 5 reads become -1, the others are renumbered by first appearance (:924-955) ----
    for (int j = lane; j < m; j += 64) { nc[j] = 0; t1[j] = 0x7fffffff; }
    wave_sync_lds();
    for (int j = lane; j < m; j += 64) { const int l = lab[j]; if (l >= 0 && l < m) atomicAdd(&nc[l], 1); }
    wave_sync_lds();
    for (int j = lane; j < m; j += 64) {
        int l = lab[j];
        if (l < 0 || l >= m || nc[l] < 5) l = -1;
        t0[j] = l;
        if (l >= 0) atomicMin(&t1[l], j);
    }
    wave_sync_lds();
    for (int j = lane; j < m; j += 64) nc[j] = t0[j] >= 0 ? t1[t0[j]] : -1;     // first position of the node's label
    wave_sync_lds();
    first_seen_ids_wave(nc, m, t1, lab, lane);
    // ---- run (:970) ----
    wave_sync_lds();
    sweeps += (unsigned long long)cw_local_wave(off_w, nbr, vis, n_visit, m, lab, cnt, lane);
    for (int j = lane; j < m; j += 64) l3[j] = lab[j];
    if (lane == 0 && stat) {
        atomicAdd(&stat[0], sweeps);
        atomicAdd(&stat[1], sweeps * (4ull * (unsigned long long)(off_w[m] - abase) + 8ull * (unsigned long long)m));
    }
    if (!finish_on_device) { bail(); return; }

    HS_TQ(1)
    // ---- first-seen renumbering (:973-984): a label's new number = how many labels appear for the first time before it does ----
    for (int j = lane; j < m; j += 64) t1[j] = 0x7fffffff;
    wave_sync_lds();
    bool bad_l = false;
    for (int j = lane; j < m; j += 64) { const int l = lab[j]; if (l >= 0) { if (l >= m) bad_l = true; else atomicMin(&t1[l], j); } }
    wave_sync_lds();
    if (__ballot(bad_l) != 0ull) { bail(); return; }
    for (int b0 = 0; b0 < m; b0 += 64) {
        const int j = b0 + lane;
        int f = -1;
        if (j < m) { const int l = lab[j]; f = l >= 0 ? t1[l] : -1; nc[j] = f; }
        Kc += __popcll(__ballot(j < m && f == j));
    }
    wave_sync_lds();
    first_seen_ids_wave(nc, m, t1, lab, lane);
    if (Kc > HS_FIN_KCAP) { bail(); return; }

    HS_TQ(2)
    // ---- merge_close_clusters (cluster_graph.cpp:402-501) ----
    if (lane < HS_FIN_KCAP) { s_initial[lane] = 0; s_tested[lane] = 0; }
    wave_sync_lds();
    for (int j = lane; j < m; j += 64) { const int l = lab[j]; if (l >= 0) atomicAdd(&s_initial[l], 1); nc[j] = l; }
    wave_sync_lds();
    for (int j = 0; j < m; ++j) {
        const int target = lab[j];                      // uniform
        if (target < 0 || s_tested[target]) continue;
        if (lane < Kc) s_count[lane] = s_initial[lane];
        wave_sync_lds();
        int changes = 3, iters = 0;
        while (changes > 0 && iters < 10) {
            changes = 0;
            for (int k0 = 0; k0 < n_visit; k0 += 64) {
                const int kk = k0 + lane;
                int i_l = 0, o0_l = 0, o1_l = 0;
                if (kk < n_visit) { i_l = vis[kk]; o0_l = (int)(off_w[i_l] - abase); o1_l = (int)(off_w[i_l + 1] - abase); }
                unsigned long long act = __ballot(kk < n_visit && nc[i_l] == target);
                int nb_next = -1;      // the first 64 neighbours of the next node of the cluster: on their way while this one votes
                if (act) {
                    const int ln = __builtin_ctzll(act);
                    const int q0 = __builtin_amdgcn_readlane(o0_l, ln), q1 = __builtin_amdgcn_readlane(o1_l, ln);
                    if (q0 + lane < q1) nb_next = anb[q0 + lane];
                }
                while (act) {
                    const int l = __builtin_ctzll(act);
                    act &= act - 1ull;
                    const int i = __builtin_amdgcn_readlane(i_l, l);
                    const int o0 = __builtin_amdgcn_readlane(o0_l, l), o1 = __builtin_amdgcn_readlane(o1_l, l);
                    const int nb0 = nb_next;
                    nb_next = -1;
                    if (act) {
                        const int ln = __builtin_ctzll(act);
                        const int q0 = __builtin_amdgcn_readlane(o0_l, ln), q1 = __builtin_amdgcn_readlane(o1_l, ln);
                        if (q0 + lane < q1) nb_next = anb[q0 + lane];
                    }
                    // the votes of the neighbours' labels from ballots: lane L keeps the count of label L (Kc <= 16 labels)
                    int v = 0;
                    for (int ob = o0; ob < o1; ob += 64) {
                        int lb = -1;
                        if (ob == o0) { if (nb0 >= 0) lb = nc[nb0]; }
                        else if (ob + lane < o1) lb = nc[anb[ob + lane]];
                        unsigned long long rem = __ballot(lb >= 0);
                        while (rem) {
                            const int L = __builtin_amdgcn_readlane(lb, __builtin_ctzll(rem));
                            const unsigned long long msk = __ballot(lb == L);
                            rem &= ~msk;
                            if (lane == L) v += __popcll(msk);
                        }
                    }
                    // largest and runner-up in ascending label order with strict '>' (:455-470): (count desc, label asc)
                    const int key = v > 0 ? ((v << 8) | (255 - lane)) : 0;
                    const int best = wave_max_i32(key);
                    const int max_value = best >> 8, max_index = best ? 255 - (best & 255) : 0;
                    const int best2 = wave_max_i32((best && lane == max_index) ? 0 : key);
                    const int second_value = best2 >> 8, second_index = best2 ? 255 - (best2 & 255) : 0;
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
            for (int q = lane; q < m; q += 64) lab[q] = nc[q];
            if (lane < Kc) s_initial[lane] = s_count[lane];
        } else {
            for (int q = lane; q < m; q += 64) nc[q] = lab[q];
        }
        wave_sync_lds();
    }

    }

    HS_TQ(3)
    // ---- merge_wrongly_split_haplotypes (separate_reads.cpp:1007-1327) ----
    if (lane < HS_FIN_KCAP) { s_index_of[lane] = -1; s_slot_of[lane] = -1; }
    wave_sync_lds();
    if (masks) {      // lane = label: its place in the order of first appearance, its slot among the labels that are left
        const bool present = fin_S != 0ull;
        const int f = present ? __builtin_ctzll(fin_S) : 0;
        const unsigned long long F = wave_or_u64(present ? 1ull << f : 0ull);
        const unsigned long long pm = __ballot(present);
        const int index = __popcll(F & ((1ull << f) - 1ull)), slot = __popcll(pm & ((1ull << lane) - 1ull));
        if (present && lane < HS_FIN_KCAP) {
            s_index_of[lane] = index;
            if (slot < HS_FIN_GCAP) { s_slot_of[lane] = slot; s_glist[slot] = lane; s_gidx[slot] = index; }
        }
        if (lane == 0) s_scalar[2] = __popcll(pm);
    } else if (lane == 0) {
        int index = 0;
        for (int j = 0; j < m; ++j) { const int cl = lab[j]; if (cl > -1 && s_index_of[cl] < 0) s_index_of[cl] = index++; }
        int G = 0;
        for (int l = 0; l < Kc; ++l) if (s_index_of[l] >= 0) { if (G < HS_FIN_GCAP) { s_slot_of[l] = G; s_glist[G] = l; s_gidx[G] = s_index_of[l]; } G++; }
        s_scalar[2] = G;
    }
    wave_sync_lds();
    const int G = s_scalar[2];
    if (G <= 1) {
        for (int j = lane; j < m; j += 64) lout[j] = 0;
        if (lane == 0) ok_out[c] = 1;
        return;
    }
    if (G > HS_FIN_GCAP) { bail(); return; }
    for (int x = lane; x < G * G; x += 64) s_incompat[x] = 0;
    wave_sync_lds();
    const int pos_lo = win_pos_lo[c], pos_hi = win_pos_hi[c];
    for (int j = lane; j < m; j += 64) t0[j] = ids[j];          // the read ids next to the labels: the look-ups below stay off global memory
    wave_sync_lds();
    const int32_t* ids_l = t0;
    // read id -> cluster slot as a byte map over the window's id range (the `agg` doubles are free by now), when the range fits:
    // one LDS byte per column entry instead of a bisection of the id list
    const int id_lo = m > 0 ? ids_l[0] : 0;
    const long long id_range = m > 0 ? (long long)ids_l[m - 1] - id_lo + 1 : 0;
    // (the map lies over the `agg` doubles, free by now, or in the extra bytes behind the arrays, whichever is larger)
    const long long map_cap = in_lds ? (8ll * stride > HS_TAIL_MAP_EXTRA ? 8ll * stride : (long long)HS_TAIL_MAP_EXTRA) : 0ll;
    const bool use_map = in_lds && id_range <= map_cap;
    uint8_t* smap = 8ll * stride > HS_TAIL_MAP_EXTRA ? reinterpret_cast<uint8_t*>(agg) : reinterpret_cast<uint8_t*>(tail_dyn + 7 * lds_cap);
    if (use_map) {
        for (int x = lane; x < (int)id_range; x += 64) smap[x] = 0xff;
        wave_sync_lds();
        for (int j = lane; j < m; j += 64) { const int cl = lab[j]; if (cl > -1) smap[ids_l[j] - id_lo] = (uint8_t)s_slot_of[cl]; }
        wave_sync_lds();
    }
    HS_TQ(5)
    // lane a * G + b keeps the pair of clusters (a, b) with label(a) > label(b): its count of incompatible SNPs and the position of
    // the last one (both entries of the symmetric tables of :1113-1135 always move together)
    const int pa = lane < G * G ? lane / G : 0, pb = lane < G * G ? lane % G : 0;
    const bool pair_owner = lane < G * G && s_glist[pa] > s_glist[pb];
    int pair_inc = 0, pair_last = -10;
#ifdef HS_TAIL_DIAG
    unsigned long long dq_fast = 0, dq_slow = 0, dq_keys = 0;
#endif
    // what the majority bases of a SNP at position p do to the pairs (:1100-1135). Clusters absent at the SNP keep majority 0, which is
    // not ' ': they take part in the comparison (sic), but they do not count as a "max base" for the `several` test (:1100-1112 only
    // inserts bases of clusters that carry reads)
    auto pairs_update = [&](int p, int major) {
        const bool counts_l = lane < G && major != 0 && major != ' ';
        const unsigned long long cm = __ballot(counts_l);
        bool several = false;
        if (cm) { const int first_max = __builtin_amdgcn_readlane(major, __builtin_ctzll(cm)); several = __ballot(counts_l && major != first_max) != 0ull; }
        if (several) {
            const int ma = __shfl(major, pa, 64), mb = __shfl(major, pb, 64);
            if (pair_owner && ma != ' ' && mb != ' ' && ma != mb && p - pair_last > 10) { pair_inc += 1; pair_last = p; }
        }
    };
    // a column of at most 64 entries, lane = entry: (largest count, runner-up count, total) of every cluster's bases from ballots, key =
    // (slot, code); lane i keeps cluster slot i's three numbers. A tied maximum yields runner-up == maximum (:1090-1099).
    auto snp_fast = [&](int p, int n, int idx, int code) {
        int slot = 0xff;
        if (lane < n) {
            if (use_map) { const long long d = (long long)idx - id_lo; slot = (d >= 0 && d < id_range) ? (int)smap[d] : 0xff; }
            else { const int j = local_index(ids_l, m, idx); const int cl = j >= 0 ? lab[j] : -2; slot = cl > -1 ? s_slot_of[cl] : 0xff; }
        }
        const int key = slot != 0xff ? ((slot << 8) | code) : -1;
        unsigned long long rem = __ballot(key >= 0);
        int t1v = 0, t2v = 0, c1 = -1, nbv = 0;
#ifdef HS_TAIL_DIAG
        dq_fast++;
#endif
        while (rem) {
#ifdef HS_TAIL_DIAG
            dq_keys++;
#endif
            const int k = __builtin_amdgcn_readlane(key, __builtin_ctzll(rem));
            const unsigned long long msk = __ballot(key == k);
            rem &= ~msk;
            const int v = __popcll(msk);
            if (lane == (k >> 8)) { nbv += v; if (v > t1v) { t2v = t1v; t1v = v; c1 = k & 255; } else if (v > t2v) t2v = v; }
        }
        int major = 0;      // lane i < G: the majority base of cluster slot i at this SNP (0: absent, ' ': none)
        if (lane < G && t1v > 0) { major = c1; if (t2v * 2 > t1v || nbv * 0.5 > t1v) major = ' '; }
        pairs_update(p, major);
    };
    // a column deeper than a wavefront (rare): one cluster at a time over a 256-bin LDS histogram, the column read once per cluster
    auto snp_deep = [&](int p, int64_t e0, int n) {
#ifdef HS_TAIL_DIAG
        dq_slow++;
#endif
        int major = 0;
        for (int i = 0; i < G; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s_cnts[lane + 64 * q] = 0;
            wave_sync_lds();
            int nb_i = 0;
            for (int64_t eb = e0; eb < e0 + n; eb += 64) {
                const int64_t e = eb + lane;
                bool mine = false;
                if (e < e0 + n) {
                    const int j = local_index(ids_l, m, col_idx[e]);
                    const int cl = j >= 0 ? lab[j] : -2;
                    mine = cl > -1 && s_slot_of[cl] == i;
                    if (mine) atomicAdd(&s_cnts[col_code[e]], 1);
                }
                nb_i += __popcll(__ballot(mine));
            }
            wave_sync_lds();
            int t1v = 0, c1 = -1, t2v = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cd = lane + 64 * q;
                const int v = s_cnts[cd];
                if (v > t1v) { t2v = t1v; t1v = v; c1 = cd; } else if (v > t2v) t2v = v;
            }
            wave_sync_lds();
            // one reduction gives the maximum and, among the lanes that hold it, the largest of their codes (counts < 2^23)
            const int top = wave_max_i32((t1v << 8) | (c1 & 255));
            const int mx = top >> 8;
            if (mx == 0) continue;                              // cluster absent at this SNP: majority stays 0
            const int n_at = __popcll(__ballot(t1v == mx)) + __popcll(__ballot(t2v == mx));
            const int below = wave_max_i32(t1v == mx ? t2v : t1v);  // best count strictly below the maximum when it is unique
            const int second_max = n_at >= 2 ? mx : below;
            int max_base = top & 255;
            if (second_max * 2 > mx || nb_i * 0.5 > mx) max_base = ' ';
            if (lane == i) major = max_base & 255;
        }
        pairs_update(p, major);
    };
    for (int64_t sb = win_snp_first[c]; sb < win_snp_last[c]; sb += 64) {
        // 64 SNPs' position, first entry and depth at once; then eight SNPs per round: their entries are all requested before the first
        // of them is counted (one memory round trip per eight columns)
        const int64_t s_l = sb + lane;
        const bool s_valid = s_l < win_snp_last[c];
        const int p_l = s_valid ? col_pos[s_l] : 0;
        const int64_t e0_l = s_valid ? col_off[s_l] : 0;
        const int n_l = s_valid ? (int)(col_off[s_l + 1] - e0_l) : 0;
        unsigned long long todo = __ballot(s_valid && p_l >= pos_lo && p_l < pos_hi);
        auto rl_e0 = [&](int l) { return ((int64_t)(unsigned)__builtin_amdgcn_readlane((int)(e0_l >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(e0_l & 0xffffffffll), l); };
        while (todo) {
            int aq[HS_TAIL_BATCH], idx_b[HS_TAIL_BATCH], code_b[HS_TAIL_BATCH];
            bool any_deep = false;
            const unsigned long long batch_todo = todo;
#pragma unroll
            for (int q = 0; q < HS_TAIL_BATCH; ++q) {
                aq[q] = -1; idx_b[q] = -1; code_b[q] = 0;
                if (todo) {
                    aq[q] = __builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    const int64_t e0 = rl_e0(aq[q]);
                    const int n = __builtin_amdgcn_readlane(n_l, aq[q]);
                    any_deep = any_deep || n > 64;
                    if (lane < n) { idx_b[q] = col_idx[e0 + lane]; code_b[q] = col_code[e0 + lane]; }
                }
            }
            if (!any_deep) {
#pragma unroll
                for (int q = 0; q < HS_TAIL_BATCH; ++q) {
                    if (aq[q] < 0) break;
                    snp_fast(__builtin_amdgcn_readlane(p_l, aq[q]), __builtin_amdgcn_readlane(n_l, aq[q]), idx_b[q], code_b[q]);
                }
            } else {
                // (a batch with a deep column: one SNP after the other, in position order)
                unsigned long long bt = batch_todo & ~todo;
                while (bt) {
                    const int a = __builtin_ctzll(bt);
                    bt &= bt - 1ull;
                    const int p = __builtin_amdgcn_readlane(p_l, a), n = __builtin_amdgcn_readlane(n_l, a);
                    const int64_t e0 = rl_e0(a);
                    if (n <= 64) { int idx = -1, code = 0; if (lane < n) { idx = col_idx[e0 + lane]; code = col_code[e0 + lane]; } snp_fast(p, n, idx, code); }
                    else snp_deep(p, e0, n);
                }
            }
        }
    }
#ifdef HS_TAIL_DIAG
    if (lane == 0 && stat) { atomicAdd(&stat[18 + 8], dq_fast); atomicAdd(&stat[18 + 9], dq_slow); atomicAdd(&stat[18 + 10], dq_keys); atomicAdd(&stat[18 + 11], use_map ? 1ull : 0ull); atomicAdd(&stat[18 + 12], (unsigned long long)G); }
#endif
    if (pair_owner) { const int i1 = s_gidx[pa], i2 = s_gidx[pb]; s_incompat[i1 * G + i2] = pair_inc; s_incompat[i2 * G + i1] = pair_inc; }
    wave_sync_lds();
    HS_TQ(6)
    // link ratios (:1189-1250): a dense (label + 2) x (label + 2) count matrix walked in ascending key order
    const int M = Kc + 2;
    for (int x = lane; x < M * M; x += 64) s_link_cnt[x] = 0;
    if (lane < M) s_links_in[lane] = 0;
    wave_sync_lds();
    if (masks) {
        // lane = node q with its neighbour mask: per label c1 (and for the neighbours without a label) how many of q's neighbours carry it
        const int c2 = fin_L + 2;
        const unsigned long long any = wave_or_u64(fin_S);
        const unsigned long long nolabel = ~any & (m >= 64 ? ~0ull : ((1ull << m) - 1ull));
        {
            const int k = __popcll(fin_adjn & nolabel);
            if (k > 0) { if (c2 != 1) atomicAdd(&s_link_cnt[1 * M + c2], k); atomicAdd(&s_links_in[1], k); }
        }
        for (unsigned long long am = __ballot(fin_S != 0ull); am; am &= am - 1ull) {
            const int l = __builtin_ctzll(am);
            const int k = __popcll(fin_adjn & readlane_u64(fin_S, l));
            if (k > 0) { if (l + 2 != c2) atomicAdd(&s_link_cnt[(l + 2) * M + c2], k); atomicAdd(&s_links_in[l + 2], k); }
        }
    } else if (n_visit > 0) {   // (an empty finalize graph has no links)
        for (int q = lane; q < m; q += 64) {
            const int c2 = lab[q] + 2;
            for (int64_t o = off_w[q]; o < off_w[q + 1]; ++o) {
                const int c1 = lab[nbr[o]] + 2;
                if (c1 != c2) atomicAdd(&s_link_cnt[c1 * M + c2], 1);
                atomicAdd(&s_links_in[c1], 1);
            }
        }
    }
    wave_sync_lds();
    HS_TQ(7)
    // the links in ascending (c1, c2) order: the non-zero cells of the count matrix, ranked by a ballot prefix
    {
        int nl_all = 0;
        for (int x0 = 0; x0 < M * M; x0 += 64) {
            const int x = x0 + lane;
            const int cv = x < M * M ? s_link_cnt[x] : 0;
            const unsigned long long nz = __ballot(cv > 0);
            const int rank = nl_all + __popcll(nz & ((1ull << lane) - 1ull));
            if (cv > 0 && rank < HS_FIN_LCAP) { const int c1 = x / M, c2 = x % M; s_lc1[rank] = c1 - 2; s_lc2[rank] = c2 - 2; s_lr[rank] = (double)cv / s_links_in[c1]; }
            nl_all += __popcll(nz);
        }
        if (lane == 0) s_scalar[4] = nl_all;
    }
    wave_sync_lds();
    if (lane == 0) {
        int* lc1 = s_lc1; int* lc2 = s_lc2; double* lr = s_lr;      // (indexed at run time: LDS, not registers)
        const int nl = s_scalar[4] > HS_FIN_LCAP ? HS_FIN_LCAP : s_scalar[4];
        const bool over = s_scalar[4] > HS_FIN_LCAP;
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
    for (int j = lane; j < m; j += 64) lout[j] = s_o2n[lab[j] + 2];
    if (lane == 0) ok_out[c] = 1;
    HS_TQ(4)
}

}  // namespace hsdev
