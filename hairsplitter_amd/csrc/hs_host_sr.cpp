// hs_host_sr.cpp -- sequential glue of stage 4 (HS_separate_reads).
//
// Device work: the read x read similarity/difference counts (k_simdiff) and every Chinese-Whispers run
// (k_chinese_whispers), batched over all windows of all contigs in three waves (per-SNP runs, the merged
// clustering, the re-clustering after small clusters are dropped). Host work, here: window/mask planning
// (separate_reads.cpp:1548-1622), the per-row neighbour selection whose tie order is std::sort's
// (:769-815), cluster bookkeeping between the waves (:840-885, :924-989), merge_close_clusters
// (cluster_graph.cpp:402-501) and merge_wrongly_split_haplotypes (separate_reads.cpp:1007-1327).
#include "hs_host_sr.h"
#include "hs_rh8.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <atomic>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <random>
#include <set>
#include <unordered_map>

namespace hs {

std::vector<int32_t> shuffled_order(int n, uint32_t seed) {
    // cluster_graph.cpp:173-177 / :254-258 / :428-432 with std::random_device pinned to `seed`
    std::vector<int32_t> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::mt19937 g(seed);
    std::shuffle(order.begin(), order.end(), g);
    return order;
}

// neighbour selection shared by both graph builders: separate_reads.cpp:769-815 (== :633-670)
static void pick_neighbors_sorted(std::vector<std::pair<int, float>>& smallest, const uint8_t* mask, float error_rate, std::vector<int>& picked) {
    std::sort(smallest.begin(), smallest.end(), [](const std::pair<int, float>& a, const std::pair<int, float>& b) { return a.second > b.second; });
    int nb = 0;
    const float below = 1 - error_rate * 2;
    float above = 1;
    if (smallest.size() > 1) above = smallest[0].second - (smallest[0].second - smallest[1].second) * 3;
    if (above == 1) {
        int idx = 0;
        while (idx < (int)smallest.size() && smallest[idx].second == 1) idx += 1;
        if (idx < (int)smallest.size()) { idx = std::min(idx + 4, (int)smallest.size() - 1); above = smallest[idx].second; }
    }
    picked.clear();
    for (const auto& s : smallest) {
        if (s.second > below && (nb < 5 || s.second == 1 || s.second >= above) && mask[s.first]) { nb++; picked.push_back(s.first); }
    }
}

static void to_csr(std::vector<std::vector<int>>& lists, SrGraph& g) {
    const int N = (int)lists.size();
    g.off.assign((size_t)N + 1, 0);
    g.adj.clear();
    for (int i = 0; i < N; ++i) {
        auto& v = lists[i];
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        g.off[i + 1] = g.off[i] + (int)v.size();
        g.adj.insert(g.adj.end(), v.begin(), v.end());
    }
}

// One row of create_read_graph_matrix (separate_reads.cpp:745-815) exactly as the reference does it. The device (K6) builds
// the graphs; it hands back the rare rows where fewer than five neighbours qualify by value and the run of equal distances
// at the cut-off is only partly taken, i.e. where std::sort's arrangement of equal keys decides.
void sr_pick_row_sorted(const int32_t* srow, const int32_t* drow, int N, int r1, const uint8_t* mask, float error_rate, std::vector<int>& picked) {
    std::vector<std::pair<int, float>> smallest((size_t)N);
    int max_compat = 0;
    for (int r = 0; r < N; ++r) {
        float d = 0;
        if (mask[r] && r != r1 && srow[r] > 0) {
            const float df = (float)std::max(0, drow[r] - 1);
            d = 1 - df / float(srow[r] + drow[r]);
            if (srow[r] > max_compat) max_compat = srow[r];
        }
        smallest[r] = std::make_pair(r, d);
    }
    for (int r = 0; r < N; ++r)
        if (mask[r] && r != r1 && srow[r] + drow[r] < 0.7 * max_compat) smallest[r].second = 0;
    pick_neighbors_sorted(smallest, mask, error_rate, picked);
}

void sr_set_window_graph(SrContigState& st, int window, const int32_t* ids, int m, const int64_t* nbr_off, const int32_t* nbr) {
    SrWindowPlan& w = st.windows[(size_t)window];
    SrGraph& g = st.graphs[(size_t)w.graph_now];
    g.off.assign((size_t)st.N + 1, 0);
    g.adj.assign(nbr + nbr_off[0], nbr + nbr_off[m]);
    int j = 0;
    for (int r = 0; r < st.N; ++r) {
        g.off[(size_t)r] = (int)(j < m ? nbr_off[j] - nbr_off[0] : nbr_off[m] - nbr_off[0]);
        if (j < m && ids[j] == r) j++;
    }
    g.off[(size_t)st.N] = (int)(nbr_off[m] - nbr_off[0]);
}

// create_read_graph_low_memory: separate_reads.cpp:538-693
static void build_graph_low_memory(const SrContigState& st, const uint8_t* mask, float error_rate, SrGraph& g) {
    const hs_sr_contig& c = *st.c;
    const int N = st.N;
    std::vector<int> first((size_t)N, -1);
    std::vector<std::vector<uint8_t>> val((size_t)N);
    for (int s = 0; s < c.n_snps; ++s)
        for (int64_t e = c.col_off[s]; e < c.col_off[s + 1]; ++e) {
            const int r = c.col_idx[e];
            if (first[r] == -1) first[r] = s;
            val[r].push_back(c.col_code[e] == c.snp_ref[s] ? 1 : (c.col_code[e] == c.snp_alt[s] ? 2 : 0));
        }
    std::vector<uint8_t> ext((size_t)N, 0);
    for (int r = 0; r < N; ++r) ext[r] = mask[r] && first[r] != -1;
    std::vector<std::vector<int>> lists((size_t)N);
    std::vector<std::pair<int, float>> smallest((size_t)N);
    std::vector<int> simv((size_t)N), difv((size_t)N), picked;
    for (int r1 = 0; r1 < N; ++r1) {
        if (!ext[r1]) continue;
        int max_compat = 0;
        for (int r = 0; r < N; ++r) { smallest[r] = std::make_pair(r, 0.0f); simv[r] = 0; difv[r] = 0; }
        for (int r2 = 0; r2 < N; ++r2) {
            if (!(ext[r2] && r1 != r2)) continue;
            int nsim = 0, ndif = 0;
            const long a = std::max(first[r1], first[r2]);
            const long b = std::min((long)val[r1].size() + first[r1] - 1, (long)val[r2].size() + first[r2] - 1);
            for (long p = a; p <= b; ++p) {
                const int v1 = val[r1][p - first[r1]], v2 = val[r2][p - first[r2]];
                if (v1 == 2 && v2 == 2) nsim += 3; else if (v1 == 1 && v2 == 1) nsim++; else if (v1 != 0 && v2 != 0) ndif++;
            }
            smallest[r2].second = 1 - std::max(0, ndif - 1) / float(ndif + nsim);
            if (nsim > max_compat) max_compat = nsim;
            simv[r2] = nsim; difv[r2] = ndif;
        }
        for (int r = 0; r < N; ++r)
            if (mask[r] && r != r1 && simv[r] + difv[r] < 0.7 * max_compat) smallest[r].second = 0;
        pick_neighbors_sorted(smallest, mask, error_rate, picked);   // 0/0 distances (NaN) exist on this path
        for (int nb : picked) { lists[r1].push_back(nb); lists[nb].push_back(r1); }
    }
    to_csr(lists, g);
}

// Window / mask planning: separate_reads.cpp:1545-1622 (the running SNP cursor is carried across windows)
void sr_plan_windows(SrContigState& st, int window_size, float error_rate, bool low_memory) {
    (void)error_rate;
    const hs_sr_contig& c = *st.c;
    const int N = st.N;
    const long L = c.length;
    st.windows.clear();
    st.graphs.clear();
    st.empty_graph = -1;
    st.snp_pos_sorted = std::is_sorted(c.snp_pos, c.snp_pos + c.n_snps);
    if (c.n_snps == 0) return;
    int cur = 0, chunk = -1, upper;
    while ((long)(chunk + 1) * window_size + 100 <= L) {
        chunk++;
        upper = (chunk + 1) * window_size;
        const bool last = (long)(chunk + 1) * window_size + 100 > L;
        if (last) upper = (int)L + 1;
        SrWindowPlan w;
        w.start = chunk * window_size;
        w.end = std::min(upper - 1, (int)L);
        if (cur >= c.n_snps || c.snp_pos[cur] > upper - 1) {          // :1565-1587
            w.has_snps = false;
            w.labels.assign((size_t)N, -2);
            int mid = (w.start + std::min(upper - 1, (int)L)) / 2;
            if (mid < 500) mid = std::min(500, (int)(L / 2));
            if (mid > (int)L - 500) mid = std::max((int)(L / 2), (int)L - 500);
            for (int r = 0; r < N; ++r) if (c.read_start[r] <= mid && c.read_end[r] >= mid) w.labels[r] = 0;
            st.windows.push_back(std::move(w));
            continue;
        }
        w.has_snps = true;
        w.mask.assign((size_t)N, 0);
        if (chunk == 0) {
            while (cur < c.n_snps - 1 && c.snp_pos[cur] < chunk * window_size + 0.2 * window_size
                   && c.snp_pos[cur + 1] < chunk * window_size + 0.4 * window_size) cur++;
        }
        for (int64_t e = c.col_off[cur]; e < c.col_off[cur + 1]; ++e) w.mask[c.col_idx[e]] = 1;
        while (cur < c.n_snps && c.snp_pos[cur] < upper - 1) cur++;
        if (cur > 0) cur--;
        if (last) {
            while (cur > 0 && c.snp_pos[cur] > upper - 1 - 0.2 * window_size && c.snp_pos[cur - 1] > upper - 1 - 0.4 * window_size) cur--;
        }
        int idxmask = 0;
        for (int64_t e = c.col_off[cur]; e < c.col_off[cur + 1]; ++e) {
            while (idxmask < c.col_idx[e]) { w.mask[idxmask] = 0; idxmask++; }
            idxmask++;
        }
        cur++;
        for (int r = 0; r < N; ++r) if (w.mask[(size_t)r]) w.mask_ids.push_back(r);
        // graph slot for this window (filled by sr_build_window_graph, one independent task per window)
        st.graphs.emplace_back();
        st.graphs.back().off.assign((size_t)N + 1, 0);
        w.graph_now = (int)st.graphs.size() - 1;
        // finalize_clustering is handed the *global* low_memory flag (:1708): with low_memory_now && !low_memory it
        // sees an Eigen matrix that was never filled
        if (st.low_memory_now && !low_memory) {
            st.graphs.emplace_back();
            st.graphs.back().off.assign((size_t)N + 1, 0);
            w.graph_final = (int)st.graphs.size() - 1;
        } else w.graph_final = w.graph_now;
        // SNPs seeding a local run (:1673-1676): inside [start, start+window) and more than 10 bp apart
        int lastpos = -10;
        for (int s = 0; s < c.n_snps; ++s) {
            const int p = c.snp_pos[s];
            if (p >= chunk * window_size && p < chunk * window_size + window_size && p > lastpos + 10) { lastpos = p; w.local_snps.push_back(s); }
        }
        w.final_lo = chunk * window_size; w.final_hi = chunk * window_size + window_size;
        st.windows.push_back(std::move(w));
    }
}

void sr_build_window_graph(SrContigState& st, int window, float error_rate) {
    SrWindowPlan& w = st.windows[(size_t)window];
    if (!w.has_snps) return;
    SrGraph& g = st.graphs[(size_t)w.graph_now];
    if (st.low_memory_now) build_graph_low_memory(st, w.mask.data(), error_rate, g);   // the matrix path is K6 (device)
}

// Per-thread scratch of the two cluster-merging steps: they run once per clustering window (tens of thousands of calls per
// batch), so nothing in them allocates, and every table is indexed by cluster label (a handful) instead of by read.
struct MergeScratch {
    std::vector<int32_t> nc, order_masked, masked;
    std::vector<int> initial, count, votes, touched;
    std::vector<char> tested;
    // merge_wrongly_split
    std::vector<int> glist, gidx, index_of, slot_of, incompat, pos_last, nb_bases, majority, link_cnt, links_in, o2n, new_index;
    std::vector<std::vector<int>> cnts;          // [slot][256], all zero between uses
    std::vector<std::vector<uint8_t>> seen;
    std::vector<std::pair<std::pair<int, int>, double>> sorted_links;
};
static MergeScratch& merge_scratch() { static thread_local MergeScratch s; return s; }

// merge_close_clusters: cluster_graph.cpp:402-501. `n_labels`: cluster labels are 0 .. n_labels-1 (first-seen numbering).
static void merge_close_clusters(const SrGraph& g, bool low_memory, std::vector<int32_t>& clusters, const uint8_t* mask,
                                 const std::vector<int32_t>& order, int n_labels) {
    MergeScratch& S = merge_scratch();
    const int K = n_labels > 0 ? n_labels : 1;
    S.order_masked.clear();                                          // S.masked: the window's reads, ascending (set by the caller)
    for (int i : order) if (mask[i]) S.order_masked.push_back(i);   // the shuffled order restricted to the window's reads
    S.initial.assign((size_t)K, 0); S.votes.assign((size_t)K, 0); S.tested.assign((size_t)K, 0);
    for (int r : S.masked) if (clusters[r] >= 0) S.initial[(size_t)clusters[r]] += 1;   // reads outside the mask carry -2
    S.nc.assign(clusters.begin(), clusters.end());
    std::vector<int32_t>& nc = S.nc;
    std::vector<int>& votes = S.votes;
    std::vector<int>& touched = S.touched;
    for (int node : S.masked) {
        if (!(clusters[node] >= 0 && !S.tested[(size_t)clusters[node]])) continue;
        const int target = clusters[node];
        S.count = S.initial;
        int changes = 3, iters = 0;
        while (changes > 0 && iters < 10) {
            changes = 0;
            for (int i : S.order_masked) {
                if (nc[i] != target) continue;
                touched.clear();
                const int o0 = g.off[i], o1 = g.off[i + 1];
                if (low_memory) {   // :441-445 iterates j < degree and asks whether j itself is a neighbour (sic)
                    for (int j = 0; j < o1 - o0; ++j)
                        if (std::binary_search(g.adj.begin() + o0, g.adj.begin() + o1, j) && nc[j] >= 0) { if (votes[nc[j]]++ == 0) touched.push_back(nc[j]); }
                } else {
                    for (int o = o0; o < o1; ++o) { const int l = nc[g.adj[o]]; if (l >= 0) { if (votes[l]++ == 0) touched.push_back(l); } }
                }
                // largest and runner-up in ascending label order with strict '>' (:455-470)
                std::sort(touched.begin(), touched.end());
                int max_index = 0, max_value = 0, second_index = 0, second_value = 0;
                for (int l : touched) {
                    const int v = votes[l];
                    if (v > max_value) { second_value = max_value; second_index = max_index; max_value = v; max_index = l; }
                    else if (v > second_value) { second_value = v; second_index = l; }
                }
                for (int l : touched) votes[l] = 0;
                if (max_value > 0 && max_index != target) { S.count[nc[i]]--; S.count[max_index]++; changes++; nc[i] = max_index; }
                else if (max_value > 0 && max_value <= 2 * second_value) { S.count[nc[i]]--; S.count[second_index]++; nc[i] = second_index; changes++; }
            }
            iters++;
        }
        S.tested[(size_t)target] = 1;
        if (S.count[(size_t)target] == 0) { for (int r : S.masked) clusters[r] = nc[r]; S.initial = S.count; }   // the cluster dissolved: keep
        else for (int r : S.masked) nc[r] = clusters[r];                                                          // undo
    }
}

// merge_wrongly_split_haplotypes: separate_reads.cpp:1007-1327. `n_labels` as above.
static std::vector<int32_t> merge_wrongly_split(const SrContigState& st, const std::vector<int32_t>& clustered, const SrGraph& g,
                                                bool low_memory, int posstart, int posend, int n_labels) {
    MergeScratch& S = merge_scratch();
    const hs_sr_contig& c = *st.c;
    const int N = st.N;
    const int K = n_labels > 0 ? n_labels : 1;
    // clusters present, ascending (glist), and their first-seen rank over the reads (index_of), as the std::set / std::map give
    S.index_of.assign((size_t)K, -1); S.slot_of.assign((size_t)K, -1);
    int index = 0;
    for (int r : S.masked) { const int cl = clustered[r]; if (cl > -1 && S.index_of[(size_t)cl] < 0) S.index_of[(size_t)cl] = index++; }   // reads outside the mask carry -2
    S.glist.clear();
    for (int l = 0; l < K; ++l) if (S.index_of[(size_t)l] >= 0) { S.slot_of[(size_t)l] = (int)S.glist.size(); S.glist.push_back(l); }
    const int G = (int)S.glist.size();
    if (G <= 1) {
        std::vector<int32_t> one((size_t)N, 0);
        for (int r = 0; r < N; ++r) if (clustered[r] == -2) one[r] = -2;
        return one;
    }
    const std::vector<int>& glist = S.glist;
    S.gidx.resize((size_t)G);
    for (int i = 0; i < G; ++i) S.gidx[(size_t)i] = S.index_of[(size_t)glist[(size_t)i]];
    S.incompat.assign((size_t)G * G, 0); S.pos_last.assign((size_t)G * G, -10);
    if ((int)S.cnts.size() < G) { S.cnts.resize((size_t)G, std::vector<int>(256, 0)); S.seen.resize((size_t)G); }
    S.nb_bases.assign((size_t)G, 0); S.majority.assign((size_t)G, 0);
    std::vector<int>& incompat = S.incompat; std::vector<int>& pos_last = S.pos_last;
    std::vector<int>& nb_bases = S.nb_bases; std::vector<int>& majority = S.majority;   // 0 == operator[] default for clusters absent at a SNP
    // SNP positions ascend (call_variants writes them in position order; parse_col keeps file order): binary search if so
    int s_first = 0, s_last = c.n_snps;
    if (st.snp_pos_sorted) {
        s_first = (int)(std::lower_bound(c.snp_pos, c.snp_pos + c.n_snps, posstart) - c.snp_pos);
        s_last = (int)(std::lower_bound(c.snp_pos, c.snp_pos + c.n_snps, posend) - c.snp_pos);
    }
    for (int s = s_first; s < s_last; ++s) {
        const int p = c.snp_pos[s];
        if (!(p >= posstart && p < posend)) continue;
        for (int i = 0; i < G; ++i) { nb_bases[i] = 0; majority[i] = 0; }
        for (int64_t e = c.col_off[s]; e < c.col_off[s + 1]; ++e) {
            const int cl = clustered[c.col_idx[e]];
            if (cl > -1) {
                const int sl = S.slot_of[(size_t)cl];
                const uint8_t b = c.col_code[e];
                if (S.cnts[(size_t)sl][b]++ == 0) S.seen[(size_t)sl].push_back(b);
                nb_bases[sl]++;
            }
        }
        // The reference walks the cluster's base counts in robin_hood order keeping (max, second max) with `>=` on the max
        // (:1090-1099). The pair of values does not depend on the order, and neither does the verdict: a unique maximum names
        // the base, a tied maximum gives second == max and is rejected by `second_max * 2 > max` just below.
        int first_max = -1; bool several = false;
        for (int i = 0; i < G; ++i) {
            if (S.seen[(size_t)i].empty()) continue;
            int second_max = 0, mx = 0;
            int max_base = ' ';
            for (uint8_t b : S.seen[(size_t)i]) {
                const int v = S.cnts[(size_t)i][b];
                if (v >= mx) { max_base = (int)(signed char)b; second_max = mx; mx = v; }
                else if (v > second_max) second_max = v;
                S.cnts[(size_t)i][b] = 0;
            }
            S.seen[(size_t)i].clear();
            if (second_max * 2 > mx || nb_bases[i] * 0.5 > mx) max_base = ' ';
            majority[i] = (int)(uint8_t)max_base;
            if (max_base != ' ') { const int mb = (int)(uint8_t)max_base; if (first_max < 0) first_max = mb; else if (mb != first_max) several = true; }
        }
        if (!several) continue;
        for (int a = 0; a < G; ++a)
            for (int b = 0; b < G; ++b) {
                if (majority[a] != ' ' && majority[b] != ' ' && glist[(size_t)a] > glist[(size_t)b]) {
                    const int i1 = S.gidx[(size_t)a], i2 = S.gidx[(size_t)b];
                    if (majority[a] != majority[b] && p - pos_last[(size_t)i1 * G + i2] > 10) {
                        incompat[(size_t)i1 * G + i2] += 1; incompat[(size_t)i2 * G + i1] += 1;
                        pos_last[(size_t)i1 * G + i2] = p; pos_last[(size_t)i2 * G + i1] = p;
                    }
                }
            }
    }
    // link ratios (:1189-1250). The reference keys a std::map on (cluster1, cluster2), clusters -2 and -1 included; a
    // dense (label + 2) x (label + 2) count matrix walked in ascending key order yields the same sequence.
    const int M = K + 2;
    S.link_cnt.assign((size_t)M * M, 0); S.links_in.assign((size_t)M, 0);
    std::vector<int>& link_cnt = S.link_cnt; std::vector<int>& links_in = S.links_in;
    auto count_link = [&](int r1, int r2) {
        const int c1 = clustered[r1] + 2, c2 = clustered[r2] + 2;
        if (c1 != c2) link_cnt[(size_t)c1 * M + c2] += 1;
        links_in[(size_t)c1] += 1;
    };
    // only reads of the window have neighbours (the counts do not depend on the visiting order)
    if (low_memory) {
        for (int r1 : S.masked) for (int o = g.off[r1]; o < g.off[r1 + 1]; ++o) count_link(r1, g.adj[o]);
    } else {
        for (int k : S.masked) for (int o = g.off[k]; o < g.off[k + 1]; ++o) count_link(g.adj[o], k);
    }
    std::vector<std::pair<std::pair<int, int>, double>>& sorted_links = S.sorted_links;
    sorted_links.clear();
    for (int c1 = 0; c1 < M; ++c1)
        for (int c2 = 0; c2 < M; ++c2)
            if (link_cnt[(size_t)c1 * M + c2] > 0)
                sorted_links.push_back(std::make_pair(std::make_pair(c1 - 2, c2 - 2), (double)link_cnt[(size_t)c1 * M + c2] / links_in[(size_t)c1]));
    std::sort(sorted_links.begin(), sorted_links.end(),
              [](const std::pair<std::pair<int, int>, double>& a, const std::pair<std::pair<int, int>, double>& b) { return a.second > b.second; });
    // old label -> new label, keys -2 .. K-1 stored at [label + 2]
    S.o2n.assign((size_t)M, 0);
    std::vector<int>& o2n = S.o2n;
    for (int gl : glist) o2n[(size_t)gl + 2] = gl;
    o2n[1] = -1; o2n[0] = -2;
    for (auto& pc : sorted_links) {
        if (!(pc.second > 0.01)) continue;
        const int c1 = pc.first.first, c2 = pc.first.second;
        if (o2n[(size_t)c1 + 2] == o2n[(size_t)c2 + 2]) continue;
        bool bad = false;
        for (int g1 : glist) {
            if (o2n[(size_t)g1 + 2] != o2n[(size_t)c1 + 2]) continue;
            for (int g2 : glist) if (o2n[(size_t)g2 + 2] == o2n[(size_t)c2 + 2] && incompat[(size_t)S.index_of[(size_t)g1] * G + S.index_of[(size_t)g2]] > 1) bad = true;
        }
        if (!bad) { const int to = o2n[(size_t)c1 + 2], from = o2n[(size_t)c2 + 2]; for (int g2 : glist) if (o2n[(size_t)g2 + 2] == from) o2n[(size_t)g2 + 2] = to; }
    }
    S.new_index.assign((size_t)M, -1);   // keyed by the (possibly negative) merged label + 2
    int ni = 0;
    for (int gl : glist) { const int v = o2n[(size_t)gl + 2]; if (S.new_index[(size_t)v + 2] < 0) S.new_index[(size_t)v + 2] = ni++; }
    for (int gl : glist) o2n[(size_t)gl + 2] = S.new_index[(size_t)o2n[(size_t)gl + 2] + 2];
    std::vector<int32_t> out((size_t)N, -1);
    for (int r = 0; r < N; ++r) out[r] = o2n[(size_t)clustered[r] + 2];
    return out;
}

#ifdef HS_SELFCHECK   // the straightforward forms, kept as the cross-check in the test-harness build
// merge_close_clusters: cluster_graph.cpp:402-501
static void merge_close_clusters_ref(const SrGraph& g, bool low_memory, std::vector<int32_t>& clusters, const uint8_t* mask,
                                 const std::vector<int32_t>& order) {
    const int N = (int)clusters.size();
    std::set<int> tested;
    std::vector<int> initial((size_t)N, 0);
    for (int v : clusters) if (v >= 0 && v < N) initial[v] += 1;
    std::vector<int> votes((size_t)N, 0), touched;
    std::vector<int32_t> nc;
    std::vector<int> count;
    std::vector<int32_t> order_masked;   // the shuffled order restricted to the window's reads (the others are skipped anyway)
    for (int i : order) if (mask[i]) order_masked.push_back(i);
    for (int node = 0; node < N; ++node) {
        if (!(clusters[node] >= 0 && tested.find(clusters[node]) == tested.end())) continue;
        const int target = clusters[node];
        nc = clusters; count = initial;
        int changes = 3, iters = 0;
        while (changes > 0 && iters < 10) {
            changes = 0;
            for (int i : order_masked) {
                if (nc[i] != target) continue;
                touched.clear();
                const int o0 = g.off[i], o1 = g.off[i + 1];
                if (low_memory) {   // :441-445 iterates j < degree and asks whether j itself is a neighbour (sic)
                    for (int j = 0; j < o1 - o0; ++j)
                        if (std::binary_search(g.adj.begin() + o0, g.adj.begin() + o1, j) && nc[j] >= 0) { if (votes[nc[j]]++ == 0) touched.push_back(nc[j]); }
                } else {
                    for (int o = o0; o < o1; ++o) { const int l = nc[g.adj[o]]; if (l >= 0) { if (votes[l]++ == 0) touched.push_back(l); } }
                }
                // largest and runner-up in ascending label order with strict '>' (:455-470)
                std::sort(touched.begin(), touched.end());
                int max_index = 0, max_value = 0, second_index = 0, second_value = 0;
                for (int l : touched) {
                    const int v = votes[l];
                    if (v > max_value) { second_value = max_value; second_index = max_index; max_value = v; max_index = l; }
                    else if (v > second_value) { second_value = v; second_index = l; }
                }
                for (int l : touched) votes[l] = 0;
                if (max_value > 0 && max_index != target) { count[nc[i]]--; count[max_index]++; changes++; nc[i] = max_index; }
                else if (max_value > 0 && max_value <= 2 * second_value) { count[nc[i]]--; count[second_index]++; nc[i] = second_index; changes++; }
            }
            iters++;
        }
        tested.insert(target);
        if (count[target] == 0) { clusters = nc; initial = count; }
    }
}

// merge_wrongly_split_haplotypes: separate_reads.cpp:1007-1327
static std::vector<int32_t> merge_wrongly_split_ref(const SrContigState& st, const std::vector<int32_t>& clustered, const SrGraph& g,
                                                bool low_memory, int posstart, int posend) {
    const hs_sr_contig& c = *st.c;
    const int N = st.N;
    std::set<int> groups;
    std::map<int, int> index_of;
    int index = 0;
    for (int r = 0; r < N; ++r)
        if (clustered[r] > -1) { groups.insert(clustered[r]); if (index_of.find(clustered[r]) == index_of.end()) index_of[clustered[r]] = index++; }
    const int G = (int)groups.size();
    if (G <= 1) {
        std::vector<int32_t> one((size_t)N, 0);
        for (int r = 0; r < N; ++r) if (clustered[r] == -2) one[r] = -2;
        return one;
    }
    std::vector<int> incompat((size_t)G * G, 0), pos_last((size_t)G * G, -10);
    std::vector<int> glist(groups.begin(), groups.end());
    std::vector<int> gidx(glist.size());
    for (size_t i = 0; i < glist.size(); ++i) gidx[i] = index_of[glist[i]];
    // per-SNP majority base of every cluster (:1056-1112); the inner map's iteration order decides ties (>=)
    std::vector<int> slot_of_group;   // group label -> dense slot
    {
        int mx = 0; for (int gl : glist) mx = std::max(mx, gl);
        slot_of_group.assign((size_t)mx + 1, -1);
        for (size_t i = 0; i < glist.size(); ++i) slot_of_group[glist[i]] = (int)i;
    }
    std::vector<std::vector<uint8_t>> seen((size_t)G);
    std::vector<std::vector<int>> cnts((size_t)G, std::vector<int>(256, 0));
    std::vector<int> nb_bases((size_t)G);
    std::vector<int> majority((size_t)G);   // 0 == the operator[] default for clusters absent at this SNP
    for (int s = 0; s < c.n_snps; ++s) {
        const int p = c.snp_pos[s];
        if (!(p >= posstart && p < posend)) continue;
        for (int i = 0; i < G; ++i) { nb_bases[i] = 0; majority[i] = 0; }
        for (int64_t e = c.col_off[s]; e < c.col_off[s + 1]; ++e) {
            const int cl = clustered[c.col_idx[e]];
            if (cl > -1) {
                const int sl = slot_of_group[cl];
                const uint8_t b = c.col_code[e];
                if (cnts[sl][b]++ == 0) seen[sl].push_back(b);
                nb_bases[sl]++;
            }
        }
        // The reference walks the cluster's base counts in robin_hood order keeping (max, second max) with `>=` on the max
        // (:1090-1099). The pair of values does not depend on the order, and neither does the verdict: a unique maximum names
        // the base, a tied maximum gives second == max and is rejected by `second_max * 2 > max` just below.
        int first_max = -1; bool several = false;
        for (int i = 0; i < G; ++i) {
            if (seen[i].empty()) continue;
            int second_max = 0, mx = 0;
            int max_base = ' ';
            for (uint8_t b : seen[i]) {
                const int v = cnts[i][b];
                if (v >= mx) { max_base = (int)(signed char)b; second_max = mx; mx = v; }
                else if (v > second_max) second_max = v;
                cnts[i][b] = 0;
            }
            seen[i].clear();
            if (second_max * 2 > mx || nb_bases[i] * 0.5 > mx) max_base = ' ';
            majority[i] = (int)(uint8_t)max_base;
            if (max_base != ' ') { const int mb = (int)(uint8_t)max_base; if (first_max < 0) first_max = mb; else if (mb != first_max) several = true; }
        }
        if (!several) continue;
        for (int a = 0; a < G; ++a)
            for (int b = 0; b < G; ++b) {
                if (majority[a] != ' ' && majority[b] != ' ' && glist[a] > glist[b]) {
                    const int i1 = gidx[a], i2 = gidx[b];
                    if (majority[a] != majority[b] && p - pos_last[(size_t)i1 * G + i2] > 10) {
                        incompat[(size_t)i1 * G + i2] += 1; incompat[(size_t)i2 * G + i1] += 1;
                        pos_last[(size_t)i1 * G + i2] = p; pos_last[(size_t)i2 * G + i1] = p;
                    }
                }
            }
    }
    // link ratios (:1189-1250). The reference keys a std::map on (cluster1, cluster2), clusters -2 and -1 included; a
    // dense (label + 2) x (label + 2) count matrix walked in ascending key order yields the same sequence.
    int max_label = -2;
    for (int r = 0; r < N; ++r) max_label = std::max(max_label, clustered[r]);
    const int M = max_label + 3;
    std::vector<int> link_cnt((size_t)M * M, 0), links_in((size_t)M, 0);
    auto count_link = [&](int r1, int r2) {
        const int c1 = clustered[r1] + 2, c2 = clustered[r2] + 2;
        if (c1 != c2) link_cnt[(size_t)c1 * M + c2] += 1;
        links_in[(size_t)c1] += 1;
    };
    if (low_memory) {
        for (int r1 = 0; r1 < N; ++r1) for (int o = g.off[r1]; o < g.off[r1 + 1]; ++o) count_link(r1, g.adj[o]);
    } else {
        for (int k = 0; k < N; ++k) for (int o = g.off[k]; o < g.off[k + 1]; ++o) count_link(g.adj[o], k);
    }
    std::vector<std::pair<std::pair<int, int>, double>> sorted_links;
    for (int c1 = 0; c1 < M; ++c1)
        for (int c2 = 0; c2 < M; ++c2)
            if (link_cnt[(size_t)c1 * M + c2] > 0)
                sorted_links.push_back(std::make_pair(std::make_pair(c1 - 2, c2 - 2), (double)link_cnt[(size_t)c1 * M + c2] / links_in[(size_t)c1]));
    std::sort(sorted_links.begin(), sorted_links.end(),
              [](const std::pair<std::pair<int, int>, double>& a, const std::pair<std::pair<int, int>, double>& b) { return a.second > b.second; });
    std::map<int, int> o2n;
    for (int gl : glist) o2n[gl] = gl;
    o2n[-1] = -1; o2n[-2] = -2;
    for (auto& pc : sorted_links) {
        if (!(pc.second > 0.01)) continue;
        const int c1 = pc.first.first, c2 = pc.first.second;
        if (o2n[c1] == o2n[c2]) continue;
        bool bad = false;
        for (int g1 : glist) {
            if (o2n[g1] != o2n[c1]) continue;
            for (int g2 : glist) if (o2n[g2] == o2n[c2] && incompat[(size_t)index_of[g1] * G + index_of[g2]] > 1) bad = true;
        }
        if (!bad) for (int g2 : glist) if (o2n[g2] == o2n[c2]) o2n[g2] = o2n[c1];
    }
    std::map<int, int> new_index;
    int ni = 0;
    for (int gl : glist) if (new_index.find(o2n[gl]) == new_index.end()) new_index[o2n[gl]] = ni++;
    for (int gl : glist) o2n[gl] = new_index[o2n[gl]];
    std::vector<int32_t> out((size_t)N, -1);
    for (int r = 0; r < N; ++r) out[r] = o2n[clustered[r]];
    return out;
}

#endif

// finalize_clustering tail: separate_reads.cpp:973-993
void sr_finish_window(const SrContigState& st, SrWindowPlan& w, const int32_t* reclustered, bool low_memory) {
    const int N = st.N;
    std::vector<int32_t> hap(reclustered, reclustered + N);
    // first-seen renumbering of the non-negative labels (:973-984); -1 and -2 keep their meaning
    int max_label = -1;
    for (int h : hap) max_label = std::max(max_label, h);
    std::vector<int> to_index((size_t)max_label + 1, -1);
    int index_h = 0;
    for (int r = 0; r < N; ++r) {
        const int h = hap[r];
        if (h == -1) hap[r] = st.c->n_snps == 0 ? 0 : -1;
        else if (h >= 0) { if (to_index[(size_t)h] < 0) to_index[(size_t)h] = index_h++; hap[r] = to_index[(size_t)h]; }
    }
    const SrGraph& g = st.graphs[(size_t)w.graph_final];
    static const bool tim = std::getenv("HS_TIMING_FIN") != nullptr;
    auto nowus = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = tim ? nowus() : 0;
#ifdef HS_SELFCHECK
    std::vector<int32_t> hap_ref = hap;
    merge_close_clusters_ref(g, low_memory, hap_ref, w.mask.data(), st.perm);
    const std::vector<int32_t> out_ref = merge_wrongly_split_ref(st, hap_ref, g, low_memory, w.final_lo, w.final_hi);
#endif
    {
        MergeScratch& S = merge_scratch();
        S.masked.clear();
        for (int r = 0; r < N; ++r) if (w.mask[(size_t)r]) S.masked.push_back(r);
    }
    merge_close_clusters(g, low_memory, hap, w.mask.data(), st.perm, index_h);
    const double t1 = tim ? nowus() : 0;
    w.labels = merge_wrongly_split(st, hap, g, low_memory, w.final_lo, w.final_hi, index_h);
#ifdef HS_SELFCHECK
    if (hap_ref != hap || out_ref != w.labels) { std::fprintf(stderr, "HS_SELFCHECK: cluster merging differs from its reference form\n"); std::abort(); }
#endif
    if (tim) {
        static std::atomic<long> a_mcc{0}, a_mws{0}, a_n{0};
        a_mcc += (long)((t1 - t0) * 1000); a_mws += (long)((nowus() - t1) * 1000);
        if (++a_n % 10 == 0) std::fprintf(stderr, "[hs timing] finish: %ld windows, merge_close_clusters %.1f us/window, merge_wrongly_split %.1f us/window\n",
                                          a_n.load(), a_mcc.load() / 1000.0 / a_n.load(), a_mws.load() / 1000.0 / a_n.load());
    }
}

// merge_haplotypes_to_fit_within_limit up to the re-clustering: separate_reads.cpp:1341-1383.
// Returns false when the limit is already met (labels untouched).
bool sr_ploidy_init_labels(const SrContigState& st, const SrWindowPlan& w, int max_haplotypes, int32_t* out) {
    std::map<int, int> count;
    for (int v : w.labels) if (v >= 0) count[v] += 1;
    if ((int)count.size() <= max_haplotypes) return false;
    std::vector<std::pair<int, int>> v;
    for (auto& c : count) v.push_back(std::make_pair(c.second, c.first));
    std::sort(v.begin(), v.end(), std::greater<std::pair<int, int>>());
    std::set<int> kept;
    for (int i = 0; i < max_haplotypes; ++i) kept.insert(v[i].second);
    for (int r = 0; r < st.N; ++r) out[r] = (w.labels[r] >= 0 && kept.find(w.labels[r]) == kept.end()) ? -1 : w.labels[r];
    return true;
}

// choosing the window size: separate_reads.cpp:1466-1498
int32_t sr_window_size(const hs_sr_contig* cs, int n, bool amplicon) {
    int reads = 0, above4000 = 0;
    uint32_t sum = 0;   // `int sumLength` in the reference; wraps the same way
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < cs[i].n_reads; ++r) {
            const int len = cs[i].read_end[r] - cs[i].read_start[r] + 1;
            reads++; sum += (uint32_t)len;
            if (len > 4000) above4000++;
        }
    const double mean = (int32_t)sum / double(reads);
    int w = 2000;
    if (above4000 < 20 && mean < 4000 && mean > 2000) w = 1000;
    else if (above4000 < 20 && mean < 2000) w = 500;
    if (amplicon) { w = 0; for (int i = 0; i < n; ++i) w = std::max(w, (int)cs[i].length); }
    return w;
}

// coverage test of separate_reads.cpp:1476-1481,1516: float accumulation in read order
bool sr_coverage_above_1000(const hs_sr_contig& c) {
    float cov = 0;
    for (int r = 0; r < c.n_reads; ++r) cov += c.read_end[r] - c.read_start[r] + 1;
    cov /= c.length;
    return cov > 1000;
}

}  // namespace hs
