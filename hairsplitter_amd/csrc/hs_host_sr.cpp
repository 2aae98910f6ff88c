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
#include <functional>
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

// One row of create_read_graph_low_memory (separate_reads.cpp:590-670) from the counts of the window's own reads: srow / drow hold
// nsim / ndif of masked read `i` against the m masked reads (local order); every other read of the contig keeps distance 0. No
// `sim > 0` guard on this path (:618): 0 / 0 is NaN and std::sort sees it, as in the reference.
void sr_pick_row_sorted_low_memory(const int32_t* srow, const int32_t* drow, const int32_t* ids, int m, int i, int N, const uint8_t* mask, float error_rate,
                                   std::vector<int>& picked) {
    std::vector<std::pair<int, float>> smallest((size_t)N);
    for (int r = 0; r < N; ++r) smallest[(size_t)r] = std::make_pair(r, 0.0f);
    int max_compat = 0;
    for (int j = 0; j < m; ++j) {
        if (j == i) continue;
        const int nsim = srow[j], ndif = drow[j];
        smallest[(size_t)ids[j]].second = 1 - std::max(0, ndif - 1) / float(ndif + nsim);
        if (nsim > max_compat) max_compat = nsim;
    }
    for (int j = 0; j < m; ++j)
        if (j != i && srow[j] + drow[j] < 0.7 * max_compat) smallest[(size_t)ids[j]].second = 0;
    pick_neighbors_sorted(smallest, mask, error_rate, picked);
}

// position of read r in the ascending list ids, or -1
static inline int local_index(const std::vector<int32_t>& ids, int r) {
    const auto it = std::lower_bound(ids.begin(), ids.end(), r);
    return (it != ids.end() && *it == r) ? (int)(it - ids.begin()) : -1;
}

// create_read_graph_low_memory: separate_reads.cpp:538-693. Only reads of the mask take part (ext = mask && present at a SNP,
// :582-586) and only they are linked, so the graph is built over the window's local indices; the neighbour selection runs
// on the reference's N-entry array (entries of the other reads are zero), because std::sort sees all of them.
void sr_build_window_graph_low_memory(const SrContigState& st, const SrWindowPlan& w, float error_rate, std::vector<std::vector<int32_t>>& lists) {
    const hs_sr_contig& c = *st.c;
    const int N = st.N;
    const int m = (int)w.ids.size();
    std::vector<int> first((size_t)N, -1);          // per-read SNP vectors of the contig (:545-575)
    std::vector<std::vector<uint8_t>> val((size_t)N);
    for (int s = 0; s < c.n_snps; ++s)
        for (int64_t e = c.col_off[s]; e < c.col_off[s + 1]; ++e) {
            const int r = c.col_idx[e];
            if (first[(size_t)r] == -1) first[(size_t)r] = s;
            val[(size_t)r].push_back(c.col_code[e] == c.snp_ref[s] ? 1 : (c.col_code[e] == c.snp_alt[s] ? 2 : 0));
        }
    std::vector<uint8_t> mask((size_t)N, 0);
    for (int r : w.ids) mask[(size_t)r] = 1;
    lists.assign((size_t)m, std::vector<int32_t>());
    std::vector<std::pair<int, float>> smallest((size_t)N);
    std::vector<int> simv((size_t)N), difv((size_t)N), picked;
    for (int j1 = 0; j1 < m; ++j1) {
        const int r1 = w.ids[(size_t)j1];
        if (first[(size_t)r1] == -1) continue;
        int max_compat = 0;
        for (int r = 0; r < N; ++r) { smallest[(size_t)r] = std::make_pair(r, 0.0f); simv[(size_t)r] = 0; difv[(size_t)r] = 0; }
        for (int j2 = 0; j2 < m; ++j2) {
            const int r2 = w.ids[(size_t)j2];
            if (first[(size_t)r2] == -1 || r1 == r2) continue;
            int nsim = 0, ndif = 0;
            const long a = std::max(first[(size_t)r1], first[(size_t)r2]);
            const long b = std::min((long)val[(size_t)r1].size() + first[(size_t)r1] - 1, (long)val[(size_t)r2].size() + first[(size_t)r2] - 1);
            for (long p = a; p <= b; ++p) {
                const int v1 = val[(size_t)r1][(size_t)(p - first[(size_t)r1])], v2 = val[(size_t)r2][(size_t)(p - first[(size_t)r2])];
                if (v1 == 2 && v2 == 2) nsim += 3; else if (v1 == 1 && v2 == 1) nsim++; else if (v1 != 0 && v2 != 0) ndif++;
            }
            smallest[(size_t)r2].second = 1 - std::max(0, ndif - 1) / float(ndif + nsim);
            if (nsim > max_compat) max_compat = nsim;
            simv[(size_t)r2] = nsim; difv[(size_t)r2] = ndif;
        }
        for (int r : w.ids)
            if (r != r1 && simv[(size_t)r] + difv[(size_t)r] < 0.7 * max_compat) smallest[(size_t)r].second = 0;
        pick_neighbors_sorted(smallest, mask.data(), error_rate, picked);   // 0/0 distances (NaN) exist on this path
        for (int nb : picked) { const int j2 = local_index(w.ids, nb); lists[(size_t)j1].push_back(j2); lists[(size_t)j2].push_back(j1); }
    }
    for (auto& v : lists) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
}

// Window / mask planning: separate_reads.cpp:1545-1622 (the running SNP cursor is carried across windows). The mask of a
// window is never materialised over the N reads of the contig: it is the reads of the window's first SNP column that are
// also in its last one -- or lie beyond that column's last read, which the reference's clearing loop (:1612-1619) never
// reaches --, i.e. a merge of two ascending lists.
// The plans of a contig are a few dozen small vectors each; a state that lives from call to call (SrWorkspace) hands their
// storage round instead of freeing and allocating it every time.
SrWindowPlan SrContigState::take_window() {
    if (spare.empty()) return SrWindowPlan();
    SrWindowPlan w = std::move(spare.back());
    spare.pop_back();
    w.ids.clear(); w.labels.clear(); w.local_snps.clear();
    w.start = w.end = 0; w.has_snps = false; w.final_lo = w.final_hi = 0; w.row0 = -1; w.final_graph_empty = false; w.col_a = w.col_b = -1;
    return w;
}

void sr_plan_windows(SrContigState& st, int window_size, float error_rate, bool low_memory, bool with_reads) {
    (void)error_rate;
    const hs_sr_contig& c = *st.c;
    const int N = st.N;
    const long L = c.length;
    for (SrWindowPlan& old : st.windows) st.spare.push_back(std::move(old));
    st.windows.clear();
    st.snp_pos_sorted = std::is_sorted(c.snp_pos, c.snp_pos + c.n_snps);
    if (c.n_snps == 0) return;
    int cur = 0, chunk = -1, upper;
    while ((long)(chunk + 1) * window_size + 100 <= L) {
        chunk++;
        upper = (chunk + 1) * window_size;
        const bool last = (long)(chunk + 1) * window_size + 100 > L;
        if (last) upper = (int)L + 1;
        SrWindowPlan w = st.take_window();
        w.start = chunk * window_size;
        w.end = std::min(upper - 1, (int)L);
        if (cur >= c.n_snps || c.snp_pos[cur] > upper - 1) {          // :1565-1587
            w.has_snps = false;
            int mid = (w.start + std::min(upper - 1, (int)L)) / 2;
            if (mid < 500) mid = std::min(500, (int)(L / 2));
            if (mid > (int)L - 500) mid = std::max((int)(L / 2), (int)L - 500);
            for (int r = 0; r < N; ++r) if (c.read_start[r] <= mid && c.read_end[r] >= mid) w.ids.push_back(r);
            w.labels.assign(w.ids.size(), 0);
            st.windows.push_back(std::move(w));
            continue;
        }
        w.has_snps = true;
        if (chunk == 0) {
            while (cur < c.n_snps - 1 && c.snp_pos[cur] < chunk * window_size + 0.2 * window_size
                   && c.snp_pos[cur + 1] < chunk * window_size + 0.4 * window_size) cur++;
        }
        const int col_a = cur;
        while (cur < c.n_snps && c.snp_pos[cur] < upper - 1) cur++;
        if (cur > 0) cur--;
        if (last) {
            while (cur > 0 && c.snp_pos[cur] > upper - 1 - 0.2 * window_size && c.snp_pos[cur - 1] > upper - 1 - 0.4 * window_size) cur--;
        }
        w.col_a = col_a; w.col_b = cur;
        if (with_reads) {
            const int32_t* a = c.col_idx + c.col_off[col_a]; const int32_t* a1 = c.col_idx + c.col_off[col_a + 1];
            const int32_t* b = c.col_idx + c.col_off[cur]; const int32_t* b1 = c.col_idx + c.col_off[cur + 1];
            auto strictly_ascending = [](const int32_t* x, const int32_t* x1) { return std::adjacent_find(x, x1, std::greater_equal<int32_t>()) == x1; };
            if (strictly_ascending(a, a1) && strictly_ascending(b, b1)) {
                const int32_t b_last = b1 > b ? b1[-1] : -1;          // an empty last column clears nothing
                for (; a < a1; ++a) {
                    if (*a > b_last) { w.ids.push_back(*a); continue; }
                    while (b < b1 && *b < *a) ++b;
                    if (b < b1 && *b == *a) w.ids.push_back(*a);
                }
            } else {   // a .col whose read indices do not ascend: the reference's two loops as they are, on a dense mask
                std::vector<uint8_t> mask((size_t)N, 0);
                for (; a < a1; ++a) mask[(size_t)*a] = 1;
                int idxmask = 0;
                for (; b < b1; ++b) { while (idxmask < *b) { mask[(size_t)idxmask] = 0; idxmask++; } idxmask++; }
                for (int r = 0; r < N; ++r) if (mask[(size_t)r]) w.ids.push_back(r);
            }
        }
        cur++;
        // finalize_clustering is handed the *global* low_memory flag (:1708): with low_memory_now && !low_memory it
        // sees an Eigen matrix that was never filled
        w.final_graph_empty = st.low_memory_now && !low_memory;
        // SNPs seeding a local run (:1673-1676): inside [start, start+window) and more than 10 bp apart
        int lastpos = -10;
        int s_lo = 0, s_hi = c.n_snps;
        if (st.snp_pos_sorted) {
            s_lo = (int)(std::lower_bound(c.snp_pos, c.snp_pos + c.n_snps, chunk * window_size) - c.snp_pos);
            s_hi = (int)(std::lower_bound(c.snp_pos, c.snp_pos + c.n_snps, chunk * window_size + window_size) - c.snp_pos);
        }
        for (int s = s_lo; s < s_hi; ++s) {
            const int p = c.snp_pos[s];
            if (p >= chunk * window_size && p < chunk * window_size + window_size && p > lastpos + 10) { lastpos = p; w.local_snps.push_back(s); }
        }
        w.final_lo = chunk * window_size; w.final_hi = chunk * window_size + window_size;
        st.windows.push_back(std::move(w));
    }
}

// Per-thread scratch of the two cluster-merging steps: they run once per clustering window the device did not finish, so
// nothing in them allocates, and every table is indexed by cluster label (a handful) or by local node.
struct MergeScratch {
    std::vector<int32_t> nc, order;
    std::vector<int> initial, count, votes, touched;
    std::vector<char> tested;
    // merge_wrongly_split
    std::vector<int> glist, gidx, index_of, slot_of, incompat, pos_last, nb_bases, majority, link_cnt, links_in, o2n, new_index;
    std::vector<std::vector<int>> cnts;          // [slot][256], all zero between uses
    std::vector<std::vector<uint8_t>> seen;
    std::vector<std::pair<std::pair<int, int>, double>> sorted_links;
};
static MergeScratch& merge_scratch() { static thread_local MergeScratch s; return s; }

// merge_close_clusters: cluster_graph.cpp:402-501, on the window's local nodes (`clusters`: m labels, first-seen numbering
// 0 .. n_labels-1 or -1; `ids`: the read of every node; S.order: the nodes in the order of the shuffled permutation).
static void merge_close_clusters(const SrLocalGraph& g, bool low_memory, std::vector<int32_t>& clusters, const std::vector<int32_t>& ids, int n_labels) {
    MergeScratch& S = merge_scratch();
    const int m = (int)clusters.size();
    const int K = n_labels > 0 ? n_labels : 1;
    S.initial.assign((size_t)K, 0); S.votes.assign((size_t)K, 0); S.tested.assign((size_t)K, 0);
    for (int j = 0; j < m; ++j) if (clusters[(size_t)j] >= 0) S.initial[(size_t)clusters[(size_t)j]] += 1;
    S.nc.assign(clusters.begin(), clusters.end());
    std::vector<int32_t>& nc = S.nc;
    std::vector<int>& votes = S.votes;
    std::vector<int>& touched = S.touched;
    for (int node = 0; node < m; ++node) {
        if (!(clusters[(size_t)node] >= 0 && !S.tested[(size_t)clusters[(size_t)node]])) continue;
        const int target = clusters[(size_t)node];
        S.count = S.initial;
        int changes = 3, iters = 0;
        while (changes > 0 && iters < 10) {
            changes = 0;
            for (int i : S.order) {
                if (nc[(size_t)i] != target) continue;
                touched.clear();
                const int64_t o0 = g.begin(i), o1 = g.end(i);
                if (low_memory) {
                    // :441-445 iterates j < degree and asks whether READ j itself is a neighbour (sic): reads outside the window
                    // carry -2 and never vote
                    for (int r = 0; r < (int)(o1 - o0); ++r) {
                        const int j = local_index(ids, r);
                        if (j >= 0 && std::binary_search(g.nbr + o0, g.nbr + o1, j) && nc[(size_t)j] >= 0) { if (votes[(size_t)nc[(size_t)j]]++ == 0) touched.push_back(nc[(size_t)j]); }
                    }
                } else {
                    for (int64_t o = o0; o < o1; ++o) { const int l = nc[(size_t)g.nbr[o]]; if (l >= 0) { if (votes[(size_t)l]++ == 0) touched.push_back(l); } }
                }
                // largest and runner-up in ascending label order with strict '>' (:455-470)
                std::sort(touched.begin(), touched.end());
                int max_index = 0, max_value = 0, second_index = 0, second_value = 0;
                for (int l : touched) {
                    const int v = votes[(size_t)l];
                    if (v > max_value) { second_value = max_value; second_index = max_index; max_value = v; max_index = l; }
                    else if (v > second_value) { second_value = v; second_index = l; }
                }
                for (int l : touched) votes[(size_t)l] = 0;
                if (max_value > 0 && max_index != target) { S.count[(size_t)nc[(size_t)i]]--; S.count[(size_t)max_index]++; changes++; nc[(size_t)i] = max_index; }
                else if (max_value > 0 && max_value <= 2 * second_value) { S.count[(size_t)nc[(size_t)i]]--; S.count[(size_t)second_index]++; nc[(size_t)i] = second_index; changes++; }
            }
            iters++;
        }
        S.tested[(size_t)target] = 1;
        if (S.count[(size_t)target] == 0) { clusters.assign(nc.begin(), nc.end()); S.initial = S.count; }   // the cluster dissolved: keep
        else nc.assign(clusters.begin(), clusters.end());                                                  // undo
    }
}

// merge_wrongly_split_haplotypes: separate_reads.cpp:1007-1327, on the window's local nodes. `n_labels` as above.
static std::vector<int32_t> merge_wrongly_split(const SrContigState& st, const std::vector<int32_t>& clustered, const std::vector<int32_t>& ids,
                                                const SrLocalGraph& g, bool low_memory, int posstart, int posend, int n_labels) {
    MergeScratch& S = merge_scratch();
    const hs_sr_contig& c = *st.c;
    const int m = (int)clustered.size();
    const int K = n_labels > 0 ? n_labels : 1;
    // clusters present, ascending (glist), and their first-seen rank over the reads (index_of), as the std::set / std::map give
    S.index_of.assign((size_t)K, -1); S.slot_of.assign((size_t)K, -1);
    int index = 0;
    for (int j = 0; j < m; ++j) { const int cl = clustered[(size_t)j]; if (cl > -1 && S.index_of[(size_t)cl] < 0) S.index_of[(size_t)cl] = index++; }
    S.glist.clear();
    for (int l = 0; l < K; ++l) if (S.index_of[(size_t)l] >= 0) { S.slot_of[(size_t)l] = (int)S.glist.size(); S.glist.push_back(l); }
    const int G = (int)S.glist.size();
    if (G <= 1) return std::vector<int32_t>((size_t)m, 0);      // (every node of the window is a masked read)
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
        for (int i = 0; i < G; ++i) { nb_bases[(size_t)i] = 0; majority[(size_t)i] = 0; }
        for (int64_t e = c.col_off[s]; e < c.col_off[s + 1]; ++e) {
            const int j = local_index(ids, c.col_idx[e]);      // reads outside the window carry -2
            const int cl = j >= 0 ? clustered[(size_t)j] : -2;
            if (cl > -1) {
                const int sl = S.slot_of[(size_t)cl];
                const uint8_t b = c.col_code[e];
                if (S.cnts[(size_t)sl][b]++ == 0) S.seen[(size_t)sl].push_back(b);
                nb_bases[(size_t)sl]++;
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
            if (second_max * 2 > mx || nb_bases[(size_t)i] * 0.5 > mx) max_base = ' ';
            majority[(size_t)i] = (int)(uint8_t)max_base;
            if (max_base != ' ') { const int mb = (int)(uint8_t)max_base; if (first_max < 0) first_max = mb; else if (mb != first_max) several = true; }
        }
        if (!several) continue;
        for (int a = 0; a < G; ++a)
            for (int b = 0; b < G; ++b) {
                if (majority[(size_t)a] != ' ' && majority[(size_t)b] != ' ' && glist[(size_t)a] > glist[(size_t)b]) {
                    const int i1 = S.gidx[(size_t)a], i2 = S.gidx[(size_t)b];
                    if (majority[(size_t)a] != majority[(size_t)b] && p - pos_last[(size_t)i1 * G + i2] > 10) {
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
    auto count_link = [&](int j1, int j2) {
        const int c1 = clustered[(size_t)j1] + 2, c2 = clustered[(size_t)j2] + 2;
        if (c1 != c2) link_cnt[(size_t)c1 * M + c2] += 1;
        links_in[(size_t)c1] += 1;
    };
    // only reads of the window have neighbours (the counts do not depend on the visiting order)
    if (low_memory) {
        for (int j1 = 0; j1 < m; ++j1) for (int64_t o = g.begin(j1); o < g.end(j1); ++o) count_link(j1, g.nbr[o]);
    } else {
        for (int k = 0; k < m; ++k) for (int64_t o = g.begin(k); o < g.end(k); ++o) count_link(g.nbr[o], k);
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
    std::vector<int32_t> out((size_t)m, -1);
    for (int j = 0; j < m; ++j) out[(size_t)j] = o2n[(size_t)clustered[(size_t)j] + 2];
    return out;
}

// finalize_clustering tail: separate_reads.cpp:973-993
void sr_finish_window(const SrContigState& st, SrWindowPlan& w, const int32_t* reclustered, const SrLocalGraph& g, bool low_memory) {
    const int m = (int)w.ids.size();
    std::vector<int32_t> hap(reclustered, reclustered + m);
    // first-seen renumbering of the non-negative labels (:973-984); -1 keeps its meaning (a window of the chain has SNPs)
    int max_label = -1;
    for (int h : hap) max_label = std::max(max_label, h);
    std::vector<int> to_index((size_t)max_label + 1, -1);
    int index_h = 0;
    for (int j = 0; j < m; ++j) {
        const int h = hap[(size_t)j];
        if (h == -1) hap[(size_t)j] = st.c->n_snps == 0 ? 0 : -1;
        else if (h >= 0) { if (to_index[(size_t)h] < 0) to_index[(size_t)h] = index_h++; hap[(size_t)j] = to_index[(size_t)h]; }
    }
    {   // the shuffled order restricted to the window's reads (the others are skipped anyway)
        MergeScratch& S = merge_scratch();
        S.order.clear();
        static thread_local std::vector<std::pair<int32_t, int32_t>> byrank;
        byrank.clear();
        for (int j = 0; j < m; ++j) byrank.push_back(std::make_pair(st.rank[(size_t)w.ids[(size_t)j]], j));
        std::sort(byrank.begin(), byrank.end());
        for (auto& pr : byrank) S.order.push_back(pr.second);
    }
    SrLocalGraph gf = g;
    std::vector<int64_t> empty_off;
    if (w.final_graph_empty) { empty_off.assign((size_t)m + 1, 0); gf.off = empty_off.data(); gf.nbr = nullptr; }
    merge_close_clusters(gf, low_memory, hap, w.ids, index_h);
    w.labels = merge_wrongly_split(st, hap, w.ids, gf, low_memory, w.final_lo, w.final_hi, index_h);
}

// merge_haplotypes_to_fit_within_limit up to the re-clustering: separate_reads.cpp:1341-1383.
// Returns false when the limit is already met (labels untouched).
bool sr_ploidy_init_labels(const SrWindowPlan& w, int max_haplotypes, int32_t* out) {
    std::map<int, int> count;
    for (int v : w.labels) if (v >= 0) count[v] += 1;
    if ((int)count.size() <= max_haplotypes) return false;
    std::vector<std::pair<int, int>> v;
    for (auto& c : count) v.push_back(std::make_pair(c.second, c.first));
    std::sort(v.begin(), v.end(), std::greater<std::pair<int, int>>());
    std::set<int> kept;
    for (int i = 0; i < max_haplotypes; ++i) kept.insert(v[(size_t)i].second);
    for (size_t j = 0; j < w.labels.size(); ++j) out[j] = (w.labels[j] >= 0 && kept.find(w.labels[j]) == kept.end()) ? -1 : w.labels[j];
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
