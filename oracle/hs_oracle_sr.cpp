// ORACLE (test infrastructure, CPU only): restatement of stage 4, HS_separate_reads
// (separate_reads.cpp + cluster_graph.cpp). Eigen::SparseMatrix<int> products are restated as dense
// N x N int arrays / sorted neighbour lists (col-major inner iteration == ascending row index).
// Citations are to /root/reference/src/<file>:<line>.
#include "hs_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <numeric>
#include <random>
#include <set>
#include <sstream>
#include <unordered_map>

namespace hso {

// separate_reads.cpp:46-190
std::vector<ColContig> parse_column_file(const std::string& file, int max_coverage, float rarest_strain_abundance) {
    std::vector<ColContig> out;
    std::ifstream infile(file);
    std::string line;
    bool numbers = false, firstsnpline = true;
    while (std::getline(infile, line)) {
        std::istringstream iss(line);
        std::string line_type;
        iss >> line_type;
        if (line_type == "CONTIG") {
            ColContig c;
            c.contig_line = line;
            std::string name, length;
            double cov = 0;
            iss >> name >> length >> cov;
            c.length = std::atoi(length.c_str());
            c.coverage = cov;
            out.push_back(c);
        } else if (line_type == "SNPS") {
            std::string pos, ref_s, sec_s, content, readsIdx;
            char ref_base, second_base;
            iss >> pos >> ref_s >> sec_s;
            if (firstsnpline && (!std::isalpha((unsigned char)ref_s[0]) && ref_s[0] != '-')) numbers = true;
            if (numbers) { ref_base = (char)std::stoi(ref_s); second_base = (char)std::stoi(sec_s); }
            else { ref_base = ref_s[0]; second_base = sec_s[0]; }
            firstsnpline = false;
            iss >> readsIdx >> content;
            std::string new_content, integer;
            for (char ch : content) {
                if (ch == ',') {
                    if (integer == " ") new_content += " ";
                    else if (numbers) new_content += (char)(unsigned char)std::stoi(integer);
                    else new_content += integer;
                    integer = "";
                } else integer += ch;
            }
            content = new_content;
            std::vector<int> readIdxs;
            std::string cur;
            for (char ch : readsIdx) {
                if (ch == ',') { readIdxs.push_back(std::atoi(cur.c_str())); cur = ""; }
                else cur += ch;
            }
            Column snp;
            snp.pos = std::atoi(pos.c_str());
            snp.ref_base = (unsigned char)ref_base;
            snp.second_base = (unsigned char)second_base;
            int cov_maj = 0, cov_sec = 0, cov = 0;
            for (size_t n = 0; n < content.size(); n++) {
                if (content[n] != ' ' && cov < max_coverage) {
                    snp.content.push_back((unsigned char)content[n]);
                    snp.readIdxs.push_back((unsigned)readIdxs[n]);
                    if (content[n] == ref_base) cov_maj++;
                    else if (content[n] == second_base) cov_sec++;
                }
                if (content[n] != ' ' && readIdxs[n] >= 0) cov++;
            }
            if ((float)cov_sec >= rarest_strain_abundance * (float)(cov_maj + cov_sec)) out.back().snps.push_back(snp);
        } else if (line_type == "READ") {
            out.back().read_lines.push_back(line);
            std::string name, sR, eR, sC, eC;
            iss >> name >> sR >> eR >> sC >> eC;
            try { out.back().readLimits.push_back(std::make_pair(std::stoi(sC), std::stoi(eC))); }
            catch (const std::invalid_argument&) {
                std::cout << "error in parsing read limits" << std::endl << "line : " << line << std::endl;
                std::exit(1);
            }
        }
    }
    return out;
}

// separate_reads.cpp:374-433 : similarity = 3*A*At + R*Rt, difference = A*Rt + R*At, diagonals zeroed
void list_similarities_and_differences(const std::vector<Column>& snps, int N, std::vector<int>& sim, std::vector<int>& diff) {
    sim.assign((size_t)N * N, 0);
    diff.assign((size_t)N * N, 0);
    std::vector<int> alt, ref;
    for (const Column& snp : snps) {
        alt.clear(); ref.clear();
        for (size_t r = 0; r < snp.readIdxs.size(); r++) {
            if (snp.content[r] == snp.ref_base) ref.push_back((int)snp.readIdxs[r]);
            else if (snp.content[r] == snp.second_base) alt.push_back((int)snp.readIdxs[r]);
        }
        for (int a : alt) for (int b : alt) sim[(size_t)a * N + b] += 3;
        for (int a : ref) for (int b : ref) sim[(size_t)a * N + b] += 1;
        for (int a : alt) for (int b : ref) { diff[(size_t)a * N + b] += 1; diff[(size_t)b * N + a] += 1; }
    }
    for (int i = 0; i < N; i++) { sim[(size_t)i * N + i] = 0; diff[(size_t)i * N + i] = 0; }
}

// shared tail of both graph builders (separate_reads.cpp:769-815 == :633-670)
static void pick_neighbors(const std::vector<float>& dist, const std::vector<bool>& mask, float errorRate,
                           std::vector<int>& picked) {
    std::vector<std::pair<int, float>> smallest;
    for (int r = 0; r < (int)dist.size(); r++) smallest.push_back(std::make_pair(r, dist[r]));
    std::sort(smallest.begin(), smallest.end(), [](const std::pair<int, float>& a, const std::pair<int, float>& b) { return a.second > b.second; });
    int nb_of_neighbors = 0;
    float below = 1 - errorRate * 2;
    float above = 1;
    if (smallest.size() > 1) above = smallest[0].second - (smallest[0].second - smallest[1].second) * 3;
    if (above == 1) {
        int idx = 0;
        while (idx < (int)smallest.size() && smallest[idx].second == 1) idx += 1;
        if (idx < (int)smallest.size()) {
            idx = std::min(idx + 4, (int)smallest.size() - 1);
            above = smallest[idx].second;
        }
    }
    picked.clear();
    for (auto& nb : smallest) {
        if (nb.second > below && (nb_of_neighbors < 5 || nb.second == 1 || nb.second >= above) && mask[nb.first]) {
            nb_of_neighbors++;
            picked.push_back(nb.first);
        }
    }
}

static void sort_unique(std::vector<std::vector<int>>& adj) {
    for (auto& v : adj) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
}

// separate_reads.cpp:706-828
void create_read_graph_matrix(const std::vector<bool>& mask, const std::vector<int>& sim, const std::vector<int>& diff,
                              int N, float errorRate, std::vector<std::vector<int>>& adj) {
    adj.assign(N, std::vector<int>());
    std::vector<int> picked;
    for (int read1 = 0; read1 < N; read1++) {
        if (!mask[read1]) continue;
        std::vector<float> dist(N, 0);
        int max_compat = 0;
        for (int r = 0; r < N; r++) {
            int s = (mask[r] && r != read1) ? sim[(size_t)r * N + read1] : 0;      // :737-749
            int d = (mask[r] && r != read1) ? diff[(size_t)r * N + read1] : 0;
            if (mask[r] && r != read1 && s > 0) {                                   // :752-760
                float df = (float)std::max(0, d - 1);
                dist[r] = 1 - df / float(s + d);
                if (s > max_compat) max_compat = s;
            }
        }
        for (int r = 0; r < N; r++) {                                               // :762-766
            int s = (mask[r] && r != read1) ? sim[(size_t)r * N + read1] : 0;
            int d = (mask[r] && r != read1) ? diff[(size_t)r * N + read1] : 0;
            if (mask[r] && r != read1 && s + d < 0.7 * max_compat) dist[r] = 0;
        }
        pick_neighbors(dist, mask, errorRate, picked);
        for (int nb : picked) { adj[read1].push_back(nb); adj[nb].push_back(read1); }
    }
    sort_unique(adj);   // setFromTriplets + "all values to 1" :820-827
}

// separate_reads.cpp:538-693
void create_read_graph_low_memory(const std::vector<Column>& snps, const std::vector<bool>& mask,
                                  std::vector<std::vector<int>>& nl, float errorRate) {
    const int N = (int)mask.size();
    std::vector<std::pair<int, std::vector<int>>> reads(N, std::make_pair(-1, std::vector<int>()));
    int idx_snp = 0;
    for (const Column& snp : snps) {
        for (size_t r = 0; r < snp.readIdxs.size(); r++) {
            auto& rd = reads[snp.readIdxs[r]];
            if (rd.first == -1) rd.first = idx_snp;
            if (snp.content[r] == snp.ref_base) rd.second.push_back(1);
            else if (snp.content[r] == snp.second_base) rd.second.push_back(2);
            else rd.second.push_back(0);
        }
        idx_snp++;
    }
    std::vector<bool> mask_extend(N, false);
    for (int r = 0; r < N; r++) if (mask[r] && reads[r].first != -1) mask_extend[r] = true;
    std::vector<int> picked;
    for (int read1 = 0; read1 < N; read1++) {
        if (!mask_extend[read1]) continue;
        std::vector<float> dist(N, 0);
        std::vector<int> simv(N, 0), difv(N, 0);
        int max_compat = 0;
        for (int read2 = 0; read2 < N; read2++) {
            if (!(mask_extend[read2] && read1 != read2)) continue;
            int nb_similar = 0, nb_different = 0;
            int first_common = std::max(reads[read1].first, reads[read2].first);
            // :598 mixes size_t and int; the values are small and non-negative unless a vector is empty (it is not: first != -1)
            long last_common = std::min((long)reads[read1].second.size() + reads[read1].first - 1, (long)reads[read2].second.size() + reads[read2].first - 1);
            for (long pos = first_common; pos <= last_common; pos++) {
                int v1 = reads[read1].second[pos - reads[read1].first];
                int v2 = reads[read2].second[pos - reads[read2].first];
                if (v1 == 2 && v2 == 2) nb_similar += 3;
                else if (v1 == 1 && v2 == 1) nb_similar++;
                else if (v1 != 0 && v2 != 0) nb_different++;
            }
            dist[read2] = 1 - std::max(0, nb_different - 1) / float(nb_different + nb_similar);   // :618 (0/0 -> NaN kept)
            if (nb_similar > max_compat) max_compat = nb_similar;
            simv[read2] = nb_similar; difv[read2] = nb_different;
        }
        for (int r = 0; r < N; r++)
            if (mask[r] && r != read1 && simv[r] + difv[r] < 0.7 * max_compat) dist[r] = 0;
        pick_neighbors(dist, mask, errorRate, picked);
        for (int nb : picked) { nl[read1].push_back(nb); nl[nb].push_back(read1); }
    }
    sort_unique(nl);
}

std::vector<int> shuffled_order(int n, unsigned seed) {                  // cluster_graph.cpp:173-177
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::mt19937 g(seed);
    std::shuffle(order.begin(), order.end(), g);
    return order;
}

// cluster_graph.cpp:152-230 and :240-310 (identical update rule; the graph is a neighbour list either way)
std::vector<int> chinese_whispers(const std::vector<std::vector<int>>& adj, const std::vector<int>& init,
                                  const std::vector<bool>& mask, unsigned seed, int* sweeps_out) {
    std::vector<int> clusters = init;
    const int N = (int)init.size();
    int changes = 3, iters = 0;
    std::vector<int> order = shuffled_order(N, seed);   // same seed every sweep => same permutation
    std::vector<int> neighbors(mask.size(), 0);
    while (changes > 2 && iters < 15) {
        changes = 0;
        for (int i : order) {
            if (!mask[i]) continue;
            std::fill(neighbors.begin(), neighbors.end(), 0);
            if (i < (int)adj.size())
                for (int nb : adj[i]) if (clusters[nb] >= 0) neighbors[clusters[nb]] += 1;
            int max_index = 0, max_value = 0;
            for (int j = 0; j < (int)neighbors.size(); j++)
                if (neighbors[j] > max_value) { max_value = neighbors[j]; max_index = j; }
            if (max_value > 0) {
                if (clusters[i] != max_index) changes++;
                clusters[i] = max_index;
            }
        }
        iters++;
    }
    for (size_t i = 0; i < mask.size(); i++) if (!mask[i]) clusters[i] = -2;
    if (sweeps_out) *sweeps_out = iters;
    return clusters;
}

// cluster_graph.cpp:402-501
void merge_close_clusters(const std::vector<std::vector<int>>& nl, const std::vector<std::vector<int>>& adj,
                          bool low_memory, std::vector<int>& clusters, const std::vector<bool>& mask, unsigned seed) {
    std::set<int> tested;
    std::vector<int> initialCount(clusters.size(), 0);
    for (int i : clusters) if (i >= 0 && (size_t)i < initialCount.size()) initialCount[i] += 1;
    std::vector<int> order = shuffled_order((int)clusters.size(), seed);
    for (size_t node = 0; node < clusters.size(); node++) {
        if (!(tested.find(clusters[node]) == tested.end() && clusters[node] >= 0)) continue;
        int clusterToTest = clusters[node];
        std::vector<int> newclusters = clusters;
        int changes = 3, iters = 0;
        std::vector<int> countOf = initialCount;
        while (changes > 0 && iters < 10) {
            changes = 0;
            for (int i : order) {
                if (!mask[i] || newclusters[i] != clusterToTest) continue;
                std::vector<int> neighbors(mask.size(), 0);
                if (low_memory) {
                    for (int j = 0; j < (int)nl[i].size(); j++)                         // :441-445 (sic: j < degree)
                        if (std::binary_search(nl[i].begin(), nl[i].end(), j) && newclusters[j] >= 0) neighbors[newclusters[j]] += 1;
                } else if ((size_t)i < adj.size()) {
                    for (int nb : adj[i]) if (newclusters[nb] >= 0) neighbors[newclusters[nb]] += 1;
                }
                int max_index = 0, max_value = 0, second_index = 0, second_value = 0;
                for (int j = 0; j < (int)neighbors.size(); j++) {
                    if (neighbors[j] > max_value) { second_value = max_value; second_index = max_index; max_value = neighbors[j]; max_index = j; }
                    else if (neighbors[j] > second_value) { second_value = neighbors[j]; second_index = j; }
                }
                if (max_value > 0 && max_index != clusterToTest) {
                    countOf[newclusters[i]]--; countOf[max_index]++; changes++; newclusters[i] = max_index;
                } else if (max_value > 0 && max_value <= 2 * second_value) {
                    countOf[newclusters[i]]--; countOf[second_index]++; newclusters[i] = second_index; changes++;
                }
            }
            iters++;
        }
        tested.emplace(clusterToTest);
        if (countOf[clusterToTest] == 0) { clusters = newclusters; initialCount = countOf; }
    }
}

// separate_reads.cpp:1007-1327
std::vector<int> merge_wrongly_split_haplotypes(const std::vector<int>& clusteredReads, const std::vector<Column>& snps,
                                                const std::vector<std::vector<int>>& nl,
                                                const std::vector<std::vector<int>>& adj, bool low_memory,
                                                int posstart, int posend) {
    std::set<int> listOfGroups;
    std::map<int, int> indexOfGroups;       // robin_hood maps used for lookup only in the reference
    int index = 0;
    for (size_t read = 0; read < clusteredReads.size(); read++) {
        if (clusteredReads[read] > -1) {
            listOfGroups.emplace(clusteredReads[read]);
            if (indexOfGroups.find(clusteredReads[read]) == indexOfGroups.end()) { indexOfGroups[clusteredReads[read]] = index; index++; }
        }
    }
    const size_t G = listOfGroups.size();
    std::vector<std::vector<int>> incompat(G, std::vector<int>(G, 0));
    std::vector<std::vector<int>> pos_last(G, std::vector<int>(G, -10));
    if (G <= 1) {
        std::vector<int> one(clusteredReads.size(), 0);
        for (size_t r = 0; r < clusteredReads.size(); r++) if (clusteredReads[r] == -2) one[r] = -2;
        return one;
    }
    for (const Column& snp : snps) {
        if (!(snp.pos >= posstart && snp.pos < posend)) continue;
        std::map<int, unsigned char> cluster_to_majority_base;               // lookups default-insert 0, as operator[] does
        std::map<int, RHMap<unsigned char, int>> bases_in_each_cluster;      // inner iteration order matters (:1090-1099)
        std::map<int, int> nb_bases_in_each_cluster;
        for (size_t r = 0; r < snp.readIdxs.size(); r++) {
            int read = (int)snp.readIdxs[r];
            unsigned char base = snp.content[r];
            int cluster = clusteredReads[read];
            if (cluster > -1) {
                RHMap<unsigned char, int>& m = bases_in_each_cluster[cluster];
                if (!m.contains(base)) m[base] = 0;
                m[base]++;
                nb_bases_in_each_cluster[cluster]++;
            }
        }
        std::set<unsigned char> maxbases;
        for (auto& cl : bases_in_each_cluster) {
            int secondMax = 0, mx = 0;
            char maxBase = ' ';
            cl.second.for_each([&](unsigned char k, int v) {
                if (v >= mx) { maxBase = (char)k; secondMax = mx; mx = v; }
                else if (v > secondMax) secondMax = v;
            });
            if (secondMax * 2 > mx || nb_bases_in_each_cluster[cl.first] * 0.5 > mx) maxBase = ' ';
            cluster_to_majority_base[cl.first] = (unsigned char)maxBase;
            if (maxBase != ' ') maxbases.emplace((unsigned char)maxBase);
        }
        if (maxbases.size() <= 1) continue;
        for (int g1 : listOfGroups) {
            for (int g2 : listOfGroups) {
                if (cluster_to_majority_base[g1] != ' ' && cluster_to_majority_base[g2] != ' ' && g1 > g2) {
                    int i1 = indexOfGroups[g1], i2 = indexOfGroups[g2];
                    if (cluster_to_majority_base[g1] != cluster_to_majority_base[g2] && snp.pos - pos_last[i1][i2] > 10) {
                        incompat[i1][i2] += 1; incompat[i2][i1] += 1;
                        pos_last[i1][i2] = snp.pos; pos_last[i2][i1] = snp.pos;
                    }
                }
            }
        }
    }

    std::map<std::pair<int, int>, double> links;
    std::map<int, int> links_in;
    auto count_link = [&](int read1, int read2) {
        int c1 = clusteredReads[read1], c2 = clusteredReads[read2];
        if (c1 != c2) links[std::make_pair(c1, c2)] += 1;
        links_in[c1] += 1;
    };
    if (low_memory) {
        for (size_t r1 = 0; r1 < clusteredReads.size(); r1++)
            for (size_t r2 = 0; r2 < clusteredReads.size(); r2++)
                if (r1 < nl.size() && std::binary_search(nl[r1].begin(), nl[r1].end(), (int)r2)) count_link((int)r1, (int)r2);
    } else {
        for (size_t k = 0; k < adj.size(); k++) for (int row : adj[k]) count_link(row, (int)k);
    }
    for (auto& l : links) l.second = l.second / links_in[l.first.first];
    std::vector<std::pair<std::pair<int, int>, double>> sorted_links(links.begin(), links.end());
    std::sort(sorted_links.begin(), sorted_links.end(),
              [](const std::pair<std::pair<int, int>, double>& a, const std::pair<std::pair<int, int>, double>& b) { return a.second > b.second; });

    std::map<int, int> o2n;
    for (int g : listOfGroups) o2n[g] = g;
    o2n[-1] = -1; o2n[-2] = -2;
    for (auto& pc : sorted_links) {
        if (!(pc.second > 0.01)) continue;
        int c1 = pc.first.first, c2 = pc.first.second;
        if (o2n[c1] == o2n[c2]) continue;
        bool bad = false;
        for (int g1 : listOfGroups) {
            if (o2n[g1] != o2n[c1]) continue;
            for (int g2 : listOfGroups)
                if (o2n[g2] == o2n[c2] && incompat[indexOfGroups[g1]][indexOfGroups[g2]] > 1) bad = true;
        }
        if (!bad) {
            for (int g2 : listOfGroups) if (o2n[g2] == o2n[c2]) o2n[g2] = o2n[c1];
        }
    }
    std::map<int, int> new_index;
    int ni = 0;
    for (int g : listOfGroups) if (new_index.find(o2n[g]) == new_index.end()) { new_index[o2n[g]] = ni; ni++; }
    for (int g : listOfGroups) o2n[g] = new_index[o2n[g]];
    std::vector<int> out(clusteredReads.size(), -1);
    for (size_t r = 0; r < clusteredReads.size(); r++) out[r] = o2n[clusteredReads[r]];
    return out;
}

thread_local std::vector<WindowTap>* g_window_taps = nullptr;

// separate_reads.cpp:840-885
static std::vector<int> merge_clusterings(const std::vector<std::vector<int>>& localClusters,
                                          const std::vector<std::vector<int>>& graph, const std::vector<bool>& mask,
                                          unsigned seed) {
    std::vector<double> agg(localClusters[0].size(), 0);
    for (size_t i = 0; i < localClusters.size(); i++)
        for (size_t j = 0; j < localClusters[i].size(); j++) agg[j] += localClusters[i][j] * std::pow(2.0, (double)i);
    std::unordered_map<double, int> ids;
    std::vector<int> ints;
    int index = 0;
    for (size_t i = 0; i < agg.size(); i++) {
        auto it = ids.find(agg[i]);
        if (it == ids.end()) { ids[agg[i]] = index; ints.push_back(index); index++; }
        else ints.push_back(it->second);
    }
    for (size_t i = 0; i < ints.size(); i++) if (!mask[i]) ints[i] = -2;
    return chinese_whispers(graph, ints, mask, seed);
}

// separate_reads.cpp:897-994. `graph` is the structure the *global* low_memory flag selects (:1708 quirk):
// neighbour list when low_memory, Eigen adjacency (possibly never built, i.e. empty) otherwise.
static void finalize_clustering(const std::vector<Column>& snps, const std::vector<std::vector<int>>& localClusters,
                                const std::vector<std::vector<int>>& nl, const std::vector<std::vector<int>>& adj,
                                bool low_memory, const std::vector<bool>& mask, std::vector<int>& haplotypes,
                                int posstart, int posend, unsigned seed) {
    if (localClusters.size() == 0) {
        for (size_t r = 0; r < mask.size(); r++) haplotypes[r] = mask[r] ? -1 : -2;
        return;
    }
    const std::vector<std::vector<int>>& graph = low_memory ? nl : adj;
    std::vector<int> clusteredReads = merge_clusterings(localClusters, graph, mask, seed);
    std::map<int, int> sizes;
    for (size_t r = 0; r < clusteredReads.size(); r++) {
        if (!mask[r]) clusteredReads[r] = -2;
        else sizes[clusteredReads[r]] += 1;
    }
    for (size_t r = 0; r < clusteredReads.size(); r++)
        if (sizes[clusteredReads[r]] < 5 && clusteredReads[r] != -2) clusteredReads[r] = -1;
    std::vector<int> merged = clusteredReads;
    std::map<int, int> toHap;
    int hap = 0;
    for (size_t r = 0; r < merged.size(); r++) {
        if (merged[r] > -1) {
            if (toHap.find(merged[r]) == toHap.end()) { toHap[merged[r]] = hap; hap++; }
            merged[r] = toHap[merged[r]];
        }
    }
    haplotypes = chinese_whispers(graph, merged, mask, seed);
    if (g_window_taps && !g_window_taps->empty()) { WindowTap& t = g_window_taps->back(); for (int r : t.mask_ids) t.third.push_back(haplotypes[(size_t)r]); }
    std::map<int, int> toIndex;
    toIndex[-1] = -1;
    if (snps.size() == 0) toIndex[-1] = 0;
    toIndex[-2] = -2;
    int index_h = 0;
    for (int h : haplotypes) if (toIndex.find(h) == toIndex.end()) { toIndex[h] = index_h; index_h++; }
    for (size_t r = 0; r < haplotypes.size(); r++) haplotypes[r] = toIndex[haplotypes[r]];
    merge_close_clusters(nl, adj, low_memory, haplotypes, mask, seed);
    haplotypes = merge_wrongly_split_haplotypes(haplotypes, snps, nl, adj, low_memory, posstart, posend);
}

// separate_reads.cpp:1341-1396
static std::vector<int> merge_haplotypes_to_fit_within_limit(int max_haplotypes, const std::vector<int>& clusters,
                                                             const std::vector<bool>& mask,
                                                             const std::vector<std::vector<int>>& graph, unsigned seed) {
    std::map<int, int> count;
    for (int c : clusters) if (c >= 0) count[c] += 1;
    if ((int)count.size() <= max_haplotypes) return clusters;
    std::vector<std::pair<int, int>> v;
    for (auto& c : count) v.push_back(std::make_pair(c.second, c.first));
    std::sort(v.begin(), v.end(), std::greater<std::pair<int, int>>());
    std::set<int> kept;
    for (int i = 0; i < max_haplotypes; i++) kept.insert(v[i].second);
    std::vector<int> nc = clusters;
    for (size_t i = 0; i < clusters.size(); i++) if (clusters[i] >= 0 && kept.find(clusters[i]) == kept.end()) nc[i] = -1;
    return chinese_whispers(graph, nc, mask, seed);
}

// separate_reads.cpp:1466-1498
int choose_window_size(const std::vector<ColContig>& cs, bool amplicon, std::vector<float>* coverages) {
    int numberOfReadsHere = 0, sumLength = 0, above4000 = 0;
    if (coverages) coverages->assign(cs.size(), 0);
    for (size_t i = 0; i < cs.size(); i++) {
        float cov = 0;
        for (auto& r : cs[i].readLimits) {
            numberOfReadsHere++;
            sumLength = (int)((unsigned)sumLength + (unsigned)(r.second - r.first + 1));   // int accumulator (:1468); wraps like the 2's-complement add
            cov += r.second - r.first + 1;
            if (r.second - r.first + 1 > 4000) above4000++;
        }
        cov /= cs[i].length;
        if (coverages) (*coverages)[i] = cov;
    }
    double meanLength = sumLength / double(numberOfReadsHere);
    int w = 2000;
    if (above4000 < 20 && meanLength < 4000 && meanLength > 2000) w = 1000;
    else if (above4000 < 20 && meanLength < 2000) w = 500;
    if (amplicon) { w = 0; for (auto& c : cs) w = std::max(w, int(c.length)); }
    return w;
}

// separate_reads.cpp:1508-1739 (one contig)
std::vector<Window> separate_reads_on_contig(const ColContig& c, int sizeOfWindow, float errorRate, bool low_memory,
                                             bool low_memory_now, int ploidy, unsigned seed) {
    std::vector<Window> out;
    const std::vector<Column>& snps = c.snps;
    const int N = (int)c.read_lines.size();
    const long L = c.length;
    if (snps.size() == 0) return out;
    std::vector<int> sim, diff;
    if (!low_memory_now) list_similarities_and_differences(snps, N, sim, diff);

    int suspectIdx = 0;
    int chunk = -1;
    int upperBound;
    while ((long)(chunk + 1) * sizeOfWindow + 100 <= L) {
        chunk++;
        upperBound = (chunk + 1) * sizeOfWindow;
        if ((long)(chunk + 1) * sizeOfWindow + 100 > L) upperBound = (int)L + 1;

        if ((size_t)suspectIdx >= snps.size() || snps[suspectIdx].pos > upperBound - 1) {       // :1565-1587
            std::vector<int> readsHere(N, -2);
            for (size_t r = 0; r < c.readLimits.size(); r++) {
                int pointLeft = chunk * sizeOfWindow;
                int pointRight = std::min(upperBound - 1, int(L));
                int middle = (pointLeft + pointRight) / 2;
                if (middle < 500) middle = std::min(500, int(L / 2));
                if (middle > int(L) - 500) middle = std::max(int(L / 2), int(L) - 500);
                if (c.readLimits[r].first <= middle && c.readLimits[r].second >= middle) readsHere[r] = 0;
            }
            out.push_back(Window{chunk * sizeOfWindow, std::min(upperBound - 1, int(L)), readsHere});
            continue;
        }

        std::vector<bool> mask(N, false);                                                       // :1590-1622
        if (chunk == 0) {
            while ((size_t)suspectIdx < snps.size() - 1 && snps[suspectIdx].pos < chunk * sizeOfWindow + 0.2 * sizeOfWindow
                   && snps[suspectIdx + 1].pos < chunk * sizeOfWindow + 0.4 * sizeOfWindow)
                suspectIdx++;
        }
        for (unsigned r : snps[suspectIdx].readIdxs) mask[r] = true;
        while ((size_t)suspectIdx < snps.size() && snps[suspectIdx].pos < upperBound - 1) suspectIdx++;
        if (suspectIdx > 0) suspectIdx--;
        if ((long)(chunk + 1) * sizeOfWindow + 100 > L) {
            while (suspectIdx > 0 && snps[suspectIdx].pos > upperBound - 1 - 0.2 * sizeOfWindow
                   && snps[suspectIdx - 1].pos > upperBound - 1 - 0.4 * sizeOfWindow)
                suspectIdx--;
        }
        unsigned idxmask = 0;
        for (size_t r = 0; r < snps[suspectIdx].readIdxs.size(); r++) {
            while (idxmask < snps[suspectIdx].readIdxs[r]) { mask[idxmask] = false; idxmask++; }
            idxmask++;
        }
        suspectIdx++;

        std::vector<std::vector<int>> nl(N), adj;
        if (!low_memory_now) create_read_graph_matrix(mask, sim, diff, N, errorRate, adj);
        else create_read_graph_low_memory(snps, mask, nl, errorRate);
        const std::vector<std::vector<int>>& graph_now = low_memory_now ? nl : adj;

        std::vector<std::vector<int>> localClusters;                                            // :1673-1705
        int lastpos = -10;
        if (g_window_taps) { g_window_taps->emplace_back(); g_window_taps->back().start = chunk * sizeOfWindow; for (int r = 0; r < N; ++r) if (mask[(size_t)r]) g_window_taps->back().mask_ids.push_back(r); }
        int snp_index = -1;
        for (const Column& snp : snps) {
            ++snp_index;
            if (snp.pos >= chunk * sizeOfWindow && snp.pos < chunk * sizeOfWindow + sizeOfWindow && snp.pos > lastpos + 10) {
                lastpos = snp.pos;
                std::map<unsigned char, int> charToIndex;
                std::vector<int> start(N);
                std::iota(start.begin(), start.end(), 0);
                for (size_t r = 0; r < snp.content.size(); r++) {
                    if (mask[snp.readIdxs[r]]) {
                        if (charToIndex.find(snp.content[r]) == charToIndex.end()) charToIndex[snp.content[r]] = (int)snp.readIdxs[r];
                        start[snp.readIdxs[r]] = charToIndex[snp.content[r]];
                    }
                }
                localClusters.push_back(chinese_whispers(graph_now, start, mask, seed));
                if (g_window_taps) { WindowTap& t = g_window_taps->back(); t.run_snp.push_back(snp_index); t.runs.emplace_back(); for (int r : t.mask_ids) t.runs.back().push_back(localClusters.back()[(size_t)r]); }
            }
        }
        std::vector<int> haplotypes(N, -2);
        // :1708 passes the global low_memory, not low_memory_now
        finalize_clustering(snps, localClusters, nl, adj, low_memory, mask, haplotypes, chunk * sizeOfWindow,
                            chunk * sizeOfWindow + sizeOfWindow, seed);
        if (ploidy > 0)                                                                         // :1711-1715
            haplotypes = merge_haplotypes_to_fit_within_limit(ploidy, haplotypes, mask, low_memory ? nl : adj, seed);
        out.push_back(Window{chunk * sizeOfWindow, std::min(upperBound - 1, int(L)), haplotypes});
    }
    return out;
}

// separate_reads.cpp:1398-1790
int run_separate_reads(int argc, char** argv) {
    const char* usage = "Usage: ./separate_reads <columns> <num_threads> <error_rate> <ploidy_of_contigs> <low_memory> <rarest-strain-abundance> <amplicon> <outfile> <DEBUG>";
    if (argc != 10) {
        std::cout << usage << std::endl;
        if (argc == 2 && (argv[1] == std::string("-h") || argv[1] == std::string("--help"))) return 0;
        return 1;
    }
    std::string columns_file = argv[1], ploidy_file = argv[4], outfile = argv[8];
    float errorRate = (float)atof(argv[3]);
    bool amplicon = bool(atoi(argv[7]));
    bool low_memory = bool(atoi(argv[5]));
    float rsa = (float)atof(argv[6]);
    // :1420-1426: the else-branch declares a shadowing local, so max_coverage is uninitialised whenever
    // rarest_strain_abundance != 0. Observed behaviour of the compiled reference: unlimited.
    int max_coverage = 1000000000;
    unsigned seed = 12345u;
    if (const char* s = std::getenv("HS_ORACLE_SEED")) seed = (unsigned)std::strtoul(s, nullptr, 10);
    { std::ofstream o(outfile); }
    std::vector<ColContig> cs = parse_column_file(columns_file, max_coverage, rsa);

    std::map<std::string, int> ploidy_of;
    bool have_ploidy = false;
    { std::ifstream pf(ploidy_file);
      if (pf) { have_ploidy = true; std::string line;
        while (std::getline(pf, line)) { std::istringstream iss(line); std::string ctg; int p; if (!(iss >> ctg >> p)) break; ploidy_of[ctg] = p; } } }
    std::vector<float> coverages;
    int w = choose_window_size(cs, amplicon, &coverages);
    std::ofstream out(outfile, std::ios_base::app);
    for (size_t n = 0; n < cs.size(); n++) {
        bool low_memory_now = low_memory || coverages[n] > 1000;
        if (cs[n].snps.size() == 0) continue;
        int ploidy = 0;
        if (have_ploidy) {
            std::string rest = cs[n].contig_line.substr(cs[n].contig_line.find("\t") + 1);
            std::string nm = rest.substr(0, rest.find("\t"));
            auto it = ploidy_of.find(nm);
            if (it != ploidy_of.end()) ploidy = it->second;
        }
        std::vector<Window> ws = separate_reads_on_contig(cs[n], w, errorRate, low_memory, low_memory_now, ploidy, seed);
        out << cs[n].contig_line << std::endl;
        for (auto& r : cs[n].read_lines) out << r << "\n";
        for (auto& win : ws) {
            out << "GROUP\t" << win.start << "\t" << win.end << "\t";
            std::string a, b;
            for (size_t h = 0; h < win.labels.size(); h++)
                if (win.labels[h] != -2) { a += std::to_string(h) + ","; b += std::to_string(win.labels[h]) + ","; }
            out << a << "\t" << b << "\n";
        }
    }
    return 0;
}

// ---- A1 oracle: plain DP edit distance (edlib.h:36-62 semantics) --------------------------------
int edit_distance(const unsigned char* q, int qn, const unsigned char* t, int tn, int mode, int* end_loc) {
    // rows = query, columns = target. NW: D[0][j] = j; HW/SHW... SHW: gaps after query end in target free (prefix of target);
    // HW: target start and end gaps free.
    std::vector<int> prev(tn + 1), cur(tn + 1);
    for (int j = 0; j <= tn; j++) prev[j] = (mode == 2) ? 0 : j;
    for (int i = 1; i <= qn; i++) {
        cur[0] = i;
        for (int j = 1; j <= tn; j++) {
            int sub = prev[j - 1] + (q[i - 1] != t[j - 1]);
            int del = prev[j] + 1, ins = cur[j - 1] + 1;
            cur[j] = std::min(sub, std::min(del, ins));
        }
        std::swap(prev, cur);
    }
    if (mode == 0) { if (end_loc) *end_loc = tn - 1; return prev[tn]; }
    // The reference's edlib pads the query to a multiple of 64 rows and reads the score of column c off column c + W (W = padding
    // rows, edlib.cpp:664-690): with W > 0 the columns "before the target" (score = query length) take part and win ties; with
    // W == 0 (and a non-empty query) there are none and the first real column that reaches the minimum is reported.
    int best = prev[0], bj = 0;
    if (qn > 0 && qn % 64 == 0 && tn > 0) { best = prev[1]; bj = 1; }
    for (int j = 1; j <= tn; j++) if (prev[j] < best) { best = prev[j]; bj = j; }
    if (end_loc) *end_loc = bj - 1;   // first (leftmost) end location, 0-based inclusive, as edlib reports endLocations[0]
    return best;
}

}  // namespace hso
