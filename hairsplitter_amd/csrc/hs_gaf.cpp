// hs_gaf.cpp -- the consumer side of the .gro file: how the reads thread through the contigs that the
// read separation implies, written as a GAF. Replaces parse_split_file (create_new_contigs.cpp:41-175), merge_intervals
// (:1427-1534) with stitch (:833-903), find_paths (:959-1112) and output_GAF (:1128-1419) of the reference's stage 5.
// Pure host code (text and small integer sets; nothing here is worth a kernel): it exists so that the labels this
// library produces can be checked end to end against what the reference's next stage makes of them, and so that a host
// that holds the stage-4 result in memory does not have to write and re-parse the .gro text.
//
// Reference behaviours kept on purpose:
//  * labels are attached to alignment records through the READ NAME (:93-95,141-143): when a read has several records
//    on one contig only the last one carries labels, the others count as absent (-2);
//  * merge_intervals dereferences begin() of an empty std::set (:1493-1500). The value it reads cannot change the outcome:
//    a set is still empty there only if the other sets already cover every cluster on the left, which k-1 one-element sets
//    cannot do for k clusters, so the junction is non-trivial whatever is read; 0 is used here;
//  * conversion[] of a label that was never stitched yields 0 (unordered_map::operator[], :1519);
//  * the path of a read is sorted with std::sort on the start coordinate only (:1290): the same call is used here, so
//    equal keys come out in the same order;
//  * find_paths marks a neighbour as visited before the length test on one of its four branches only (:981).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "hs_host.h"
#include "hs_driver.h"
#include "../../include/hairsplitter_hip.h"

namespace hs {

namespace {

struct GafLink { long n1; int e1; long n2; int e2; };
typedef std::pair<std::pair<int, int>, std::vector<int>> GafInterval;             // (start, end), label of every record of the contig
typedef std::vector<std::pair<std::string, bool>> GafSteps;                      // (new contig name, same orientation as the read)
struct GafPath { std::pair<int, int> on_read; GafSteps steps; long backbone; };

// "allreads" of the reference: reads first (file order), then the contigs (GFA order)
struct GafModel {
    const CvFileInput* in = nullptr;
    long n_reads = 0, n_contigs = 0;
    std::vector<GafLink> links;
    std::vector<std::vector<size_t>> links_left, links_right;                    // per contig (Read::add_link, read.cpp:70-77)
    std::unordered_map<long, std::vector<GafInterval>> partitions;               // keyed by allreads index of the contig
    const std::string& name(long i) const { return i < n_reads ? in->read_names[(size_t)i] : in->contig_names[(size_t)(i - n_reads)]; }
    long size(long i) const {
        return i < n_reads ? (long)(in->read_off[(size_t)i + 1] - in->read_off[(size_t)i])
                           : (long)(in->contig_off[(size_t)(i - n_reads) + 1] - in->contig_off[(size_t)(i - n_reads)]);
    }
    bool has_partitions(long i) const { return partitions.find(i) != partitions.end(); }
};

// 'L' lines of the GFA (input_output.cpp:173-252): + on the first segment = its right end (1), + on the second = its left end (0)
int parse_links(const std::string& gfa, GafModel& m) {
    std::unordered_map<std::string, long> idx;
    for (long c = 0; c < m.n_contigs; ++c) idx[m.in->contig_names[(size_t)c]] = m.n_reads + c;   // later duplicates win, like indices[name] = id
    std::ifstream f(gfa);
    if (!f) { set_error("cannot open " + gfa); return HS_EIO; }
    std::string line, field;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] != 'L') continue;
        std::istringstream ls(line);
        int k = 0;
        GafLink l{0, 0, 0, 0};
        std::string n1, n2;
        bool ok = true;
        while (std::getline(ls, field, '\t')) {
            if (k == 1) n1 = field;
            else if (k == 2) { if (field != "+" && field != "-") ok = false; l.e1 = field == "+" ? 1 : 0; }
            else if (k == 3) n2 = field;
            else if (k == 4) { if (field != "+" && field != "-") ok = false; l.e2 = field == "+" ? 0 : 1; }
            ++k;
        }
        if (!ok || k < 5 || !idx.count(n1) || !idx.count(n2)) { set_error("invalid link in " + gfa + ": " + line); return HS_EINVAL; }
        l.n1 = idx[n1]; l.n2 = idx[n2];
        m.links.push_back(l);
        const size_t li = m.links.size() - 1;
        (l.e1 == 0 ? m.links_left : m.links_right)[(size_t)(l.n1 - m.n_reads)].push_back(li);
        (l.e2 == 0 ? m.links_left : m.links_right)[(size_t)(l.n2 - m.n_reads)].push_back(li);
    }
    return HS_OK;
}

// the .gro text -> partitions (create_new_contigs.cpp:41-175)
int parse_gro(const std::string& path, GafModel& m) {
    std::ifstream f(path);
    if (!f.good()) { set_error("could not open file " + path); return HS_EIO; }
    const CvFileInput& in = *m.in;
    std::unordered_map<std::string, long> name_of;       // name -> allreads index (reads and contigs share the namespace, :47-50)
    for (long i = 0; i < m.n_reads + m.n_contigs; ++i) name_of[m.name(i)] = i;
    std::string line;
    long contig = 0;
    std::unordered_map<std::string, int> name_of_neighbors;
    std::vector<std::string> read_names;
    while (std::getline(f, line)) {
        std::istringstream iss(line);
        std::string cat;
        iss >> cat;
        if (cat == "CONTIG") {
            name_of_neighbors.clear(); read_names.clear();
            std::string cname;
            iss >> cname;
            auto it = name_of.find(cname);
            contig = it == name_of.end() ? 0 : it->second;   // operator[] of the reference: an unknown name becomes index 0
            if (contig < m.n_reads) { set_error("contig of the .gro file is not in the assembly: " + cname); return HS_EINVAL; }
            m.partitions[contig] = {};
            const int c = (int)(contig - m.n_reads);
            for (int r = in.contig_rec_off[(size_t)c]; r < in.contig_rec_off[(size_t)c + 1]; ++r)
                name_of_neighbors[in.read_names[(size_t)in.rec_read[(size_t)r]]] = r - in.contig_rec_off[(size_t)c];
        } else if (cat == "READ") {
            std::string rname;
            iss >> rname;
            read_names.push_back(rname);
        } else if (cat == "GROUP") {
            int start = 0, end = 0;
            std::string idx_s, lab_s;
            iss >> start >> end >> idx_s >> lab_s;
            if (idx_s == "," || lab_s == ",") continue;
            std::vector<int> idxs, labs;
            std::string tok;
            { std::istringstream a(idx_s); while (std::getline(a, tok, ',')) idxs.push_back(std::stoi(tok)); }
            { std::istringstream a(lab_s); while (std::getline(a, tok, ',')) labs.push_back(std::stoi(tok)); }
            if (!m.partitions.count(contig)) { set_error("GROUP line before any CONTIG line in " + path); return HS_EINVAL; }
            const int c = (int)(contig - m.n_reads);
            std::vector<int> full((size_t)(in.contig_rec_off[(size_t)c + 1] - in.contig_rec_off[(size_t)c]), -2);
            for (size_t r = 0; r < idxs.size(); ++r) {
                if (idxs[r] < 0 || (size_t)idxs[r] >= read_names.size() || r >= labs.size()) { set_error("malformed GROUP line in " + path); return HS_EINVAL; }
                auto nb = name_of_neighbors.find(read_names[(size_t)idxs[r]]);
                if (nb != name_of_neighbors.end()) full[(size_t)nb->second] = labs[r];
            }
            m.partitions[contig].push_back(std::make_pair(std::make_pair(start, end), full));
        }
    }
    return HS_OK;
}

// which clusters of the next interval each cluster continues into (create_new_contigs.cpp:833-903)
std::unordered_map<int, std::set<int>> stitch(const std::vector<int>& par, const std::vector<int>& neighbor) {
    std::unordered_map<int, std::unordered_map<int, int>> fit_left, fit_right;
    std::unordered_map<int, int> cluster_size;
    std::unordered_map<int, std::set<int>> st;
    for (size_t r = 0; r < par.size(); ++r) {
        if (par[r] > -1 && neighbor[r] > -1) {
            if (fit_left.find(par[r]) != fit_left.end()) { fit_left[par[r]][neighbor[r]] += 1; cluster_size[par[r]] += 1; }
            else { fit_left[par[r]][neighbor[r]] = 1; cluster_size[par[r]] = 1; st[par[r]] = {}; }
            fit_right[neighbor[r]][par[r]] += 1;
        }
    }
    for (auto& fit : fit_left)
        for (auto& cand : fit.second)
            if (cand.second >= std::min(5.0, 0.7 * cluster_size[fit.first])) st[fit.first].emplace(cand.first);
    for (auto& fit : fit_right)
        for (auto& cand : fit.second)
            if (cand.second >= std::min(5.0, 0.7 * cluster_size[cand.first])) st[cand.first].emplace(fit.first);
    return st;
}

struct GafStats { long intervals_in = 0, intervals_out = 0, empty_stitch_sets = 0, chained = 0, ambiguous = 0, lines = 0; };

// consecutive intervals whose clusters map one to one are fused (create_new_contigs.cpp:1427-1534)
void merge_intervals(std::vector<GafInterval>& ivs, GafStats& st) {
    st.intervals_in += (long)ivs.size();
    if (ivs.empty()) return;
    std::vector<GafInterval> out;
    std::vector<int> group = ivs[0].second;
    int c_start = ivs[0].first.first, c_end = ivs[0].first.second;
    for (size_t k = 1; k < ivs.size(); ++k) {
        const std::vector<int>& there = ivs[k].second;
        std::unordered_map<int, std::set<int>> stitch_left = stitch(group, there);
        std::unordered_map<int, std::set<int>> stitches = stitch_left;
        std::set<int> left(group.begin(), group.end()), right(there.begin(), there.end());
        left.erase(-1); left.erase(-2); right.erase(-1); right.erase(-2);
        std::set<int> stitched;
        for (auto& s : stitches) for (int nb : s.second) stitched.emplace(nb);
        for (int cl : left)
            if (stitched.find(cl) == stitched.end())
                for (auto& s : stitch_left) stitches[s.first].emplace(cl);
        bool trivial = true;
        std::unordered_map<int, int> conversion;
        std::set<int> seen;
        for (auto& s : stitches) {
            if (s.second.size() > 1) { trivial = false; continue; }
            if (s.second.empty()) st.empty_stitch_sets++;
            const int only = s.second.empty() ? 0 : *s.second.begin();   // see the header: the value read from an empty set is immaterial
            if (seen.find(only) != seen.end()) trivial = false; else seen.emplace(only);
            conversion[only] = s.first;
        }
        if (seen.size() < left.size() || left.size() != right.size()) trivial = false;
        if (!trivial) {
            out.push_back(std::make_pair(std::make_pair(c_start, c_end), group));
            group = there; c_start = ivs[k].first.first; c_end = ivs[k].first.second;
        } else {
            c_end = ivs[k].first.second;
            for (size_t r = 0; r < group.size(); ++r)
                if (group[r] < 0 && there[r] > -1) group[r] = conversion[there[r]];
        }
    }
    out.push_back(std::make_pair(std::make_pair(c_start, c_end), group));
    st.intervals_out += (long)out.size();
    ivs.swap(out);
}

// name under which a neighbouring contig appears in the new graph when it is crossed on the way (:984-990)
void crossed_name(const GafModel& m, long nb, std::string& name, int& copies) {
    name = m.name(nb); copies = 1;
    auto it = m.partitions.find(nb);
    if (it == m.partitions.end()) return;
    if (it->second.size() == 1 && it->second[0].second.size() == 1)
        name = m.name(nb) + "_" + std::to_string(it->second[0].first.first) + "_" + std::to_string(it->second[0].second[0]);
    else copies = 2;
}

// all walks from one contig end to another that are shorter than max_length (create_new_contigs.cpp:959-1112)
std::vector<GafSteps> find_paths(const GafModel& m, long contig1, int end1, long contig2, int end2, int max_length, GafSteps path,
                                 std::set<std::string> visited) {
    std::vector<GafSteps> all;
    const std::vector<size_t>& ls = end1 == 1 ? m.links_right[(size_t)(contig1 - m.n_reads)] : m.links_left[(size_t)(contig1 - m.n_reads)];
    for (size_t li : ls) {
        const GafLink& l = m.links[li];
        long nb; int nb_end; bool first_form;
        if (l.n1 == contig1 && l.e1 == end1) { nb = l.n2; nb_end = l.e2; first_form = true; }
        else if (l.n2 == contig1 && l.e2 == end1) { nb = l.n1; nb_end = l.e1; first_form = false; }
        else continue;
        if (nb == contig2 && nb_end == end2) { all.push_back(path); continue; }
        if (visited.find(m.name(nb)) != visited.end()) return {path, path};   // two copies: "ambiguous"
        const size_t len = (size_t)m.size(nb);
        const bool early_mark = end1 == 1 && first_form;                      // :981 marks before the length test
        if (early_mark) visited.emplace(m.name(nb));
        if (max_length > 0 && len < (size_t)max_length) {
            std::string nm; int copies;
            crossed_name(m, nb, nm, copies);
            path.push_back(std::make_pair(nm, (bool)(1 - nb_end)));
            if (!early_mark) visited.emplace(m.name(nb));
            std::vector<GafSteps> sub = find_paths(m, nb, 1 - nb_end, contig2, end2, (int)((size_t)max_length - len), path, visited);
            for (const GafSteps& p : sub) for (int i = 0; i < copies; ++i) all.push_back(p);
            path.pop_back();
        }
    }
    return all;
}

bool is_marker(const GafSteps& s) {
    const std::string& last = s.back().first;
    const char c = last[last.size() - 1];
    return c == '&' || c == '+' || c == '-';
}

void push_end_marker(GafSteps& steps, bool strand, bool firsthere, bool lasthere) {   // :1218-1226, :1266-1274
    const bool open_end = (strand && !lasthere) || (!strand && !firsthere);
    const bool open_begin = (strand && !firsthere) || (!strand && !lasthere);
    if (open_end && open_begin) steps.push_back(std::make_pair("&", strand));
    else if (open_end) steps.push_back(std::make_pair("+", strand));
    else if (open_begin) steps.push_back(std::make_pair("-", strand));
}

int write_gaf(const GafModel& m, const std::string& out_path, GafStats& st) {
    const CvFileInput& in = *m.in;
    std::vector<std::vector<GafPath>> read_paths((size_t)(m.n_reads + m.n_contigs));
    for (long c = 0; c < m.n_contigs; ++c) {
        const long backbone = m.n_reads + c;
        const std::string& bname = in.contig_names[(size_t)c];
        auto pit = m.partitions.find(backbone);
        const int r0 = in.contig_rec_off[(size_t)c], r1 = in.contig_rec_off[(size_t)c + 1];
        if (pit != m.partitions.end() && !pit->second.empty()) {
            const std::vector<GafInterval>& ivs = pit->second;
            for (int r = r0; r < r1; ++r) {
                const int n = r - r0;
                const bool strand = in.rec_strand[(size_t)r] != 0;
                GafSteps steps;
                short stop = 0;
                bool firsthere = false, lasthere = false;
                int inter = 0;
                for (const GafInterval& iv : ivs) {
                    if (iv.second[(size_t)n] > -1 && stop < 2) {
                        steps.push_back(std::make_pair(bname + "_" + std::to_string(iv.first.first) + "_" + std::to_string(iv.second[(size_t)n]), strand));
                        if (inter == 0) firsthere = true;
                        stop = 1;
                    } else if (stop == 1) stop = 2;
                    inter++;
                }
                if (stop < 2) {   // the stretch after the last interval exists in one copy
                    lasthere = true;
                    steps.push_back(std::make_pair(bname + "_" + std::to_string(ivs.back().first.second + 1) + "_0", strand));
                }
                if (!strand) std::reverse(steps.begin(), steps.end());
                push_end_marker(steps, strand, firsthere, lasthere);
                if (!steps.empty())
                    read_paths[(size_t)in.rec_read[(size_t)r]].push_back(GafPath{std::make_pair(in.rec_r0[(size_t)r], in.rec_r1[(size_t)r]), steps, backbone});
            }
        } else {
            for (int r = r0; r < r1; ++r) {
                const bool strand = in.rec_strand[(size_t)r] != 0;
                const long read = in.rec_read[(size_t)r];
                const int start = in.rec_r0[(size_t)r], end = in.rec_r1[(size_t)r];
                const bool firsthere = start > 100;
                const bool lasthere = (size_t)end < (size_t)m.size(read) - 100;   // size_t arithmetic as in :1257
                GafSteps steps = {std::make_pair(bname, strand)};
                push_end_marker(steps, strand, firsthere, lasthere);
                read_paths[(size_t)read].push_back(GafPath{std::make_pair(start, end), steps, backbone});
            }
        }
    }
    // paths of one read on different contigs are chained when the graph offers exactly one way between them (:1286-1392)
    for (std::vector<GafPath>& rp : read_paths) {
        if (rp.empty()) continue;
        std::sort(rp.begin(), rp.end(), [](const GafPath& x, const GafPath& y) { return x.on_read.first < y.on_read.first; });
        std::vector<GafPath> merged;
        GafPath cur = rp[0];
        for (size_t p = 0; p + 1 < rp.size(); ++p) {
            const GafPath& next = rp[p + 1];
            const long contig = cur.backbone;
            const bool orientation = cur.steps.back().second;
            if (contig != next.backbone) {
                const int max_len = next.on_read.first - cur.on_read.second + 1000;
                std::vector<GafSteps> between = find_paths(m, contig, orientation ? 1 : 0, next.backbone, 1 - (next.steps[0].second ? 1 : 0), max_len, {},
                                                           std::set<std::string>());
                if (is_marker(cur.steps)) cur.steps.pop_back();
                if (between.size() > 1) st.ambiguous++;
                if (between.size() == 1) {
                    st.chained++;
                    cur.steps.insert(cur.steps.end(), between[0].begin(), between[0].end());
                    cur.steps.insert(cur.steps.end(), next.steps.begin(), next.steps.end());
                    cur.backbone = next.backbone;
                } else { merged.push_back(cur); cur = next; }
            } else {
                if (is_marker(cur.steps)) cur.steps.pop_back();
                merged.push_back(cur);
                cur = next;
            }
        }
        if (is_marker(cur.steps)) cur.steps.pop_back();
        merged.push_back(cur);
        rp.swap(merged);
    }
    std::string text;
    for (size_t p = 0; p < read_paths.size(); ++p) {
        for (const GafPath& path : read_paths[p]) {
            if (path.steps.empty()) continue;
            text += m.name((long)p); text += "\t-1\t"; text += std::to_string(path.on_read.first); text += "\t-1\t+\t";
            for (const auto& s : path.steps) { text += s.second ? '>' : '<'; text += s.first; }
            text += "\t-1\t-1\t-1\t-1\t-1\t255\n";
            st.lines++;
        }
    }
    std::FILE* f = std::fopen(out_path.c_str(), "wb");
    if (!f) { set_error("cannot write " + out_path); return HS_EIO; }
    const bool ok = text.empty() || std::fwrite(text.data(), 1, text.size(), f) == text.size();
    std::fclose(f);
    if (!ok) { set_error("short write on " + out_path); return HS_EIO; }
    if (std::getenv("HS_TIMING"))
        std::fprintf(stderr, "[hs timing] gaf: %ld windows -> %ld intervals (%ld empty stitch sets), %ld lines, %ld chained across contigs, %ld ambiguous\n",
                     st.intervals_in, st.intervals_out, st.empty_stitch_sets, st.lines, st.chained, st.ambiguous);
    return HS_OK;
}

int build_model(const std::string& gfa, const CvFileInput& in, GafModel& m) {
    m.in = &in;
    m.n_reads = (long)in.read_names.size(); m.n_contigs = (long)in.contig_names.size();
    m.links_left.assign((size_t)m.n_contigs, {}); m.links_right.assign((size_t)m.n_contigs, {});
    return parse_links(gfa, m);
}

}  // namespace

int gaf_from_files(const std::string& gfa, const std::string& reads, const std::string& sam, const std::string& gro, bool amplicon,
                   const std::string& out_gaf, int n_threads) {
    CvFileInput in;
    if (int rc = load_cv_inputs(gfa, reads, sam, amplicon, in, n_threads)) return rc;
    GafModel m;
    if (int rc = build_model(gfa, in, m)) return rc;
    if (int rc = parse_gro(gro, m)) return rc;
    GafStats st;
    for (auto& kv : m.partitions) merge_intervals(kv.second, st);
    return write_gaf(m, out_gaf, st);
}

// the same from the stage-4 result in memory: window w of contig c covers [win_start, win_end] with one label per record
// (-2 = absent); contigs without SNPs are the ones the .gro writer skips (separate_reads.cpp:1522-1524)
int gaf_from_labels(const std::string& gfa, const CvFileInput& in, int n_contigs, const int64_t* win_off, const int32_t* win_start,
                    const int32_t* win_end, const int64_t* label_off, const int32_t* labels, const uint8_t* contig_has_snps,
                    const std::string& out_gaf) {
    GafModel m;
    if (int rc = build_model(gfa, in, m)) return rc;
    if (n_contigs != (int)m.n_contigs) { set_error("gaf_from_labels: the result does not cover the contigs of the assembly"); return HS_EINVAL; }
    for (long c = 0; c < m.n_contigs; ++c) {
        if (contig_has_snps ? !contig_has_snps[c] : win_off[c + 1] == win_off[c]) continue;   // a contig with SNPs has at least one window
        const int r0 = in.contig_rec_off[(size_t)c], n = in.contig_rec_off[(size_t)c + 1] - r0;
        // the record that represents a read name on this contig: the last one (see the header)
        std::unordered_map<std::string, int> last_of;
        for (int k = 0; k < n; ++k) last_of[in.read_names[(size_t)in.rec_read[(size_t)(r0 + k)]]] = k;
        std::vector<GafInterval>& ivs = m.partitions[m.n_reads + c];
        for (int64_t w = win_off[c]; w < win_off[c + 1]; ++w) {
            if (label_off[w + 1] - label_off[w] != n) { set_error("gaf_from_labels: a window does not hold one label per record"); return HS_EINVAL; }
            const int32_t* lab = labels + label_off[w];
            std::vector<int> full((size_t)n, -2);
            for (int k = 0; k < n; ++k)
                if (lab[k] != -2) full[(size_t)last_of[in.read_names[(size_t)in.rec_read[(size_t)(r0 + k)]]]] = lab[k];
            ivs.push_back(std::make_pair(std::make_pair((int)win_start[w], (int)win_end[w]), full));
        }
    }
    GafStats st;
    for (auto& kv : m.partitions) merge_intervals(kv.second, st);
    return write_gaf(m, out_gaf, st);
}

}  // namespace hs

// ---- C ABI (include/hairsplitter_hip.h) --------------------------------------------------------------
extern "C" int hs_gaf_from_files(const char* gfa, const char* reads, const char* sam, const char* gro, int32_t amplicon, const char* out_gaf,
                                 int32_t n_threads) {
    if (!gfa || !reads || !sam || !gro || !out_gaf) { hs::set_error("hs_gaf_from_files: null path"); return HS_EINVAL; }
    try {
        return hs::gaf_from_files(gfa, reads, sam, gro, amplicon != 0, out_gaf, n_threads < 1 ? 1 : n_threads);
    } catch (const std::exception& e) { hs::set_error(std::string("hs_gaf_from_files: ") + e.what()); return HS_EINVAL; }
}

extern "C" int hs_gaf_from_labels(const char* gfa, const char* reads, const char* sam, int32_t amplicon, int32_t n_contigs,
                                  const int64_t* win_off, const int32_t* win_start, const int32_t* win_end, const int64_t* label_off,
                                  const int32_t* labels, const uint8_t* contig_has_snps, const char* out_gaf, int32_t n_threads) {
    if (!gfa || !reads || !sam || !out_gaf || !win_off || !label_off) { hs::set_error("hs_gaf_from_labels: null argument"); return HS_EINVAL; }
    try {
        hs::CvFileInput in;
        if (int rc = hs::load_cv_inputs(gfa, reads, sam, amplicon != 0, in, n_threads < 1 ? 1 : n_threads)) return rc;
        return hs::gaf_from_labels(gfa, in, n_contigs, win_off, win_start, win_end, label_off, labels, contig_has_snps, out_gaf);
    } catch (const std::exception& e) { hs::set_error(std::string("hs_gaf_from_labels: ") + e.what()); return HS_EINVAL; }
}

// hs_gro_to_gaf <assembly.gfa> <reads> <aln.sam> <reads_haplo.gro> <amplicon:0|1> <out.gaf> [threads]
extern "C" int hs_gro_to_gaf_main(int argc, char** argv) {
    if (argc < 7) {
        std::printf("Usage: hs_gro_to_gaf <original_assembly.gfa> <reads_file> <sam_file> <gro_file> <amplicon:0|1> <output_gaf> [num_threads]\n");
        return argc == 2 ? 0 : 1;
    }
    const int rc = hs_gaf_from_files(argv[1], argv[2], argv[3], argv[4], std::atoi(argv[5]), argv[6], argc > 7 ? std::atoi(argv[7]) : 1);
    if (rc) std::printf("ERROR: %s\n", hs_last_error());
    return rc ? 1 : 0;
}
