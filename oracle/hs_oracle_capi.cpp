// ORACLE (test infrastructure, CPU only): flat-array C entry points so that the parity tests can compare the
// HIP kernels with the CPU restatement buffer by buffer (ctypes). Never linked into the product.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "hs_oracle.h"

extern "C" {

// K1 oracle: generate_msa (call_variants.cpp:50-437) on the flat batch layout of include/hairsplitter_hip.h.
// pile is read-major (pile_off as computed by the caller), rec_stats = {q_end, n_err, n_len, 0} per record.
int hso_pileup(const uint8_t* contig_seq, const int64_t* contig_off, int32_t n_contigs, const uint8_t* read_seq,
               const int64_t* read_off, int32_t n_reads, const int32_t* rec_read, const int32_t* rec_pos,
               const uint8_t* rec_strand, const int64_t* rec_cig_off, const uint32_t* cigar,
               const int32_t* contig_rec_off, const int64_t* pile_off, uint8_t* pile, int32_t* rec_stats,
               float* mean_distance) {
    const char* opc = "MIDNSHP=X";
    std::vector<std::string> reads((size_t)n_reads);
    std::vector<char> loaded((size_t)n_reads, 0);
    for (int c = 0; c < n_contigs; ++c) {
        hso::Contig ctg;
        for (int64_t k = contig_off[c]; k < contig_off[c + 1]; ++k) ctg.seq += "ACGT"[contig_seq[k] & 3];
        for (int r = contig_rec_off[c]; r < contig_rec_off[c + 1]; ++r) {
            hso::Record rec;
            rec.read = rec_read[r]; rec.position_2_1 = rec_pos[r]; rec.strand = rec_strand[r] != 0;
            for (int64_t o = rec_cig_off[r]; o < rec_cig_off[r + 1]; ++o) rec.cigar += std::to_string(cigar[o] >> 4) + opc[cigar[o] & 15u];
            if (!loaded[(size_t)rec.read]) {
                std::string s;
                for (int64_t k = read_off[rec.read]; k < read_off[rec.read + 1]; ++k) s += "ACGT"[read_seq[k] & 3];
                reads[(size_t)rec.read] = s; loaded[(size_t)rec.read] = 1;
            }
            ctg.recs.push_back(rec);
        }
        hso::MsaResult m = hso::generate_msa(ctg, reads);
        if (mean_distance) mean_distance[c] = m.meanDistance;
        for (size_t k = 0; k < ctg.recs.size(); ++k) {
            const int r = contig_rec_off[c] + (int)k;
            rec_stats[4 * r + 0] = m.q_end[k]; rec_stats[4 * r + 1] = (int32_t)m.n_err[k]; rec_stats[4 * r + 2] = (int32_t)m.n_len[k]; rec_stats[4 * r + 3] = 0;
        }
        for (size_t p = 0; p < m.cols.size(); ++p)
            for (size_t k = 0; k < m.cols[p].content.size(); ++k) {
                const int r = contig_rec_off[c] + (int)m.cols[p].readIdxs[k];
                pile[pile_off[r] + ((int64_t)p - rec_pos[r])] = m.cols[p].content[k];
            }
    }
    return 0;
}

// K4 oracle: loops C and D of keep_only_robust_variants (call_variants.cpp:721-764) on flat arrays, through
// distance(Partition, Column) (:778-967) and computeChiSquare (:1135-1163). Same argument meaning as
// hs_column_partition_test; n_reads_of_contig[c] = length of every state array of contig c. chi_out (optional) receives
// the chi-square of each column against the first partition of its contig, tab_out (optional) its 2x2 table.
int hso_column_partition_test(const int64_t* col_off, const int32_t* col_idx, const uint8_t* col_code, const int32_t* col_contig,
                              const uint8_t* col_k0, const uint8_t* col_k1, const int32_t* col_c1, const uint8_t* col_is_cand,
                              int32_t n_cols, const int32_t* part_off, const int64_t* part_state_off, const int8_t* part_state,
                              const int32_t* n_reads_of_contig, int32_t n_contigs, uint8_t* keep, float* chi_out, int32_t* tab_out) {
    std::vector<std::vector<hso::Partition>> parts((size_t)n_contigs);
    for (int c = 0; c < n_contigs; ++c)
        for (int f = part_off[c]; f < part_off[c + 1]; ++f) {
            hso::Partition P;
            for (int r = 0; r < n_reads_of_contig[c]; ++r) {
                const int8_t st = part_state[part_state_off[f] + r];
                if (st == 2) continue;
                P.readIdx.push_back(r); P.mostFrequentBases.push_back(st); P.moreFrequence.push_back(0); P.lessFrequence.push_back(0);
            }
            parts[(size_t)c].push_back(P);
        }
    for (int i = 0; i < n_cols; ++i) {
        hso::Column col;
        for (int64_t k = col_off[i]; k < col_off[i + 1]; ++k) { col.readIdxs.push_back((unsigned)col_idx[k]); col.content.push_back(col_code[k]); }
        col.ref_base = col_k0[i]; col.second_base = col_k1[i];
        const std::vector<hso::Partition>& fin = parts[(size_t)col_contig[i]];
        if (!fin.empty() && (chi_out || tab_out)) {
            hso::DistRes d = hso::distance(fin[0], col, (char)col.ref_base);
            if (chi_out) chi_out[i] = hso::computeChiSquare(d);
            if (tab_out) { tab_out[4 * i] = d.n00; tab_out[4 * i + 1] = d.n01; tab_out[4 * i + 2] = d.n10; tab_out[4 * i + 3] = d.n11; }
        }
        bool kept = false;
        if (col_is_cand[i])
            for (const hso::Partition& P : fin) {
                hso::DistRes d = hso::distance(P, col, (char)col.ref_base);
                if (d.n00 + d.n01 + d.n10 + d.n11 > 0.5 * col.content.size() && hso::computeChiSquare(d) > 15) { kept = true; break; }
            }
        if (!kept && col_c1[i] >= 5) {
            const int rb = col.ref_base, sb = col.second_base;
            if (rb % 5 != sb % 5 && ((sb - '!') % 5 != 4 || (sb / 5 % 5 != rb % 5 && sb / 25 % 5 != rb % 5)))
                for (const hso::Partition& P : fin) {
                    hso::DistRes d = hso::distance(P, col, (char)col.ref_base);
                    if (hso::computeChiSquare(d) > 20.0 && d.n10 + d.n00 > 4 && d.n01 + d.n11 > 4) { kept = true; break; }
                }
        }
        keep[i] = kept ? 1 : 0;
    }
    return 0;
}

// V5 oracle: distance(Partition&, Partition&, threshold_p) (call_variants.cpp:977-1127) for pairs of partitions given as dense arrays
// (state 2 = read absent); out = 6 ints per pair {n00, n01, n10, n11, phased, augmented}
int hso_partition_pair_distance(const int8_t* state, const int32_t* more, const int32_t* less, const int64_t* part_off, const int32_t* part_n,
                                const int32_t* pair_a, const int32_t* pair_b, int32_t n_pairs, int32_t threshold_p, int32_t* out) {
    auto make = [&](int p) {
        hso::Partition P;
        for (int r = 0; r < part_n[p]; ++r) {
            const int8_t st = state[part_off[p] + r];
            if (st == 2) continue;
            P.readIdx.push_back(r); P.mostFrequentBases.push_back(st); P.moreFrequence.push_back(more[part_off[p] + r]); P.lessFrequence.push_back(less[part_off[p] + r]);
        }
        return P;
    };
    for (int k = 0; k < n_pairs; ++k) {
        const hso::Partition A = make(pair_a[k]), B = make(pair_b[k]);
        const hso::DistRes d = hso::distance(A, B, threshold_p);
        out[6 * k] = d.n00; out[6 * k + 1] = d.n01; out[6 * k + 2] = d.n10; out[6 * k + 3] = d.n11; out[6 * k + 4] = d.phased; out[6 * k + 5] = d.augmented ? 1 : 0;
    }
    return 0;
}

// exact (reference tie order) top-3 of every position: call_variants.cpp:477-507 on a read-major pileup
int hso_column_top3(const uint8_t* pile, const int64_t* pile_off, const int32_t* rec_pos, const int32_t* rec_qend,
                    int32_t r0, int32_t r1, int64_t L, uint8_t* k0, uint8_t* k1, int32_t* c0, int32_t* c1, int32_t* c2,
                    int32_t* depth) {
    std::vector<hso::Column> cols((size_t)L);
    for (int r = r0; r < r1; ++r)
        for (int q = rec_pos[r]; q < rec_qend[r]; ++q) { cols[(size_t)q].readIdxs.push_back((unsigned)(r - r0)); cols[(size_t)q].content.push_back(pile[pile_off[r] + (q - rec_pos[r])]); }
    std::string ref((size_t)L, 'A');
    hso::CallResult cr = hso::call_variants(cols, ref, 0.05f, 0.33f);
    for (int64_t p = 0; p < L; ++p) { k0[p] = cr.k0[(size_t)p]; k1[p] = cr.k1[(size_t)p]; c0[p] = cr.c0[(size_t)p]; c1[p] = cr.c1[(size_t)p]; c2[p] = cr.c2[(size_t)p]; depth[p] = (int32_t)cols[(size_t)p].content.size(); }
    return 0;
}

// the candidate SNPs of a contig (call_variants.cpp:447-567: the predicate of :525-529 with the greedy spacing, the automatic ones of :531)
// on a read-major pileup: flags[p] bit 0 = suspicious, bit 1 = automatic. Returns the number of candidates.
int hso_call_variants_flags(const uint8_t* pile, const int64_t* pile_off, const int32_t* rec_pos, const int32_t* rec_qend,
                            int32_t r0, int32_t r1, int64_t L, float mean_error, float automatic_snp_threshold, uint8_t* flags) {
    std::vector<hso::Column> cols((size_t)L);
    for (int r = r0; r < r1; ++r)
        for (int q = rec_pos[r]; q < rec_qend[r]; ++q) { cols[(size_t)q].readIdxs.push_back((unsigned)(r - r0)); cols[(size_t)q].content.push_back(pile[pile_off[r] + (q - rec_pos[r])]); }
    std::string ref((size_t)L, 'A');
    hso::CallResult cr = hso::call_variants(cols, ref, mean_error, automatic_snp_threshold);
    for (int64_t p = 0; p < L; ++p) flags[p] = 0;
    for (const hso::Column& c : cr.suspicious) flags[c.pos] |= 1;
    for (const hso::Column& c : cr.automatic) flags[c.pos] |= 2;
    return (int)cr.suspicious.size();
}

// K5 oracle: separate_reads.cpp:374-433 from CSR SNP columns
int hso_simdiff(int32_t n_reads, int32_t n_snps, const uint8_t* snp_ref, const uint8_t* snp_alt, const int64_t* col_off,
                const int32_t* col_idx, const uint8_t* col_code, int32_t* sim, int32_t* diff) {
    std::vector<hso::Column> snps((size_t)n_snps);
    for (int s = 0; s < n_snps; ++s) {
        snps[(size_t)s].ref_base = snp_ref[s]; snps[(size_t)s].second_base = snp_alt[s];
        for (int64_t e = col_off[s]; e < col_off[s + 1]; ++e) { snps[(size_t)s].readIdxs.push_back((unsigned)col_idx[e]); snps[(size_t)s].content.push_back(col_code[e]); }
    }
    std::vector<int> S, D;
    hso::list_similarities_and_differences(snps, n_reads, S, D);
    std::memcpy(sim, S.data(), S.size() * sizeof(int)); std::memcpy(diff, D.data(), D.size() * sizeof(int));
    return 0;
}

// K6 oracle: create_read_graph_matrix (separate_reads.cpp:706-828) for one window. adj_off[N+1], adj (capacity N*N) =
// sorted neighbour lists of the symmetric graph.
int hso_read_graph(int32_t n, const int32_t* sim, const int32_t* diff, const uint8_t* mask, float error_rate, int32_t* adj_off, int32_t* adj) {
    std::vector<bool> m((size_t)n);
    for (int i = 0; i < n; ++i) m[(size_t)i] = mask[i] != 0;
    std::vector<int> S(sim, sim + (size_t)n * n), D(diff, diff + (size_t)n * n);
    std::vector<std::vector<int>> a;
    hso::create_read_graph_matrix(m, S, D, n, error_rate, a);
    adj_off[0] = 0;
    for (int i = 0; i < n; ++i) {
        for (size_t k = 0; k < a[(size_t)i].size(); ++k) adj[adj_off[i] + (int)k] = a[(size_t)i][k];
        adj_off[i + 1] = adj_off[i] + (int)a[(size_t)i].size();
    }
    return 0;
}

// K7 oracle: cluster_graph.cpp:240-310
int hso_chinese_whispers(int32_t n, const int32_t* adj_off, const int32_t* adj, const uint8_t* mask, const int32_t* init,
                         uint32_t seed, int32_t* out, int32_t* sweeps) {
    std::vector<std::vector<int>> a((size_t)n);
    for (int i = 0; i < n; ++i) a[(size_t)i].assign(adj + adj_off[i], adj + adj_off[i + 1]);
    std::vector<bool> m((size_t)n);
    for (int i = 0; i < n; ++i) m[(size_t)i] = mask[i] != 0;
    std::vector<int> in(init, init + n);
    int sw = 0;
    std::vector<int> r = hso::chinese_whispers(a, in, m, seed, &sw);
    std::memcpy(out, r.data(), (size_t)n * sizeof(int));
    if (sweeps) *sweeps = sw;
    return 0;
}

int hso_shuffled_order(int32_t n, uint32_t seed, int32_t* out) {
    std::vector<int> o = hso::shuffled_order(n, seed);
    std::memcpy(out, o.data(), (size_t)n * sizeof(int));
    return 0;
}

int hso_edit_distance(const uint8_t* q, int32_t qn, const uint8_t* t, int32_t tn, int32_t mode, int32_t* end_loc) {
    return hso::edit_distance(q, qn, t, tn, mode, end_loc);
}

// robin_hood iteration order of unsigned char keys (hs_oracle_rh.h), keys inserted left to right
int hso_rh_order_u8(const uint8_t* keys, int32_t n, uint8_t* out) {
    hso::RHMap<unsigned char, int> m;
    for (int i = 0; i < n; ++i) m[keys[i]] += 1;
    int k = 0;
    m.for_each([&](unsigned char key, int) { out[k++] = key; });
    return k;
}
int hso_rh_order_int(const int32_t* keys, int32_t n, int32_t* out) {
    hso::RHMap<int, int> m;
    for (int i = 0; i < n; ++i) m[keys[i]] += 1;
    int k = 0;
    m.for_each([&](int key, int) { out[k++] = key; });
    return k;
}

}  // extern "C"

// ---- reading the reference's OUTPUT files into flat arrays (checker side of bench.py's parity gate and of the tests that compare the
// in-memory pipeline with the reference at full size). The formats are the writers' of call_variants.cpp:1184-1211 (.col) and
// separate_reads.cpp:1756-1784 (.gro): CONTIG <name> <fields...> / READ ... / <tag> a b [c] idx,idx,..., val,val,...,
struct hso_blocks {
    int32_t n_contigs;
    int64_t names_len;      // '\n'-separated contig names, in file order
    char* names;
    int64_t extra_len;      // '\n'-separated rest of every CONTIG line (after the name: "L\tdepth")
    char* extra;
    int32_t* n_read_lines;  // [C] READ lines of the contig
    int64_t* rec_off;       // [C+1] records (SNPS / GROUP lines) of the contig
    int64_t n_records;
    int32_t *a, *b, *c;     // [R] the two or three numbers after the tag (.col: pos, ref, second; .gro: start, end, 0)
    int64_t* ent_off;       // [R+1]
    int32_t *idx, *val;     // the two comma lists (they must have the same length: otherwise the call fails)
};

static bool hso_read_file(const char* path, std::string& s) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    s.resize((size_t)n);
    size_t got = n ? std::fread(&s[0], 1, (size_t)n, f) : 0;
    std::fclose(f);
    return got == (size_t)n;
}

template <class T> static T* hso_dup(const std::vector<T>& v) {
    T* p = (T*)std::malloc(std::max<size_t>(1, v.size()) * sizeof(T));
    if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

extern "C" {

// One contig through separate_reads_on_contig (separate_reads.cpp:1508-1739) with the test taps on: the windows that build a read graph,
// their reads, the labels of their per-SNP Chinese-Whispers runs, of the run behind the small-cluster filter, and the finished labels of
// EVERY window of the contig. Arrays are malloc'ed (hso_sr_taps_free).
struct hso_sr_taps {
    int32_t n_tap; int32_t* tap_start; int64_t* tap_row0; int32_t* mask_ids; int64_t* run_begin; int32_t* run_snp; int32_t* run_labels /* runs x m, window after window */; int32_t* third;
    int32_t n_windows; int32_t* win_start; int32_t* win_end; int32_t* win_labels /* n_windows x N */;
};
void hso_sr_taps_free(hso_sr_taps* t) {
    if (!t) return;
    std::free(t->tap_start); std::free(t->tap_row0); std::free(t->mask_ids); std::free(t->run_begin); std::free(t->run_snp); std::free(t->run_labels); std::free(t->third);
    std::free(t->win_start); std::free(t->win_end); std::free(t->win_labels); std::free(t);
}
int hso_sr_contig_taps(int32_t n_reads, int64_t length, const int32_t* read_start, const int32_t* read_end, int32_t n_snps, const int32_t* snp_pos,
                       const uint8_t* snp_ref, const uint8_t* snp_alt, const int64_t* col_off, const int32_t* col_idx, const uint8_t* col_code,
                       int32_t window, float error_rate, int32_t low_memory, uint32_t seed, hso_sr_taps** out) {
    hso::ColContig c;
    c.length = (long)length;
    c.read_lines.assign((size_t)n_reads, std::string());
    for (int r = 0; r < n_reads; ++r) c.readLimits.push_back(std::make_pair((int)read_start[r], (int)read_end[r]));
    for (int s = 0; s < n_snps; ++s) {
        hso::Column col;
        col.pos = snp_pos[s]; col.ref_base = snp_ref[s]; col.second_base = snp_alt[s];
        for (int64_t e = col_off[s]; e < col_off[s + 1]; ++e) { col.readIdxs.push_back((unsigned)col_idx[e]); col.content.push_back(col_code[e]); }
        c.snps.push_back(col);
    }
    std::vector<hso::WindowTap> taps;
    hso::g_window_taps = &taps;
    std::vector<hso::Window> ws = hso::separate_reads_on_contig(c, window, error_rate, low_memory != 0, low_memory != 0, 0, seed);
    hso::g_window_taps = nullptr;
    hso_sr_taps* t = (hso_sr_taps*)std::calloc(1, sizeof(hso_sr_taps));
    std::vector<int32_t> start, mask, rsnp, rlab, third, wst, wen, wlab; std::vector<int64_t> row0(1, 0), rb(1, 0);
    for (const hso::WindowTap& w : taps) {
        start.push_back(w.start);
        mask.insert(mask.end(), w.mask_ids.begin(), w.mask_ids.end()); row0.push_back((int64_t)mask.size());
        for (size_t k = 0; k < w.runs.size(); ++k) { rsnp.push_back(w.run_snp[k]); rlab.insert(rlab.end(), w.runs[k].begin(), w.runs[k].end()); }
        rb.push_back((int64_t)rsnp.size());
        if (w.third.size() == w.mask_ids.size()) third.insert(third.end(), w.third.begin(), w.third.end());
        else third.insert(third.end(), w.mask_ids.size(), -9);      // (a window without runs never gets there)
    }
    for (const hso::Window& w : ws) { wst.push_back(w.start); wen.push_back(w.end); wlab.insert(wlab.end(), w.labels.begin(), w.labels.end()); }
    t->n_tap = (int32_t)taps.size(); t->tap_start = hso_dup(start); t->tap_row0 = hso_dup(row0); t->mask_ids = hso_dup(mask); t->run_begin = hso_dup(rb);
    t->run_snp = hso_dup(rsnp); t->run_labels = hso_dup(rlab); t->third = hso_dup(third);
    t->n_windows = (int32_t)ws.size(); t->win_start = hso_dup(wst); t->win_end = hso_dup(wen); t->win_labels = hso_dup(wlab);
    *out = t;
    return 0;
}

void hso_blocks_free(hso_blocks* b) {
    if (!b) return;
    std::free(b->names); std::free(b->extra); std::free(b->n_read_lines); std::free(b->rec_off); std::free(b->a); std::free(b->b); std::free(b->c);
    std::free(b->ent_off); std::free(b->idx); std::free(b->val);
    delete b;
}

// n_fixed = 3 for "SNPS" (.col), 2 for "GROUP" (.gro). Returns 0, or a negative code (-1 unreadable, -2 malformed line).
int hso_parse_blocks(const char* path, const char* tag, int32_t n_fixed, hso_blocks** out) {
    std::string s;
    if (!hso_read_file(path, s)) return -1;
    const size_t tl = std::strlen(tag);
    std::string names, extra;
    std::vector<int32_t> nread, a, b, c, idx, val;
    std::vector<int64_t> rec_off, ent_off;
    ent_off.push_back(0);
    const char* p = s.data();
    const char* end = p + s.size();
    auto number = [&](const char*& q, const char* e, int32_t& v) -> bool {
        bool neg = false;
        if (q < e && *q == '-') { neg = true; ++q; }
        if (q >= e || *q < '0' || *q > '9') return false;
        int64_t x = 0;
        while (q < e && *q >= '0' && *q <= '9') { x = x * 10 + (*q - '0'); ++q; }
        v = (int32_t)(neg ? -x : x);
        return true;
    };
    while (p < end) {
        const char* e = (const char*)std::memchr(p, '\n', (size_t)(end - p));
        if (!e) e = end;
        if (e - p >= 7 && std::memcmp(p, "CONTIG\t", 7) == 0) {
            const char* q = p + 7;
            const char* t = (const char*)std::memchr(q, '\t', (size_t)(e - q));
            if (!t) t = e;
            names.append(q, t); names.push_back('\n');
            if (t < e) extra.append(t + 1, e);
            extra.push_back('\n');
            nread.push_back(0);
            rec_off.push_back((int64_t)a.size());
        } else if (e - p >= 5 && std::memcmp(p, "READ\t", 5) == 0) {
            if (nread.empty()) return -2;
            nread.back() += 1;
        } else if ((size_t)(e - p) > tl && std::memcmp(p, tag, tl) == 0 && p[tl] == '\t') {
            if (nread.empty()) return -2;
            const char* q = p + tl + 1;
            int32_t v[3] = {0, 0, 0};
            for (int k = 0; k < n_fixed; ++k) {
                if (!number(q, e, v[k])) return -2;
                if (q < e && *q == '\t') ++q;
            }
            a.push_back(v[0]); b.push_back(v[1]); c.push_back(v[2]);
            size_t n0 = idx.size();
            while (q < e && *q != '\t') {
                int32_t x;
                if (!number(q, e, x)) return -2;
                idx.push_back(x);
                if (q < e && *q == ',') ++q;
            }
            if (q < e && *q == '\t') ++q;
            while (q < e && *q != '\t') {
                int32_t x;
                if (!number(q, e, x)) return -2;
                val.push_back(x);
                if (q < e && *q == ',') ++q;
            }
            if (val.size() != idx.size()) return -2;
            (void)n0;
            ent_off.push_back((int64_t)idx.size());
        }
        p = e + 1;
    }
    rec_off.push_back((int64_t)a.size());
    hso_blocks* B = new hso_blocks();
    B->n_contigs = (int32_t)nread.size();
    B->names_len = (int64_t)names.size(); B->names = (char*)std::malloc(names.size() + 1); std::memcpy(B->names, names.c_str(), names.size() + 1);
    B->extra_len = (int64_t)extra.size(); B->extra = (char*)std::malloc(extra.size() + 1); std::memcpy(B->extra, extra.c_str(), extra.size() + 1);
    B->n_read_lines = hso_dup(nread); B->rec_off = hso_dup(rec_off); B->n_records = (int64_t)a.size();
    B->a = hso_dup(a); B->b = hso_dup(b); B->c = hso_dup(c); B->ent_off = hso_dup(ent_off); B->idx = hso_dup(idx); B->val = hso_dup(val);
    *out = B;
    return 0;
}

}  // extern "C"
