// TEST INFRASTRUCTURE (never shipped): runs the product's *host glue* (hs_driver.cpp, hs_host_cv.cpp,
// hs_host_sr.cpp, hs_io.cpp) with the device interface implemented by the CPU oracle, so that the sequential
// logic around the HIP kernels can be checked against the reference goldens on a machine without a GPU.
// The product library itself has no such backend (hs_capi.hip only knows HIP).
//   host_harness call_variants  <11 positional args of HS_call_variants>
//   host_harness separate_reads <9 positional args of HS_separate_reads>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <cmath>
#include <map>
#include <unordered_map>

#include "../../hairsplitter_amd/csrc/hs_driver.h"
#include "../../oracle/hs_oracle.h"

namespace hs {
static std::string g_err;
void set_error(const std::string& m) { g_err = m; }
}
extern "C" const char* hs_last_error(void) { return hs::g_err.c_str(); }

namespace {

struct OracleCvOps : hs::CvDeviceOps {
    const hs::CvFileInput& in;
    std::vector<hso::MsaResult> msa;              // per contig
    explicit OracleCvOps(const hs::CvFileInput& i) : in(i) {}

    // K0 + K1 through the oracle's generate_msa
    int pileup(std::vector<int32_t>& rec_stats, float k_ms[4]) override {
        k_ms[0] = k_ms[1] = k_ms[3] = 0;
        const int C = (int)in.contig_names.size();
        std::vector<std::string> read_seq(in.read_names.size());
        for (size_t r = 0; r < read_seq.size(); ++r) {
            std::string s;
            for (int64_t k = in.read_off[r]; k < in.read_off[r + 1]; ++k) s += "ACGT"[in.read_seq[(size_t)k]];
            read_seq[r] = s;
        }
        msa.clear(); msa.resize((size_t)C);
        const char* opc = "MIDNSHP=X";
        for (int c = 0; c < C; ++c) {
            hso::Contig ctg;
            ctg.name = in.contig_names[(size_t)c];
            for (int64_t k = in.contig_off[(size_t)c]; k < in.contig_off[(size_t)c + 1]; ++k) ctg.seq += "ACGT"[in.contig_seq[(size_t)k]];
            for (int r = in.contig_rec_off[(size_t)c]; r < in.contig_rec_off[(size_t)c + 1]; ++r) {
                hso::Record rec;
                rec.read = in.rec_read[(size_t)r]; rec.position_2_1 = in.rec_pos[(size_t)r]; rec.strand = in.rec_strand[(size_t)r] != 0;
                for (int64_t o = in.rec_cig_off[(size_t)r]; o < in.rec_cig_off[(size_t)r + 1]; ++o)
                    rec.cigar += std::to_string(in.cigar[(size_t)o] >> 4) + opc[in.cigar[(size_t)o] & 15u];
                ctg.recs.push_back(rec);
            }
            hso::MsaResult m = hso::generate_msa(ctg, read_seq);
            for (size_t k = 0; k < ctg.recs.size(); ++k) {
                const size_t r = (size_t)in.contig_rec_off[(size_t)c] + k;
                rec_stats[r * 4 + 0] = m.q_end[k]; rec_stats[r * 4 + 1] = (int32_t)m.n_err[k]; rec_stats[r * 4 + 2] = (int32_t)m.n_len[k];
            }
            msa[(size_t)c] = std::move(m);
        }
        return 0;
    }

    // the columns of the current range: every position whose second count is >= 4 (what the device extracts), in position order
    struct XCol { int contig, pos; hs_colrec rec; };
    int r0 = 0, r1 = 0;
    std::vector<XCol> xcols;
    std::vector<hs_colrec> pk_rec; std::vector<int32_t> pk_col; std::vector<int64_t> pk_off; std::vector<int32_t> pk_idx; std::vector<uint8_t> pk_code;
    void pack(int flag) {
        pk_rec.clear(); pk_col.clear(); pk_off.assign(1, 0); pk_idx.clear(); pk_code.clear();
        for (size_t k = 0; k < xcols.size(); ++k) {
            if (!(xcols[k].rec.flags & flag)) continue;
            const hso::Column& col = msa[(size_t)xcols[k].contig].cols[(size_t)xcols[k].pos];
            pk_rec.push_back(xcols[k].rec); pk_col.push_back((int32_t)k);
            for (size_t e = 0; e < col.content.size(); ++e) { pk_idx.push_back((int32_t)col.readIdxs[e]); pk_code.push_back(col.content[e]); }
            pk_off.push_back((int64_t)pk_idx.size());
        }
        pk_idx.push_back(0); pk_code.push_back(0);      // (never null)
    }
    // K2 + K3 + K3b + V1 through the oracle's call_variants (call_variants.cpp:447-567): its per-position top-3 (the reference's
    // order of equal counts) and its candidate / automatic lists
    int extract_candidates(int c0, int c1, const std::vector<int32_t>& min_reads, float thr, hs::CvCandidates& out, float k_ms[3], bool) override {
        k_ms[0] = k_ms[1] = k_ms[2] = 0;
        r0 = c0; r1 = c1;
        xcols.clear();
        out = hs::CvCandidates();
        out.contig_n_cand.assign((size_t)(c1 - c0), 0);
        for (int c = c0; c < c1; ++c) {
            hso::MsaResult& m = msa[(size_t)c];
            if (((m.meanDistance < 0.015f) ? 3 : 5) != min_reads[(size_t)(c - c0)]) { std::cerr << "harness: the driver's read minimum differs from the oracle's\n"; return 3; }
            hso::CallResult cr = hso::call_variants(m.cols, m.newref, m.meanDistance, thr);
            std::vector<char> is_cand(m.cols.size(), 0), is_auto(m.cols.size(), 0);
            for (const hso::Column& s : cr.suspicious) is_cand[(size_t)s.pos] = 1;
            for (const hso::Column& s : cr.automatic) is_auto[(size_t)s.pos] = 1;
            for (size_t p = 0; p < m.cols.size(); ++p) {
                const int c1v = cr.c1[p], c2v = cr.c2[p];
                if (!(c1v > 4 || (c1v == 4 && c2v == 0))) { if (is_cand[p]) { std::cerr << "harness: a candidate outside the extracted columns\n"; return 3; } continue; }
                XCol x; x.contig = c; x.pos = (int)p;
                hs_colrec& r = x.rec;
                r.pos = (int32_t)p; r.contig = c; r.c0 = (uint16_t)cr.c0[p]; r.c1 = (uint16_t)c1v; r.k0 = cr.k0[p]; r.k1 = cr.k1[p]; r.c2 = (uint8_t)(c2v < 63 ? c2v : 63);
                r.flags = 0;
                if (c1v > 5 * c2v) r.flags |= HS_COL_C1GT5C2;
                if (is_cand[p]) r.flags |= HS_COL_CAND;
                if (is_auto[p]) r.flags |= HS_COL_AUTO;
                const int rb = r.k0, sb = r.k1;
                if (c1v >= 5 && rb % 5 != sb % 5 && ((sb - '!') % 5 != 4 || (sb / 5 % 5 != rb % 5 && sb / 25 % 5 != rb % 5))) r.flags |= HS_COL_LOOPD;
                out.n_entries += (int64_t)m.cols[p].content.size();
                if (is_cand[p]) out.contig_n_cand[(size_t)(c - c0)]++;
                xcols.push_back(x);
            }
        }
        out.n_columns = (int64_t)xcols.size();
        pack(HS_COL_CAND);
        out.n_cand = (int64_t)pk_rec.size();
        out.rec = pk_rec.data(); out.col = pk_col.data(); out.off = pk_off.data(); out.idx = pk_idx.data(); out.code = pk_code.data();
        // the candidates as bit sets over the reads ranked by start position (what the device's k_cand_bits hands to loop A),
        // contig by contig with the product's own restatement of that kernel; word offsets count from the start of cb_words
        cb_bits.assign(pk_rec.size(), hs::CandBits()); cb_words.clear();
        {
            size_t k = 0;
            for (int c = c0; c < c1; ++c) {
                const int n = out.contig_n_cand[(size_t)(c - c0)];
                const int rr0 = in.contig_rec_off[(size_t)c], nr = in.contig_rec_off[(size_t)c + 1] - rr0;
                std::vector<int32_t> rank_of, orig_of, read_end((size_t)nr);
                hs::cv_rank_reads(nr, in.rec_pos.data() + rr0, rank_of, orig_of);
                for (int r = 0; r < nr; ++r) {
                    int64_t span = 0;
                    for (int64_t o = in.rec_cig_off[(size_t)(rr0 + r)]; o < in.rec_cig_off[(size_t)(rr0 + r) + 1]; ++o) {
                        const uint32_t op = in.cigar[(size_t)o] & 15u;
                        if (op == 0 || op == 2 || op == 7 || op == 8) span += in.cigar[(size_t)o] >> 4;
                    }
                    read_end[(size_t)r] = (int32_t)std::min<int64_t>((int64_t)in.rec_pos[(size_t)(rr0 + r)] + span, 0x7fffffff);
                }
                hs::cv_build_cand_bits(n, pk_off.data() + k, pk_idx.data(), pk_code.data(), rank_of.data(), read_end.data(), cb_bits.data() + k, cb_words);
                k += (size_t)n;
            }
        }
        cb_words.push_back(0);
        out.bits = cb_bits.data(); out.words = cb_words.data();
        return 0;
    }
    std::vector<hs::CandBits> cb_bits; std::vector<uint64_t> cb_words;
    // loops C and D of keep_only_robust_variants through the oracle's distance() / computeChiSquare(), then the reference's own
    // two-pointer merge of the automatic and the filtered SNPs (call_variants.cpp:1335-1352)
    int finish_columns(const hs::CvPartitionTest& t, bool want_entries, hs::CvSnpSet& out, float* k_ms) override {
        if (k_ms) *k_ms = 0;
        const int C = r1 - r0;
        out = hs::CvSnpSet();
        out.contig_n_snp.assign((size_t)C, 0);
        std::vector<std::vector<hso::Partition>> parts((size_t)C);
        for (int c = 0; c < C; ++c) {
            const int nreads = in.contig_rec_off[(size_t)(r0 + c) + 1] - in.contig_rec_off[(size_t)(r0 + c)];
            if (nreads != t.contig_n_reads[(size_t)c]) { std::cerr << "harness: read counts of the partitions differ\n"; return 3; }
            for (int f = t.part_off[(size_t)c]; f < t.part_off[(size_t)c + 1]; ++f) {
                hso::Partition P;
                for (int r = 0; r < nreads; ++r) {
                    const int8_t st = t.part_state[(size_t)t.part_state_off[(size_t)f] + (size_t)r];
                    if (st == 2) continue;
                    P.readIdx.push_back(r); P.mostFrequentBases.push_back(st); P.moreFrequence.push_back(0); P.lessFrequence.push_back(0);
                }
                parts[(size_t)c].push_back(P);
            }
        }
        std::vector<std::vector<size_t>> autos((size_t)C), filt((size_t)C);
        for (size_t i = 0; i < xcols.size(); ++i) {
            hs_colrec& r = xcols[i].rec;
            const int c = xcols[i].contig - r0;
            hso::Column col = msa[(size_t)xcols[i].contig].cols[(size_t)xcols[i].pos];
            col.ref_base = r.k0; col.second_base = r.k1;
            const std::vector<hso::Partition>& fin = parts[(size_t)c];
            bool kept = false;
            if (r.flags & HS_COL_CAND) {
                for (const hso::Partition& P : fin) {
                    hso::DistRes d = hso::distance(P, col, (char)col.ref_base);
                    if (d.n00 + d.n01 + d.n10 + d.n11 > 0.5 * col.content.size() && hso::computeChiSquare(d) > 15) { kept = true; break; }
                }
            }
            if (!kept && (r.flags & HS_COL_LOOPD)) {
                for (const hso::Partition& P : fin) {
                    hso::DistRes d = hso::distance(P, col, (char)col.ref_base);
                    if (hso::computeChiSquare(d) > 20.0 && d.n10 + d.n00 > 4 && d.n01 + d.n11 > 4) { kept = true; break; }
                }
            }
            r.flags &= ~(HS_COL_KEEP | HS_COL_SNP);
            if (kept) { r.flags |= HS_COL_KEEP; filt[(size_t)c].push_back(i); }
            if (r.flags & HS_COL_AUTO) autos[(size_t)c].push_back(i);
        }
        for (int c = 0; c < C; ++c) {
            size_t ia = 0, ifi = 0;
            const std::vector<size_t>& A = autos[(size_t)c]; const std::vector<size_t>& F = filt[(size_t)c];
            while (ia < A.size() && ifi < F.size()) {
                const int pa = xcols[A[ia]].pos, pf = xcols[F[ifi]].pos;
                if (pa < pf) { xcols[A[ia]].rec.flags |= HS_COL_SNP; ia++; }
                else if (pa > pf) { xcols[F[ifi]].rec.flags |= HS_COL_SNP; ifi++; }
                else { xcols[A[ia]].rec.flags |= HS_COL_SNP; ia++; ifi++; }
                out.contig_n_snp[(size_t)c]++;
            }
        }
        pack(HS_COL_SNP);
        out.n_snp = (int64_t)pk_rec.size(); out.n_entries = pk_off.back();
        out.rec = pk_rec.data(); out.off = pk_off.data();
        if (want_entries) { out.idx = pk_idx.data(); out.code = pk_code.data(); }
        return 0;
    }
};

struct OracleSrOps : hs::SrDeviceOps {
    hs::SrWindowSet ws;
    std::vector<int64_t> g_off;      // CSR over the rows of the window set, neighbours = local ids
    std::vector<int32_t> g_nbr;
    std::vector<int32_t> sim, diff;
    uint32_t seed;
    explicit OracleSrOps(uint32_t s) : seed(s) {}
    // K5a + K5 through the oracle's list_similarities_and_differences (separate_reads.cpp:374-433) on the job's SNP columns
    int simdiff_columns(const hs::SimdiffJob& job, float* k_ms) override {
        (void)k_ms;
        sd_off = job.out_off; sd_n = job.n_reads;
        sim.assign((size_t)job.out_total, 0); diff.assign((size_t)job.out_total, 0);
        const hs::CwChain& ch = *job.cols;
        const int C = (int)job.n_reads.size();
        for (int c = 0; c < C; ++c) {
            const int N = job.n_reads[(size_t)c];
            if (N == 0) continue;
            const int64_t s0 = job.contig_snp_base[(size_t)c];
            const int64_t s1 = c + 1 < C ? job.contig_snp_base[(size_t)c + 1] : (int64_t)job.snp_ref.size();
            std::vector<hso::Column> snps;
            for (int64_t s = s0; s < s1; ++s) {
                hso::Column col;
                col.ref_base = job.snp_ref[(size_t)s]; col.second_base = job.snp_alt[(size_t)s];
                for (int64_t e = ch.col_off[(size_t)s]; e < ch.col_off[(size_t)s + 1]; ++e) { col.readIdxs.push_back((unsigned)ch.col_idx[(size_t)e]); col.content.push_back(ch.col_code[(size_t)e]); }
                snps.push_back(std::move(col));
            }
            std::vector<int> S, D;
            hso::list_similarities_and_differences(snps, N, S, D);
            std::copy(S.begin(), S.end(), sim.begin() + job.out_off[(size_t)c]);
            std::copy(D.begin(), D.end(), diff.begin() + job.out_off[(size_t)c]);
        }
        return 0;
    }
    std::vector<int64_t> sd_off; std::vector<int32_t> sd_n;
    int n_reads_of_contig(int c) const { return (int)((c + 1 < (int)ws.ctg_rank_off.size() ? ws.ctg_rank_off[(size_t)c + 1] : (int64_t)ws.rank.size()) - ws.ctg_rank_off[(size_t)c]); }
    // K6 through the oracle's create_read_graph_matrix (N-space), stored in the windows' local index space
    int build_graphs(const hs::SrWindowSet& w_in, int64_t* rows_on_host, float* k_ms) override {
        (void)k_ms;
        ws = w_in;
        if (rows_on_host) *rows_on_host = 0;
        g_off.assign(1, 0); g_nbr.clear();
        for (int w = 0; w < ws.n_dev_windows; ++w) {
            const int c = ws.win_contig[(size_t)w];
            const int N = sd_n[(size_t)c];
            const int64_t m0 = ws.win_row0[(size_t)w], m1 = ws.win_row0[(size_t)w + 1];
            std::vector<bool> mask((size_t)N, false);
            std::vector<int> local((size_t)N, -1);
            for (int64_t k = m0; k < m1; ++k) { mask[(size_t)ws.mask_ids[(size_t)k]] = true; local[(size_t)ws.mask_ids[(size_t)k]] = (int)(k - m0); }
            std::vector<int> S(sim.begin() + sd_off[(size_t)c], sim.begin() + sd_off[(size_t)c] + (size_t)N * N);
            std::vector<int> D(diff.begin() + sd_off[(size_t)c], diff.begin() + sd_off[(size_t)c] + (size_t)N * N);
            std::vector<std::vector<int>> adj;
            hso::create_read_graph_matrix(mask, S, D, N, ws.error_rate, adj);
            for (int r = 0; r < N; ++r) if (!mask[(size_t)r] && !adj[(size_t)r].empty()) { std::cerr << "harness: a read outside the mask has neighbours\n"; return 3; }
            for (int64_t k = m0; k < m1; ++k) {
                for (int nb : adj[(size_t)ws.mask_ids[(size_t)k]]) {
                    if (local[(size_t)nb] < 0) { std::cerr << "harness: a neighbour outside the mask\n"; return 3; }
                    g_nbr.push_back(local[(size_t)nb]);
                }
                g_off.push_back((int64_t)g_nbr.size());
            }
        }
        const int64_t base = (int64_t)g_nbr.size();
        for (size_t r = 1; r < ws.host_off.size(); ++r) g_off.push_back(base + ws.host_off[r]);
        g_nbr.insert(g_nbr.end(), ws.host_nbr.begin(), ws.host_nbr.end());
        if ((int64_t)g_off.size() != ws.rows() + 1) { std::cerr << "harness: rows of the window set do not add up\n"; return 3; }
        return 0;
    }
    int fetch_graphs(std::vector<int64_t>& off, std::vector<int32_t>& nbr) override { off = g_off; nbr = g_nbr; return 0; }

    // one run of the oracle's Chinese Whispers on window w: the local graph expanded to the contig's N reads, so that the
    // visiting order is the oracle's own shuffle of 0..N-1 (the product's kernels work on the local nodes)
    std::vector<int> run_cw(int w, const std::vector<int>& init_n, bool empty_graph) {
        const int N = n_reads_of_contig(ws.win_contig[(size_t)w]);
        const int64_t m0 = ws.win_row0[(size_t)w], m1 = ws.win_row0[(size_t)w + 1];
        std::vector<std::vector<int>> adj((size_t)N);
        std::vector<bool> mask((size_t)N, false);
        for (int64_t k = m0; k < m1; ++k) {
            const int r = ws.mask_ids[(size_t)k];
            mask[(size_t)r] = true;
            if (!empty_graph) for (int64_t e = g_off[(size_t)k]; e < g_off[(size_t)k + 1]; ++e) adj[(size_t)r].push_back(ws.mask_ids[(size_t)(m0 + g_nbr[(size_t)e])]);
        }
        return hso::chinese_whispers(adj, init_n, mask, seed);
    }

    // CPU statement of the device-resident chain (separate_reads.cpp:1674-1705, :840-885, :924-971)
    int cw_chain(const hs::CwChain& ch, std::vector<int32_t>& labels, std::vector<int32_t>& final_labels, std::vector<uint8_t>& final_ok,
                 float k_ms[3], hs::SrChainStats* stats) override {
        final_labels.clear(); final_ok.clear();   // the tail of finalize_clustering stays with the product's host code here
        (void)k_ms; (void)stats;
        const int Wc = (int)ch.win.size();
        labels.assign((size_t)ch.chain_row0.back(), 0);
        for (int k = 0; k < Wc; ++k) {
            const int w = ch.win[(size_t)k];
            const int N = n_reads_of_contig(ws.win_contig[(size_t)w]);
            const int64_t m0 = ws.win_row0[(size_t)w], m1 = ws.win_row0[(size_t)w + 1];
            std::vector<bool> mask((size_t)N, false);
            for (int64_t q = m0; q < m1; ++q) mask[(size_t)ws.mask_ids[(size_t)q]] = true;
            const bool fe = ws.win_final_empty[(size_t)w] != 0;
            std::vector<std::vector<int>> local;
            for (int64_t i = ch.win_seed_begin[(size_t)k]; i < ch.win_seed_begin[(size_t)k + 1]; ++i) {
                const int64_t s = ch.seed_col[(size_t)i];
                std::vector<int> start((size_t)N);
                for (int r = 0; r < N; ++r) start[(size_t)r] = r;
                std::map<unsigned char, int> first;
                for (int64_t e = ch.col_off[(size_t)s]; e < ch.col_off[(size_t)s + 1]; ++e) {
                    const int r = ch.col_idx[(size_t)e];
                    if (!mask[(size_t)r]) continue;
                    if (!first.count(ch.col_code[(size_t)e])) first[ch.col_code[(size_t)e]] = r;
                    start[(size_t)r] = first[ch.col_code[(size_t)e]];
                }
                local.push_back(run_cw(w, start, false));
            }
            // merge_clusterings :840-874
            std::vector<double> agg((size_t)N, 0.0);
            for (size_t i = 0; i < local.size(); ++i) for (int j = 0; j < N; ++j) agg[(size_t)j] += local[i][(size_t)j] * std::pow(2.0, (double)i);
            std::unordered_map<double, int> ids; std::vector<int> merged((size_t)N); int index = 0;
            for (int j = 0; j < N; ++j) { auto it = ids.find(agg[(size_t)j]); if (it == ids.end()) { ids[agg[(size_t)j]] = index; merged[(size_t)j] = index++; } else merged[(size_t)j] = it->second; }
            for (int j = 0; j < N; ++j) if (!mask[(size_t)j]) merged[(size_t)j] = -2;
            std::vector<int> c2 = run_cw(w, merged, fe);
            // finalize_clustering :924-955
            std::map<int, int> sizes;
            for (int r = 0; r < N; ++r) { if (!mask[(size_t)r]) c2[(size_t)r] = -2; else sizes[c2[(size_t)r]] += 1; }
            for (int r = 0; r < N; ++r) if (c2[(size_t)r] != -2 && sizes[c2[(size_t)r]] < 5) c2[(size_t)r] = -1;
            std::map<int, int> to_hap; int hap = 0;
            for (int r = 0; r < N; ++r) if (c2[(size_t)r] > -1) { if (!to_hap.count(c2[(size_t)r])) to_hap[c2[(size_t)r]] = hap++; c2[(size_t)r] = to_hap[c2[(size_t)r]]; }
            std::vector<int> c3 = run_cw(w, c2, fe);
            for (int64_t q = m0; q < m1; ++q) labels[(size_t)(ch.chain_row0[(size_t)k] + (q - m0))] = c3[(size_t)ws.mask_ids[(size_t)q]];
        }
        return 0;
    }
    int cw(hs::CwWave& wv, float* k_ms) override {
        (void)k_ms;
        for (size_t i = 0; i < wv.inst_win.size(); ++i) {
            const int w = wv.inst_win[i];
            const int N = n_reads_of_contig(ws.win_contig[(size_t)w]);
            const int64_t m0 = ws.win_row0[(size_t)w], m1 = ws.win_row0[(size_t)w + 1];
            std::vector<int> init((size_t)N, -2);
            for (int64_t q = m0; q < m1; ++q) init[(size_t)ws.mask_ids[(size_t)q]] = wv.labels[(size_t)(wv.inst_label_off[i] + (q - m0))];
            std::vector<int> res = run_cw(w, init, ws.win_final_empty[(size_t)w] != 0);
            for (int64_t q = m0; q < m1; ++q) wv.labels[(size_t)(wv.inst_label_off[i] + (q - m0))] = res[(size_t)ws.mask_ids[(size_t)q]];
        }
        return 0;
    }
};

}  // namespace

// `selftest`: host-side pieces of the drivers that need no device and no input files
static int selftest() {
    int bad = 0;
    auto expect = [&](bool ok, const char* what) { if (!ok) { std::fprintf(stderr, "selftest FAILED: %s\n", what); bad++; } };
    {   // sr_expand_labels: -2 where a window does not hold the read
        hs::SrSparseLabels sp;
        sp.off = {0, 2, 2, 5}; sp.ids = {1, 3, 0, 2, 4}; sp.labels = {7, -1, 0, 1, 0};
        const int64_t label_off[4] = {0, 5, 10, 15};
        std::vector<int32_t> dense(15, 99);
        hs::sr_expand_labels(sp, label_off, 0, 3, dense.data());
        const int32_t want[15] = {-2, 7, -2, -1, -2,  -2, -2, -2, -2, -2,  0, -2, 1, -2, 0};
        expect(std::equal(dense.begin(), dense.end(), want), "sr_expand_labels");
        std::vector<int32_t> part(15, 99);     // windows [1, 3) alone: the first five stay as they are
        hs::sr_expand_labels(sp, label_off, 1, 3, part.data());
        const int32_t want2[15] = {99, 99, 99, 99, 99,  -2, -2, -2, -2, -2,  0, -2, 1, -2, 0};
        expect(std::equal(part.begin(), part.end(), want2), "sr_expand_labels on a range of windows");
    }
    {   // the dense label block is recycled between calls
        int32_t* a = hs::sr_labels_alloc(5u << 20);
        a[0] = 1; a[(5u << 20) - 1] = 2;
        hs::sr_labels_free(a);
        int32_t* b = hs::sr_labels_alloc((5u << 20) - 1000);
        expect(a == b, "a freed label block of 20 MB is handed out again");
        hs::sr_labels_free(b);
        int32_t* c = hs::sr_labels_alloc(16);
        hs::sr_labels_free(c);
        int32_t* plain = (int32_t*)std::malloc(64);
        std::memset(plain, 0, 64);
        hs::sr_labels_free(nullptr);      // (accepted)
        std::free(plain);
    }
    std::printf("selftest %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    if (!std::strcmp(argv[1], "selftest")) return selftest();
    if (!std::strcmp(argv[1], "call_variants")) {
        if (argc < 13) return 2;
        char** a = argv + 1;
        hs::CvFileInput in;
        if (int rc = hs::load_cv_inputs(a[1], a[2], a[3], std::atoi(a[7]) != 0, in, 4)) { std::cerr << hs::g_err << "\n"; return rc; }
        hs::CvMeta meta;
        meta.n_contigs = (int)in.contig_names.size(); meta.n_rec = (int)in.rec_read.size();
        meta.contig_off = in.contig_off; meta.contig_rec_off = in.contig_rec_off; meta.total_len = in.contig_off.back();
        meta.pile_off.assign((size_t)meta.n_rec + 1, 0);
        meta.rec_pos = in.rec_pos; meta.rec_refspan.assign((size_t)meta.n_rec, 0);
        for (int c = 0; c < meta.n_contigs; ++c) {
            const int64_t L = in.contig_off[(size_t)c + 1] - in.contig_off[(size_t)c];
            for (int r = in.contig_rec_off[(size_t)c]; r < in.contig_rec_off[(size_t)c + 1]; ++r) {
                int64_t span = 0;
                for (int64_t o = in.rec_cig_off[(size_t)r]; o < in.rec_cig_off[(size_t)r + 1]; ++o) { uint32_t k = in.cigar[(size_t)o] & 15u; if (k == 0 || k == 2 || k == 7 || k == 8) span += in.cigar[(size_t)o] >> 4; }
                const int64_t pos = in.rec_pos[(size_t)r];
                meta.rec_refspan[(size_t)r] = span;
                const int64_t qend = pos >= L ? pos : std::min(pos + span, L);
                meta.pile_off[(size_t)r + 1] = meta.pile_off[(size_t)r] + (qend - pos);
            }
        }
        OracleCvOps ops(in);
        hs_cv_result* res = nullptr;
        if (std::getenv("HS_HARNESS_RANGE_SELECT")) {   // the way the contig groups go: the pileup of the batch, then stage 3 range by range, results put together
            hs::CvSelection whole;
            if (int rc = hs::cv_pileup(ops, meta, whole)) return rc;
            const int mid = meta.n_contigs / 2;
            hs_cv_result* ra = nullptr; hs_cv_result* rb = nullptr;
            if (int rc = hs::cv_run_range(ops, meta, whole.rec_stats, 0, mid, std::strtof(a[11], nullptr), 1, &ra)) return rc;
            if (int rc = hs::cv_run_range(ops, meta, whole.rec_stats, mid, meta.n_contigs, std::strtof(a[11], nullptr), 1, &rb)) return rc;
            res = hs::cv_concat_results(ra, rb);
        } else
        if (int rc = hs::cv_run(ops, meta, std::strtof(a[11], nullptr), 1, &res)) return rc;
        hs::write_cv_outputs(in, res, a[6], a[9], a[10], 4);
        if (const char* e = std::getenv("HS_HARNESS_DUMP_MD")) {      // the contigs' mean distances, exactly (what the ranks of a sharded job exchange)
            std::ofstream o(e);
            for (int c = 0; c < res->n_contigs; ++c) { char buf[64]; std::snprintf(buf, sizeof buf, "%a\n", (double)res->mean_distance[c]); o << buf; }
        }
        hs::free_cv_result(res);
        return 0;
    }
    if (!std::strcmp(argv[1], "separate_reads")) {
        if (argc != 11) return 2;
        char** a = argv + 1;
        std::vector<hs::ColFileContig> cs;
        // as the product's HS_separate_reads: the binary companion of the .col if it still describes this file, else the text
        if (hs::read_col_sidecar(a[1], (float)std::atof(a[6]), cs, 4) != 1) { if (int rc = hs::parse_col(a[1], (float)std::atof(a[6]), cs, 4)) return rc; }
        std::map<std::string, int> ploidy_of; bool have = false;
        { std::ifstream pf(a[4]); if (pf) { have = true; std::string c; int p; while (pf >> c >> p) ploidy_of[c] = p; } }
        std::vector<hs_sr_contig> hc(cs.size());
        for (size_t i = 0; i < cs.size(); ++i) {
            hs::ColFileContig& c = cs[i]; hs_sr_contig& h = hc[i];
            h.length = c.length; h.n_reads = (int32_t)c.read_lines.size(); h.read_start = c.read_start.data(); h.read_end = c.read_end.data();
            h.n_snps = (int32_t)c.snp_pos.size(); h.snp_pos = c.snp_pos.data(); h.snp_ref = c.snp_ref.data(); h.snp_alt = c.snp_alt.data();
            h.col_off = c.col_off.data(); h.col_idx = c.col_idx.data(); h.col_code = c.col_code.data();
            h.ploidy = (have && ploidy_of.count(c.name)) ? ploidy_of[c.name] : 0;
        }
        int w = hs::sr_window_size(hc.data(), (int)hc.size(), std::atoi(a[7]) != 0);
        if (const char* e = std::getenv("HS_HARNESS_WINDOW_SIZE")) w = std::atoi(e);      // (a shard of a job: the window size is chosen over the whole job)
        uint32_t seed = 12345u;   // as hs_main.cpp: the pinned std::random_device of the reference, HS_SEED overrides
        if (const char* e = std::getenv("HS_SEED")) seed = (uint32_t)std::strtoul(e, nullptr, 10);
        OracleSrOps ops(seed);
        hs_sr_result* res = nullptr;
        if (int rc = hs::sr_run(ops, hc.data(), (int)hc.size(), w, (float)std::atof(a[3]), std::atoi(a[5]), seed, 1, &res)) return rc;
        { std::ofstream o(a[8]); }
        hs::write_gro(cs, res, a[8], 4);
        hs::free_sr_result(res);
        return 0;
    }
    return 2;
}
