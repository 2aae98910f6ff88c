// hs_realign.cpp -- SURVEY.md 8(f) N3, third part: a CIGAR-less input for stage 3. The reference takes the base-level alignment of
// every read from the CIGAR of a SAM file and refuses anything else (call_variants.cpp:1256-1267; CIGAR required at
// input_output.cpp:357-368). Here a PAF file -- which read interval lies on which contig interval, on which strand, no CIGAR --
// is turned into the SAM the path expects by aligning every read segment against its contig window on the device with A1
// (k_myers_hw_path / k_myers_hw_path_grouped: banded Myers bit vectors with edlib's own traceback, hs_kernels_myers.hip),
// i.e. exactly what edlibAlign(read segment, window, k = -1, EDLIB_MODE_HW, EDLIB_TASK_PATH) of the reference's bundled edlib
// (edlib.cpp:94-106,560-700,947-1130) returns for the pair: start and end on the window, and the path.
//
//   window of a PAF line   = contig[max(0, tstart - HS_REALIGN_PAD) .. min(L, tend + HS_REALIGN_PAD))      (pad: 100 bases)
//   query                  = read[qstart .. qend), reverse-complemented for strand '-'
//   SAM line               = qname, flag 0 / 16, tname, POS = window start + start location + 1, MAPQ 60,
//                            CIGAR = <left clip>S <path as M / I / D runs> <right clip>S, SEQ / QUAL '*', NM:i:<edit distance>, LN:i:<read length>
//                            (edlib's moves: 0 '=' and 3 'X' -> M, 1 -> I (query base the window lacks), 2 -> D)
// The SAM is a file of its own (the caller names it: HS_call_variants puts it into its tmpDir), so the rest of the path --
// and the reference itself, which is how the mode is checked -- reads it like any other SAM.
// Opt-in at the drop-in boundary: HS_call_variants only takes a .paf when HS_REALIGN=1; without it the reference's refusal stands.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "hs_host.h"

namespace hs {

namespace {
struct PafRec { long read; long contig; int qlen, qs, qe, ts, te; bool minus; };
struct DevMem {      // hs_malloc'ed block, freed with the scope
    void* p = nullptr;
    ~DevMem() { if (p) hs_free(p); }
    int alloc(size_t n) { if (p) { hs_free(p); p = nullptr; } return hs_malloc(&p, n ? n : 16); }
};
}  // namespace

int realign_paf_to_sam(const std::string& gfa, const std::string& reads, const std::string& paf, const std::string& out_sam, int n_threads, RealignStats* stats) {
    if (n_threads < 1) n_threads = 1;
    const auto t_begin = std::chrono::steady_clock::now();
    SeqSet seqs;
    if (int rc = load_sequences(gfa, reads, seqs, n_threads)) return rc;
    std::unordered_map<std::string, long> read_of, contig_of;
    for (size_t i = 0; i < seqs.read_names.size(); ++i) read_of[seqs.read_names[i]] = (long)i;      // (a later read of the same name replaces the earlier one, as the path's own name table)
    for (size_t i = 0; i < seqs.contig_names.size(); ++i) contig_of[seqs.contig_names[i]] = (long)i;
    // ---- the PAF lines: qname qlen qstart qend strand tname tlen tstart tend ... ----
    std::vector<PafRec> recs;
    long n_unknown = 0;      // PAF lines that name a read or a contig of neither input file
    {
        std::ifstream f(paf);
        if (!f) { set_error("Input file '" + paf + "' could not be read"); std::cout << "problem reading PAF file " << paf << std::endl; return HS_EIO; }
        std::string line, qn, strand, tn;
        while (std::getline(f, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream ss(line);
            PafRec r; long tlen = 0;
            if (!(ss >> qn >> r.qlen >> r.qs >> r.qe >> strand >> tn >> tlen >> r.ts >> r.te)) continue;
            auto iq = read_of.find(qn); auto it = contig_of.find(tn);
            if (iq == read_of.end() || it == contig_of.end()) { n_unknown++; continue; }      // (a name of neither file: no record; counted, reported below)
            r.read = iq->second; r.contig = it->second; r.minus = strand == "-";
            const long rl = (long)(seqs.read_off[(size_t)r.read + 1] - seqs.read_off[(size_t)r.read]);
            const long cl = (long)(seqs.contig_off[(size_t)r.contig + 1] - seqs.contig_off[(size_t)r.contig]);
            if (r.qlen != rl || r.qs < 0 || r.qe > rl || r.qs >= r.qe || r.ts < 0 || r.te > cl || r.ts >= r.te) {
                set_error("PAF line does not fit its read / contig: " + line.substr(0, 200));
                std::cout << "ERROR: a PAF line names coordinates outside its read or contig: " << line.substr(0, 200) << std::endl;
                return HS_EFORMAT;
            }
            recs.push_back(r);
        }
    }
    static const int pad = []() { const char* e = std::getenv("HS_REALIGN_PAD"); const int v = e ? std::atoi(e) : 100; return v >= 0 ? v : 100; }();
    const size_t n = recs.size();
    std::vector<std::string> sam_line(n);
    // ---- the pairs, in chunks (bounded device memory: a pair needs query + window bases and as many move bytes) ----
    static const size_t chunk_bases = []() { const char* e = std::getenv("HS_REALIGN_CHUNK_MB"); const long v = e ? std::atol(e) : 512; return (size_t)std::max(16l, v) << 20; }();
    size_t done = 0;
    long n_aligned = 0; double ms_device = 0;
    while (done < n) {
        size_t k1 = done, bases = 0;
        while (k1 < n && (k1 == done || bases < chunk_bases) && k1 - done < 1000000) {
            const PafRec& r = recs[k1];
            const int w0 = std::max(0, r.ts - pad);
            const long cl = (long)(seqs.contig_off[(size_t)r.contig + 1] - seqs.contig_off[(size_t)r.contig]);
            const int w1 = (int)std::min<long>(cl, (long)r.te + pad);
            bases += (size_t)(r.qe - r.qs) + (size_t)(w1 - w0);
            ++k1;
        }
        const size_t m = k1 - done;
        std::vector<int64_t> q_off(m + 1, 0), t_off(m + 1, 0), o_off(m + 1, 0);
        std::vector<int> win0(m);
        for (size_t i = 0; i < m; ++i) {
            const PafRec& r = recs[done + i];
            const int w0 = std::max(0, r.ts - pad);
            const long cl = (long)(seqs.contig_off[(size_t)r.contig + 1] - seqs.contig_off[(size_t)r.contig]);
            const int w1 = (int)std::min<long>(cl, (long)r.te + pad);
            win0[i] = w0;
            q_off[i + 1] = q_off[i] + (r.qe - r.qs);
            t_off[i + 1] = t_off[i] + (w1 - w0);
            o_off[i + 1] = o_off[i] + (r.qe - r.qs) + (w1 - w0);
        }
        std::vector<uint8_t> hq((size_t)q_off[m] + 16), ht((size_t)t_off[m] + 16);
        hs_parallel_for((int)m, n_threads, [&](int i) {
            const PafRec& r = recs[done + (size_t)i];
            const uint8_t* rs = seqs.read_seq.data() + seqs.read_off[(size_t)r.read];
            uint8_t* q = hq.data() + q_off[(size_t)i];
            const int len = r.qe - r.qs;
            if (!r.minus) std::memcpy(q, rs + r.qs, (size_t)len);
            else for (int j = 0; j < len; ++j) q[j] = (uint8_t)(3 - rs[r.qe - 1 - j]);      // reverse complement (codes A C G T = 0 1 2 3)
            const uint8_t* cs = seqs.contig_seq.data() + seqs.contig_off[(size_t)r.contig];
            std::memcpy(ht.data() + t_off[(size_t)i], cs + win0[(size_t)i], (size_t)(t_off[(size_t)i + 1] - t_off[(size_t)i]));
        });
        DevMem dq, dt, dd, ds, de, dops, dlen;
        if (dq.alloc(hq.size()) || dt.alloc(ht.size()) || dd.alloc(m * 4) || ds.alloc(m * 4) || de.alloc(m * 4) || dops.alloc((size_t)o_off[m] + 16) || dlen.alloc(m * 4)) return HS_EHIP;
        if (hs_memcpy_h2d(dq.p, hq.data(), hq.size()) || hs_memcpy_h2d(dt.p, ht.data(), ht.size())) return HS_EHIP;
        const auto t0 = std::chrono::steady_clock::now();
        if (int rc = hs_edlib_hw_align((const uint8_t*)dq.p, q_off.data(), (const uint8_t*)dt.p, t_off.data(), (int32_t)m, (int32_t*)dd.p, (int32_t*)ds.p, (int32_t*)de.p,
                                       (uint8_t*)dops.p, o_off.data(), (int32_t*)dlen.p, nullptr)) return rc;
        if (hs_device_synchronize()) return HS_EHIP;
        ms_device += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::vector<int32_t> dist(m), start(m), olen(m);
        std::vector<uint8_t> ops((size_t)o_off[m] + 16);
        if (hs_memcpy_d2h(dist.data(), dd.p, m * 4) || hs_memcpy_d2h(start.data(), ds.p, m * 4) || hs_memcpy_d2h(olen.data(), dlen.p, m * 4)
            || hs_memcpy_d2h(ops.data(), dops.p, (size_t)o_off[m])) return HS_EHIP;
        hs_parallel_for((int)m, n_threads, [&](int i) {
            const PafRec& r = recs[done + (size_t)i];
            if (olen[(size_t)i] < 0) return;      // (edlib has no alignment for this pair: no record)
            std::string& s = sam_line[done + (size_t)i];
            s.reserve(256 + (size_t)olen[(size_t)i] / 4);
            s += seqs.read_names[(size_t)r.read]; s += '\t'; s += r.minus ? "16" : "0"; s += '\t'; s += seqs.contig_names[(size_t)r.contig]; s += '\t';
            s += std::to_string(win0[(size_t)i] + start[(size_t)i] + 1); s += "\t60\t";
            const int clip_l = r.minus ? r.qlen - r.qe : r.qs, clip_r = r.minus ? r.qs : r.qlen - r.qe;
            if (clip_l > 0) { s += std::to_string(clip_l); s += 'S'; }
            const uint8_t* mv = ops.data() + o_off[(size_t)i];
            int run = 0; char cur = 0;
            for (int j = 0; j < olen[(size_t)i]; ++j) {
                const char c = mv[j] == 1 ? 'I' : (mv[j] == 2 ? 'D' : 'M');
                if (c == cur) { run++; continue; }
                if (run) { s += std::to_string(run); s += cur; }
                cur = c; run = 1;
            }
            if (run) { s += std::to_string(run); s += cur; }
            if (clip_r > 0) { s += std::to_string(clip_r); s += 'S'; }
            s += "\t*\t0\t0\t*\t*\tNM:i:"; s += std::to_string(dist[(size_t)i]); s += "\tLN:i:"; s += std::to_string(r.qlen); s += '\n';
        });
        for (size_t i = 0; i < m; ++i) if (olen[i] >= 0) n_aligned++;
        done = k1;
    }
    {
        std::ofstream out(out_sam, std::ios::binary);
        if (!out) { set_error("cannot write " + out_sam); return HS_EIO; }
        std::string hdr = "@HD\tVN:1.6\tSO:unknown\n";
        for (size_t c = 0; c < seqs.contig_names.size(); ++c)
            hdr += "@SQ\tSN:" + seqs.contig_names[c] + "\tLN:" + std::to_string(seqs.contig_off[c + 1] - seqs.contig_off[c]) + "\n";
        hdr += "@PG\tID:hairsplitter_amd\tPN:hs_realign_paf\tDS:read segments aligned to their contig windows on the device (Myers bit vectors, edlib's HW path)\n";
        out.write(hdr.data(), (std::streamsize)hdr.size());
        for (const std::string& s : sam_line) if (!s.empty()) out.write(s.data(), (std::streamsize)s.size());
    }
    if (n_unknown) std::fprintf(stderr, "hairsplitter: realign: %ld PAF lines name a read or a contig that is in neither input file and were left out\n", n_unknown);
    if (n_aligned != (int64_t)n)      // (never silent: these records are missing from the SAM the stage goes on with)
        std::fprintf(stderr, "hairsplitter: realign: %ld of %ld PAF records have no alignment (the device aligner found none, or the read segment is longer than 2^20 bases) and were left out\n",
                     (long)((int64_t)n - n_aligned), (long)n);
    if (stats) {
        stats->n_lines = (int64_t)n; stats->n_aligned = n_aligned; stats->ms_device = ms_device;
        stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        int64_t qb = 0; for (const PafRec& r : recs) qb += r.qe - r.qs;
        stats->query_bases = qb;
    }
    return HS_OK;
}

}  // namespace hs

extern "C" int hs_realign_paf(const char* gfa, const char* reads, const char* paf, const char* out_sam, int32_t n_threads, hs_realign_stats* stats) {
    if (!gfa || !reads || !paf || !out_sam) { hs::set_error("hs_realign_paf: null argument"); return HS_EINVAL; }
    if (hs_device_count() <= 0) { hs::set_error("no HIP device available: the HairSplitter MI355X path has no CPU fallback"); return HS_ENODEVICE; }
    hs::RealignStats st;
    const int rc = hs::realign_paf_to_sam(gfa, reads, paf, out_sam, n_threads > 0 ? n_threads : hs::host_threads(), &st);
    if (stats) { stats->n_lines = st.n_lines; stats->n_aligned = st.n_aligned; stats->query_bases = st.query_bases; stats->ms_device = st.ms_device; stats->ms_total = st.ms_total; }
    return rc;
}
