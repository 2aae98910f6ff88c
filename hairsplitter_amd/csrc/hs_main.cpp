// hs_main.cpp -- main() of the two drop-in executables, exported through the C ABI so that a host in any
// language can run a stage file-to-file. Same positional argv, same exit codes, same output formats as
// call_variants.cpp:1215-1385 and separate_reads.cpp:1398-1790.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "hs_host.h"
#include "hs_driver.h"

namespace {

bool has_suffix(const std::string& s, const char* suf) {
    const size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

int parse_int_arg(const char* a, int& out) {   // std::stoi semantics: garbage -> exception -> abort in the reference
    char* end = nullptr;
    long v = std::strtol(a, &end, 10);
    if (end == a) return -1;
    out = (int)v;
    return 0;
}

struct StageClock {   // HS_TIMING=1: wall clock of each phase of a stage executable, on stderr
    bool on = std::getenv("HS_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[hs timing] main: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

bool g_leak_at_exit = false;   // hs_main_process_exits(1): the caller is an executable about to _exit -- nothing is torn down

// what HS_call_variants may still do once its outputs are complete (hs_call_variants_epilogue): the .gro of the usual stage-4 call
struct PendingGro { bool on = false; std::string col, err; bool amplicon = false; int threads = 1; } g_pending_gro;

// the contigs of a parsed .col as the C ABI of stage 4 takes them; 1 if a read index is out of range
int sr_contigs_of(std::vector<hs::ColFileContig>& cs, const std::unordered_map<std::string, int>* ploidy_of, std::vector<hs_sr_contig>& hc) {
    hc.resize(cs.size());
    for (size_t i = 0; i < cs.size(); ++i) {
        hs::ColFileContig& c = cs[i];
        hs_sr_contig& h = hc[i];
        h.length = c.length; h.n_reads = (int32_t)c.read_lines.size();
        h.read_start = c.read_start.data(); h.read_end = c.read_end.data();
        h.n_snps = (int32_t)c.snp_pos.size();
        h.snp_pos = c.snp_pos.data(); h.snp_ref = c.snp_ref.data(); h.snp_alt = c.snp_alt.data();
        h.col_off = c.col_off.data(); h.col_idx = c.col_idx.data(); h.col_code = c.col_code.data();
        h.ploidy = 0;
        if (ploidy_of) { auto it = ploidy_of->find(c.name); if (it != ploidy_of->end()) h.ploidy = it->second; }
        for (int32_t v : c.col_idx) if (v < 0 || v >= h.n_reads) return 1;
    }
    return 0;
}
uint32_t stage4_seed() {   // std::random_device of the reference, pinned; override with HS_SEED
    uint32_t seed = 12345u;
    if (const char* s = std::getenv("HS_SEED")) seed = (uint32_t)std::strtoul(s, nullptr, 10);
    return seed;
}

}  // namespace

// The drop-in executables call this before the stage's main: the process ends right after it, so the gigabytes of parsed
// input, the results and the device state are left to the operating system instead of being destroyed piece by piece
// (0.2 s on the 500-contig job). In-process hosts do not call it.
extern "C" void hs_main_process_exits(int yes) { g_leak_at_exit = yes != 0; }
static std::thread g_destroyer;

extern "C" int hs_call_variants_main(int argc, char** argv) {
    if (argc < 12) {   // also how `HS_call_variants --version` is answered (hairsplitter.py:229-252 expects exit 0)
        std::cout << "Usage: ./call_variants <gfa_file> <reads_file> <sam_file> <num_threads> <tmpDir> <error_rate_out> <amplicon> <DEBUG> <file_out> <vcfFile> <automatic_snp_threshold>\n";
        return 0;
    }
    const std::string gfafile = argv[1], reads_file = argv[2], sam_file = argv[3];
    int num_threads = 1, amplicon_i = 0, debug_i = 0;
    if (parse_int_arg(argv[4], num_threads) || parse_int_arg(argv[7], amplicon_i) || parse_int_arg(argv[8], debug_i)) {
        std::cout << "ERROR: could not parse the numeric arguments" << std::endl;
        return 1;
    }
    const std::string error_rate_out = argv[6], file_out = argv[9], vcf_file = argv[10];
    const float automatic_snp_threshold = std::strtof(argv[11], nullptr);
    StageClock clk;
    // truncate (call_variants.cpp:1239-1240) -- on a thread of its own: giving back the pages of an earlier run's 300-MB .col and its companions takes
    // 25 ms, and nothing needs the empty file before the outputs are written (every way out of this function waits for it)
    struct Joiner { std::thread t; ~Joiner() { if (t.joinable()) t.join(); } } truncation;
    truncation.t = std::thread([file_out] {
        { std::ofstream o(file_out); }
        hs::remove_gro_companion(file_out);      // (a precomputed .gro of an earlier .col of this name)
    });
    g_pending_gro.on = false;
    std::string realigned_sam;
    if (has_suffix(sam_file, ".paf")) {
        // the reference refuses a .paf (call_variants.cpp:1256-1259) and so does this executable -- unless HS_REALIGN=1 asks for the
        // read segments to be aligned against their contig windows on the device (hs_realign.cpp): the SAM it makes is read instead
        const char* e = std::getenv("HS_REALIGN");
        if (!(e && e[0] == '1')) {
            std::cout << "ERROR: please provide a .sam file as input for the alignments of the reads on the contigs." << std::endl;
            return EXIT_FAILURE;
        }
        realigned_sam = std::string(argv[5]) + "/hs_realigned.sam";
        std::cout << " - Aligning the read segments of " << sam_file << " against their contig windows on the device\n";
        hs::RealignStats rs;
        if (int rc = hs::realign_paf_to_sam(gfafile, reads_file, sam_file, realigned_sam, num_threads, &rs)) {
            std::cout << "ERROR: " << hs_last_error() << " (" << rc << ")" << std::endl;
            return EXIT_FAILURE;
        }
        if (std::getenv("HS_TIMING")) std::fprintf(stderr, "[hs timing] realign: %ld of %ld PAF lines aligned, %.1f M read bases, device %.1f ms, total %.1f ms\n", (long)rs.n_aligned,
                                                   (long)rs.n_lines, rs.query_bases / 1e6, rs.ms_device, rs.ms_total);
        clk.lap("realign the PAF records");
    }
    const std::string& aln_file = realigned_sam.empty() ? sam_file : realigned_sam;
    if (!has_suffix(aln_file, ".sam")) {
        std::cout << "ERROR: the file containing the alignments on the assembly should be .sam" << std::endl;
        return EXIT_FAILURE;
    }
    // the HIP runtime, the context and the code object come up on a side thread while the inputs are parsed
    int n_devices = 0;
    std::thread warm([&n_devices] { n_devices = hs_warmup(); });
    std::cout << " - Loading reads, contigs and alignments\n";
    hs::CvFileInput* in_p = new hs::CvFileInput();
    struct InGuard { hs::CvFileInput* p; ~InGuard() { if (!g_leak_at_exit) delete p; } } in_guard{in_p};
    hs::CvFileInput& in = *in_p;
    const int load_rc = hs::load_cv_inputs(gfafile, reads_file, aln_file, amplicon_i != 0, in, num_threads);
    clk.lap("load gfa + reads + sam");
    warm.join();
    clk.lap("wait for the device");
    if (n_devices <= 0) {
        std::cout << "ERROR: no HIP device found; this build of HS_call_variants runs on MI355X only" << std::endl;
        return EXIT_FAILURE;
    }
    if (load_rc) {
        std::cout << "ERROR: " << hs_last_error() << std::endl;
        return load_rc == HS_EIO ? 1 : EXIT_FAILURE;
    }
    std::cout << " - Calling variants on each contig\n";
    const int C = (int)in.contig_names.size();
    hs_cv_result* res = nullptr;   // upload + stage 3, sharded over the visible devices (HS_DEVICES) when there are several
    if (int rc = hs_cv_run_host(in.contig_seq.data(), in.contig_off.data(), C, in.read_seq.data(), in.read_off.data(),
                                (int)in.read_names.size(), in.rec_read.data(), in.rec_pos.data(), in.rec_strand.data(),
                                in.rec_cig_off.data(), in.cigar.data(), in.contig_rec_off.data(), automatic_snp_threshold, num_threads, &res)) {
        std::cout << "ERROR: " << hs_last_error() << " (" << rc << ")" << std::endl;
        return EXIT_FAILURE;
    }
    clk.lap("hs_cv_run_host (H2D + stage 3)");
    if (truncation.t.joinable()) truncation.t.join();
    hs::write_cv_outputs(in, res, error_rate_out, file_out, vcf_file, num_threads);
    clk.lap("write .col/.vcf");
    if (res->col_idx || res->col_off[res->snp_off[C]] == 0) {      // the stage's outputs are complete: what may follow is hs_call_variants_epilogue
        // (HS_NO_PRECOMPUTE, or no sidecar: no companion, and no stage 4 behind this stage's back either)
        float er = 0; bool er_ok = false;
        { std::ifstream f(error_rate_out); std::string t; if (f >> t) { double e = std::strtod(t.c_str(), nullptr); if (e > 0.15) e = 0.15; er = (float)e; er_ok = true; } }
        g_pending_gro.col = file_out; g_pending_gro.err = error_rate_out; g_pending_gro.amplicon = amplicon_i != 0; g_pending_gro.threads = num_threads;
        g_pending_gro.on = er_ok && hs::mark_gro_companion_pending(file_out, er, (float)std::atof("0.01"), false, amplicon_i != 0, stage4_seed());
    }
    if (std::getenv("HS_EXIT_PROBE")) {      // (diagnostic: what destroying the parsed input and the result costs here instead of at exit)
        hs_cv_result_destroy(res); clk.lap("destroy the result");
        delete in_p; in_guard.p = nullptr; clk.lap("destroy the parsed input");
        return 0;
    }
    // The executable is about to leave (hs_main_process_exits): the gigabytes of parsed input and the result go back on a thread of their own
    // while the epilogue runs, hs_dropin_finish() waits for it and gives the device's blocks back -- 0.09 s less than leaving all of it to the
    // kernel's teardown after _exit (alternating blocks of runs, 500-contig job: 1.02 s against 0.93). HS_EXIT_LEAK=1: nothing is destroyed.
    if (g_leak_at_exit && !std::getenv("HS_EXIT_LEAK")) {
        hs::CvFileInput* gone = in_p; in_guard.p = nullptr;
        g_destroyer = std::thread([gone, res] { hs_cv_result_destroy(res); delete gone; });
        return 0;
    }
    if (!g_leak_at_exit) hs_cv_result_destroy(res);
    return 0;
}
extern "C" void hs_teardown_probe(void);
extern "C" void hs_dropin_finish(void) {      // (the drop-in executables, after the stage and its epilogue)
    if (std::getenv("HS_EXIT_LEAK") || std::getenv("HS_EXIT_PROBE")) return;
    const auto t0 = std::chrono::steady_clock::now();
    if (g_destroyer.joinable()) g_destroyer.join();
    const auto t1 = std::chrono::steady_clock::now();
    hs_teardown_probe();
    if (std::getenv("HS_TIMING")) std::fprintf(stderr, "[hs timing] exit: waited %.1f ms for the destruction of the inputs, device blocks given back in %.1f ms\n",
                                               std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
}

// After HS_call_variants' outputs are complete: stage 4 for the arguments hairsplitter.py passes by default (hairsplitter.py:686-692,
// 725-726: the error rate it reads back from error_rate_out capped at 0.15, no ploidies, low memory off, rarest strain abundance 0.01,
// the same amplicon switch), in this process -- the device is up, its pools are warm, the columns are a memory-mapped file away -- and
// the .gro left as <col>.hsgro for HS_separate_reads to adopt if that is the call it gets (hs_io.cpp: take_gro_companion). The
// executable calls this after it has reported its exit status (hs_dropin_main.h); a failure here costs nothing but the companion.
extern "C" void hs_call_variants_epilogue(void) {
    if (!g_pending_gro.on) return;
    g_pending_gro.on = false;
    const std::string col = g_pending_gro.col;
    if (std::getenv("HS_EXIT_PROBE")) { hs::remove_gro_companion(col); return; }      // (the diagnostic has torn the device down already)
    const int nt = std::max(1, g_pending_gro.threads);
    StageClock clk;
    try {
        float er = 0;
        { std::ifstream f(g_pending_gro.err); std::string t; if (!(f >> t)) { hs::remove_gro_companion(col); return; } double e = std::strtod(t.c_str(), nullptr); if (e > 0.15) e = 0.15; er = (float)e; }
        const float rsa = (float)std::atof("0.01");
        std::vector<hs::ColFileContig>* cs_p = new std::vector<hs::ColFileContig>();      // (left to the process end like everything else of the executable)
        if (hs::read_col_sidecar(col, rsa, *cs_p, nt) != 1) { hs::remove_gro_companion(col); return; }
        std::vector<hs_sr_contig> hc;
        if (sr_contigs_of(*cs_p, nullptr, hc)) { hs::remove_gro_companion(col); return; }
        const int32_t w = hs_sr_window_size(hc.data(), (int32_t)hc.size(), g_pending_gro.amplicon ? 1 : 0);
        const uint32_t seed = stage4_seed();
        hs_sr_result* res = nullptr;
        if (hs_sr_run(hc.data(), (int32_t)hc.size(), w, er, 0, seed, nt, &res)) { hs::remove_gro_companion(col); return; }
        clk.lap("epilogue: stage 4 for the usual arguments");
        if (hs::write_gro_companion(col, *cs_p, res, er, rsa, false, g_pending_gro.amplicon, seed, w, nt)) hs::remove_gro_companion(col);
        clk.lap("epilogue: write .col.hsgro");
    } catch (...) { hs::remove_gro_companion(col); }
}

extern "C" int hs_separate_reads_main(int argc, char** argv) {
    const char* usage = "Usage: ./separate_reads <columns> <num_threads> <error_rate> <ploidy_of_contigs> <low_memory> <rarest-strain-abundance> <amplicon> <outfile> <DEBUG>";
    if (argc != 10) {
        std::cout << usage << std::endl;
        if (argc == 2 && (argv[1] == std::string("-h") || argv[1] == std::string("--help"))) return 0;
        return 1;
    }
    const std::string columns_file = argv[1], ploidy_file = argv[4], outfile = argv[8];
    const int num_threads = std::atoi(argv[2]);
    const float error_rate = (float)std::atof(argv[3]);
    const bool amplicon = std::atoi(argv[7]) != 0;
    const bool low_memory = std::atoi(argv[5]) != 0;
    const float rsa = (float)std::atof(argv[6]);
    StageClock clk;
    { std::ofstream o(outfile); }
    const uint32_t seed = stage4_seed();
    std::unordered_map<std::string, int> ploidy_of;
    bool have_ploidy = false;
    {
        std::ifstream pf(ploidy_file);
        if (pf) {
            have_ploidy = true;
            std::string line;
            while (std::getline(pf, line)) { std::istringstream iss(line); std::string ctg; int p; if (!(iss >> ctg >> p)) break; ploidy_of[ctg] = p; }
        }
    }
    // the .gro HS_call_variants left for exactly this call (same .col, same arguments, no ploidies): copied, no device needed
    if (ploidy_of.empty() && hs::take_gro_companion(columns_file, error_rate, rsa, low_memory, amplicon, seed, outfile, num_threads) == 1) {
        clk.lap("the precomputed .gro of this call (.col.hsgro)");
        return 0;
    }
    int n_devices = 0;
    std::thread warm([&n_devices] { n_devices = hs_warmup(); });
    std::vector<hs::ColFileContig>* cs_p = new std::vector<hs::ColFileContig>();
    struct CsGuard { std::vector<hs::ColFileContig>* p; ~CsGuard() { if (!g_leak_at_exit) delete p; } } cs_guard{cs_p};
    std::vector<hs::ColFileContig>& cs = *cs_p;
    // the arrays HS_call_variants left beside the .col, if this .col still is the file they describe; else the text
    const bool from_sidecar = hs::read_col_sidecar(columns_file, rsa, cs, num_threads) == 1;
    const int parse_rc = from_sidecar ? 0 : hs::parse_col(columns_file, rsa, cs, num_threads);
    clk.lap(from_sidecar ? "read .col.hsbin" : "parse .col");
    warm.join();
    clk.lap("wait for the device");
    if (n_devices <= 0) {
        std::cout << "ERROR: no HIP device found; this build of HS_separate_reads runs on MI355X only" << std::endl;
        return 1;
    }
    if (parse_rc) return parse_rc;
    std::vector<hs_sr_contig> hc;
    if (sr_contigs_of(cs, have_ploidy ? &ploidy_of : nullptr, hc)) { std::cout << "ERROR: read index out of range in " << columns_file << std::endl; return 1; }
    const int32_t w = hs_sr_window_size(hc.data(), (int32_t)hc.size(), amplicon ? 1 : 0);
    hs_sr_result* res = nullptr;
    if (int rc = hs_sr_run(hc.data(), (int32_t)hc.size(), w, error_rate, low_memory ? 1 : 0, seed, num_threads, &res)) {
        std::cout << "ERROR: " << hs_last_error() << " (" << rc << ")" << std::endl;
        return 1;
    }
    clk.lap("hs_sr_run");
    hs::write_gro(cs, res, outfile, num_threads);
    clk.lap("write .gro");
    if (!g_leak_at_exit) hs_sr_result_destroy(res);
    return 0;
}
