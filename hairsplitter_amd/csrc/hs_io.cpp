// hs_io.cpp -- file boundary of the two drop-in executables: GFA / FASTA|FASTQ / SAM readers that flatten
// straight into the device batch layout, and the .col / .vcf / error_rate / .gro writers.
// Contracts follow the reference parsers (input_output.cpp:39-109 parse_reads, :120-264 parse_assembly,
// :274-536 parse_SAM, :546-569 parse_reads_on_contig) and writers (call_variants.cpp:1174-1213,1377;
// separate_reads.cpp:1754-1786); see SURVEY.md §8(b).
#include "hs_host.h"
#include "hs_driver.h"

#include <algorithm>
#include <chrono>
#include <cctype>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string_view>
#include <thread>
#include <unordered_map>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <signal.h>
#include <cerrno>
#include <unistd.h>

namespace hs {

// read-only view of a whole file (mmap; plain read() for things that cannot be mapped)
struct FileView {
    const char* p = nullptr;
    size_t n = 0;
    void* map = nullptr;
    std::string fallback;
    // n_threads > 1: the page tables of a large mapping are filled on the host threads (MADV_POPULATE_READ per slice: one thread needs
    // 25 ms per gigabyte of cached file for it), else by the mmap call itself
    bool open(const std::string& path, int n_threads = 1) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (::fstat(fd, &st) != 0) { ::close(fd); return false; }
        if (S_ISREG(st.st_mode) && st.st_size > 0) {
            const size_t len = (size_t)st.st_size, SL = (size_t)32 << 20;
            const bool par = n_threads > 1 && len >= 2 * SL;
            void* m = ::mmap(nullptr, len, PROT_READ, MAP_PRIVATE | (par ? 0 : MAP_POPULATE), fd, 0);
            if (m != MAP_FAILED) {
                map = m; p = (const char*)m; n = len; ::close(fd);
#ifdef MADV_POPULATE_READ
                if (par) {
                    const int slices = (int)((len + SL - 1) / SL);
                    hs_parallel_for(slices, n_threads, [&](int k) {
                        const size_t a = (size_t)k * SL, e = std::min(len, a + SL);
                        (void)::madvise((char*)m + a, e - a, MADV_POPULATE_READ);      // (refused by an older kernel: the pages come in as they are touched)
                    });
                }
#endif
                return true;
            }
        }
        char buf[1 << 16];
        ssize_t k;
        while ((k = ::read(fd, buf, sizeof buf)) > 0) fallback.append(buf, (size_t)k);
        ::close(fd);
        p = fallback.data(); n = fallback.size();
        return true;
    }
    ~FileView() { if (map) ::munmap(map, n); }
};

struct BaseLut {                                   // sequence.cpp:13-23: everything that is not A/C/G is T
    uint8_t t[256];
    BaseLut() { std::memset(t, 3, sizeof t); t[(unsigned char)'A'] = 0; t[(unsigned char)'C'] = 1; t[(unsigned char)'G'] = 2; }
};
static const BaseLut g_lut;

struct Line { const char* p; size_t n; };
// lines of a buffer ('\n' separated, a last line without newline counts, no empty line after a final newline): the buffer
// is cut into one piece per thread at line starts, the pieces are scanned with memchr and concatenated
static std::vector<Line> split_lines(const char* s, size_t n, int n_threads) {
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, n_threads), n / (1 << 20) + 1));
    std::vector<size_t> cut((size_t)T + 1, n);
    cut[0] = 0;
    for (int t = 1; t < T; ++t) {
        const size_t raw = n / T * t;
        const void* nl = std::memchr(s + raw - 1, '\n', n - (raw - 1));
        cut[(size_t)t] = nl ? (size_t)((const char*)nl - s) + 1 : n;
    }
    std::vector<std::vector<Line>> part((size_t)T);
    hs_parallel_for(T, n_threads, [&](int t) {
        std::vector<Line>& v = part[(size_t)t];
        size_t i = cut[(size_t)t];
        const size_t e = cut[(size_t)t + 1];
        while (i < e) {
            const void* nl = std::memchr(s + i, '\n', e - i);
            const size_t j = nl ? (size_t)((const char*)nl - s) : e;
            v.push_back(Line{s + i, j - i});
            i = j + 1;
        }
    });
    if (T == 1) return std::move(part[0]);
    std::vector<Line> all;
    size_t tot = 0;
    for (auto& v : part) tot += v.size();
    all.reserve(tot);
    for (auto& v : part) all.insert(all.end(), v.begin(), v.end());
    return all;
}
static std::string_view first_token(const char* p, size_t n) {   // name up to the first blank
    size_t k = 0;
    while (k < n && p[k] != ' ') k++;
    return std::string_view(p, k);
}
static int atoi_n(const char* p, size_t n) {        // std::atoi on a field that is not NUL terminated
    size_t i = 0;
    while (i < n && (p[i] == ' ' || (p[i] >= '\t' && p[i] <= '\r'))) i++;
    bool neg = false;
    if (i < n && (p[i] == '+' || p[i] == '-')) { neg = p[i] == '-'; i++; }
    long v = 0;
    while (i < n && p[i] >= '0' && p[i] <= '9') { v = v * 10 + (p[i] - '0'); i++; }
    return (int)(neg ? -v : v);
}

template <class Ops>
static int parse_cigar(const char* cg, size_t n, Ops& ops) {
    // tools.cpp:27-57 semantics: digits accumulate, any other byte closes a run of that operation.
    if (n == 1 && cg[0] == '*') return 0;
    long num = -1;
    for (size_t i = 0; i < n; ++i) {
        const char c = cg[i];
        if (c >= '0' && c <= '9') { num = (num < 0 ? 0 : num) * 10 + (c - '0'); if (num > 0x0fffffff) return -1; }
        else {
            if (num < 0) return -1;   // the reference's stoi("") throws -> "could not convert" exit(1)
            uint32_t op;
            switch (c) {
                case 'M': op = 0; break; case 'I': op = 1; break; case 'D': op = 2; break; case 'N': op = 3; break;
                case 'S': op = 4; break; case 'H': op = 5; break; case 'P': op = 6; break; case '=': op = 7; break;
                case 'X': op = 8; break; default: op = 6; break;   // unknown letters are inert in generate_msa
            }
            ops.push_back(((uint32_t)num << 4) | op);
            num = -1;
        }
    }
    return 0;
}

namespace {
// the CIGAR ops of a record are [cig_off, cig_off + cig_n) of the buffer its block (or the replay pass) parses into: one growing
// array per block instead of one heap allocation per record (171 k allocations of 16 KB on 16 threads: a quarter of the stage's CPU
// time went into glibc growing its per-thread arenas, mprotect by mprotect)
using CigarBuf = std::vector<uint32_t, NoInitAlloc<uint32_t>>;
struct SamRec { size_t line; int32_t contig; int32_t read, pos; uint8_t strand; int32_t r0, r1, c0, c1; int64_t need; size_t cig_off, cig_n; const CigarBuf* buf; };
enum SamLineResult { SAM_SKIP = 0, SAM_REC = 1, SAM_DEFER = 2, SAM_BAD_CIGAR = 3 };

// One alignment line (input_output.cpp:300-528). `lookup(name, is_query, &found)` resolves a name to its index in the
// reference's single name table (reads first, then contigs); the parallel pass defers lines with unknown names.
template <class Lookup>
SamLineResult parse_sam_line(const Line& l, size_t line_no, long n_reads, long n_contigs, bool amplicon, Lookup&& lookup, SamRec& r, std::string* bad_cigar, CigarBuf& ops) {
    const char* cg = nullptr; size_t cgn = 0;
    long seq1 = -1, seq2 = -2;
    int length1 = 0, pos2_1 = -1, flag = 0, nonmatching = 0;
    bool positive = true, allgood = true;
    int fieldnumber = 0;
    size_t a = 0;
    while (true) {
        // (the sequence and quality fields of a long read are tens of kilobytes: memchr, not a byte loop)
        const char* tab = a < l.n ? (const char*)std::memchr(l.p + a, '\t', l.n - a) : nullptr;
        const size_t b = tab ? (size_t)(tab - l.p) : l.n;
        const char* f = l.p + a; const size_t fn = b - a;
        if (fieldnumber == 0) {
            bool found = true, defer = false;
            seq1 = lookup(std::string_view(f, fn), true, &found, &defer);
            if (defer) return SAM_DEFER;
            if (!found) {
                std::cout << "WARNING: read in the sam file not found in reads file, ignoring: " << std::string(f, fn) << std::endl;
                allgood = false;
            }
        } else if (fieldnumber == 1) {
            flag = atoi_n(f, fn);
            if (flag % 8 >= 4) allgood = false;
            if (flag % 32 >= 16) positive = false;
        } else if (fieldnumber == 2) {
            bool found = true, defer = false;
            seq2 = lookup(std::string_view(f, fn), false, &found, &defer);
            if (defer) return SAM_DEFER;
        } else if (fieldnumber == 3) pos2_1 = atoi_n(f, fn);
        else if (fieldnumber == 5) { cg = f; cgn = fn; }
        else if (fn >= 5 && std::memcmp(f, "LN:i:", 5) == 0) length1 = atoi_n(f + 5, fn - 5);
        else if (fn >= 5 && std::memcmp(f, "NM:i:", 5) == 0) nonmatching = atoi_n(f + 5, fn - 5);
        fieldnumber++;
        if (b >= l.n) break;
        a = b + 1;
    }
    if (!(allgood && fieldnumber > 10 && seq2 != seq1)) return SAM_SKIP;
    const size_t cig0 = ops.size();
    auto drop = [&](SamLineResult k) { ops.resize(cig0); return k; };      // (a line that yields no record leaves nothing behind)
    if (parse_cigar(cg, cgn, ops) != 0) { if (bad_cigar) bad_cigar->assign(cg, cgn); return drop(SAM_BAD_CIGAR); }
    const uint32_t* cig = ops.data() + cig0;
    const size_t ncig = ops.size() - cig0;
    auto clip = [&](bool front, uint32_t what) -> int {
        if (ncig == 0) return 0;
        uint32_t op = front ? cig[0] : cig[ncig - 1];
        return (op & 15u) == what ? (int)(op >> 4) : 0;
    };
    int nbH_start = clip(true, 5), nbH_end = clip(false, 5);
    int nbS_start = clip(true, 4), nbS_end = clip(false, 4);
    if (ncig == 1) {   // single-op CIGAR: the reference's backward scan sees the same run from both ends
        nbH_end = nbH_start; nbS_end = nbS_start;
    }
    if (!positive) { std::swap(nbH_start, nbH_end); std::swap(nbS_start, nbS_end); }
    if (nbH_start + nbH_end > 0.2 * length1 && flag < 2048) allgood = false;
    else if (flag % 512 >= 256) allgood = false;
    if (amplicon && nonmatching > 0.2 * length1) allgood = false;
    if (!allgood) return drop(SAM_SKIP);
    int length_read = 0, length_contig = 0;
    int64_t need = 0;
    for (size_t q = 0; q < ncig; ++q) {
        const uint32_t op = cig[q];
        const uint32_t c = op & 15u; const int len = (int)(op >> 4);
        if (c == 0 || c == 7 || c == 8) { length_read += len; length_contig += len; }
        else if (c == 1) length_read += len;
        else if (c == 2) length_contig += len;
        if (c == 0 || c == 1 || c == 4 || c == 5 || c == 7 || c == 8) need += len;
    }
    r.line = line_no; r.need = need; r.cig_off = cig0; r.cig_n = ncig; r.buf = &ops;
    r.read = (int32_t)seq1; r.pos = pos2_1 - 1; r.strand = positive ? 1 : 0;
    r.r0 = nbS_start + nbH_start; r.r1 = nbS_start + nbH_start + length_read;
    r.c0 = pos2_1 - 1; r.c1 = pos2_1 + length_contig;
    const long ci = seq2 - n_reads;
    if (ci < 0 || ci >= n_contigs) return drop(SAM_SKIP);   // target is not a contig of the GFA
    if (seq1 >= n_reads) return drop(SAM_SKIP);             // contig-on-contig records are outside this path's contract
    r.contig = (int32_t)ci;
    return SAM_REC;
}
}  // namespace

// every read (FASTA / FASTQ, records as input_output.cpp:39-109 cuts them) and every contig (S lines) of a job, coded
int load_sequences(const std::string& gfa, const std::string& reads, SeqSet& out, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    FileView rtxt;
    if (!rtxt.open(reads)) { std::cout << "problem reading files in index_reads, while trying to read " << reads << std::endl; set_error("Input file could not be read: " + reads); return HS_EIO; }
    char format = '@';
    if ((reads.size() > 6 && reads.substr(reads.size() - 6, 6) == ".fasta") || (reads.size() >= 3 && reads.substr(reads.size() - 3, 3) == ".fa")) format = '>';
    std::vector<Line> rl = split_lines(rtxt.p, rtxt.n, n_threads);
    std::vector<Line> seq_of_read;
    {
        std::vector<size_t> buffer;
        char lastlinestart = '+';
        auto flush = [&]() {
            const Line& h = rl[buffer[0]];
            out.read_names.emplace_back(first_token(h.p + (h.n ? 1 : 0), h.n ? h.n - 1 : 0));
            seq_of_read.push_back(rl[buffer[1]]);
        };
        for (size_t li = 0; li < rl.size(); ++li) {
            const Line& l = rl[li];
            const char first = l.n ? l.p[0] : '\0';
            if (first == format && buffer.size() >= 2 && (((lastlinestart != '+' || buffer.size() == 4) && format == '@') || format == '>')) { flush(); buffer.clear(); buffer.push_back(li); }
            else buffer.push_back(li);
            if (l.n > 0) lastlinestart = l.p[0];
        }
        if (buffer.size() >= 2) flush();
    }
    const long n_reads = (long)out.read_names.size();
    out.read_off.assign((size_t)n_reads + 1, 0);
    for (long i = 0; i < n_reads; ++i) out.read_off[(size_t)i + 1] = out.read_off[(size_t)i] + (int64_t)seq_of_read[(size_t)i].n;
    out.read_seq.resize((size_t)out.read_off.back());
    {
        const int RB = (int)std::max<long>(1, std::min<long>((long)n_threads * 8, std::max<long>(n_reads, 1)));
        hs_parallel_for(RB, n_threads, [&](int bi) {
            const long i0 = n_reads * bi / RB, i1 = n_reads * (bi + 1) / RB;
            for (long i = i0; i < i1; ++i) {
                const Line& s = seq_of_read[(size_t)i];
                uint8_t* o = out.read_seq.data() + out.read_off[(size_t)i];
                for (size_t k = 0; k < s.n; ++k) o[k] = g_lut.t[(unsigned char)s.p[k]];
            }
        });
    }
    FileView gtxt;
    if (!gtxt.open(gfa)) { std::cout << "problem reading files in index_reads, while trying to read " << gfa << std::endl; set_error("Input file could not be read: " + gfa); return HS_EIO; }
    out.contig_off.assign(1, 0);
    std::vector<Line> seqs;
    for (const Line& l : split_lines(gtxt.p, gtxt.n, n_threads)) {
        if (!l.n || l.p[0] != 'S') continue;
        size_t a = 0; int field = 0; std::string_view name;
        while (a <= l.n) {
            const char* tab = a < l.n ? (const char*)std::memchr(l.p + a, '\t', l.n - a) : nullptr;
            const size_t b = tab ? (size_t)(tab - l.p) : l.n;
            if (field == 1) name = first_token(l.p + a, b - a);
            else if (field == 2) { seqs.push_back(Line{l.p + a, b - a}); out.contig_off.push_back(out.contig_off.back() + (int64_t)(b - a)); out.contig_names.emplace_back(name); }
            field++;
            if (b >= l.n) break;
            a = b + 1;
        }
    }
    out.contig_seq.resize((size_t)out.contig_off.back());
    hs_parallel_for((int)seqs.size(), n_threads, [&](int c) {
        uint8_t* o = out.contig_seq.data() + out.contig_off[(size_t)c];
        const Line& s = seqs[(size_t)c];
        for (size_t k = 0; k < s.n; ++k) o[k] = g_lut.t[(unsigned char)s.p[k]];
    });
    return HS_OK;
}

int load_cv_inputs(const std::string& gfa, const std::string& reads, const std::string& sam, bool amplicon, CvFileInput& in, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    const bool tim = std::getenv("HS_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!tim) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[hs timing]   load: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t_last).count());
        t_last = n;
    };
    // one name table for reads and contigs, as in the reference (reads first, contigs appended: input_output.cpp:134,241);
    // a later entry of the same name replaces the earlier one
    std::unordered_map<std::string_view, long> indices;
    // ---- reads: names, lengths, sequence line of each record (input_output.cpp:39-109) ----
    FileView rtxt;
    if (!rtxt.open(reads, n_threads)) {
        std::cout << "problem reading files in index_reads, while trying to read " << reads << std::endl;
        set_error("Input file could not be read: " + reads);
        return HS_EIO;
    }
    char format = '@';
    if ((reads.size() > 6 && reads.substr(reads.size() - 6, 6) == ".fasta") || (reads.size() >= 3 && reads.substr(reads.size() - 3, 3) == ".fa")) format = '>';
    lap("map reads");
    std::vector<Line> rl = split_lines(rtxt.p, rtxt.n, n_threads);
    lap("split reads");
    std::vector<Line> seq_of_read;
    {
        std::vector<size_t> buffer;   // line indices
        char lastlinestart = '+';
        auto flush = [&]() {
            const Line& h = rl[buffer[0]];
            const std::string_view name = first_token(h.p + (h.n ? 1 : 0), h.n ? h.n - 1 : 0);
            in.read_names.emplace_back(name);
            seq_of_read.push_back(rl[buffer[1]]);
            indices[name] = (long)in.read_names.size() - 1;
        };
        for (size_t li = 0; li < rl.size(); ++li) {
            const Line& l = rl[li];
            const char first = l.n ? l.p[0] : '\0';
            if (first == format && buffer.size() >= 2 && (((lastlinestart != '+' || buffer.size() == 4) && format == '@') || format == '>')) {
                flush();
                buffer.clear();
                buffer.push_back(li);
            } else buffer.push_back(li);
            if (l.n > 0) lastlinestart = l.p[0];
        }
        if (buffer.size() >= 2) flush();
    }
    const long n_reads = (long)in.read_names.size();
    lap("index reads");

    // ---- contigs (input_output.cpp:120-264, S lines) ----
    FileView gtxt;
    if (!gtxt.open(gfa)) {
        std::cout << "problem reading files in index_reads, while trying to read " << gfa << std::endl;
        set_error("Input file could not be read: " + gfa);
        return HS_EIO;
    }
    in.contig_off.assign(1, 0);
    {
        std::vector<Line> seqs;
        for (const Line& l : split_lines(gtxt.p, gtxt.n, n_threads)) {
            if (!l.n || l.p[0] != 'S') continue;
            // fields are tab separated: S <name> <sequence> ...
            size_t a = 0; int field = 0; std::string_view name;
            while (a <= l.n) {
                const char* tab = a < l.n ? (const char*)std::memchr(l.p + a, '\t', l.n - a) : nullptr;
                const size_t b = tab ? (size_t)(tab - l.p) : l.n;
                if (field == 1) name = first_token(l.p + a, b - a);
                else if (field == 2) {
                    seqs.push_back(Line{l.p + a, b - a});
                    in.contig_off.push_back(in.contig_off.back() + (int64_t)(b - a));
                    indices[name] = n_reads + (long)in.contig_names.size();
                    in.contig_names.emplace_back(name);
                    in.contig_skip.push_back(name == "edge_124@009" ? 1 : 0);   // call_variants.cpp:1283
                }
                field++;
                if (b >= l.n) break;
                a = b + 1;
            }
        }
        in.contig_seq.resize((size_t)in.contig_off.back());
        hs_parallel_for((int)seqs.size(), n_threads, [&](int c) {
            uint8_t* o = in.contig_seq.data() + in.contig_off[(size_t)c];
            const Line& s = seqs[(size_t)c];
            for (size_t k = 0; k < s.n; ++k) o[k] = g_lut.t[(unsigned char)s.p[k]];
        });
    }
    const long n_contigs = (long)in.contig_names.size();
    lap("contigs");

    // ---- alignments (input_output.cpp:274-536): lines are independent once the names resolve, so they are parsed in
    // blocks on all threads. A name that is in neither file goes through the reference's operator[] (:325), which inserts
    // it with index 0: its first appearance as a query is refused, later ones are taken as read 0. Lines with such names are
    // set aside and replayed in file order afterwards. ----
    FileView stxt;
    if (!stxt.open(sam, n_threads)) {
        std::cout << "problem reading SAM file " << sam << std::endl;
        set_error("Input file '" + sam + "' could not be read");
        return HS_EIO;
    }
    std::vector<Line> sl = split_lines(stxt.p, stxt.n, n_threads);
    lap("map + split sam");
    const int NB = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads * 4, sl.size() / 64 + 1));
    struct Block { std::vector<SamRec> recs; CigarBuf ops; std::vector<size_t> deferred; size_t bad_line = (size_t)-1; std::string bad_cigar; };
    std::vector<Block> blocks((size_t)NB);
    hs_parallel_for(NB, n_threads, [&](int bi) {
        Block& B = blocks[(size_t)bi];
        const size_t l0 = sl.size() * (size_t)bi / NB, l1 = sl.size() * ((size_t)bi + 1) / NB;
        auto lookup = [&](std::string_view name, bool, bool* found, bool* defer) -> long {
            auto it = indices.find(name);
            if (it == indices.end()) { *found = false; *defer = true; return 0; }
            return it->second;
        };
        SamRec r;
        {   // room for the block's ops up front: a CIGAR op is at least two characters and the CIGAR a fraction of its line
            size_t bytes = 0;
            for (size_t li = l0; li < l1; ++li) bytes += sl[li].n;
            B.ops.reserve(bytes / 6 + 1024);
        }
        for (size_t li = l0; li < l1; ++li) {
            const Line& l = sl[li];
            if (l.n == 0 || l.p[0] == '@') continue;
            const SamLineResult k = parse_sam_line(l, li, n_reads, n_contigs, amplicon, lookup, r, &B.bad_cigar, B.ops);
            if (k == SAM_REC) B.recs.push_back(r);
            else if (k == SAM_DEFER) B.deferred.push_back(li);
            else if (k == SAM_BAD_CIGAR) { B.bad_line = li; break; }
        }
    });
    lap("parse sam");
    for (const Block& B : blocks)
        if (B.bad_line != (size_t)-1) {
            std::cout << "ERROR : could not convert " << B.bad_cigar << " to int" << std::endl;
            set_error("malformed CIGAR " + B.bad_cigar);
            return HS_EFORMAT;
        }
    std::vector<SamRec> replayed;
    CigarBuf replayed_ops;
    {
        std::unordered_map<std::string, long> inserted;   // names operator[] would have added, all with index 0
        std::string bad;
        for (const Block& B : blocks)
            for (size_t li : B.deferred) {
                auto lookup = [&](std::string_view name, bool, bool* found, bool* defer) -> long {
                    *defer = false;
                    auto it = indices.find(name);
                    if (it != indices.end()) return it->second;
                    auto ins = inserted.emplace(std::string(name), 0L);
                    *found = !ins.second;
                    return 0;
                };
                SamRec r;
                const SamLineResult k = parse_sam_line(sl[li], li, n_reads, n_contigs, amplicon, lookup, r, &bad, replayed_ops);
                if (k == SAM_REC) replayed.push_back(r);
                else if (k == SAM_BAD_CIGAR) {
                    std::cout << "ERROR : could not convert " << bad << " to int" << std::endl;
                    set_error("malformed CIGAR " + bad);
                    return HS_EFORMAT;
                }
            }
    }
    // records of each contig in file order
    std::vector<std::vector<SamRec*>> per_contig((size_t)n_contigs);
    for (Block& B : blocks) for (SamRec& r : B.recs) per_contig[(size_t)r.contig].push_back(&r);
    if (!replayed.empty()) {
        for (SamRec& r : replayed) per_contig[(size_t)r.contig].push_back(&r);
        for (auto& v : per_contig) std::stable_sort(v.begin(), v.end(), [](const SamRec* x, const SamRec* y) { return x->line < y->line; });
    }

    // ---- flatten; load only the reads that are aligned somewhere (input_output.cpp:546-569) ----
    std::vector<char> needed((size_t)n_reads, 0);
    for (auto& v : per_contig) for (SamRec* r : v) needed[(size_t)r->read] = 1;
    in.read_off.assign((size_t)n_reads + 1, 0);
    for (long i = 0; i < n_reads; ++i) in.read_off[(size_t)i + 1] = in.read_off[(size_t)i] + (needed[(size_t)i] ? (int64_t)seq_of_read[(size_t)i].n : 0);
    in.read_seq.resize((size_t)in.read_off.back());
    {
        const int RB = (int)std::max<long>(1, std::min<long>((long)n_threads * 8, n_reads));
        hs_parallel_for(RB, n_threads, [&](int bi) {
            const long i0 = n_reads * bi / RB, i1 = n_reads * (bi + 1) / RB;
            for (long i = i0; i < i1; ++i) {
                if (!needed[(size_t)i]) continue;
                const Line& s = seq_of_read[(size_t)i];
                uint8_t* o = in.read_seq.data() + in.read_off[(size_t)i];
                for (size_t k = 0; k < s.n; ++k) o[k] = g_lut.t[(unsigned char)s.p[k]];
            }
        });
    }
    lap("code reads");
    size_t n_rec = 0;
    in.contig_rec_off.assign(1, 0);
    for (long c = 0; c < n_contigs; ++c) { n_rec += per_contig[(size_t)c].size(); in.contig_rec_off.push_back((int32_t)n_rec); }
    in.rec_read.resize(n_rec); in.rec_pos.resize(n_rec); in.rec_strand.resize(n_rec);
    in.rec_r0.resize(n_rec); in.rec_r1.resize(n_rec); in.rec_c0.resize(n_rec); in.rec_c1.resize(n_rec);
    in.rec_cig_off.assign(n_rec + 1, 0);
    std::vector<SamRec*> flat(n_rec);
    {
        size_t k = 0;
        for (long c = 0; c < n_contigs; ++c)
            for (SamRec* r : per_contig[(size_t)c]) {
                // validate: the CIGAR must not run past the read (the reference would index past the string)
                const int64_t have = in.read_off[(size_t)r->read + 1] - in.read_off[(size_t)r->read];
                if (r->need > have) {
                    set_error("CIGAR of read " + in.read_names[(size_t)r->read] + " consumes more bases than the read has");
                    std::cout << "ERROR: CIGAR of read " << in.read_names[(size_t)r->read] << " is longer than the read" << std::endl;
                    return HS_EFORMAT;
                }
                flat[k] = r;
                in.rec_cig_off[k + 1] = in.rec_cig_off[k] + (int64_t)r->cig_n;
                k++;
            }
    }
    in.cigar.resize((size_t)in.rec_cig_off.back());
    {
        const int FB = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads * 4, n_rec));
        hs_parallel_for(FB, n_threads, [&](int bi) {
            const size_t k0 = n_rec * (size_t)bi / FB, k1 = n_rec * ((size_t)bi + 1) / FB;
            for (size_t k = k0; k < k1; ++k) {
                const SamRec* r = flat[k];
                in.rec_read[k] = r->read; in.rec_pos[k] = r->pos; in.rec_strand[k] = r->strand;
                in.rec_r0[k] = r->r0; in.rec_r1[k] = r->r1; in.rec_c0[k] = r->c0; in.rec_c1[k] = r->c1;
                if (r->cig_n) std::memcpy(in.cigar.data() + in.rec_cig_off[k], r->buf->data() + r->cig_off, r->cig_n * sizeof(uint32_t));
            }
        });
    }
    lap("flatten");
    return HS_OK;
}


// ---------------------------------------------------------------------------------------------------
// .col reader and the writers. Text formatting / parsing is per contig block and independent: blocks are handled on all
// threads, the file itself is written with one write per block, in contig order.
// ---------------------------------------------------------------------------------------------------
namespace {

inline void put_int(std::string& s, long long v) {
    char buf[24];
    auto r = std::to_chars(buf, buf + sizeof buf, v);
    s.append(buf, (size_t)(r.ptr - buf));
}
inline void put_float_g(std::string& s, double v) {   // what `ostream << float` prints: %g with 6 significant digits
    char buf[48];
    const int n = std::snprintf(buf, sizeof buf, "%g", v);
    s.append(buf, (size_t)n);
}

struct Tok { const char* p; size_t n; };
// whitespace-separated tokens of a line, as `istringstream >>` yields them
inline bool next_tok(const char*& c, const char* e, Tok& t) {
    while (c < e && (*c == ' ' || (*c >= '\t' && *c <= '\r'))) ++c;
    if (c >= e) return false;
    t.p = c;
    while (c < e && !(*c == ' ' || (*c >= '\t' && *c <= '\r'))) ++c;
    t.n = (size_t)(c - t.p);
    return true;
}

}  // namespace

// parse_column_file: separate_reads.cpp:46-190 (integer- or character-encoded .col, decided by the first SNPS line)
int parse_col(const std::string& path, float rsa, std::vector<ColFileContig>& cs, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    FileView txt;
    if (!txt.open(path)) return 0;   // the reference's ifstream simply reads nothing
    const std::vector<Line> lines = split_lines(txt.p, txt.n, n_threads);
    const int max_coverage = 1000000000;   // :1420-1426: uninitialised shadowed variable, observed "unlimited"
    // block boundaries (CONTIG lines) and the encoding, which the first complete SNPS line after a CONTIG decides (:93-95)
    std::vector<size_t> starts;
    bool numbers = false, decided = false;
    for (size_t li = 0; li < lines.size(); ++li) {
        const char* c = lines[li].p; const char* e = c + lines[li].n;
        Tok t;
        if (!next_tok(c, e, t)) continue;
        if (t.n == 6 && std::memcmp(t.p, "CONTIG", 6) == 0) starts.push_back(li);
        else if (!decided && !starts.empty() && t.n == 4 && std::memcmp(t.p, "SNPS", 4) == 0) {
            Tok pos, r, s2;
            if (next_tok(c, e, pos) && next_tok(c, e, r) && next_tok(c, e, s2)) {
                numbers = !std::isalpha((unsigned char)r.p[0]) && r.p[0] != '-';
                decided = true;
            }
        }
    }
    cs.assign(starts.size(), ColFileContig());
    std::vector<int> rc_of(starts.size(), 0);
    std::vector<std::string> err_of(starts.size());
    hs_parallel_for((int)starts.size(), n_threads, [&](int b) {
        ColFileContig& cc = cs[(size_t)b];
        const size_t l0 = starts[(size_t)b], l1 = (size_t)b + 1 < starts.size() ? starts[(size_t)b + 1] : lines.size();
        std::vector<char> codes;
        std::vector<int> ridx;
        for (size_t li = l0; li < l1 && !rc_of[(size_t)b]; ++li) {
            const Line& L = lines[li];
            const char* c = L.p; const char* e = c + L.n;
            Tok type;
            if (!next_tok(c, e, type)) continue;
            if (li == l0) {   // the CONTIG line itself
                cc.contig_line.assign(L.p, L.n);
                Tok name, len;
                if (next_tok(c, e, name)) cc.name.assign(name.p, name.n);
                cc.length = next_tok(c, e, len) ? atoi_n(len.p, len.n) : 0;
            } else if (type.n == 4 && std::memcmp(type.p, "SNPS", 4) == 0) {
                Tok pos, ref_s, sec_s, idx_s, content;
                if (!next_tok(c, e, pos)) continue;
                if (!next_tok(c, e, ref_s) || !next_tok(c, e, sec_s)) continue;
                char ref_base, sec_base;
                if (numbers) { ref_base = (char)atoi_n(ref_s.p, ref_s.n); sec_base = (char)atoi_n(sec_s.p, sec_s.n); }
                else { ref_base = ref_s.p[0]; sec_base = sec_s.p[0]; }
                const bool has_idx = next_tok(c, e, idx_s), has_content = has_idx && next_tok(c, e, content);
                codes.clear(); ridx.clear();
                if (has_content) {   // comma-terminated tokens
                    size_t t0 = 0;
                    for (size_t k = 0; k < content.n; ++k)
                        if (content.p[k] == ',') {
                            const char* tp = content.p + t0; const size_t tn = k - t0;
                            if (tn == 1 && tp[0] == ' ') codes.push_back(' ');
                            else if (numbers) codes.push_back((char)(unsigned char)atoi_n(tp, tn));
                            else for (size_t q = 0; q < tn; ++q) codes.push_back(tp[q]);
                            t0 = k + 1;
                        }
                }
                if (has_idx) {
                    size_t t0 = 0;
                    for (size_t k = 0; k < idx_s.n; ++k)
                        if (idx_s.p[k] == ',') { ridx.push_back(atoi_n(idx_s.p + t0, k - t0)); t0 = k + 1; }
                }
                int cov_maj = 0, cov_sec = 0, cov = 0;
                const size_t keep_from = cc.col_idx.size();
                for (size_t n = 0; n < codes.size() && n < ridx.size(); ++n) {
                    if (codes[n] != ' ' && cov < max_coverage) {
                        cc.col_code.push_back((uint8_t)codes[n]); cc.col_idx.push_back(ridx[n]);
                        if (codes[n] == ref_base) cov_maj++; else if (codes[n] == sec_base) cov_sec++;
                    }
                    if (codes[n] != ' ' && ridx[n] >= 0) cov++;
                }
                if ((float)cov_sec >= rsa * (float)(cov_maj + cov_sec)) {
                    cc.snp_pos.push_back(atoi_n(pos.p, pos.n)); cc.snp_ref.push_back((uint8_t)ref_base); cc.snp_alt.push_back((uint8_t)sec_base);
                    cc.col_off.push_back((int64_t)cc.col_idx.size());
                } else { cc.col_idx.resize(keep_from); cc.col_code.resize(keep_from); }
            } else if (type.n == 4 && std::memcmp(type.p, "READ", 4) == 0) {
                cc.read_lines.emplace_back(L.p, L.n);
                Tok name, sR, eR, sC, eC;
                const bool ok = next_tok(c, e, name) && next_tok(c, e, sR) && next_tok(c, e, eR) && next_tok(c, e, sC) && next_tok(c, e, eC);
                auto is_num = [](const Tok& t) {   // what strtol accepts at the start of the token
                    size_t i = 0;
                    if (i < t.n && (t.p[i] == '+' || t.p[i] == '-')) ++i;
                    return i < t.n && t.p[i] >= '0' && t.p[i] <= '9';
                };
                if (!ok || !is_num(sC) || !is_num(eC)) { rc_of[(size_t)b] = 1; err_of[(size_t)b].assign(L.p, L.n); break; }
                cc.read_start.push_back((int32_t)atoi_n(sC.p, sC.n)); cc.read_end.push_back((int32_t)atoi_n(eC.p, eC.n));
            }
        }
        // a SNPS line names reads by their place among the contig's READ lines: an index outside them (a damaged file) would be read
        // as it stands by the reference (out of bounds there) and by the kernels here -- refused instead
        if (!rc_of[(size_t)b]) {
            const int32_t nr = (int32_t)cc.read_start.size();
            for (int32_t v : cc.col_idx) if (v < 0 || v >= nr) { rc_of[(size_t)b] = 2; err_of[(size_t)b] = cc.contig_line; break; }
        }
    });
    for (size_t b = 0; b < starts.size(); ++b) {
        if (rc_of[b] == 1) { std::cout << "error in parsing read limits" << std::endl << "line : " << err_of[b] << std::endl; return 1; }
        if (rc_of[b] == 2) { std::cout << "error in parsing SNPS: a read index outside the contig's READ lines" << std::endl << "contig : " << err_of[b] << std::endl; return 1; }
    }
    return 0;
}

int write_col_sidecar(const CvFileInput& in, const hs_cv_result* res, const std::vector<std::string>& col_block, const std::string& col_path, int n_threads);
int write_cv_outputs(const CvFileInput& in, const hs_cv_result* res, const std::string& error_rate_out, const std::string& col_path,
                     const std::string& vcf_path, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    const int C = (int)in.contig_names.size();
    float total = 0; int n = 0;
    for (int c = 0; c < C; ++c) {
        if (in.contig_skip[(size_t)c]) continue;   // call_variants.cpp:1283
        if (res->mean_distance[c] > 0) { total += res->mean_distance[c]; n += 1; }
    }
    {
        std::ofstream er(error_rate_out);
        std::cout << "total error rate : " << total << " number of contigs : " << n << std::endl;
        er << total / n << std::endl;
    }
    // the VCF header of :1242-1247 is overwritten when output_files reopens the file (:1177): rows only
    std::vector<std::string> col_block((size_t)C), vcf_block((size_t)C);
    hs_parallel_for(C, n_threads, [&](int c) {
        if (in.contig_skip[(size_t)c]) return;
        std::string& out = col_block[(size_t)c];
        std::string& vcf = vcf_block[(size_t)c];
        const int64_t L = in.contig_off[(size_t)c + 1] - in.contig_off[(size_t)c];
        const int64_t s0 = res->snp_off[c], s1 = res->snp_off[c + 1];
        out.reserve((size_t)(res->col_off[s1] - res->col_off[s0]) * 7 + (size_t)(in.contig_rec_off[(size_t)c + 1] - in.contig_rec_off[(size_t)c]) * 48 + 256);
        out += "CONTIG\t"; out += in.contig_names[(size_t)c]; out += '\t'; put_int(out, L); out += '\t'; put_float_g(out, res->depth[c]); out += '\n';
        for (int r = in.contig_rec_off[(size_t)c]; r < in.contig_rec_off[(size_t)c + 1]; ++r) {
            out += "READ\t"; out += in.read_names[(size_t)in.rec_read[(size_t)r]]; out += '\t';
            put_int(out, in.rec_r0[(size_t)r]); out += '\t'; put_int(out, in.rec_r1[(size_t)r]); out += '\t';
            put_int(out, in.rec_c0[(size_t)r]); out += '\t'; put_int(out, in.rec_c1[(size_t)r]); out += '\t';
            out += in.rec_strand[(size_t)r] ? '1' : '0'; out += '\n';
        }
        for (int64_t s = s0; s < s1; ++s) {
            out += "SNPS\t"; put_int(out, res->snp_pos[s]); out += '\t'; put_int(out, (int)res->snp_ref[s]); out += '\t'; put_int(out, (int)res->snp_alt[s]); out += '\t';
            {   // the two lists of a SNPS line (nine tenths of the file's bytes): digits written straight into the string's storage
                const int64_t e0 = res->col_off[s], e1 = res->col_off[s + 1];
                const size_t at = out.size();
                out.resize(at + (size_t)(e1 - e0) * (11 + 4) + 2);      // ("4294967295," and "255," at most per entry)
                char* w = &out[at];
                for (int64_t e = e0; e < e1; ++e) {
                    uint32_t v = (uint32_t)res->col_idx[e];      // (read indices: never negative)
                    char tmp[10]; int k = 0;
                    do { tmp[k++] = (char)('0' + v % 10u); v /= 10u; } while (v);
                    while (k) *w++ = tmp[--k];
                    *w++ = ',';
                }
                *w++ = '\t';
                for (int64_t e = e0; e < e1; ++e) {
                    const unsigned v = res->col_code[e];
                    if (v >= 100) { *w++ = (char)('0' + v / 100); *w++ = (char)('0' + v / 10 % 10); *w++ = (char)('0' + v % 10); }
                    else if (v >= 10) { *w++ = (char)('0' + v / 10); *w++ = (char)('0' + v % 10); }
                    else *w++ = (char)('0' + v);
                    *w++ = ',';
                }
                *w++ = '\n';
                out.resize((size_t)(w - out.data()));
            }
            vcf += in.contig_names[(size_t)c]; vcf += '\t'; put_int(vcf, res->snp_pos[s]); vcf += "\t.\t"; vcf += "ACGT-"[(res->snp_ref[s] - '!') % 5];
            vcf += '\t'; vcf += "ACGT-"[(res->snp_alt[s] - '!') % 5]; vcf += "\t.\t.\tDP="; put_int(vcf, res->col_off[s + 1] - res->col_off[s]); vcf += '\n';
        }
        out += '\n';
        vcf += '\n';
    });
    // the binary companion for HS_separate_reads is formed and written on a thread of its own while this one writes the text (the
    // companion is only ever used next to a .col of exactly the size and the block hashes it records: whichever file is complete first,
    // a process that dies in between leaves nothing a reader would take)
    std::thread side;
    const std::string side_path = col_path + ".hsbin";
    std::remove(side_path.c_str());      // (a companion of an earlier .col of this name must not outlive it)
    if (res->col_idx || res->col_off[res->snp_off[C]] == 0) side = std::thread([&] {
        // the companion is an optimisation: whatever goes wrong with it (memory, the disk) must leave the .col / .vcf alone and no half-written companion behind
        try { write_col_sidecar(in, res, col_block, col_path, std::max(1, n_threads / 2)); }
        catch (const std::exception& e) { std::fprintf(stderr, "hairsplitter: the binary companion of %s was not written (%s)\n", col_path.c_str(), e.what()); std::remove(side_path.c_str()); }
        catch (...) { std::fprintf(stderr, "hairsplitter: the binary companion of %s was not written\n", col_path.c_str()); std::remove(side_path.c_str()); }
    });
    {
        std::ofstream out(col_path, std::ios::binary), vcf(vcf_path, std::ios::binary);
        for (int c = 0; c < C; ++c) {
            if (in.contig_skip[(size_t)c]) continue;
            out.write(col_block[(size_t)c].data(), (std::streamsize)col_block[(size_t)c].size());
            vcf.write(vcf_block[(size_t)c].data(), (std::streamsize)vcf_block[(size_t)c].size());
        }
    }
    if (side.joinable()) side.join();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// <out.col>.hsbin -- the binary side-channel between the two executables. The reference hands stage 3's result to
// stage 4 as text (call_variants.cpp:1197-1204 writes, separate_reads.cpp:84-170 parses it back): 300 MB of decimal numbers for the
// 500-contig job, and parsing them is most of what HS_separate_reads does before the GPU gets anything. The .col file stays what
// it is (written first, complete, the file every other tool and --resume read); next to it HS_call_variants leaves the same
// content as flat arrays -- per contig: where its block lies in the .col, a hash of that block, the READ limits, the SNPs with
// their counts, the columns as CSR. HS_separate_reads takes the arrays only if the .col it was given still IS that file (size and
// every block's hash), applies the rarest-strain filter of :151-167 from the counts, and points at the READ lines of the .col
// itself for the .gro; anything else -- no companion, another size, one differing block -- and it parses the text as before.
// HS_NO_SIDECAR=1: neither written nor read.
// ---------------------------------------------------------------------------------------------------
namespace {
constexpr uint64_t kSidecarMagic = 0x0001004e49425348ull;      // "HSBIN\0\1\0"
constexpr uint64_t kSidecarVersion = 1;
struct SidecarHeader { uint64_t magic, version, col_size, n_contigs, table_off, file_size, pad[2]; };
struct SidecarEntry { uint64_t col_off, col_bytes, hash, header_len, n_reads, n_snps, n_entries, length, name_len, data_off, data_bytes, pad; };
// 64-bit block hash: four interleaved multiply-xorshift lanes over 8-byte words (speed of memory; not cryptographic -- it guards
// against a .col that was edited or replaced, not against an adversary)
uint64_t block_hash(const char* p, size_t n) {
    uint64_t h[4] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
    size_t i = 0;
    for (; i + 32 <= n; i += 32)
        for (int k = 0; k < 4; ++k) { uint64_t w; std::memcpy(&w, p + i + 8 * k, 8); h[k] = (h[k] ^ w) * 0x9FB21C651E98DF25ull; h[k] ^= h[k] >> 29; }
    uint64_t tail = 0x2545F4914F6CDD1Dull;
    for (; i < n; ++i) tail = (tail ^ (unsigned char)p[i]) * 0x100000001B3ull;
    uint64_t r = (uint64_t)n * 0xD6E8FEB86659FD93ull;
    for (int k = 0; k < 4; ++k) { r = (r ^ h[k]) * 0x9FB21C651E98DF25ull; r ^= r >> 32; }
    return (r ^ tail) * 0x9E3779B97F4A7C15ull;
}
bool sidecar_off() { static const bool off = std::getenv("HS_NO_SIDECAR") != nullptr; return off; }
template <class T> void put_arr(std::string& o, const T* p, size_t n) { o.append(reinterpret_cast<const char*>(p), n * sizeof(T)); while (o.size() & 7) o += '\0'; }
}  // namespace

int write_col_sidecar(const CvFileInput& in, const hs_cv_result* res, const std::vector<std::string>& col_block, const std::string& col_path, int n_threads) {
    if (sidecar_off()) return 0;
    const int C = (int)in.contig_names.size();
    std::vector<int> order;
    for (int c = 0; c < C; ++c) if (!in.contig_skip[(size_t)c]) order.push_back(c);
    const size_t K = order.size();
    std::vector<SidecarEntry> tab(K);
    std::vector<std::string> data(K);
    uint64_t off = 0;
    for (size_t k = 0; k < K; ++k) { tab[k] = SidecarEntry(); tab[k].col_off = off; tab[k].col_bytes = col_block[(size_t)order[k]].size(); off += tab[k].col_bytes; }
    const uint64_t col_size = off;
    hs_parallel_for((int)K, n_threads, [&](int k) {
        const int c = order[(size_t)k];
        const std::string& blk = col_block[(size_t)c];
        SidecarEntry& e = tab[(size_t)k];
        e.hash = block_hash(blk.data(), blk.size());
        const int r0 = in.contig_rec_off[(size_t)c], nr = in.contig_rec_off[(size_t)c + 1] - r0;
        const int64_t s0 = res->snp_off[c], s1 = res->snp_off[c + 1], S = s1 - s0;
        const int64_t e0 = res->col_off[s0], E = res->col_off[s1] - e0;
        e.n_reads = (uint64_t)nr; e.n_snps = (uint64_t)S; e.n_entries = (uint64_t)E;
        e.length = (uint64_t)(in.contig_off[(size_t)c + 1] - in.contig_off[(size_t)c]);
        e.name_len = 0;      // (what parse_column_file takes for the name: the CONTIG line's second token)
        while (e.name_len < in.contig_names[(size_t)c].size() && !std::isspace((unsigned char)in.contig_names[(size_t)c][(size_t)e.name_len])) e.name_len++;
        // the lines of the block: CONTIG, then the READ lines (their offsets in the block), then SNPS
        std::vector<int32_t> line_off((size_t)nr + 1);
        size_t pos = blk.find('\n');
        e.header_len = pos == std::string::npos ? blk.size() : pos;
        pos = pos == std::string::npos ? blk.size() : pos + 1;
        for (int r = 0; r < nr; ++r) { line_off[(size_t)r] = (int32_t)pos; const size_t nl = blk.find('\n', pos); pos = nl == std::string::npos ? blk.size() : nl + 1; }
        line_off[(size_t)nr] = (int32_t)pos;
        std::vector<int32_t> n_ref((size_t)S), n_alt((size_t)S);
        std::vector<int64_t> coff((size_t)S + 1);
        for (int64_t q = 0; q < S; ++q) {      // what parse_column_file recounts (separate_reads.cpp:151-167)
            int a = 0, b2 = 0;
            const uint8_t rb = res->snp_ref[s0 + q], sb = res->snp_alt[s0 + q];
            for (int64_t x = res->col_off[s0 + q]; x < res->col_off[s0 + q + 1]; ++x) { if (res->col_code[x] == rb) a++; else if (res->col_code[x] == sb) b2++; }
            n_ref[(size_t)q] = a; n_alt[(size_t)q] = b2; coff[(size_t)q] = res->col_off[s0 + q] - e0;
        }
        coff[(size_t)S] = E;
        std::string& o = data[(size_t)k];
        o.reserve((size_t)nr * 12 + (size_t)S * 30 + (size_t)E * 5 + 256);
        put_arr(o, line_off.data(), line_off.size());
        put_arr(o, in.rec_c0.data() + r0, (size_t)nr);
        put_arr(o, in.rec_c1.data() + r0, (size_t)nr);
        put_arr(o, res->snp_pos + s0, (size_t)S);
        put_arr(o, n_ref.data(), (size_t)S);
        put_arr(o, n_alt.data(), (size_t)S);
        put_arr(o, coff.data(), coff.size());
        put_arr(o, res->col_idx ? res->col_idx + e0 : nullptr, (size_t)E);
        put_arr(o, res->snp_ref + s0, (size_t)S);
        put_arr(o, res->snp_alt + s0, (size_t)S);
        put_arr(o, res->col_code ? res->col_code + e0 : nullptr, (size_t)E);
        put_arr(o, in.contig_names[(size_t)c].data(), (size_t)e.name_len);
        e.data_bytes = o.size();
    });
    SidecarHeader h; std::memset(&h, 0, sizeof h);
    h.magic = kSidecarMagic; h.version = kSidecarVersion; h.col_size = col_size; h.n_contigs = K; h.table_off = sizeof h;
    uint64_t doff = sizeof h + K * sizeof(SidecarEntry);
    for (size_t k = 0; k < K; ++k) { tab[k].data_off = doff; doff += tab[k].data_bytes; }
    h.file_size = doff;
    const std::string tmp = col_path + ".hsbin.tmp", fin = col_path + ".hsbin";
    {
        std::ofstream out(tmp, std::ios::binary);
        if (!out) return 1;
        out.write(reinterpret_cast<const char*>(&h), sizeof h);
        out.write(reinterpret_cast<const char*>(tab.data()), (std::streamsize)(K * sizeof(SidecarEntry)));
        for (size_t k = 0; k < K; ++k) out.write(data[k].data(), (std::streamsize)data[k].size());
        if (!out) { std::remove(tmp.c_str()); return 1; }
    }
    if (std::rename(tmp.c_str(), fin.c_str()) != 0) { std::remove(tmp.c_str()); return 1; }      // (complete or absent, never half a file)
    return 0;
}

// 1: the contigs were taken from the companion of `col_path` (cs filled as parse_col would have filled it); 0: no usable companion
int read_col_sidecar(const std::string& col_path, float rsa, std::vector<ColFileContig>& cs, int n_threads) {
    if (sidecar_off()) return 0;
    FileView bin;
    { struct stat st; if (::stat((col_path + ".hsbin").c_str(), &st) != 0) return 0; }
    if (!bin.open(col_path + ".hsbin") || bin.n < sizeof(SidecarHeader)) return 0;
    SidecarHeader h; std::memcpy(&h, bin.p, sizeof h);
    if (h.magic != kSidecarMagic || h.version != kSidecarVersion || h.file_size != bin.n || h.table_off != sizeof h) return 0;
    if (h.n_contigs > (bin.n - sizeof h) / sizeof(SidecarEntry)) return 0;
    FileView txt;
    if (!txt.open(col_path) || txt.n != h.col_size) return 0;
    const size_t K = (size_t)h.n_contigs;
    const SidecarEntry* tab = reinterpret_cast<const SidecarEntry*>(bin.p + h.table_off);
    std::vector<char> bad(K, 0);
    if (n_threads < 1) n_threads = 1;
    // the .col must still be the file these arrays were written beside: every block in its place with its hash, the arrays inside the companion
    hs_parallel_for((int)K, n_threads, [&](int k) {
        const SidecarEntry& e = tab[k];
        const uint64_t R = e.n_reads, S = e.n_snps, E = e.n_entries;
        if (R > bin.n || S > bin.n || E > bin.n || e.name_len > bin.n) { bad[(size_t)k] = 1; return; }      // (counts no file of this size can hold: the sums below stay far from 2^64)
        auto a8 = [](uint64_t x) { return (x + 7) & ~(uint64_t)7; };
        const uint64_t need = a8((R + 1) * 4) + 2 * a8(R * 4) + 3 * a8(S * 4) + a8((S + 1) * 8) + a8(E * 4) + 2 * a8(S) + a8(E) + a8(e.name_len);
        if (e.col_off > txt.n || e.col_bytes > txt.n - e.col_off || e.data_off > bin.n || e.data_bytes > bin.n - e.data_off || need != e.data_bytes || e.header_len > e.col_bytes
            || R > 0x7fffffff || S > 0x7fffffff) { bad[(size_t)k] = 1; return; }
        if (block_hash(txt.p + e.col_off, (size_t)e.col_bytes) != e.hash) bad[(size_t)k] = 1;
    });
    for (size_t k = 0; k < K; ++k) if (bad[k]) return 0;
    cs.assign(K, ColFileContig());
    hs_parallel_for((int)K, n_threads, [&](int k) {
        const SidecarEntry& e = tab[k];
        ColFileContig& cc = cs[(size_t)k];
        const size_t R = (size_t)e.n_reads, S = (size_t)e.n_snps, E = (size_t)e.n_entries;
        auto a8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
        const char* d = bin.p + e.data_off;
        const int32_t* line_off = reinterpret_cast<const int32_t*>(d); d += a8((R + 1) * 4);
        const int32_t* c0 = reinterpret_cast<const int32_t*>(d); d += a8(R * 4);
        const int32_t* c1 = reinterpret_cast<const int32_t*>(d); d += a8(R * 4);
        const int32_t* spos = reinterpret_cast<const int32_t*>(d); d += a8(S * 4);
        const int32_t* n_ref = reinterpret_cast<const int32_t*>(d); d += a8(S * 4);
        const int32_t* n_alt = reinterpret_cast<const int32_t*>(d); d += a8(S * 4);
        const int64_t* coff = reinterpret_cast<const int64_t*>(d); d += a8((S + 1) * 8);
        const int32_t* cidx = reinterpret_cast<const int32_t*>(d); d += a8(E * 4);
        const uint8_t* sref = reinterpret_cast<const uint8_t*>(d); d += a8(S);
        const uint8_t* salt = reinterpret_cast<const uint8_t*>(d); d += a8(S);
        const uint8_t* ccode = reinterpret_cast<const uint8_t*>(d); d += a8(E);
        const char* blk = txt.p + e.col_off;
        cc.contig_line.assign(blk, (size_t)e.header_len);
        cc.name.assign(d, (size_t)e.name_len);
        cc.length = (long)e.length;
        cc.read_lines.resize(R);
        bool ok = true;
        for (size_t r = 0; r < R && ok; ++r) {
            const int64_t a = line_off[r], b2 = line_off[r + 1];
            if (a < 0 || b2 <= a || (uint64_t)b2 > e.col_bytes) { ok = false; break; }
            cc.read_lines[r].assign(blk + a, (size_t)(b2 - a - 1));      // (without the newline)
        }
        if (!ok) { bad[(size_t)k] = 1; return; }
        cc.read_start.assign(c0, c0 + R); cc.read_end.assign(c1, c1 + R);
        // parse_column_file keeps a SNP iff its second allele is frequent enough among the reads that carry one of the two (:151-167)
        cc.col_off.assign(1, 0);
        bool all = true;
        for (size_t q = 0; q < S && all; ++q) if (!((float)n_alt[q] >= rsa * (float)(n_ref[q] + n_alt[q]))) all = false;
        if (coff[0] != 0 || (uint64_t)coff[S] != E) { bad[(size_t)k] = 1; return; }
        if (all) {
            cc.snp_pos.assign(spos, spos + S); cc.snp_ref.assign(sref, sref + S); cc.snp_alt.assign(salt, salt + S);
            cc.col_off.assign(coff, coff + S + 1); cc.col_idx.assign(cidx, cidx + E); cc.col_code.assign(ccode, ccode + E);
        } else {
            for (size_t q = 0; q < S; ++q) {
                if (!((float)n_alt[q] >= rsa * (float)(n_ref[q] + n_alt[q]))) continue;
                if (coff[q] < 0 || coff[q + 1] < coff[q] || (uint64_t)coff[q + 1] > E) { bad[(size_t)k] = 1; return; }
                cc.snp_pos.push_back(spos[q]); cc.snp_ref.push_back(sref[q]); cc.snp_alt.push_back(salt[q]);
                cc.col_idx.insert(cc.col_idx.end(), cidx + coff[q], cidx + coff[q + 1]);
                cc.col_code.insert(cc.col_code.end(), ccode + coff[q], ccode + coff[q + 1]);
                cc.col_off.push_back((int64_t)cc.col_idx.size());
            }
        }
        for (size_t q = 0; q + 1 < cc.col_off.size(); ++q) if (cc.col_off[q + 1] < cc.col_off[q]) { bad[(size_t)k] = 1; return; }
        const int32_t nr = (int32_t)R;
        for (int32_t v : cc.col_idx) if (v < 0 || v >= nr) { bad[(size_t)k] = 1; return; }
    });
    for (size_t k = 0; k < K; ++k) if (bad[k]) { cs.clear(); return 0; }
    if (std::getenv("HS_SIDECAR_REPORT")) std::fprintf(stderr, "[hs] %s: %zu contigs from the binary companion\n", col_path.c_str(), K);
    return 1;
}

// ---------------------------------------------------------------------------------------------------
// <out.col>.hsgro -- the .gro HS_separate_reads would write for this .col with the orchestrator's usual arguments, left by
// HS_call_variants (hs_call_variants_epilogue: stage 4 run in the process that still has the job's device state, instead of a second
// process that brings the device up again, reads the columns back and runs on cold pools). HS_separate_reads copies it to its output
// ONLY when everything that determines the .gro is what the companion was made with: the .col still is the file the binary companion
// (.hsbin) describes -- size and every block's hash --, the .hsgro belongs to that .hsbin (hash of its table), and error rate, rarest
// strain abundance, low-memory and amplicon switches, seed and the absence of ploidies are equal. Anything else: the normal path.
// Written to <col>.hsgro.tmp and renamed when complete. HS_NO_PRECOMPUTE=1: neither written nor read.
// ---------------------------------------------------------------------------------------------------
namespace {
constexpr uint64_t kGroMagic = 0x0100004F52475348ull;      // "HSGRO", two zero bytes, format 1
struct GroCompanionHeader { uint64_t magic, version, col_size, table_hash, n_contigs, gro_bytes; uint32_t error_rate_bits, rsa_bits, low_memory, amplicon, seed, window; uint64_t pad[2]; };
static_assert(sizeof(GroCompanionHeader) == 88, "GroCompanionHeader layout");
uint32_t f32_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
// the .col at `col_path` is the file its binary companion describes: 1 and the hash of the companion's table, else 0
int sidecar_tag(const std::string& col_path, int n_threads, uint64_t* tag, uint64_t* col_size, uint64_t* n_contigs) {
    if (sidecar_off()) return 0;
    FileView bin;
    { struct stat st; if (::stat((col_path + ".hsbin").c_str(), &st) != 0) return 0; }
    if (!bin.open(col_path + ".hsbin") || bin.n < sizeof(SidecarHeader)) return 0;
    SidecarHeader h; std::memcpy(&h, bin.p, sizeof h);
    if (h.magic != kSidecarMagic || h.version != kSidecarVersion || h.file_size != bin.n || h.table_off != sizeof h) return 0;
    if (h.n_contigs > (bin.n - sizeof h) / sizeof(SidecarEntry)) return 0;
    FileView txt;
    if (!txt.open(col_path) || txt.n != h.col_size) return 0;
    const size_t K = (size_t)h.n_contigs;
    const SidecarEntry* tab = reinterpret_cast<const SidecarEntry*>(bin.p + h.table_off);
    std::vector<char> bad(K, 0);
    hs_parallel_for((int)K, std::max(1, n_threads), [&](int k) {
        const SidecarEntry& e = tab[k];
        if (e.col_off > txt.n || e.col_bytes > txt.n - e.col_off) { bad[(size_t)k] = 1; return; }
        if (block_hash(txt.p + e.col_off, (size_t)e.col_bytes) != e.hash) bad[(size_t)k] = 1;
    });
    for (size_t k = 0; k < K; ++k) if (bad[k]) return 0;
    *tag = block_hash(bin.p + h.table_off, K * sizeof(SidecarEntry)) ^ (h.col_size * 0x9E3779B97F4A7C15ull);
    *col_size = h.col_size; *n_contigs = h.n_contigs;
    return 1;
}
bool precompute_off() { static const bool off = std::getenv("HS_NO_PRECOMPUTE") != nullptr; return off || sidecar_off(); }
}  // namespace

int write_gro_companion(const std::string& col_path, const std::vector<ColFileContig>& cs, const hs_sr_result* res, float error_rate, float rsa, bool low_memory, bool amplicon,
                        uint32_t seed, int32_t window, int n_threads) {
    if (precompute_off()) return 0;
    GroCompanionHeader h; std::memset(&h, 0, sizeof h);
    if (!sidecar_tag(col_path, n_threads, &h.table_hash, &h.col_size, &h.n_contigs)) return 0;
    h.magic = kGroMagic; h.version = 1; h.error_rate_bits = f32_bits(error_rate); h.rsa_bits = f32_bits(rsa); h.low_memory = low_memory; h.amplicon = amplicon; h.seed = seed; h.window = (uint32_t)window;
    const std::string tmp = col_path + ".hsgro.tmp", fin = col_path + ".hsgro";
    { std::ofstream out(tmp, std::ios::binary); if (!out) return 1; out.write(reinterpret_cast<const char*>(&h), sizeof h); if (!out) { std::remove(tmp.c_str()); return 1; } }
    if (write_gro(cs, res, tmp, n_threads)) { std::remove(tmp.c_str()); return 1; }      // (appends)
    struct stat st;
    if (::stat(tmp.c_str(), &st) != 0 || (uint64_t)st.st_size < sizeof h) { std::remove(tmp.c_str()); return 1; }
    h.gro_bytes = (uint64_t)st.st_size - sizeof h;
    { std::fstream out(tmp, std::ios::binary | std::ios::in | std::ios::out); if (!out) { std::remove(tmp.c_str()); return 1; } out.seekp(0); out.write(reinterpret_cast<const char*>(&h), sizeof h); if (!out) { std::remove(tmp.c_str()); return 1; } }
    if (std::rename(tmp.c_str(), fin.c_str()) != 0) { std::remove(tmp.c_str()); return 1; }
    return 0;
}
void remove_gro_companion(const std::string& col_path) { std::remove((col_path + ".hsgro").c_str()); std::remove((col_path + ".hsgro.tmp").c_str()); }
// the marker "a companion is being made": holds the process id of its maker, so that a reader can tell a maker at work from one that died
// ... and the arguments the companion is being made for, so that a reader with other arguments does not wait for it. Returns false when
// no companion is to be made (HS_NO_PRECOMPUTE / no sidecar): the caller then skips the epilogue's stage 4 altogether
bool mark_gro_companion_pending(const std::string& col_path, float error_rate, float rsa, bool low_memory, bool amplicon, uint32_t seed) {
    if (precompute_off()) return false;
    std::ofstream out(col_path + ".hsgro.tmp", std::ios::binary);
    out << (long)::getpid() << "\n" << f32_bits(error_rate) << " " << f32_bits(rsa) << " " << (low_memory ? 1 : 0) << " " << (amplicon ? 1 : 0) << " " << seed << "\n";
    return (bool)out;
}
// 1: the marker names other arguments than these (nothing usable will come of it); 0: the same, or a marker without arguments
static int companion_marker_mismatch(const std::string& tmp, float error_rate, float rsa, bool low_memory, bool amplicon, uint32_t seed) {
    std::ifstream in(tmp, std::ios::binary);
    long pid = 0; unsigned long er = 0, rs = 0, lm = 0, am = 0, sd = 0;
    if (!(in >> pid) || pid <= 0) return 0;
    if (!(in >> er >> rs >> lm >> am >> sd)) return 0;
    return (er != f32_bits(error_rate) || rs != f32_bits(rsa) || lm != (unsigned long)(low_memory ? 1 : 0) || am != (unsigned long)(amplicon ? 1 : 0) || sd != (unsigned long)seed) ? 1 : 0;
}
static bool companion_maker_gone(const std::string& tmp) {
    std::ifstream in(tmp, std::ios::binary);
    long pid = 0;
    if (!(in >> pid) || pid <= 0) return false;      // (no process id in it: the companion itself, about to be renamed -- or somebody else's file: patience decides)
    return ::kill((pid_t)pid, 0) != 0 && errno == ESRCH;
}

// 1: `outfile` holds the .gro (the companion was made for exactly this call); 0: no usable companion
int take_gro_companion(const std::string& col_path, float error_rate, float rsa, bool low_memory, bool amplicon, uint32_t seed, const std::string& outfile, int n_threads) {
    if (precompute_off()) return 0;
    const std::string fin = col_path + ".hsgro", tmp = col_path + ".hsgro.tmp";
    struct stat st;
    if (::stat(fin.c_str(), &st) != 0) {
        // HS_call_variants may still be at it (it writes the companion after its own outputs, its caller has moved on): a moment's patience
        if (::stat(tmp.c_str(), &st) != 0) return 0;
        if (companion_maker_gone(tmp)) { std::remove(tmp.c_str()); return 0; }      // (it died in its epilogue: nothing will come)
        if (companion_marker_mismatch(tmp, error_rate, rsa, low_memory, amplicon, seed)) return 0;      // (it is being made for another call)
        static const long wait_ms = []() { const char* e = std::getenv("HS_PRECOMPUTE_WAIT_MS"); const long v = e ? std::atol(e) : 1500; return v >= 0 ? v : 1500; }();
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            if (::stat(fin.c_str(), &st) == 0) break;
            if (::stat(tmp.c_str(), &st) != 0) { if (::stat(fin.c_str(), &st) == 0) break; return 0; }      // (given up on, or renamed just now)
            if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > (double)wait_ms) return 0;
            if (companion_maker_gone(tmp)) { if (::stat(fin.c_str(), &st) == 0) break; std::remove(tmp.c_str()); return 0; }
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    }
    FileView g;
    if (!g.open(fin) || g.n < sizeof(GroCompanionHeader)) return 0;
    GroCompanionHeader h; std::memcpy(&h, g.p, sizeof h);
    if (h.magic != kGroMagic || h.version != 1 || h.gro_bytes != g.n - sizeof h) return 0;
    if (h.error_rate_bits != f32_bits(error_rate) || h.rsa_bits != f32_bits(rsa) || h.low_memory != (uint32_t)low_memory || h.amplicon != (uint32_t)amplicon || h.seed != seed) return 0;
    uint64_t tag = 0, col_size = 0, nc = 0;
    if (!sidecar_tag(col_path, n_threads, &tag, &col_size, &nc)) return 0;
    if (tag != h.table_hash || col_size != h.col_size || nc != h.n_contigs) return 0;
    std::ofstream out(outfile, std::ios::binary | std::ios::trunc);
    if (!out) return 0;
    out.write(g.p + sizeof h, (std::streamsize)h.gro_bytes);
    return out ? 1 : 0;
}

int write_gro(const std::vector<ColFileContig>& cs, const hs_sr_result* res, const std::string& path, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    std::vector<std::string> block(cs.size());
    hs_parallel_for((int)cs.size(), n_threads, [&](int i) {
        const ColFileContig& cc = cs[(size_t)i];
        if (cc.snp_pos.empty()) return;   // separate_reads.cpp:1522-1524
        std::string& out = block[(size_t)i];
        out += cc.contig_line; out += '\n';
        for (auto& r : cc.read_lines) { out += r; out += '\n'; }
        std::string b;
        for (int64_t w = res->win_off[i]; w < res->win_off[i + 1]; ++w) {
            out += "GROUP\t"; put_int(out, res->win_start[w]); out += '\t'; put_int(out, res->win_end[w]); out += '\t';
            b.clear();
            const int32_t* lab = res->labels + res->label_off[w];
            const int64_t n = res->label_off[w + 1] - res->label_off[w];
            for (int64_t h = 0; h < n; ++h) if (lab[h] != -2) { put_int(out, h); out += ','; put_int(b, lab[h]); b += ','; }
            out += '\t'; out += b; out += '\n';
        }
    });
    std::ofstream out(path, std::ios_base::app | std::ios::binary);
    for (const std::string& b : block) out.write(b.data(), (std::streamsize)b.size());
    return 0;
}

}  // namespace hs
