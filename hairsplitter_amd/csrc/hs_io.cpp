// hs_io.cpp -- file boundary of the two drop-in executables: GFA / FASTA|FASTQ / SAM readers that flatten
// straight into the device batch layout, and the .col / .vcf / error_rate / .gro writers.
// Contracts follow the reference parsers (input_output.cpp:39-109 parse_reads, :120-264 parse_assembly,
// :274-536 parse_SAM, :546-569 parse_reads_on_contig) and writers (call_variants.cpp:1174-1213,1377;
// separate_reads.cpp:1754-1786); see SURVEY.md §8(b).
#include "hs_host.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <unordered_map>

namespace hs {

static bool slurp(const std::string& path, std::string& out) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    in.seekg(0, std::ios::end);
    std::streamoff n = in.tellg();
    in.seekg(0);
    out.resize((size_t)n);
    if (n) in.read(&out[0], n);
    return true;
}

static inline uint8_t base_code(char c) {          // sequence.cpp:13-23: everything that is not A/C/G is T
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3;
}

struct Line { const char* p; size_t n; };
static std::vector<Line> split_lines(const std::string& s) {
    std::vector<Line> v;
    size_t i = 0;
    while (i < s.size()) {
        size_t j = s.find('\n', i);
        if (j == std::string::npos) j = s.size();
        v.push_back(Line{s.data() + i, j - i});
        i = j + 1;
    }
    return v;
}
static std::string first_token(const char* p, size_t n) {   // name up to the first blank
    size_t k = 0;
    while (k < n && p[k] != ' ') k++;
    return std::string(p, k);
}

static int parse_cigar(const std::string& cg, std::vector<uint32_t>& ops) {
    // tools.cpp:27-57 semantics: digits accumulate, any other byte closes a run of that operation.
    if (cg == "*") return 0;
    long num = -1;
    for (char c : cg) {
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

int load_cv_inputs(const std::string& gfa, const std::string& reads, const std::string& sam, bool amplicon, CvFileInput& in) {
    std::unordered_map<std::string, long> indices;
    // ---- reads: names, lengths, sequence line of each record (input_output.cpp:39-109) ----
    std::string rtxt;
    if (!slurp(reads, rtxt)) {
        std::cout << "problem reading files in index_reads, while trying to read " << reads << std::endl;
        set_error("Input file could not be read: " + reads);
        return HS_EIO;
    }
    char format = '@';
    if ((reads.size() > 6 && reads.substr(reads.size() - 6, 6) == ".fasta") || (reads.size() >= 3 && reads.substr(reads.size() - 3, 3) == ".fa")) format = '>';
    std::vector<Line> rl = split_lines(rtxt);
    std::vector<Line> seq_of_read;
    {
        std::vector<size_t> buffer;   // line indices
        char lastlinestart = '+';
        auto flush = [&]() {
            const Line& h = rl[buffer[0]];
            std::string name = first_token(h.p + (h.n ? 1 : 0), h.n ? h.n - 1 : 0);
            in.read_names.push_back(name);
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

    // ---- contigs (input_output.cpp:120-264, S lines) ----
    std::string gtxt;
    if (!slurp(gfa, gtxt)) {
        std::cout << "problem reading files in index_reads, while trying to read " << gfa << std::endl;
        set_error("Input file could not be read: " + gfa);
        return HS_EIO;
    }
    in.contig_off.assign(1, 0);
    for (const Line& l : split_lines(gtxt)) {
        if (!l.n || l.p[0] != 'S') continue;
        // fields are tab separated: S <name> <sequence> ...
        size_t a = 0; int field = 0; std::string name;
        while (a <= l.n) {
            size_t b = a;
            while (b < l.n && l.p[b] != '\t') b++;
            if (field == 1) name = first_token(l.p + a, b - a);
            else if (field == 2) {
                for (size_t k = a; k < b; ++k) in.contig_seq.push_back(base_code(l.p[k]));
                in.contig_off.push_back((int64_t)in.contig_seq.size());
                indices[name] = n_reads + (long)in.contig_names.size();
                in.contig_names.push_back(name);
                in.contig_skip.push_back(name == "edge_124@009" ? 1 : 0);   // call_variants.cpp:1283
            }
            field++;
            if (b >= l.n) break;
            a = b + 1;
        }
    }
    const long n_contigs = (long)in.contig_names.size();

    // ---- alignments (input_output.cpp:274-536) ----
    std::string stxt;
    if (!slurp(sam, stxt)) {
        std::cout << "problem reading SAM file " << sam << std::endl;
        set_error("Input file '" + sam + "' could not be read");
        return HS_EIO;
    }
    struct Rec { int32_t read, pos; uint8_t strand; int32_t r0, r1, c0, c1; std::vector<uint32_t> cig; };
    std::vector<std::vector<Rec>> per_contig((size_t)n_contigs);
    for (const Line& l : split_lines(stxt)) {
        if (l.n && l.p[0] == '@') continue;
        std::string cigar;
        long seq1 = -1, seq2 = -2;
        int length1 = 0, pos2_1 = -1, flag = 0, nonmatching = 0;
        bool positive = true, allgood = true;
        int fieldnumber = 0;
        size_t a = 0;
        if (l.n == 0) continue;
        while (true) {
            size_t b = a;
            while (b < l.n && l.p[b] != '\t') b++;
            std::string field(l.p + a, b - a);
            if (fieldnumber == 0) {
                if (indices.find(field) == indices.end()) {
                    std::cout << "WARNING: read in the sam file not found in reads file, ignoring: " << field << std::endl;
                    allgood = false;
                }
                seq1 = indices[field];   // default-inserts 0 for unknown names, as the reference's operator[] does
            } else if (fieldnumber == 1) {
                flag = std::atoi(field.c_str());
                if (flag % 8 >= 4) allgood = false;
                if (flag % 32 >= 16) positive = false;
            } else if (fieldnumber == 2) seq2 = indices[field];
            else if (fieldnumber == 3) pos2_1 = std::atoi(field.c_str());
            else if (fieldnumber == 5) cigar = field;
            else if (field.compare(0, 5, "LN:i:") == 0) length1 = std::atoi(field.c_str() + 5);
            else if (field.compare(0, 5, "NM:i:") == 0) nonmatching = std::atoi(field.c_str() + 5);
            fieldnumber++;
            if (b >= l.n) break;
            a = b + 1;
        }
        if (!(allgood && fieldnumber > 10 && seq2 != seq1)) continue;
        Rec r;
        if (parse_cigar(cigar, r.cig) != 0) {
            std::cout << "ERROR : could not convert " << cigar << " to int" << std::endl;
            set_error("malformed CIGAR " + cigar);
            return HS_EFORMAT;
        }
        auto clip = [&](bool front, uint32_t what) -> int {
            if (r.cig.empty()) return 0;
            uint32_t op = front ? r.cig.front() : r.cig.back();
            return (op & 15u) == what ? (int)(op >> 4) : 0;
        };
        int nbH_start = clip(true, 5), nbH_end = clip(false, 5);
        int nbS_start = clip(true, 4), nbS_end = clip(false, 4);
        if (r.cig.size() == 1) {   // single-op CIGAR: the reference's backward scan sees the same run from both ends
            nbH_end = nbH_start; nbS_end = nbS_start;
        }
        if (!positive) { std::swap(nbH_start, nbH_end); std::swap(nbS_start, nbS_end); }
        if (nbH_start + nbH_end > 0.2 * length1 && flag < 2048) allgood = false;
        else if (flag % 512 >= 256) allgood = false;
        if (amplicon && nonmatching > 0.2 * length1) allgood = false;
        if (!allgood) continue;
        int length_read = 0, length_contig = 0;
        for (uint32_t op : r.cig) {
            const uint32_t c = op & 15u; const int len = (int)(op >> 4);
            if (c == 0 || c == 7 || c == 8) { length_read += len; length_contig += len; }
            else if (c == 1) length_read += len;
            else if (c == 2) length_contig += len;
        }
        r.read = (int32_t)seq1; r.pos = pos2_1 - 1; r.strand = positive ? 1 : 0;
        r.r0 = nbS_start + nbH_start; r.r1 = nbS_start + nbH_start + length_read;
        r.c0 = pos2_1 - 1; r.c1 = pos2_1 + length_contig;
        const long ci = seq2 - n_reads;
        if (ci < 0 || ci >= n_contigs) continue;   // target is not a contig of the GFA
        if (seq1 >= n_reads) continue;             // contig-on-contig records are outside this path's contract
        per_contig[(size_t)ci].push_back(std::move(r));
    }

    // ---- flatten; load only the reads that are aligned somewhere (input_output.cpp:546-569) ----
    std::vector<char> needed((size_t)n_reads, 0);
    for (auto& v : per_contig) for (auto& r : v) needed[(size_t)r.read] = 1;
    in.read_off.assign(1, 0);
    for (long i = 0; i < n_reads; ++i) {
        if (needed[(size_t)i]) {
            const Line& s = seq_of_read[(size_t)i];
            for (size_t k = 0; k < s.n; ++k) in.read_seq.push_back(base_code(s.p[k]));
        }
        in.read_off.push_back((int64_t)in.read_seq.size());
    }
    in.contig_rec_off.assign(1, 0);
    in.rec_cig_off.assign(1, 0);
    for (long c = 0; c < n_contigs; ++c) {
        for (auto& r : per_contig[(size_t)c]) {
            // validate: the CIGAR must not run past the read (the reference would index past the string)
            int64_t need = 0;
            for (uint32_t op : r.cig) { uint32_t k = op & 15u; if (k == 0 || k == 1 || k == 4 || k == 5 || k == 7 || k == 8) need += op >> 4; }
            const int64_t have = in.read_off[(size_t)r.read + 1] - in.read_off[(size_t)r.read];
            if (need > have) {
                set_error("CIGAR of read " + in.read_names[(size_t)r.read] + " consumes more bases than the read has");
                std::cout << "ERROR: CIGAR of read " << in.read_names[(size_t)r.read] << " is longer than the read" << std::endl;
                return HS_EFORMAT;
            }
            in.rec_read.push_back(r.read); in.rec_pos.push_back(r.pos); in.rec_strand.push_back(r.strand);
            in.rec_r0.push_back(r.r0); in.rec_r1.push_back(r.r1); in.rec_c0.push_back(r.c0); in.rec_c1.push_back(r.c1);
            in.cigar.insert(in.cigar.end(), r.cig.begin(), r.cig.end());
            in.rec_cig_off.push_back((int64_t)in.cigar.size());
        }
        in.contig_rec_off.push_back((int32_t)in.rec_read.size());
    }
    return HS_OK;
}

}  // namespace hs
