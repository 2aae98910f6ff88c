// hs_gfa_tools.cpp -- the two upstream feeders of stage 3 that are plain text transforms, native:
//   hs_cut_gfa      == src/cut_gfa.py (hairsplitter.py:583 cuts every contig in pieces of <= 300 kb before the reads are
//                      aligned): S lines -> <name>@<k> pieces linked by 0M edges, the L lines re-attached to the first /
//                      last piece of their contigs (cut_gfa.py:33-66)
//   hs_gfa_to_fasta == src/gfa2fa.cpp: ">name tags\nsequence\n" for every S line with a sequence
// Host code (mmap + one pass); byte-identical to the reference's outputs on the fixtures of tests/golden/gfa_tools.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "hs_host.h"

namespace {

std::vector<std::string_view> split_tabs(std::string_view s) {
    std::vector<std::string_view> f;
    size_t a = 0;
    for (;;) {
        const size_t b = s.find('\t', a);
        if (b == std::string_view::npos) { f.push_back(s.substr(a)); break; }
        f.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    return f;
}

// python's str.strip() with no argument: ASCII whitespace on both ends
std::string_view py_strip(std::string_view s) {
    auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; };
    while (!s.empty() && ws(s.front())) s.remove_prefix(1);
    while (!s.empty() && ws(s.back())) s.remove_suffix(1);
    return s;
}

bool read_file(const std::string& path, std::string& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    f.seekg(0);
    out.resize((size_t)n);
    if (n) f.read(&out[0], n);
    return true;
}

// floor((len - 1) / length) as cut_gfa.py computes it (float division, numpy floor): -1 for an empty sequence
long last_piece(long len, long length) { return len >= 1 ? (len - 1) / length : -1; }

}  // namespace

extern "C" int hs_cut_gfa(const char* gfa_in, int64_t length, const char* gfa_out) {
    if (!gfa_in || !gfa_out || length <= 0) { hs::set_error("hs_cut_gfa: bad arguments"); return HS_EINVAL; }
    std::string txt;
    if (!read_file(gfa_in, txt)) { hs::set_error(std::string("hs_cut_gfa: cannot read ") + gfa_in); return HS_EIO; }
    std::string out;
    out.reserve(txt.size() + txt.size() / 64 + 1024);
    std::unordered_map<std::string, long> length_of;      // cut_gfa.py:25,38-41
    std::vector<std::string_view> l_lines;
    size_t a = 0;
    while (a < txt.size()) {   // python iterates lines including their '\n'; the last line may lack it
        size_t b = txt.find('\n', a);
        const bool has_nl = b != std::string::npos;
        if (!has_nl) b = txt.size();
        const std::string_view line(txt.data() + a, b - a + (has_nl ? 1 : 0));
        a = b + 1;
        if (!line.empty() && line[0] == 'S') {
            const std::vector<std::string_view> ls = split_tabs(py_strip(line));      // :37
            if (ls.size() < 3) { if (ls.size() >= 2) length_of[std::string(ls[1])] = 0; else { hs::set_error("hs_cut_gfa: S line without a name"); return HS_EFORMAT; } continue; }
            const std::string name(ls[1]);
            const long len = (long)ls[2].size();
            length_of[name] = len;
            std::string tags;                                                          // "\t".join(ls[3:]).strip("\n")
            for (size_t k = 3; k < ls.size(); ++k) { if (k > 3) tags += '\t'; tags.append(ls[k].data(), ls[k].size()); }
            while (!tags.empty() && tags.back() == '\n') tags.pop_back();
            while (!tags.empty() && tags.front() == '\n') tags.erase(tags.begin());
            const long n_pieces = last_piece(len, length) + 1;                         // :42
            for (long k = 0; k < n_pieces; ++k) {
                if (k * length >= len) continue;
                const long e = std::min((k + 1) * length, len + 1);
                out += "S\t"; out += name; out += '@'; out += std::to_string(k); out += '\t';
                out.append(ls[2].data() + k * length, (size_t)(std::min(e, len) - k * length));
                out += '\t'; out += tags; out += '\n';
                if (k > 0) { out += "L\t"; out += name; out += '@'; out += std::to_string(k - 1); out += "\t+\t"; out += name; out += '@'; out += std::to_string(k); out += "\t+\t0M\n"; }
            }
        } else if (!line.empty() && line[0] == 'L') l_lines.push_back(line);
    }
    for (std::string_view line : l_lines) {                                            // :56-66
        std::string_view body = line;
        while (!body.empty() && body.back() == '\n') body.remove_suffix(1);
        while (!body.empty() && body.front() == '\n') body.remove_prefix(1);
        const std::vector<std::string_view> ls = split_tabs(body);
        if (ls.size() < 6) { hs::set_error("hs_cut_gfa: L line with fewer than six fields (the reference raises IndexError)"); return HS_EFORMAT; }
        auto len_of = [&](std::string_view n, long& v) { auto it = length_of.find(std::string(n)); if (it == length_of.end()) return false; v = it->second; return true; };
        long l1 = 0, l3 = 0;
        if (ls[2] == "+") { if (!len_of(ls[1], l1)) { hs::set_error("hs_cut_gfa: L line names an unknown contig (the reference raises KeyError)"); return HS_EFORMAT; } }
        if (ls[4] == "-") { if (!len_of(ls[3], l3)) { hs::set_error("hs_cut_gfa: L line names an unknown contig (the reference raises KeyError)"); return HS_EFORMAT; } }
        out += "L\t"; out.append(ls[1].data(), ls[1].size());
        if (ls[2] == "+") { out += '@'; out += std::to_string(last_piece(l1, length)); out += "\t+\t"; } else out += "@0\t-\t";
        out.append(ls[3].data(), ls[3].size());
        if (ls[4] == "-") { out += '@'; out += std::to_string(last_piece(l3, length)); out += "\t-\t"; } else out += "@0\t+\t";
        out.append(ls[5].data(), ls[5].size());
        out += '\n';
    }
    std::ofstream o(gfa_out, std::ios::binary);
    if (!o) { hs::set_error(std::string("hs_cut_gfa: cannot write ") + gfa_out); return HS_EIO; }
    o.write(out.data(), (std::streamsize)out.size());
    return HS_OK;
}

// fasta_out == NULL or "-": standard output (as the reference's HS_gfa2fa)
extern "C" int hs_gfa_to_fasta(const char* gfa_in, const char* fasta_out) {
    if (!gfa_in) { hs::set_error("hs_gfa_to_fasta: bad arguments"); return HS_EINVAL; }
    std::string txt;
    if (!read_file(gfa_in, txt)) txt.clear();           // the reference's ifstream simply reads nothing
    std::string out;
    out.reserve(txt.size());
    size_t a = 0;
    while (a < txt.size()) {
        size_t b = txt.find('\n', a);
        if (b == std::string::npos) b = txt.size();
        const std::string_view line(txt.data() + a, b - a);
        a = b + 1;
        if (line.empty() || line[0] != 'S') continue;
        // std::getline(line2, field, '\t'): a trailing empty field after the last tab is not produced
        std::vector<std::string_view> f = split_tabs(line);
        if (!f.empty() && f.back().empty() && line.back() == '\t') f.pop_back();
        if (f.size() < 3 || f[2].empty()) continue;     // gfa2fa.cpp:46-48
        out += '>'; out.append(f[1].data(), f[1].size()); out += ' ';
        for (size_t k = 3; k < f.size(); ++k) { if (k > 3) out += '\t'; out.append(f[k].data(), f[k].size()); }
        out += '\n'; out.append(f[2].data(), f[2].size()); out += '\n';
    }
    if (!fasta_out || !std::strcmp(fasta_out, "-")) { std::fwrite(out.data(), 1, out.size(), stdout); std::fflush(stdout); return HS_OK; }
    std::ofstream o(fasta_out, std::ios::binary);
    if (!o) { hs::set_error(std::string("hs_gfa_to_fasta: cannot write ") + fasta_out); return HS_EIO; }
    o.write(out.data(), (std::streamsize)out.size());
    return HS_OK;
}

// argv of src/cut_gfa.py: --assembly/-a <gfa> --length/-l <n> --output/-o <out>
extern "C" int hs_cut_gfa_main(int argc, char** argv) {
    const char* a = nullptr; const char* l = nullptr; const char* o = nullptr;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        if (k == "-a" || k == "--assembly") a = argv[i + 1];
        else if (k == "-l" || k == "--length") l = argv[i + 1];
        else if (k == "-o" || k == "--output") o = argv[i + 1];
    }
    if (!a || !l || !o) { std::cout << "usage: hs_cut_gfa --assembly ASSEMBLY --length LENGTH --output OUTPUT" << std::endl; return 2; }
    const int rc = hs_cut_gfa(a, std::atoll(l), o);
    if (rc) { std::cout << "ERROR: " << hs_last_error() << std::endl; return 1; }
    return 0;
}
// argv of src/gfa2fa.cpp: <gfa>; FASTA on standard output
extern "C" int hs_gfa2fa_main(int argc, char** argv) {
    if (argc < 2) return 1;
    return hs_gfa_to_fasta(argv[1], nullptr) ? 1 : 0;
}
