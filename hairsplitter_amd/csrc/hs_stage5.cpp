// hs_stage5.cpp -- the two places where the reference's stage 5 calls its bundled edlib, batched on the A1 kernel
// (SURVEY.md §8f N4). Both take what racon / medaka hand back (external tools: not part of this build) and fix the ends:
//   hs_reattach_ends   tools.cpp:505-536  racon drops the ends of the sequence: the first / last 200 bases of the consensus are
//                      placed inside the first / last 300 bases of the backbone (HW alignment) and what the backbone has before /
//                      after that placement is attached again
//   hs_trim_polished   create_new_contigs.cpp:556-629  the polished sequence was built from the backbone piece plus overhangs:
//                      the overhangs are located on it through the alignment PATH of the piece's ends and cut off
// Host code around ONE batched call of hs_edlib_hw_align (two alignments per item). Alphabet: edlib compares BYTES (its default
// equality: 'N' only matches 'N', 'a' is not 'A'), the kernel compares codes 0..3; distance and path only depend on which
// positions are equal, so every pair gets its own bijection from the bytes it contains to the codes -- exact for any pair with at
// most four distinct bytes (ACGT in either case, ACG + N, ...). A pair with more than four (ACGT + N) is an error (HS_EFORMAT),
// not an approximation.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "hs_host.h"

namespace {

struct Dev {                     // device buffers through the raw-memory entry points of the C ABI
    std::vector<void*> ptrs;
    ~Dev() { for (void* p : ptrs) hs_free(p); }
    int alloc(void** p, size_t n) { const int rc = hs_malloc(p, n ? n : 16); if (!rc) ptrs.push_back(*p); return rc; }
};

struct Aln { int32_t dist = 0, start = 0, end = 0; std::vector<uint8_t> ops; };

// the pair's bytes -> codes 0..3 in order of first appearance; false when the pair has more than four distinct bytes
bool encode_pair(const std::string& q, const std::string& t, uint8_t* cq, uint8_t* ct) {
    int code[256];
    for (int i = 0; i < 256; ++i) code[i] = -1;
    int n = 0;
    auto enc = [&](const std::string& s, uint8_t* out) {
        for (size_t k = 0; k < s.size(); ++k) {
            const unsigned char b = (unsigned char)s[k];
            if (code[b] < 0) { if (n == 4) return false; code[b] = n++; }
            out[k] = (uint8_t)code[b];
        }
        return true;
    };
    return enc(q, cq) && enc(t, ct);
}

// edlibAlign(query, target, HW, k = -1, TASK_PATH) for every pair
int align_all(const std::vector<std::string>& q, const std::vector<std::string>& t, bool path, std::vector<Aln>& out) {
    const int n = (int)q.size();
    out.assign((size_t)n, Aln());
    if (n == 0) return HS_OK;
    std::vector<int64_t> qo((size_t)n + 1, 0), to((size_t)n + 1, 0), oo((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) { qo[(size_t)i + 1] = qo[(size_t)i] + (int64_t)q[(size_t)i].size(); to[(size_t)i + 1] = to[(size_t)i] + (int64_t)t[(size_t)i].size(); oo[(size_t)i + 1] = oo[(size_t)i] + (int64_t)(q[(size_t)i].size() + t[(size_t)i].size()); }
    std::vector<uint8_t> hq((size_t)qo.back() + 1), ht((size_t)to.back() + 1);
    for (int i = 0; i < n; ++i)
        if (!encode_pair(q[(size_t)i], t[(size_t)i], hq.data() + qo[(size_t)i], ht.data() + to[(size_t)i])) {
            hs::set_error("stage 5 alignment " + std::to_string(i) + ": more than four distinct bytes in one query / target pair (edlib compares bytes; this path has four codes)");
            return HS_EFORMAT;
        }
    Dev dev;
    void *dq, *dt, *dd, *ds, *de, *dl, *dops = nullptr;
    if (int rc = dev.alloc(&dq, hq.size())) return rc;
    if (int rc = dev.alloc(&dt, ht.size())) return rc;
    if (int rc = dev.alloc(&dd, (size_t)n * 4)) return rc;
    if (int rc = dev.alloc(&ds, (size_t)n * 4)) return rc;
    if (int rc = dev.alloc(&de, (size_t)n * 4)) return rc;
    if (int rc = dev.alloc(&dl, (size_t)n * 4)) return rc;
    if (path) { if (int rc = dev.alloc(&dops, (size_t)oo.back() + 1)) return rc; }
    if (int rc = hs_memcpy_h2d(dq, hq.data(), hq.size())) return rc;
    if (int rc = hs_memcpy_h2d(dt, ht.data(), ht.size())) return rc;
    if (int rc = hs_edlib_hw_align((const uint8_t*)dq, qo.data(), (const uint8_t*)dt, to.data(), n, (int32_t*)dd, (int32_t*)ds, (int32_t*)de, (uint8_t*)dops,
                                   oo.data(), (int32_t*)dl, nullptr)) return rc;
    std::vector<int32_t> hd((size_t)n), hs_((size_t)n), he((size_t)n), hl((size_t)n, 0);
    if (int rc = hs_memcpy_d2h(hd.data(), dd, (size_t)n * 4)) return rc;
    if (int rc = hs_memcpy_d2h(hs_.data(), ds, (size_t)n * 4)) return rc;
    if (int rc = hs_memcpy_d2h(he.data(), de, (size_t)n * 4)) return rc;
    std::vector<uint8_t> hops;
    if (path) {
        if (int rc = hs_memcpy_d2h(hl.data(), dl, (size_t)n * 4)) return rc;
        hops.resize((size_t)oo.back() + 1);
        if (int rc = hs_memcpy_d2h(hops.data(), dops, hops.size())) return rc;
    }
    for (int i = 0; i < n; ++i) {
        Aln& a = out[(size_t)i];
        a.dist = hd[(size_t)i]; a.start = hs_[(size_t)i]; a.end = he[(size_t)i];
        if (path && hl[(size_t)i] > 0) a.ops.assign(hops.begin() + oo[(size_t)i], hops.begin() + oo[(size_t)i] + hl[(size_t)i]);
    }
    return HS_OK;
}

char** to_c_strings(const std::vector<std::string>& v) {
    char** out = (char**)std::malloc(std::max<size_t>(1, v.size()) * sizeof(char*));
    for (size_t i = 0; i < v.size(); ++i) { out[i] = (char*)std::malloc(v[i].size() + 1); std::memcpy(out[i], v[i].c_str(), v[i].size() + 1); }
    return out;
}

std::string substr_like_std(const std::string& s, long pos, long len) {   // std::string::substr with a length that went through size_t
    if (pos < 0 || (size_t)pos > s.size()) return std::string();           // (the reference would throw std::out_of_range)
    if (len < 0) return s.substr((size_t)pos);
    return s.substr((size_t)pos, (size_t)len);
}

}  // namespace

extern "C" int hs_reattach_ends(const char* const* backbone, const char* const* consensus, int32_t n, char*** out) {
    if (n < 0 || !out || (n && (!backbone || !consensus))) { hs::set_error("hs_reattach_ends: bad arguments"); return HS_EINVAL; }
    std::vector<std::string> q, t;
    for (int i = 0; i < n; ++i) {
        const std::string b = backbone[i], c = consensus[i];
        if (b.empty() || c.empty()) { hs::set_error("hs_reattach_ends: empty sequence (the reference returns the backbone before it gets here)"); return HS_EINVAL; }
        const size_t before = std::min<size_t>(300, b.size()), after = std::min<size_t>(200, c.size());      // tools.cpp:508-509
        q.push_back(c.substr(0, after)); t.push_back(b.substr(0, before));                                   // :512-516
        q.push_back(c.substr(c.size() - after, after)); t.push_back(b.substr(b.size() - before, before));    // :525-529
    }
    std::vector<Aln> al;
    if (int rc = align_all(q, t, false, al)) return rc;
    std::vector<std::string> res((size_t)n);
    for (int i = 0; i < n; ++i) {
        const std::string& bs = t[(size_t)2 * i]; const std::string& be = t[(size_t)2 * i + 1];
        const int start_pos = al[(size_t)2 * i].start;                                                       // :519-520
        const int end_pos = al[(size_t)2 * i + 1].end + 1;                                                   // :532-533
        res[(size_t)i] = substr_like_std(bs, 0, start_pos) + consensus[i] + substr_like_std(be, end_pos, (long)be.size() - end_pos);
    }
    *out = to_c_strings(res);
    return HS_OK;
}

extern "C" int hs_trim_polished(const char* const* to_polish, const char* const* newcontig, const int32_t* overhang_left, const int32_t* overhang_right,
                                int32_t n, char*** out) {
    if (n < 0 || !out || (n && (!to_polish || !newcontig || !overhang_left || !overhang_right))) { hs::set_error("hs_trim_polished: bad arguments"); return HS_EINVAL; }
    std::vector<std::string> q, t;
    std::vector<int> begin_of_end((size_t)n);
    for (int i = 0; i < n; ++i) {
        const std::string tp = to_polish[i], nc = newcontig[i];
        if (tp.empty() || nc.empty()) { hs::set_error("hs_trim_polished: empty sequence"); return HS_EINVAL; }
        q.push_back(tp.substr(0, (size_t)std::max(300, overhang_left[i] * 2))); t.push_back(nc);             // create_new_contigs.cpp:559
        const int boe = std::max(0, std::min((int)tp.size() - overhang_right[i] * 2, (int)tp.size() - 300)); // :594
        begin_of_end[(size_t)i] = boe;
        q.push_back(tp.substr((size_t)boe, tp.size() - (size_t)boe)); t.push_back(nc);                      // :595
    }
    std::vector<Aln> al;
    if (int rc = align_all(q, t, true, al)) return rc;
    std::vector<std::string> res((size_t)n);
    for (int i = 0; i < n; ++i) {
        const std::string tp = to_polish[i], nc = newcontig[i];
        const Aln& a = al[(size_t)2 * i]; const Aln& b = al[(size_t)2 * i + 1];
        int on_tp = 0, on_nc = a.start, pos_start = 0;                                                       // :566-568
        for (uint8_t op : a.ops) {                                                                           // :571-586 ('=' / 'X' / 'M' both, 'D' contig, 'I' read)
            if (op == 0 || op == 3) { on_tp++; on_nc++; } else if (op == 2) on_nc++; else if (op == 1) on_tp++;
            if (on_tp == overhang_left[i]) { pos_start = on_nc; break; }
        }
        if (a.dist > 0.3 * (double)q[(size_t)2 * i].size()) pos_start = 0;                                   // :587-590
        on_tp = begin_of_end[(size_t)i]; on_nc = b.start;
        int pos_end = 0;
        const size_t stop_at = tp.size() - (size_t)overhang_right[i] - 1;                                    // (size_t arithmetic as in the reference, :616)
        for (uint8_t op : b.ops) {
            if (op == 0 || op == 3) { on_tp++; on_nc++; } else if (op == 2) on_nc++; else if (op == 1) on_tp++;
            if ((size_t)on_tp == stop_at) { pos_end = on_nc; break; }
        }
        if (b.dist > 0.3 * (double)q[(size_t)2 * i + 1].size()) pos_end = (int)nc.size();                    // :622-625
        res[(size_t)i] = substr_like_std(nc, pos_start, std::min(pos_end - pos_start + 1, (int)nc.size() - pos_start));   // :627
    }
    *out = to_c_strings(res);
    return HS_OK;
}

extern "C" void hs_free_strings(char** s, int32_t n) {
    if (!s) return;
    for (int i = 0; i < n; ++i) std::free(s[i]);
    std::free(s);
}
