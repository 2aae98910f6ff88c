// Test-infrastructure driver (build-owned): calls the bundled edlib of the reference
// (/root/reference/src/edlib/include/edlib.h:242-246 edlibAlign) on (query,target,mode) triples read
// from stdin and prints editDistance + first end location. Used only to generate golden vectors for
// the Myers bit-vector HIP kernel; linked against the reference's edlib.cpp where it lies.
// stdin lines: <mode NW|HW|SHW> <k> <query> <target>      -> "<distance> <numLocations> <start> <end>"
//              HWPATH <k> <query> <target>                 -> "<distance> <start> <end> <extended cigar or *>": what the stage-5 call
//              sites ask for (edlibNewAlignConfig(k, EDLIB_MODE_HW, EDLIB_TASK_PATH, NULL, 0) + edlibAlignmentToCigar(.., EDLIB_CIGAR_EXTENDED),
//              create_new_contigs.cpp:560-564, tools.cpp:515-534)
//              REATTACH 0 <backbone> <consensus>          -> the sequence tools.cpp:505-536 forms (restated here around the reference's edlib)
//              TRIM <overhangLeft>,<overhangRight> <toPolish> <newcontig> -> the sequence create_new_contigs.cpp:556-629 forms
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <string>
#include "edlib.h"
// one move per character, as convert_cigar(edlibAlignmentToCigar(.., EDLIB_CIGAR_EXTENDED)) yields (tools.cpp:27-57)
static std::string moves(const EdlibAlignResult& r) {
    std::string s;
    for (int i = 0; i < r.alignmentLength; ++i) s += "=IDX"[r.alignment[i]];
    return s;
}
static EdlibAlignResult hw_path(const std::string& q, const std::string& t) {
    return edlibAlign(q.c_str(), (int)q.size(), t.c_str(), (int)t.size(), edlibNewAlignConfig(-1, EDLIB_MODE_HW, EDLIB_TASK_PATH, NULL, 0));
}
// tools.cpp:505-536
static std::string reattach(const std::string& backbone, const std::string& consensus) {
    auto before_size = std::min(size_t(300), backbone.size());
    auto after_size = std::min(size_t(200), consensus.size());
    std::string before_start = backbone.substr(0, before_size), after_start = consensus.substr(0, after_size);
    EdlibAlignResult result = hw_path(after_start, before_start);
    int start_pos = result.startLocations[0];
    std::string additional_start_seq = before_start.substr(0, start_pos);
    edlibFreeAlignResult(result);
    std::string before_end = backbone.substr(backbone.size() - before_size, before_size), after_end = consensus.substr(consensus.size() - after_size, after_size);
    result = hw_path(after_end, before_end);
    int end_pos = result.endLocations[0] + 1;
    std::string additional_end_seq = before_end.substr(end_pos, before_end.size() - end_pos);
    edlibFreeAlignResult(result);
    return additional_start_seq + consensus + additional_end_seq;
}
// create_new_contigs.cpp:556-629
static std::string trim(const std::string& toPolish, std::string newcontig, int overhangLeft, int overhangRight) {
    std::string toPolishStart = toPolish.substr(0, std::max(300, overhangLeft * 2));
    EdlibAlignResult result = hw_path(toPolishStart, newcontig);
    std::string cigar = moves(result);
    int posOnToPolish = 0, posOnNewContig = result.startLocations[0], posStartOnNewContig = 0;
    for (char c : cigar) {
        if (c == 'M' || c == 'X' || c == '=') { posOnToPolish++; posOnNewContig++; }
        else if (c == 'D') posOnNewContig++;
        else if (c == 'I') posOnToPolish++;
        if (posOnToPolish == overhangLeft) { posStartOnNewContig = posOnNewContig; break; }
    }
    if (result.editDistance > 0.3 * toPolishStart.size()) posStartOnNewContig = 0;
    edlibFreeAlignResult(result);
    int beginning_of_end = std::max(0, std::min(int(toPolish.size()) - overhangRight * 2, int(toPolish.size()) - 300));
    std::string toPolishEnd = toPolish.substr(beginning_of_end, int(toPolish.size()) - beginning_of_end);
    result = hw_path(toPolishEnd, newcontig);
    cigar = moves(result);
    posOnToPolish = beginning_of_end; posOnNewContig = result.startLocations[0];
    int posEndOnNewContig = 0;
    for (char c : cigar) {
        if (c == 'M' || c == 'X' || c == '=') { posOnToPolish++; posOnNewContig++; }
        else if (c == 'D') posOnNewContig++;
        else if (c == 'I') posOnToPolish++;
        if (posOnToPolish == toPolish.size() - overhangRight - 1) { posEndOnNewContig = posOnNewContig; break; }
    }
    if (result.editDistance > 0.3 * toPolishEnd.size()) posEndOnNewContig = newcontig.size();
    newcontig = newcontig.substr(posStartOnNewContig, std::min(posEndOnNewContig - posStartOnNewContig + 1, int(newcontig.size()) - posStartOnNewContig));
    edlibFreeAlignResult(result);
    return newcontig;
}

int main() {
    std::string mode, q, t, kk;
    while (std::cin >> mode >> kk >> q >> t) {
        if (mode == "REATTACH") { std::printf("%s\n", reattach(q, t).c_str()); continue; }
        if (mode == "TRIM") {
            const size_t comma = kk.find(',');
            std::printf("%s\n", trim(q, t, std::atoi(kk.substr(0, comma).c_str()), std::atoi(kk.substr(comma + 1).c_str())).c_str());
            continue;
        }
        const int k = std::atoi(kk.c_str());
        if (mode == "HWPATH") {
            EdlibAlignResult r = edlibAlign(q == "-" ? "" : q.c_str(), q == "-" ? 0 : (int)q.size(), t == "-" ? "" : t.c_str(), t == "-" ? 0 : (int)t.size(),
                                            edlibNewAlignConfig(k, EDLIB_MODE_HW, EDLIB_TASK_PATH, NULL, 0));
            char* cig = (r.alignment && r.alignmentLength > 0) ? edlibAlignmentToCigar(r.alignment, r.alignmentLength, EDLIB_CIGAR_EXTENDED) : nullptr;
            std::printf("%d %d %d %s\n", r.editDistance, (r.numLocations > 0 && r.startLocations) ? r.startLocations[0] : -1,
                        (r.numLocations > 0 && r.endLocations) ? r.endLocations[0] : -1, cig ? cig : "*");
            if (cig) free(cig);
            edlibFreeAlignResult(r);
            continue;
        }
        EdlibAlignMode m = EDLIB_MODE_NW;
        if (mode == "HW") m = EDLIB_MODE_HW; else if (mode == "SHW") m = EDLIB_MODE_SHW;
        EdlibAlignResult r = edlibAlign(q.c_str(), (int)q.size(), t.c_str(), (int)t.size(),
                                        edlibNewAlignConfig(k, m, EDLIB_TASK_LOC, NULL, 0));
        int endloc = (r.numLocations > 0 && r.endLocations) ? r.endLocations[0] : -1;
        int startloc = (r.numLocations > 0 && r.startLocations) ? r.startLocations[0] : -1;
        std::printf("%d %d %d %d\n", r.editDistance, r.numLocations, startloc, endloc);
        edlibFreeAlignResult(r);
    }
    return 0;
}
