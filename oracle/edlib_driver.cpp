// Test-infrastructure driver (build-owned): calls the bundled edlib of the reference
// (/root/reference/src/edlib/include/edlib.h:242-246 edlibAlign) on (query,target,mode) triples read
// from stdin and prints editDistance + first end location. Used only to generate golden vectors for
// the Myers bit-vector HIP kernel; linked against the reference's edlib.cpp where it lies.
// stdin lines: <mode NW|HW|SHW> <k> <query> <target>      -> "<distance> <numLocations> <start> <end>"
//              HWPATH <k> <query> <target>                 -> "<distance> <start> <end> <extended cigar or *>": what the stage-5 call
//              sites ask for (edlibNewAlignConfig(k, EDLIB_MODE_HW, EDLIB_TASK_PATH, NULL, 0) + edlibAlignmentToCigar(.., EDLIB_CIGAR_EXTENDED),
//              create_new_contigs.cpp:560-564, tools.cpp:515-534)
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include "edlib.h"
int main() {
    std::string mode, q, t; int k;
    while (std::cin >> mode >> k >> q >> t) {
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
