// Test-infrastructure driver (build-owned): calls the bundled edlib of the reference
// (/root/reference/src/edlib/include/edlib.h:242-246 edlibAlign) on (query,target,mode) triples read
// from stdin and prints editDistance + first end location. Used only to generate golden vectors for
// the Myers bit-vector HIP kernel; linked against the reference's edlib.cpp where it lies.
// stdin lines: <mode NW|HW|SHW> <k> <query> <target>
#include <cstdio>
#include <iostream>
#include <string>
#include "edlib.h"
int main() {
    std::string mode, q, t; int k;
    while (std::cin >> mode >> k >> q >> t) {
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
