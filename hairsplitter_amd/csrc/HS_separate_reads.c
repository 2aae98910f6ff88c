/* Drop-in replacement of the reference's HS_separate_reads executable (src/CMakeLists.txt:105-114). */
#include "../../include/hairsplitter_hip.h"
int main(int argc, char** argv) { return hs_separate_reads_main(argc, argv); }
