/* Drop-in replacement of the reference's HS_call_variants executable (src/CMakeLists.txt:96-103). */
#include "../../include/hairsplitter_hip.h"
int main(int argc, char** argv) { return hs_call_variants_main(argc, argv); }
