#include "../../include/hairsplitter_hip.h"
int main(int argc, char** argv) { return hs_cut_gfa_main(argc, argv); }
