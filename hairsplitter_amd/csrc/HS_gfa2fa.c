#include "../../include/hairsplitter_hip.h"
int main(int argc, char** argv) { return hs_gfa2fa_main(argc, argv); }
