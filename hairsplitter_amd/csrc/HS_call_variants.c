/* Drop-in executable: same argv, exit codes and output files as the reference's HS_call_variants; a thin host over
 * the C ABI. How the process starts and ends: hs_dropin_main.h. */
#include "hs_dropin_main.h"
int main(int argc, char** argv) { return hs_dropin_main2(hs_call_variants_main, hs_call_variants_epilogue, argc, argv); }
