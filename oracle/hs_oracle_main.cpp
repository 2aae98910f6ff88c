// ORACLE (test infrastructure, CPU only): command-line front end of the CPU restatement.
//   hs_oracle call_variants  <the 11 positional arguments of HS_call_variants>
//   hs_oracle separate_reads <the 9 positional arguments of HS_separate_reads>
#include <cstring>
#include <iostream>
#include "hs_oracle.h"
int main(int argc, char** argv) {
    if (argc < 2) { std::cerr << "usage: hs_oracle call_variants|separate_reads ...\n"; return 2; }
    if (!std::strcmp(argv[1], "call_variants")) return hso::run_call_variants(argc - 1, argv + 1);
    if (!std::strcmp(argv[1], "separate_reads")) return hso::run_separate_reads(argc - 1, argv + 1);
    std::cerr << "unknown subcommand " << argv[1] << "\n";
    return 2;
}
