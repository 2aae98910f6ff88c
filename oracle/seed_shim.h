// Build-owned shim (NOT reference code): pins std::random_device to a constant so the
// reference HS_separate_reads becomes deterministic (it re-seeds std::mt19937 from
// std::random_device on every Chinese-Whispers sweep, cluster_graph.cpp:175-177,256-258,430-432).
// Force-included with `g++ -include seed_shim.h`; no reference source is modified.
#pragma once
#include <random>
#ifndef HS_ORACLE_SEED
#define HS_ORACLE_SEED 12345u
#endif
namespace std {
struct hs_fixed_rd {
    using result_type = unsigned int;
    hs_fixed_rd() {}
    unsigned int operator()() { return HS_ORACLE_SEED; }
};
}
#define random_device hs_fixed_rd
