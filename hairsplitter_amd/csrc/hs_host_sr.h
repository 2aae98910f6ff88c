// hs_host_sr.h -- per-contig state of stage 4 (HS_separate_reads) between the device waves.
#pragma once
#include <cstdint>
#include <vector>
#include "hs_host.h"

namespace hs {

struct SrWindowPlanEx;

struct SrContigState {
    const hs_sr_contig* c = nullptr;
    int N = 0;
    int words = 0;
    bool low_memory_now = false;
    bool snp_pos_sorted = false;                    // SNP positions ascend (lets the per-window SNP range be a binary search)
    std::vector<SrGraph> graphs;
    int empty_graph = -1;
    std::vector<struct SrWindowPlan> windows;
    std::vector<int32_t> perm;                      // std::shuffle(mt19937(seed)) of 0..N-1
};

std::vector<int32_t> shuffled_order(int n, uint32_t seed);
void sr_plan_windows(SrContigState& st, int window_size, float error_rate, bool low_memory);
void sr_build_window_graph(SrContigState& st, int window, float error_rate);   // low-memory path only
// one row of create_read_graph_matrix in the reference's own way (std::sort + walk, separate_reads.cpp:745-815): used for
// the rows the device reports as depending on std::sort's arrangement of equal distances
void sr_pick_row_sorted(const int32_t* srow, const int32_t* drow, int N, int r1, const uint8_t* mask, float error_rate, std::vector<int>& picked);
// neighbour lists of the masked reads (device result) -> the window's CSR over all N reads
void sr_set_window_graph(SrContigState& st, int window, const int32_t* ids, int m, const int64_t* nbr_off, const int32_t* nbr);
void sr_finish_window(const SrContigState& st, SrWindowPlan& w, const int32_t* reclustered, bool low_memory);
bool sr_ploidy_init_labels(const SrContigState& st, const SrWindowPlan& w, int max_haplotypes, int32_t* out);
int32_t sr_window_size(const hs_sr_contig* cs, int n, bool amplicon);
bool sr_coverage_above_1000(const hs_sr_contig& c);

}  // namespace hs
