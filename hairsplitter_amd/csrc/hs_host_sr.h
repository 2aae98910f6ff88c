// hs_host_sr.h -- per-contig state of stage 4 (HS_separate_reads) between the device passes.
#pragma once
#include <cstdint>
#include <vector>
#include "hs_host.h"

namespace hs {

struct SrContigState {
    const hs_sr_contig* c = nullptr;
    int N = 0;
    int words = 0;
    bool low_memory_now = false;
    bool snp_pos_sorted = false;                    // SNP positions ascend (lets the per-window SNP range be a binary search)
    std::vector<struct SrWindowPlan> windows;
    std::vector<int32_t> perm;                      // std::shuffle(mt19937(seed)) of 0..N-1
    std::vector<int32_t> rank;                      // rank[perm[k]] = k
    uint32_t perm_seed = 0; int perm_n = -1;        // what perm / rank were made for (a state kept from call to call skips the shuffle)
    // the reads ranked by start position (ties by index): the row order of the contig's sim / diff matrices (K5 then only computes the
    // blocks of reads that lie next to each other); kept from call to call while the starts are the same
    std::vector<int32_t> pos_rank, pos_orig, pos_key;
    std::vector<struct SrWindowPlan> spare;         // window plans of the previous call: their vectors' storage is used again
    struct SrWindowPlan take_window();              // an empty plan, recycled if one is there
};

// a window's read graph in its local index space: CSR rows [0, m), neighbours = local ids (ascending)
struct SrLocalGraph {
    const int64_t* off = nullptr;    // m + 1 absolute offsets into nbr
    const int32_t* nbr = nullptr;
    int64_t begin(int j) const { return off[j]; }
    int64_t end(int j) const { return off[j + 1]; }
};

std::vector<int32_t> shuffled_order(int n, uint32_t seed);
// with_reads = false: the windows' read lists (ids) are left empty -- the columns are not here; col_a / col_b say which two make them
void sr_plan_windows(SrContigState& st, int window_size, float error_rate, bool low_memory, bool with_reads = true);
// one row of that builder from the window-local counts the device formed (the rows it hands back: a NaN distance, or a run of equal
// distances at the cut-off)
void sr_pick_row_sorted_low_memory(const int32_t* srow, const int32_t* drow, const int32_t* ids, int m, int i, int N, const uint8_t* mask, float error_rate,
                                   std::vector<int>& picked);
// create_read_graph_low_memory (separate_reads.cpp:538-693) for one window, in local index space: deg/nbr lists per node
void sr_build_window_graph_low_memory(const SrContigState& st, const SrWindowPlan& w, float error_rate, std::vector<std::vector<int32_t>>& lists);
// one row of create_read_graph_matrix in the reference's own way (std::sort + walk, separate_reads.cpp:745-815): used for
// the rows the device reports as depending on std::sort's arrangement of equal distances
void sr_pick_row_sorted(const int32_t* srow, const int32_t* drow, int N, int r1, const uint8_t* mask, float error_rate, std::vector<int>& picked);
// the tail of finalize_clustering (:973-993) on the host, for the windows the device did not finish: `reclustered` = the
// m labels the third Chinese-Whispers run left; result in w.labels
void sr_finish_window(const SrContigState& st, SrWindowPlan& w, const int32_t* reclustered, const SrLocalGraph& g, bool low_memory);
// merge_haplotypes_to_fit_within_limit up to its re-clustering (:1341-1383): false when the limit is met already
bool sr_ploidy_init_labels(const SrWindowPlan& w, int max_haplotypes, int32_t* out);
int32_t sr_window_size(const hs_sr_contig* cs, int n, bool amplicon);
bool sr_coverage_above_1000(const hs_sr_contig& c);

}  // namespace hs
