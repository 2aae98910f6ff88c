// hs_host.h -- host-side (C++17) data model and glue of the MI355X HairSplitter hot path.
// The data-parallel work runs in the HIP kernels (hs_kernels.hip); what is declared here is the sequential
// glue the reference keeps between them, re-expressed over flat / dense arrays.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <utility>
#include <string>
#include <vector>
#include "../../include/hairsplitter_hip.h"

namespace hs {

void set_error(const std::string& msg);

// ---- stage 3 ---------------------------------------------------------------------------------
// The candidate columns of one contig (call_variants.cpp:525-536), position order: views into what the device handed over
// (the device extracts the columns, names their two leading codes in the reference's order and runs the spacing scan).
struct CandidateSet {
    int n = 0;
    const hs_colrec* rec = nullptr;  // [n] position, codes k0 / k1, counts
    const int64_t* off = nullptr;    // [n+1] into idx / code
    const int32_t* idx = nullptr;    // read indices (ascending inside a column)
    const uint8_t* code = nullptr;
};

struct ContigCvResult {
    float mean_distance = 0;
    float depth = 0;
    // diagnostics
    int n_candidates = 0, n_partitions = 0, n_final_partitions = 0;
};

struct CvContigState;   // per-contig state between the phases of the stage-3 glue (hs_host_cv.cpp)
CvContigState* cv_state_new();
void cv_state_free(CvContigState* st);
// The host part of keep_only_robust_variants in steps: loop A on the host (cv_phase_a_host) or imported from the device
// (k_loop_a -> cv_phase_a_import), then loop B (cv_phase_b); the final partitions leave for loops C / D on the device.
// read_start / read_end: [n_reads] reference interval [start, end) of every record of the contig (POS-1, POS-1 + reference span)
struct CvPartRecord { int32_t left, right, n_occ, n_corr, lo, hi, reach, pad; int64_t elem; };   // what k_loop_a_pack writes per partition
void cv_phase_begin(CvContigState& st, int n_reads, int n_candidates, float mean_distance, ContigCvResult& out);
void cv_phase_a_host(CvContigState& st, const CandidateSet& cs, const int32_t* read_start, const int32_t* read_end);
// bits: the contig's partitions, 3 W words each (present, plus, minus over the reads ranked by start position); cnt: N counters each
// (more | less << 16); rec[p].elem is not used here
void cv_phase_a_import(CvContigState& st, const int32_t* read_start, int n_parts, const CvPartRecord* rec, const uint64_t* bits, const int32_t* cnt);
void cv_phase_b(CvContigState& st, ContigCvResult& out, const int32_t* pair_table = nullptr);
// loop B with the pair distances from the device (k_partition_pair_distance): the partitions that pass loop B's gate, their dense
// arrays, the host's table of 3-sigma thresholds
int cv_loop_b_survivors(CvContigState& st);
void cv_export_survivors(const CvContigState& st, int8_t* state, int32_t* more, int32_t* less);
const std::vector<float>& cv_three_sigma_table();
int cv_final_partitions(const CvContigState& st);
// the final partitions' dense state arrays (n_reads bytes each) written at `state`, their offsets (state_base + ...) at state_off
void cv_export_partitions(const CvContigState& st, int8_t* state, int64_t state_base, int64_t* state_off);

// generate_msa's return value from the integer event counts of the pileup kernel (call_variants.cpp:67-68,434)
float mean_distance_from_counts(int64_t n_err, int64_t n_len);

// ---- stage 4 ---------------------------------------------------------------------------------
// A clustering window lives in its own LOCAL index space: node j = its j-th masked read (ascending read id). Everything the
// reference does with a window only ever touches those reads (see hs_kernels_cw.hip), so graphs, labels and every table are
// m-sized (m = masked reads of the window) instead of N-sized (N = reads of the contig).
struct SrWindowPlan {
    int start = 0, end = 0;
    bool has_snps = false;
    std::vector<int32_t> ids;        // reads the window reports (ascending): its masked reads, or, for a window without SNPs, the reads over its midpoint
    std::vector<int32_t> labels;     // final label of every entry of `ids` (every other read of the contig is -2)
    std::vector<int32_t> local_snps; // SNP indices whose allele seeds a local Chinese-Whispers run
    int final_lo = 0, final_hi = 0;  // [posstart, posend) handed to merge_wrongly_split_haplotypes
    int64_t row0 = -1;               // first row of the window in the graph set of the call (window-local CSR rows)
    bool final_graph_empty = false;  // finalize_clustering sees a graph that was never filled (separate_reads.cpp:1708 quirk)
    int col_a = -1, col_b = -1;      // first and last SNP column of the window (indices on the contig): its reads are those present at both
};

// ---- files -----------------------------------------------------------------------------------
// std::allocator that default-initialises: resize() of a multi-gigabyte byte vector does not memset it first (every element
// is written right afterwards, by many threads)
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
    template <class U, class... A> void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};

struct CvFileInput {
    // flattened parse_reads / parse_assembly / parse_SAM result
    std::vector<std::string> contig_names;
    std::vector<uint8_t, NoInitAlloc<uint8_t>> contig_seq;
    std::vector<int64_t> contig_off;
    std::vector<std::string> read_names;
    std::vector<uint8_t, NoInitAlloc<uint8_t>> read_seq;
    std::vector<int64_t> read_off;
    std::vector<int32_t> rec_read, rec_pos;
    std::vector<uint8_t> rec_strand;
    std::vector<int64_t> rec_cig_off;
    std::vector<uint32_t, NoInitAlloc<uint32_t>> cigar;
    std::vector<int32_t> contig_rec_off;
    std::vector<int32_t> rec_r0, rec_r1, rec_c0, rec_c1;   // the four coordinates of the READ line
    std::vector<uint8_t> contig_skip;                      // call_variants.cpp:1283
};
int load_cv_inputs(const std::string& gfa, const std::string& reads, const std::string& sam, bool amplicon,
                   CvFileInput& in, int n_threads = 1);
// the .gro consumer of the next stage (hs_gaf.cpp)
int gaf_from_files(const std::string& gfa, const std::string& reads, const std::string& sam, const std::string& gro, bool amplicon,
                   const std::string& out_gaf, int n_threads);
int gaf_from_labels(const std::string& gfa, const CvFileInput& in, int n_contigs, const int64_t* win_off, const int32_t* win_start,
                    const int32_t* win_end, const int64_t* label_off, const int32_t* labels, const uint8_t* contig_has_snps,
                    const std::string& out_gaf);

// the calling thread's persistent worker pool (hs_driver.cpp)
void hs_parallel_for(int n, int n_threads, const std::function<void(int)>& f);

}  // namespace hs
