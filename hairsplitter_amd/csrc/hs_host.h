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
// A candidate column as loop A reads it: the reads of the column as one bit set per distinct code, bit k = the read of RANK k in
// the order of the reads' start positions on the contig (ties by read index). The reads of a column all cover its position, so
// they sit in a few neighbouring 64-read words whatever the order of the records in the SAM file. Built on the device
// (k_cand_bits) from the packed candidate columns; 32 bytes + a block of 64-bit words per column.
struct CandBits {
    int32_t wlo;                // first word the column's reads lie in
    uint16_t n_words;           // words it spans (0: a column without entries)
    uint16_t n_slots;           // distinct codes, in the order their first read (ascending read index) brings them
    int32_t idx_min, idx_max;   // first / last read index of the column
    int32_t reach;              // largest (exclusive) end position of its reads
    int32_t n_entries;
    int64_t word_off;           // its block in the word array: any[n_words], slot bits [n_slots][n_words], the slots' codes 8 per word
};
static_assert(sizeof(CandBits) == 32, "CandBits layout");
inline int64_t cand_bits_block_words(int n_words, int n_slots) { return (int64_t)n_words * (n_slots + 1) + (n_slots + 7) / 8; }

// The candidate columns of one contig (call_variants.cpp:525-536), position order: views into what the device handed over
// (the device extracts the columns, names their two leading codes in the reference's order, runs the spacing scan and turns
// every candidate into bit sets). off / idx / code (the raw entries) are optional: only the cross-check of the test harness walks them.
struct CandidateSet {
    int n = 0;
    const hs_colrec* rec = nullptr;  // [n] position, codes k0 / k1, counts
    const CandBits* bits = nullptr;  // [n]
    const uint64_t* words = nullptr; // the blocks (CandBits::word_off)
    const int64_t* off = nullptr;    // [n+1] into idx / code
    const int32_t* idx = nullptr;    // read indices (ascending inside a column)
    const uint8_t* code = nullptr;
};
// The same from raw entries on the host: what k_cand_bits computes, restated (the test harness's device interface and the kernel's
// unit test use it; the product gets the bit sets from the device). rank_of: read -> rank, read_end: read -> exclusive end position.
// Appends the blocks to `words`, fills bits[0..n).
void cv_build_cand_bits(int n, const int64_t* off, const int32_t* idx, const uint8_t* code, const int32_t* rank_of, const int32_t* read_end,
                        CandBits* bits, std::vector<uint64_t>& words);
// reads of a contig ranked by start position (ties by index): rank_of[read] and orig_of[rank] (padded to a multiple of 64)
void cv_rank_reads(int n_reads, const int32_t* read_start, std::vector<int32_t>& rank_of, std::vector<int32_t>& orig_of);

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
// read_start: [n_reads] POS-1 of every record of the contig (the bit order of every bit set: ranks by start position)
struct CvPartRecord { int32_t left, right, n_occ, n_corr, lo, hi, reach, w0, w1, pad; int64_t word_off; };   // what k_loop_a_pack writes per partition (w0..w1: the words of the contig's ranked reads its reads lie in)
void cv_phase_begin(CvContigState& st, int n_reads, int n_candidates, float mean_distance, ContigCvResult& out);
void cv_phase_a_host(CvContigState& st, const CandidateSet& cs, const int32_t* read_start, const int32_t* rank_of = nullptr, const int32_t* orig_of = nullptr /* the contig's reads ranked already (cv_rank_reads' result), or NULL */);
// bits: the contig's partitions, 3 W words each (present, plus, minus over the reads ranked by start position); cnt: N counters each
// (more | less << 16); rec[p].elem is not used here
void cv_phase_a_import(CvContigState& st, const int32_t* read_start, int n_parts, const CvPartRecord* rec, const uint64_t* bits, const int32_t* cnt, const int32_t* rank_of = nullptr, const int32_t* orig_of = nullptr);
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
// every read and every contig of a job, coded 0..3, with their names (the PAF ingest aligns read segments against contig windows)
struct SeqSet {
    std::vector<std::string> read_names, contig_names;
    std::vector<uint8_t, NoInitAlloc<uint8_t>> read_seq, contig_seq;
    std::vector<int64_t> read_off, contig_off;
};
int load_sequences(const std::string& gfa, const std::string& reads, SeqSet& out, int n_threads = 1);
// CIGAR-less input (hs_realign.cpp): a PAF file becomes the SAM the path reads, every read segment aligned against its contig window
// on the device with edlib's HW path (A1)
struct RealignStats { int64_t n_lines = 0, n_aligned = 0, query_bases = 0; double ms_device = 0, ms_total = 0; };
int realign_paf_to_sam(const std::string& gfa, const std::string& reads, const std::string& paf, const std::string& out_sam, int n_threads, RealignStats* stats = nullptr);
int host_threads();
// the .gro consumer of the next stage (hs_gaf.cpp)
int gaf_from_files(const std::string& gfa, const std::string& reads, const std::string& sam, const std::string& gro, bool amplicon,
                   const std::string& out_gaf, int n_threads);
int gaf_from_labels(const std::string& gfa, const CvFileInput& in, int n_contigs, const int64_t* win_off, const int32_t* win_start,
                    const int32_t* win_end, const int64_t* label_off, const int32_t* labels, const uint8_t* contig_has_snps,
                    const std::string& out_gaf);

// the calling thread's persistent worker pool (hs_driver.cpp)
void hs_parallel_for(int n, int n_threads, const std::function<void(int)>& f);

}  // namespace hs
