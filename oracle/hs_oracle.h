// ORACLE (test infrastructure, CPU only). A from-scratch CPU restatement of the reference's hot path
// (HS_call_variants -> HS_separate_reads). It is the checker for the HIP product path; nothing in the
// product links, imports or executes it. Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use it. Parity of this restatement itself is pinned against the compiled
// reference (oracle/_ref, seeded) through tests/golden/* (see oracle/gen_goldens.py).
//
// Every function cites the reference file:line (relative to /root/reference/src) it follows.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <utility>
#include "hs_oracle_rh.h"

namespace hso {

// ---- data model (Partition.h:8-14, read.h:12-24) -------------------------------------------------
struct Column {
    int pos = 0;
    std::vector<unsigned int> readIdxs;
    std::vector<unsigned char> content;
    unsigned char ref_base = 0;
    unsigned char second_base = 0;
};

struct Record {            // one kept SAM line == one "read" n of a contig (input_output.cpp:493-521)
    long read = -1;        // index into Dataset::read_seq
    std::string read_name;
    int position_1_1 = 0, position_1_2 = 0, position_2_1 = 0, position_2_2 = 0;
    bool strand = true;
    std::string cigar;
};

struct Contig {
    std::string name;
    std::string seq;       // already passed through the 2-bit filter (non-ACG -> T, sequence.cpp:13-23)
    std::vector<Record> recs;
    float depth = -1;
};

struct Dataset {
    std::vector<std::string> read_names;
    std::vector<long> read_len;
    std::vector<long> read_offset;      // byte offset of the sequence line (input_output.cpp:78)
    std::vector<std::string> read_seq;  // loaded lazily, 2-bit filtered
    std::vector<Contig> contigs;
};

// ---- I/O (input_output.cpp:39-569, call_variants.cpp:1174-1213) ----------------------------------
std::string convert_cigar(const std::string& cigar);                                   // tools.cpp:27-57
std::string two_bit_filter(const std::string& s);                                      // sequence.cpp:13-52
std::string reverse_complement(const std::string& s);                                  // sequence.cpp:54-65
void parse_inputs(const std::string& gfa, const std::string& reads, const std::string& sam,
                  bool amplicon, Dataset& ds);

// ---- stage 3 (call_variants.cpp) ----------------------------------------------------------------
struct MsaResult {
    std::vector<Column> cols;
    std::string newref;
    float meanDistance = 0;
    // integer side-products used by kernel-level parity tests
    std::vector<int> q_end;          // per record: final indexQuery (call_variants.cpp:354)
    std::vector<long> n_err, n_len;  // per record: +1 events of totalDistance / totalLengthOfAlignment
};
MsaResult generate_msa(const Contig& c, const std::vector<std::string>& read_seq);     // :50-437

struct CallResult {
    std::vector<Column> suspicious;      // candidates
    std::vector<Column> automatic;       // automatic_snps
    float depth = 0;
    // per-position top-3 (after the stable/unstable std::sort of the reference)
    std::vector<unsigned char> k0, k1;
    std::vector<int> c0, c1, c2;
};
CallResult call_variants(std::vector<Column>& cols, const std::string& ref, float meanError,
                         float automatic_snp_threshold);                                // :447-567

struct DistRes {
    int n00 = 0, n01 = 0, n10 = 0, n11 = 0, solid11 = 0, solid10 = 0, solid01 = 0, solid00 = 0;
    short phased = 1;
    bool augmented = true;
    unsigned char secondBase = ' ';
    Column partition_to_augment;
};

class Partition {                                                                        // Partition.cpp
public:
    Partition() {}
    Partition(const Column& snp, int pos, unsigned char ref_base);                      // :32-83
    void augmentPartition(const Column& supp, int pos);                                 // :243-397
    void mergePartition(const Partition& p, short phased);                              // :401-537
    bool isInformative(bool lastReadBiased, float meanError) const;                     // :141-179
    float isSignificant(int total_columns) const;                                       // :197-233
    float compute_conf();                                                               // :716-732
    std::vector<float> getConfidence() const;                                           // :811-827
    int number() const { return numberOfOccurences; }
    int get_left() const { return pos_left; }
    int get_right() const { return pos_right; }

    std::vector<int> readIdx;
    std::vector<short> mostFrequentBases;
    std::vector<int> moreFrequence, lessFrequence;
    int numberOfOccurences = 0;
    float conf_score = 0;
    int pos_left = -1, pos_right = -1;
    int number_of_correlating_snps = 0;
};

DistRes distance(const Partition& p, const Column& col, char ref_base);                 // :778-967
DistRes distance(const Partition& a, const Partition& b, int threshold_p);              // :977-1127
float computeChiSquare(const DistRes& d);                                               // :1135-1163
void keep_only_robust_variants(std::vector<Column>& msa, std::vector<Column>& snps_in,
                               std::vector<Column>& snps_out, float mean_error,
                               std::vector<Partition>& parts);                          // :577-768

struct ContigVariants {
    std::vector<Column> merged;      // what goes to the .col
    float meanDistance = 0;
    float depth = 0;
    std::vector<Partition> partitions;
    std::vector<int> candidate_pos, automatic_pos, filtered_pos;
};
ContigVariants call_variants_on_contig(const Contig& c, const std::vector<std::string>& read_seq,
                                       float automatic_snp_threshold);                  // :1280-1367

int run_call_variants(int argc, char** argv);                                           // main :1215-1385

// ---- stage 4 (separate_reads.cpp, cluster_graph.cpp) --------------------------------------------
struct ColContig {
    std::string contig_line;                       // whole CONTIG line (separate_reads.cpp:70-71)
    long length = 0;
    double coverage = 0;
    std::vector<std::string> read_lines;
    std::vector<std::pair<int, int>> readLimits;
    std::vector<Column> snps;
};
std::vector<ColContig> parse_column_file(const std::string& file, int max_coverage,
                                         float rarest_strain_abundance);                // :46-190

struct Window { int start, end; std::vector<int> labels; };
// test taps (tests only): when g_window_taps is set, separate_reads_on_contig leaves, for every window that builds a read graph, its reads,
// the labels of every per-SNP Chinese-Whispers run (separate_reads.cpp:1674-1705) and of the run behind finalize_clustering's small-cluster
// filter (:924-970), both restricted to the window's reads
struct WindowTap { int start = 0; std::vector<int> mask_ids; std::vector<int> run_snp; std::vector<std::vector<int>> runs; std::vector<int> third; };
extern thread_local std::vector<WindowTap>* g_window_taps;

// dense restatement of the two Eigen products (:374-433)
void list_similarities_and_differences(const std::vector<Column>& snps, int N,
                                       std::vector<int>& sim, std::vector<int>& diff);
// adjacency as sorted neighbour lists (Eigen col-major inner iteration == ascending row)
void create_read_graph_matrix(const std::vector<bool>& mask, const std::vector<int>& sim,
                              const std::vector<int>& diff, int N, float errorRate,
                              std::vector<std::vector<int>>& adj);                      // :706-828
void create_read_graph_low_memory(const std::vector<Column>& snps, const std::vector<bool>& mask,
                                  std::vector<std::vector<int>>& nl, float errorRate);  // :538-693
std::vector<int> shuffled_order(int n, unsigned seed);       // libstdc++ mt19937 + std::shuffle
std::vector<int> chinese_whispers(const std::vector<std::vector<int>>& adj, const std::vector<int>& init,
                                  const std::vector<bool>& mask, unsigned seed,
                                  int* sweeps_out = nullptr);                           // cluster_graph.cpp:152-310
void merge_close_clusters(const std::vector<std::vector<int>>& nl, const std::vector<std::vector<int>>& adj,
                          bool low_memory, std::vector<int>& clusters, const std::vector<bool>& mask,
                          unsigned seed);                                               // cluster_graph.cpp:402-501
std::vector<int> merge_wrongly_split_haplotypes(const std::vector<int>& clusteredReads,
                                                const std::vector<Column>& snps,
                                                const std::vector<std::vector<int>>& nl,
                                                const std::vector<std::vector<int>>& adj, bool low_memory,
                                                int posstart, int posend);              // :1007-1327
std::vector<Window> separate_reads_on_contig(const ColContig& c, int sizeOfWindow, float errorRate,
                                             bool low_memory, bool low_memory_now, int ploidy,
                                             unsigned seed);                            // :1508-1739
int choose_window_size(const std::vector<ColContig>& cs, bool amplicon, std::vector<float>* coverages); // :1466-1498
int run_separate_reads(int argc, char** argv);                                          // main :1398-1790

// ---- A1: edit distance (oracle for the Myers bit-vector kernel; edlib.h:36-62 modes) -------------
// mode 0 = NW (global), 1 = SHW (prefix: gaps at target end free), 2 = HW (infix)
int edit_distance(const unsigned char* q, int qn, const unsigned char* t, int tn, int mode, int* end_loc);

}  // namespace hso
