// hs_driver.h -- the two stage drivers, written against a small device interface.
// The shipped library implements the interface with HIP kernels only (hs_capi.hip); there is no CPU
// implementation in the product. tests/harness implements it with the oracle so the host glue can be
// checked on a machine without a GPU (that harness is test infrastructure and never ships).
#pragma once
#include <cstdint>
#include <functional>
#include <vector>
#include "hs_host.h"
#include "hs_host_sr.h"

namespace hs {

struct CvMeta {                       // host-side description of a stage-3 batch
    int32_t n_contigs = 0, n_rec = 0;
    std::vector<int64_t> contig_off;      // [C+1]
    std::vector<int32_t> contig_rec_off;  // [C+1]
    std::vector<int64_t> pile_off;        // [NREC+1]
    std::vector<int32_t> rec_pos;         // [NREC] POS-1
    std::vector<int64_t> rec_refspan;     // [NREC] reference bases consumed by the CIGAR (unclipped)
    int64_t total_len = 0;
    std::vector<int32_t> ploidy;          // [C] or empty: ploidy of the contigs for the stage 3 -> 4 hand-over (0 = none)
    // optional, [NREC] each: the reads of every contig ranked by start position (ties by index) -- rank of a read on its contig, and the
    // read at a rank -- when the batch has them already (it ranks the reads once for the device); empty: loop A sorts per call
    std::vector<int32_t> rank_of, orig_of;
};

// K4 input: the final partitions of every contig of a range as dense per-read state arrays (2 = read not in the partition)
struct CvPartitionTest {
    std::vector<int32_t> contig_n_reads;    // [C]
    std::vector<int32_t> part_off;          // [C+1] range of partitions of each contig
    std::vector<int64_t> part_state_off;    // [sum F] offset of each partition's state array
    std::vector<int8_t> part_state;
};

// What the device hands to the host in the middle of stage 3: the candidate columns of a contig range (call_variants.cpp:525-536),
// contig by contig, position order inside a contig. The arrays belong to the implementation and stay valid until its next call.
struct CvCandidates {
    int64_t n_columns = 0, n_entries = 0;    // columns extracted for the range (second count >= 4: everything that can become a SNP) -- they stay with the implementation
    int64_t n_tie = 0, n_tie_big = 0;        // of those, columns whose two leading codes needed the reference's order of equal counts / std::sort beyond 16 keys
    std::vector<int32_t> contig_n_cand;      // [C]
    std::vector<float> contig_mean_distance; // [C] when the implementation brought up the range's pileup itself (min_reads passed empty), else empty
    int64_t n_cand = 0;
    const hs_colrec* rec = nullptr;          // [n_cand]
    const int32_t* col = nullptr;            // [n_cand] index of the candidate in the implementation's column list
    const int64_t* off = nullptr;            // [n_cand + 1]
    const int32_t* idx = nullptr;            // read indices (ascending inside a column); idx / code may be null when `bits` is given
    const uint8_t* code = nullptr;
    // the same columns as bit sets over the reads ranked by start position (hs::CandBits, hs_host.h): what loop A reads.
    // word_off counts from `words`; bits[k] belongs to rec[k]
    const CandBits* bits = nullptr;          // [n_cand]
    const uint64_t* words = nullptr;
    std::vector<uint8_t> contig_on_device;   // [C] != 0: loop A of this contig runs on the device (collect_partitions); empty: none does
};
// ... and at its end: the SNPs (call_variants.cpp:1335-1352), same layout; idx / code only when they were asked for
struct CvSnpSet {
    std::vector<int32_t> contig_n_snp;       // [C]
    int64_t n_snp = 0, n_entries = 0;
    const hs_colrec* rec = nullptr;
    const int64_t* off = nullptr;            // [n_snp + 1]
    const int32_t* idx = nullptr;
    const uint8_t* code = nullptr;
};

// Loop A of keep_only_robust_variants on the device (k_loop_a), for the contigs of the range its tables hold: the partitions of every
// such contig as the host imports them (cv_phase_a_import) -- a record, and for the words [w0, w1] of the contig's reads (ranked by
// start position) that the partition's reads lie in: present / state +1 / state -1 (3 x span words at 3 x word_off of `bits`) and one
// counter per read of those words (more | less << 16; 64 x span at 64 x word_off of `cnt`)
struct CvLoopAResult {
    std::vector<uint8_t> on_device;            // [C] != 0: the contig was the device's (empty: none was)
    std::vector<int32_t> failed;               // [C] != 0: the device gave up on this contig (its tables do not hold it), the host does it
    std::vector<int64_t> part_base;            // [C+1] first partition of every contig in `rec`
    const CvPartRecord* rec = nullptr;
    const uint64_t* bits = nullptr;
    const int32_t* cnt = nullptr;
};

// The device side of stage 3. One object serves one range of contigs at a time: pileup() once per batch, then per range
// extract_candidates() -> [collect_partitions()] -> finish_columns().
struct CvDeviceOps {
    virtual ~CvDeviceOps() {}
    // K0 + K1 over the whole batch: per-record {q_end, n_err, n_len, n_events}; k_ms = {pileup, -, -, cigar scan}
    virtual int pileup(std::vector<int32_t>& rec_stats, float k_ms[4]) = 0;
    // K2 -> K3 -> K3b -> V1 for the contigs [c0, c1): every position whose second count is >= 4 becomes a column (it stays with the
    // implementation), its two leading codes are named in the reference's order, the candidates are chosen (min_reads[c - c0] = 3 or 5,
    // call_variants.cpp:463-466) and packed for the host. k_ms = {column statistics, gather, top-3 + candidates}
    // want_entries = false: only the counts come back (n_columns, contig_n_cand, n_cand ...); the candidates stay with the
    // implementation for robust_partitions() and can still be fetched (fetch_candidates) if the host has to walk some after all
    virtual int extract_candidates(int c0, int c1, const std::vector<int32_t>& min_reads, float automatic_snp_threshold, CvCandidates& out,
                                   float k_ms[3], bool want_entries = true) = 0;
    virtual int fetch_candidates(CvCandidates& out) { (void)out; return -1; }
    // K4 (loops C and D of keep_only_robust_variants on the extracted columns against the final partitions) and the merge of the
    // automatic and the filtered SNPs; want_entries = false leaves idx / code of the result null (the SNP columns stay with the
    // implementation for stage 4: take_snp_columns)
    virtual int finish_columns(const CvPartitionTest& t, bool want_entries, CvSnpSet& out, float* k_ms) = 0;
    // An implementation may bring the entries of the SNP columns (want_entries) to the host BEHIND the call -- idx / code of the result
    // null, the transfer on its way beside whatever the caller does next (stage 4 reads the columns on the device) -- and hand them over
    // here: waits for the transfer, fills idx / code (valid while the implementation lives). Default: nothing was deferred.
    virtual int late_entries(CvSnpSet& out) { (void)out; return 0; }
    // Loop A of keep_only_robust_variants (call_variants.cpp:590-638) on the candidate columns, contig by contig.
    // Optional: an implementation without it leaves the loop to the host (cv_phase_a_host). The result arrays are owned by
    // the implementation and stay valid until the next call.
    // V5 distance(Partition, Partition, 2) for a list of pairs (loop B, opt-in): dense arrays of the partitions one after the other
    // (state 2 = absent), part_off / part_n per partition, out = 8 ints per pair {n00, n01, n10, n11, phased, augmented, valid, comparable}
    virtual bool has_partition_pairs() const { return false; }
    virtual int partition_pairs(const std::vector<int8_t>& state, const std::vector<int32_t>& more, const std::vector<int32_t>& less, const std::vector<int64_t>& part_off,
                                const std::vector<int32_t>& part_n, const std::vector<int32_t>& pair_a, const std::vector<int32_t>& pair_b, const std::vector<float>& sigma3,
                                std::vector<int32_t>& out) { (void)state; (void)more; (void)less; (void)part_off; (void)part_n; (void)pair_a; (void)pair_b; (void)sigma3; (void)out; return -1; }
    // loop A on the device: an implementation that has it queues it inside extract_candidates() for the contigs it marks in
    // CvCandidates::contig_on_device (their candidates come without bit sets) and hands the partitions over here, once the host has
    // walked the other contigs; fetch_candidates() brings the bit sets of every candidate after all (a contig the device gave up on)
    virtual int collect_partitions(CvLoopAResult& out) { out = CvLoopAResult(); return 0; }
};

// result of the whole-batch streaming pass (K0 + K1): the per-record counters
struct CvSelection {
    std::vector<int32_t> rec_stats;
    float k_ms[4] = {0, 0, 0, 0};          // pileup, -, -, cigar_scan
    double t_device_ms = 0, t_host_ms = 0;
};
int cv_pileup(CvDeviceOps& dev, const CvMeta& meta, CvSelection& sel);
// stage 3 for the contigs [c0, c1) on top of the pileup; `resident` (optional): the SNP columns are not downloaded into the result
// (col_idx / col_code stay null) because the caller hands them to stage 4 on the device
// rec_stats empty: the device interface brings up the range's share of the pileup itself and reports the contigs' mean distances with
// the candidates (CvCandidates::contig_mean_distance); on_mean_distance (optional) is called with them [c1 - c0] as soon as they are known
int cv_run_range(CvDeviceOps& dev, const CvMeta& meta, const std::vector<int32_t>& rec_stats, int c0, int c1, float automatic_snp_threshold, int n_threads,
                 hs_cv_result** out, bool resident = false, const std::function<void(const float*)>* on_mean_distance = nullptr);
int cv_run(CvDeviceOps& dev, const CvMeta& meta, float automatic_snp_threshold, int n_threads, hs_cv_result** out);
hs_cv_result* cv_concat_results(hs_cv_result* a, hs_cv_result* b);   // two consecutive contig ranges (both consumed)
// the entries of the SNP columns of a result whose device interface deferred them (CvDeviceOps::late_entries): col_idx / col_code filled in
int cv_attach_entries(CvDeviceOps& dev, hs_cv_result* r, int n_threads, bool borrow = false);      // (borrow: the result points into the implementation's block, which the caller keeps alive)

// Every clustering window of a stage-4 call, each in its own LOCAL index space: node j of window w is the read
// mask_ids[win_row0[w] + j] (ascending inside a window). "Row" = (window, node); the read graphs of the call are ONE CSR
// over the rows whose neighbours are local ids. Windows [0, n_dev_windows) take their graph from the sim / diff matrices of
// the last simdiff_columns() call (K6); the others (contigs on the low-memory path) bring theirs in host_off / host_nbr.
struct SrWindowSet {
    std::vector<int32_t> win_contig;       // [W] contig index as passed to simdiff_columns
    std::vector<int64_t> win_row0;         // [W+1]
    std::vector<int32_t> mask_ids;         // [rows]
    int32_t n_dev_windows = 0;             // windows whose graphs the device builds: the first n_matrix_windows from the contig's sim / diff
    int32_t n_matrix_windows = 0;          // matrices, the others (low-memory path) from window-local matrices; behind them the host's
    std::vector<int32_t> ctg_reads;        // [C] reads of every contig
    std::vector<int64_t> host_off;         // [host rows + 1], starts at 0
    std::vector<int32_t> host_nbr;
    std::vector<uint8_t> win_final_empty;  // [W] finalize_clustering sees an empty graph for this window (separate_reads.cpp:1708)
    std::vector<int32_t> rank;             // position of every read in its contig's shuffled visiting order, contigs concatenated
    std::vector<int32_t> pos_rank;         // row of every read in its contig's sim / diff matrices (SimdiffJob::pos_orig inverted), same layout as `rank`; empty: the read index
    std::vector<int64_t> ctg_rank_off;     // [C] first entry of each contig in `rank`
    float error_rate = 0;
    int64_t rows() const { return win_row0.empty() ? 0 : win_row0.back(); }
};

// The dependent Chinese-Whispers runs of every clustering window that has seeding SNPs, kept on the device end to end:
// per-SNP runs seeded from the SNP columns (separate_reads.cpp:1674-1705) -> merged ids (:840-874) -> run on the
// finalize graph (:881) -> small clusters dropped + renumbered (:924-955) -> run (:970) [-> K8: the tail of
// finalize_clustering]. Labels are local: m per window, chain window k at chain_row0[k].
// a read-only array that either views memory of the caller (the SNP columns stage 3 just produced, already contiguous in
// contig order) or owns a copy: no second copy of a hundred megabytes of columns per call when the first case applies
template <class T> struct ArrayView {
    const T* p = nullptr;
    size_t n = 0;
    std::vector<T> own;
    void view(const T* ptr, size_t count) { own.clear(); p = ptr; n = count; }
    T* alloc(size_t count) { own.resize(count); p = own.data(); n = count; return own.data(); }
    const T* data() const { return p; }
    size_t size() const { return n; }
    const T& operator[](size_t i) const { return p[i]; }
};

struct CwChain {
    ArrayView<int64_t> col_off;            // SNP columns of all contigs, concatenated CSR [S+1]
    ArrayView<int32_t> col_idx;
    ArrayView<uint8_t> col_code;
    std::vector<int32_t> win;              // [Wc] window index in the SrWindowSet
    std::vector<int64_t> chain_row0;       // [Wc+1] offset of the window's m labels
    std::vector<int64_t> win_seed_begin;   // [Wc+1] range of the window's per-SNP runs in seed_col
    std::vector<int64_t> seed_col;         // global column index of every per-SNP run
    // K8 (optional): the tail of finalize_clustering on the device. finish_on_device = every window of the chain may be
    // finished there (global low_memory off, SNP positions ascending); per window the range of its SNP columns (global
    // column indices) and the position interval [pos_lo, pos_hi) that merge_wrongly_split looks at.
    bool finish_on_device = false;
    std::vector<int32_t> col_pos;          // [S] position of every SNP column
    std::vector<int64_t> win_snp_first, win_snp_last;
    std::vector<int32_t> win_pos_lo, win_pos_hi;
};

struct CwWave {                       // one batched launch of single runs on the finalize graphs: (window, m initial labels)
    std::vector<int32_t> inst_win;
    std::vector<int64_t> inst_label_off;
    std::vector<int32_t> labels;      // in: initial labels, out: result
};

// K5 input: the SNP columns of every contig of the batch (concatenated in CwChain::col_*) with their two alleles
struct SimdiffJob {
    const CwChain* cols = nullptr;
    std::vector<uint8_t> snp_ref, snp_alt;       // per column
    std::vector<int32_t> snp_contig;             // per column
    std::vector<int64_t> contig_snp_base;        // [C] first column of each contig
    std::vector<int64_t> plane_off, out_off;     // [C] offsets of the contig's bit rows (uint64 words) / matrices (int32)
    std::vector<int32_t> n_reads, words;         // [C]; n_reads == 0: contig not on the matrix path
    std::vector<int32_t> plane_n;                // [C] reads of every contig that has bit rows (words > 0): the matrix path AND the low-memory path
    int64_t plane_total = 0, out_total = 0;
    // row order of the matrices: row k of contig c is the read pos_orig[read_base[c] + k] (reads by start position); read_base[c] = first
    // read of the contig among the reads of all contigs with bit rows. Empty pos_orig: rows = read indices.
    std::vector<int32_t> pos_orig;
    std::vector<int64_t> read_base;              // [C]
};

struct SrChainStats {                 // what the clustering chain did (bench.py's whole-path roofline)
    int64_t n_instances = 0;          // Chinese-Whispers runs
    int64_t sweeps = 0;               // sweeps over all runs
    int64_t bytes = 0;                // sum over runs of sweeps * (4 * nnz + 8 * m)
    int64_t graph_nnz = 0;            // neighbour entries of all window graphs
};

struct SrDeviceOps {
    virtual ~SrDeviceOps() {}
    // K5a + K5: bit-planes from the SNP columns, then sim / diff for every contig with n_reads[c] > 0. The columns (the same
    // object cw_chain() receives later) and the matrices stay with the implementation.
    virtual int simdiff_columns(const SimdiffJob& job, float* k_ms) = 0;
    // K6 (create_read_graph_matrix) for the device windows + the rows the host brings, one CSR, and the visiting order of
    // every window; all of it stays with the implementation. rows_on_host: rows whose cut-off fell inside a run of equal
    // distances (std::sort's order decides: resolved with std::sort itself)
    virtual int build_graphs(const SrWindowSet& ws, int64_t* rows_on_host, float* k_ms) = 0;
    // the same in two halves, for an implementation that can leave the device working on the rows while the caller plans the
    // Chinese-Whispers chain of the call (which needs the window plans only): begin queues, end resolves what the host has to resolve
    virtual bool two_phase_graphs() const { return false; }
    virtual int build_graphs_begin(const SrWindowSet& ws, float* k_ms) { (void)ws; (void)k_ms; return -1; }
    virtual int build_graphs_end(const SrWindowSet& ws, int64_t* rows_on_host) { (void)ws; (void)rows_on_host; return -1; }
    // host copy of the CSR of the last build_graphs(): off[rows + 1] absolute, nbr local ids
    virtual int fetch_graphs(std::vector<int64_t>& off, std::vector<int32_t>& nbr) = 0;
    // labels = what the third run leaves (m per chain window). If the implementation also ran K8, final_labels holds the
    // finished labels and final_ok[k] != 0 marks the windows it could finish (the others go through the host code); else
    // both stay empty. An implementation that finished EVERY window may leave `labels` empty (nobody reads them then).
    virtual int cw_chain(const CwChain& chain, std::vector<int32_t>& labels, std::vector<int32_t>& final_labels,
                         std::vector<uint8_t>& final_ok, float k_ms[3], SrChainStats* stats) = 0;
    virtual int cw(CwWave& wave, float* k_ms) = 0;
    // (test taps: the next cw_chain() also leaves the labels of its per-SNP runs -- run_off / run_labels -- and always fills `labels`)
    virtual void tap_chain(std::vector<int64_t>* run_off, std::vector<int32_t>* run_labels) { (void)run_off; (void)run_labels; }
    // The SNP columns of the call may be with the implementation already (stage 3 left them on the device, in the order and with
    // the offsets of CwChain::col_off): then CwChain's col_idx / col_code stay empty and ...
    // create_read_graph_low_memory on the device (window-local sim / diff from the bit rows); false: the caller builds those rows
    virtual bool low_memory_graphs() const { return false; }
    virtual bool columns_resident() const { return false; }
    virtual void drop_resident_columns() {}
    // ... the reads of every clustering window -- those present at its first AND its last SNP column (separate_reads.cpp:1590-1622),
    // global column indices col_a / col_b -- come from there: the reads of window w at ids[slot_off[w] ...], win_m[w] of them
    // (slot_off: room for the whole first column of every window) ...
    virtual int window_masks(const std::vector<int64_t>& col_a, const std::vector<int64_t>& col_b, const std::vector<int64_t>& slot_off,
                             std::vector<int32_t>& ids, std::vector<int32_t>& win_m) { (void)col_a; (void)col_b; (void)slot_off; (void)ids; (void)win_m; return -1; }
    // ... and the columns themselves can be fetched for the few places that walk them on the host (low-memory graphs, windows the
    // device did not finish)
    virtual int fetch_columns(std::vector<int32_t>& idx, std::vector<uint8_t>& code) { (void)idx; (void)code; return -1; }
};

void set_trace_origin();   // HS_TIMING=abs: laps are printed relative to this moment
double trace_ms();         // milliseconds since then
int host_threads();      // default number of host worker threads: usable cores (cgroup quota), at most 32

// The labels of a result before they are spread over the N reads of each window: per window the reads it holds (ascending
// ids) and their labels. A caller that merges several partial results (contig groups, device shards) asks for this form
// and writes the dense array once, in its final place (sr_expand_labels), instead of building it per part and copying it.
struct SrSparseLabels {
    std::vector<int64_t> off;           // [W+1]
    std::vector<int32_t> ids, labels;
};
// sparse != nullptr: filled, and out->labels stays nullptr (label_off is filled as usual)
// state of a stage-4 call that is worth keeping for the next one on the same contigs (a pipeline group runs the same contigs step
// after step): the per-contig plans with their storage, the shuffled visiting orders
struct SrWorkspace { std::vector<SrContigState> st; };
// Test taps of the clustering chain of a stage-4 call (hs_sr_run_taps): the windows that have seeding SNPs, in chain order, with what the
// kernels of the chain left -- the labels of every per-SNP Chinese-Whispers run (k_cw_seed_sets + k_cw_seeded_lanes / _rows / _wave) and
// of the third run (inside k_window_tail), as window-local node ids / cluster indices
struct SrTaps {
    std::vector<int32_t> win_contig, win_start;   // contig index in the call, first position of the window
    std::vector<int64_t> win_row0;                // [Wc + 1] into mask_ids
    std::vector<int32_t> mask_ids;                // the window's reads (ascending)
    std::vector<int64_t> run_begin;               // [Wc + 1] the window's per-SNP runs
    std::vector<int32_t> run_snp;                 // SNP index on its contig of every run
    std::vector<int64_t> run_off;                 // [runs + 1] into run_labels (m labels per run)
    std::vector<int32_t> run_labels, third;       // third: [win_row0.back()] what the third run left
};
int sr_run(SrDeviceOps& dev, const hs_sr_contig* contigs, int32_t n_contigs, int32_t window_size, float error_rate,
           int32_t low_memory, uint32_t seed, int32_t n_threads, hs_sr_result** out, SrSparseLabels* sparse = nullptr, SrWorkspace* keep = nullptr, SrTaps* taps = nullptr);

int sr_run_from_cv(SrDeviceOps& dev, const CvMeta& meta, int c0, int c1, const hs_cv_result* cv, float error_rate, float rarest_strain_abundance,
                   int32_t low_memory, int32_t amplicon, uint32_t seed, int32_t n_threads, int32_t window_size, hs_sr_result** out,
                   SrSparseLabels* sparse = nullptr, SrWorkspace* keep = nullptr);

// the dense label array of an hs_sr_result (recycled big blocks; free_sr_result returns it)
int32_t* sr_labels_alloc(size_t n_labels);
void sr_labels_free(int32_t* labels);
// windows [w0, w1) of `sp` into dense[label_off[w] - label_off[0] ...]: -2 for the reads a window does not hold
void sr_expand_labels(const SrSparseLabels& sp, const int64_t* label_off, int64_t w0, int64_t w1, int32_t* dense);

// .col reader of HS_separate_reads (separate_reads.cpp:46-190)
struct ColFileContig {
    std::string contig_line, name;
    long length = 0;
    std::vector<std::string> read_lines;
    std::vector<int32_t> read_start, read_end;
    std::vector<int32_t> snp_pos;
    std::vector<uint8_t> snp_ref, snp_alt;
    std::vector<int64_t> col_off{0};
    std::vector<int32_t> col_idx;
    std::vector<uint8_t> col_code;
};
int parse_col(const std::string& path, float rarest_strain_abundance, std::vector<ColFileContig>& cs, int n_threads = 1);
// the binary companion HS_call_variants leaves next to the .col (<col>.hsbin, hs_io.cpp): 1 = cs filled from it (it matched the .col
// block by block), 0 = no usable companion, parse the text
int read_col_sidecar(const std::string& col_path, float rarest_strain_abundance, std::vector<ColFileContig>& cs, int n_threads = 1);

// writers shared by the executables and the test harness
int write_cv_outputs(const CvFileInput& in, const hs_cv_result* res, const std::string& error_rate_out,
                     const std::string& col_path, const std::string& vcf_path, int n_threads = 1);
int write_gro(const std::vector<ColFileContig>& cs, const hs_sr_result* res, const std::string& path, int n_threads = 1);
// <col>.hsgro: the .gro of the orchestrator's usual stage-4 call, precomputed by HS_call_variants (hs_io.cpp)
int write_gro_companion(const std::string& col_path, const std::vector<ColFileContig>& cs, const hs_sr_result* res, float error_rate, float rsa, bool low_memory, bool amplicon,
                        uint32_t seed, int32_t window, int n_threads);
int take_gro_companion(const std::string& col_path, float error_rate, float rsa, bool low_memory, bool amplicon, uint32_t seed, const std::string& outfile, int n_threads);
void remove_gro_companion(const std::string& col_path);
bool mark_gro_companion_pending(const std::string& col_path, float error_rate, float rsa, bool low_memory, bool amplicon, uint32_t seed);

void free_cv_result(hs_cv_result* r);
void free_sr_result(hs_sr_result* r);

}  // namespace hs
