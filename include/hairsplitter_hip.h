/*
 * hairsplitter_hip.h -- C ABI of the MI355X (gfx950) implementation of HairSplitter's hot path
 * (HS_call_variants -> HS_separate_reads). Plain pointers and sizes only; no C++/torch types.
 *
 * The reference has no FFI: its boundary is two executables and five text formats.
 * The drop-in executables shipped here (hairsplitter_amd/bin/HS_call_variants, HS_separate_reads) are thin
 * hosts over this library; every entry point below names the reference function (file:line under
 * /root/reference/src) whose work it replaces, so a maintainer could also call it in-process.
 *
 * Conventions
 *  - Pointers named d_* are DEVICE (HBM) pointers, h_* are host pointers. `stream` is a hipStream_t passed
 *    as void* (NULL = the default stream). Kernel-level calls are asynchronous on `stream`.
 *  - Every function returns 0 on success, a negative HS_E* code otherwise; hs_last_error() gives the text.
 *  - Sequences are 1 byte per base, codes A=0 C=1 G=2 T=3 (non-ACG input bases become T exactly like the
 *    reference's 2-bit Sequence, sequence.cpp:13-23). Reads are stored as sequenced (FASTA orientation); the
 *    kernels reverse-complement on the fly for records whose strand is 0 (call_variants.cpp:108-115).
 *  - CIGARs are BAM-style uint32 (len << 4 | op), op: M=0 I=1 D=2 N=3 S=4 H=5 P=6 '='=7 X=8.
 *  - "record" = one kept SAM line = one read index n of a contig (call_variants.cpp:85-86). Records are
 *    grouped by contig, in SAM file order inside a contig.
 */
#ifndef HAIRSPLITTER_HIP_H
#define HAIRSPLITTER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HS_OK 0
#define HS_ENODEVICE (-1)   /* no HIP device / extension unusable: the product never falls back to a CPU path */
#define HS_EINVAL (-2)
#define HS_EHIP (-3)
#define HS_EIO (-4)
#define HS_EFORMAT (-5)

/* ------------------------------------------------------------------------------------------------
 * Library / device
 * ---------------------------------------------------------------------------------------------- */
const char* hs_version(void);
const char* hs_last_error(void);
int hs_device_count(void);                       /* hipGetDeviceCount; 0 when there is no GPU */
int hs_warmup(void);                             /* runtime + context + code-object load (first launch); returns the device count.
                                                    The executables run it on a side thread while they parse their inputs. */
int hs_set_device(int device);
int hs_device_synchronize(void);
/* raw HBM helpers so that hosts without a HIP binding (ctypes, cgo, JNI) can stage buffers */
int hs_malloc(void** d_ptr, size_t bytes);
int hs_free(void* d_ptr);
int hs_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
int hs_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
int hs_memset(void* d_ptr, int value, size_t bytes);
/* event timing on the stream the kernels run on (bench.py's roofline leg) */
int hs_event_create(void** ev);
int hs_event_destroy(void* ev);
int hs_event_record(void* ev, void* stream);
int hs_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);   /* synchronises on ev_stop */

/* ------------------------------------------------------------------------------------------------
 * Per-kernel accounting of everything the stage drivers launch (bench.py's roofline leg): every launch is bracketed by HIP
 * events on the stream it runs on; the elapsed times, the number of launches and the ALGORITHMIC bytes of each launch
 *  are accumulated per kernel family, process-wide, until hs_kernel_stats_reset().
 * ---------------------------------------------------------------------------------------------- */
#define HS_NKERNELS 28
enum {   /* one slot per kernel of the path (a slot's helper launches -- prefix scans, block sums -- are counted with it) */
    HS_K_CIGAR_SCAN = 0, HS_K_PILEUP, HS_K_COLUMN_STATS, HS_K_COLUMNS_COMPACT, HS_K_GATHER_COLUMNS, HS_K_COLUMN_TOP3, HS_K_CANDIDATES_SCAN,
    HS_K_PACK_COLUMNS, HS_K_PARTITION_TRANSPOSE, HS_K_PARTITION_LANES, HS_K_PARTITION_TEST, HS_K_SNP_SELECT, HS_K_WINDOW_MASKS,
    HS_K_SNP_PLANES, HS_K_SIMDIFF, HS_K_GRAPH_ROWS, HS_K_GRAPH_CSR, HS_K_VISIT_LISTS, HS_K_CW_SEED_SETS, HS_K_CW_SEEDED, HS_K_CW_SEEDED_WIDE,
    HS_K_WINDOW_TAIL, HS_K_CW_LOCAL, HS_K_ROBUST_PARTITIONS, HS_K_OTHER, HS_K_CAND_BITS, HS_K_SHIP, HS_K_PARTITION_GROUPED
};
typedef struct hs_kernel_stats {
    double ms[HS_NKERNELS];        /* sum of the launch durations (hipEventElapsedTime) */
    int64_t launches[HS_NKERNELS];
    int64_t bytes[HS_NKERNELS];    /* sum of the algorithmic bytes of the launches */
} hs_kernel_stats;
const char* hs_kernel_name(int k);          /* name of the (dominant) kernel of family k as rocprofv3 prints it */
void hs_kernel_stats_reset(void);
void hs_kernel_stats_get(hs_kernel_stats* out);
/* Host waits for the device since the library was loaded: each is one round trip in the chain of a contig group (bench.py reports
 * them per step). The time spent in them is only accumulated under HS_TIMING. A wait polls hipStreamQuery and sleeps 25 us between two
 * looks; for the duration of the wait the calling thread's timer slack is set to 1 us (prctl PR_SET_TIMERSLACK; HS_TIMER_SLACK_NS=0: not
 * touched) and the thread's own value is put back before the call returns. */
/* Timing events of the library (two per kernel launch for hs_kernel_stats, a few per phase for the t_kernel_* fields) are recorded on every
 * pipeline step by default; hs_kernel_stats_every(n) records them on every n-th step only (steps counted from this call; hs_pipeline_select /
 * hs_pipeline_run_fused begin a step). hs_kernel_stats then holds the launches of the timed steps; the t_kernel_* fields of an untimed step are 0.
 * About 1200 event records per step of the 500-contig job cost it 1.3 ms of 16.9. */
void hs_kernel_stats_every(int32_t n);
void hs_host_wait_stats(int64_t* n_waits, double* ms_in_waits);

/* ------------------------------------------------------------------------------------------------
 * K1 -- pileup.  Replaces generate_msa (call_variants.cpp:50-437) + convert_cigar (tools.cpp:27-57).
 * One wavefront per task (a range of alignment events of one record). Launches K0 (CIGAR scan: fills d_chunk_scratch with the
 * cursors at every 64-op chunk and flags the records that hold a clip between aligned bases), the packed pileup kernel (four
 * events per lane) over the task list and the per-event kernel over the flagged records. For record r with reference start pos[r]:
 *   d_pile[pile_off[r] + (q - pos[r])] = 33 + 5*i(c-2) + i(c-1) + 25*i(c0)   for every M/=/X/D event at q < L
 * (i() = index in "ACGT-", previous chars initialised C,G as call_variants.cpp:212-214 leave them).
 * d_rec_stats[r] = {q_end, n_err, n_len, n_events}: final reference cursor (call_variants.cpp:354) and the number
 * of +1's applied to totalDistance / totalLengthOfAlignment (call_variants.cpp:255-257,305-306,337-338).
 * ---------------------------------------------------------------------------------------------- */
int hs_pileup(const uint8_t* d_contig_seq, const int64_t* d_contig_off,   /* [C+1] */
              const uint8_t* d_read_seq, const int64_t* d_read_off,        /* [NR+1] */
              const int32_t* d_rec_read, const int32_t* d_rec_contig, const int32_t* d_rec_pos,
              const uint8_t* d_rec_strand, const int64_t* d_rec_cig_off,   /* [NREC+1] */
              const uint32_t* d_cigar, const int64_t* d_pile_off,          /* [NREC+1] */
              int32_t n_rec,
              /* launch plan from hs_pileup_plan (work is split in equal ranges of alignment events, not per record) */
              const int64_t* d_rec_chunk_off, int32_t* d_chunk_scratch /* [4 * rec_chunk_off[NREC]] */,
              const int32_t* d_task_rec, const int32_t* d_task_ev0, int32_t n_tasks, int32_t ev_per_task,
              uint8_t* d_pile, int32_t* d_rec_stats /* [NREC*4] */, void* stream);

/* Host-side launch plan of hs_pileup, from host copies of the CIGARs (the per-record aggregates parse_SAM also
 * derives, input_output.cpp:486-505): rec_chunk_off[r] = number of 64-op chunks before record r; tasks = (record,
 * first event) pairs covering every record in ranges of ev_per_task events. The two task arrays are malloc'ed;
 * release them with hs_free_host. */
int hs_pileup_plan(const int64_t* h_rec_cig_off, const uint32_t* h_cigar, int32_t n_rec, int32_t ev_per_task,
                   int64_t* h_rec_chunk_off /* [NREC+1] out */, int32_t* n_tasks, int32_t** h_task_rec, int32_t** h_task_ev0);
void hs_free_host(void* p);

/* ------------------------------------------------------------------------------------------------
 * K2 -- per-position code histogram and top-5.  Replaces the counting half of call_variants
 * (call_variants.cpp:471-507): for every contig position the five most frequent pileup codes.
 * Output record (16 B / position): u8 key[4]; u16 cnt[5]; u16 depth -- ordered by (count desc, code asc);
 * missing entries have key 0 / count 0. The reference's tie order (robin_hood iteration order + std::sort) is
 * applied afterwards by the host only where it is observable (hs_colstat flags).
 * d_contig_rec_off[C+1] gives each contig's record range; d_rec_qend[r] = exclusive end of record r's pileup.
 * ---------------------------------------------------------------------------------------------- */
typedef struct hs_colstat {
    uint8_t key[4];
    uint16_t cnt[5];
    uint16_t depth;
} hs_colstat;

/* ------------------------------------------------------------------------------------------------
 * Tile plan for K2 / K3: which records overlap each tile of 256 consecutive positions of the concatenated contigs, in
 * ascending record order. Built on the host from the same per-record aggregates as the pileup layout (one pass over the
 * records; the batch keeps it for its lifetime). entry.first = lane of the tile that holds the record's first position
 * (negative when the record starts in an earlier tile), entry.len = positions the record covers from there,
 * entry.base = offset in the pileup of the byte that lane 0 of the tile would read.
 * Outputs are malloc'ed; release them with hs_free_host.
 * ---------------------------------------------------------------------------------------------- */
typedef struct hs_tile_entry { int32_t first, len; int64_t base; } hs_tile_entry;
int hs_tile_plan(const int64_t* h_contig_off, int32_t n_contigs, const int32_t* h_contig_rec_off, const int32_t* h_rec_pos,
                 const int32_t* h_rec_qend, const int64_t* h_pile_off, int64_t** tile_off /* [n_tiles+1] */,
                 hs_tile_entry** tile_ent, int32_t** tile_rec, int64_t* n_tiles);
/* K2 on a tile plan in its full-statistics form (total_len = contig_off[C]): d_stats [sum L] or NULL = selection only; optional compact
 * selection (all NULL / 0 to skip): global positions (index into the concatenated contigs) whose second count is >= min_second,
 * unordered, with their depth; *d_sel_count must be 0; max_depth: upper bound on the depth of any position, if known (1..255
 * selects 8-bit LDS counters), 0 = unknown. (The stage drivers run the selection-only form k_column_stats_tiled_dw: see
 * hs_cv_column_pass_taps.) */
int hs_column_stats_tiled(const uint8_t* d_pile, const int64_t* d_tile_off, const hs_tile_entry* d_tile_ent, int64_t total_len,
                          hs_colstat* d_stats, int32_t min_second, int32_t* d_sel_count, int64_t* d_sel_gpos,
                          int32_t* d_sel_depth, int32_t sel_cap, int32_t max_depth, void* stream);
/* Exclusive prefix sum of n non-negative ints into n + 1 64-bit offsets (d_out[n] = total): the CSR offsets of the read
 * graphs (K6) and of the selection list (K2) are built with it. Synchronous with respect to `stream`. */
int hs_exclusive_scan_i32(const int32_t* d_in, int32_t n, int64_t* d_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K4 -- SNP column x partition correlation.  Replaces distance(Partition&, Column&) + computeChiSquare
 * (call_variants.cpp:778-967, :1135-1163) as used by loops C and D of keep_only_robust_variants (:721-764).
 * Columns are a CSR as the column pass of stage 3 leaves it (hs_cv_taps); col_k0 / col_k1 their two most frequent codes in the reference's
 * tie order; col_c1 the second count; col_is_cand marks candidate SNPs (loop C). Partitions are dense int8 state
 * arrays over the contig's reads (1, -1, 0, or 2 = read absent); contig c owns partitions [part_off[c], part_off[c+1]).
 * Two kernels: lanes = partitions on a [read][partition] table built on the device (64 partitions per step), then the
 * one-partition-at-a-time form for the few columns whose verdict hinges on the reference's order of tied second alleles.
 * d_keep[i] = 1 if column i is kept by loop C (chi2 > 15 on more than half of its reads) or rescued by loop D
 * (chi2 > 20, both alleles carried by more than four partition reads).
 * ---------------------------------------------------------------------------------------------- */
int hs_column_partition_test(const int64_t* d_col_off, const int32_t* d_col_idx, const uint8_t* d_col_code,
                             const int32_t* d_col_contig, const uint8_t* d_col_k0, const uint8_t* d_col_k1,
                             const int32_t* d_col_c1, const uint8_t* d_col_is_cand, int32_t n_cols,
                             const int32_t* d_part_off, const int64_t* d_part_state_off, const int8_t* d_part_state,
                             const int32_t* h_contig_n_reads /* HOST, [C]: reads of every contig (length of its state arrays) */, int32_t n_contigs,
                             uint8_t* d_keep, void* stream);
/* Test tap: what the kernels of the LAST hs_column_partition_test call of this thread left to one another -- out[0] columns the first kernel
 * (k_column_partition_lanes) passed on to k_column_partition_grouped, out[1] columns that one passed on whole to k_column_partition_test,
 * out[2] (column, partition) pairs it passed on to k_column_partition_pairs. */
void hs_column_partition_last_counts(int32_t out[3]);

/* ------------------------------------------------------------------------------------------------
 * V5 -- distance(Partition&, Partition&, threshold_p) of loop B (call_variants.cpp:977-1127) for a list of partition pairs.
 * Partitions as dense arrays one after the other: partition p occupies [part_off[p], part_off[p] + part_n[p]) of d_state
 * (mostFrequentBases in {-1, 0, 1}, 2 = read absent), d_more / d_less (moreFrequence / lessFrequence); the two partitions of a
 * pair belong to the same contig (same part_n). d_sigma3[n], n < 4096 = (float)(0.5 n + 3 sqrt(0.25 n)) as the host forms it.
 * d_out: 8 ints per pair {n00, n01, n10, n11, phased, augmented, valid, comparable}; valid = 0 when a read's vote count is
 * beyond the table (the caller's own walk decides). pair_a = par1, pair_b = par2 of the reference's call.
 * ---------------------------------------------------------------------------------------------- */
int hs_partition_pair_distance(const int8_t* d_state, const int32_t* d_more, const int32_t* d_less, const int64_t* d_part_off, const int32_t* d_part_n,
                               const int32_t* d_pair_a, const int32_t* d_pair_b, int32_t n_pairs, int32_t threshold_p, const float* d_sigma3, int32_t* d_out,
                               void* stream);

/* ------------------------------------------------------------------------------------------------
 * K5a -- SNP bit-planes for K5 from the SNP columns (separate_reads.cpp:386-405): bit s of row r of d_ref / d_alt = read r
 * carries snp_ref[s] / snp_alt[s]. Columns of all contigs concatenated (CSR); snp_contig[s] = contig of column s,
 * contig_snp_base[c] = first column of contig c; plane_off / words / n_reads as for hs_simdiff (words[c] == 0 skips the contig).
 * Every word of the rows of a contig with words[c] > 0 is written (the planes need no clearing). h_words: the HOST copy of
 * words[] (the launch has one workgroup per contig and four words).
 * ---------------------------------------------------------------------------------------------- */
int hs_snp_planes(const int64_t* d_col_off, const int32_t* d_col_idx, const uint8_t* d_col_code, const uint8_t* d_snp_ref,
                  const uint8_t* d_snp_alt, const int32_t* d_snp_contig, const int64_t* d_contig_snp_base,
                  const int64_t* d_plane_off, const int32_t* d_words, const int32_t* d_n_reads, const int32_t* h_words /* HOST, [C] */,
                  int32_t n_contigs, int32_t n_snps, uint64_t* d_alt, uint64_t* d_ref, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K5 -- read x read similarity / difference.  Replaces list_similarities_and_differences_between_reads3
 * (separate_reads.cpp:374-433): sim = 3*A*At + R*Rt, diff = A*Rt + R*At with zero diagonals, as popcounts of
 * bit-planes. d_alt / d_ref: N rows of `words` uint64 (bit s of row r = read r carries second_base / ref_base
 * at SNP s; a read carries at most one of the two at a SNP: the planes are disjoint, which the kernel relies on). Batched over
 * contigs: plane_off[c] (in uint64 words), out_off[c] (in int32 elements).
 * ---------------------------------------------------------------------------------------------- */
int hs_simdiff(const uint64_t* d_alt, const uint64_t* d_ref, const int64_t* d_plane_off,
               const int32_t* d_n_reads, const int32_t* d_words, const int64_t* d_out_off, int32_t n_contigs,
               int32_t* d_sim, int32_t* d_diff, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K6 -- read graph of every clustering window. Replaces create_read_graph_matrix (separate_reads.cpp:706-828).
 * d_sim / d_diff: the matrices of hs_simdiff, contig c at ctg_out_off[c] with ctg_n_reads[c] rows. A window is
 * (contig, ascending list of its masked read ids); row = one masked read. Output (malloc'ed, release with
 * hs_free_host): nbr_off[rows+1] and nbr = the reads linked to every masked read, ascending (the symmetric adjacency
 * the reference stores in its Eigen matrix / neighbour lists). Rows whose result depends on how std::sort arranges
 * equal distances are finished on the host with std::sort itself; *n_rows_host reports how many.
 * All array arguments except d_sim / d_diff are host pointers.
 * ---------------------------------------------------------------------------------------------- */
int hs_read_graphs(const int32_t* d_sim, const int32_t* d_diff, const int64_t* ctg_out_off, const int32_t* ctg_n_reads,
                   int32_t n_contigs, const int32_t* win_contig, const int64_t* win_mask_off, const int32_t* mask_ids,
                   int32_t n_windows, float error_rate, int64_t** nbr_off, int32_t** nbr, int64_t* n_rows_host, void* stream);

/* ------------------------------------------------------------------------------------------------
 * A1 -- Myers bit-vector edit distance, batched and banded (64 query rows per lane, carries passed between lanes;
 * a wavefront per pair, queries up to 2048 bases 8 / 16 / 32 lanes per pair; the band's bound is found on the way, as
 * edlib's k = -1 does). The reference's hot path takes base-level alignments from the SAM CIGAR;
 * its bundled edlib (edlib.h:242-246, modes edlib.h:36-62) is the behavioural oracle for this kernel.
 * mode: 0 = NW (global), 1 = SHW (prefix), 2 = HW (infix). Outputs: edit distance and the first end location
 * on the target (0-based, inclusive), like edlibAlign's editDistance / endLocations[0].
 * ---------------------------------------------------------------------------------------------- */
int hs_edit_distance(const uint8_t* d_query, const int64_t* d_query_off, const uint8_t* d_target,
                     const int64_t* d_target_off, int32_t n_pairs, int32_t mode, int32_t* d_dist,
                     int32_t* d_end, void* stream);

/* A1 as the stage-5 call sites of the reference use their bundled edlib (create_new_contigs.cpp:558-629, tools.cpp:515-534):
 * edlibAlign(query, target, edlibNewAlignConfig(-1, EDLIB_MODE_HW, EDLIB_TASK_PATH, NULL, 0)), batched, one wavefront per
 * pair. d_dist = editDistance, d_end = endLocations[0], d_start = startLocations[0] (edlib.cpp:226-258), d_ops[h_ops_off[i] ..
 * + d_ops_len[i]) = alignment of pair i in edlib's move codes (0 '=', 1 insertion, 2 deletion, 3 mismatch; every pair needs
 * room for query + target operations). d_ops == NULL: locations only (EDLIB_TASK_LOC). d_ops_len[i] = -1 where edlib
 * itself has no alignment to offer (end location -1) or the query has more than 2^20 bases. Long pairs are cut in halves the way
 * edlib cuts them (Hirschberg, edlib.cpp:1166-1404), so the alignment is edlib's at any length. Offsets are HOST arrays [n+1];
 * sequences are base codes 0..3. Synchronous with respect to `stream`. */
int hs_edlib_hw_align(const uint8_t* d_query, const int64_t* h_query_off, const uint8_t* d_target, const int64_t* h_target_off,
                      int32_t n_pairs, int32_t* d_dist, int32_t* d_start, int32_t* d_end, uint8_t* d_ops, const int64_t* h_ops_off,
                      int32_t* d_ops_len, void* stream);

/* A1 on the path: a CIGAR-less input. The reference reads the base-level alignments from the CIGARs of a SAM file and
 * refuses a .paf (call_variants.cpp:1256-1267; CIGAR required at input_output.cpp:357-368). hs_realign_paf turns a PAF file (read
 * interval, strand, contig interval per line) into that SAM: every read segment is aligned against its contig window (the PAF
 * interval + HS_REALIGN_PAD = 100 bases on both sides) on the device exactly as edlibAlign(segment, window, k = -1, EDLIB_MODE_HW,
 * EDLIB_TASK_PATH) of the reference's bundled edlib aligns it; POS = window start + start location + 1, CIGAR = clips (S) + the
 * path as M / I / D runs, NM:i: = the edit distance, LN:i: = the read length. HS_call_variants takes a .paf this way when
 * HS_REALIGN=1 is set (the SAM goes to <tmpDir>/hs_realigned.sam); without it the reference's refusal stands. */
typedef struct hs_realign_stats { int64_t n_lines, n_aligned, query_bases; double ms_device, ms_total; } hs_realign_stats;
int hs_realign_paf(const char* gfa, const char* reads, const char* paf, const char* out_sam, int32_t n_threads, hs_realign_stats* stats /* may be NULL */);

/* The two stage-5 computations that sit on those edlib calls, batched (two alignments per item in ONE hs_edlib_hw_align call):
 * hs_reattach_ends == tools.cpp:505-536 (the ends of the backbone that racon dropped are attached to the consensus again),
 * hs_trim_polished == create_new_contigs.cpp:556-629 (the overhangs the piece was polished with are cut off the polished
 * sequence through the alignment path of the piece's ends). Strings are NUL-terminated ACGT; *out = n malloc'ed strings,
 * released with hs_free_strings. */
int hs_reattach_ends(const char* const* backbone, const char* const* consensus, int32_t n, char*** out);
int hs_trim_polished(const char* const* to_polish, const char* const* newcontig, const int32_t* overhang_left, const int32_t* overhang_right,
                     int32_t n, char*** out);
void hs_free_strings(char** s, int32_t n);

/* ------------------------------------------------------------------------------------------------
 * Stage level (host buffers in, host buffers out). These run the whole stage exactly as the drop-in
 * executables do: device kernels for pileup / histogram / extraction / sim-diff / Chinese Whispers, host code
 * for the sequential glue. Split in upload + run so callers can time the path with inputs resident in HBM.
 * ---------------------------------------------------------------------------------------------- */
typedef struct hs_cv_batch hs_cv_batch;   /* opaque: a batch of contigs + reads + records resident in HBM */

/* call_variants.cpp:1249-1262 hand-off: what parse_reads/parse_assembly/parse_SAM produce, flattened. */
int hs_cv_batch_create(const uint8_t* h_contig_seq, const int64_t* h_contig_off, int32_t n_contigs,
                       const uint8_t* h_read_seq, const int64_t* h_read_off, int32_t n_reads,
                       const int32_t* h_rec_read, const int32_t* h_rec_pos, const uint8_t* h_rec_strand,
                       const int64_t* h_rec_cig_off, const uint32_t* h_cigar,
                       const int32_t* h_contig_rec_off, hs_cv_batch** out);
void hs_cv_batch_destroy(hs_cv_batch* b);
/* Optional ploidy of every contig (0 = none), the in-memory counterpart of the <ploidy_of_contigs> file of HS_separate_reads
 * (separate_reads.cpp:1444-1463, used at :1711-1715): the stage-3 -> stage-4 entry points below (hs_sr_run_cv, hs_sr_run_cv_range,
 * hs_pipeline_run) apply it. ploidy == NULL clears it. */
int hs_cv_batch_set_ploidy(hs_cv_batch* b, const int32_t* ploidy /* [n_contigs] or NULL */);
int64_t hs_cv_batch_aligned_bp(const hs_cv_batch* b);   /* number of pileup entries (the metric's unit) */
/* The device the batch lives on (the one that was current on the creating thread). Every entry point that takes a batch binds
 * the calling thread -- and every thread the library starts for it -- to that device first. */
int hs_cv_batch_device(const hs_cv_batch* b);

/* A column of the pileup as the device keeps it between the kernels of stage 3 (16 bytes). */
typedef struct hs_colrec {
    int32_t pos;          /* position on its contig */
    int32_t contig;       /* contig index in the batch */
    uint16_t c0, c1;      /* reads carrying the two most frequent codes (call_variants.cpp:497-507) */
    uint8_t k0, k1;       /* those codes, equal counts in the reference's order (robin_hood iteration order + std::sort) */
    uint8_t flags;        /* HS_COL_* */
    uint8_t c2;           /* the third count, saturating at 63 */
} hs_colrec;
#define HS_COL_CAND 1     /* candidate SNP of call_variants.cpp:525-536 */
#define HS_COL_AUTO 2     /* ... that also passes the automatic threshold (:532) */
#define HS_COL_LOOPD 4    /* may be rescued by loop D (:745-764) */
#define HS_COL_KEEP 8     /* kept by loop C or D */
#define HS_COL_SNP 16     /* in the output (:1335-1352) */
#define HS_COL_TIE 32     /* equal counts among its leading codes */
#define HS_COL_C1GT5C2 64 /* second count > 5 x third count (:526) */

/* Per-contig result of stage 3 == what output_files (call_variants.cpp:1174-1213) prints. */
typedef struct hs_cv_result {
    int32_t n_contigs;
    float* mean_distance;      /* [C] generate_msa's return value */
    float* depth;              /* [C] call_variants.cpp:565 */
    int64_t* snp_off;          /* [C+1] */
    int32_t* snp_pos;          /* [S] */
    uint8_t* snp_ref;          /* [S] */
    uint8_t* snp_alt;          /* [S] */
    int32_t* snp_n_ref;        /* [S] reads carrying snp_ref / snp_alt (what parse_column_file recounts, separate_reads.cpp:151-167) */
    int32_t* snp_n_alt;        /* [S] */
    int64_t* col_off;          /* [S+1] */
    int32_t* col_idx;          /* read indices, ascending */
    uint8_t* col_code;         /* pileup codes */
    float error_rate;          /* call_variants.cpp:1312-1315,1377 */
    int32_t n_contigs_with_error_rate;
    double t_device_ms;        /* wall time of the device phase (uploads of selections, kernels, downloads) */
    double t_host_ms;          /* wall time of the host glue */
    float t_kernel_ms[4];      /* hipEvent time of k_pileup, k_column_stats, k_gather_columns, k_cigar_scan */
    float t_kernel_k4_ms;      /* hipEvent time of k_column_partition_test */
    int64_t n_columns_extracted;        /* K3: columns of the selected positions (they stay on the device) */
    int64_t n_columns_downloaded;       /* of those, candidate SNPs: the columns the host walks (loops A / B) */
    int64_t n_columns_downloaded_late;  /* columns whose leading codes had equal counts (ordered on the device as the reference orders them) */
    int32_t entries_borrowed;           /* col_idx / col_code point into a block of the pipeline that made the result (valid until its next call): not freed with it */
} hs_cv_result;

int hs_cv_run(hs_cv_batch* b, float automatic_snp_threshold, int32_t n_threads, hs_cv_result** out);

/* Test taps of the column pass of stage 3 exactly as the pipeline queues it for the contigs [c0, c1) of a resident batch -- K0, K1, K2
 * (k_column_stats_tiled_dw), k_columns_compact, k_gather_tiles_direct / k_gather_tiles, k_column_top3_exact, k_candidates_scan,
 * k_flag_block_sums / _offsets, k_pack_flagged, k_cand_bits -- with what those kernels left on the device copied out, so that each of
 * them can be held against the oracle on its own (tests/test_gpu_kernels.py). The reference has no counterpart: call_variants.cpp:
 * 471-536 walks positions one by one. */
typedef struct hs_cand_bits {        /* == hs::CandBits (hs_host.h): a candidate column as bit sets over the reads ranked by start position */
    int32_t wlo; uint16_t n_words, n_slots; int32_t idx_min, idx_max, reach, n_entries; int64_t word_off;
} hs_cand_bits;
typedef struct hs_cv_taps {
    int32_t n_contigs;               /* c1 - c0 */
    int64_t n_cols, n_entries;       /* the extracted columns (K2's selection: every position that can still become a SNP) and their entries */
    int64_t* col_gpos;               /* [n_cols] position in the concatenated contigs of the batch, ascending (k_columns_compact) */
    hs_colrec* col_rec;              /* [n_cols] leading codes and counts (K2's second pass / k_column_top3_exact), flags (k_candidates_scan) */
    int64_t* col_off;                /* [n_cols + 1] */
    int32_t* col_idx;                /* [n_entries] read index on the contig, ascending inside a column (k_gather_tiles*) */
    uint8_t* col_code;               /* [n_entries] pileup code */
    int64_t n_cand;                  /* candidates (HS_COL_CAND) */
    hs_colrec* cand_rec;             /* [n_cand] k_pack_flagged: their records ... */
    int32_t* cand_col;               /* ... and their index among the columns */
    hs_cand_bits* cand_bits;         /* [n_cand] k_cand_bits */
    uint64_t* cand_words;            /* [n_cand_words] */
    int64_t n_cand_words;
    int32_t* contig_n_cand;          /* [n_contigs] */
    float* contig_mean_distance;     /* [n_contigs] k_contig_error */
} hs_cv_taps;
int hs_cv_column_pass_taps(hs_cv_batch* b, int32_t c0, int32_t c1, float automatic_snp_threshold, hs_cv_taps** out);
void hs_cv_taps_destroy(hs_cv_taps* t);

/* Several GPUs in one process. Contigs are the independent units of both stages (call_variants.cpp:1276-1280,
 * separate_reads.cpp:1506-1508): hs_cv_run_host and hs_sr_run shard them over the devices of hs_devices() by
 * longest-processing-time (stage 3: bases of the reads aligned to the contig; stage 4: N^2 + N * S), one host thread and one
 * device per shard, results merged in contig order (the error rate over the whole job). hs_devices: the device list =
 * HS_DEVICES="0,1,.." or every visible device; a device may be listed more than once (two shards on one GPU).
 * hs_cv_run_host = hs_cv_batch_create + hs_cv_run + destroy per shard (host buffers as for hs_cv_batch_create). */
int hs_devices(int32_t* out, int32_t cap);       /* returns the number of devices of the list */
int hs_cv_run_host(const uint8_t* h_contig_seq, const int64_t* h_contig_off, int32_t n_contigs,
                   const uint8_t* h_read_seq, const int64_t* h_read_off, int32_t n_reads,
                   const int32_t* h_rec_read, const int32_t* h_rec_pos, const uint8_t* h_rec_strand,
                   const int64_t* h_rec_cig_off, const uint32_t* h_cigar, const int32_t* h_contig_rec_off,
                   float automatic_snp_threshold, int32_t n_threads, hs_cv_result** out);

/* hs_cv_run in two steps, for hosts that overlap the sequential glue of several contig groups:
 * hs_cv_select = the streaming pass over the WHOLE batch (K0 CIGAR scan, K1 pileup, K2 column statistics; one launch each),
 * hs_cv_run_range = everything after it (K3, partitions, K4, merge) for the contigs [c0, c1) of the batch. Contigs are
 * independent (call_variants.cpp:1280): ranges may run concurrently from different host threads; the result of a range
 * holds c1 - c0 contigs and its error_rate covers that range only (form the job-wide mean from mean_distance). */
typedef struct hs_cv_selection {
    int64_t n_selected;        /* positions whose second allele count makes them worth extracting */
    float t_kernel_ms[4];      /* hipEvent time of k_pileup, k_column_stats, -, k_cigar_scan */
    double t_device_ms, t_host_ms;
    void* impl;
} hs_cv_selection;
int hs_cv_select(hs_cv_batch* b, hs_cv_selection** out);
int hs_cv_run_range(hs_cv_batch* b, const hs_cv_selection* sel, int32_t c0, int32_t c1, float automatic_snp_threshold,
                    int32_t n_threads, hs_cv_result** out);
void hs_cv_selection_destroy(hs_cv_selection* sel);
void hs_cv_result_destroy(hs_cv_result* r);

/* Stage 4 on SNP columns already in memory (what parse_column_file, separate_reads.cpp:46-190, yields). */
typedef struct hs_sr_contig {
    int64_t length;            /* contig length */
    int32_t n_reads;           /* number of READ lines */
    const int32_t* read_start; /* [n_reads] readLimits .first */
    const int32_t* read_end;   /* [n_reads] readLimits .second */
    int32_t n_snps;
    const int32_t* snp_pos;
    const uint8_t* snp_ref;
    const uint8_t* snp_alt;
    const int64_t* col_off;    /* [n_snps+1] offsets into col_idx / col_code (need not start at 0) */
    const int32_t* col_idx;
    const uint8_t* col_code;
    int32_t ploidy;            /* 0 = unlimited (separate_reads.cpp:1454-1458) */
} hs_sr_contig;

typedef struct hs_sr_result {
    int32_t n_contigs;
    int64_t* win_off;          /* [C+1] windows per contig */
    int32_t* win_start;        /* [W] */
    int32_t* win_end;          /* [W] */
    int64_t* label_off;        /* [W+1] offsets into labels (n_reads of the contig each) */
    int32_t* labels;           /* -2 absent, -1 unclustered, >=0 group */
    double t_device_ms;
    double t_host_ms;
    int64_t n_cw_instances;
    float t_kernel_ms[4];      /* hipEvent time of k_simdiff and of the three k_chinese_whispers waves */
    float t_kernel_graph_ms;   /* hipEvent time of k_read_graph_rows */
    int64_t n_graph_rows_host; /* graph rows whose neighbour cut-off depended on std::sort's order of equal keys */
    int64_t n_windows_finished_on_host;   /* clustering windows whose cluster merging (K8) fell back to the host code */
    /* counts behind the whole-path roofline of SURVEY.md 8(d): emitted by the kernels / the driver, not estimated */
    int64_t n_cw_sweeps;       /* Chinese-Whispers sweeps, summed over every run */
    int64_t cw_bytes;          /* sum over runs of sweeps * (4 * nnz + 8 * m), nnz / m = neighbour entries / nodes of the run's window graph */
    int64_t graph_nnz;         /* neighbour entries of all window graphs */
    int64_t n_graph_rows;      /* rows (window, masked read) of all window graphs */
    int64_t simdiff_bytes;     /* sum over matrix-path contigs of N * S / 4 + 16 * N * N */
} hs_sr_result;

int hs_sr_run(const hs_sr_contig* contigs, int32_t n_contigs, int32_t window_size, float error_rate,
              int32_t low_memory, uint32_t seed, int32_t n_threads, hs_sr_result** out);

/* Test taps of the clustering chain of stage 4 exactly as hs_sr_run queues it: for every window that has seeding SNPs (chain order) its
 * contig, first position and reads, and what the kernels of the chain left -- the labels of every per-SNP Chinese-Whispers run
 * (separate_reads.cpp:1674-1705: k_cw_seed_sets + k_cw_seeded_lanes / k_cw_seeded_rows / k_cw_seeded_wave) as READ indices, and the labels of
 * the run behind finalize_clustering's small-cluster filter (:924-970, inside k_window_tail). The finished labels are in the result. */
typedef struct hs_sr_taps {
    int32_t n_windows;
    int32_t* win_contig; int32_t* win_start;      /* [n_windows] */
    int64_t* win_row0;                            /* [n_windows + 1] into mask_ids / third */
    int32_t* mask_ids;                            /* the window's reads, ascending */
    int64_t* run_begin;                           /* [n_windows + 1] the window's per-SNP runs */
    int32_t* run_snp;                             /* SNP index on its contig of every run */
    int64_t* run_off;                             /* [runs + 1] into run_labels: m labels per run, in the order of mask_ids */
    int32_t* run_labels;                          /* label = the read index the label stands for */
    int32_t* third;                               /* cluster index (-1: dropped with its small cluster) */
} hs_sr_taps;
int hs_sr_run_taps(const hs_sr_contig* contigs, int32_t n_contigs, int32_t window_size, float error_rate, int32_t low_memory, uint32_t seed,
                   int32_t n_threads, hs_sr_result** out, hs_sr_taps** taps);
void hs_sr_taps_destroy(hs_sr_taps* t);
void hs_sr_result_destroy(hs_sr_result* r);
/* Stage 4 directly on the result of hs_cv_run, without the .col text round trip (call_variants.cpp:1197-1204 <->
 * separate_reads.cpp:84-170). READ limits come from the records (input_output.cpp:503-511); SNPs whose second base is
 * rarer than rarest_strain_abundance are dropped as parse_column_file does (separate_reads.cpp:167). window_size <= 0:
 * computed as separate_reads.cpp:1466-1498 from this batch. */
int hs_sr_run_cv_range(const hs_cv_batch* b, int32_t c0, int32_t c1, const hs_cv_result* cv, float error_rate,
                       float rarest_strain_abundance, int32_t low_memory, int32_t amplicon, uint32_t seed, int32_t n_threads,
                       int32_t window_size, hs_sr_result** out);   /* cv = result of hs_cv_run_range(b, sel, c0, c1) */
int hs_sr_run_cv(const hs_cv_batch* b, const hs_cv_result* cv, float error_rate, float rarest_strain_abundance,
                 int32_t low_memory, int32_t amplicon, uint32_t seed, int32_t n_threads, int32_t window_size,
                 hs_sr_result** out);
/* separate_reads.cpp:1466-1498: window size from the read limits of all contigs of the .col */
/* ------------------------------------------------------------------------------------------------
 * Both stages over one resident batch with the contigs processed as n_groups consecutive ranges on persistent host threads
 * (one HIP stream and one worker pool each). hs_pipeline_select runs hs_cv_select (the streaming kernels, once over the whole
 * batch) and returns the per-contig mean distances, which only need K1's per-record counters; the caller forms the error
 * rate from them (job-wide, contig order, call_variants.cpp:1312-1315,1377 -- across processes if the job is sharded);
 * hs_pipeline_run then takes every group through hs_cv_run_range and hs_sr_run_cv_range back to back, no barrier between the
 * stages, and concatenates the results in contig order. Equivalent to hs_cv_run + hs_sr_run_cv on the whole batch.
 * ---------------------------------------------------------------------------------------------- */
typedef struct hs_pipeline hs_pipeline;
typedef struct hs_pipeline_stats {
    int64_t n_snps, n_cw_instances, n_graph_rows_host;
    double t_device_ms, t_host_ms;      /* summed over the groups */
    float t_kernel_cv_ms[4];            /* k_pileup, k_column_stats, k_gather_columns (+top3), k_cigar_scan */
    float t_kernel_k4_ms;
    float t_kernel_sr_ms[4];            /* k_simdiff and the three k_chinese_whispers waves */
    float t_kernel_graph_ms;
    int64_t n_columns_extracted, n_columns_downloaded, n_columns_downloaded_late;   /* see hs_cv_result */
    int64_t n_cw_sweeps, cw_bytes, graph_nnz, n_graph_rows, simdiff_bytes;          /* see hs_sr_result */
} hs_pipeline_stats;
int hs_pipeline_create(hs_cv_batch* b, int32_t n_groups, hs_pipeline** out);
int hs_pipeline_select(hs_pipeline* p, float* mean_distance /* [C] out */, hs_pipeline_stats* stats);
/* window_size <= 0: chosen over the whole batch as separate_reads.cpp:1466-1498 does over the whole .col file */
int hs_pipeline_run(hs_pipeline* p, float automatic_snp_threshold, float error_rate, float rarest_strain_abundance, int32_t low_memory,
                    int32_t amplicon, uint32_t seed, int32_t n_threads, int32_t window_size, hs_sr_result** out,
                    hs_pipeline_stats* stats);
/* hs_pipeline_select + the job-wide error rate + hs_pipeline_run in one call, for a job that lives in ONE process: every contig
 * group brings up its own share of the pileup (one group at a time), the error rate (call_variants.cpp:1312-1315, then %g and
 * the 0.15 cap of hairsplitter.py:686-692,725) is formed inside when the last group has its counters. mean_distance
 * [n_contigs] and error_rate_out are optional outputs. Same results as the two calls. */
int hs_pipeline_run_fused(hs_pipeline* p, float automatic_snp_threshold, float rarest_strain_abundance, int32_t low_memory, int32_t amplicon,
                          uint32_t seed, int32_t n_threads, int32_t window_size, float* mean_distance, float* error_rate_out, hs_sr_result** out,
                          hs_pipeline_stats* st);
/* the HIP device each contig-group thread of the pipeline is bound to (== hs_cv_batch_device of its batch); returns the number
 * of groups, fills at most cap entries */
int hs_pipeline_thread_devices(hs_pipeline* p, int32_t* out, int32_t cap);
/* What a pipeline call leaves with the host besides the labels. Stage 3's SNP columns are handed to stage 4 on the device; with
 * HS_PIPELINE_KEEP_COLUMNS != 0 their entries (.col's payload: read indices and codes of every SNP column, the SNPS lines of
 * call_variants.cpp:1184-1204) are ALSO brought to the host inside the call, and the groups' stage-3 results stay available through
 * hs_pipeline_group_cv until the next call on the pipeline (they are dropped when it starts; their col_idx / col_code point into pinned
 * blocks of the pipeline -- hs_cv_result::entries_borrowed -- and are not the caller's to free). Default 0: positions, alleles and
 * counts of the SNPs only. */
#define HS_PIPELINE_KEEP_COLUMNS 1
/* HS_PIPELINE_SPARSE_LABELS != 0: the labels come back per window as the reads the window holds and their labels -- what the GROUP lines
 * of the .gro list (separate_reads.cpp:1756-1784), nothing else: hs_sr_result::labels is NULL (label_off still describes the dense
 * form) and hs_pipeline_sparse_labels returns, valid until the next call on the pipeline, win_row_off [W+1], ids [rows] (ascending
 * read indices inside a window) and labels [rows]; every read a window does not list has the label -2. */
#define HS_PIPELINE_SPARSE_LABELS 2
int hs_pipeline_set_option(hs_pipeline* p, int32_t option, int64_t value);
int hs_pipeline_groups(const hs_pipeline* p);                                  /* number of contig groups */
int hs_pipeline_group_range(const hs_pipeline* p, int32_t g, int32_t* c0, int32_t* c1);   /* contigs [c0, c1) of group g */
int hs_pipeline_sparse_labels(const hs_pipeline* p, const int64_t** win_row_off, const int32_t** ids, const int32_t** labels, int64_t* n_windows, int64_t* n_rows);
const hs_cv_result* hs_pipeline_group_cv(const hs_pipeline* p, int32_t g);    /* stage-3 result of group g from the last call (NULL before the first) */
void hs_pipeline_destroy(hs_pipeline* p);

int32_t hs_sr_window_size(const hs_sr_contig* contigs, int32_t n_contigs, int32_t amplicon);

/* File-level entry points == main() of the two reference executables (same argv, same exit codes).
 * call_variants.cpp:1215-1385 and separate_reads.cpp:1398-1790. */
int hs_call_variants_main(int argc, char** argv);
/* What HS_call_variants may still do once hs_call_variants_main has returned 0 and its outputs are complete: stage 4 for the arguments
 * hairsplitter.py passes by default (hairsplitter.py:686-692,725-726), in the process that still holds the job's device state; the
 * .gro is left as <out.col>.hsgro and hs_separate_reads_main copies it ONLY if it is called with that .col (size and block hashes
 * of the .hsbin companion) and exactly those arguments (error rate, rarest strain abundance 0.01, low memory 0, the same amplicon
 * switch, HS_SEED, no ploidies); anything else runs stage 4 as always. The executable calls it after it has reported its exit
 * status (csrc/hs_dropin_main.h). HS_NO_PRECOMPUTE=1: neither written nor read. No-op without a preceding successful main. */
void hs_call_variants_epilogue(void);
int hs_separate_reads_main(int argc, char** argv);
/* for executables that _exit right after one of the two: nothing is torn down at the end (parsed inputs, results, device state) */
void hs_main_process_exits(int yes);

/* ------------------------------------------------------------------------------------------------
 * Next stage, consumer side of the .gro (host code, no device work): how every read threads through the contigs that the
 * read separation implies, as a GAF.  Replaces parse_split_file + merge_intervals (+ stitch) + find_paths + output_GAF
 * (create_new_contigs.cpp:41-175, :1427-1534, :833-903, :959-1112, :1128-1419) as called from its main (:1582-1590).
 * hs_gaf_from_files reads the same four files the reference's HS_create_new_contigs reads; hs_gaf_from_labels takes the
 * windows and labels of an hs_sr_result instead of the .gro text (contig_has_snps[c] = 0 for the contigs the .gro writer
 * skips, separate_reads.cpp:1522-1524; NULL = the contigs without windows, which is what hs_sr_run leaves them with).  hs_gro_to_gaf_main: argv = <gfa> <reads> <sam> <gro> <amplicon> <out.gaf> [threads].
 * ---------------------------------------------------------------------------------------------- */
int hs_gaf_from_files(const char* gfa, const char* reads, const char* sam, const char* gro, int32_t amplicon, const char* out_gaf,
                      int32_t n_threads);
int hs_gaf_from_labels(const char* gfa, const char* reads, const char* sam, int32_t amplicon, int32_t n_contigs,
                       const int64_t* win_off, const int32_t* win_start, const int32_t* win_end, const int64_t* label_off,
                       const int32_t* labels, const uint8_t* contig_has_snps, const char* out_gaf, int32_t n_threads);
int hs_gro_to_gaf_main(int argc, char** argv);

/* ------------------------------------------------------------------------------------------------
 * Upstream feeders of stage 3 that are plain text transforms (host code): the 300 kb cutter the orchestrator runs before the
 * reads are aligned (src/cut_gfa.py:33-66, hairsplitter.py:583) and the GFA -> FASTA converter (src/gfa2fa.cpp).
 * hs_cut_gfa_main: argv of cut_gfa.py (--assembly/-a, --length/-l, --output/-o); hs_gfa2fa_main: argv[1] = gfa, FASTA on
 * standard output. Byte-identical to the reference's outputs (tests/golden/gfa_tools).
 * ---------------------------------------------------------------------------------------------- */
int hs_cut_gfa(const char* gfa_in, int64_t length, const char* gfa_out);
int hs_gfa_to_fasta(const char* gfa_in, const char* fasta_out /* NULL or "-": standard output */);
int hs_cut_gfa_main(int argc, char** argv);
int hs_gfa2fa_main(int argc, char** argv);

#ifdef __cplusplus
}
#endif
#endif /* HAIRSPLITTER_HIP_H */
