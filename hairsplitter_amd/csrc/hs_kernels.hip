// hs_kernels.hip -- hand-written CDNA4 (gfx950, wave64) kernels of the HairSplitter hot path.
// Integer / bitwise work bounded by HBM: no MFMA. One wavefront owns one unit of sequential work
// (an alignment record, a Chinese-Whispers instance, a sequence pair); lanes cover the data-parallel axis.
// Reference semantics are cited per kernel (paths relative to /root/reference/src).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hs_device.h"
#include "hs_rh8.h"

namespace hsdev {

static __device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
// index of the wavefront in its workgroup, declared wave-uniform to the compiler: everything derived from it (the unit of
// work, its metadata, loop bounds) then lives in SGPRs, loads through the scalar cache and branches on SCC instead of EXEC
static __device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// wave-wide max over a 64-bit key (6 butterfly steps through ds_bpermute)
static __device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        unsigned long long o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}
// wave64 inclusive prefix sum on the DPP network (row_shr within 16-lane rows, then row_bcast across rows): six
// v_add_u32_dpp, no LDS round trips (hipcc lowers __shfl_* to ds_bpermute, ~10x the latency)
static __device__ __forceinline__ int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return v;
}
static __device__ __forceinline__ int wave_max_i32(int v) {
    int o;
    o = __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false); v = o > v ? o : v;
    return __builtin_amdgcn_readlane(v, 63);
}
static __device__ __forceinline__ int wave_sum_i32(int v) { return __builtin_amdgcn_readlane(wave_scan_incl(v), 63); }
// value of the lane to the left (lane 0 receives `fill`)
static __device__ __forceinline__ int wave_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
// x clamped into [0, hi] (hi >= 0, wave-uniform): one v_med3_i32
static __device__ __forceinline__ int clamp_i32(int x, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
}
static __device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------
// K0 CIGAR scan: per alignment record, the running (event, read, reference) offsets at every 64-op chunk
// boundary, so that K1 can start in the middle of a record. One wavefront per record, lane = op, wave-level
// inclusive scans. Also writes the record's final reference cursor (call_variants.cpp:354) and zeroes its
// error/length counters. tools.cpp:27-57 (convert_cigar) never materialises: ops stay run-length encoded.
// ------------------------------------------------------------------------------------------------
struct OpAdv { int ev, rd, rf; };
static __device__ __forceinline__ OpAdv op_advances(uint32_t op, bool in_range) {
    OpAdv a; a.ev = 0; a.rd = 0; a.rf = 0;
    if (!in_range) return a;
    const int len = (int)(op >> 4), code = (int)(op & 15u);
    const bool isM = code == 0 || code == 7 || code == 8;
    if (isM || code == 1 || code == 2) a.ev = len;
    if (isM || code == 1 || code == 4 || code == 5) a.rd = len;
    if (isM || code == 2) a.rf = len;
    return a;
}

__global__ __launch_bounds__(256) void k_cigar_scan(
    const int64_t* __restrict__ contig_off, const int32_t* __restrict__ rec_contig, const int32_t* __restrict__ rec_pos,
    const int64_t* __restrict__ rec_cig_off, const uint32_t* __restrict__ cigar, const int64_t* __restrict__ rec_chunk_off,
    int n_rec, int32_t* __restrict__ chunk_start /* [n_chunks][4] */, int32_t* __restrict__ rec_stats) {
    const int lane = lane_id();
    const int r = (int)blockIdx.x * 4 + wave_id();
    if (r >= n_rec) return;
    const int64_t cig0 = rec_cig_off[r], cig1 = rec_cig_off[r + 1];
    int32_t* __restrict__ cs = chunk_start + 4 * rec_chunk_off[r];
    const int pos = rec_pos[r];
    int ev_cur = 0, t_cur = 0, q_cur = pos;
    int k = 0;
    // smallest number of events that precede an op which moves a cursor without owning events (S, H, N) and has events before
    // it: if events also follow it, the record is not one run of M/I/D events (see k_pileup_packed)
    int first_gap = 0x7fffffff;
    // what the two events before the chunk were (k_pileup_runs forms the 3-mer context of an op's first events from it):
    // 0 = an event that consumed a read base (M / I), 1 = a deletion, 2 / 3 = none yet: the initial 'G' / 'C' of call_variants.cpp:212-214
    int ctx1 = 2, ctx2 = 3;
    for (int64_t ob = cig0; ob < cig1; ob += 64, ++k) {
        const int64_t oi = ob + lane;
        const uint32_t opw = oi < cig1 ? cigar[oi] : 0u;
        const OpAdv a = op_advances(opw, oi < cig1);
        const int ev_incl = wave_scan_incl(a.ev);
        const int ev = __builtin_amdgcn_readlane(ev_incl, 63), rd = wave_sum_i32(a.rd), rf = wave_sum_i32(a.rf);
        const bool gap = a.ev == 0 && (a.rd > 0 || a.rf > 0);
        if (__ballot(gap)) {   // wave-uniform; clips sit in the first and last chunk of a record
            const int before = ev_cur + ev_incl - a.ev;
            const int cand = (gap && before > 0) ? before : 0x7fffffff;
            const int m = -wave_max_i32(-cand);
            first_gap = m < first_gap ? m : first_gap;
        }
        // bit 0 of the 4th slot of the record's FIRST chunk: the record is not one run of events (set below); bits 4..7: ctx1 | ctx2 << 2
        if (lane == 0) { cs[4 * k + 0] = ev_cur; cs[4 * k + 1] = t_cur; cs[4 * k + 2] = q_cur; cs[4 * k + 3] = (ctx1 | (ctx2 << 2)) << 4; }
        ev_cur += ev; t_cur += rd; q_cur += rf;
        {   // the last two events of the chunk (wave-uniform: three ballots)
            const unsigned long long nz = __ballot(a.ev > 0), dm = __ballot(a.ev > 0 && (opw & 15u) == 2u), lg = __ballot(a.ev > 1);
            if (nz) {
                const int l1 = 63 - __builtin_clzll(nz);
                const int t1 = (int)((dm >> l1) & 1ull);
                if ((lg >> l1) & 1ull) { ctx1 = t1; ctx2 = t1; }
                else {
                    const unsigned long long below = nz & ((1ull << l1) - 1ull);
                    ctx2 = below ? (int)((dm >> (63 - __builtin_clzll(below))) & 1ull) : ctx1;
                    ctx1 = t1;
                }
            }
        }
    }
    if (lane == 0 && k > 0 && first_gap < ev_cur) cs[3] |= 1;   // 4th slot of the record's first chunk entry
    if (lane == 0) {
        const int ctg = rec_contig[r];
        const int L = (int)(contig_off[ctg + 1] - contig_off[ctg]);
        rec_stats[4 * r + 0] = pos >= L ? pos : (q_cur < L ? q_cur : L);
        rec_stats[4 * r + 1] = 0; rec_stats[4 * r + 2] = 0; rec_stats[4 * r + 3] = ev_cur;
    }
}

// ------------------------------------------------------------------------------------------------
// K1 pileup: generate_msa, call_variants.cpp:189-354. One wavefront per TASK = a fixed-size range of
// alignment events of one record (balanced: a 60 kb read is 15+ tasks, not one long wave). The task finds its
// 64-op chunk by bisection on K0's table, then walks chunks: wave inclusive scan of the ops (lane = op), events
// processed 64 per step (lane = event), owning op by a 6-step binary search across lanes. The 3-mer context
// (previous two emitted characters, insertions and deletions included) comes from the two lanes to the left or a
// wave-uniform carry; a task warms the carry up by replaying the two events before its range without
// committing them. Pileup writes are contiguous per run of M/D events (coalesced).
// ------------------------------------------------------------------------------------------------
#ifndef HS_K1_PU
#define HS_K1_PU 4
#endif
// per-wave LDS of the per-event form: the 64 ops of the current chunk (first event, read offset, reference offset, op code),
// the lanes of the ops that own events, and one "an op starts here" flag per event of the current 64-event window
struct PileupLds {
    int4 op[4][64];
    uint8_t nzlane[4][64];
    uint8_t flag[4][HS_K1_PU][64];
};

static __device__ __forceinline__ void pileup_task_per_event(
    PileupLds& lds, const int lane, const int wv, const int r, const int e0, const int ev_per_task,
    const uint8_t* __restrict__ contig_seq, const int64_t* __restrict__ contig_off,
    const uint8_t* __restrict__ read_seq, const int64_t* __restrict__ read_off,
    const int32_t* __restrict__ rec_read, const int32_t* __restrict__ rec_contig,
    const int32_t* __restrict__ rec_pos, const uint8_t* __restrict__ rec_strand,
    const int64_t* __restrict__ rec_cig_off, const uint32_t* __restrict__ cigar,
    const int64_t* __restrict__ pile_off, const int64_t* __restrict__ rec_chunk_off,
    const int32_t* __restrict__ chunk_start, uint8_t* __restrict__ pile, int32_t* __restrict__ rec_stats) {
    constexpr int PU = HS_K1_PU;
    auto& s_op = lds.op; auto& s_nzlane = lds.nzlane; auto& s_flag = lds.flag;
    const int e1 = e0 + ev_per_task;
    const int e_first = e0 >= 2 ? e0 - 2 : 0;

    const int ctg = rec_contig[r];
    const int64_t coff = contig_off[ctg];
    const int L = (int)(contig_off[ctg + 1] - coff);
    const int rd = rec_read[r];
    const int64_t roff = read_off[rd];
    const int rlen = (int)(read_off[rd + 1] - roff);
    const int pos = rec_pos[r];
    const bool fwd = rec_strand[r] != 0;
    const int64_t cig0 = rec_cig_off[r], cig1 = rec_cig_off[r + 1];
    const int n_chunks = (int)((cig1 - cig0 + 63) >> 6);
    const int32_t* __restrict__ cs = chunk_start + 4 * rec_chunk_off[r];
    uint8_t* __restrict__ out = pile + pile_off[r];
    const uint8_t* __restrict__ ctgp = contig_seq + coff;
    const uint8_t* __restrict__ rdp = read_seq + roff;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // reverse-strand records read the reverse complement: index rlen - 1 - t == (t ^ -1) + rlen, base 3 - b == b ^ 3
    const int rd_xor = fwd ? 0 : -1, rd_add = fwd ? 0 : rlen, rd_cmpl = fwd ? 0 : 3;

    // last chunk whose first event is <= e_first (uniform bisection)
    int klo = 0, khi = n_chunks - 1;
    while (klo < khi) { const int mid = (klo + khi + 1) >> 1; if (cs[4 * mid] <= e_first) klo = mid; else khi = mid - 1; }

    int p1 = 2, p2 = 1;   // previous char 'G', the one before 'C' (call_variants.cpp:212-214 after one shift)
    int nerr = 0, nlen = 0;
    for (int k = klo; k < n_chunks; ++k) {
        const int ev_base = cs[4 * k + 0];
        if (ev_base >= e1) break;
        const int t_cur = cs[4 * k + 1], q_cur = cs[4 * k + 2];
        const int64_t oi = cig0 + ((int64_t)k << 6) + lane;
        const bool in_range = oi < cig1;
        const uint32_t op = in_range ? cigar[oi] : 0xFu;
        const int code = in_range ? (int)(op & 15u) : 15;
        const OpAdv a = op_advances(op, in_range);
        const int ev_inc = wave_scan_incl(a.ev), rd_inc = wave_scan_incl(a.rd), rf_inc = wave_scan_incl(a.rf);
        const int ev_ex = ev_inc - a.ev;
        const int chunk_ev = __builtin_amdgcn_readlane(ev_inc, 63);
        const bool nz = a.ev > 0;
        const unsigned long long nzmask = __ballot(nz);
        wave_lds_sync();   // the previous chunk's readers are done
        // op record: first event, read offset, reference offset, class (bit 0: the op consumes the reference (M/=/X/D), bit 1: deletion)
        const bool isM = code == 0 || code == 7 || code == 8;
        s_op[wv][lane] = make_int4(ev_ex, t_cur + rd_inc - a.rd, q_cur + rf_inc - a.rf, (isM ? 1 : 0) | (code == 2 ? 3 : 0));
        if (nz) s_nzlane[wv][__popcll(nzmask & lt_mask)] = (uint8_t)lane;
        const int lo_el = e_first > ev_base ? e_first - ev_base : 0;
        const int hi_el = (e1 - ev_base) < chunk_ev ? (e1 - ev_base) : chunk_ev;
        const int lo_commit = e0 - ev_base;                                   // events before it only warm the 3-mer context up
        const unsigned span_commit = hi_el > lo_commit ? (unsigned)(hi_el - lo_commit) : 0u;

        // K1 is bound by VALU issue (a wave64 instruction takes four cycles per SIMD): four 64-event windows are decoded per
        // iteration so that their eight loads are in flight together, predicates are kept as wave masks (counters are scalar
        // popcounts of ballots), and the owner lookup shares one flag clear / one flag scatter between the four windows.
        uint8_t* const flags = &s_flag[wv][0][0];
        for (int eb = lo_el; eb < hi_el; eb += 64 * PU) {
#pragma unroll
            for (int z = 0; z < PU / 4; ++z) reinterpret_cast<uint32_t*>(flags)[lane + 64 * z] = 0u;   // 64 lanes x 4 B = 256 flags each
            wave_lds_sync();
            {
                const int rel = ev_ex - eb;
                if (nz && (unsigned)rel < (unsigned)(64 * PU)) flags[rel] = 1;   // an op starts at this event
            }
            wave_lds_sync();
            int before = __popcll(__ballot(nz && ev_ex < eb));                // ops (owning events) that start before the iteration
            int q_[PU], c_[PU], cref_[PU];
            bool act_[PU], wr_[PU], jm_[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                // owner of every event of the window: the (ops started before the window + flags at or left of it)-th op that
                // owns events
                const int w0 = eb + 64 * u;
                const int own = (int)flags[64 * u + lane];                    // 1 when an op starts at this event
                const unsigned long long fm = __ballot(own != 0);
                const int e = w0 + lane;
                // flags strictly left of the lane (v_mbcnt) + its own + ops started before the window - 1
                // (never negative for an event of the chunk: the op that owns it has started at or before it; lanes past the end of
                // the chunk are masked below and may read any slot of the table, which holds 64 entries)
                const int rank = (before - 1 + own + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u))) & 63;
                before += __popcll(fm);
                const int4 od = s_op[wv][s_nzlane[wv][rank]];
                const int off = e - od.x;
                const bool refc = (od.w & 1) != 0, jD = (od.w & 2) != 0;
                const int t = od.y + off;
                const int q = od.z + (refc ? off : 0);
                // committed: inside the task's event range and on the contig (call_variants.cpp:217)
                act_[u] = (unsigned)(e - lo_commit) < span_commit && (unsigned)q < (unsigned)L;
                wr_[u] = refc; jm_[u] = od.w == 1; q_[u] = q;
                // both loads are unconditional (indices clamped into the read / the contig) so that they issue back to back
                const int tt = clamp_i32(t, rlen - 1);                         // the host validates CIGAR vs read length
                // scalar base + unsigned 32-bit lane offset: no 64-bit address arithmetic per lane
                const int bb = (int)rdp[(unsigned)((tt ^ rd_xor) + rd_add)];   // forward: tt, reverse: rlen - 1 - tt
                cref_[u] = (int)ctgp[(unsigned)clamp_i32(q, L - 1)];
                c_[u] = jD ? 4 : (bb ^ rd_cmpl);                               // 4 == '-'; reverse strand reads the complement
            }
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int w0 = eb + 64 * u;
                if (w0 >= hi_el) break;                                      // wave-uniform
                const int c = c_[u];
                const int cu1 = wave_shr1(c, p1);
                const int cu2 = wave_shr1(cu1, p2);
                // M: call_variants.cpp:238-240,254-256; D: :287-290; I: :337 (no column written)
                nlen += __popcll(__ballot(act_[u]));
                nerr += __popcll(__ballot(act_[u] && !(jm_[u] && c == cref_[u])));
                if (act_[u] && wr_[u]) out[(unsigned)(q_[u] - pos)] = (uint8_t)(33 + 5 * cu2 + cu1 + 25 * c);
                const int nv = (hi_el - w0) < 64 ? (hi_el - w0) : 64;
                const int last = __builtin_amdgcn_readlane(c, nv - 1);
                const int last2 = nv >= 2 ? __builtin_amdgcn_readlane(c, nv - 2) : p1;
                p2 = last2; p1 = last;
            }
        }
    }
    if (lane == 0 && nlen > 0) {   // nlen / nerr are wave-uniform (scalar popcounts)
        atomicAdd(&rec_stats[4 * r + 1], nerr);
        atomicAdd(&rec_stats[4 * r + 2], nlen);
    }
}

// unaligned vector types and packed-byte helpers shared by K1 (run form) and K2
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint16_t __attribute__((aligned(1))) u16_unaligned;
typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
// keeps the compiler from folding a chain of shifts and adds back into a 32-bit multiply (quarter rate on the VALU)
static __device__ __forceinline__ uint32_t opaque(uint32_t x) { asm("" : "+v"(x)); return x; }
// 0x01 bytes -> 0xff bytes
static __device__ __forceinline__ uint32_t bytes_ff(uint32_t x01) { return opaque(x01 << 8) - x01; }
static __device__ __forceinline__ uint32_t byte_sum(uint32_t x, uint32_t acc) { return __builtin_amdgcn_sad_u8(x, 0u, acc); }
// bytes 0 .. n-1 set (n in 0..4)
static __device__ __forceinline__ uint32_t low_bytes(int n) { return n >= 4 ? 0xffffffffu : ((1u << (8 * n)) - 1u); }

#include "hs_kernels_runs.inc"      // K1, run form (round 5): lanes = 16-event pieces of one CIGAR op

// the records k_pileup_runs leaves out (K0 flagged them: a clip or a skip between aligned bases), in the per-event form.
// A small persistent grid: every wave checks 64 records per step and walks the tasks of the flagged ones.
__global__ __launch_bounds__(256) void k_pileup_flagged_records(
    const uint8_t* __restrict__ contig_seq, const int64_t* __restrict__ contig_off,
    const uint8_t* __restrict__ read_seq, const int64_t* __restrict__ read_off,
    const int32_t* __restrict__ rec_read, const int32_t* __restrict__ rec_contig,
    const int32_t* __restrict__ rec_pos, const uint8_t* __restrict__ rec_strand,
    const int64_t* __restrict__ rec_cig_off, const uint32_t* __restrict__ cigar,
    const int64_t* __restrict__ pile_off, const int64_t* __restrict__ rec_chunk_off,
    const int32_t* __restrict__ chunk_start, int n_rec, int ev_per_task, uint8_t* __restrict__ pile, int32_t* __restrict__ rec_stats) {
    __shared__ PileupLds lds;
    const int lane = lane_id();
    const int wv = wave_id();
    const int n_waves = (int)gridDim.x * 4;
    for (int base = ((int)blockIdx.x * 4 + wv) * 64; base < n_rec; base += n_waves * 64) {
        const int rr = base + lane;
        bool flagged = false;
        if (rr < n_rec && rec_chunk_off[rr + 1] > rec_chunk_off[rr]) flagged = (chunk_start[4 * rec_chunk_off[rr] + 3] & 1) != 0;
        unsigned long long m = __ballot(flagged);
        while (m) {
            const int r = base + __builtin_ctzll(m);
            m &= m - 1ull;
            const int n_ev = rec_stats[4 * r + 3];
            for (int e0 = 0; e0 < n_ev; e0 += ev_per_task)
                pileup_task_per_event(lds, lane, wv, r, e0, ev_per_task, contig_seq, contig_off, read_seq, read_off, rec_read, rec_contig, rec_pos,
                                      rec_strand, rec_cig_off, cigar, pile_off, rec_chunk_off, chunk_start, pile, rec_stats);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2 column statistics: the histogram of call_variants.cpp:477-501 for 256 consecutive positions per
// workgroup. Every lane owns one position and a private 125-bin u16 histogram column in LDS
// (hist[bin][lane]: lanes hitting the same bin are conflict-free). The records overlapping the tile are first
// compacted into an LDS list (coalesced metadata loads + wave ballot), then walked four at a time so that the
// pileup byte loads of a group are in flight together. Output: five largest (count desc, code asc) + depth,
// one 16-B store per lane; positions whose second count reaches `min_second` are also appended to a compact
// selection list (wave-aggregated atomic), so the host never scans the per-position array.
// ------------------------------------------------------------------------------------------------
#define HS_NBINS 125
#define HS_LIST_CAP 512
#ifndef HS_K2_INFLIGHT
#define HS_K2_INFLIGHT 16
#endif
// CB = bytes per counter: 1 when no position of the batch is deeper than 255 reads (32 KiB of LDS per workgroup, 4-5
// workgroups per CU), 2 otherwise (63 KiB). Counters are packed 4 (or 2) per dword, dword-major ([word][lane]), so
// the final scan reads one dword per 4 bins and skips empty ones.
// FULL = false is the form the stage driver uses: it only needs the two largest counts, whether a third allele exists and
// the depth (the exact top-3 of the selected columns is recomputed on the host in the reference's tie order), which takes a
// handful of branch-free VALU instructions per dword of counters instead of a five-deep insertion per bin.
// The second half of K2, shared by both variants: scan of the lane's counters (two largest counts, third allele, depth -- or
// the five largest with their codes when FULL), the 16-B record, and the wave-aggregated append to the selection list.
template <int CB, bool FULL>
static __device__ __forceinline__ void column_stats_tail(const uint32_t* __restrict__ hw, int tid, int lane, int64_t g, int64_t total,
                                                         hs_colstat_dev* __restrict__ stats, int min_second, int32_t* __restrict__ sel_count,
                                                         int64_t* __restrict__ sel_gpos, int32_t* __restrict__ sel_depth, int sel_cap,
                                                         int64_t g_lo = 0, int64_t g_hi = 0x7fffffffffffffffll, int32_t* __restrict__ sel_ent = nullptr) {
    constexpr int PER_WORD = 4 / CB;
    constexpr int NWORDS = (HS_NBINS + PER_WORD - 1) / PER_WORD;
    int k0 = 0, k1 = 0, k2 = 0, k3 = 0;
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
    int depth = 0;
    if (!FULL) {
        if (g < total) {
            // two largest counters over all bins with packed 16-bit maxima: the bytes of a word are split over two registers of
            // two 16-bit fields; every field keeps its own (largest, runner-up); the four pairs are merged at the end
            typedef unsigned short us2 __attribute__((ext_vector_type(2)));
            us2 m0a = {0, 0}, m1a = {0, 0}, m0b = {0, 0}, m1b = {0, 0};
#pragma unroll 4
            for (int w = 0; w < NWORDS; ++w) {
                const uint32_t word = hw[w * 256 + tid];
                us2 va, vb;
                if (CB == 1) {
                    depth = (int)__builtin_amdgcn_sad_u8(word, 0u, (uint32_t)depth);      // sum of the four byte counters
                    va = __builtin_bit_cast(us2, word & 0x00ff00ffu);
                    vb = __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, word, 0x0c030c01u));      // bytes 1 and 3, zero-extended: one v_perm
                } else {
                    depth += (int)(word & 0xffffu) + (int)(word >> 16);
                    va = __builtin_bit_cast(us2, word);
                    vb = us2{0, 0};
                }
                const us2 la = __builtin_elementwise_min(va, m0a);
                m1a = __builtin_elementwise_max(m1a, la);
                m0a = __builtin_elementwise_max(m0a, va);
                if (CB == 1) {
                    const us2 lb = __builtin_elementwise_min(vb, m0b);
                    m1b = __builtin_elementwise_max(m1b, lb);
                    m0b = __builtin_elementwise_max(m0b, vb);
                }
            }
            const int tops[8] = {m0a.x, m1a.x, m0a.y, m1a.y, m0b.x, m1b.x, m0b.y, m1b.y};
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                const int v = tops[f];
                const int lo2 = v < c0 ? v : c0;     // branch-free two largest
                c1 = c1 > lo2 ? c1 : lo2;
                c0 = c0 > v ? c0 : v;
            }
        }
        c2 = depth > c0 + c1 ? 1 : 0;   // only "is there a third allele" is needed: a third non-empty bin <=> reads beyond the two largest counts
    } else if (g < total) {
        for (int w = 0; w < NWORDS; ++w) {
            const uint32_t word = hw[w * 256 + tid];
            if (word == 0u) continue;
#pragma unroll
            for (int f = 0; f < PER_WORD; ++f) {
                const int v = (int)((word >> (8 * CB * f)) & (CB == 1 ? 0xFFu : 0xFFFFu));
                if (v == 0) continue;
                depth += v;
                if (v > c4) {
                    const int key = w * PER_WORD + f + 33;
                    if (v > c0) { c4 = c3; c3 = c2; k3 = k2; c2 = c1; k2 = k1; c1 = c0; k1 = k0; c0 = v; k0 = key; }
                    else if (v > c1) { c4 = c3; c3 = c2; k3 = k2; c2 = c1; k2 = k1; c1 = v; k1 = key; }
                    else if (v > c2) { c4 = c3; c3 = c2; k3 = k2; c2 = v; k2 = key; }
                    else if (v > c3) { c4 = c3; c3 = v; k3 = key; }
                    else c4 = v;
                }
            }
        }
        uint4 o;
        o.x = (uint32_t)k0 | ((uint32_t)k1 << 8) | ((uint32_t)k2 << 16) | ((uint32_t)k3 << 24);
        o.y = (uint32_t)c0 | ((uint32_t)c1 << 16);
        o.z = (uint32_t)c2 | ((uint32_t)c3 << 16);
        o.w = (uint32_t)c4 | ((uint32_t)(depth > 65535 ? 65535 : depth) << 16);
        reinterpret_cast<uint4*>(stats)[g] = o;
    }
    if (sel_count) {
        // Selection: second count above the floor, or exactly at it with no third allele at all (the only way c1 > 5*c2 can
        // hold there). Every tile owns 256 slots of a scratch list and writes its selected positions there in lane order with
        // its count (no global atomic: ten thousand returning atomics on one counter cost more than the histogram itself);
        // k_selection_compact then packs the tiles in order, so the list comes out sorted by position.
        // [g_lo, g_hi): the positions of the caller's contig range (a boundary tile also holds positions of its neighbours);
        // sel_ent (optional): entries of the tile's selected columns = sum of their depths
        __shared__ int s_wc[4], s_we[4];
        const bool sel = g < total && g >= g_lo && g < g_hi && (c1 > min_second || (c1 == min_second && c2 == 0));
        const unsigned long long m = __ballot(sel);
        const int wv = tid >> 6;
        if (lane == 0) s_wc[wv] = __popcll(m);
        if (sel_ent) { const int we = wave_sum_i32(sel ? depth : 0); if (lane == 0) s_we[wv] = we; }
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wv; ++w) base += s_wc[w];
        const int64_t tile0 = (int64_t)blockIdx.x * 256;
        if (sel) {
            const int64_t slot = tile0 + base + __popcll(m & ((1ull << lane) - 1ull));
            sel_gpos[slot] = g; sel_depth[slot] = depth;
        }
        if (tid == 0) { sel_count[blockIdx.x] = s_wc[0] + s_wc[1] + s_wc[2] + s_wc[3]; if (sel_ent) sel_ent[blockIdx.x] = s_we[0] + s_we[1] + s_we[2] + s_we[3]; }
    }
}

// packs the per-tile selections (256 slots each, tile_base = exclusive prefix of the tile counts) into one sorted list
__global__ __launch_bounds__(256) void k_selection_compact(const int32_t* __restrict__ tile_cnt, const int64_t* __restrict__ tile_base,
                                                          const int64_t* __restrict__ scratch_gpos, const int32_t* __restrict__ scratch_depth,
                                                          int64_t n_tiles, int32_t* __restrict__ out_count, int64_t* __restrict__ out_gpos,
                                                          int32_t* __restrict__ out_depth, int out_cap) {
    const int64_t t = blockIdx.x;
    const int tid = (int)threadIdx.x;
    if (t == 0 && tid == 0) *out_count = (int32_t)tile_base[n_tiles];
    if (tid < tile_cnt[t]) {
        const int64_t o = tile_base[t] + tid;
        if (o < out_cap) { out_gpos[o] = scratch_gpos[t * 256 + tid]; out_depth[o] = scratch_depth[t * 256 + tid]; }
    }
}

template <int CB, bool FULL, bool PAD>
__global__ __launch_bounds__(256) void k_column_stats_tiled(
    const uint8_t* __restrict__ pile, const int64_t* __restrict__ tile_off, const int4* __restrict__ tile_ent, int64_t total,
    hs_colstat_dev* __restrict__ stats, int min_second, int32_t* __restrict__ sel_count, int64_t* __restrict__ sel_gpos,
    int32_t* __restrict__ sel_depth, int sel_cap, int64_t tile0 /* first tile of the launch: the selection scratch is indexed from it */,
    int64_t g_lo, int64_t g_hi, int32_t* __restrict__ sel_ent) {
    constexpr int PER_WORD = 4 / CB;
    constexpr int NWORDS = (HS_NBINS + PER_WORD - 1) / PER_WORD;
    __shared__ __attribute__((aligned(16))) uint32_t hw[NWORDS * 256];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int64_t tile = tile0 + (int64_t)blockIdx.x;
    const int64_t g = tile * 256 + tid;
    // a wavefront clears the counters of its own 64 positions, four words per store instruction (no workgroup barrier anywhere)
#pragma unroll
    for (int w4 = 0; w4 < NWORDS; w4 += 4) {
        const int w = w4 + (lane >> 4);
        if (w < NWORDS) *reinterpret_cast<uint4*>(&hw[w * 256 + (tid & ~63) + (lane & 15) * 4]) = make_uint4(0u, 0u, 0u, 0u);
    }
    char* const my_col = reinterpret_cast<char*>(hw) + tid * 4;      // counter word w of this position: my_col + 1024 w
    auto bump = [&](unsigned code, bool valid) {
        // bytes (or halves) of a word = consecutive codes: word code / PER_WORD, field code % PER_WORD
        if (valid) {
            const unsigned inc = 1u << ((code & (unsigned)(PER_WORD - 1)) * (unsigned)(8 * CB));
            uint32_t* at = reinterpret_cast<uint32_t*>(my_col + ((code & ~(unsigned)(PER_WORD - 1)) << (CB == 1 ? 8 : 9)));
            __hip_atomic_fetch_add(at, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    const int64_t e0 = tile_off[tile], e1 = tile_off[tile + 1];
    for (int64_t i0 = e0; i0 < e1; i0 += 64) {
        const int nrec = (e1 - i0) < 64 ? (int)(e1 - i0) : 64;
        const int4 held = tile_ent[i0 + (lane < nrec ? lane : nrec - 1)];   // lane j keeps record i0 + j
        // HS_K2_INFLIGHT records per step: K2 waits on the pileup bytes (two thirds of its wave cycles are s_waitcnt), so the
        // more byte loads are in flight per wait, the fewer waits a tile costs; what is left of the list goes four, then one at a time
        // (no loads of records that are not there)
        auto step = [&](int i, auto n_const) {
            constexpr int NU = decltype(n_const)::value;
            unsigned code[NU]; bool in_[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int j = i + u;
                const int first = __builtin_amdgcn_readlane(held.x, j), len = __builtin_amdgcn_readlane(held.y, j);
                const uint32_t plo = (uint32_t)__builtin_amdgcn_readlane(held.z, j), phi = (uint32_t)__builtin_amdgcn_readlane(held.w, j);
                const uint8_t* __restrict__ base = pile + (int64_t)(((uint64_t)phi << 32) | plo);   // pileup byte of lane 0 of the tile
                const bool in = (unsigned)(tid - first) < (unsigned)len;
                if (PAD) code[u] = (unsigned)base[tid] - 33u;
                else {
                    const unsigned off = (unsigned)(in ? tid : (first < 0 ? 0 : first));   // lanes outside the record re-read one of its bytes
                    code[u] = (unsigned)base[off] - 33u;
                }
                in_[u] = in;
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) bump(code[u], in_[u] && code[u] < (unsigned)HS_NBINS);
        };
        int i = 0;
        for (; i + HS_K2_INFLIGHT <= nrec; i += HS_K2_INFLIGHT) step(i, std::integral_constant<int, HS_K2_INFLIGHT>());
        for (; i + 4 <= nrec; i += 4) step(i, std::integral_constant<int, 4>());
        for (; i < nrec; ++i) step(i, std::integral_constant<int, 1>());
    }
    column_stats_tail<CB, FULL>(hw, tid, lane, g, total, stats, min_second, sel_count, sel_gpos, sel_depth, sel_cap, g_lo, g_hi, sel_ent);
}

// K2 on a tile plan, four positions per lane (the form the stage driver runs: byte counters, padded pileup, only the selection):
// a record's 256 bytes over the tile are ONE dword load per lane of one wavefront (lane l: positions 4 l .. 4 l + 3) instead of
// four byte loads in four wavefronts, and the wavefronts of the workgroup take the tile's records in turn (record j goes to wave
// j mod 4), all four counting into the same histogram columns with LDS atomics. A record that covers the whole tile -- 19 of 20
// at 30x with 10-kb reads -- needs no range test at all (wave-uniform branch). Counter word w of position p lives at
// [w][p % 4][p / 4]: for a fixed byte of the dword the 64 lanes hit 64 different banks. The final scan gives every thread one
// column (thread t: position 4 (t % 64) + t / 64); the selection is written in position order through the four ballots.
template <bool TOP /* the leading codes of the selection from the counters, only the positions the path reads go on */>
static __device__ __forceinline__ void column_stats_tiled_dw_body(
    const uint8_t* __restrict__ pile, const int64_t* __restrict__ tile_off, const int4* __restrict__ tile_ent, int64_t total, int min_second,
    int32_t* __restrict__ sel_count, int64_t* __restrict__ sel_gpos, int32_t* __restrict__ sel_depth, int64_t tile0, int64_t g_lo, int64_t g_hi,
    int32_t* __restrict__ sel_ent, uint2* __restrict__ sel_info) {
    constexpr int NWORDS = (HS_NBINS + 3) / 4;
    static_assert(NWORDS == 32, "the second pass splits 32 counter words over four wavefronts");
    __shared__ __attribute__((aligned(16))) uint32_t hw[NWORDS * 256];
    __shared__ unsigned long long s_b[4];
    __shared__ int s_e[4];
    __shared__ uint8_t s_item[TOP ? 256 : 4];       // TOP: the scanning threads of the positions that go to the second pass, in position order
    __shared__ uint32_t s_part[TOP ? 4 : 1][4][64];  // per wavefront and item of the round: the three largest (count << 8 | code) of its eight counter words, two per dword; their sum
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wv = wave_id();
    const int64_t tile = tile0 + (int64_t)blockIdx.x;
#pragma unroll
    for (int x = 0; x < NWORDS * 256 / (256 * 4); ++x) reinterpret_cast<uint4*>(hw)[x * 256 + tid] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    char* const my_cols = reinterpret_cast<char*>(hw) + lane * 4;      // byte k of this lane's dword: column k * 64 + lane, i.e. + 256 k bytes
    auto bump = [&](unsigned code, bool valid, int k) {
        if (valid) {
            const unsigned inc = 1u << ((code & 3u) * 8u);
            uint32_t* at = reinterpret_cast<uint32_t*>(my_cols + 256 * k + ((code & ~3u) << 8));
#if defined(HS_K2_ABL) && HS_K2_ABL == 1      /* (timing experiment, wrong results: the counting without its LDS atomics) */
            asm volatile("" :: "v"(at), "v"(inc));
#else
            __hip_atomic_fetch_add(at, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
        }
    };
    const int64_t e0 = tile_off[tile], e1 = tile_off[tile + 1];
    for (int64_t i0 = e0; i0 < e1; i0 += 64) {
        const int nrec = (e1 - i0) < 64 ? (int)(e1 - i0) : 64;
        const int4 held = tile_ent[i0 + (lane < nrec ? lane : nrec - 1)];   // lane j keeps record i0 + j
        auto step = [&](int j, auto n_const) {      // records j, j + 4, ... (this wavefront's), NU of them with their loads in flight together
            constexpr int NU = decltype(n_const)::value;
            uint32_t v[NU]; int first[NU], len[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int jj = j + 4 * u;
                first[u] = __builtin_amdgcn_readlane(held.x, jj); len[u] = __builtin_amdgcn_readlane(held.y, jj);
                const uint32_t plo = (uint32_t)__builtin_amdgcn_readlane(held.z, jj), phi = (uint32_t)__builtin_amdgcn_readlane(held.w, jj);
                const uint8_t* __restrict__ base = pile + (int64_t)(((uint64_t)phi << 32) | plo);   // pileup byte of position 0 of the tile
#if defined(HS_K2_ABL) && HS_K2_ABL == 2      /* (timing experiment, wrong results: the counting without its loads of the pileup) */
                v[u] = 0x41424344u + (uint32_t)lane + (uint32_t)(reinterpret_cast<uintptr_t>(base) & 3u);
#else
                v[u] = *reinterpret_cast<const u32_unaligned*>(base + 4 * lane);      // (the buffer is padded: positions the record does not cover are readable)
#endif
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (first[u] == 0 && len[u] == 256) {      // the record covers the tile (wave-uniform)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const unsigned code = ((v[u] >> (8 * k)) & 255u) - 33u; bump(code, code < (unsigned)HS_NBINS, k); }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned code = ((v[u] >> (8 * k)) & 255u) - 33u;
                        bump(code, (unsigned)(4 * lane + k - first[u]) < (unsigned)len[u] && code < (unsigned)HS_NBINS, k);
                    }
                }
            }
        };
        int j = wv;
        for (; j + 4 * 7 < nrec; j += 4 * 8) step(j, std::integral_constant<int, 8>());
        for (; j + 4 * 1 < nrec; j += 4 * 2) step(j, std::integral_constant<int, 2>());
        for (; j < nrec; j += 4) step(j, std::integral_constant<int, 1>());
    }
    __syncthreads();
    // ---- thread t scans column t = the counters of position 4 (t % 64) + t / 64 ----
    const int64_t g = tile * 256 + 4 * lane + wv;
    int c0 = 0, c1 = 0, depth = 0;
    bool sel;
    if (TOP) {
        // Only "can the second count reach the floor": at least two counters >= min_second (4: the bits above the low two). Exactly what
        // is selected is decided in the second pass, which has the three largest counts -- this scan is 5 instructions per counter
        // word instead of 9 (it was 55 % of the kernel's instructions).
        uint32_t big = 0;      // counters of 4 and more (of the 128 bytes of the column)
        if (g < total) {
#pragma unroll 8
            for (int w = 0; w < NWORDS; ++w) {
                const uint32_t word = hw[w * 256 + tid];
                const uint32_t t = word & 0xfcfcfcfcu;                              // the bits above the low two: non-zero <=> the counter is >= 4
                // v_msad_u8: sum of |a - b| over the bytes whose REFERENCE byte (second operand) is not zero -- with a = t | 1 that is
                // one per non-zero byte of t: three instructions per counter word
                big = __builtin_amdgcn_msad_u8(t | 0x01010101u, t, big);
            }
        }
        sel = g < total && g >= g_lo && g < g_hi && (big >= 2u || min_second < 4);      // (a floor below 4: every position to the second pass)
    } else {
    if (g < total) {
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        us2 m0a = {0, 0}, m1a = {0, 0}, m0b = {0, 0}, m1b = {0, 0};
#pragma unroll 4
        for (int w = 0; w < NWORDS; ++w) {
            const uint32_t word = hw[w * 256 + tid];
            depth = (int)__builtin_amdgcn_sad_u8(word, 0u, (uint32_t)depth);      // sum of the four byte counters
            const us2 va = __builtin_bit_cast(us2, word & 0x00ff00ffu);
            const us2 vb = __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, word, 0x0c030c01u));      // bytes 1 and 3, zero-extended: one v_perm
            const us2 la = __builtin_elementwise_min(va, m0a);
            m1a = __builtin_elementwise_max(m1a, la);
            m0a = __builtin_elementwise_max(m0a, va);
            const us2 lb = __builtin_elementwise_min(vb, m0b);
            m1b = __builtin_elementwise_max(m1b, lb);
            m0b = __builtin_elementwise_max(m0b, vb);
        }
        const int tops[8] = {m0a.x, m1a.x, m0a.y, m1a.y, m0b.x, m1b.x, m0b.y, m1b.y};
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const int v = tops[f];
            const int lo2 = v < c0 ? v : c0;     // branch-free two largest
            c1 = c1 > lo2 ? c1 : lo2;
            c0 = c0 > v ? c0 : v;
        }
    }
    const bool third = depth > c0 + c1;      // a third non-empty bin <=> reads beyond the two largest counts
    // selection in position order: second count above the floor, or exactly at it with no third allele (see column_stats_tail)
    sel = g < total && g >= g_lo && g < g_hi && (c1 > min_second || (c1 == min_second && !third));
    }
    const unsigned long long mine = __ballot(sel);
    const int we = wave_sum_i32(sel ? depth : 0);
    if (lane == 0) { s_b[wv] = mine; s_e[wv] = we; }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    int rank = 0, n_sel = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long bk = s_b[k];
        rank += __popcll(bk & below) + ((k < wv) ? (int)((bk >> lane) & 1ull) : 0);
        n_sel += __popcll(bk);
    }
    if (!TOP) {
        if (sel) {
            const int64_t slot = (int64_t)blockIdx.x * 256 + rank;
            sel_gpos[slot] = g; sel_depth[slot] = depth;
        }
        if (tid == 0) { sel_count[blockIdx.x] = n_sel; if (sel_ent) sel_ent[blockIdx.x] = s_e[0] + s_e[1] + s_e[2] + s_e[3]; }
        return;
    }
    // ---- the leading codes of the selected positions, from the counters that are still in LDS (call_variants.cpp:477-507: the three
    // largest counts and the codes of the first two), and only the positions the path can use go on: a position whose three
    // counts are distinct has ONE order of its leading codes, the predicate of :527-528 (also loop D's, :751-752) can be asked here;
    // what fails it, or has neither five reads of the second code nor more than five times the third count, is read by nobody
    // (k_candidates_scan, K4). Positions with equal leading counts go on undecided (flag HS_COL_TIE): k_column_top3_exact orders
    // them as the reference does once their reads are gathered. Items = selected positions in position order, lane = item, the 32
    // counter words of an item's column split over the four wavefronts; keys (count << 8 | code) as packed 16-bit pairs. ----
    if (n_sel == 0) { if (tid == 0) { sel_count[blockIdx.x] = 0; if (sel_ent) sel_ent[blockIdx.x] = 0; } return; }
    if (sel) s_item[rank] = (uint8_t)tid;
    __syncthreads();
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    auto pk = [](uint32_t x) { return __builtin_bit_cast(us2, x); };
    auto insert3 = [](us2& t0, us2& t1, us2& t2, const us2 v) {
        const us2 lo0 = __builtin_elementwise_min(v, t0); t0 = __builtin_elementwise_max(v, t0);
        const us2 lo1 = __builtin_elementwise_min(lo0, t1); t1 = __builtin_elementwise_max(lo0, t1);
        t2 = __builtin_elementwise_max(lo1, t2);
    };
    auto merge3 = [](us2& a0, us2& a1, us2& a2, const us2 b0, const us2 b1, const us2 b2) {      // two descending triples -> the three largest of the six
        const us2 m = __builtin_elementwise_min(a0, b0), x = __builtin_elementwise_max(a1, b1), y = __builtin_elementwise_min(a1, b1);
        a0 = __builtin_elementwise_max(a0, b0);
        a1 = __builtin_elementwise_max(m, x);
        a2 = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_elementwise_min(m, x), y), __builtin_elementwise_max(a2, b2));
    };
    int n_kept = 0, e_kept = 0;      // (wavefront 0's)
    for (int r0 = 0; r0 < n_sel; r0 += 64) {
        const int item = r0 + lane;
        const uint32_t t_scan = s_item[item < n_sel ? item : 0];
        {
            const uint32_t* __restrict__ col = hw + t_scan + (unsigned)(8 * wv) * 256u;
            us2 a0 = {0, 0}, a1 = {0, 0}, a2 = {0, 0}, b0 = {0, 0}, b1 = {0, 0}, b2 = {0, 0};
            uint32_t dsum = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t word = col[j * 256];
                dsum = __builtin_amdgcn_sad_u8(word, 0u, dsum);
                const uint32_t idw = (uint32_t)(4 * (8 * wv + j)) * 0x00010001u;      // (wave-uniform)
                // bytes 0 and 2 -> codes 4 w, 4 w + 2; bytes 1 and 3 -> codes 4 w + 1, 4 w + 3
                insert3(a0, a1, a2, pk(((word & 0x00ff00ffu) << 8) | (idw | 0x00020000u)));
                insert3(b0, b1, b2, pk((word & 0xff00ff00u) | (idw | 0x00030001u)));
            }
            merge3(a0, a1, a2, b0, b1, b2);
            s_part[wv][0][lane] = __builtin_bit_cast(uint32_t, a0); s_part[wv][1][lane] = __builtin_bit_cast(uint32_t, a1); s_part[wv][2][lane] = __builtin_bit_cast(uint32_t, a2);
            s_part[wv][3][lane] = dsum;
        }
        __syncthreads();
        if (wv == 0) {
            us2 a0 = pk(s_part[0][0][lane]), a1 = pk(s_part[0][1][lane]), a2 = pk(s_part[0][2][lane]);
#pragma unroll
            for (int w = 1; w < 4; ++w) merge3(a0, a1, a2, pk(s_part[w][0][lane]), pk(s_part[w][1][lane]), pk(s_part[w][2][lane]));
            // the two halves of the packed triple -> one triple of 32-bit keys
            const uint32_t l0 = a0.x, l1 = a1.x, l2 = a2.x, h0 = a0.y, h1 = a1.y, h2 = a2.y;
            const uint32_t m = l0 < h0 ? l0 : h0, x = l1 > h1 ? l1 : h1, y = l1 < h1 ? l1 : h1;
            const uint32_t K0 = l0 > h0 ? l0 : h0, K1 = m > x ? m : x;
            uint32_t K2 = m < x ? m : x; K2 = K2 > y ? K2 : y; K2 = K2 > l2 ? K2 : l2; K2 = K2 > h2 ? K2 : h2;
            const int n0 = (int)(K0 >> 8), n1 = (int)(K1 >> 8), n2 = (int)(K2 >> 8);
            const int k0 = 33 + (int)(K0 & 255u), k1 = 33 + (int)(K1 & 255u);
            const bool tie = n0 == n1 || n1 == n2 || n1 == 0;
            const bool gt5 = n1 > 5 * n2;
            // call_variants.cpp:527-528 / :751-752 on the raw code bytes
            const bool central = k0 % 5 != k1 % 5 && ((k1 - 33) % 5 != 4 || (k1 / 5 % 5 != k0 % 5 && k1 / 25 % 5 != k0 % 5));
            // the selection itself (second count above the floor, or exactly at it with no third allele: column_stats_tail), then what the path reads
            const bool chosen = n1 > min_second || (n1 == min_second && n2 == 0);
            const bool keep = item < n_sel && chosen && (tie || (central && (n1 >= 5 || gt5)));
            const unsigned long long km = __ballot(keep);
            const int d = (int)(s_part[0][3][lane] + s_part[1][3][lane] + s_part[2][3][lane] + s_part[3][3][lane]);
            if (keep) {
                const int64_t slot = (int64_t)blockIdx.x * 256 + n_kept + __popcll(km & below);
                sel_gpos[slot] = tile * 256 + 4 * (int)(t_scan & 63u) + (int)(t_scan >> 6);
                sel_depth[slot] = d;
                sel_info[slot] = make_uint2((uint32_t)n0 | ((uint32_t)n1 << 16),
                                            (uint32_t)k0 | ((uint32_t)k1 << 8) | ((tie ? 32u /* HS_COL_TIE */ : 0u) | (gt5 ? 64u /* HS_COL_C1GT5C2 */ : 0u)) << 16 | (uint32_t)(n2 < 63 ? n2 : 63) << 24 | 0x80000000u);
            }
            n_kept += __popcll(km);
            e_kept += wave_sum_i32(keep ? d : 0);
        }
        if (r0 + 64 < n_sel) __syncthreads();      // (s_part is written again)
    }
    if (tid == 0) { sel_count[blockIdx.x] = n_kept; if (sel_ent) sel_ent[blockIdx.x] = e_kept; }
}
__global__ __launch_bounds__(256) void k_column_stats_tiled_dw(
    const uint8_t* __restrict__ pile, const int64_t* __restrict__ tile_off, const int4* __restrict__ tile_ent, int64_t total, int min_second,
    int32_t* __restrict__ sel_count, int64_t* __restrict__ sel_gpos, int32_t* __restrict__ sel_depth, int64_t tile0, int64_t g_lo, int64_t g_hi,
    int32_t* __restrict__ sel_ent, uint2* __restrict__ sel_info) {
    column_stats_tiled_dw_body<true>(pile, tile_off, tile_ent, total, min_second, sel_count, sel_gpos, sel_depth, tile0, g_lo, g_hi, sel_ent, sel_info);
}

// ------------------------------------------------------------------------------------------------
// K5a SNP bit-planes: the 0/1 matrices A (read carries second_base) and R (read carries ref_base) of
// list_similarities_and_differences_between_reads3 (separate_reads.cpp:386-405) as bit rows, built from the SNP columns that
// are resident for the Chinese-Whispers seeding anyway. One workgroup per (contig, HS_SP_WORDS consecutive words = 256 SNP
// columns): the bits of HS_SP_READS reads at a time are ORed together in LDS (one ds_or_b64 per column entry) and every word of
// those rows is then stored once, 32 contiguous bytes per read and plane -- zero words included, so the planes need no clearing.
// blk_contig / blk_word0: the launch's list of (contig, first word); contigs with words[c] == 0 (low-memory path, no SNPs) have
// no workgroup.
// ------------------------------------------------------------------------------------------------
#define HS_SP_WORDS 4
#define HS_SP_READS 512
#define HS_SP_WAVES 16      // wavefronts of a workgroup: each walks every sixteenth of its 256 columns (sixteen columns in a row, not 64: at eight contig groups
                           // a launch is a few hundred workgroups and lasts as long as one wavefront's chain)
#define HS_SP_THREADS (64 * HS_SP_WAVES)
__global__ __launch_bounds__(HS_SP_THREADS) void k_snp_planes(
    const int64_t* __restrict__ col_off, const int32_t* __restrict__ col_idx, const uint8_t* __restrict__ col_code,
    const uint8_t* __restrict__ snp_ref, const uint8_t* __restrict__ snp_alt, const int32_t* __restrict__ snp_contig,
    const int64_t* __restrict__ contig_snp_base, const int64_t* __restrict__ plane_off, const int32_t* __restrict__ words,
    const int32_t* __restrict__ n_reads, const int32_t* __restrict__ blk_contig, const int32_t* __restrict__ blk_word0, int n_snps,
    unsigned long long* __restrict__ alt, unsigned long long* __restrict__ ref,
    const int64_t* __restrict__ read_base /* [C] first row of the contig in rng_lo / rng_hi, or NULL */, int32_t* __restrict__ rng_lo, int32_t* __restrict__ rng_hi) {
    __shared__ unsigned long long s_a[HS_SP_READS][HS_SP_WORDS], s_r[HS_SP_READS][HS_SP_WORDS];
    // which of the workgroup's words a read is PRESENT in (any code, also a third allele that sets no bit in either plane): K5 only
    // compares two reads over the words both are present in, and skips pairs of read blocks that share none
    __shared__ unsigned int s_p[HS_SP_READS];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = blk_contig[blockIdx.x];
    const int w0 = blk_word0[blockIdx.x];
    const int W = words[c], N = n_reads[c];
    const int nw = W - w0 < HS_SP_WORDS ? W - w0 : HS_SP_WORDS;
    const int64_t s0 = contig_snp_base[c] + 64ll * w0;      // first column of the workgroup
    unsigned long long* __restrict__ A = alt + plane_off[c];
    unsigned long long* __restrict__ R = ref + plane_off[c];
    for (int rb = 0; rb < N; rb += HS_SP_READS) {
        for (int x = tid; x < HS_SP_READS * HS_SP_WORDS; x += HS_SP_THREADS) { (&s_a[0][0])[x] = 0ull; (&s_r[0][0])[x] = 0ull; }
        if (read_base) for (int x = tid; x < HS_SP_READS; x += HS_SP_THREADS) s_p[x] = 0u;
        __syncthreads();
        {
            // a wavefront takes every sixteenth column of the workgroup: lane t first reads what column wv + 16 t needs (one round trip for
            // all of them), then the columns go by one after the other, lanes = entries, the next column's entries already on their way
            const int q_l = wv + HS_SP_WAVES * lane;
            const int64_t s_l = s0 + q_l;
            const bool v_l = lane < 256 / HS_SP_WAVES && q_l < 64 * nw && s_l < n_snps && snp_contig[s_l] == c;      // (the contig's last word is partly filled)
            const int64_t e0_l = v_l ? col_off[s_l] : 0;
            const int n_l = v_l ? (int)(col_off[s_l + 1] - e0_l) : 0;
            const int al_l = v_l ? ((int)snp_ref[s_l] | ((int)snp_alt[s_l] << 8)) : 0;
            unsigned long long todo = __ballot(v_l);
            auto rl_e0 = [&](int l) { return ((int64_t)(unsigned)__builtin_amdgcn_readlane((int)(e0_l >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(e0_l & 0xffffffffll), l); };
            int idx_n = 0, code_n = -1;
            if (todo) {
                const int l = __builtin_ctzll(todo);
                const int64_t e0 = rl_e0(l); const int n = __builtin_amdgcn_readlane(n_l, l);
                if (lane < n) { idx_n = col_idx[e0 + lane]; code_n = col_code[e0 + lane]; }
            }
            while (todo) {
                const int t = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                const int q = wv + HS_SP_WAVES * t;
                const int64_t e0 = rl_e0(t);
                const int n = __builtin_amdgcn_readlane(n_l, t);
                const int al = __builtin_amdgcn_readlane(al_l, t);
                const int rbv = al & 255, abv = al >> 8;
                int idx = idx_n, code = code_n;
                idx_n = 0; code_n = -1;
                if (todo) {
                    const int l = __builtin_ctzll(todo);
                    const int64_t e0n = rl_e0(l); const int nn = __builtin_amdgcn_readlane(n_l, l);
                    if (lane < nn) { idx_n = col_idx[e0n + lane]; code_n = col_code[e0n + lane]; }
                }
                const unsigned long long bit = 1ull << (q & 63);
                const int wq = q >> 6;
                for (int eb = 0; eb < n; eb += 64) {
                    if (eb > 0) { idx = 0; code = -1; if (eb + lane < n) { idx = col_idx[e0 + eb + lane]; code = col_code[e0 + eb + lane]; } }
                    const int r = idx - rb;
                    if (code >= 0 && r >= 0 && r < HS_SP_READS) {
                        if (code == rbv) atomicOr(&s_r[r][wq], bit);
                        else if (code == abv) atomicOr(&s_a[r][wq], bit);
                        if (read_base && !(s_p[r] & (1u << wq))) atomicOr(&s_p[r], 1u << wq);
                    }
                }
            }
        }
        __syncthreads();
        const int nr = N - rb < HS_SP_READS ? N - rb : HS_SP_READS;
        for (int x = tid; x < nr * HS_SP_WORDS; x += HS_SP_THREADS) {      // (consecutive threads: consecutive words of a row)
            const int r = x / HS_SP_WORDS, wq = x % HS_SP_WORDS;
            if (wq < nw) {
                const int64_t at = (int64_t)(rb + r) * W + w0 + wq;
                A[at] = s_a[r][wq]; R[at] = s_r[r][wq];
            }
        }
        if (read_base)
            for (int r = tid; r < nr; r += HS_SP_THREADS) {
                const unsigned int pm = s_p[r];
                if (pm) {
                    const int64_t at = read_base[c] + rb + r;
                    atomicMin(&rng_lo[at], w0 + __builtin_ctz(pm));
                    atomicMax(&rng_hi[at], w0 + 31 - __builtin_clz(pm));
                }
            }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// K5 sim/diff: separate_reads.cpp:374-433 as bit-plane popcounts. 64x64 read-pair tile per workgroup,
// 4x4 pairs per thread, planes staged through LDS 16 words at a time (padded rows: conflict-free b64 reads).
// ------------------------------------------------------------------------------------------------
#define SD_KW 16
__global__ __launch_bounds__(256) void k_simdiff(
    const uint64_t* __restrict__ alt, const uint64_t* __restrict__ ref, const int64_t* __restrict__ plane_off,
    const int32_t* __restrict__ n_reads, const int32_t* __restrict__ words, const int64_t* __restrict__ out_off,
    const int32_t* __restrict__ tile_contig, const int32_t* __restrict__ tile_i, const int32_t* __restrict__ tile_j,
    int32_t* __restrict__ sim, int32_t* __restrict__ diff, int es /* element stride: 1 = two arrays, 2 = (sim, diff) pairs in one (diff = sim + 1) */,
    const int64_t* __restrict__ read_base /* [C] or NULL */, const int32_t* __restrict__ orig_of /* row of the matrices -> read, per contig at read_base[c] */,
    const int32_t* __restrict__ rng_lo, const int32_t* __restrict__ rng_hi) {
    // With orig_of the rows / columns of the matrices are the reads in the order of their START POSITIONS (row k = read orig_of[k]): the
    // reads of a 64-row block then lie next to each other on the contig, a tile of two blocks far apart has no read pair that shares a SNP
    // -- it is not computed and not written (nobody reads it: a window only ever asks for pairs of reads that are both present at its
    // first and its last SNP) -- and a tile that is computed only walks the words both blocks are present in.
    __shared__ uint64_t s_planes[4][64][SD_KW + 1];      // one array: the mirrored store below reuses it as a 64 x 65 int tile
    __shared__ int s_rows[2][64];
    __shared__ int s_rng[4];
    uint64_t (*sAi)[SD_KW + 1] = s_planes[0]; uint64_t (*sRi)[SD_KW + 1] = s_planes[1];
    uint64_t (*sAj)[SD_KW + 1] = s_planes[2]; uint64_t (*sRj)[SD_KW + 1] = s_planes[3];
    const int tid = (int)threadIdx.x;
    const int c = tile_contig[blockIdx.x];
    const int i0 = tile_i[blockIdx.x] * 64, j0 = tile_j[blockIdx.x] * 64;
    const int N = n_reads[c], W = words[c];
    const uint64_t* __restrict__ A = alt + plane_off[c];
    const uint64_t* __restrict__ R = ref + plane_off[c];
    int w_begin = 0, w_end = W;
    if (orig_of) {
        const int64_t rb = read_base[c];
        if (tid < 128) {      // wave 0: the i rows, wave 1: the j rows -- the read of every row and the words it is present in
            const int side = tid >> 6, k = (side ? j0 : i0) + (tid & 63);
            int lo = 0x7fffffff, hi = -1, rd = -1;
            if (k < N) { rd = orig_of[rb + k]; lo = rng_lo[rb + rd]; hi = rng_hi[rb + rd]; }
            s_rows[side][tid & 63] = rd;
            lo = -wave_max_i32(-lo); hi = wave_max_i32(hi);
            if ((tid & 63) == 0) { s_rng[2 * side] = lo; s_rng[2 * side + 1] = hi; }
        }
        __syncthreads();
        w_begin = s_rng[0] > s_rng[2] ? s_rng[0] : s_rng[2];
        const int e = s_rng[1] < s_rng[3] ? s_rng[1] : s_rng[3];
        w_end = e + 1;
        if (w_begin >= w_end) return;      // no word in common: every pair of the tile is (0, 0), and nobody asks
        w_begin &= ~(SD_KW - 1);
        if (w_end > W) w_end = W;
    }
    const int ti = tid >> 4, tj = tid & 15;
    int s_acc[4][4], d_acc[4][4], u_acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { s_acc[a][b] = 0; d_acc[a][b] = 0; u_acc[a][b] = 0; }

    for (int w0 = w_begin; w0 < w_end; w0 += SD_KW) {
        // stage 64 rows x SD_KW words of both planes for both sides (coalesced along words)
        for (int x = tid; x < 64 * SD_KW; x += 256) {
            const int row = x / SD_KW, w = x % SD_KW;
            const int gw = w0 + w;
            const int gi = orig_of ? s_rows[0][row] : (i0 + row < N ? i0 + row : -1), gj = orig_of ? s_rows[1][row] : (j0 + row < N ? j0 + row : -1);
            const bool wi = gi >= 0 && gw < W, wj = gj >= 0 && gw < W;
            sAi[row][w] = wi ? A[(int64_t)gi * W + gw] : 0ull;
            sRi[row][w] = wi ? R[(int64_t)gi * W + gw] : 0ull;
            sAj[row][w] = wj ? A[(int64_t)gj * W + gw] : 0ull;
            sRj[row][w] = wj ? R[(int64_t)gj * W + gw] : 0ull;
        }
        __syncthreads();
#pragma unroll 4
        for (int w = 0; w < SD_KW; ++w) {
            uint64_t ai[4], ri[4], aj[4], rj[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { ai[a] = sAi[ti + 16 * a][w]; ri[a] = sRi[ti + 16 * a][w]; }
#pragma unroll
            for (int b = 0; b < 4; ++b) { aj[b] = sAj[tj + 16 * b][w]; rj[b] = sRj[tj + 16 * b][w]; }
            // a read carries at most one of the two alleles at a SNP (the planes are disjoint), so with U = A | R:
            // pop(Ui & Uj) = pop(Ai & Aj) + pop(Ri & Rj) + [pop(Ai & Rj) + pop(Ri & Aj)] -- three popcounts per pair instead of four
            uint64_t ui[4], uj[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { ui[a] = ai[a] | ri[a]; uj[a] = aj[a] | rj[a]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    s_acc[a][b] += __popcll(ai[a] & aj[b]);      // (pop(A A), pop(R R), pop(U U): combined after the loop)
                    d_acc[a][b] += __popcll(ri[a] & rj[b]);
                    u_acc[a][b] += __popcll(ui[a] & uj[b]);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int paa = s_acc[a][b], prr = d_acc[a][b], puu = u_acc[a][b];
            s_acc[a][b] = 3 * paa + prr; d_acc[a][b] = puu - paa - prr;
        }
    int32_t* __restrict__ S = sim + out_off[c] * es;
    int32_t* __restrict__ D = diff + out_off[c] * es;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gi = i0 + ti + 16 * a, gj = j0 + tj + 16 * b;
            if (gi < N && gj < N) {
                const bool dg = gi == gj;
                S[((int64_t)gi * N + gj) * es] = dg ? 0 : s_acc[a][b];
                D[((int64_t)gi * N + gj) * es] = dg ? 0 : d_acc[a][b];
            }
        }
    // Both matrices are symmetric (sim = 3 A A^T + R R^T, diff = A R^T + R A^T): only the tiles on and above the diagonal are
    // computed, the mirror image of an off-diagonal tile goes out through LDS so that its rows are written contiguously too
    if (i0 != j0) {
        int32_t* tile = reinterpret_cast<int32_t*>(&s_planes[0][0][0]);      // 64 x 65 ints (16.6 KB of the 34.8 KB staging area; the loop above is done with it)
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            __syncthreads();
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) tile[(tj + 16 * b) * 65 + (ti + 16 * a)] = which == 0 ? s_acc[a][b] : d_acc[a][b];
            __syncthreads();
            int32_t* __restrict__ O = which == 0 ? S : D;
            for (int x = tid; x < 64 * 64; x += 256) {
                const int rj = x >> 6, ci = x & 63;              // row of the mirrored tile = a read of the j side
                const int gj = j0 + rj, gi = i0 + ci;
                if (gj < N && gi < N) O[((int64_t)gj * N + gi) * es] = tile[rj * 65 + ci];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K5 for the low-memory path (create_read_graph_low_memory, separate_reads.cpp:538-693): the reference compares every pair of
// masked reads of a WINDOW over the SNPs both reads cover and never forms the N x N matrices. With the 0/1/2 vectors as bit rows
// that comparison is the same popcount as k_simdiff (2 & 2 -> +3, 1 & 1 -> +1, 1 & 2 or 2 & 1 -> difference; a read has no bits
// outside its span, so all words of the contig can be walked), restricted to the window's m masked reads: an m x m matrix per
// window. One workgroup per 64 x 64 tile of a window, every tile (no mirroring), rows gathered through the window's read list.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_simdiff_windows(
    const uint64_t* __restrict__ alt, const uint64_t* __restrict__ ref, const int64_t* __restrict__ plane_off, const int32_t* __restrict__ words,
    const int32_t* __restrict__ win_contig, const int64_t* __restrict__ win_mask_off, const int32_t* __restrict__ mask_ids,
    const int64_t* __restrict__ win_mat_off, const int32_t* __restrict__ tile_win, const int32_t* __restrict__ tile_i, const int32_t* __restrict__ tile_j,
    int32_t* __restrict__ wsim, int32_t* __restrict__ wdiff, int es) {
    __shared__ uint64_t s_planes[4][64][SD_KW + 1];
    uint64_t (*sAi)[SD_KW + 1] = s_planes[0]; uint64_t (*sRi)[SD_KW + 1] = s_planes[1];
    uint64_t (*sAj)[SD_KW + 1] = s_planes[2]; uint64_t (*sRj)[SD_KW + 1] = s_planes[3];
    const int tid = (int)threadIdx.x;
    const int w = tile_win[blockIdx.x];
    const int c = win_contig[w];
    const int i0 = tile_i[blockIdx.x] * 64, j0 = tile_j[blockIdx.x] * 64;
    const int64_t m0 = win_mask_off[w];
    const int m = (int)(win_mask_off[w + 1] - m0);
    const int32_t* __restrict__ ids = mask_ids + m0;
    const int W = words[c];
    const uint64_t* __restrict__ A = alt + plane_off[c];
    const uint64_t* __restrict__ R = ref + plane_off[c];
    const int ti = tid >> 4, tj = tid & 15;
    int s_acc[4][4], d_acc[4][4], u_acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { s_acc[a][b] = 0; d_acc[a][b] = 0; u_acc[a][b] = 0; }
    for (int w0 = 0; w0 < W; w0 += SD_KW) {
        for (int x = tid; x < 64 * SD_KW; x += 256) {
            const int row = x / SD_KW, ww = x % SD_KW;
            const int gw = w0 + ww;
            const bool wi = i0 + row < m && gw < W, wj = j0 + row < m && gw < W;
            const int64_t ri = wi ? ids[i0 + row] : 0, rj = wj ? ids[j0 + row] : 0;
            sAi[row][ww] = wi ? A[ri * W + gw] : 0ull;
            sRi[row][ww] = wi ? R[ri * W + gw] : 0ull;
            sAj[row][ww] = wj ? A[rj * W + gw] : 0ull;
            sRj[row][ww] = wj ? R[rj * W + gw] : 0ull;
        }
        __syncthreads();
#pragma unroll 4
        for (int ww = 0; ww < SD_KW; ++ww) {
            uint64_t ai[4], ri[4], aj[4], rj[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { ai[a] = sAi[ti + 16 * a][ww]; ri[a] = sRi[ti + 16 * a][ww]; }
#pragma unroll
            for (int b = 0; b < 4; ++b) { aj[b] = sAj[tj + 16 * b][ww]; rj[b] = sRj[tj + 16 * b][ww]; }
            // a read carries at most one of the two alleles at a SNP (the planes are disjoint), so with U = A | R:
            // pop(Ui & Uj) = pop(Ai & Aj) + pop(Ri & Rj) + [pop(Ai & Rj) + pop(Ri & Aj)] -- three popcounts per pair instead of four
            uint64_t ui[4], uj[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { ui[a] = ai[a] | ri[a]; uj[a] = aj[a] | rj[a]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    s_acc[a][b] += __popcll(ai[a] & aj[b]);      // (pop(A A), pop(R R), pop(U U): combined after the loop)
                    d_acc[a][b] += __popcll(ri[a] & rj[b]);
                    u_acc[a][b] += __popcll(ui[a] & uj[b]);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int paa = s_acc[a][b], prr = d_acc[a][b], puu = u_acc[a][b];
            s_acc[a][b] = 3 * paa + prr; d_acc[a][b] = puu - paa - prr;
        }
    int32_t* __restrict__ S = wsim + win_mat_off[w] * es;
    int32_t* __restrict__ D = wdiff + win_mat_off[w] * es;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gi = i0 + ti + 16 * a, gj = j0 + tj + 16 * b;
            if (gi < m && gj < m) { S[((int64_t)gi * m + gj) * es] = s_acc[a][b]; D[((int64_t)gi * m + gj) * es] = d_acc[a][b]; }
        }
}

// ------------------------------------------------------------------------------------------------
// K4 SNP column x partition correlation: distance(Partition&, Column&) + computeChiSquare
// (call_variants.cpp:778-967, :1135-1163) for loops C (:721-738) and D (:745-764) of keep_only_robust_variants.
// One wavefront per extracted column, lanes = reads of the column. For every final partition of the contig:
//  * the reads present in both are found with one gather of the partition state per lane;
//  * the codes carried by those reads are enumerated leader by leader (v_readlane + __ballot), each distinct code gets
//    a slot (lane j holds slot j) with its total / "+1" / "-1" counts = __popcll of ballots -- the whole 2x2 table
//    falls out of popcounts once the second allele is known;
//  * second allele = most frequent non-reference code; ties go to the first key in the reference's hash-map iteration
//    order (hs_rh8.h emulator, run by one lane, rare). A reference code >= 128 never equals a key in the reference's
//    signed/unsigned comparison (:838) and then competes as well.
//  * chi-square with the reference's operation types (float marginals, double squares, float result); hipcc is run
//    with -ffp-contract=off and IEEE division so the bits match the host.
// keep[col] = 1 if the column is kept by loop C (candidates) or rescued by loop D (second count >= 5 + byte predicate).
// ------------------------------------------------------------------------------------------------
struct Table2x2 { int n00, n01, n10, n11; };

static __device__ __forceinline__ float chi_square_dev(const Table2x2& d) {
    const int n = d.n00 + d.n01 + d.n10 + d.n11;
    if (n == 0) return 0;
    const float pmax1 = float(d.n10 + d.n11) / n;
    const float pmax2 = float(d.n01 + d.n11) / n;
    if (pmax1 * (1 - pmax1) == 0 && pmax2 * (1 - pmax2) == 0) return -1;
    if (pmax1 * pmax2 * (1 - pmax1) * (1 - pmax2) == 0) return 0;
    const float e00 = (1 - pmax1) * (1 - pmax2) * n, e01 = (1 - pmax1) * pmax2 * n;
    const float e10 = pmax1 * (1 - pmax2) * n, e11 = pmax1 * pmax2 * n;
    const double d00 = (double)(float)(d.n00 - e00), d01 = (double)(float)(d.n01 - e01);
    const double d10 = (double)(float)(d.n10 - e10), d11 = (double)(float)(d.n11 - e11);
    return (float)(d00 * d00 / (double)e00 + d01 * d01 / (double)e01 + d10 * d10 / (double)e10 + d11 * d11 / (double)e11);
}

static __device__ __forceinline__ bool central_base_test_dev(int k0, int k1) {
    // call_variants.cpp:527-528 and :751-752 (same predicate on raw code bytes)
    return k0 % 5 != k1 % 5 && ((k1 - '!') % 5 != 4 || (k1 / 5 % 5 != k0 % 5 && k1 / 25 % 5 != k0 % 5));
}

// the 2x2 table of one column against one partition; wave-uniform result
// place of a byte key in the iteration order of the reference's hash map while it has at most 12 keys (tests/harness/rh8_static_order.cpp): up to 6
// keys 8 buckets and the first multiplier, 7 to 12 keys (`wide`) 16 buckets and the second one; rank = home bucket << 5 | 31 - low five hash bits
static __device__ __forceinline__ int rh8_rank_dev(int key, bool wide) {
    unsigned long long x = (unsigned long long)(key & 255);
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33;
    x *= wide ? (0xc4ceb9fe1a85ec53ull + 0xc4ceb9fe1a85ec54ull) : 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return (int)((((x >> 5) & (wide ? 15ull : 7ull)) << 5) | (31ull - (x & 31ull)));
}

static __device__ Table2x2 column_vs_partition_dev(const int32_t* __restrict__ idx, const uint8_t* __restrict__ code, int n,
                                                   const int8_t* __restrict__ state, int ref, uint8_t* s_seen /* [128] */,
                                                   uint8_t* s_ord /* [260] */, int* s_ord_n /* [1] */, uint8_t* s_map /* [3 * 512]: the hash map's tables */) {
    const int lane = lane_id();
    // slot j of the distinct-code table lives in lane j (codes 33..157: at most 125 distinct -> two slots per lane)
    int sc[2] = {-1, -1}, st_tot[2] = {0, 0}, st_pos[2] = {0, 0}, st_neg[2] = {0, 0};
    int nseen = 0, shared = 0;
    for (int base = 0; base < n; base += 64) {
        const int e = base + lane;
        const bool valid = e < n;
        const int cd = valid ? (int)code[e] : -1;
        const int stv = valid ? (int)state[idx[e]] : 2;
        const bool take = valid && stv != 2;                 // 2 == read not in the partition
        const unsigned long long plus = __ballot(take && stv == 1), minus = __ballot(take && stv == -1);
        unsigned long long rem = __ballot(take);
        shared += __popcll(rem);
        while (rem) {
            const int leader = __builtin_ctzll(rem);
            const int c = __builtin_amdgcn_readlane(cd, leader);
            const unsigned long long m = __ballot(take && cd == c);
            rem &= ~m;
            const int kt = __popcll(m), kp = __popcll(m & plus), kn = __popcll(m & minus);
            const unsigned long long hit0 = __ballot(sc[0] == c), hit1 = __ballot(sc[1] == c);
            if (hit0) { if (sc[0] == c) { st_tot[0] += kt; st_pos[0] += kp; st_neg[0] += kn; } }
            else if (hit1) { if (sc[1] == c) { st_tot[1] += kt; st_pos[1] += kp; st_neg[1] += kn; } }
            else {
                const int slot = nseen & 63;
                if (nseen < 64) { if (lane == slot) { sc[0] = c; st_tot[0] = kt; st_pos[0] = kp; st_neg[0] = kn; } }
                else { if (lane == slot) { sc[1] = c; st_tot[1] = kt; st_pos[1] = kp; st_neg[1] = kn; } }
                nseen++;
            }
        }
    }
    Table2x2 r; r.n00 = r.n01 = r.n10 = r.n11 = 0;
    if (shared == 0) return r;                               // not comparable (call_variants.cpp:817-828)
    // reference allele counts
    const unsigned long long ref0 = __ballot(sc[0] == ref), ref1 = __ballot(sc[1] == ref);
    const bool ref_seen = (ref0 | ref1) != 0ull;
    if (ref0) { const int l = __builtin_ctzll(ref0); r.n11 = __builtin_amdgcn_readlane(st_pos[0], l); r.n01 = __builtin_amdgcn_readlane(st_neg[0], l); }
    else if (ref1) { const int l = __builtin_ctzll(ref1); r.n11 = __builtin_amdgcn_readlane(st_pos[1], l); r.n01 = __builtin_amdgcn_readlane(st_neg[1], l); }
    // second allele: most frequent eligible code (call_variants.cpp:832-844)
    const bool ref_eligible = ref >= 128;
    int key0 = (lane < nseen && (sc[0] != ref || ref_eligible)) ? st_tot[0] : -1;
    int key1 = (lane + 64 < nseen && (sc[1] != ref || ref_eligible)) ? st_tot[1] : -1;
    int best = key0 > key1 ? key0 : key1;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(best, d, 64); best = o > best ? o : best; }
    int second = ' ';
    bool second_is_unseen_ref = false;
    if (ref_eligible && !ref_seen && best < 0) { second = ref; second_is_unseen_ref = true; best = 0; }   // content2[ref] inserts a zero-count key
    if (best >= 0 && !second_is_unseen_ref) {
        const unsigned long long b0 = __ballot(key0 == best), b1 = __ballot(key1 == best);
        const int nbest = __popcll(b0) + __popcll(b1) + ((ref_eligible && !ref_seen && best == 0) ? 1 : 0);
        if (nbest == 1) {
            second = b0 ? __builtin_amdgcn_readlane(sc[0], __builtin_ctzll(b0)) : __builtin_amdgcn_readlane(sc[1], __builtin_ctzll(b1));
        } else {
            // tie: first of the tied keys in the hash map's iteration order (keys inserted in first-appearance order, then ref). While the map has
            // at most 12 keys that order is the keys' static rank (unless two tied keys share a rank, or a key sits six slots from its bucket) ...
            bool resolved = false;
            const int nkeys = nseen + (ref_seen ? 0 : 1);
            if (nkeys <= 12) {
                const bool wide = nkeys > 6;
                const int rk = lane < nseen ? rh8_rank_dev(sc[0], wide) : 0x7fffffff;      // (at most 12 codes: slot 0 of the lanes only)
                const int rref = rh8_rank_dev(ref, wide);
                bool far = false;
                if (wide) {
                    int carry = 0;
                    for (int bkt = 0; bkt < 16; ++bkt) {
                        const int cb = __popcll(__ballot(lane < nseen && (rk >> 5) == bkt)) + ((!ref_seen && (rref >> 5) == bkt) ? 1 : 0);
                        if (cb > 0 && carry + cb - 1 >= 6) far = true;
                        carry = carry + cb - 1 > 0 ? carry + cb - 1 : 0;
                    }
                }
                if (!far) {
                    const bool ref_cand = ref_eligible && !ref_seen && best == 0;
                    const int mine = key0 == best ? rk : 0x7fffffff;
                    int lo = -wave_max_i32(-mine);
                    if (ref_cand && rref < lo) lo = rref;
                    const int n_lo = __popcll(__ballot(key0 == best && rk == lo)) + ((ref_cand && rref == lo) ? 1 : 0);
                    if (n_lo == 1) {
                        const unsigned long long w = __ballot(key0 == best && rk == lo);
                        second = w ? __builtin_amdgcn_readlane(sc[0], __builtin_ctzll(w)) : ref;
                        resolved = true;
                    }
                }
            }
            if (!resolved) {
            // ... else one lane replays the insertions on the emulator and publishes the order; the tied slots are then looked up in that order.
            if (lane < nseen) s_seen[lane] = (uint8_t)sc[0];
            if (lane + 64 < nseen) s_seen[lane + 64] = (uint8_t)sc[1];
            wave_lds_sync();
            if (lane == 0) {
                hs::Rh8View rh; rh.init(s_map, s_map + 512, s_map + 1024, 512);      // (cap 512: every set of byte keys fits, no overflow)
                for (int i = 0; i < nseen; ++i) rh.insert(s_seen[i]);
                rh.insert((uint8_t)ref);
                if (rh.overflow) __builtin_trap();      // (never silently another order)
                s_ord_n[0] = rh.order(s_ord);
            }
            wave_lds_sync();
            const int m = s_ord_n[0];
            second = -1;
            for (int i = 0; i < m && second < 0; ++i) {
                const int k = s_ord[i];
                if (k == ref && !ref_eligible) continue;
                const unsigned long long h0 = __ballot(lane < nseen && sc[0] == k), h1 = __ballot(lane + 64 < nseen && sc[1] == k);
                int cnt = 0;                                     // an unseen ref has count 0
                if (h0) cnt = __builtin_amdgcn_readlane(st_tot[0], __builtin_ctzll(h0));
                else if (h1) cnt = __builtin_amdgcn_readlane(st_tot[1], __builtin_ctzll(h1));
                if (cnt == best) second = k;
            }
            if (second < 0) second = ' ';
            }
        }
    }
    if (second != ref) {   // c == mostFrequent is tested first in the reference (:899-936): nothing is left for an equal second
        const unsigned long long s0 = __ballot(lane < nseen && sc[0] == second), s1 = __ballot(lane + 64 < nseen && sc[1] == second);
        if (s0) { const int l = __builtin_ctzll(s0); r.n10 = __builtin_amdgcn_readlane(st_pos[0], l); r.n00 = __builtin_amdgcn_readlane(st_neg[0], l); }
        else if (s1) { const int l = __builtin_ctzll(s1); r.n10 = __builtin_amdgcn_readlane(st_pos[1], l); r.n00 = __builtin_amdgcn_readlane(st_neg[1], l); }
    }
    return r;
}

// One workgroup of 16 wavefronts per listed column (the columns k_column_partition_lanes left undecided, or every column when
// `list` is null): wavefront w takes the partitions w, w + 16, ... of the column's contig, the verdict is the OR.
__global__ __launch_bounds__(1024) void k_column_partition_test(
    const int64_t* __restrict__ col_off, const int32_t* __restrict__ col_idx, const uint8_t* __restrict__ col_code,
    const int32_t* __restrict__ col_contig, const uint8_t* __restrict__ col_k0, const uint8_t* __restrict__ col_k1,
    const int32_t* __restrict__ col_c1, const uint8_t* __restrict__ col_is_cand, int n_cols,
    const int32_t* __restrict__ part_off /* [C+1] */, const int64_t* __restrict__ part_state_off /* [sum F] */,
    const int8_t* __restrict__ part_state, uint8_t* __restrict__ keep, const int32_t* __restrict__ list, const int32_t* __restrict__ n_list) {
    __shared__ uint8_t s_seen[16][128];
    __shared__ uint8_t s_ord[16][264];
    __shared__ uint8_t s_map[16][3 * 512];
    __shared__ int s_ord_n[16];
    __shared__ int s_kept;
    const int lane = lane_id();
    const int wv = wave_id();
    const int total = list ? *n_list : n_cols;
    for (int f = (int)blockIdx.x; f < total; f += (int)gridDim.x) {
        const int col = list ? list[f] : f;
        if (threadIdx.x == 0) s_kept = 0;
        __syncthreads();
        const int c = col_contig[col];
        const int p0 = part_off[c], p1 = part_off[c + 1];
        const int64_t e0 = col_off[col];
        const int n = (int)(col_off[col + 1] - e0);
        const int32_t* __restrict__ idx = col_idx + e0;
        const uint8_t* __restrict__ code = col_code + e0;
        const int k0 = col_k0[col], k1 = col_k1[col];
        const bool loop_c = col_is_cand[col] != 0;                                   // loop C (:721-738)
        const bool loop_d = (col_c1[col] & 0xffff) >= 5 && central_base_test_dev(k0, k1);     // loop D (:745-764) on the columns that can be rescued
        bool kept = false;
        if (loop_c || loop_d) {
            for (int p = p0 + wv; p < p1 && !kept; p += 16) {
                const Table2x2 d = column_vs_partition_dev(idx, code, n, part_state + part_state_off[p], k0, s_seen[wv], s_ord[wv], &s_ord_n[wv], s_map[wv]);
                const float chi = chi_square_dev(d);
                if (loop_c && (double)(d.n00 + d.n01 + d.n10 + d.n11) > 0.5 * (double)n && chi > 15) kept = true;
                if (loop_d && (double)chi > 20.0 && d.n10 + d.n00 > 4 && d.n01 + d.n11 > 4) kept = true;
            }
        }
        if (kept && lane == 0) atomicOr(&s_kept, 1);
        __syncthreads();
        if (threadIdx.x == 0) keep[col] = s_kept ? 1 : 0;
        __syncthreads();
    }
}

// The same test for listed (column, partition) PAIRS, one wavefront each: the pairs k_column_partition_grouped could not settle (the
// candidates for the second allele give different verdicts, so the reference's order of equal counts decides). A pair that keeps its
// column writes keep = 1; nobody writes 0 (the grouped kernel has).
__global__ __launch_bounds__(256) void k_column_partition_pairs(
    const int64_t* __restrict__ col_off, const int32_t* __restrict__ col_idx, const uint8_t* __restrict__ col_code,
    const int32_t* __restrict__ col_contig, const uint8_t* __restrict__ col_k0, const uint8_t* __restrict__ col_k1,
    const int32_t* __restrict__ col_c1, const uint8_t* __restrict__ col_is_cand, const int32_t* __restrict__ part_off,
    const int64_t* __restrict__ part_state_off, const int8_t* __restrict__ part_state, uint8_t* __restrict__ keep,
    const int2* __restrict__ pair_list, const int32_t* __restrict__ n_pairs, int pair_cap) {
    __shared__ uint8_t s_seen[4][128];
    __shared__ uint8_t s_ord[4][264];
    __shared__ uint8_t s_map[4][3 * 512];
    __shared__ int s_ord_n[4];
    const int wv = wave_id();
    const int total = *n_pairs < pair_cap ? *n_pairs : pair_cap;
    for (int f = (int)blockIdx.x * 4 + wv; f < total; f += (int)gridDim.x * 4) {
        const int2 pr = pair_list[f];
        const int col = pr.x;
        if (keep[col] == 1) continue;      // (another partition has kept the column already)
        const int c = col_contig[col];
        const int p = part_off[c] + pr.y;
        const int64_t e0 = col_off[col];
        const int n = (int)(col_off[col + 1] - e0);
        const int k0 = col_k0[col], k1 = col_k1[col];
        const bool loop_c = col_is_cand[col] != 0;
        const bool loop_d = (col_c1[col] & 0xffff) >= 5 && central_base_test_dev(k0, k1);
        const Table2x2 d = column_vs_partition_dev(col_idx + e0, col_code + e0, n, part_state + part_state_off[p], k0, s_seen[wv], s_ord[wv], &s_ord_n[wv], s_map[wv]);
        const float chi = chi_square_dev(d);
        bool kept = false;
        if (loop_c && (double)(d.n00 + d.n01 + d.n10 + d.n11) > 0.5 * (double)n && chi > 15) kept = true;
        if (loop_d && (double)chi > 20.0 && d.n10 + d.n00 > 4 && d.n01 + d.n11 > 4) kept = true;
        if (kept && lane_id() == 0) keep[col] = 1;
    }
}

// ------------------------------------------------------------------------------------------------
// K4, fast form (k_column_partition_lanes below): LANES = (column, partition) PAIRS THAT SHARE A READ.
// The partition states are read from a per-contig table transposed to [read][partition] (k_partition_transpose), one
// byte per (read, partition) holding 8 x {0 absent, 1 present with state 0, 2 state +1, 3 state -1}, rows padded to 16
// partitions; beside it one 16-bit word per (read, block of 16 partitions) saying in which of them the read is present:
// the OR of those words over a column's reads names the partitions whose table with the column is not empty -- the only
// ones that can keep it (call_variants.cpp:817-828: no shared read, no verdict).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_partition_transpose(
    const int32_t* __restrict__ part_off, const int64_t* __restrict__ part_state_off, const int8_t* __restrict__ part_state,
    const int32_t* __restrict__ ctg_n, const int64_t* __restrict__ tab_off, int n_contigs, uint8_t* __restrict__ tab, uint16_t* __restrict__ pres) {
    const int c = (int)blockIdx.y;
    if (c >= n_contigs) return;
    const int p0 = part_off[c], P = part_off[c + 1] - p0;
    const int N = ctg_n[c];
    const int nblk = (P + 15) >> 4;
    const int64_t total = (int64_t)N * nblk;
    uint4* __restrict__ rows = reinterpret_cast<uint4*>(tab + tab_off[c]);
    uint16_t* __restrict__ pr = pres + (tab_off[c] >> 4);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int blk = (int)(i / N), r = (int)(i % N);      // (reads fastest: the loads of one partition's states are consecutive)
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        uint32_t mask = 0u;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int p = blk * 16 + q;
            if (p < P) {
                const int st = part_state[part_state_off[p0 + p] + r];
                const uint32_t e = st == 2 ? 0u : (st == 0 ? 8u : (st == 1 ? 16u : 24u));
                if (e) mask |= 1u << q;
                w[q >> 2] |= e << (8 * (q & 3));
            }
        }
        rows[(int64_t)r * nblk + blk] = make_uint4(w[0], w[1], w[2], w[3]);
        pr[(int64_t)r * nblk + blk] = (uint16_t)mask;
    }
}

// ------------------------------------------------------------------------------------------------
// V5 distance(Partition&, Partition&, threshold_p) of loop B (call_variants.cpp:977-1127) for a list of partition pairs of the same
// contig: one wavefront per pair, lanes = reads. A read counts when both partitions hold it with more than one vote; the two
// phasings' tables are mirror images (same-sign reads are n11 / n00 of phasing 0 and n10 / n01 of phasing 1, opposite-sign reads
// the other way round), so nine per-lane counters and their wave sums give everything: the four counts of the better phasing, the
// phasing, and `augmented` (fewer than threshold_p surely divergent reads on either side ...). The 3-sigma threshold of a read
// (0.5 n + 3 sqrt(0.25 n), double then float) comes from the host's table for n < 4096; a pair with a larger vote count is marked
// not valid and left to the host. out[k] = {n00, n01, n10, n11, phased, augmented, valid, comparable}.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_partition_pair_distance(
    const int8_t* __restrict__ state, const int32_t* __restrict__ more, const int32_t* __restrict__ less, const int64_t* __restrict__ part_off,
    const int32_t* __restrict__ part_n, const int32_t* __restrict__ pair_a, const int32_t* __restrict__ pair_b, int n_pairs, int threshold_p,
    const float* __restrict__ sigma3 /* [4096] */, int32_t* __restrict__ out) {
    const int lane = lane_id();
    const int k = (int)blockIdx.x * 4 + wave_id();
    if (k >= n_pairs) return;
    const int a = pair_a[k], b = pair_b[k];
    const int N = part_n[a];
    const int8_t* __restrict__ sa = state + part_off[a]; const int8_t* __restrict__ sb = state + part_off[b];
    const int32_t* __restrict__ ma = more + part_off[a]; const int32_t* __restrict__ mb = more + part_off[b];
    const int32_t* __restrict__ la = less + part_off[a]; const int32_t* __restrict__ lb = less + part_off[b];
    int comparable = 0, n_pp = 0, n_mm = 0, n_pm = 0, n_mp = 0;      // (state of b, state of a): ++, --, +-, -+
    int div_same = 0, div_opp = 0, uns_same = 0, uns_opp = 0;
    bool big = false;
    for (int r = lane; r < N; r += 64) {
        const int s1 = sa[r], s2 = sb[r];
        if (s1 == 2 || s2 == 2) continue;                              // absent from one of them
        const int m1 = ma[r], m2 = mb[r];
        if (!(m1 > 1 && m2 > 1)) continue;
        comparable++;
        const int t1n = m1 + la[r], t2n = m2 + lb[r];
        if (t1n >= 4096 || t2n >= 4096) { big = true; continue; }
        const float t1 = sigma3[t1n], t2 = sigma3[t2n];
        const bool c1 = (float)m1 > t1, c2 = (float)m2 > t2;
        const bool both = c1 && c2, either = c1 || c2;
        if (s2 == 1) {
            if (s1 == 1) { n_pp++; div_same += both; uns_same += either; }
            else if (s1 == -1) { n_pm++; div_opp += both; uns_opp += either; }
        } else if (s2 == -1) {
            if (s1 == 1) { n_mp++; div_opp += both; uns_opp += either; }
            else if (s1 == -1) { n_mm++; div_same += both; uns_same += either; }
        }
    }
    comparable = wave_sum_i32(comparable);
    n_pp = wave_sum_i32(n_pp); n_mm = wave_sum_i32(n_mm); n_pm = wave_sum_i32(n_pm); n_mp = wave_sum_i32(n_mp);
    div_same = wave_sum_i32(div_same); div_opp = wave_sum_i32(div_opp); uns_same = wave_sum_i32(uns_same); uns_opp = wave_sum_i32(uns_opp);
    const bool any_big = __ballot(big) != 0ull;
    if (lane == 0) {
        // phasing 0: n11 = ++, n01 = +- (b plus, a minus), n10 = -+, n00 = --; phasing 1: n10 = ++, n00 = +-, n11 = -+, n01 = --
        const int score0 = (n_pp + n_mm) - (n_pm + n_mp);
        const int kk = -score0 > score0 ? 1 : 0;
        const short ndiv0 = (short)div_opp, ndiv1 = (short)div_same, nuns0 = (short)uns_opp, nuns1 = (short)uns_same;      // (the reference counts in shorts)
        const bool augmented = !((ndiv0 >= threshold_p && ndiv1 >= threshold_p) || (nuns0 >= 5 && nuns1 >= 5) || comparable == 0);
        int32_t* o = out + (int64_t)k * 8;
        o[0] = kk ? n_pm : n_mm;      // n00
        o[1] = kk ? n_mm : n_pm;      // n01
        o[2] = kk ? n_pp : n_mp;      // n10
        o[3] = kk ? n_mp : n_pp;      // n11
        o[4] = -2 * kk + 1; o[5] = augmented ? 1 : 0; o[6] = any_big ? 0 : 1; o[7] = comparable;
    }
}

// ------------------------------------------------------------------------------------------------
// K4, first form (k_column_partition_lanes): 16 lanes per column = the 16 partitions of a block of the table, four columns per wavefront,
// NO grouping of the entries by code. The four columns' entries go to LDS as they lie (lanes = entries: row offset of the read in the
// table, class: reference code / the column's own second code k1 / another code); the lanes of a group then walk their column eight
// entries at a time (four 16-byte LDS reads, eight table bytes in flight, each 16 consecutive bytes per group) and count in one
// accumulator of 5-bit fields: the reference code's plus / minus reads (-> n11 / n01), k1's zero / plus / minus reads, the other codes'
// present reads together. Where the partition holds more reads with k1 than any other non-reference code can have -- all of them
// together, or the column's third count --, k1 IS its most frequent other code (call_variants.cpp:832-844), no tie possible, and the
// table is exact: chi-square and the verdicts of loops C and D (:721-764) follow as in the grouped kernel below. Where it does not, the
// table is unknown but bounded (n10 + n00 <= that bound or k1's reads): a pair whose bound cannot reach loop C's "more than half of the
// column's reads" nor loop D's five reads per side is settled as well; the rest are marked (keep = 2) for k_column_partition_grouped.
// Columns deeper than 255, with a reference code >= 128, or that do not fit the wavefront's 512 LDS entries go there too.
// ------------------------------------------------------------------------------------------------
#define HS_K4L_WAVE 512     // entries of its four columns a wavefront has room for in LDS (each column rounded up to eight)
// per entry class (reference code / the column's second code k1 / any other code) a word of four shift amounts, one per table byte
// 0 / 8 / 16 / 24 = absent / zero / plus / minus: the 5-bit field of the accumulator the entry counts in
//   0 reference plus (n11), 5 reference minus (n01), 10 / 15 / 20 k1 zero / plus / minus, 25 another code and present, 30 nothing
#define HS_K4L_REF 0x05001E1Eu
#define HS_K4L_K1 0x140F0A1Eu
#define HS_K4L_OTHER 0x1919191Eu
#define HS_K4L_NONE 0x1E1E1E1Eu
__global__ __launch_bounds__(256) void k_column_partition_lanes(
    const int64_t* __restrict__ col_off, const int32_t* __restrict__ col_idx, const uint8_t* __restrict__ col_code,
    const int32_t* __restrict__ col_contig, const uint8_t* __restrict__ col_k0, const uint8_t* __restrict__ col_k1,
    const int32_t* __restrict__ col_c1, const uint8_t* __restrict__ col_is_cand, int n_cols,
    const int32_t* __restrict__ part_off, const int32_t* __restrict__ ctg_n, const int64_t* __restrict__ tab_off, const uint8_t* __restrict__ tab,
    uint8_t* __restrict__ keep, int32_t* __restrict__ grouped_list, int32_t* __restrict__ n_grouped) {
    __shared__ __attribute__((aligned(16))) uint32_t s_off[4][HS_K4L_WAVE];      // per entry: where its read's row starts in the contig's table
    __shared__ __attribute__((aligned(16))) uint32_t s_cls[4][HS_K4L_WAVE];      // per entry: the shift amounts of its class
    const int lane = lane_id();
    const int wv = wave_id();
    const int grp = lane >> 4, pl = lane & 15;
    const int col = (((int)blockIdx.x * 4 + wv) * 4) + grp;
    bool tested = false, passed_on = false;
    int n = 0, k0 = 0, k1 = 0, P = 0, c2 = 0;
    bool is_cand = false, loop_d = false;
    int64_t e0 = 0;
    uint32_t tb = 0u, ppad = 0u;
    if (col < n_cols) {
        const int c = col_contig[col];
        P = part_off[c + 1] - part_off[c];
        e0 = col_off[col];
        n = (int)(col_off[col + 1] - e0);
        k0 = col_k0[col]; k1 = col_k1[col];
        is_cand = col_is_cand[col] != 0;
        const int c1w = col_c1[col];
        loop_d = (c1w & 0xffff) >= 5 && central_base_test_dev(k0, k1);
        c2 = 63 - ((c1w >> 16) & 63);      // the column's third count (63: that or more, or not known)
        if (c2 >= 63) c2 = 1 << 20;
        ppad = (uint32_t)((P + 15) & ~15);
        const int64_t t0 = tab_off[c];
        tb = (uint32_t)t0;
        if (P == 0 || !(is_cand || loop_d) || n == 0) { if (pl == 0) keep[col] = 0; }      // (an empty column shares no read with anything)
        else if (n > 255 || k0 >= 128 || t0 + (int64_t)ctg_n[c] * ppad > 0xffffffffll) passed_on = true;
        else tested = true;
    }
    if (__ballot(tested || passed_on) == 0ull) return;
    // ---- the four columns' entries into LDS, lanes = entries (the loads of all four first); a column the wavefront has no room left for is passed on ----
    int row0 = 0;      // where this group's column starts
    {
        int ng[4], og[4]; int64_t eg[4]; uint32_t pg[4]; int kr[4], kk[4]; int32_t rr[4]; int cd[4];
        int used = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            ng[g] = __builtin_amdgcn_readlane(tested ? n : 0, 16 * g);
            og[g] = used;
            if (used + ((ng[g] + 7) & ~7) > HS_K4L_WAVE) { ng[g] = 0; if (grp == g && tested) { tested = false; passed_on = true; } }
            used += (ng[g] + 7) & ~7;
            if (grp == g) row0 = og[g];
            eg[g] = ((int64_t)(unsigned)__builtin_amdgcn_readlane((int)(e0 >> 32), 16 * g) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(e0 & 0xffffffffll), 16 * g);
            pg[g] = (uint32_t)__builtin_amdgcn_readlane((int)ppad, 16 * g);
            kr[g] = __builtin_amdgcn_readlane(k0, 16 * g); kk[g] = __builtin_amdgcn_readlane(k1, 16 * g);
            rr[g] = 0; cd[g] = -1;
            if (lane < ng[g]) { rr[g] = col_idx[eg[g] + lane]; cd[g] = (int)col_code[eg[g] + lane]; }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (lane < ((ng[g] + 7) & ~7)) {      // (the tail of the last eight counts nowhere)
                s_off[wv][og[g] + lane] = (uint32_t)rr[g] * pg[g];
                s_cls[wv][og[g] + lane] = cd[g] < 0 ? HS_K4L_NONE : (cd[g] == kr[g] ? HS_K4L_REF : (cd[g] == kk[g] ? HS_K4L_K1 : HS_K4L_OTHER));
            }
            for (int e = 64 + lane; e < ((ng[g] + 7) & ~7); e += 64) {      // (deeper than 64: rare)
                const bool in = e < ng[g];
                const int code = in ? (int)col_code[eg[g] + e] : -1;
                s_off[wv][og[g] + e] = in ? (uint32_t)col_idx[eg[g] + e] * pg[g] : 0u;
                s_cls[wv][og[g] + e] = code < 0 ? HS_K4L_NONE : (code == kr[g] ? HS_K4L_REF : (code == kk[g] ? HS_K4L_K1 : HS_K4L_OTHER));
            }
        }
    }
    wave_lds_sync();
    const uint32_t* __restrict__ so = s_off[wv] + row0;
    const uint32_t* __restrict__ sc = s_cls[wv] + row0;
    const unsigned gshift = 16u * (unsigned)grp;
    const int n8 = (n + 7) & ~7;
    bool kept = false, unsettled = false;
    for (uint32_t pb = 0u; __ballot(tested && !kept && pb < ppad) != 0ull; pb += 16u) {
        const bool on = tested && !kept && pb < ppad;
        const uint32_t base = tb + pb + (uint32_t)pl;
        uint32_t acc = 0u;                       // six 5-bit fields: spilled into the wide counters before one can overflow (every 24 entries)
        uint32_t w_r = 0u, w_k = 0u, w_o = 0u;   // reference plus | minus << 16; k1 zero | plus << 10 | minus << 20; the others, present
        for (int e = 0; __ballot(on && e < n8) != 0ull; e += 16) {
            if (on && e < n8) {      // sixteen entries at a time where the column has them (the second eight under their own test: n8 is a multiple of 8)
                const bool two = e + 8 < n8;
                const uint4 oa = *reinterpret_cast<const uint4*>(so + e), ob = *reinterpret_cast<const uint4*>(so + e + 4);
                const uint4 ca = *reinterpret_cast<const uint4*>(sc + e), cb = *reinterpret_cast<const uint4*>(sc + e + 4);
                uint4 oc = make_uint4(0u, 0u, 0u, 0u), od = oc, cc = make_uint4(HS_K4L_NONE, HS_K4L_NONE, HS_K4L_NONE, HS_K4L_NONE), cd = cc;
                if (two) { oc = *reinterpret_cast<const uint4*>(so + e + 8); od = *reinterpret_cast<const uint4*>(so + e + 12); cc = *reinterpret_cast<const uint4*>(sc + e + 8); cd = *reinterpret_cast<const uint4*>(sc + e + 12); }
                const uint32_t b0 = tab[base + oa.x], b1 = tab[base + oa.y], b2 = tab[base + oa.z], b3 = tab[base + oa.w];
                const uint32_t b4 = tab[base + ob.x], b5 = tab[base + ob.y], b6 = tab[base + ob.z], b7 = tab[base + ob.w];
                const uint32_t b8 = tab[base + oc.x], b9 = tab[base + oc.y], b10 = tab[base + oc.z], b11 = tab[base + oc.w];
                const uint32_t b12 = tab[base + od.x], b13 = tab[base + od.y], b14 = tab[base + od.z], b15 = tab[base + od.w];
                acc += (1u << __builtin_amdgcn_ubfe(ca.x, b0, 5u)) + (1u << __builtin_amdgcn_ubfe(ca.y, b1, 5u)) + (1u << __builtin_amdgcn_ubfe(ca.z, b2, 5u)) + (1u << __builtin_amdgcn_ubfe(ca.w, b3, 5u));
                acc += (1u << __builtin_amdgcn_ubfe(cb.x, b4, 5u)) + (1u << __builtin_amdgcn_ubfe(cb.y, b5, 5u)) + (1u << __builtin_amdgcn_ubfe(cb.z, b6, 5u)) + (1u << __builtin_amdgcn_ubfe(cb.w, b7, 5u));
                acc += (1u << __builtin_amdgcn_ubfe(cc.x, b8, 5u)) + (1u << __builtin_amdgcn_ubfe(cc.y, b9, 5u)) + (1u << __builtin_amdgcn_ubfe(cc.z, b10, 5u)) + (1u << __builtin_amdgcn_ubfe(cc.w, b11, 5u));
                acc += (1u << __builtin_amdgcn_ubfe(cd.x, b12, 5u)) + (1u << __builtin_amdgcn_ubfe(cd.y, b13, 5u)) + (1u << __builtin_amdgcn_ubfe(cd.z, b14, 5u)) + (1u << __builtin_amdgcn_ubfe(cd.w, b15, 5u));
            }
            {      // (at most 16 per field in one round: spilled every round, wave-uniform)
                w_r += (acc & 31u) | (((acc >> 5) & 31u) << 16);
                w_k += ((acc >> 10) & 31u) | (((acc >> 15) & 31u) << 10) | (((acc >> 20) & 31u) << 20);
                w_o += (acc >> 25) & 31u;
                acc = 0u;
            }
        }
        w_r += (acc & 31u) | (((acc >> 5) & 31u) << 16);
        w_k += ((acc >> 10) & 31u) | (((acc >> 15) & 31u) << 10) | (((acc >> 20) & 31u) << 20);
        w_o += (acc >> 25) & 31u;
        bool ok = false, open = false;
        if (on && (int)(pb + (uint32_t)pl) < P) {
            const int n11 = (int)(w_r & 0xffffu), n01 = (int)(w_r >> 16);
            const int n10 = (int)((w_k >> 10) & 1023u), n00 = (int)(w_k >> 20);
            const int take_k = (int)(w_k & 1023u) + n10 + n00, rest = (int)w_o;
            const int other_ub = rest < c2 ? rest : c2;      // what one other code can have here at most: all of them together, or the column's third count
            if (take_k > other_ub) {      // k1 is the partition's most frequent other code, strictly
                const int total = n00 + n01 + n10 + n11;
                const bool pre_c = is_cand && 2 * total > n;                       // loop C (:721-738): (double)total > 0.5 * (double)n
                const bool pre_d = loop_d && n10 + n00 > 4 && n01 + n11 > 4;       // loop D (:745-764)
                const int r1 = n10 + n11, c1 = n01 + n11;
                if ((pre_c || pre_d) && r1 > 0 && r1 < total && c1 > 0 && c1 < total) {      // (a margin of 0 or all: chi-square is -1 or 0)
                    const float det = (float)(n11 * n00 - n10 * n01);
                    float chi = (float)total * det * det * __builtin_amdgcn_rcpf((float)((r1 * (total - r1)) * (c1 * (total - c1))));
                    const bool near = (pre_c && fabsf(chi - 15.0f) < 0.05f) || (pre_d && fabsf(chi - 20.0f) < 0.05f);
                    if (near) { Table2x2 d; d.n00 = n00; d.n01 = n01; d.n10 = n10; d.n11 = n11; chi = chi_square_dev(d); }
                    ok = (pre_c && chi > 15) || (pre_d && (double)chi > 20.0);
                }
            } else {
                const int ub = other_ub > take_k ? other_ub : take_k;      // what the second allele can have at most
                open = ub > 0 && ((is_cand && 2 * (n11 + n01 + ub) > n) || (loop_d && ub >= 5 && n01 + n11 > 4));
            }
        }
        kept = kept || ((__ballot(ok) >> gshift) & 0xffffull) != 0ull;
        unsettled = unsettled || ((__ballot(open) >> gshift) & 0xffffull) != 0ull;
    }
    const bool pass = (tested || passed_on) && pl == 0 && (passed_on || (!kept && unsettled));
    if ((tested || passed_on) && pl == 0) keep[col] = kept ? 1 : (pass ? 2 : 0);
    const unsigned long long pm = __ballot(pass);      // the columns left to k_column_partition_grouped: one atomic per wavefront
    if (pm) {
        int at = 0;
        if (lane == 0) at = atomicAdd(n_grouped, __popcll(pm));
        at = __builtin_amdgcn_readfirstlane(at);
        if (pass) grouped_list[at + __popcll(pm & ((1ull << lane) - 1ull))] = col;
    }
}

// K4, second form (k_column_partition_grouped): the columns k_column_partition_lanes could not settle with its counters (see there), four
// of its list per wavefront (= one workgroup) at a time.
//  1. the first lanes read the headers and decide which columns are tested at all;
//  2. column by column, lanes = entries: the entries' read indices go to LDS grouped by code -- the reference code first (ballot ranks),
//     the other codes behind it from the next multiple of 8 by a counting sort on LDS counters (rank = the old value of an atomic add,
//     starts = one wave scan over the 128 counters) with one bit per entry saying "last of its code" --, and per block of 16 partitions
//     the OR of the reads' presence words names the PAIRS {column, partition that shares a read with it};
//  3. 64 pending pairs at a time, one per lane: the lane walks its column's grouped entries eight at a time (one 16-byte LDS read, eight
//     table bytes in flight, 1 << byte added to an accumulator whose byte fields count absent / zero / plus / minus reads): the
//     reference code -> n11 / n01, then the others with the accumulator evaluated at every code's last entry: the table this code would
//     give as the second allele and its verdict for loops C and D (:721-764, k4_verdict); the codes the partition holds most reads of
//     are the candidates for the second allele (call_variants.cpp:832-936; the reference breaks their tie by hash-map order).
// Not decided here (keep = 2, re-done by the exact kernel above): a column with a partition whose tied candidates give different
// verdicts, reference codes >= 128 (signed-char quirk), columns
// deeper than 255, codes outside 33..160, contigs with more than 65535 reads or partitions or a table beyond 4 GB.
// the verdict of loops C (:721-738) and D (:745-764) on one table: chi-square as N (ad - bc)^2 / (r1 r2 c1 c2) in float unless that comes
// within 0.05 of a threshold, then the reference's own sequence of float and double operations (chi_square_dev)
static __device__ __noinline__ bool k4_verdict(int n11, int n01, int n10, int n00, int n, bool is_cand, bool loop_d) {
    const int total = n00 + n01 + n10 + n11;
    const bool pre_c = is_cand && 2 * total > n;                       // (double)total > 0.5 * (double)n
    const bool pre_d = loop_d && n10 + n00 > 4 && n01 + n11 > 4;
    const int r1 = n10 + n11, c1 = n01 + n11;
    if (!((pre_c || pre_d) && r1 > 0 && r1 < total && c1 > 0 && c1 < total)) return false;      // (a margin of 0 or all: chi-square is -1 or 0)
    const float det = (float)(n11 * n00 - n10 * n01);
    float chi = (float)total * det * det * __builtin_amdgcn_rcpf((float)((r1 * (total - r1)) * (c1 * (total - c1))));
    const bool near = (pre_c && fabsf(chi - 15.0f) < 0.05f) || (pre_d && fabsf(chi - 20.0f) < 0.05f);
    if (near) { Table2x2 d; d.n00 = n00; d.n01 = n01; d.n10 = n10; d.n11 = n11; chi = chi_square_dev(d); }
    return (pre_c && chi > 15) || (pre_d && (double)chi > 20.0);
}
#define HS_K4_COLS 4              // columns a wavefront takes at a time: their pairs fill about one round of 64
#define HS_K4_ROW 264             // u16 entries of a column's row in LDS: 255 entries + the gap behind the reference code's + the tail
#define HS_K4_PAIRS 80
static __device__ __forceinline__ int wave_or_i32(int v) {
    v |= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return __builtin_amdgcn_readlane(v, 63);
}
__global__ __launch_bounds__(64) void k_column_partition_grouped(
    const int64_t* __restrict__ col_off, const int32_t* __restrict__ col_idx, const uint8_t* __restrict__ col_code,
    const int32_t* __restrict__ col_contig, const uint8_t* __restrict__ col_k0, const uint8_t* __restrict__ col_k1,
    const int32_t* __restrict__ col_c1, const uint8_t* __restrict__ col_is_cand, const int32_t* __restrict__ list, const int32_t* __restrict__ n_list,
    const int32_t* __restrict__ part_off, const int32_t* __restrict__ ctg_n, const int64_t* __restrict__ tab_off, const uint8_t* __restrict__ tab,
    const uint16_t* __restrict__ pres, uint8_t* __restrict__ keep, int32_t* __restrict__ undecided_list, int32_t* __restrict__ n_undecided,
    int2* __restrict__ pair_list /* {column, partition of its contig}: the pairs whose candidates disagree */, int32_t* __restrict__ n_pairs, int pair_cap) {
    __shared__ __attribute__((aligned(16))) uint16_t s_idx[HS_K4_COLS * HS_K4_ROW];
    __shared__ int s_col[HS_K4_COLS];
    __shared__ uint32_t s_last[HS_K4_COLS][9];      // bit e: entry e of the row is the last of its code (other codes only)
    __shared__ uint32_t s_cnt[128];
    __shared__ uint32_t s_pairs[HS_K4_PAIRS];       // slot << 28 | block << 4 | partition within the block
    __shared__ uint4 s_hdr[HS_K4_COLS];             // {table offset of the contig, row length, n | flags << 16, entries with the reference code | end of the others << 16}
    __shared__ uint32_t s_flags;                    // bit s: column s kept; bit 16 + s: undecided
    const int lane = lane_id();
    const int n_listed = *n_list;
    for (int first = (int)blockIdx.x * HS_K4_COLS; first < n_listed; first += (int)gridDim.x * HS_K4_COLS) {
    // ---- the column headers: lane l < HS_K4_COLS holds the l-th listed column of this round ----
    const bool hvalid = lane < HS_K4_COLS && first + lane < n_listed;
    const int hc = hvalid ? list[first + lane] : 0;
    int h_n = 0, h_k0 = 0, h_flags = 0, h_ppad = 0;
    uint32_t h_tb = 0u;
    int64_t h_e0 = 0;
    bool tested = false, h_bad = false;
    if (hvalid) {
        const int c = col_contig[hc];
        const int P = part_off[c + 1] - part_off[c];
        h_e0 = col_off[hc];
        h_n = (int)(col_off[hc + 1] - h_e0);
        h_k0 = col_k0[hc];
        const int k1 = col_k1[hc];
        const bool is_cand = col_is_cand[hc] != 0;
        const bool loop_d = (col_c1[hc] & 0xffff) >= 5 && central_base_test_dev(h_k0, k1);
        h_flags = (is_cand ? 1 : 0) | (loop_d ? 2 : 0);
        h_ppad = (P + 15) & ~15;
        const int64_t tb = tab_off[c];
        const int N = ctg_n[c];
        h_tb = (uint32_t)tb;
        if (P == 0 || h_flags == 0 || h_n == 0) keep[hc] = 0;      // (an empty column shares no read with anything)
        else if (h_n > 255 || h_k0 >= 128 || N > 65535 || P > 65535 || tb + (int64_t)N * h_ppad > 0xffffffffll) h_bad = true;
        else tested = true;
    }
    if (lane < HS_K4_COLS) { s_hdr[lane] = make_uint4(h_tb, (uint32_t)h_ppad, (uint32_t)h_n | ((uint32_t)h_flags << 16), 0u); s_col[lane] = hc; }
    if (lane == 0) s_flags = 0u;
    unsigned bad_cols = (unsigned)(__ballot(h_bad) & ((1ull << HS_K4_COLS) - 1ull));
    unsigned todo = (unsigned)(__ballot(tested) & ((1ull << HS_K4_COLS) - 1ull));
    int npairs = 0;
    int cur_s = -1;
    uint32_t cur_blk = 0u, cur_nblk = 0u;
    int cur_nch = 0;
    uint32_t prow[4] = {0u, 0u, 0u, 0u};      // per entry of the current column: where its presence words start
    bool pvalid[4] = {false, false, false, false};
    for (;;) {
        // ---- step 2: columns (and their blocks of 16 partitions) until 64 pairs are pending ----
        while (npairs < 64) {
            if (cur_s < 0) {
                if (!todo) break;
                const int s = __builtin_ctz(todo);
                todo &= todo - 1u;
                const int64_t e0 = ((int64_t)(unsigned)__builtin_amdgcn_readlane((int)(h_e0 >> 32), s) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(h_e0 & 0xffffffffll), s);
                const int n = __builtin_amdgcn_readlane(h_n, s), k0 = __builtin_amdgcn_readlane(h_k0, s);
                const uint32_t tb16 = (uint32_t)__builtin_amdgcn_readlane((int)h_tb, s) >> 4, ppb = (uint32_t)__builtin_amdgcn_readlane(h_ppad, s) >> 4;
                const int nch = (n + 63) >> 6;
                int r_i[4], c_i[4], rank[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = k * 64 + lane;
                    r_i[k] = 0; c_i[k] = -1; rank[k] = 0;
                    if (k < nch && e < n) { r_i[k] = col_idx[e0 + e]; c_i[k] = (int)col_code[e0 + e]; }      // (k < nch: wave-uniform, most columns are one chunk)
                }
                reinterpret_cast<uint2*>(s_cnt)[lane] = make_uint2(0u, 0u);
                if (lane < 9) s_last[s][lane] = 0u;
                wave_lds_sync();
                uint16_t* __restrict__ ix = s_idx + s * HS_K4_ROW;
                int nref = 0;
                bool strange = false;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k >= nch) break;
                    const bool isref = c_i[k] == k0;
                    const unsigned long long m = __ballot(isref);
                    if (isref) ix[nref + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)r_i[k];
                    nref += __popcll(m);
                    if (c_i[k] >= 0 && !isref) {
                        if (c_i[k] < 33 || c_i[k] > 160) strange = true;
                        else rank[k] = (int)atomicAdd(&s_cnt[c_i[k] - 33], 1u);
                    }
                }
                wave_lds_sync();
                const uint2 ab = reinterpret_cast<const uint2*>(s_cnt)[lane];
                const int both = (int)(ab.x + ab.y);
                const int incl = wave_scan_incl(both);
                const int o0 = (nref + 7) & ~7, oend = o0 + __builtin_amdgcn_readlane(incl, 63);
                const uint32_t start_a = (uint32_t)(o0 + incl - both), start_b = start_a + ab.x;
                if (ab.x > 0u) { const uint32_t last = start_a + ab.x - 1u; atomicOr(&s_last[s][last >> 5], 1u << (last & 31u)); }
                if (ab.y > 0u) { const uint32_t last = start_b + ab.y - 1u; atomicOr(&s_last[s][last >> 5], 1u << (last & 31u)); }
                reinterpret_cast<uint2*>(s_cnt)[lane] = make_uint2(start_a, start_b);
                if (lane < 8) {      // the gap behind the reference code's entries and the tail of the last eight: read 0
                    if (nref + lane < o0) ix[nref + lane] = 0;
                    if (oend + lane < ((oend + 7) & ~7)) ix[oend + lane] = 0;
                }
                if (lane == 0) s_hdr[s].w = (uint32_t)nref | ((uint32_t)oend << 16);
                wave_lds_sync();
                if (__ballot(strange) != 0ull) { bad_cols |= 1u << s; continue; }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k >= nch) break;
                    if (c_i[k] >= 0 && c_i[k] != k0) ix[s_cnt[c_i[k] - 33] + (uint32_t)rank[k]] = (uint16_t)r_i[k];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { prow[k] = tb16 + (uint32_t)r_i[k] * ppb; pvalid[k] = c_i[k] >= 0; }
                cur_s = s; cur_blk = 0u; cur_nblk = ppb; cur_nch = nch;
                continue;
            }
            if (cur_blk == cur_nblk) { cur_s = -1; continue; }
            int m = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { if (k >= cur_nch) break; if (pvalid[k]) m |= (int)pres[prow[k] + cur_blk]; }
            m = wave_or_i32(m);
            if (m) {
                if (lane < 16 && ((m >> lane) & 1)) s_pairs[npairs + __popc((unsigned)m & ((1u << lane) - 1u))] = ((uint32_t)cur_s << 28) | (cur_blk << 4) | (uint32_t)lane;
                npairs += __popc((unsigned)m);
            }
            ++cur_blk;
        }
        if (npairs == 0) break;
        wave_lds_sync();
        // ---- step 3: 64 pairs, one per lane ----
        {
            const bool act = lane < npairs;
            const uint32_t pr = s_pairs[act ? lane : 0];
            const uint32_t moved = (lane + 64 < npairs) ? s_pairs[lane + 64] : 0u;
            const int slot = (int)(pr >> 28);
            const uint4 hd = s_hdr[slot];
            const uint32_t base = hd.x + (pr & 0xfffffffu), ppad = hd.y;      // (block << 4 | partition = the byte's place in the table row)
            const int n = (int)(hd.z & 255u);
            const bool is_cand = (hd.z >> 16) & 1u, loop_d = (hd.z >> 17) & 1u;
            const int nref = act ? (int)(hd.w & 0xffffu) : 0, oend = act ? (int)(hd.w >> 16) : 0;
            const int o0 = (nref + 7) & ~7;
            const uint16_t* __restrict__ ix = s_idx + slot * HS_K4_ROW;
            // the reference code's entries (the gap up to the next multiple of 8 reads row 0: taken out again below)
            uint32_t acc = 0u;
            for (int e0 = 0; __ballot(e0 < nref) != 0ull; e0 += 8) {
                if (e0 < nref) {
                    const uint4 v = *reinterpret_cast<const uint4*>(ix + e0);
                    const uint32_t b0 = tab[base + __umul24(v.x & 0xffffu, ppad)], b1 = tab[base + __umul24(v.x >> 16, ppad)];
                    const uint32_t b2 = tab[base + __umul24(v.y & 0xffffu, ppad)], b3 = tab[base + __umul24(v.y >> 16, ppad)];
                    const uint32_t b4 = tab[base + __umul24(v.z & 0xffffu, ppad)], b5 = tab[base + __umul24(v.z >> 16, ppad)];
                    const uint32_t b6 = tab[base + __umul24(v.w & 0xffffu, ppad)], b7 = tab[base + __umul24(v.w >> 16, ppad)];
                    acc += (1u << b0) + (1u << b1) + (1u << b2) + (1u << b3) + (1u << b4) + (1u << b5) + (1u << b6) + (1u << b7);
                }
            }
            if (o0 > nref) acc -= (uint32_t)(o0 - nref) << (uint32_t)tab[base];
            const int n11 = (int)((acc >> 16) & 255u), n01 = (int)(acc >> 24);
            // the other codes: the accumulator is evaluated where an entry is the last of its code -- the table this code would give as
            // the second allele and its verdict; the codes the partition holds most of are the candidates for the second allele (the
            // reference breaks their tie by hash-map order): where all of them give the same verdict, it does not matter which it takes
            acc = 0u;
            uint32_t bt = 0u;
            bool ok_all = false, ok_any = false;
            const uint8_t* __restrict__ lastb = reinterpret_cast<const uint8_t*>(s_last[slot]);
            for (int e0 = o0; __ballot(e0 < oend) != 0ull; e0 += 8) {
                if (e0 < oend) {
                    const uint4 v = *reinterpret_cast<const uint4*>(ix + e0);
                    const uint32_t fb = lastb[e0 >> 3];
                    uint32_t b[8];
                    b[0] = tab[base + __umul24(v.x & 0xffffu, ppad)]; b[1] = tab[base + __umul24(v.x >> 16, ppad)];
                    b[2] = tab[base + __umul24(v.y & 0xffffu, ppad)]; b[3] = tab[base + __umul24(v.y >> 16, ppad)];
                    b[4] = tab[base + __umul24(v.z & 0xffffu, ppad)]; b[5] = tab[base + __umul24(v.z >> 16, ppad)];
                    b[6] = tab[base + __umul24(v.w & 0xffffu, ppad)]; b[7] = tab[base + __umul24(v.w >> 16, ppad)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        acc += 1u << b[u];
                        if ((fb >> u) & 1u) {
                            const uint32_t take = __builtin_amdgcn_sad_u8(acc & 0xffffff00u, 0u, 0u);      // zero + plus + minus: the code's reads the partition holds
                            if (take >= bt && take > 0u) {
                                const bool vd = k4_verdict(n11, n01, (int)((acc >> 16) & 255u), (int)(acc >> 24), n, is_cand, loop_d);
                                if (take > bt) { bt = take; ok_all = vd; ok_any = vd; }
                                else { ok_all = ok_all && vd; ok_any = ok_any || vd; }
                            }
                            acc = 0u;
                        }
                    }
                }
            }
            const bool ok = act && ok_all, und = act && ok_any && !ok_all;
            if (ok) atomicOr(&s_flags, 1u << slot);
            {   // a pair whose candidates for the second allele disagree goes to the exact kernel as a pair (one atomic per wavefront); when
                // the list is full the whole column goes there instead
                const unsigned long long um = __ballot(und);
                if (um) {
                    int at = 0;
                    if (lane == 0) at = atomicAdd(n_pairs, __popcll(um));
                    at = __builtin_amdgcn_readfirstlane(at);
                    const int mine = at + __popcll(um & ((1ull << lane) - 1ull));
                    if (und) { if (mine < pair_cap) pair_list[mine] = make_int2(s_col[slot], (int)(pr & 0xfffffffu)); else atomicOr(&s_flags, 0x10000u << slot); }
                }
            }
            wave_lds_sync();
            if (lane + 64 < npairs) s_pairs[lane] = moved;
            npairs = npairs > 64 ? npairs - 64 : 0;
            wave_lds_sync();
        }
    }
    wave_lds_sync();
    const uint32_t fl = s_flags;
    if (hvalid && (tested || h_bad)) {
        const bool kept = (fl >> lane) & 1u;
        const bool undecided = ((fl >> (16 + lane)) & 1u) || ((bad_cols >> lane) & 1u);
        keep[hc] = kept ? 1 : (undecided ? 2 : 0);      // (0: unless one of its listed pairs says otherwise)
    }
    {   // the columns left to the exact kernel WHOLE (not testable here, or the pair list full): one atomic per wavefront
        const bool und = hvalid && (tested || h_bad) && !((fl >> lane) & 1u) && ((((fl >> (16 + lane)) & 1u) != 0u) || (((bad_cols >> lane) & 1u) != 0u));
        const unsigned long long um = __ballot(und);
        if (um) {
            int at = 0;
            if (lane == 0) at = atomicAdd(n_undecided, __popcll(um));
            at = __builtin_amdgcn_readfirstlane(at);
            if (und) undecided_list[at + __popcll(um & ((1ull << lane) - 1ull))] = hc;
        }
    }
    wave_lds_sync();
    }      // (the next listed columns)
}

}  // namespace hsdev
