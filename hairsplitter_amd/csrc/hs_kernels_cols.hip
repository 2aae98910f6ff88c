// hs_kernels_cols.hip -- the column pass of stage 3 kept on the device from the pileup to the candidate SNPs and from the
// final partitions to the SNP columns stage 4 reads (call_variants.cpp:471-536 and :721-764, :1335-1352). The host sees two
// things only: the candidate columns (loops A and B of keep_only_robust_variants are sequential per contig and go through
// libm, hs_host_cv.cpp) and the SNPs that come out.
//
//   k_columns_compact     K2's per-tile selections -> the sorted column list of a contig range with its CSR offsets
//   k_gather_tiles        K3: one wavefront per 256-position tile; the tile's pileup bytes are staged in LDS row by row
//                         (one coalesced 256-byte row per record) and every selected position reads its column out of
//                         LDS (lane = record): each pileup byte of the tile is fetched once, where the per-position form
//                         (k_gather_columns_tiled) touched one 64-byte line per entry
//   k_column_top3_exact   K3b: the two most frequent codes of every column in the reference's order of equal counts --
//                         robin_hood iteration order (hs::Rh8View) + libstdc++'s std::sort (hs::CountSort), replayed by
//                         one lane on LDS tables for the columns where counts tie; nothing goes to the host
//   k_candidates_scan     V1: the greedy spacing scan of call_variants.cpp:525-536 per contig (lanes = columns, the
//                         dependency only runs along the few columns that pass the predicate)
//   k_flag_block_sums / k_flag_block_offsets / k_pack_flagged   the flagged columns (candidates, later the SNPs) packed back to back with their records
//   k_cand_bits           the packed candidate columns as bit sets over the reads ranked by start position (what loop A reads)
//   k_ship / k_fill16     transfers and fills as kernels (pinned host memory is mapped into the device's address space)
//   k_snp_bounds / k_snp_flags   the two-pointer merge of automatic and filtered SNPs (:1335-1352) as a per-contig bound
// Included by hs_capi.hip after hs_kernels.hip.
#pragma once

namespace hsdev {

// per-column record the device keeps (== hs_colrec of include/hairsplitter_hip.h)
struct alignas(16) hs_colrec_dev {
    int32_t pos;            // position on its contig
    int32_t contig;         // contig index in the batch
    uint16_t c0, c1;        // counts of the two most frequent codes (saturate at 65535: the depth limit of the path)
    uint8_t k0, k1;         // the codes, reference order of equal counts
    uint8_t flags;          // HS_COL_*
    uint8_t c2;             // the third count, saturating at 63
};
static_assert(sizeof(hs_colrec_dev) == 16, "hs_colrec must be 16 bytes");
#define HS_COL_CAND 1
#define HS_COL_AUTO 2
#define HS_COL_LOOPD 4      // can be rescued by loop D (second count >= 5 + byte predicate)
#define HS_COL_KEEP 8       // kept by loop C or D
#define HS_COL_SNP 16       // in the output
#define HS_COL_TIE 32       // its top-3 needed the reference's order of equal counts
#define HS_COL_C1GT5C2 64   // second count > 5 x third count (call_variants.cpp:526; the third count itself is not kept)
#define HS_COL_OPEN 128     // between k_columns_compact and k_column_top3_exact: the leading codes are not decided yet (equal counts, or K2 ran without its second pass)

struct ColumnsHeader {      // what the host reads between the phases (one small download)
    int64_t n_cols, n_entries;          // extracted columns / their entries
    int64_t n_flagged, n_flagged_entries;   // candidates (after k_candidates_scan + k_flag_block_sums/_offsets) or SNPs (after k_snp_flags + ...)
    int64_t n_tie, n_tie_big;           // columns whose order went through the emulator / through std::sort's non-stable part
    int64_t ok;                         // the column arrays' capacities hold n_cols / n_entries (k_columns_compact): 0 makes every later kernel of the pass a no-op
    int64_t pad;
};
static __device__ __forceinline__ int64_t header_cols(const ColumnsHeader* __restrict__ h) { return h->ok ? h->n_cols : 0; }

// the totals of the two tile scans into the header the host reads before it sizes the column arrays
__global__ void k_columns_totals(const int64_t* __restrict__ n_cols, const int64_t* __restrict__ n_entries, ColumnsHeader* __restrict__ header) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { header->n_cols = *n_cols; header->n_entries = *n_entries; }
}

// ------------------------------------------------------------------------------------------------
// K2's selection of the tiles [tile0, tile0 + n_tiles) packed into the column list, in position order. One workgroup per tile.
// tile_base / tile_ebase: exclusive scans of the tiles' column counts / entry counts.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_columns_compact(
    const int32_t* __restrict__ tile_cnt, const int64_t* __restrict__ tile_base, const int64_t* __restrict__ tile_ebase,
    const int64_t* __restrict__ scratch_gpos, const int32_t* __restrict__ scratch_depth, int64_t n_tiles,
    const int64_t* __restrict__ contig_off, int n_contigs, int64_t* __restrict__ col_gpos, hs_colrec_dev* __restrict__ col_rec,
    int64_t* __restrict__ col_off, int32_t* __restrict__ col_len, ColumnsHeader* __restrict__ header, int64_t cap_cols, int64_t cap_entries,
    const uint2* __restrict__ scratch_info /* K2's leading codes per slot, or NULL */) {
    // one wavefront per tile (a tile keeps a handful of its 256 positions since K2 drops what nobody reads), 64 slots at a time
    const int lane = lane_id();
    const int64_t t = (int64_t)blockIdx.x * 4 + wave_id();
    if (t >= n_tiles) return;
    if (t == n_tiles - 1 && lane == 0) {
        header->n_cols = tile_base[n_tiles]; header->n_entries = tile_ebase[n_tiles];
        header->ok = (tile_base[n_tiles] <= cap_cols && tile_ebase[n_tiles] <= cap_entries) ? 1 : 0;
        if (tile_base[n_tiles] <= cap_cols) col_off[tile_base[n_tiles]] = tile_ebase[n_tiles];
    }
    const int cnt = tile_cnt[t];
    if (cnt == 0) return;
    const int64_t k0 = tile_base[t], e0 = tile_ebase[t];
    int before = 0;
    for (int s0 = 0; s0 < cnt; s0 += 64) {
        const int slot = s0 + lane;
        const bool mine = slot < cnt;
        const int depth = mine ? scratch_depth[t * 256 + slot] : 0;
        const int incl = wave_scan_incl(depth);
        const int64_t k = k0 + slot;
        const bool live = mine && k < cap_cols;
        if (live) {
            const int64_t g = scratch_gpos[t * 256 + slot];
            int lo = 0, hi = n_contigs - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (contig_off[mid] <= g) lo = mid; else hi = mid - 1; }
            col_gpos[k] = g;
            col_off[k] = e0 + before + incl - depth;
            col_len[k] = depth;
            hs_colrec_dev r;
            r.pos = (int32_t)(g - contig_off[lo]); r.contig = lo; r.c0 = 0; r.c1 = 0; r.k0 = 0; r.k1 = 0; r.flags = HS_COL_OPEN; r.c2 = 63;
            if (scratch_info) {
                const uint2 inf = scratch_info[t * 256 + slot];
                if (inf.y & 0x80000000u) {
                    r.c0 = (uint16_t)(inf.x & 0xffffu); r.c1 = (uint16_t)(inf.x >> 16);
                    r.k0 = (uint8_t)(inf.y & 255u); r.k1 = (uint8_t)((inf.y >> 8) & 255u); r.flags = (uint8_t)((inf.y >> 16) & 255u); r.c2 = (uint8_t)((inf.y >> 24) & 63u);
                    if (r.flags & HS_COL_TIE) r.flags |= HS_COL_OPEN;      // (k_column_top3_exact orders it as the reference does)
                }
            }
            col_rec[k] = r;
        }
        before += __builtin_amdgcn_readlane(incl, 63);
    }
}

// ------------------------------------------------------------------------------------------------
// K3, tile-cooperative. One wavefront per tile that has selected positions. RC records at a time: lane j keeps plan entry j
// (first position in the tile, length, pileup address of tile position 0); the RC x 256 bytes the records lay over the tile
// go to LDS with one dword load per lane and record (rows of 260 bytes: a column read -- lane = record, fixed position -- then
// hits 64 different banks). For every selected position the covering records are a ballot; rank = prefix popcount, so the
// read indices of a column come out ascending (the plan lists a tile's records in ascending order) and a column's entries are
// written with one coalesced store of indices and one of codes per RC records.
// The pileup buffer is padded by 256 bytes on both sides: a row is loaded whole, also where the record covers part of the tile.
// ------------------------------------------------------------------------------------------------
#define HS_GT_RC 32
#define HS_GT_ROW 260
#define HS_GT_DIRECT 16      // tiles with at most this many kept positions fetch their bytes directly
__global__ __launch_bounds__(256) void k_gather_tiles(
    const uint8_t* __restrict__ pile, const int64_t* __restrict__ tile_off, const int4* __restrict__ tile_ent, const int32_t* __restrict__ tile_lrec,
    int64_t tile0, int64_t n_tiles, const int32_t* __restrict__ tile_cnt, const int64_t* __restrict__ tile_base,
    const int64_t* __restrict__ col_gpos, const int64_t* __restrict__ col_off, int32_t* __restrict__ col_idx, uint8_t* __restrict__ col_code,
    const ColumnsHeader* __restrict__ header) {
    __shared__ __attribute__((aligned(16))) uint8_t s_rows[4][HS_GT_RC * HS_GT_ROW];
    if (!header->ok) return;      // (the column arrays are too small for this range: the caller sizes up and runs the pass again)
    const int lane = lane_id();
    const int wv = wave_id();
    const int64_t tl = (int64_t)blockIdx.x * 4 + wv;      // tile index in the launch
    if (tl >= n_tiles) return;
    const int cnt = tile_cnt[tl];
    if (cnt == 0) return;
    uint8_t* __restrict__ rows = s_rows[wv];
    const int64_t kbase = tile_base[tl];
    const int64_t e0 = tile_off[tile0 + tl], e1 = tile_off[tile0 + tl + 1];
    if (cnt <= HS_GT_DIRECT) return;      // (k_gather_tiles_direct does those)
    for (int sc = 0; sc < cnt; sc += 64) {      // (a tile rarely selects more than 64 of its 256 positions)
        const int ns = (cnt - sc) < 64 ? (cnt - sc) : 64;
        int my_x = 0;
        int64_t my_w = 0;
        if (lane < ns) { my_x = (int)(col_gpos[kbase + sc + lane] & 255); my_w = col_off[kbase + sc + lane]; }
        for (int64_t rb = e0; rb < e1; rb += HS_GT_RC) {
            const int nrec = (e1 - rb) < HS_GT_RC ? (int)(e1 - rb) : HS_GT_RC;
            int4 en = make_int4(0, 0, 0, 0);
            int lrec = 0;
            if (lane < nrec) { en = tile_ent[rb + lane]; lrec = tile_lrec[rb + lane]; }
            __builtin_amdgcn_wave_barrier();
            // stage the rows: record j -> bytes of tile positions 4 * lane .. 4 * lane + 3
            // (all rows of the chunk requested before the first is written: with one load in flight per wavefront 86 % of the wave cycles
            // were spent waiting -- SQ_WAIT_ANY -- and the kernel took 1.25 ms instead of 0.73)
            for (int j0 = 0; j0 < nrec; j0 += 32) {
                uint32_t v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    const int j = (j0 + u) < nrec ? (j0 + u) : (nrec - 1);
                    const uint32_t plo = (uint32_t)__builtin_amdgcn_readlane(en.z, j), phi = (uint32_t)__builtin_amdgcn_readlane(en.w, j);
                    const uint8_t* __restrict__ base = pile + (int64_t)(((uint64_t)phi << 32) | plo);
                    v[u] = *reinterpret_cast<const u32_unaligned*>(base + 4 * lane);      // one (unaligned) dword per lane: the row
                }
#pragma unroll
                for (int u = 0; u < 32; ++u)
                    if (j0 + u < nrec) *reinterpret_cast<uint32_t*>(rows + (j0 + u) * HS_GT_ROW + 4 * lane) = v[u];
            }
            wave_lds_sync();
            for (int s = 0; s < ns; ++s) {
                const int x = __builtin_amdgcn_readlane(my_x, s);
                const bool cov = lane < nrec && (unsigned)(x - en.x) < (unsigned)en.y;
                const unsigned long long m = __ballot(cov);
                if (m == 0ull) continue;
                const uint32_t wlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_w & 0xffffffffll), s);
                const uint32_t whi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)my_w >> 32), s);
                const int64_t w = (int64_t)(((uint64_t)whi << 32) | wlo);
                if (cov) {
                    const int rank = __popcll(m & ((1ull << lane) - 1ull));
                    col_idx[w + rank] = lrec;
                    col_code[w + rank] = rows[lane * HS_GT_ROW + x];
                }
                if (lane == s) my_w += __popcll(m);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// K3 for the tiles that keep few positions (nearly all of them since K2 drops what nobody reads): no LDS, so a CU holds eight times the
// wavefronts of the staging form -- the kernel is a latency chain of plan entry -> byte loads -> stores per tile.
__global__ __launch_bounds__(256) void k_gather_tiles_direct(
    const uint8_t* __restrict__ pile, const int64_t* __restrict__ tile_off, const int4* __restrict__ tile_ent, const int32_t* __restrict__ tile_lrec,
    int64_t tile0, int64_t n_tiles, const int32_t* __restrict__ tile_cnt, const int64_t* __restrict__ tile_base,
    const int64_t* __restrict__ col_gpos, const int64_t* __restrict__ col_off, int32_t* __restrict__ col_idx, uint8_t* __restrict__ col_code,
    const ColumnsHeader* __restrict__ header) {
    if (!header->ok) return;
    const int lane = lane_id();
    const int64_t tl = (int64_t)blockIdx.x * 4 + wave_id();
    if (tl >= n_tiles) return;
    const int cnt = tile_cnt[tl];
    if (cnt == 0 || cnt > HS_GT_DIRECT) return;
    const int64_t kbase = tile_base[tl];
    const int64_t e0 = tile_off[tile0 + tl], e1 = tile_off[tile0 + tl + 1];
    // Round 4: K2 keeps a handful of positions per tile (5.5 on average), so the bytes a column needs are fetched directly -- lane =
    // record, one byte load per kept position, all of a chunk's loads in flight together -- instead of staging 32 x 256 bytes of rows
    // in LDS for them. Same ranks (ballot + prefix popcount over the records in plan order), same stores.
    int my_x = 0;
    int64_t my_w = 0;
    if (lane < cnt) { my_x = (int)(col_gpos[kbase + lane] & 255); my_w = col_off[kbase + lane]; }
    for (int64_t rb = e0; rb < e1; rb += 64) {
        const int nrec = (e1 - rb) < 64 ? (int)(e1 - rb) : 64;
        int4 en = make_int4(0, 0, 0, 0);
        int lrec = 0;
        if (lane < nrec) { en = tile_ent[rb + lane]; lrec = tile_lrec[rb + lane]; }
        const uint8_t* __restrict__ base = pile + (int64_t)(((uint64_t)(uint32_t)en.w << 32) | (uint32_t)en.z);
        for (int s0 = 0; s0 < cnt; s0 += 8) {
            uint8_t code[8]; bool cov[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int x = __builtin_amdgcn_readlane(my_x, (s0 + u) & 63);
                cov[u] = s0 + u < cnt && lane < nrec && (unsigned)(x - en.x) < (unsigned)en.y;
                code[u] = cov[u] ? base[x] : (uint8_t)0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (s0 + u >= cnt) break;      // wave-uniform
                const unsigned long long m = __ballot(cov[u]);
                if (m == 0ull) continue;
                const int s = s0 + u;
                const uint32_t wlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_w & 0xffffffffll), s);
                const uint32_t whi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)my_w >> 32), s);
                const int64_t w = (int64_t)(((uint64_t)whi << 32) | wlo);
                if (cov[u]) {
                    const int rank = __popcll(m & ((1ull << lane) - 1ull));
                    col_idx[w + rank] = lrec;
                    col_code[w + rank] = code[u];
                }
                if (lane == s) my_w += __popcll(m);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K3b, exact: top-3 of every extracted column as call_variants.cpp:477-507 forms it -- the counts of the column's codes go
// into a robin_hood map (keys in the order the reads bring them, then the zero-count fillers 0, 1, 2), the map is walked into
// a vector and std::sort orders it by count. Where the three largest counts differ from each other and from the fourth, no
// order of equal keys is involved: histogram in LDS, three wave arg-max rounds. Else one lane replays the reference: the codes
// in first-appearance order into hs::Rh8View (tables in LDS), its iteration order into hs::CountSort (std::sort's own
// sequence of moves). Writes counts and codes into the column record.
// ------------------------------------------------------------------------------------------------
#define HS_T3_CHUNK 16
__global__ __launch_bounds__(256) void k_column_top3_exact(const int64_t* __restrict__ col_off, const int32_t* __restrict__ col_len,
                                                           const uint8_t* __restrict__ col_code, const ColumnsHeader* __restrict__ header,
                                                           hs_colrec_dev* __restrict__ col_rec, unsigned long long* __restrict__ n_tie /* [4][2]: {ties, of them sorted beyond 16 keys}, four slots the host adds up */) {
    __shared__ int s_hist[4][128];
    __shared__ uint8_t s_info[4][512], s_key[4][512], s_tmp[4][512];
    __shared__ uint32_t s_sort[4][136];
    __shared__ int s_stack[4][120];
    __shared__ uint8_t s_first[4][128];
    __shared__ int s_fpos[4][128];
    // place of every byte key in the iteration order of the reference's hash map while it has not grown past 16 buckets (tests/harness/
    // rh8_static_order.cpp): up to 6 keys 8 buckets and the first multiplier, 7 to 12 keys 16 buckets and the second one; rank = home bucket << 5 |
    // 31 - low five hash bits. Keys of different rank iterate in rank order whatever order they were inserted in.
    __shared__ uint16_t s_rank8[256], s_rank16[256];
    __shared__ int s_bucket[4][16];
    const int lane = lane_id();
    const int wv = wave_id();
    const int64_t n_cols = header_cols(header);
    int* __restrict__ h = s_hist[wv];
    __shared__ int s_ties[2];
    if (threadIdx.x < 2) s_ties[threadIdx.x] = 0;
    {
        unsigned long long x = (unsigned long long)threadIdx.x;
        x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33;
        unsigned long long a = x * 0xc4ceb9fe1a85ec53ull; a ^= a >> 33;
        unsigned long long b = x * (0xc4ceb9fe1a85ec53ull + 0xc4ceb9fe1a85ec54ull); b ^= b >> 33;
        s_rank8[threadIdx.x] = (uint16_t)((((a >> 5) & 7ull) << 5) | (31ull - (a & 31ull)));
        s_rank16[threadIdx.x] = (uint16_t)((((b >> 5) & 15ull) << 5) | (31ull - (b & 31ull)));
    }
    __syncthreads();
    int my_ties = 0, my_big = 0;
    // a wavefront looks at 64 column records at a time and does the ones k_columns_compact left open (since K2 forms the leading codes itself:
    // the columns with equal counts, about one in 35), one after the other, all lanes on one column
    // (HS_T3_CHUNK records per wavefront and round: the kernel is as long as the wavefront with the most tied columns -- a tie takes 14 us of
    // one lane's dependent LDS accesses -- and ties come in runs; 64 records per wavefront: 0.49 ms for the C4 job, 16: see DESIGN 4.3)
    for (int64_t base = ((int64_t)blockIdx.x * 4 + wv) * HS_T3_CHUNK; base < n_cols; base += (int64_t)gridDim.x * 4 * HS_T3_CHUNK) {
      unsigned long long todo = __ballot(lane < HS_T3_CHUNK && base + lane < n_cols && (col_rec[base + lane < n_cols ? base + lane : 0].flags & HS_COL_OPEN) != 0);
      for (; todo; todo &= todo - 1ull) {
        const int64_t col = base + __builtin_ctzll(todo);
        h[lane] = 0; h[lane + 64] = 0;
        wave_lds_sync();
        const int64_t b = col_off[col];
        const int n = col_len[col];
        for (int j = lane; j < n; j += 64) {
            const int c = (int)col_code[b + j] - 33;
            if (c >= 0 && c < HS_NBINS) atomicAdd(&h[c], 1);
        }
        wave_lds_sync();
        int k_a = (h[lane] << 8) | (255 - lane), k_b = (h[lane + 64] << 8) | (255 - (lane + 64));
        int top[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mine = k_a > k_b ? k_a : k_b;
            const int best = wave_max_i32(mine);
            top[r] = best;
            if (k_a == best) k_a = -1;
            if (k_b == best) k_b = -1;
        }
        int c0 = top[0] >> 8, c1 = top[1] >> 8, c2 = top[2] >> 8;
        const int c3 = top[3] >> 8;
        int k0 = 33 + 255 - (top[0] & 255), k1 = 33 + 255 - (top[1] & 255);
        // which keys take places 0 and 1 is open when c0 == c1 or c1 == c2 (c2 == c3 only permutes places 2.. -- the third COUNT
        // is all the path uses); a second count of zero means the fillers take part
        const bool tie = c0 == c1 || c1 == c2 || c1 == 0;
        (void)c3;
        bool settled = false;
        if (tie && c0 < 8192) {
            // Equal counts: std::sort on up to 16 pairs is an insertion sort, i.e. stable -- the pairs keep the order in which the hash map
            // hands them over, and that order is the keys' rank (above) while the map holds at most 12 keys and no key has been pushed six
            // slots from its bucket. All lanes: key = count << 18 | 1023 - rank << 8 | code, the three largest by wave maxima; two of the
            // leading keys with the same count AND rank (the insertion order would decide), more keys, a far key: the emulator below.
            const bool on_a = h[lane] > 0, on_b = h[lane + 64] > 0;
            const int nd = __popcll(__ballot(on_a)) + __popcll(__ballot(on_b));
            const int mkeys = nd + 3;      // + the zero-count fillers 0, 1, 2 (:492-494)
            if (mkeys <= 12) {
                const bool wide = mkeys > 6;
                const uint16_t* __restrict__ rt = wide ? s_rank16 : s_rank8;
                const int ra = rt[lane + 33], rb = lane + 64 + 33 < 256 ? rt[lane + 64 + 33] : 0, rf = rt[lane & 3];
                bool far = false;
                if (wide) {      // a key six or more slots from its home bucket takes the map's overflow path (info byte): not the static order
                    if (lane < 16) s_bucket[wv][lane] = 0;
                    wave_lds_sync();
                    if (on_a) atomicAdd(&s_bucket[wv][ra >> 5], 1);
                    if (on_b) atomicAdd(&s_bucket[wv][rb >> 5], 1);
                    if (lane < 3) atomicAdd(&s_bucket[wv][rf >> 5], 1);
                    wave_lds_sync();
                    int carry = 0;
                    for (int bkt = 0; bkt < 16; ++bkt) { const int cb = s_bucket[wv][bkt]; if (cb > 0 && carry + cb - 1 >= 6) far = true; carry = carry + cb - 1 > 0 ? carry + cb - 1 : 0; }
                }
                if (!far) {
                    int ka = on_a ? ((h[lane] << 18) | ((1023 - ra) << 8) | (lane + 33)) : -1;
                    int kb = on_b ? ((h[lane + 64] << 18) | ((1023 - rb) << 8) | (lane + 64 + 33)) : -1;
                    int kf = lane < 3 ? (((1023 - rf) << 8) | lane) : -1;
                    int tv[3];
                    bool amb = false;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        int mine = ka > kb ? ka : kb;
                        mine = kf > mine ? kf : mine;
                        const int best = wave_max_i32(mine);
                        tv[r] = best;
                        const int same = __popcll(__ballot(ka >= 0 && (ka >> 8) == (best >> 8))) + __popcll(__ballot(kb >= 0 && (kb >> 8) == (best >> 8)))
                                         + __popcll(__ballot(kf >= 0 && (kf >> 8) == (best >> 8)));
                        amb = amb || same > 1;
                        if (ka == best) ka = -1;
                        if (kb == best) kb = -1;
                        if (kf == best) kf = -1;
                    }
                    if (!amb) {
                        k0 = tv[0] & 255; k1 = tv[1] & 255; c0 = tv[0] >> 18; c1 = tv[1] >> 18; c2 = tv[2] >> 18;
                        settled = true;
                        my_ties++;
                    }
                }
            }
        }
        if (tie && !settled) {
            // the column's codes in the order in which its reads bring them: the first position of every code by LDS atomic minima
            // over a coalesced pass (instead of one lane walking the column byte by byte out of global memory), the rank of a code =
            // how many codes appear before it
            int* __restrict__ fp = s_fpos[wv];
            fp[lane] = 0x7fffffff; fp[lane + 64] = 0x7fffffff;
            wave_lds_sync();
            for (int j = lane; j < n; j += 64) {
                const int bin = (int)col_code[b + j] - 33;
                if (bin >= 0 && bin < HS_NBINS) atomicMin(&fp[bin], j);
            }
            wave_lds_sync();
            {
                const int p_a = fp[lane], p_b = fp[lane + 64];      // (0x7fffffff: the code does not occur)
                const unsigned long long m_a = __ballot(p_a != 0x7fffffff), m_b = __ballot(p_b != 0x7fffffff);
                int r_a = 0, r_b = 0;
                for (unsigned long long m = m_a; m; m &= m - 1ull) { const int pq = __builtin_amdgcn_readlane(p_a, __builtin_ctzll(m)); r_a += pq < p_a; r_b += pq < p_b; }
                for (unsigned long long m = m_b; m; m &= m - 1ull) { const int pq = __builtin_amdgcn_readlane(p_b, __builtin_ctzll(m)); r_a += pq < p_a; r_b += pq < p_b; }
                uint8_t* first = s_first[wv];
                if (p_a != 0x7fffffff) { first[r_a] = (uint8_t)(lane + 33); h[lane] = -h[lane]; }      // negative = listed
                if (p_b != 0x7fffffff) { first[r_b] = (uint8_t)(lane + 64 + 33); h[lane + 64] = -h[lane + 64]; }
                if (lane == 0) s_sort[wv][131] = (uint32_t)(__popcll(m_a) + __popcll(m_b));
            }
            wave_lds_sync();
            if (lane == 0) {
                const int nd = (int)s_sort[wv][131];
                const uint8_t* first = s_first[wv];
                hs::Rh8View rh; rh.init(s_info[wv], s_key[wv], s_tmp[wv], 512);
                for (int i = 0; i < nd; ++i) rh.insert(first[i]);
                rh.insert(0); rh.insert(1); rh.insert(2);
                if (rh.overflow) __builtin_trap();      // (cap 512 holds every set of byte keys: tests/harness/rh8_selftest worstcase; never silently another order)
                const int m = rh.order(s_tmp[wv]);
                uint32_t* v = s_sort[wv];
                for (int i = 0; i < m; ++i) {
                    const int key = s_tmp[wv][i];
                    const int bin = key - 33;
                    const int cnt = (bin >= 0 && bin < HS_NBINS) ? -h[bin] : 0;
                    v[i] = ((uint32_t)cnt << 8) | (uint32_t)key;
                }
                hs::CountSort::sort_with_stack(v, m, s_stack[wv]);
                s_sort[wv][132] = v[0]; s_sort[wv][133] = v[1]; s_sort[wv][134] = v[2]; s_sort[wv][135] = (uint32_t)m;
            }
            wave_lds_sync();
            const uint32_t v0 = s_sort[wv][132], v1 = s_sort[wv][133], v2 = s_sort[wv][134];
            const int m = (int)s_sort[wv][135];
            k0 = (int)(v0 & 255u); k1 = (int)(v1 & 255u); c0 = (int)(v0 >> 8); c1 = (int)(v1 >> 8); c2 = (int)(v2 >> 8);
            my_ties++; if (m > 16) my_big++;
            wave_lds_sync();
        }
        if (lane == 0) {
            hs_colrec_dev r = col_rec[col];
            r.c0 = (uint16_t)(c0 > 65535 ? 65535 : c0); r.c1 = (uint16_t)(c1 > 65535 ? 65535 : c1);
            r.k0 = (uint8_t)k0; r.k1 = (uint8_t)k1; r.c2 = (uint8_t)(c2 < 63 ? c2 : 63);
            r.flags = tie ? HS_COL_TIE : 0;
            // c1 > c2 * 5 is all the path asks of the third count (call_variants.cpp:526): kept as a bit next to c2 == 0
            if (c1 > c2 * 5) r.flags |= HS_COL_C1GT5C2;
            col_rec[col] = r;
        }
      }
    }
    // one atomic per workgroup and counter, over four slots (39 k atomics on one address took 0.47 ms: 12 ns each, one after the other)
    if (lane == 0 && my_ties) { atomicAdd(&s_ties[0], my_ties); if (my_big) atomicAdd(&s_ties[1], my_big); }
    __syncthreads();
    if (threadIdx.x < 2 && s_ties[threadIdx.x]) atomicAdd(&n_tie[2 * (blockIdx.x & 3) + threadIdx.x], (unsigned long long)s_ties[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// generate_msa's return value per contig (call_variants.cpp:434) from the integer counters of K1 -- totalDistance is a float that
// is incremented by one (saturates at 2^24), totalLength a double starting at 1 (:67-68) -- and the read minimum that follows from
// it (:463-466). One wavefront per contig of the range; rec_stats = {q_end, errors, events, -} per record.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_contig_error(const int32_t* __restrict__ rec_stats, const int32_t* __restrict__ contig_rec_off, int c_first, int c_count,
                                                      float* __restrict__ mean_distance, int32_t* __restrict__ min_reads) {
    const int lane = lane_id();
    const int c = (int)blockIdx.x * 4 + wave_id();
    if (c >= c_count) return;
    const int r0 = contig_rec_off[c_first + c], r1 = contig_rec_off[c_first + c + 1];
    long long nerr = 0, nlen = 0;
    for (int r = r0 + lane; r < r1; r += 64) { nerr += rec_stats[(size_t)r * 4 + 1]; nlen += rec_stats[(size_t)r * 4 + 2]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { nerr += __shfl_xor(nerr, d, 64); nlen += __shfl_xor(nlen, d, 64); }
    if (lane == 0) {
        const float total_distance = nerr > 16777216 ? 16777216.0f : (float)nerr;
        const double total_length = 1.0 + (double)nlen;
        const float md = (float)((double)total_distance / total_length);
        mean_distance[c] = md;
        min_reads[c] = (double)md < 0.015 ? 3 : 5;
    }
}

// ------------------------------------------------------------------------------------------------
// Host <-> device transfers as kernels over pinned host memory (hipHostMalloc'ed blocks are mapped into the device's address
// space): a launch costs the host a few microseconds where hipMemcpyAsync costs 50 and an interrupt-driven completion signal,
// and -- what matters more -- the number of bytes may be a value that only exists on the device when the kernel is queued
// (a count a previous kernel of the stream produced), so that a chain of kernels ends in ONE host wait instead of "wait for the
// count, then queue the copy, then wait again". Up to 8 segments per launch; a segment's length is `bytes`, or, with `count`,
// min(*count, cap) * stride (+ extra) bytes, rounded up to 16 (every block of the pools is 256-byte aligned and padded).
// ------------------------------------------------------------------------------------------------
struct ShipSeg { const void* src; void* dst; long long bytes; const long long* count; long long stride, cap, extra; };
struct ShipList { ShipSeg seg[8]; int n; };
__global__ __launch_bounds__(256) void k_ship(const ShipList L) {
    for (int s = 0; s < L.n; ++s) {
        const ShipSeg& g = L.seg[s];
        long long bytes = g.bytes;
        if (g.count) { long long c = *g.count; if (c > g.cap) c = g.cap; if (c < 0) c = 0; bytes = c * g.stride + g.extra; }
        const long long n16 = (bytes + 15) >> 4;
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(g.src);
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(g.dst);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) dst[i] = src[i];
    }
}
// (fills: the same reasoning for hipMemsetAsync)
__global__ __launch_bounds__(256) void k_fill16(uint4* __restrict__ dst, long long n16, unsigned v) {
    const uint4 x = make_uint4(v, v, v, v);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) dst[i] = x;
}

// ------------------------------------------------------------------------------------------------
// The candidate columns as bit sets over the reads ranked by start position: what loop A of keep_only_robust_variants reads on
// the host (hs::CandBits in hs_host.h; hs::cv_build_cand_bits is the same on the host, for the tests). One wavefront per packed
// candidate column: lanes = entries. The distinct codes in the order their first entry brings them (LDS minima over the entry
// index, then one wave minimum per code), the words the reads' ranks span, the largest alignment end, and the block
// [any][slot bits][codes] -- built in LDS when it has at most HS_CB_LDS_WORDS words (nearly always: two words x five codes),
// else in place with global atomics. Blocks are bump-allocated from `out_words` (counter[0] = words used; counter[1] != 0: the
// capacity did not suffice or a column spans more than 65535 words -- the caller sizes up and runs the kernel again).
// ------------------------------------------------------------------------------------------------
struct alignas(16) CandBitsDev { int32_t wlo; uint16_t n_words, n_slots; int32_t idx_min, idx_max, reach, n_entries; long long word_off; };
static_assert(sizeof(CandBitsDev) == 32, "CandBitsDev must be 32 bytes");
#define HS_CB_LDS_WORDS 96
#define HS_CB_WAVES 4       // wavefronts per workgroup
#define HS_CB_PER_WAVE 4    // columns per wavefront, one after the other with their loads in flight together: 16 columns and ONE allocation atomic per
                            // workgroup (one per column: 60 k atomics on one address per launch, 0.8 ms). Round 4, first form: one column per wavefront, 16
                            // wavefronts per workgroup -- 0.47 ms, a latency chain per column at full occupancy, and 1024-thread workgroups find room late
                            // on a device that other contig groups' kernels share.
__global__ __launch_bounds__(64 * HS_CB_WAVES) void k_cand_bits(
    const hs_colrec_dev* __restrict__ cand_rec, const int64_t* __restrict__ cand_off, const int32_t* __restrict__ cand_idx, const uint8_t* __restrict__ cand_code,
    const ColumnsHeader* __restrict__ header, long long cap_cand, const int32_t* __restrict__ contig_rec_off, const int2* __restrict__ rank_end /* per record: {rank on its contig, alignment end} */,
    CandBitsDev* __restrict__ out_bits, unsigned long long* out_words, long long cap_words,
    unsigned long long* counter, long long cap_entries, const int32_t* __restrict__ cand_len /* non-NULL: cand_off[k] is column k's place in cand_idx / cand_code (the
    range's own column arrays) and cand_len[k] its length; NULL: the packed arrays, lengths from consecutive offsets */,
    const uint8_t* __restrict__ skip_contig /* non-NULL: [contig - c_first] != 0 -- loop A of that contig runs on the device (k_loop_a): its columns get an empty header and no block */, int c_first) {
    constexpr int PW = HS_CB_PER_WAVE;
    __shared__ int s_fp[HS_CB_WAVES][PW][256];
    __shared__ uint8_t s_slot[HS_CB_WAVES][PW][256];
    __shared__ uint8_t s_codes[HS_CB_WAVES][PW][128];
    __shared__ unsigned long long s_blk[HS_CB_WAVES][HS_CB_LDS_WORDS];
    __shared__ int s_size[HS_CB_WAVES * PW];
    __shared__ long long s_base;
    const int lane = lane_id();
    const int wv = wave_id();
    long long n_cand = header->n_flagged;
    if (n_cand > cap_cand || (!cand_len && header->n_flagged_entries > cap_entries)) {      // the packed block did not hold the candidates: nothing to read
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(counter + 1, 2ull);
        return;
    }
    const long long k0 = ((long long)blockIdx.x * HS_CB_WAVES + wv) * PW;
    // ---- phase 1: per column the codes in first-appearance order, the words its reads' ranks span, the largest alignment end -> its size ----
    bool live[PW]; int r0[PW], n[PW]; int64_t e0[PW];
#pragma unroll
    for (int c = 0; c < PW; ++c) {
        const long long k = k0 + c;
        live[c] = k < n_cand;      // (a wavefront without a column still meets the others at the barriers)
        const hs_colrec_dev rec = cand_rec[live[c] ? k : 0];
        if (live[c] && skip_contig && skip_contig[rec.contig - c_first]) {
            live[c] = false;
            if (lane == 0) { CandBitsDev h; h.wlo = 0; h.n_words = 0; h.n_slots = 0; h.idx_min = 0; h.idx_max = -1; h.reach = -1; h.n_entries = 0; h.word_off = 0; out_bits[k] = h; }
        }
        r0[c] = live[c] ? contig_rec_off[rec.contig] : 0;
        e0[c] = live[c] ? cand_off[k] : 0;
        n[c] = live[c] ? (cand_len ? cand_len[k] : (int)(cand_off[k + 1] - e0[c])) : 0;
    }
    int rc0[PW];      // the first 64 entries of every column stay in registers (rank << 8 | code): nearly every column is that shallow
    int2 re0[PW]; int cd0[PW];
#pragma unroll
    for (int c = 0; c < PW; ++c) {      // (the four columns' gathers in flight together)
        re0[c] = make_int2(0, -1); cd0[c] = 0;
        if (lane < n[c]) { re0[c] = rank_end[r0[c] + cand_idx[e0[c] + lane]]; cd0[c] = cand_code[e0[c] + lane]; }
    }
    // A column of up to 64 entries (nearly all) needs no table: its distinct codes in first-appearance order come from ballots -- the lowest
    // remaining lane's code is the next one -- and every lane keeps the slot of its code in a register. Deeper columns: the first position
    // of every code by LDS minima, then one wave minimum per code.
#pragma unroll
    for (int c = 0; c < PW; ++c) {
        if (n[c] <= 64) continue;      // (wave-uniform)
#pragma unroll
        for (int i = 0; i < 4; ++i) s_fp[wv][c][lane + 64 * i] = 0x7fffffff;
    }
    wave_lds_sync();
    int lo[PW], hi[PW], reach[PW], nslots[PW], W[PW], cells[PW], size[PW], my_slot[PW];
#pragma unroll
    for (int c = 0; c < PW; ++c) {
        int* __restrict__ fp = s_fp[wv][c];
        int l_lo = 0x7fffffff, l_hi = -1, l_reach = -1;
        rc0[c] = -1; my_slot[c] = -1; nslots[c] = 0;
        if (lane < n[c]) {
            const int w = re0[c].x >> 6;
            l_lo = w; l_hi = w; l_reach = re0[c].y;
            if (n[c] > 64) atomicMin(&fp[cd0[c]], lane);
            rc0[c] = (re0[c].x << 8) | cd0[c];
        }
        if (n[c] <= 64) {
            const bool valid = lane < n[c];
            unsigned long long rem = __ballot(valid);
            int ns = 0;
            while (rem) {
                const int code = __builtin_amdgcn_readlane(cd0[c], __builtin_ctzll(rem));
                const bool mine = valid && cd0[c] == code;
                if (mine) my_slot[c] = ns;
                if (lane == 0) s_codes[wv][c][ns] = (uint8_t)code;
                rem &= ~__ballot(mine);
                ns++;
            }
            nslots[c] = ns;
        }
        for (int j0 = 64; j0 < n[c]; j0 += 64) {      // (deep columns)
            const int j = j0 + lane;
            if (j < n[c]) {
                const int2 re = rank_end[r0[c] + cand_idx[e0[c] + j]];
                const int cd = cand_code[e0[c] + j];
                const int w = re.x >> 6;
                l_lo = w < l_lo ? w : l_lo; l_hi = w > l_hi ? w : l_hi;
                l_reach = re.y > l_reach ? re.y : l_reach;
                atomicMin(&fp[cd], j);
            }
        }
        lo[c] = -wave_max_i32(-l_lo); hi[c] = wave_max_i32(l_hi); reach[c] = wave_max_i32(l_reach);
    }
    wave_lds_sync();
#pragma unroll
    for (int c = 0; c < PW; ++c) {
        // the slots: the codes by first appearance
        int ns = nslots[c];
        if (n[c] > 64) {
            int f[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = s_fp[wv][c][lane + 64 * i];
            for (;;) {
                int m = f[0];
#pragma unroll
                for (int i = 1; i < 4; ++i) m = f[i] < m ? f[i] : m;
                const int best = -wave_max_i32(-m);
                if (best == 0x7fffffff || ns == 128) break;
#pragma unroll
                for (int i = 0; i < 4; ++i) if (f[i] == best) { f[i] = 0x7fffffff; s_slot[wv][c][lane + 64 * i] = (uint8_t)ns; s_codes[wv][c][ns] = (uint8_t)(lane + 64 * i); }
                ns++;
            }
        }
        W[c] = n[c] > 0 ? hi[c] - lo[c] + 1 : 0;
        if (n[c] == 0) { lo[c] = 0; ns = 0; }
        nslots[c] = ns;
        cells[c] = W[c] * (ns + 1);
        size[c] = live[c] ? cells[c] + (ns + 7) / 8 : 0;
        if (lane == 0) s_size[wv * PW + c] = size[c];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int w = 0; w < HS_CB_WAVES * PW; ++w) total += s_size[w];
        s_base = total ? (long long)atomicAdd(counter, (unsigned long long)total) : 0ll;
    }
    __syncthreads();      // (also: the slots and codes are written)
    // ---- phase 2: the blocks [any][slot bits][codes], built in LDS when they have at most HS_CB_LDS_WORDS words, else in place ----
    long long off_w = s_base;
    for (int w = 0; w < wv * PW; ++w) off_w += s_size[w];
    unsigned long long* __restrict__ blk = s_blk[wv];
#pragma unroll
    for (int c = 0; c < PW; ++c) {
        if (!live[c]) continue;      // (wave-uniform)
        const long long k = k0 + c;
        long long off = off_w;
        off_w += size[c];
        if (off + size[c] > cap_words || W[c] > 65535) { if (lane == 0) atomicOr(counter + 1, 1ull); off = -1; }
        const uint8_t* __restrict__ slot_of = s_slot[wv][c];
        const uint8_t* __restrict__ codes = s_codes[wv][c];
        if (off >= 0 && n[c] > 0 && n[c] <= 64 && W[c] <= 4) {
            // every entry is another read, so (word, bit) is another cell for each: a byte per cell holding the entry's slot + 1, written without a
            // conflict, and the rows of the block are ballots over those bytes -- no atomic, no word that 30 lanes OR into at once
            unsigned long long* __restrict__ out = out_words + off;
            uint8_t* __restrict__ cellb = reinterpret_cast<uint8_t*>(blk);
            reinterpret_cast<uint32_t*>(cellb)[lane] = 0u;      // 256 bytes: four words of 64 cells
            wave_lds_sync();
            if (lane < n[c]) { const int rk = rc0[c] >> 8; cellb[(((rk >> 6) - lo[c]) << 6) + (rk & 63)] = (uint8_t)(my_slot[c] + 1); }
            wave_lds_sync();
            unsigned long long mine_w = 0ull;      // lane x keeps word x of the block
            for (int w = 0; w < W[c]; ++w) {
                const int v = cellb[(w << 6) + lane];
                const unsigned long long any = __ballot(v != 0);
                if (lane == w) mine_w = any;
                for (int q = 0; q < nslots[c]; ++q) {
                    const unsigned long long row = __ballot(v == q + 1);
                    if (lane == (q + 1) * W[c] + w) mine_w = row;
                }
            }
            if (cells[c] <= 64) { if (lane < cells[c]) out[lane] = mine_w; }
            else {      // (more than 64 words: 16 and more codes over four words -- the rows once more, one word at a time)
                for (int w = 0; w < W[c]; ++w) {
                    const int v = cellb[(w << 6) + lane];
                    const unsigned long long any = __ballot(v != 0);
                    if (lane == 0) out[w] = any;
                    for (int q = 0; q < nslots[c]; ++q) { const unsigned long long row = __ballot(v == q + 1); if (lane == 0) out[(long long)(q + 1) * W[c] + w] = row; }
                }
            }
            if (lane < (nslots[c] + 7) / 8) {
                unsigned long long cw = 0;
                for (int q = 0; q < 8; ++q) if (8 * lane + q < nslots[c]) cw |= (unsigned long long)codes[8 * lane + q] << (8 * q);
                out[cells[c] + lane] = cw;
            }
            wave_lds_sync();      // (blk is the next column's)
        } else if (off >= 0 && n[c] > 0) {
            unsigned long long* __restrict__ out = out_words + off;
            const bool in_lds = cells[c] <= HS_CB_LDS_WORDS;
            if (in_lds) { for (int x = lane; x < cells[c]; x += 64) blk[x] = 0ull; }
            else { for (int x = lane; x < cells[c]; x += 64) out[x] = 0ull; __threadfence(); }
            wave_lds_sync();
            for (int j0 = 0; j0 < n[c]; j0 += 64) {
                const int j = j0 + lane;
                if (j < n[c]) {
                    int rk, cd;
                    if (j0 == 0) { rk = rc0[c] >> 8; cd = rc0[c] & 255; }
                    else { rk = rank_end[r0[c] + cand_idx[e0[c] + j]].x; cd = cand_code[e0[c] + j]; }
                    const int w = (rk >> 6) - lo[c];
                    const unsigned long long bit = 1ull << (rk & 63);
                    const int sl = n[c] <= 64 ? my_slot[c] : (int)slot_of[cd];
                    if (in_lds) { atomicOr(&blk[w], bit); atomicOr(&blk[(sl + 1) * W[c] + w], bit); }
                    else { atomicOr(&out[w], bit); atomicOr(&out[(long long)(sl + 1) * W[c] + w], bit); }
                }
            }
            wave_lds_sync();
            if (in_lds) for (int x = lane; x < cells[c]; x += 64) out[x] = blk[x];
            if (lane < (nslots[c] + 7) / 8) {
                unsigned long long cw = 0;
                for (int q = 0; q < 8; ++q) if (8 * lane + q < nslots[c]) cw |= (unsigned long long)codes[8 * lane + q] << (8 * q);
                out[cells[c] + lane] = cw;
            }
            wave_lds_sync();      // (blk is the next column's)
        }
        if (lane == 0) {
            CandBitsDev h;
            h.wlo = lo[c]; h.n_words = (uint16_t)(W[c] > 65535 ? 65535 : W[c]); h.n_slots = (uint16_t)nslots[c];
            h.idx_min = n[c] > 0 ? cand_idx[e0[c]] : 0; h.idx_max = n[c] > 0 ? cand_idx[e0[c] + n[c] - 1] : -1; h.reach = reach[c]; h.n_entries = n[c]; h.word_off = off;
            out_bits[k] = h;
        }
    }
}

static __device__ __forceinline__ bool central_base_test_cols(int k0, int k1) {
    // call_variants.cpp:527-528 and :751-752 (same predicate on raw code bytes)
    return k0 % 5 != k1 % 5 && ((k1 - '!') % 5 != 4 || (k1 / 5 % 5 != k0 % 5 && k1 / 25 % 5 != k0 % 5));
}

// ------------------------------------------------------------------------------------------------
// V1: the candidate and the "automatic" columns of every contig of the range (call_variants.cpp:525-536). One thread per
// column. The reference walks a contig's positions and takes a column that passes the predicate when it lies more than five
// positions after the last one it took (posoflastvariant starts at -5, so position 0 is never taken). A passing column with no
// passing column in the five positions before it is therefore always taken, whatever happened earlier: the dependency only
// runs along a CHAIN of passing columns each within five positions of the one before. A thread whose column passes walks back
// to the head of its chain (nearly always itself) and replays the greedy rule from there. Also splits the record into the
// arrays K4 reads, finds every contig's first column (threads 0..c_count of the launch) and counts the candidates per contig
// (contig_n_cand zeroed by the caller).
// The record is rewritten in place while neighbours may read it: everything the predicate reads (counts, codes, the
// HS_COL_C1GT5C2 bit, position, contig) is written back unchanged.
// min_reads[c] = 3 or 5 (:463-466, from the contig's mean distance); thr = automatic_snp_threshold.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_candidates_scan(
    const int64_t* __restrict__ col_gpos, const ColumnsHeader* __restrict__ header, const int64_t* __restrict__ contig_off, int c_first, int c_count,
    const int32_t* __restrict__ min_reads /* [c_count] */, float thr, hs_colrec_dev* col_rec,
    int32_t* __restrict__ col_contig_local, uint8_t* __restrict__ col_k0, uint8_t* __restrict__ col_k1, int32_t* __restrict__ col_c1, uint8_t* __restrict__ col_is_cand,
    int64_t* __restrict__ contig_col_off /* [c_count + 1] */, int32_t* contig_n_cand /* [c_count], zeroed */) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n_cols = header_cols(header);
    if (k < c_count) {
        // the contig's columns start at the first one at or after its global offset in the sorted list
        auto lower = [&](int64_t g) { int64_t lo = 0, hi = n_cols; while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (col_gpos[mid] < g) lo = mid + 1; else hi = mid; } return lo; };
        contig_col_off[k] = lower(contig_off[c_first + k]);
        if (k == c_count - 1) contig_col_off[c_count] = lower(contig_off[c_first + c_count]);
    }
    bool cand = false;
    int ci = -1;
    if (k < n_cols) {
        const int4* recs = reinterpret_cast<const int4*>(col_rec);
        auto unpack = [](const int4 v) { hs_colrec_dev r; __builtin_memcpy(&r, &v, 16); return r; };
        hs_colrec_dev r = unpack(recs[k]);
        ci = r.contig - c_first;
        const int mr = min_reads[ci];
        auto passes = [&](const hs_colrec_dev& q) {
            return q.pos > 0 && (int)q.c1 > mr && (q.flags & HS_COL_C1GT5C2) && central_base_test_cols(q.k0, q.k1);
        };
        if (passes(r)) {
            // back to the head of the chain
            int64_t h = k;
            int pos_h = r.pos;
            for (;;) {
                int64_t q = h - 1;
                bool linked = false;
                while (q >= 0) {
                    const hs_colrec_dev rq = unpack(recs[q]);
                    if (rq.contig != r.contig || pos_h - rq.pos > 5) break;
                    if (passes(rq)) { h = q; pos_h = rq.pos; linked = true; break; }
                    --q;
                }
                if (!linked) break;
            }
            if (h == k) cand = true;
            else {
                int last = pos_h;      // the head is taken
                for (int64_t q = h + 1; q <= k; ++q) {
                    const hs_colrec_dev rq = unpack(recs[q]);
                    if (passes(rq) && rq.pos - last > 5) { last = rq.pos; cand = q == k; }
                }
            }
        }
        uint8_t f = r.flags & (HS_COL_TIE | HS_COL_C1GT5C2);
        if (cand) { f |= HS_COL_CAND; if ((float)r.c1 > thr * (float)r.c0) f |= HS_COL_AUTO; }
        if ((int)r.c1 >= 5 && central_base_test_cols(r.k0, r.k1)) f |= HS_COL_LOOPD;
        r.flags = f;
        int4 w; __builtin_memcpy(&w, &r, 16);
        reinterpret_cast<int4*>(col_rec)[k] = w;
        col_contig_local[k] = ci; col_k0[k] = r.k0; col_k1[k] = r.k1; col_c1[k] = (int32_t)r.c1 | ((63 - (int32_t)r.c2) << 16); col_is_cand[k] = cand ? 1 : 0;      // (bits 16-21 of the count's word: how far the third count lies below 63; 0 = not known)
    }
    // candidates per contig: one atomic per contig and wavefront (a wavefront's columns nearly always lie in one contig)
    unsigned long long m = __ballot(cand);
    while (m) {
        const int l = __builtin_ctzll(m);
        const int c_l = __builtin_amdgcn_readlane(ci, l);
        const unsigned long long same = __ballot(cand && ci == c_l);
        if ((int)lane_id() == l) atomicAdd(&contig_n_cand[c_l], __popcll(same));
        m &= ~same;
    }
}

// ------------------------------------------------------------------------------------------------
// Columns whose record carries `flag` packed back to back: exclusive prefix of (1, length) over the column list in two kernels
// (per 1024-column block sums, then one workgroup orders the blocks), then one wavefront per flagged column copies its entries.
// ------------------------------------------------------------------------------------------------
#define HS_FP_BLOCK 1024
__global__ __launch_bounds__(256) void k_flag_block_sums(const hs_colrec_dev* __restrict__ col_rec, const int32_t* __restrict__ col_len, const ColumnsHeader* __restrict__ header,
                                                         int flag, long long* __restrict__ blk_cnt, long long* __restrict__ blk_ent) {
    __shared__ long long s_c[4], s_e[4];
    const int64_t n_cols = header_cols(header);
    const int64_t base = (int64_t)blockIdx.x * HS_FP_BLOCK;
    if (base >= n_cols) { if (threadIdx.x == 0) { blk_cnt[blockIdx.x] = 0; blk_ent[blockIdx.x] = 0; } return; }
    int c = 0; long long e = 0;
    for (int i = (int)threadIdx.x; i < HS_FP_BLOCK; i += 256) {
        const int64_t k = base + i;
        if (k < n_cols && (col_rec[k].flags & flag)) { c++; e += col_len[k]; }
    }
    const int lane = lane_id(), wv = (int)(threadIdx.x >> 6);
    const int cs = wave_sum_i32(c);
    long long es = e;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) es += __shfl_xor(es, d, 64);
    if (lane == 0) { s_c[wv] = cs; s_e[wv] = es; }
    __syncthreads();
    if (threadIdx.x == 0) { blk_cnt[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3]; blk_ent[blockIdx.x] = s_e[0] + s_e[1] + s_e[2] + s_e[3]; }
}
__global__ __launch_bounds__(1024) void k_flag_block_offsets(long long* __restrict__ blk_cnt, long long* __restrict__ blk_ent, int n_blocks, ColumnsHeader* __restrict__ header) {
    // serial over chunks of 1024 blocks, parallel inside (a range has a few hundred blocks)
    __shared__ long long s_c[1024], s_e[1024];
    __shared__ long long carry_c, carry_e;
    if (threadIdx.x == 0) { carry_c = 0; carry_e = 0; }
    __syncthreads();
    for (int b0 = 0; b0 < n_blocks; b0 += 1024) {
        const int i = b0 + (int)threadIdx.x;
        const long long c = i < n_blocks ? blk_cnt[i] : 0, e = i < n_blocks ? blk_ent[i] : 0;
        s_c[threadIdx.x] = c; s_e[threadIdx.x] = e;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            long long ac = 0, ae = 0;
            if ((int)threadIdx.x >= d) { ac = s_c[threadIdx.x - d]; ae = s_e[threadIdx.x - d]; }
            __syncthreads();
            s_c[threadIdx.x] += ac; s_e[threadIdx.x] += ae;
            __syncthreads();
        }
        if (i < n_blocks) { blk_cnt[i] = carry_c + s_c[threadIdx.x] - c; blk_ent[i] = carry_e + s_e[threadIdx.x] - e; }
        __syncthreads();
        if (threadIdx.x == 1023) { carry_c += s_c[1023]; carry_e += s_e[1023]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { header->n_flagged = carry_c; header->n_flagged_entries = carry_e; }
}
// out_rec[f] = record of the f-th flagged column, out_col[f] = its index in the column list, out_off[f] = first entry in the
// packed arrays (out_off[n_flagged] = total); entries copied by the wavefront that owns the column
__global__ __launch_bounds__(1024) void k_pack_flagged(
    const hs_colrec_dev* __restrict__ col_rec, const int64_t* __restrict__ col_off, const int32_t* __restrict__ col_len, const int32_t* __restrict__ col_idx,
    const uint8_t* __restrict__ col_code, const ColumnsHeader* __restrict__ header, int flag, const long long* __restrict__ blk_cnt,
    const long long* __restrict__ blk_ent, hs_colrec_dev* __restrict__ out_rec, int32_t* __restrict__ out_col, int64_t* __restrict__ out_off,
    int32_t* __restrict__ out_idx, uint8_t* __restrict__ out_code, int64_t cap_flagged, int64_t cap_entries,
    int32_t* __restrict__ out_len /* non-NULL: the LIGHT form -- no entry is copied, out_off[f] is the column's place in col_idx / col_code and out_len[f] its length */) {
    // one workgroup per block of HS_FP_BLOCK columns, one wavefront per 64 of them (sixteen short chains instead of four long ones: at
    // eight contig groups a launch is a few hundred wavefronts and lasts as long as one of them)
    const int64_t n_cols = header_cols(header);
    const int64_t base = (int64_t)blockIdx.x * HS_FP_BLOCK;
    if (base >= n_cols) return;
    __shared__ long long s_wc[16], s_we[16];
    __shared__ int s_incl[16][64];
    __shared__ int64_t s_src[16][64];
    const int lane = lane_id(), wv = wave_id();
    const int64_t k = base + wv * 64 + lane;
    const bool on = k < n_cols && (col_rec[k].flags & flag);
    const int len = on ? col_len[k] : 0;
    const int64_t my_src = on ? col_off[k] : 0;      // (every lane its own column's offset, one coalesced load)
    const unsigned long long m = __ballot(on);
    const int incl = wave_scan_incl(len);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (lane == 0) { s_wc[wv] = __popcll(m); s_we[wv] = total; }
    s_incl[wv][lane] = incl; s_src[wv][lane] = my_src;
    __syncthreads();
    long long f = blk_cnt[blockIdx.x], o = blk_ent[blockIdx.x];
    for (int w = 0; w < wv; ++w) { f += s_wc[w]; o += s_we[w]; }
    const long long my_f = f + __popcll(m & ((1ull << lane) - 1ull));
    const long long my_o = o + incl - len;
    if (out_len) {      // (the candidates: k_cand_bits reads their entries where they lie)
        if (on && my_f < cap_flagged) { out_rec[my_f] = col_rec[k]; out_col[my_f] = (int32_t)k; out_off[my_f] = my_src; out_len[my_f] = len; }
    } else {
        if (on && my_f < cap_flagged) { out_rec[my_f] = col_rec[k]; out_col[my_f] = (int32_t)k; out_off[my_f] = my_o; }
        // the entries of the wavefront's flagged columns lie one behind the other in the output: lane t, t + 64, ... of that range each
        // finds its column by bisection of the lengths' prefix sums (LDS) and copies one entry, four in flight per lane
        if (o + total <= cap_entries) {
            auto source = [&](int t) -> int64_t {
                int c = 0;      // the first column whose inclusive prefix exceeds t
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) if (s_incl[wv][c + step - 1] <= t) c += step;
                return s_src[wv][c] + (t - (c ? s_incl[wv][c - 1] : 0));
            };
            int t = lane;
            for (; t + 192 < total; t += 256) {
                const int64_t s0 = source(t), s1 = source(t + 64), s2 = source(t + 128), s3 = source(t + 192);
                const int32_t i0 = col_idx[s0], i1 = col_idx[s1], i2 = col_idx[s2], i3 = col_idx[s3];
                const uint8_t c0 = col_code[s0], c1 = col_code[s1], c2 = col_code[s2], c3 = col_code[s3];
                out_idx[o + t] = i0; out_idx[o + t + 64] = i1; out_idx[o + t + 128] = i2; out_idx[o + t + 192] = i3;
                out_code[o + t] = c0; out_code[o + t + 64] = c1; out_code[o + t + 128] = c2; out_code[o + t + 192] = c3;
            }
            for (; t < total; t += 64) { const int64_t s0 = source(t); out_idx[o + t] = col_idx[s0]; out_code[o + t] = col_code[s0]; }
        } else {      // (the packed block does not hold them all: column by column, each only if it fits whole)
            unsigned long long rem = m;
            while (rem) {
                const int l = __builtin_ctzll(rem); rem &= rem - 1ull;
                const int n = __builtin_amdgcn_readlane(len, l);
                const uint32_t olo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_o & 0xffffffffll), l), ohi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)my_o >> 32), l);
                const int64_t dst = (int64_t)(((uint64_t)ohi << 32) | olo);
                const uint32_t slo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_src & 0xffffffffll), l), shi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)my_src >> 32), l);
                const int64_t src = (int64_t)(((uint64_t)shi << 32) | slo);
                if (dst + n <= cap_entries)
                    for (int j = lane; j < n; j += 64) { out_idx[dst + j] = col_idx[src + j]; out_code[dst + j] = col_code[src + j]; }
            }
        }
    }
    const bool last_block = base + HS_FP_BLOCK >= n_cols;
    if (last_block && wv == 15 && lane == 0 && f + (long long)__popcll(m) <= cap_flagged) out_off[f + __popcll(m)] = o + total;
}

// ------------------------------------------------------------------------------------------------
// The merge of the automatic and the filtered SNPs of a contig (call_variants.cpp:1335-1352) walks both lists in position
// order and stops when either ends: what it emits is every column of either list up to min(last automatic, last filtered).
// Two passes over the column list, lanes = columns: the two maxima per contig (position + 1; 0 = the list is empty; `bounds`
// = [2][C] zeroed by the caller, as is contig_n_snp), then the SNP flag and the count per contig. keep[k]: verdict of loops
// C / D (K4). A wavefront's 64 columns nearly always belong to one contig: one atomic per wavefront then.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_snp_bounds(const hs_colrec_dev* __restrict__ col_rec, const int32_t* __restrict__ col_contig_local,
                                                    const uint8_t* __restrict__ keep, int64_t n_cols, int c_count, int32_t* __restrict__ bounds) {
    const int lane = lane_id();
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int ci = -1, pa = 0, pf = 0;
    if (k < n_cols) {
        const hs_colrec_dev r = col_rec[k];
        ci = col_contig_local[k];
        if (r.flags & HS_COL_AUTO) pa = r.pos + 1;
        if (keep[k] == 1) pf = r.pos + 1;
    }
    const int c_first = __builtin_amdgcn_readfirstlane(ci);
    if (__ballot(ci != c_first && ci >= 0) == 0ull && c_first >= 0) {
        const int ma = wave_max_i32(pa), mf = wave_max_i32(pf);
        if (lane == 0) { if (ma) atomicMax(&bounds[c_first], ma); if (mf) atomicMax(&bounds[c_count + c_first], mf); }
    } else if (ci >= 0) {
        if (pa) atomicMax(&bounds[ci], pa);
        if (pf) atomicMax(&bounds[c_count + ci], pf);
    }
}
__global__ __launch_bounds__(256) void k_snp_flags(hs_colrec_dev* __restrict__ col_rec, const int32_t* __restrict__ col_contig_local,
                                                   const uint8_t* __restrict__ keep, int64_t n_cols, int c_count, const int32_t* __restrict__ bounds,
                                                   int32_t* __restrict__ contig_n_snp) {
    const int lane = lane_id();
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int ci = -1;
    bool snp = false;
    if (k < n_cols) {
        hs_colrec_dev r = col_rec[k];
        ci = col_contig_local[k];
        const int ma = bounds[ci], mf = bounds[c_count + ci];
        const int bound = (ma == 0 || mf == 0) ? -1 : (ma < mf ? ma : mf) - 1;
        const bool kept = keep[k] == 1;
        snp = ((r.flags & HS_COL_AUTO) || kept) && r.pos <= bound;
        uint8_t f = r.flags & ~(HS_COL_KEEP | HS_COL_SNP);
        if (kept) f |= HS_COL_KEEP;
        if (snp) f |= HS_COL_SNP;
        r.flags = f;
        col_rec[k] = r;
    }
    const int c_first = __builtin_amdgcn_readfirstlane(ci);
    if (__ballot(ci != c_first && ci >= 0) == 0ull && c_first >= 0) {
        const int n = __popcll(__ballot(snp));
        if (lane == 0 && n) atomicAdd(&contig_n_snp[c_first], n);
    } else if (snp) atomicAdd(&contig_n_snp[ci], 1);
}

// ------------------------------------------------------------------------------------------------
// The reads of a clustering window = the reads present at its first AND its last SNP (separate_reads.cpp:1590-1622: the mask
// is set from the first column, then cleared up to the last read of the last column wherever that column lacks the read --
// reads beyond its last read keep their bit). One wavefront per window; both columns are ascending read lists of the SNP CSR,
// lanes take the reads of the first column and bisect the second. Window w writes its reads at slot_off[w] (room for the
// whole first column) and their number to win_m[w].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_window_masks(const int64_t* __restrict__ col_off, const int32_t* __restrict__ col_idx, const int64_t* __restrict__ win_col_a,
                                                      const int64_t* __restrict__ win_col_b, const int64_t* __restrict__ slot_off, int n_windows,
                                                      int32_t* __restrict__ ids, int32_t* __restrict__ win_m) {
    const int lane = lane_id();
    const int w = (int)blockIdx.x * 4 + wave_id();
    if (w >= n_windows) return;
    const int64_t a0 = col_off[win_col_a[w]], a1 = col_off[win_col_a[w] + 1];
    const int64_t b0 = col_off[win_col_b[w]], b1 = col_off[win_col_b[w] + 1];
    const int b_last = b1 > b0 ? col_idx[b1 - 1] : -1;      // an empty last column clears nothing
    int32_t* __restrict__ out = ids + slot_off[w];
    int cnt = 0;
    for (int64_t base = a0; base < a1; base += 64) {
        const int64_t e = base + lane;
        bool keep = false;
        int x = 0;
        if (e < a1) {
            x = col_idx[e];
            if (x > b_last) keep = true;
            else {
                int64_t lo = b0, hi = b1;
                while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (col_idx[mid] < x) lo = mid + 1; else hi = mid; }
                keep = lo < b1 && col_idx[lo] == x;
            }
        }
        const unsigned long long m = __ballot(keep);
        if (keep) out[cnt + __popcll(m & ((1ull << lane) - 1ull))] = x;
        cnt += __popcll(m);
    }
    if (lane == 0) win_m[w] = cnt;
}

}  // namespace hsdev
