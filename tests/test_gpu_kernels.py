"""GPU parity, kernel by kernel, through the C ABI (hairsplitter_amd.api -> libhairsplitter_hip.so):
every HIP kernel against the CPU oracle on the same seeded inputs. Integer / byte work: bit-exact."""
import json
import os

import numpy as np
import pytest

import golden_util as gu
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _contigs(kind):
    from hairsplitter_amd import synth
    if kind == "dip":
        return [synth.make_contig(11, 0, 30_000, 2, 0.01, 40, "ont")]
    if kind == "multi":
        return [synth.make_contig(12, 0, 9_000, 3, 0.01, 35, "ont"), synth.make_contig(12, 1, 700, 1, 0.0, 20, "ont", read_len_override=(100, 600)),
                synth.make_contig(12, 2, 5_000, 1, 0.0, 0, "ont"), synth.make_contig(12, 3, 20_011, 4, 0.012, 50, "ont", clip_prob=0.4)]
    if kind == "edge":
        return [synth.make_contig(14, 0, 15_000, 2, 0.01, 40, "ont", eqx=True, overhang_prob=0.2, inert_ops_prob=0.3, clip_prob=0.3,
                                  read_len_override=(300, 5000)),
                synth.make_contig(14, 1, 300, 1, 0.0, 30, "ont", read_len_override=(50, 300), overhang_prob=0.5)]
    if kind == "hifi":
        return [synth.make_contig(13, 0, 40_000, 2, 0.003, 25, "hifi")]
    raise ValueError(kind)


@pytest.fixture(scope="module", params=["dip", "multi", "hifi", "edge"])
def batch(request, built):
    from hairsplitter_amd import api
    api.require_gpu()
    flat = api.FlatBatch(_contigs(request.param))
    t = api.device_tensors(flat)
    return flat, t


def test_pileup_bytes_and_counters(batch):
    """K1 == generate_msa (call_variants.cpp:50-437): every pileup byte and the per-record integer counters."""
    from hairsplitter_amd import api
    flat, t = batch
    pile, stats = api.pileup(t, flat)
    o_pile, o_stats, _ = ol.pileup(flat)
    assert np.array_equal(pile.cpu().numpy(), o_pile)
    assert np.array_equal(stats.cpu().numpy()[:, :3], o_stats[:, :3])
    # task granularity must not change a byte (tasks of 64 / 1000 / one huge range per record)
    for ev in (64, 1000, 1 << 30):
        p2, s2 = api.pileup(t, flat, ev_per_task=ev)
        assert np.array_equal(p2.cpu().numpy(), o_pile), ev
        assert np.array_equal(s2.cpu().numpy()[:, :3], o_stats[:, :3]), ev


def test_pileup_records_with_a_clip_between_aligned_bases(built):
    """Records whose CIGAR holds an S / H run BETWEEN aligned bases (the reference just steps over the read bases,
    call_variants.cpp:269-273) are not one run of M/I/D events: K0 flags them and the per-event form handles them, next to
    the packed form for all other records. Also a clip longer than a task, and clips at several places of one record."""
    from hairsplitter_amd import api, synth
    rng = np.random.default_rng(11)
    c = synth.make_contig(41, 0, 20_000, 2, 0.01, 30, "ont", clip_prob=0.3)
    n_mod = 0
    for ai, a in enumerate(c.alns):
        if ai % 3 != 0 or len(a.cigar) < 12:
            continue
        for rep in range(1 + (ai % 2)):
            k = int(rng.integers(2, len(a.cigar) - 2))
            ops, lens = a.cigar & 0xF, (a.cigar >> 4).astype(np.int64)
            t = int(lens[:k][np.isin(ops[:k], (synth.OP_M, synth.OP_I, synth.OP_S, synth.OP_H, synth.OP_EQ, synth.OP_X))].sum())
            n = int(rng.integers(1, 40)) if ai % 9 else 5000
            junk = rng.integers(0, 4, size=n).astype(np.uint8)
            read = c.reads[a.read]
            at = t if a.strand else len(read) - t            # reads are stored as sequenced
            c.reads[a.read] = np.concatenate((read[:at], junk, read[at:]))
            tok = np.array([(n << 4) | (synth.OP_S if rep == 0 else synth.OP_H)], dtype=np.uint32)
            a.cigar = np.concatenate((a.cigar[:k], tok, a.cigar[k:]))
        n_mod += 1
    assert n_mod > 20
    flat = api.FlatBatch([c, synth.make_contig(41, 1, 8_000, 2, 0.01, 25, "ont")])
    t = api.device_tensors(flat)
    o_pile, o_stats, _ = ol.pileup(flat)
    for ev in (4096, 64, 1 << 30):
        pile, stats = api.pileup(t, flat, ev_per_task=ev)
        assert np.array_equal(pile.cpu().numpy(), o_pile), ev
        assert np.array_equal(stats.cpu().numpy()[:, :3], o_stats[:, :3]), ev


def test_pileup_tiny_reads_and_contigs(built):
    """Reads and contigs shorter than the four bytes a lane of the packed form loads at once (the loads are moved back into
    the sequence, never past its first byte), single-event records, records that start at the last base of the contig"""
    from hairsplitter_amd import api, synth
    rng = np.random.default_rng(3)
    M, I, D, S = synth.OP_M, synth.OP_I, synth.OP_D, synth.OP_S

    def cig(*ops):
        return np.array([(n << 4) | o for n, o in ops], dtype=np.uint32)

    def contig(name, L, recs):
        seq = rng.integers(0, 4, size=L).astype(np.uint8)
        reads, names, alns = [], [], []
        for k, (pos, strand, ops) in enumerate(recs):
            c = cig(*ops)
            rl = int(sum(n for n, o in ops if o in (M, I, S)))
            reads.append(rng.integers(0, 4, size=max(rl, 1)).astype(np.uint8)[:rl] if rl else np.zeros(0, np.uint8))
            names.append(f"{name}_r{k}")
            alns.append(synth.Alignment(k, pos, strand, c, 0))
        return synth.ContigData(name, seq, reads, names, alns, np.zeros(len(recs), np.int32))

    cs = [contig("t3", 3, [(0, True, [(3, M)]), (1, False, [(2, M)]), (2, True, [(1, M)]), (0, True, [(1, M), (1, I), (1, M)]),
                           (0, False, [(1, M), (1, D), (1, M)]), (2, True, [(1, M), (2, M)]), (0, True, [(1, S), (2, M)])]),
          contig("t1", 1, [(0, True, [(1, M)]), (0, False, [(1, M)]), (0, True, [(1, I), (1, M)])]),
          contig("t9", 9, [(0, True, [(2, M), (3, D), (2, M)]), (5, False, [(1, M), (2, I), (3, M)]), (8, True, [(1, M)]),
                           (7, False, [(3, M)]), (0, True, [(9, M)]), (3, False, [(1, M), (1, I), (1, D), (1, M), (1, I), (1, M)])]),
          synth.make_contig(43, 3, 3_000, 2, 0.01, 20, "ont")]
    flat = api.FlatBatch(cs)
    t = api.device_tensors(flat)
    o_pile, o_stats, _ = ol.pileup(flat)
    for ev in (4096, 64, 1 << 30):
        pile, stats = api.pileup(t, flat, ev_per_task=ev)
        assert np.array_equal(pile.cpu().numpy(), o_pile), ev
        assert np.array_equal(stats.cpu().numpy()[:, :3], o_stats[:, :3]), ev


def test_column_stats_counts(batch):
    """K2 == histogram of call_variants.cpp:477-501: sorted counts, depth; keys wherever they are unambiguous."""
    from hairsplitter_amd import api
    flat, t = batch
    pile, _ = api.pileup(t, flat)
    st, sel_g, sel_d = api.column_stats(t, flat, pile, min_second=4)
    # 8-bit counter variant (valid: no position of these batches is deeper than 255) must give the same bytes
    st8, sel_g8, sel_d8 = api.column_stats(t, flat, pile, min_second=4, max_depth=255)
    assert int(st["depth"].max()) <= 255
    assert np.array_equal(st8.view(np.uint8), st.view(np.uint8)) and np.array_equal(sel_g8, sel_g) and np.array_equal(sel_d8, sel_d)
    hp = pile.cpu().numpy()
    exp_sel = np.flatnonzero((st["cnt"][:, 1] > 4) | ((st["cnt"][:, 1] == 4) & (st["cnt"][:, 2] == 0)))
    assert np.array_equal(sel_g, exp_sel) and np.array_equal(sel_d, st["depth"][exp_sel].astype(np.int32))
    for c in range(flat.n_contigs):
        k0, k1, c0, c1, c2, depth = ol.column_top3(flat, hp, c)
        s = st[int(flat.contig_off[c]):int(flat.contig_off[c + 1])]
        assert np.array_equal(s["depth"].astype(np.int32), depth)
        assert np.array_equal(s["cnt"][:, 0].astype(np.int32), c0)
        assert np.array_equal(s["cnt"][:, 1].astype(np.int32), c1)
        assert np.array_equal(s["cnt"][:, 2].astype(np.int32), c2)
        strict = (c0 > c1) & (c1 > c2) & (c1 > 0)
        assert np.array_equal(s["key"][:, 0][strict], k0[strict])
        assert np.array_equal(s["key"][:, 1][strict], k1[strict])
        # counts are sorted and sum to at most depth
        cnt = s["cnt"].astype(np.int64)
        assert np.all(np.diff(cnt, axis=1) <= 0) and np.all(cnt.sum(axis=1) <= depth)


def test_tile_plan_lists_every_overlap_once_in_record_order(batch):
    """the host-built tile plan of K2 / K3: every (tile, record) overlap once, ascending record ids per tile"""
    from hairsplitter_amd import api
    flat, t = batch
    plan = api.tile_plan(flat)
    n = 0
    for tl in range(len(plan["h_off"]) - 1):
        recs = plan["h_rec"][plan["h_off"][tl]:plan["h_off"][tl + 1]]
        assert np.all(np.diff(recs) > 0)
        n += len(recs)
    exp = 0
    for c in range(flat.n_contigs):
        g0 = int(flat.contig_off[c])
        for r in range(int(flat.contig_rec_off[c]), int(flat.contig_rec_off[c + 1])):
            if flat.rec_qend[r] > flat.rec_pos[r]:
                exp += ((g0 + int(flat.rec_qend[r]) - 1) >> 8) - ((g0 + int(flat.rec_pos[r])) >> 8) + 1
    assert n == exp


@pytest.fixture(scope="module")
def taps(batch):
    """the column pass of stage 3 AS THE PIPELINE QUEUES IT over the whole batch (hs_cv_column_pass_taps: K0, K1, k_column_stats_tiled_dw,
    k_columns_compact, k_gather_tiles_direct / k_gather_tiles, k_column_top3_exact, k_candidates_scan, k_flag_block_*, k_pack_flagged, k_cand_bits),
    with what those kernels left on the device, + the oracle's pileup of the same batch"""
    from hairsplitter_amd import api
    flat, t = batch
    b = api.CvBatch(flat)
    tp = api.cv_column_pass_taps(b, 0, flat.n_contigs, 0.33)
    b.close()
    o_pile, _, _ = ol.pileup(flat)
    return flat, tp, o_pile


def test_column_pass_selection_and_gather(taps):
    """K2's selection (k_column_stats_tiled_dw + k_columns_compact) and K3 (k_gather_tiles_direct / k_gather_tiles): the extracted columns
    ascend by position, hold every position that can still become a SNP (second count >= 5, or a candidate of call_variants.cpp:525-529) and
    nothing whose second count is below 4, and every column is the reference's Column (Partition.h:8-14) of its position: the reads covering
    it in ascending index with their pileup codes."""
    flat, tp, hp = taps
    g = tp["col_gpos"]
    assert np.all(np.diff(g) > 0)
    ctg = np.searchsorted(flat.contig_off, g, side="right") - 1
    pos = g - flat.contig_off[ctg]
    assert np.array_equal(tp["col_rec"]["contig"], ctg.astype(np.int32)) and np.array_equal(tp["col_rec"]["pos"], pos.astype(np.int32))
    off, idx, code = tp["col_off"], tp["col_idx"], tp["col_code"]
    assert off[0] == 0 and off[-1] == len(idx) == len(code)
    have = set(int(x) for x in g)
    for c in range(flat.n_contigs):
        r0, r1 = int(flat.contig_rec_off[c]), int(flat.contig_rec_off[c + 1])
        g0 = int(flat.contig_off[c])
        k0, k1, c0, c1, c2, depth = ol.column_top3(flat, hp, c)
        md = float(tp["contig_mean_distance"][c])
        fl = ol.call_variants_flags(flat, hp, c, md)
        # (what K2 drops: positions nobody reads -- neither a candidate nor a column loop D could rescue, call_variants.cpp:751-756)
        rescue = np.array([p for p in np.flatnonzero(c1 >= 5) if (int(k0[p]) % 5 != int(k1[p]) % 5 and ((int(k1[p]) - 33) % 5 != 4 or (int(k1[p]) // 5 % 5 != int(k0[p]) % 5 and int(k1[p]) // 25 % 5 != int(k0[p]) % 5)))], dtype=np.int64)
        for p in list(np.flatnonzero((fl & 1).astype(bool))) + list(rescue):
            assert g0 + int(p) in have, (c, int(p))
        for i in np.flatnonzero(ctg == c):
            p = int(pos[i])
            assert c1[p] >= 4
            rs = np.arange(r0, r1)
            cover = rs[(flat.rec_pos[r0:r1] <= p) & (p < flat.rec_qend[r0:r1])]
            assert idx[off[i]:off[i + 1]].tolist() == (cover - r0).tolist()
            assert code[off[i]:off[i + 1]].tolist() == [int(hp[flat.pile_off[r] + p - flat.rec_pos[r]]) for r in cover]


def test_column_pass_leading_codes(taps):
    """K2's second pass + k_column_top3_exact: the two leading codes and their counts of EVERY extracted column are the reference's
    (call_variants.cpp:477-507), equal counts in the order of its hash map and its std::sort included"""
    flat, tp, hp = taps
    rec = tp["col_rec"]
    n = 0
    for c in range(flat.n_contigs):
        k0, k1, c0, c1, c2, _ = ol.column_top3(flat, hp, c)
        sel = np.flatnonzero(rec["contig"] == c)
        p = rec["pos"][sel]
        assert np.array_equal(rec["k0"][sel], k0[p]) and np.array_equal(rec["k1"][sel], k1[p])
        assert np.array_equal(rec["c0"][sel].astype(np.int32), c0[p]) and np.array_equal(rec["c1"][sel].astype(np.int32), c1[p])
        assert np.array_equal(rec["c2"][sel], np.minimum(c2[p], 63))
        n += len(sel)
    assert n == len(rec)


def test_column_pass_candidates(taps):
    """k_candidates_scan: the candidate SNPs (predicate + greedy spacing of call_variants.cpp:525-529, column-parallel on the device) and the
    automatic ones (:531) of every contig, with the contig's own mean distance deciding the read minimum (:463-466); k_flag_block_* +
    k_pack_flagged: the packed candidates are those columns, in order, records intact"""
    from hairsplitter_amd import api  # noqa: F401
    flat, tp, hp = taps
    rec = tp["col_rec"]
    HS_COL_CAND, HS_COL_AUTO = 1, 2
    n_cand = 0
    for c in range(flat.n_contigs):
        fl = ol.call_variants_flags(flat, hp, c, float(tp["contig_mean_distance"][c]))
        sel = np.flatnonzero(rec["contig"] == c)
        p = rec["pos"][sel]
        got_c = (rec["flags"][sel] & HS_COL_CAND) != 0
        got_a = (rec["flags"][sel] & HS_COL_AUTO) != 0
        assert np.array_equal(got_c, (fl[p] & 1) != 0)
        assert np.array_equal(got_a, (fl[p] & 2) != 0)
        assert int(tp["contig_n_cand"][c]) == int((fl & 1).sum()) == int(got_c.sum())      # (no candidate outside the extracted columns)
        n_cand += int(got_c.sum())
    cand_cols = np.flatnonzero((rec["flags"] & HS_COL_CAND) != 0)
    assert len(tp["cand_rec"]) == n_cand and np.array_equal(tp["cand_col"], cand_cols.astype(np.int32))
    assert np.array_equal(tp["cand_rec"].view(np.uint8), rec[cand_cols].view(np.uint8))


def test_column_pass_candidate_bit_sets(taps):
    """k_cand_bits: every candidate column as loop A reads it (hs::CandBits) -- one bit set per distinct code in the order the column's
    entries bring them, bit k = the read of rank k by (start position, index) on its contig -- holds exactly the column's (read, code) pairs"""
    flat, tp, hp = taps
    off, idx, code = tp["col_off"], tp["col_idx"], tp["col_code"]
    bits, words = tp["cand_bits"], tp["cand_words"]
    rank_of = {}
    for c in range(flat.n_contigs):
        r0, r1 = int(flat.contig_rec_off[c]), int(flat.contig_rec_off[c + 1])
        order = np.lexsort((np.arange(r1 - r0), flat.rec_pos[r0:r1]))
        rk = np.zeros(r1 - r0, np.int64); rk[order] = np.arange(r1 - r0)
        rank_of[c] = rk
    ref_span_end = flat.rec_pos + np.asarray(flat.rec_refspan)
    for k, col in enumerate(tp["cand_col"]):
        h = bits[k]
        c = int(tp["cand_rec"]["contig"][k])
        r0 = int(flat.contig_rec_off[c])
        ci, cc = idx[off[col]:off[col + 1]], code[off[col]:off[col + 1]]
        assert int(h["n_entries"]) == len(ci) and int(h["idx_min"]) == int(ci[0]) and int(h["idx_max"]) == int(ci[-1])
        rk = rank_of[c][ci]
        assert int(h["wlo"]) == int(rk.min() >> 6) and int(h["n_words"]) == int(rk.max() >> 6) - int(rk.min() >> 6) + 1
        assert int(h["reach"]) == int(ref_span_end[r0 + ci].max())
        slots = list(dict.fromkeys(int(x) for x in cc))
        assert int(h["n_slots"]) == len(slots)
        W, o = int(h["n_words"]), int(h["word_off"])
        blk = words[o:o + W * (len(slots) + 1) + (len(slots) + 7) // 8]
        exp = np.zeros((len(slots) + 1, W), np.uint64)
        for r, x in zip(rk, cc):
            w, b = int(r >> 6) - int(h["wlo"]), np.uint64(1) << np.uint64(int(r) & 63)
            exp[0, w] |= b; exp[1 + slots.index(int(x)), w] |= b
        assert np.array_equal(blk[:W * (len(slots) + 1)].reshape(len(slots) + 1, W), exp)
        assert blk[W * (len(slots) + 1):].view(np.uint8)[:len(slots)].tolist() == slots


@pytest.mark.parametrize("n", [0, 1, 63, 4095, 4096, 4097, 70_001, 1_000_000, 4_300_000])
def test_exclusive_scan_sizes(n):
    """one tile, tile boundaries, many tiles, more tiles than one pass of the tile-offset scan takes (> 1024 tiles)"""
    from hairsplitter_amd import api
    rng = np.random.default_rng(n)
    v = rng.integers(0, 600, size=n, dtype=np.int32)
    if n > 10:
        v[rng.integers(0, n, size=n // 7)] = 0
        v[3] = 20_000
    exp = np.zeros(n + 1, np.int64)
    np.cumsum(v, dtype=np.int64, out=exp[1:])
    assert np.array_equal(api.exclusive_scan(v), exp)


def test_stage3_result_equals_oracle_pipeline(request, batch, built):
    """hs_cv_run on the resident batch (no files on the product side): SNP positions, ref / alt codes, read indices, codes,
    depth and mean distance of every contig equal what the oracle restatement writes into its .col / error_rate for the
    same contigs."""
    import subprocess, tempfile
    from hairsplitter_amd import api, synth, canon
    flat, _ = batch
    b = api.CvBatch(flat)
    assert b.aligned_bp == flat.aligned_bp
    out = b.run(0.33)
    b.close()
    contigs = _contigs(request.node.callspec.params["batch"])
    with tempfile.TemporaryDirectory() as td:
        f = synth.write_files(contigs, td)
        col, vcf, err = (os.path.join(td, x) for x in ("o.col", "o.vcf", "o.err"))
        subprocess.run([built["oracle"], "call_variants", f["gfa"], f["reads"], f["sam"], "1", td, err, "0", "0", col, vcf, "0.33"],
                       check=True, stdout=subprocess.DEVNULL)
        blocks = canon.split_blocks(col)
        o_err = open(err).read()
    assert "%g" % np.float32(out["error_rate"]) == o_err.strip()      # default ostream precision (call_variants.cpp:1377)
    n_checked = 0
    for c, cd in enumerate(contigs):
        lines = blocks[cd.name]
        head = lines[0].split("\t")
        assert int(head[2]) == len(cd.seq)
        assert "%g" % np.float32(out["depth"][c]) == head[3]
        snps = [l.split("\t") for l in lines if l.startswith("SNPS")]
        s0, s1 = int(out["snp_off"][c]), int(out["snp_off"][c + 1])
        assert s1 - s0 == len(snps)
        for k, fld in enumerate(snps):
            s = s0 + k
            assert int(fld[1]) == int(out["snp_pos"][s]) and int(fld[2]) == int(out["snp_ref"][s]) and int(fld[3]) == int(out["snp_alt"][s])
            e0, e1 = int(out["col_off"][s]), int(out["col_off"][s + 1])
            assert [int(x) for x in fld[4].split(",") if x] == out["col_idx"][e0:e1].tolist()
            assert [int(x) for x in fld[5].split(",") if x] == out["col_code"][e0:e1].tolist()
            n_checked += 1
    assert n_checked == len(out["snp_pos"])


def _partition_test_case(rng, n_contigs, cols_per_contig, mode):
    """random columns x partitions; mode picks the regime: 'snp' = correlated two-allele columns, 'ties' = many codes with equal counts
    (exercises the robin_hood order of the second allele), 'high' = codes >= 128 as most frequent (signed-char quirk), 'many' = like
    'snp' with up to 150 partitions per contig (more than one block of 64 lanes) and columns deeper than 255 reads"""
    col_off = [0]; col_idx = []; col_code = []; col_contig = []; k0s = []; k1s = []; c1s = []; cand = []
    part_off = [0]; pso = []; states = []; nreads = []
    tot_state = 0
    for c in range(n_contigs):
        N = int(rng.integers(6, 180)) if mode != "many" else int(rng.integers(200, 700))
        nreads.append(N)
        F = int(rng.integers(0, 4)) if mode != "many" else int(rng.choice([1, 63, 64, 65, 150]))
        hap = rng.integers(0, 2, N)
        for f in range(F):
            st = np.where(hap == 1, 1, -1).astype(np.int8)
            st[rng.random(N) < 0.1] = 0
            st[rng.random(N) < rng.choice([0.0, 0.3, 0.8])] = 2
            if rng.random() < 0.3:
                st = rng.choice(np.array([1, -1, 0, 2], np.int8), N)
            pso.append(tot_state); states.append(st); tot_state += N
        part_off.append(len(pso))
        for _ in range(cols_per_contig):
            n = int(rng.integers(1, N + 1))
            if mode == "many":
                n = int(rng.choice([rng.integers(5, 70), 255, 256, rng.integers(257, max(N, 258) + 1)]))
                n = min(n, N)
            idx = np.sort(rng.choice(N, n, replace=False)).astype(np.int32)
            if mode in ("snp", "many"):
                a, b = rng.choice(np.arange(33, 158), 2, replace=False)
                code = np.where(hap[idx] == 1, a, b)
                noise = rng.random(n) < 0.08
                code = np.where(noise, rng.integers(33, 158, n), code)
            elif mode == "ties":
                pool = rng.choice(np.arange(33, 158), int(rng.integers(3, 7)), replace=False)
                alt = pool[1 + np.arange(n) % (len(pool) - 1)]          # the non-ref reads split evenly over several codes
                if rng.random() < 0.5:
                    alt = np.where(rng.random(n) < 0.5, alt, pool[1 + (np.arange(n) + 1) % (len(pool) - 1)])   # different codes per haplotype side
                code = np.where(hap[idx] == 1, pool[0], alt)
            else:
                a, b = rng.choice(np.arange(128, 158), 2, replace=False)
                code = np.where(hap[idx] == 1, a, b)
            code = code.astype(np.uint8)
            vals, cnt = np.unique(code, return_counts=True)
            order = np.argsort(-cnt, kind="stable")
            k0 = int(vals[order[0]]) if rng.random() < 0.9 else int(rng.integers(33, 158))   # sometimes a ref the column does not hold
            rest = [(int(cnt[o]), int(vals[o])) for o in order if int(vals[o]) != k0]
            k1 = rest[0][1] if rest else 32
            c1 = rest[0][0] if rest else 0
            col_idx.append(idx); col_code.append(code); col_off.append(col_off[-1] + n); col_contig.append(c)
            k0s.append(k0); k1s.append(k1); c1s.append(c1); cand.append(int(rng.random() < 0.6))
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    return dict(col_off=np.array(col_off, np.int64), col_idx=cat(col_idx, np.int32), col_code=cat(col_code, np.uint8),
                col_contig=np.array(col_contig, np.int32), col_k0=np.array(k0s, np.uint8), col_k1=np.array(k1s, np.uint8),
                col_c1=np.array(c1s, np.int32), col_is_cand=np.array(cand, np.uint8), part_off=np.array(part_off, np.int32),
                part_state_off=np.array(pso, np.int64), part_state=cat(states, np.int8)), nreads


@pytest.mark.parametrize("mode", ["snp", "ties", "high", "many", "snp_third_count", "ties_third_count"])
def test_column_partition_test_matches_oracle(built, mode):
    """K4 (k_column_partition_lanes -> _grouped -> _test) == loops C and D of keep_only_robust_variants. The `_third_count` modes hand the
    kernels what the pipeline does: the column's third count in bits 16-21 of the second count's word (63 - min(c2, 63); 0 = not known),
    which lets the first kernel settle a pair whose second allele is provably the column's own second code."""
    from hairsplitter_amd import api
    base = mode.replace("_third_count", "")
    rng = np.random.default_rng({"snp": 11, "ties": 12, "high": 13, "many": 14}[base])
    case, nreads = _partition_test_case(rng, 24 if base != "many" else 10, 60 if base != "many" else 40, base)
    want, _, _ = ol.column_partition_test(n_reads_of_contig=nreads, **case)
    if mode.endswith("_third_count"):
        c1w = case["col_c1"].copy()
        for k in range(len(c1w)):
            e0, e1 = int(case["col_off"][k]), int(case["col_off"][k + 1])
            vals, cnt = np.unique(case["col_code"][e0:e1], return_counts=True)
            rest = sorted((int(c) for v, c in zip(vals, cnt) if int(v) not in (int(case["col_k0"][k]), int(case["col_k1"][k]))), reverse=True)
            c2 = rest[0] if rest else 0
            c1w[k] = int(c1w[k]) | ((63 - min(c2, 63)) << 16)
        case = dict(case, col_c1=c1w)
    got = api.column_partition_test(n_reads=nreads, **case)
    assert np.array_equal(got, want)
    assert 0 < int(want.sum()) < len(want)          # both verdicts occur
    # every kernel of the chain had its share: the first one settles most columns but not all, ties reach the pair test, reference codes
    # >= 128 and columns deeper than 255 the whole-column test
    cnt = api.column_partition_last_counts()
    assert 0 < cnt["to_grouped"] < len(want)
    if base == "ties":
        assert cnt["pairs_to_exact"] > 0
    if base in ("high", "many"):
        assert cnt["whole_columns_to_exact"] > 0
    if mode == "snp_third_count":
        assert cnt["to_grouped"] < 0.5 * len(want)      # (with the third count known the first kernel settles most columns on its own)


def test_simdiff_matches_oracle(built):
    """K5 == list_similarities_and_differences_between_reads3 (separate_reads.cpp:374-433)."""
    from hairsplitter_amd import api
    rng = np.random.default_rng(3)
    for (N, S) in [(1, 1), (37, 5), (64, 64), (130, 700), (257, 1030)]:
        snp_ref = rng.integers(33, 158, S).astype(np.uint8)
        snp_alt = ((snp_ref.astype(np.int32) - 33 + rng.integers(1, 124, S)) % 125 + 33).astype(np.uint8)
        col_off, col_idx, col_code = [0], [], []
        for s in range(S):
            reads = np.flatnonzero(rng.random(N) < 0.6)
            u = rng.random(len(reads))
            codes = np.where(u < 0.5, snp_ref[s], np.where(u < 0.9, snp_alt[s], 40)).astype(np.uint8)
            col_idx += reads.tolist(); col_code += codes.tolist(); col_off.append(len(col_idx))
        W = (S + 63) // 64
        alt = np.zeros((N, W), np.uint64); ref = np.zeros((N, W), np.uint64)
        for s in range(S):
            for e in range(col_off[s], col_off[s + 1]):
                if col_code[e] == snp_ref[s]:
                    ref[col_idx[e], s >> 6] |= np.uint64(1) << np.uint64(s & 63)
                elif col_code[e] == snp_alt[s]:
                    alt[col_idx[e], s >> 6] |= np.uint64(1) << np.uint64(s & 63)
        d_alt, d_ref = api.snp_planes(N, snp_ref, snp_alt, col_off, col_idx, col_code)      # K5a: the same planes, built on the device
        assert np.array_equal(d_alt, alt) and np.array_equal(d_ref, ref)
        sim, diff = api.simdiff(alt, ref)
        o_sim, o_diff = ol.simdiff(N, snp_ref, snp_alt, col_off, col_idx, col_code)
        assert np.array_equal(sim, o_sim) and np.array_equal(diff, o_diff)
        assert np.array_equal(sim, sim.T) and np.all(np.diag(sim) == 0)


def _simdiff_np(alt, ref):
    a = alt.astype(np.int32); r = ref.astype(np.int32)
    sim = 3 * a @ a.T + r @ r.T
    diff = a @ r.T + r @ a.T
    np.fill_diagonal(sim, 0); np.fill_diagonal(diff, 0)
    return sim.astype(np.int32), diff.astype(np.int32)


@pytest.mark.parametrize("regime", ["haplotypes", "ties", "sparse"])
def test_read_graphs_match_oracle(built, regime):
    """K6 against create_read_graph_matrix, window by window; 'ties' has so few SNPs that many rows fall back to std::sort"""
    from hairsplitter_amd import api
    rng = np.random.default_rng({"haplotypes": 21, "ties": 22, "sparse": 23}[regime])
    sims, diffs, windows = [], [], []
    for c in range(6):
        N = int(rng.integers(2, 260)) if c else 70
        S = {"haplotypes": 120, "ties": 16, "sparse": 30}[regime]
        hap = rng.integers(0, 3, N)
        truth = rng.integers(0, 2, (3, S))
        cover = rng.random((N, S)) < {"haplotypes": 0.6, "ties": 1.0, "sparse": 0.15}[regime]
        allele = truth[hap] ^ (rng.random((N, S)) < (0.15 if regime == "ties" else 0.05))
        alt = (cover & (allele == 1)); ref = (cover & (allele == 0))
        sim, diff = _simdiff_np(alt, ref)
        sims.append(sim); diffs.append(diff)
        for _ in range(5):
            frac = rng.choice([0.1, 0.5, 1.0])
            ids = np.flatnonzero(rng.random(N) < frac).astype(np.int32)
            windows.append((c, ids))
    windows.append((0, np.zeros(0, np.int32)))          # a window without masked reads
    for err in (0.05, 0.15):
        got, n_host = api.read_graphs(sims, diffs, windows, err)
        for (c, ids), g in zip(windows, got):
            N = sims[c].shape[0]
            mask = np.zeros(N, np.uint8); mask[ids] = 1
            want = ol.read_graph(sims[c], diffs[c], mask, err)
            assert sorted(g.keys()) == ids.tolist()
            for r in range(N):
                assert g.get(r, []) == want[r], (regime, err, c, r)
        if regime == "ties":
            assert n_host > 0


@pytest.mark.parametrize("kind", ["dip", "multi", "penta"])
def test_clustering_chain_kernels_match_oracle(built, kind):
    """The clustering chain of stage 4 as hs_sr_run queues it, kernel by kernel against the oracle's separate_reads_on_contig (test taps on both
    sides): for every window with seeding SNPs the same reads (k_window_masks), the labels of every per-SNP Chinese-Whispers run
    (separate_reads.cpp:1674-1705, cluster_graph.cpp:240-310: k_cw_seed_sets + k_cw_seeded_lanes / _rows / _wave on the graphs of
    k_read_graph_rows, visiting orders of k_cw_visit_lists), the labels of the run behind finalize_clustering's small-cluster filter (:897-970,
    first half of k_window_tail) as a partition of the window's reads, and the finished labels of every window (:973-993, cluster_graph.cpp:402-501,
    separate_reads.cpp:1007-1327: the rest of k_window_tail)."""
    from hairsplitter_amd import api, synth
    if kind == "penta":
        contigs = [synth.make_contig(21, 0, 24_000, 5, 0.012, 60, "ont")]
    else:
        contigs = _contigs(kind)
    flat = api.FlatBatch(contigs)
    b = api.CvBatch(flat)
    cv = b.run(0.33)
    b.close()
    er = min(float("%g" % np.float32(cv["error_rate"])), 0.15)
    out = api.separate_reads(cv, flat, er, taps=True)
    tp = out["taps"]
    n_runs = n_win = 0
    for c, ctg in enumerate(out["contigs"]):
        o = ol.sr_contig_taps(ctg, out["window_size"], er)
        N = len(ctg["read_start"])
        # the finished labels of every window of the contig
        w0, w1 = int(out["win_off"][c]), int(out["win_off"][c + 1])
        assert w1 - w0 == len(o["win_start"])
        for k in range(w1 - w0):
            assert int(out["win_start"][w0 + k]) == int(o["win_start"][k]) and int(out["win_end"][w0 + k]) == int(o["win_end"][k])
            got = out["labels"][int(out["label_off"][w0 + k]):int(out["label_off"][w0 + k + 1])]
            assert np.array_equal(got, o["win_labels"][k]), (c, k)
        # the chain's windows of this contig, in order: those of the oracle that have runs
        mine = np.flatnonzero(tp["win_contig"] == c)
        theirs = [k for k in range(len(o["tap_start"])) if o["run_begin"][k + 1] > o["run_begin"][k]]
        assert [int(tp["win_start"][k]) for k in mine] == [int(o["tap_start"][k]) for k in theirs]
        o_lab_off = 0
        o_off = {}
        for k in range(len(o["tap_start"])):
            m = int(o["tap_row0"][k + 1] - o["tap_row0"][k])
            o_off[k] = o_lab_off
            o_lab_off += m * int(o["run_begin"][k + 1] - o["run_begin"][k])
        for km, ko in zip(mine, theirs):
            ids = tp["mask_ids"][int(tp["win_row0"][km]):int(tp["win_row0"][km + 1])]
            oid = o["mask_ids"][int(o["tap_row0"][ko]):int(o["tap_row0"][ko + 1])]
            assert np.array_equal(ids, oid)
            m = len(ids)
            r0, r1 = int(tp["run_begin"][km]), int(tp["run_begin"][km + 1])
            assert np.array_equal(tp["run_snp"][r0:r1], o["run_snp"][int(o["run_begin"][ko]):int(o["run_begin"][ko + 1])])
            for i in range(r0, r1):
                got = tp["run_labels"][int(tp["run_off"][i]):int(tp["run_off"][i]) + m]
                exp = o["run_labels"][o_off[ko] + (i - r0) * m:o_off[ko] + (i - r0 + 1) * m]
                assert np.array_equal(got, exp), (c, int(tp["win_start"][km]), i - r0)
                n_runs += 1
            # the third run: the same partition of the window's reads (the oracle numbers the clusters by first appearance before it goes on, :972-981)
            g3 = tp["third"][int(tp["win_row0"][km]):int(tp["win_row0"][km + 1])]
            e3 = o["third"][int(o["tap_row0"][ko]):int(o["tap_row0"][ko + 1])]
            assert np.array_equal(g3 < 0, e3 < 0)
            pairs = set(zip(g3[g3 >= 0].tolist(), e3[e3 >= 0].tolist()))
            assert len(pairs) == len(set(p[0] for p in pairs)) == len(set(p[1] for p in pairs)), (c, int(tp["win_start"][km]))
            n_win += 1
        assert N >= 0
    assert n_win > 0 and n_runs > n_win


def test_myers_matches_edlib_vectors_and_oracle(built):
    """A1 == edlibAlign distance / first end location (golden vectors from the reference's bundled edlib)."""
    from hairsplitter_amd import api
    vec = json.load(open(os.path.join(gu.GOLD, "edlib_vectors.json"))) + json.load(open(os.path.join(gu.GOLD, "edlib_edge_vectors.json")))
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    for mode in ("NW", "HW", "SHW"):
        vs = [v for v in vec if v["mode"] == mode]
        q = [np.array([code[c] for c in v["query"]], np.uint8) for v in vs]
        t = [np.array([code[c] for c in v["target"]], np.uint8) for v in vs]
        d, e = api.edit_distance(q, t, mode)
        assert d.tolist() == [v["distance"] for v in vs], mode
        assert e.tolist() == [v["end"] for v in vs], mode
    # multi-pass path (query > 4096 rows) against the DP oracle
    rng = np.random.default_rng(2)
    qs, ts = [], []
    for qn in (4097, 5000, 9001):
        base = rng.integers(0, 4, qn).astype(np.uint8)
        keep = rng.random(qn) > 0.04
        tt = base[keep].copy()
        m = rng.random(len(tt)) < 0.05
        tt[m] = (tt[m] + 1) & 3
        qs.append(base); ts.append(tt)
    for mode, mi in (("NW", 0), ("HW", 2)):
        d, e = api.edit_distance(qs, ts, mode)
        for k in range(len(qs)):
            od, oe = ol.edit_distance(qs[k], ts[k], mi)
            assert d[k] == od and e[k] == oe, (mode, k, d[k], od, e[k], oe)


def test_edlib_hw_path_matches_the_reference_edlib(built):
    """A1 as the stage-5 call sites use the reference's bundled edlib (HW, k = -1, TASK_PATH; create_new_contigs.cpp:558-629,
    tools.cpp:515-534): edit distance, first end location, its start location and the alignment itself, move by move."""
    from hairsplitter_amd import api
    vec = json.load(open(os.path.join(gu.GOLD, "edlib_path_vectors.json")))
    got = api.edlib_hw_align([(v["query"], v["target"]) for v in vec])
    sym = "=IDX"
    n_paths = 0
    for v, g in zip(vec, got):
        assert g["distance"] == v["distance"], (v["query"][:30], v["target"][:30])
        assert g["end"] == v["end"]
        if v["end"] >= 0:
            assert g["start"] == v["start"]
        if v["cigar"] != "*":
            ops = g["ops"]
            assert ops is not None
            runs = []
            for o in ops.tolist():
                if runs and runs[-1][1] == sym[o]:
                    runs[-1][0] += 1
                else:
                    runs.append([1, sym[o]])
            assert "".join("%d%s" % (c, s) for c, s in runs) == v["cigar"]
            n_paths += 1
    assert n_paths > 200
    # locations only (edlib's TASK_LOC): same numbers without the path
    loc = api.edlib_hw_align([(v["query"], v["target"]) for v in vec[:50]], path=False)
    assert [(g["distance"], g["end"]) for g in loc] == [(v["distance"], v["end"]) for v in vec[:50]]
    # short queries share a wavefront (8 lanes per pair here); a wavefront per pair gives the same
    os.environ["HS_MYERS_NO_GROUPS"] = "1"
    try:
        alone = api.edlib_hw_align([(v["query"], v["target"]) for v in vec])
    finally:
        del os.environ["HS_MYERS_NO_GROUPS"]
    for g, a in zip(got, alone):
        assert (g["distance"], g["start"], g["end"]) == (a["distance"], a["start"], a["end"])
        assert (g["ops"] is None) == (a["ops"] is None) and (g["ops"] is None or np.array_equal(g["ops"], a["ops"]))


def test_edlib_hw_path_beyond_one_leaf_matches_reference_edlib(built):
    """Queries of 1.5-60 kb: above 1 MB of its own bookkeeping edlib cuts the alignment in halves (Hirschberg, edlib.cpp:1166-1404) and
    the cuts decide which optimal alignment comes out -- k_myers_hw_path makes the same cuts. Vectors: the reference's edlib
    (oracle/gen_goldens.py --edlib-long)."""
    import gzip
    from hairsplitter_amd import api
    vec = json.loads(gzip.open(os.path.join(gu.GOLD, "edlib_long_path_vectors.json.gz")).read())
    got = api.edlib_hw_align([(v["query"], v["target"]) for v in vec])
    sym = "=IDX"
    for v, g in zip(vec, got):
        assert (g["distance"], g["start"], g["end"]) == (v["distance"], v["start"], v["end"]), (len(v["query"]), len(v["target"]))
        ops = g["ops"]
        assert ops is not None, (len(v["query"]), len(v["target"]))
        cut = np.flatnonzero(np.diff(ops)) + 1
        runs = np.diff(np.concatenate(([0], cut, [len(ops)])))
        heads = ops[np.concatenate(([0], cut))]
        assert "".join("%d%s" % (c, sym[o]) for c, o in zip(runs.tolist(), heads.tolist())) == v["cigar"], (len(v["query"]), len(v["target"]))
    assert max(len(v["query"]) for v in vec) >= 60000


def _check_paths_against_vectors(vec, got):
    sym = "=IDX"
    for v, g in zip(vec, got):
        assert (g["distance"], g["end"]) == (v["distance"], v["end"]), (len(v["query"]), len(v["target"]))
        if v["end"] >= 0:
            assert g["start"] == v["start"]
        if v["cigar"] == "*":
            continue
        ops = g["ops"]
        assert ops is not None, (len(v["query"]), len(v["target"]))
        cut = np.flatnonzero(np.diff(ops)) + 1
        runs = np.diff(np.concatenate(([0], cut, [len(ops)])))
        heads = ops[np.concatenate(([0], cut))]
        assert "".join("%d%s" % (c, sym[o]) for c, o in zip(runs.tolist(), heads.tolist())) == v["cigar"], (len(v["query"]), len(v["target"]))


def test_edlib_hw_path_every_lane_grouping_in_one_call(built):
    """Queries of 330-2048 bases whose matrix edlib keeps whole take 8, 16 or 32 lanes of a wavefront (k_myers_hw_path_grouped);
    here one call holds all of those classes, stage-5-sized pairs and pairs that need a wavefront and Hirschberg's cuts. Vectors:
    the reference's edlib (oracle/gen_goldens.py --edlib-mid / --edlib-path / --edlib-long)."""
    import gzip
    from hairsplitter_amd import api
    mid = json.loads(gzip.open(os.path.join(gu.GOLD, "edlib_mid_path_vectors.json.gz")).read())
    short = json.load(open(os.path.join(gu.GOLD, "edlib_path_vectors.json")))[:40]
    long_ = [v for v in json.loads(gzip.open(os.path.join(gu.GOLD, "edlib_long_path_vectors.json.gz")).read()) if len(v["query"]) <= 5000][:8]
    vec = []
    for i in range(max(len(mid), len(short), len(long_))):      # interleaved: neighbours in the call are of different classes
        vec += [x[i] for x in (mid, short, long_) if i < len(x)]
    nb = [(len(v["query"]) + 63) // 64 for v in mid]
    assert any(b <= 8 for b in nb) and any(8 < b <= 16 for b in nb) and any(16 < b <= 32 for b in nb)
    got = api.edlib_hw_align([(v["query"], v["target"]) for v in vec])
    _check_paths_against_vectors(vec, got)
    loc = api.edlib_hw_align([(v["query"], v["target"]) for v in vec], path=False)
    assert [(g["distance"], g["end"]) for g in loc] == [(v["distance"], v["end"]) for v in vec]
    os.environ["HS_MYERS_NO_GROUPS"] = "1"
    try:
        alone = api.edlib_hw_align([(v["query"], v["target"]) for v in vec])
    finally:
        del os.environ["HS_MYERS_NO_GROUPS"]
    _check_paths_against_vectors(vec, alone)
    # the scratch of a call is bounded: with 2 MB of it the pairs go out in many chunks that take the same scratch in turn
    os.environ["HS_MYERS_SCRATCH_MB"] = "2"
    try:
        chunked = api.edlib_hw_align([(v["query"], v["target"]) for v in vec])
    finally:
        del os.environ["HS_MYERS_SCRATCH_MB"]
    _check_paths_against_vectors(vec, chunked)


def test_edit_distance_long_and_mixed_pairs_match_reference_edlib(built):
    """hs_edit_distance (NW / SHW / HW distance + first end location) on pairs of 0.3-60 kb, every lane grouping and the
    wavefront-per-pair kernel in one call: the banded sweeps with their bound found on the way. Expected: the reference's edlib
    (oracle/gen_goldens.py --edlib-long-dist)."""
    import gzip
    from hairsplitter_amd import api
    exp = json.load(open(os.path.join(gu.GOLD, "edlib_long_dist_vectors.json")))
    code = np.full(256, 3, np.uint8)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    enc = lambda x: code[np.frombuffer(x.encode(), dtype=np.uint8)]
    pairs, want = [], {"NW": [], "SHW": [], "HW": []}
    for name in ("edlib_mid_path_vectors", "edlib_long_path_vectors"):
        vec = json.loads(gzip.open(os.path.join(gu.GOLD, name + ".json.gz")).read())
        for v, e in zip(vec, exp[name]):
            pairs.append((enc(v["query"]), enc(v["target"])))
            for m in want:
                want[m].append(e[m])
    assert max(len(q) for q, _ in pairs) >= 60000
    for mode in ("NW", "SHW", "HW"):
        d, e = api.edit_distance([q for q, _ in pairs], [t for _, t in pairs], mode)
        w = np.array(want[mode])
        bad = np.flatnonzero((d != w[:, 0]) | (e != w[:, 1]))
        assert len(bad) == 0, (mode, [(int(i), len(pairs[i][0]), len(pairs[i][1]), int(d[i]), int(e[i]), w[i].tolist()) for i in bad[:5]])


def test_edlib_hw_path_random_pairs_against_the_numpy_restatement(built):
    """Seeded random pairs of 1-2600 bases (substitutions, insertions, deletions at 0-35 %, unrelated pairs, repeats) through
    hs_edlib_hw_align, against oracle/edlib_path_oracle.py -- the numpy restatement that the CPU suite pins on the reference's
    edlib vectors. Covers sizes and error rates between the committed vectors."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import edlib_path_oracle as eo
    from hairsplitter_amd import api
    rng = np.random.default_rng(20261002)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rs = lambda n: acgt[rng.integers(0, 4, size=n)].tobytes().decode()

    def mutate(q, rate):
        out = []
        for ch in q:
            u = rng.random()
            if u < rate / 3:
                out.append("ACGT"[rng.integers(0, 4)])
            elif u < 2 * rate / 3:
                continue
            elif u < rate:
                out.append(ch); out.append("ACGT"[rng.integers(0, 4)])
            else:
                out.append(ch)
        return "".join(out)
    pairs = []
    for i in range(64):
        qn = int(rng.choice([1, 7, 63, 64, 65, 130, 300, 511, 513, 900, 1024, 1500, 2048, 2600]))
        q = rs(qn)
        kind = i % 4
        if kind == 0:
            t = rs(int(rng.integers(0, 400))) + mutate(q, float(rng.choice([0.0, 0.02, 0.1, 0.35]))) + rs(int(rng.integers(0, 400)))
        elif kind == 1:
            t = rs(int(rng.integers(1, 2 * qn + 2)))
        elif kind == 2:
            unit = rs(int(rng.integers(1, 6)))
            q = (unit * (qn // len(unit) + 1))[:qn]
            t = mutate(q, 0.08) + unit * 3
        else:
            t = mutate(q, 0.05)[: max(1, qn // 2)]
        pairs.append((q, t or "A"))
    got = api.edlib_hw_align(pairs)
    for (q, t), g in zip(pairs, got):
        e = eo.hw_align(q, t)
        assert (g["distance"], g["end"]) == (e["distance"], e["end"]), (len(q), len(t))
        if e["end"] >= 0:
            assert g["start"] == e["start"], (len(q), len(t))
        assert (g["ops"] is None) == (e["ops"] is None)
        if e["ops"] is not None:
            assert g["ops"].tolist() == e["ops"], (len(q), len(t))


def test_stage5_edlib_call_sites(built):
    """The two stage-5 computations that sit on the reference's edlib calls, batched on the A1 kernel: the ends racon dropped are
    attached again (tools.cpp:505-536) and the overhangs are cut off the polished piece (create_new_contigs.cpp:556-629).
    Expected sequences: those lines restated around the reference's own edlib (oracle/edlib_driver.cpp)."""
    from hairsplitter_amd import api
    cases = json.load(open(os.path.join(gu.GOLD, "stage5_edlib_cases.json")))
    re_ = [c for c in cases if c["kind"] == "reattach"]
    tr = [c for c in cases if c["kind"] == "trim"]
    assert len(re_) > 30 and len(tr) > 30
    got = api.reattach_ends([c["backbone"] for c in re_], [c["consensus"] for c in re_])
    assert got == [c["expected"] for c in re_]
    got = api.trim_polished([c["to_polish"] for c in tr], [c["newcontig"] for c in tr], [c["overhang_left"] for c in tr], [c["overhang_right"] for c in tr])
    assert got == [c["expected"] for c in tr]
    assert any(len(c["expected"]) < len(c["newcontig"]) for c in tr)


def test_stage5_call_sites_compare_bytes_like_edlib(built):
    """edlib's default equality compares bytes: lower case is not upper case, N only matches N. The kernel has four codes, so every
    query / target pair gets its own bijection bytes -> codes (distance and path only depend on which positions are equal): cases
    over acgt, ACGN, mixed case ... against the reference's edlib; a pair with five distinct bytes is refused, not approximated."""
    from hairsplitter_amd import api
    cases = json.load(open(os.path.join(gu.GOLD, "stage5_alphabet_cases.json")))
    re_ = [c for c in cases if c["kind"] == "reattach"]
    tr = [c for c in cases if c["kind"] == "trim"]
    assert len(re_) >= 40 and len(tr) >= 40
    assert api.reattach_ends([c["backbone"] for c in re_], [c["consensus"] for c in re_]) == [c["expected"] for c in re_]
    assert api.trim_polished([c["to_polish"] for c in tr], [c["newcontig"] for c in tr], [c["overhang_left"] for c in tr], [c["overhang_right"] for c in tr]) == [c["expected"] for c in tr]
    bad = [c for c in cases if c["kind"] == "refused"]
    assert bad
    with pytest.raises(Exception, match="distinct bytes"):
        api.reattach_ends([bad[0]["backbone"]], [bad[0]["consensus"]])


def test_partition_pair_distance_matches_oracle(built):
    """V5 distance(Partition, Partition, 2) (call_variants.cpp:977-1127), one wavefront per pair, against the oracle on random
    partitions: overlapping and disjoint read sets, zero states, vote counts around the 3-sigma thresholds, both phasings"""
    from hairsplitter_amd import api
    import oracle_lib as ol
    rng = np.random.default_rng(11)
    state, more, less, part_off, part_n, pa, pb = [], [], [], [], [], [], []
    off = 0
    for N in (1, 63, 64, 65, 200, 1000, 3000):
        first = len(part_n)
        base = rng.choice([-1, 1], N)
        for p in range(8):
            lo = int(rng.integers(0, max(1, N // 2))); hi = int(rng.integers(lo, N)) + 1
            st = np.full(N, 2, np.int8)
            phase = 1 if p % 2 == 0 else -1
            flip = rng.random(N) < (0.05 if p < 6 else 0.5)
            vals = np.where(flip, -base, base) * phase
            vals = np.where(rng.random(N) < 0.1, 0, vals)
            st[lo:hi] = vals[lo:hi]
            st[rng.random(N) < 0.15] = 2
            mo = rng.integers(0, 12, N).astype(np.int32); le = rng.integers(0, 6, N).astype(np.int32)
            mo[rng.random(N) < 0.05] += 40
            state.append(st); more.append(mo); less.append(le); part_off.append(off); part_n.append(N); off += N
        for i in range(8):
            for j in range(8):
                if i != j:
                    pa.append(first + i); pb.append(first + j)
    state = np.concatenate(state); more = np.concatenate(more); less = np.concatenate(less)
    got = api.partition_pair_distance(state, more, less, part_off, part_n, pa, pb)
    want = ol.partition_pair_distance(state, more, less, part_off, part_n, pa, pb)
    assert got[:, 6].all()
    assert (got[:, :6] == want).all(), np.nonzero((got[:, :6] != want).any(axis=1))[0][:10]
    assert (want[:, 5] == 1).any() and (want[:, 5] == 0).any() and (want[:, 4] == -1).any() and (want[:, 4] == 1).any()
