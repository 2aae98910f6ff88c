"""GPU parity, file to file: the HIP drop-in executables against the reference goldens (and the oracle)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", gu.case_names())
def test_dropin_binaries_match_reference_goldens(built, case):
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack(case, td)
        outs = gu.run_stage_pair([built["cv"]], [built["sr"]], td, meta)
        assert gu.compare(td, outs) == []
        # ... and what the next stage makes of the .gro this build wrote: the .gaf the reference derives from its own .gro
        gaf = os.path.join(td, "t_reads.gaf")
        subprocess.run([built["gaf"], os.path.join(td, "assembly.gfa"), gu.reads_path(td, meta), os.path.join(td, "aln.sam"), outs["gro"],
                        str(meta.get("kwargs", {}).get("amplicon", 0)), gaf], check=True, stdout=subprocess.DEVNULL)
        assert open(gaf, "rb").read() == open(os.path.join(td, "reads_haplo.gaf"), "rb").read()


@pytest.mark.parametrize("switch", ["HS_SPIN_WAIT", "HS_FINISH_ON_HOST", "HS_LOOP_A_ON_DEVICE", "HS_LOOP_B_PAIRS_ON_DEVICE",
                                    "HS_NO_REEXEC", "HS_EXIT_LEAK", "HS_PINNED_HOSTMALLOC", "HS_NO_DETACH"])
@pytest.mark.parametrize("case", ["penta30k", "edge_ops"])
def test_dropin_binaries_with_the_alternative_paths(built, case, switch):
    """The switches select other ways to the same result (no huge-page restart, nothing destroyed at the exit, pinned blocks from hipHostMalloc,
    one process; back-to-back polling waits, cluster merging on the host, loop A of keep_only_robust_variants on the device, loop B's pair distances
    from the device): the executables must still reproduce the reference goldens"""
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack(case, td)
        outs = gu.run_stage_pair([built["cv"]], [built["sr"]], td, meta, env=dict(os.environ, **{switch: "1"}))
        assert gu.compare(td, outs) == []


@pytest.mark.parametrize("case", [c for c in gu.case_names() if "lowmem" in c or "deep" in c])
def test_dropin_binaries_with_low_memory_graphs_on_the_host(built, case):
    """create_read_graph_low_memory runs on the device by default (window-local sim / diff from the bit rows, the row kernel with
    that path's distance, rows with a NaN resolved on the host); HS_LOW_MEMORY_GRAPHS_ON_HOST=1 keeps the host builder: same files"""
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack(case, td)
        outs = gu.run_stage_pair([built["cv"]], [built["sr"]], td, meta, env=dict(os.environ, HS_LOW_MEMORY_GRAPHS_ON_HOST="1"))
        assert gu.compare(td, outs) == []


@pytest.mark.parametrize("case", gu.case_names())
def test_dropin_binaries_with_loop_b_pair_distances_from_the_device(built, case):
    """distance(Partition, Partition) of loop B from k_partition_pair_distance (opt-in) against every golden case"""
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack(case, td)
        outs = gu.run_stage_pair([built["cv"]], [built["sr"]], td, meta, env=dict(os.environ, HS_LOOP_B_PAIRS_ON_DEVICE="1"))
        assert gu.compare(td, outs) == []


@pytest.mark.parametrize("case", gu.case_names())
def test_dropin_binaries_with_loop_a_on_the_device(built, case):
    """k_loop_a (loop A of keep_only_robust_variants as one wave per contig over bit sets in LDS, opt-in) against every golden case"""
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack(case, td)
        outs = gu.run_stage_pair([built["cv"]], [built["sr"]], td, meta, env=dict(os.environ, HS_LOOP_A_ON_DEVICE="1"))
        assert gu.compare(td, outs) == []


@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0,0"])
@pytest.mark.parametrize("case", ["multi", "linked", "simple_mock", "tetra25k_ploidy2"])
def test_dropin_binaries_sharded_over_devices(built, case, devices):
    """Several GPUs in one process (hs_cv_run_host / hs_sr_run shard the contigs over hs_devices() by LPT, one host thread per
    device, results merged in contig order). HS_DEVICES may list a device more than once: on this single-GPU box every shard
    runs on device 0, which exercises the sharding, the per-shard batches with remapped reads and the merge -- the outputs
    must still be the reference's."""
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack(case, td)
        outs = gu.run_stage_pair([built["cv"]], [built["sr"]], td, meta, env=dict(os.environ, HS_DEVICES=devices))
        assert gu.compare(td, outs) == []


def test_device_list_follows_the_environment(built):
    import ctypes
    from hairsplitter_amd import api
    lib = api.load()
    buf = (ctypes.c_int32 * 16)()
    old = os.environ.get("HS_DEVICES")
    try:
        os.environ["HS_DEVICES"] = "0,0,0"
        assert lib.hs_devices(buf, 16) == 3 and list(buf[:3]) == [0, 0, 0]
        os.environ.pop("HS_DEVICES")
        assert lib.hs_devices(buf, 16) == lib.hs_device_count()
    finally:
        if old is not None:
            os.environ["HS_DEVICES"] = old


def test_inmemory_labels_to_gaf(built):
    """resident batch -> pipeline -> hs_gaf_from_labels, never touching .col / .gro text == drop-in executables -> hs_gro_to_gaf"""
    from hairsplitter_amd import api, synth
    contigs = [synth.make_contig(31, 0, 20_000, 3, 0.01, 40, "ont", name="a"), synth.make_contig(31, 1, 8_000, 1, 0.0, 30, "ont", name="b_noSNP"),
               synth.make_contig(31, 2, 25_000, 2, 0.01, 35, "ont", name="c")]
    with tempfile.TemporaryDirectory() as td:
        f = synth.write_files(contigs, td)
        col, vcf, err, gro = (os.path.join(td, x) for x in ("o.col", "o.vcf", "o.err", "o.gro"))
        subprocess.run([built["cv"], f["gfa"], f["reads"], f["sam"], "2", td, err, "0", "0", col, vcf, "0.33"], check=True, stdout=subprocess.DEVNULL)
        e = min(float(open(err).read()), 0.15)
        subprocess.run([built["sr"], col, "2", str(e), os.path.join(td, "none"), "0", "0.01", "0", gro, "0"], check=True, stdout=subprocess.DEVNULL)
        a, b_ = os.path.join(td, "files.gaf"), os.path.join(td, "mem.gaf")
        api.gaf_from_files(f["gfa"], f["reads"], f["sam"], gro, a)
        flat = api.FlatBatch(contigs)
        b = api.CvBatch(flat)
        cv, sr = b.run_pipeline(0.33, 2)
        b.close()
        assert int(sr["win_off"][2] - sr["win_off"][1]) == 0          # the contig without SNPs has no windows
        api.gaf_from_labels(f["gfa"], f["reads"], f["sam"], sr, b_)
        assert open(a, "rb").read() == open(b_, "rb").read() and os.path.getsize(a) > 0


def test_native_library_is_what_ran(built):
    """The executables are linked against the in-tree HIP library (no site-packages copy, no fallback)."""
    out = subprocess.run(["ldd", built["cv"]], stdout=subprocess.PIPE).stdout.decode()
    assert "libhairsplitter_hip.so" in out and "hairsplitter_amd/lib" in out.replace("bin/../lib", "lib")


def _parse_col(path):
    out = {}
    cur = None
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        if f[0] == "CONTIG":
            cur = out.setdefault(f[1], [])
        elif f[0] == "SNPS":
            cur.append((int(f[1]), int(f[2]), int(f[3]), [int(x) for x in f[4].split(",") if x], [int(x) for x in f[5].split(",") if x]))
    return out


def _parse_gro(path):
    out = {}
    cur = None
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        if f[0] == "CONTIG":
            cur = out.setdefault(f[1], [])
        elif f[0] == "GROUP":
            cur.append((int(f[1]), int(f[2]), [int(x) for x in f[3].split(",") if x], [int(x) for x in f[4].split(",") if x]))
    return out


def test_inmemory_pipeline_equals_oracle(built):
    """The single-batch in-memory calls (hs_cv_batch resident in HBM -> hs_cv_run -> hs_sr_run, no files) give the same SNP
    columns and partition labels as the oracle run file-to-file on the same synthetic contigs. (What bench.py times is the grouped,
    fused call -- api.PipelineGroups.run_fused -- compared with the reference itself at full size in
    tests/test_gpu_full_configs.py::test_bench_path_equals_reference_c3 / _c4 and with this single-batch form in
    test_pipeline_groups_equal_single_batch.)"""
    import numpy as np
    from hairsplitter_amd import api, synth
    contigs = [synth.make_contig(21, i, 30_000, 2 + i, 0.01, 40, "ont") for i in range(3)]
    flat = api.FlatBatch(contigs)
    b = api.CvBatch(flat)
    cv = b.run(0.33, 2)
    b.close()
    # only the columns the host walks cross PCIe: candidates and tie-order columns up front, rescued SNPs afterwards
    assert 0 < cv["n_columns_downloaded"] < cv["n_columns_extracted"] and cv["n_columns_downloaded_late"] > 0
    e = float("%g" % cv["error_rate"])
    sr = api.separate_reads(cv, flat, min(e, 0.15), rarest_strain_abundance=0.01, n_threads=2)
    with tempfile.TemporaryDirectory() as td:
        f = synth.write_files(contigs, td)
        col, vcf, err, gro = (os.path.join(td, x) for x in ("o.col", "o.vcf", "o.err", "o.gro"))
        subprocess.run([built["oracle"], "call_variants", f["gfa"], f["reads"], f["sam"], "1", td, err, "0", "0", col, vcf, "0.33"], check=True, stdout=subprocess.DEVNULL)
        assert float(open(err).read()) == e
        subprocess.run([built["oracle"], "separate_reads", col, "1", str(min(e, 0.15)), os.path.join(td, "none"), "0", "0.01", "0", gro, "0"], check=True, stdout=subprocess.DEVNULL)
        ocol, ogro = _parse_col(col), _parse_gro(gro)
    for c, ctg in enumerate(contigs):
        s0, s1 = int(cv["snp_off"][c]), int(cv["snp_off"][c + 1])
        exp = ocol[ctg.name]
        assert s1 - s0 == len(exp)
        for k, s in enumerate(range(s0, s1)):
            e0, e1 = int(cv["col_off"][s]), int(cv["col_off"][s + 1])
            assert (int(cv["snp_pos"][s]), int(cv["snp_ref"][s]), int(cv["snp_alt"][s])) == exp[k][:3]
            assert cv["col_idx"][e0:e1].tolist() == exp[k][3] and cv["col_code"][e0:e1].tolist() == exp[k][4]
        w0, w1 = int(sr["win_off"][c]), int(sr["win_off"][c + 1])
        gexp = ogro.get(ctg.name, [])
        assert w1 - w0 == len(gexp)
        for k, w in enumerate(range(w0, w1)):
            lab = sr["labels"][int(sr["label_off"][w]):int(sr["label_off"][w + 1])]
            present = np.flatnonzero(lab != -2)
            assert (int(sr["win_start"][w]), int(sr["win_end"][w])) == gexp[k][:2]
            assert present.tolist() == gexp[k][2] and lab[present].tolist() == gexp[k][3]


def test_fused_pipeline_equals_two_step_path(built):
    """hs_cv_run -> hs_sr_run_cv (stage 3 -> 4 handed over inside the library) == hs_cv_run -> Python hand-over -> hs_sr_run."""
    import numpy as np
    from hairsplitter_amd import api, synth
    contigs = [synth.make_contig(22, i, 25_000, 2 + (i % 3), 0.01, 40, "ont") for i in range(4)]
    flat = api.FlatBatch(contigs)
    b = api.CvBatch(flat)
    cv = b.run(0.33, 2)
    e = min(float("%g" % cv["error_rate"]), 0.15)
    sr = api.separate_reads(cv, flat, e, rarest_strain_abundance=0.01, n_threads=2)
    cv2, sr2 = b.run_pipeline(0.33, 2)
    b.close()
    assert cv2["n_snps"] == int(cv["snp_off"][-1]) and cv2["error_rate"] == cv["error_rate"]
    for k in ("win_off", "win_start", "win_end", "label_off", "labels"):
        assert np.array_equal(sr[k], sr2[k]), k


def test_fused_pipeline_takes_the_ploidy_of_the_contigs(built):
    """hs_cv_batch_set_ploidy: the in-memory hand-over caps the clusters per window as HS_separate_reads does with a ploidy file
    (separate_reads.cpp:1711-1715); equal to the two-step path that is handed the same ploidies."""
    import numpy as np
    from hairsplitter_amd import api, synth
    contigs = [synth.make_contig(23, i, 25_000, 4, 0.01, 40, "ont") for i in range(3)]
    flat = api.FlatBatch(contigs)
    b = api.CvBatch(flat)
    cv = b.run(0.33, 2)
    e = min(float("%g" % cv["error_rate"]), 0.15)
    ploidy = [2, 0, 3]
    want = api.separate_reads(cv, flat, e, rarest_strain_abundance=0.01, n_threads=2, ploidy=ploidy)
    free = api.separate_reads(cv, flat, e, rarest_strain_abundance=0.01, n_threads=2)
    b.set_ploidy(ploidy)
    _, got = b.run_pipeline(0.33, 2)
    b.set_ploidy(None)
    _, got_free = b.run_pipeline(0.33, 2)
    b.close()
    assert not np.array_equal(want["labels"], free["labels"])      # the cap does something on these tetraploid contigs
    for k in ("win_off", "win_start", "win_end", "label_off", "labels"):
        assert np.array_equal(want[k], got[k]), k
        assert np.array_equal(free[k], got_free[k]), k


@pytest.mark.parametrize("shape", ["tetra100k", "hifi300k", "meta_ploidy1to8", "deep1200x", "shallow14x_ploidy3", "shallow10x_ploidy2", "dense60x_ploidy8", "dense120x_ploidy6"])
def test_dropin_equals_oracle_at_larger_sizes(built, shape):
    """Beyond the committed fixtures: BASELINE-shaped contigs (tetraploid ONT as C3, a metagenome slice with ploidies 1..8 at
    30x total depth as C4, HiFi 300 kb chunk as C5) through the drop-in executables, against the oracle restatement run on
    the same files."""
    from hairsplitter_amd import synth, canon
    if shape == "tetra100k":
        contigs = [synth.make_contig(41, 0, 100_000, 4, 0.01, 40, "ont")]
    elif shape == "deep1200x":
        # coverage > 1000: 16-bit histogram counters in K2, the per-contig low-memory graph path (separate_reads.cpp:1515-1518),
        # finalize_clustering on a graph that was never filled (:1708), 2700 reads per Chinese-Whispers instance, 500-bp windows
        contigs = [synth.make_contig(77, 0, 4000, 2, 0.01, 1200, "ont", read_len_override=(1000, 2500))]
    elif shape == "shallow14x_ploidy3":
        # fourteen reads over three haplotypes: leading counts tie all the time (5 : 5 : 4, 4 : 4 ...), so K2's second pass hands those
        # positions to k_column_top3_exact, and the floor of the selection (second count 4 with no third allele) decides many columns
        contigs = [synth.make_contig(44, 0, 60_000, 3, 0.02, 14, "ont")]
    elif shape == "shallow10x_ploidy2":
        contigs = [synth.make_contig(44, 0, 60_000, 2, 0.02, 10, "ont")]
    elif shape == "dense60x_ploidy8":
        # 5 % divergence over eight haplotypes at 60x: a 256-position tile holds more positions with two counters of four reads than
        # a wavefront has lanes (K2's second pass in several rounds), SNPs every few bases (V1's chains of passing columns)
        contigs = [synth.make_contig(46, 0, 40_000, 8, 0.05, 60, "ont")]
    elif shape == "dense120x_ploidy6":
        contigs = [synth.make_contig(46, 0, 30_000, 6, 0.08, 120, "ont")]
    elif shape == "meta_ploidy1to8":
        contigs = [synth.make_contig(43, i, 20_000 + 5_000 * i, 1 + i, 0.01 if i else 0.0, 30, "ont") for i in range(8)]
    else:
        contigs = [synth.make_contig(42, 0, 300_000, 2, 0.001, 30, "hifi")]
    with tempfile.TemporaryDirectory() as td:
        f = synth.write_files(contigs, td)
        outs = {}
        for tag, cv, sr in (("hip", [built["cv"]], [built["sr"]]), ("orc", [built["oracle"], "call_variants"], [built["oracle"], "separate_reads"])):
            col, vcf, err, gro = (os.path.join(td, tag + x) for x in (".col", ".vcf", ".err", ".gro"))
            subprocess.run(cv + [f["gfa"], f["reads"], f["sam"], "4", td, err, "0", "0", col, vcf, "0.33"], check=True, stdout=subprocess.DEVNULL)
            e = min(float(open(err).read().strip()), 0.15)
            subprocess.run(sr + [col, "4", str(e), os.path.join(td, "no_ploidy"), "0", "0.01", "0", gro, "0"], check=True, stdout=subprocess.DEVNULL)
            outs[tag] = (col, vcf, err, gro)
        assert canon.split_blocks(outs["hip"][0]) == canon.split_blocks(outs["orc"][0])
        assert canon.vcf_blocks(outs["hip"][1]) == canon.vcf_blocks(outs["orc"][1])
        assert open(outs["hip"][2]).read() == open(outs["orc"][2]).read()
        assert canon.split_blocks(outs["hip"][3]) == canon.split_blocks(outs["orc"][3])
        assert sum(1 for l in open(outs["hip"][3]) if l.startswith("GROUP")) > {"meta_ploidy1to8": 20, "deep1200x": 5, "shallow14x_ploidy3": 20, "shallow10x_ploidy2": 20, "dense60x_ploidy8": 10, "dense120x_ploidy6": 10}.get(shape, 40)


def test_pipeline_groups_equal_single_batch(built):
    """concurrent sub-batches (one host thread + HIP stream each) == the same contigs as one batch"""
    from hairsplitter_amd import api, synth
    contigs = [synth.make_contig(7, i, 30_000, 2 + (i % 3), 0.01, 40, "ont") for i in range(6)]
    single = api.CvBatch(api.FlatBatch(contigs))
    cv1, sr1 = single.run_pipeline(0.33, 8)
    groups = api.PipelineGroups(contigs, 4)
    for _ in range(2):
        cv2, sr2 = groups.run(0.33, 8)
        assert np.array_equal(cv1["mean_distance"], cv2["mean_distance"])
        assert np.float32(cv1["error_rate"]) == np.float32(cv2["error_rate"])
        assert cv1["n_snps"] == cv2["n_snps"]
        assert np.array_equal(sr1["labels"], sr2["labels"])
        assert np.array_equal(sr1["win_start"], sr2["win_start"]) and np.array_equal(sr1["win_end"], sr2["win_end"])
    # the same job in ONE library call (hs_pipeline_run_fused: pileup per group, the error rate formed inside)
    for _ in range(2):
        cv3, sr3 = groups.run_fused(0.33, 8)
        assert np.array_equal(cv1["mean_distance"], cv3["mean_distance"])
        assert np.float32(min(float("%g" % cv1["error_rate"]), 0.15)) == np.float32(cv3["error_rate"])
        assert cv1["n_snps"] == cv3["n_snps"]
        assert np.array_equal(sr1["labels"], sr3["labels"])
        assert np.array_equal(sr1["win_start"], sr3["win_start"]) and np.array_equal(sr1["win_end"], sr3["win_end"])
    single.close(); groups.close()



def test_pipeline_options_sparse_labels_and_kept_columns(built):
    """HS_PIPELINE_SPARSE_LABELS: per window the reads it holds and their labels == the non -2 entries of the dense array;
    HS_PIPELINE_KEEP_COLUMNS: the groups' stage-3 results then carry the SNP columns' entries (.col's payload), equal to what the
    single-batch call returns; neither option changes the labels; step after step (the second step runs on the sizes the first left)"""
    import ctypes as C
    from hairsplitter_amd import api, synth
    contigs = [synth.make_contig(11, i, 24_000, 2 + (i % 3), 0.012, 36, "ont") for i in range(5)]
    single = api.CvBatch(api.FlatBatch(contigs))
    cv1, sr1 = single.run_pipeline(0.33, 8)
    full = single.run(0.33, 8)      # stage 3 alone, entries on the host
    pg = api.PipelineGroups(contigs, 3)
    lib = api.load()
    try:
        dense = pg.run_fused(0.33, 8)[1]["labels"].copy()
        assert np.array_equal(dense, sr1["labels"])
        pg.sparse_labels(True)
        pg.keep_columns(True)
        for _ in range(3):
            cv2, sr2 = pg.run_fused(0.33, 8)
            assert sr2["labels"] is None
            off, ids, lab = sr2["sparse"]
            lo = sr2["label_off"]
            rebuilt = np.full(int(lo[-1]), -2, np.int32)
            for w in range(len(off) - 1):
                rebuilt[lo[w] + ids[off[w]:off[w + 1]]] = lab[off[w]:off[w + 1]]
            assert np.array_equal(rebuilt, sr1["labels"])
            assert np.all(lab != -2)
            # the groups' stage-3 results, concatenated in contig order, are the single call's
            lib.hs_pipeline_group_cv.restype = C.POINTER(api.CvResult)
            pos, idx, code = [], [], []
            for g in range(lib.hs_pipeline_groups(pg.handle)):
                r = lib.hs_pipeline_group_cv(pg.handle, C.c_int32(g)).contents
                S = int(r.snp_off[r.n_contigs]); E = int(r.col_off[S])
                assert bool(r.col_idx) and bool(r.col_code)
                pos.append(np.ctypeslib.as_array(r.snp_pos, (max(S, 1),))[:S].copy())
                idx.append(np.ctypeslib.as_array(r.col_idx, (max(E, 1),))[:E].copy())
                code.append(np.ctypeslib.as_array(r.col_code, (max(E, 1),))[:E].copy())
            assert np.array_equal(np.concatenate(pos), full["snp_pos"])
            assert np.array_equal(np.concatenate(idx), full["col_idx"]) and np.array_equal(np.concatenate(code), full["col_code"])
    finally:
        pg.close(); single.close()


def test_degenerate_batches(built):
    """empty and ragged inputs through the in-memory path: no contigs; a single contig spread over more groups than contigs;
    a contig without reads and a contig without SNPs next to a normal one"""
    from hairsplitter_amd import api, synth
    empty = api.CvBatch(api.FlatBatch([]))
    cv = empty.run(0.33, 4)
    assert len(cv["mean_distance"]) == 0 and len(cv["snp_pos"]) == 0
    empty.close()
    one = [synth.make_contig(9, 0, 20_000, 2, 0.01, 40, "ont")]
    ref_cv, ref_sr = api.CvBatch(api.FlatBatch(one)).run_pipeline(0.33, 4)
    g = api.PipelineGroups(one, 8)
    cv2, sr2 = g.run(0.33, 4)
    assert np.array_equal(ref_sr["labels"], sr2["labels"]) and ref_cv["n_snps"] == cv2["n_snps"]
    g.close()
    ragged = [synth.make_contig(9, 1, 15_000, 1, 0.0, 0, "ont", name="noreads"), synth.make_contig(9, 2, 12_000, 1, 0.0, 30, "ont", name="nosnp"),
              synth.make_contig(9, 3, 18_000, 3, 0.012, 45, "ont", name="tri")]
    a_cv, a_sr = api.CvBatch(api.FlatBatch(ragged)).run_pipeline(0.33, 4)
    g = api.PipelineGroups(ragged, 3)
    b_cv, b_sr = g.run(0.33, 4)
    assert np.array_equal(a_cv["mean_distance"], b_cv["mean_distance"]) and a_cv["mean_distance"][0] == 0
    assert np.array_equal(a_sr["labels"], b_sr["labels"]) and np.array_equal(a_sr["win_off"], b_sr["win_off"])
    c_cv, c_sr = g.run_fused(0.33, 4)
    assert np.array_equal(a_cv["mean_distance"], c_cv["mean_distance"])
    assert np.array_equal(a_sr["labels"], c_sr["labels"]) and np.array_equal(a_sr["win_off"], c_sr["win_off"])
    assert a_sr["win_off"][1] == 0 and a_sr["win_off"][2] == 0        # contigs without SNPs produce no windows (separate_reads.cpp:1522-1524)
    g.close()


def test_cluster_merging_on_device_equals_host_code(built):
    """K8 (first-seen renumbering + merge_close_clusters + merge_wrongly_split per window on the device) against the same
    steps in the product's host code (HS_FINISH_ON_HOST=1 routes every window there), on polyploid contigs where clusters
    do get merged; and the device path must be the one that ran."""
    from hairsplitter_amd import api, synth
    contigs = [synth.make_contig(31, i, 40_000, 2 + i, 0.012, 45, "ont") for i in range(4)]
    b = api.CvBatch(api.FlatBatch(contigs))
    cv, sr = b.run_pipeline(0.33, 8)
    os.environ["HS_FINISH_ON_HOST"] = "1"
    try:
        cv_h, sr_h = b.run_pipeline(0.33, 8)
    finally:
        del os.environ["HS_FINISH_ON_HOST"]
    b.close()
    assert np.array_equal(sr["labels"], sr_h["labels"])
    n_windows = int(sr["win_off"][-1])
    assert sr_h["n_windows_finished_on_host"] > 0.5 * n_windows          # the switch works
    assert sr["n_windows_finished_on_host"] < 0.1 * n_windows            # and the device is what normally runs
    assert len(set(sr["labels"].tolist())) > 3                           # real clusterings, not all-zero windows


@pytest.mark.gpu
def test_separate_reads_with_many_alleles_per_column(built):
    """Stage 4 on a hand-made .col whose SNP columns carry up to two dozen different codes, on contigs of 40 .. 300 reads that all
    span the contig: the per-SNP Chinese-Whispers runs then start from more labels than the one-run-per-lane kernel keeps in
    registers (-> its overflow list), windows of exactly 64 and of 65 reads sit on either side of that kernel's limit, and the
    300-read contig takes the one-wavefront-per-run kernel. Byte-equal .gro against the oracle."""
    import numpy as np
    rng = np.random.default_rng(77)
    L = 6300
    lines = []
    for ci, N in enumerate([40, 64, 65, 130, 300]):
        lines.append(f"CONTIG\tc{ci}\t{L}\t{N}.0")
        for r in range(N):
            lines.append(f"READ\tc{ci}_r{r}\t0\t{L}\t0\t{L}\t1")
        hap = rng.integers(0, 3, N)
        for pos in range(150, L - 150, 45):
            ref, alt = (int(x) for x in rng.choice(np.arange(40, 64), 2, replace=False))
            allele = rng.integers(0, 2, 3)
            if allele.min() == allele.max():
                allele[0] ^= 1
            codes = np.where(allele[hap] == 0, ref, alt)
            noisy = rng.random(N) < (0.5 if pos % 2 else 0.1)
            codes = np.where(noisy, rng.integers(70, 94, N), codes)
            covered = np.flatnonzero(rng.random(N) < 0.97)
            lines.append(f"SNPS\t{pos}\t{ref}\t{alt}\t" + "".join(f"{i}," for i in covered) + "\t" + "".join(f"{int(codes[i])}," for i in covered))
    with tempfile.TemporaryDirectory() as td:
        col = os.path.join(td, "many.col")
        open(col, "w").write("\n".join(lines) + "\n")
        outs = []
        for exe, tag in ((built["sr"], "hip"), (built["oracle"], "oracle")):
            gro = os.path.join(td, tag + ".gro")
            cmd = ([exe] if tag == "hip" else [exe, "separate_reads"]) + [col, "2", "0.05", os.path.join(td, "none"), "0", "0", "0", gro, "0"]
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
            outs.append(open(gro, "rb").read())
        assert len(outs[0]) > 1000 and outs[0] == outs[1]


def test_pipeline_group_threads_run_on_the_device_of_their_batch(built):
    """The HIP current device is a per-thread setting that starts at 0: every contig-group thread of a pipeline must bind itself
    to the device its batch was created on (with one visible device that is device 0 -- the binding itself is what is checked:
    a thread that failed to bind reports -1)."""
    import ctypes
    from hairsplitter_amd import api, synth
    lib = api.load()
    contigs = [synth.make_contig(3, k, 6000, 2, 0.01, 20, "ont") for k in range(4)]
    pg = api.PipelineGroups(contigs, 3)
    try:
        dev = lib.hs_cv_batch_device(pg.batch.handle)
        assert dev == 0
        buf = (ctypes.c_int32 * 8)(*([-7] * 8))
        n = lib.hs_pipeline_thread_devices(pg.handle, buf, 8)
        assert n == 3 and list(buf[:3]) == [dev] * 3
        cv, sr = pg.run()          # ... and the pipeline works from a thread that was never bound by the caller
        assert cv["n_snps"] > 0
    finally:
        pg.close()


def test_killing_the_dropin_leaves_no_worker_behind(built):
    """The executables do their work in a forked child (hs_dropin_main.h). A caller that kills the process it started must not
    leave that child holding the GPU and writing the outputs: SIGTERM is passed on, and the child dies with its parent."""
    import signal
    import time
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack("penta30k", td)
        args = [os.path.join(td, "assembly.gfa"), gu.reads_path(td, meta), os.path.join(td, "aln.sam"), "2", td, os.path.join(td, "err.txt"),
                "0", "0", os.path.join(td, "o.col"), os.path.join(td, "o.vcf"), "0.33"]
        for sig in (signal.SIGTERM, signal.SIGKILL):
            p = subprocess.Popen([built["cv"]] + args, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            kids = []
            for _ in range(200):      # the child appears within milliseconds
                try:
                    kids = [int(x) for x in open(f"/proc/{p.pid}/task/{p.pid}/children").read().split()]
                except OSError:
                    kids = []
                if kids or p.poll() is not None:
                    break
                time.sleep(0.005)
            if p.poll() is not None:
                continue      # finished before it could be killed: nothing to check
            assert kids, "the drop-in did not fork its worker"
            p.send_signal(sig)
            p.wait(timeout=30)
            deadline = time.time() + 20
            while time.time() < deadline and any(os.path.exists(f"/proc/{k}") and "Z" not in open(f"/proc/{k}/stat").read().split(")")[-1][:3] for k in kids):
                time.sleep(0.05)
            for k in kids:
                alive = os.path.exists(f"/proc/{k}") and "Z" not in open(f"/proc/{k}/stat").read().split(")")[-1][:3]
                assert not alive, f"worker {k} survived its parent being killed with signal {int(sig)}"


# ---- the precomputed .gro (<col>.hsgro): HS_call_variants' epilogue, adopted by HS_separate_reads only for exactly the call it was made for

def _stage3(built, td, meta, env):
    col, vcf, err = (os.path.join(td, "t_" + n) for n in ("variants.col", "variants.vcf", "error_rate.txt"))
    amplicon = str(meta.get("kwargs", {}).get("amplicon", 0))
    r = subprocess.run([built["cv"], os.path.join(td, "assembly.gfa"), gu.reads_path(td, meta), os.path.join(td, "aln.sam"), "1", td, err,
                        amplicon, "0", col, vcf, "0.33"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    return col


def _stage4(built, td, col, out, env, error_rate, ploidy="absent_ploidy.txt", low_memory="0", rsa="0.01", amplicon="0"):
    r = subprocess.run([built["sr"], col, "1", error_rate, os.path.join(td, ploidy), low_memory, rsa, amplicon, os.path.join(td, out), "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(env, HS_TIMING="1"))
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    return "precomputed .gro" in r.stdout.decode(), open(os.path.join(td, out), "rb").read()


def test_precomputed_gro_is_adopted_for_the_default_call_only(built):
    """HS_call_variants leaves <col>.hsgro (stage 4 for the arguments hairsplitter.py passes by default, hairsplitter.py:686-692,725-726);
    HS_separate_reads copies it when its own call is that one, and computes -- to the same file as with HS_NO_PRECOMPUTE=1 -- whenever an
    argument, the seed, the ploidies or the .col differ"""
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack("penta30k", td)
        env = dict(os.environ, HS_NO_DETACH="1")
        col = _stage3(built, td, meta, env)
        assert os.path.exists(col + ".hsgro") and not os.path.exists(col + ".hsgro.tmp")
        er = meta["error_rate_arg"]
        off = dict(env, HS_NO_PRECOMPUTE="1")
        took, gro = _stage4(built, td, col, "a.gro", env, er)
        assert took, "the default call must find its companion"
        took_off, gro_off = _stage4(built, td, col, "b.gro", off, er)
        assert not took_off and gro == gro_off
        from hairsplitter_amd import canon
        assert canon.split_blocks(os.path.join(td, "a.gro")) == canon.split_blocks(os.path.join(td, "reads_haplo.gro"))
        open(os.path.join(td, "ploidy.txt"), "w").write("ctg0\t3\n")
        other = [dict(error_rate="0.031"), dict(error_rate=er, low_memory="1"), dict(error_rate=er, rsa="0.05"), dict(error_rate=er, amplicon="1"),
                 dict(error_rate=er, ploidy="ploidy.txt")]
        for k, kw in enumerate(other):
            took, a = _stage4(built, td, col, "c%d.gro" % k, env, **kw)
            took_off, b = _stage4(built, td, col, "d%d.gro" % k, off, **kw)
            assert not took and not took_off and a == b, kw
        took, a = _stage4(built, td, col, "e.gro", dict(env, HS_SEED="777"), er)
        _, b = _stage4(built, td, col, "f.gro", dict(off, HS_SEED="777"), er)
        assert not took and a == b
        # the .col edited after stage 3 (one allele of one SNPS line, same size): neither the arrays beside it nor the .gro describe it any more
        text = open(col, "rb").read()
        at = text.index(b"SNPS\t")
        line_end = text.index(b"\n", at)
        fields = text[at:line_end].split(b"\t")
        fields[2] = b"A" if fields[2] != b"A" else b"C"
        open(col, "wb").write(text[:at] + b"\t".join(fields) + text[line_end:])
        took, a = _stage4(built, td, col, "g.gro", env, er)
        _, b = _stage4(built, td, col, "h.gro", off, er)
        assert not took and a == b


def test_precomputed_gro_that_never_arrives_is_not_waited_for_long(built):
    """A marker without a companion (HS_call_variants died in its epilogue): HS_separate_reads waits HS_PRECOMPUTE_WAIT_MS at most, then computes"""
    import time
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack("dip20k", td)
        env = dict(os.environ, HS_NO_DETACH="1")
        col = _stage3(built, td, meta, env)
        os.rename(col + ".hsgro", col + ".hsgro.tmp")
        t0 = time.time()
        took, a = _stage4(built, td, col, "a.gro", dict(env, HS_PRECOMPUTE_WAIT_MS="100"), meta["error_rate_arg"])
        assert not took and time.time() - t0 < 30
        os.remove(col + ".hsgro.tmp")
        _, b = _stage4(built, td, col, "b.gro", dict(env, HS_NO_PRECOMPUTE="1"), meta["error_rate_arg"])
        assert a == b


def test_marker_of_a_maker_that_died_is_not_waited_for(built):
    """The marker holds the process id of its maker: one that no longer exists (HS_call_variants killed in its epilogue) means nothing will come"""
    import time
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack("dip20k", td)
        env = dict(os.environ, HS_NO_DETACH="1")
        col = _stage3(built, td, meta, env)
        os.remove(col + ".hsgro")
        gone = subprocess.Popen(["true"]); gone.wait()
        open(col + ".hsgro.tmp", "w").write("%d\n" % gone.pid)
        t0 = time.time()
        took, a = _stage4(built, td, col, "a.gro", dict(env, HS_PRECOMPUTE_WAIT_MS="60000"), meta["error_rate_arg"])
        assert not took and time.time() - t0 < 20 and not os.path.exists(col + ".hsgro.tmp")
        _, b = _stage4(built, td, col, "b.gro", dict(env, HS_NO_PRECOMPUTE="1"), meta["error_rate_arg"])
        assert a == b


@pytest.mark.parametrize("case", ["multi", "linked", "simple_mock"])
def test_stage_4_right_behind_a_detached_stage_3(built, case):
    """As hairsplitter.py runs them: HS_separate_reads started the moment HS_call_variants' exit status is in, while its worker is still
    writing the companion -- whichever way stage 4 goes (adopts, waits and adopts, computes), the .gro is the reference's"""
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack(case, td)
        outs = gu.run_stage_pair([built["cv"]], [built["sr"]], td, meta, env={k: v for k, v in os.environ.items() if k != "HS_NO_DETACH"})
        assert gu.compare(td, outs) == []
