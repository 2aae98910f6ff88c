"""CPU tests (no GPU): the oracle restatement against the reference goldens, library-behaviour vectors,
the product's host glue (through the oracle-backed harness), and the C-ABI surface."""
import json
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

import golden_util as gu
import oracle_lib as ol

ROOT = gu.ROOT


@pytest.mark.parametrize("case", gu.case_names())
def test_oracle_matches_reference_goldens(built, case):
    """oracle/_build/hs_oracle, run like the reference binaries, reproduces the reference's own outputs
    (tests/golden/*, produced by oracle/gen_goldens.py from oracle/_ref)."""
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack(case, td)
        outs = gu.run_stage_pair([built["oracle"], "call_variants"], [built["oracle"], "separate_reads"], td, meta)
        assert gu.compare(td, outs) == []


@pytest.mark.parametrize("case", gu.case_names())
def test_host_glue_matches_reference_goldens(built, case):
    """The product's host-side logic (dense partitions, window planning, cluster merging, file I/O) with the
    device interface served by the oracle: must equal the reference goldens."""
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack(case, td)
        outs = gu.run_stage_pair([built["harness"], "call_variants"], [built["harness"], "separate_reads"], td, meta)
        assert gu.compare(td, outs) == []


@pytest.mark.parametrize("case", ["multi", "penta30k", "edge_ops"])
def test_host_glue_through_the_per_range_selection(built, case):
    """Same, the way the contig groups go through stage 3: pileup of the batch (cv_pileup), then cv_run_range on two consecutive
    ranges of contigs with the batch's per-record counters, the two results put together (cv_concat_results)"""
    if case not in gu.case_names():
        pytest.skip("golden case not present")
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack(case, td)
        env = dict(os.environ, HS_HARNESS_RANGE_SELECT="1")
        outs = gu.run_stage_pair([built["harness"], "call_variants"], [built["harness"], "separate_reads"], td, meta, env=env)
        assert gu.compare(td, outs) == []


def test_robin_hood_order_vectors(built):
    vec = json.load(open(os.path.join(gu.GOLD, "robin_hood_order.json")))
    for v in vec:
        got = ol.rh_order_u8(v["keys"]) if v["type"] == "u8" else ol.rh_order_int(v["keys"])
        assert got == v["order"], v
    # the product's own emulator (hs_rh8.h) against the same vectors
    u8 = [v for v in vec if v["type"] == "u8"]
    inp = "\n".join("u8 " + " ".join(map(str, v["keys"])) for v in u8) + "\n"
    exe = os.path.join(ROOT, "tests", "harness", "_build", "rh8_selftest")
    out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.splitlines()
    for o, v in zip(out, u8):
        assert list(map(int, o.split())) == v["order"]


def test_host_driver_selftest(built):
    """Pieces of the stage drivers that need neither a device nor input files (tests/harness/host_harness.cpp `selftest`): the
    labels of the contig groups spread into the dense result (sr_expand_labels), the recycled result block, and the per-range
    selection of the contig groups (cv_select_range: boundary tiles bring positions of the neighbouring ranges)"""
    r = subprocess.run([built["harness"], "selftest"], capture_output=True, text=True)
    assert r.returncode == 0 and "selftest ok" in r.stdout, r.stdout + r.stderr


def test_hash_map_view_and_std_sort_restatements(built):
    """What the device uses to order equal counts exactly as the reference does (k_column_top3_exact): hs::Rh8View against the
    pinned emulator, hs::CountSort against std::sort of this libstdc++ -- random, tie-heavy and adversarial (heap-sort branch) inputs"""
    exe = os.path.join(ROOT, "tests", "harness", "_build", "sort_selftest")
    r = subprocess.run([exe, "200000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert " bad 0;" in r.stdout and r.stdout.strip().split(" bad ")[2].startswith("0 ")
    assert int(r.stdout.split("heap-paths")[1]) > 50


def test_hash_map_view_holds_every_key_set_of_the_path(built):
    """hs::Rh8View at the capacity the kernels give it (512 bytes per table in LDS): 128 distinct byte keys never set its overflow flag and
    iterate like the unbounded emulator (the kernels trap on the flag instead of going on with another order)"""
    exe = os.path.join(os.path.dirname(built["harness"]), "rh8_selftest")
    r = subprocess.run([exe, "worstcase"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout


def test_rh8_static_order(built):
    """The closed form k_robust_partitions uses for the iteration order of small hash maps (hs_kernels_parts.hip) against the
    emulator, on two million random key sets"""
    exe = os.path.join(ROOT, "tests", "harness", "_build", "rh8_static_order")
    r = subprocess.run([exe, "2000000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert " bad 0 " in r.stdout


def test_shuffle_permutations_pinned(built):
    """libstdc++ mt19937(12345) + std::shuffle (SURVEY.md appendix B known answers)."""
    assert ol.shuffled_order(5).tolist() == [0, 1, 4, 3, 2]
    assert ol.shuffled_order(10).tolist() == [6, 8, 3, 5, 1, 2, 9, 7, 4, 0]
    assert ol.shuffled_order(17).tolist() == [11, 9, 13, 16, 2, 1, 12, 5, 0, 10, 7, 8, 14, 15, 6, 4, 3]


def test_edit_distance_oracle_matches_edlib_vectors(built):
    # edlib_edge_vectors.json: nothing in the target beats the all-insertions score (queries of k * 64 bases differ from the others)
    vec = json.load(open(os.path.join(gu.GOLD, "edlib_vectors.json"))) + json.load(open(os.path.join(gu.GOLD, "edlib_edge_vectors.json")))
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    mode = {"NW": 0, "SHW": 1, "HW": 2}
    for v in vec:
        q = np.array([code[c] for c in v["query"]], np.uint8)
        t = np.array([code[c] for c in v["target"]], np.uint8)
        d, e = ol.edit_distance(q, t, mode[v["mode"]])
        assert d == v["distance"], v["mode"]
        assert e == v["end"], (v["mode"], d, e, v["end"])


def test_alignment_path_oracle_matches_edlib_vectors():
    """oracle/edlib_path_oracle.py (distance, locations and the alignment move by move, Hirschberg cuts included) against the
    reference's own edlib: the stage-5 shapes and, for queries up to 5 kb, the long vectors (the GPU test takes all of them)."""
    import gzip
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import edlib_path_oracle as eo
    short = json.load(open(os.path.join(gu.GOLD, "edlib_path_vectors.json")))
    long_ = [v for v in json.loads(gzip.open(os.path.join(gu.GOLD, "edlib_long_path_vectors.json.gz")).read()) if len(v["query"]) <= 5000]
    assert len(long_) >= 20
    long_ += json.loads(gzip.open(os.path.join(gu.GOLD, "edlib_mid_path_vectors.json.gz")).read())[::2]
    stats = {}
    for v in short[::3] + long_:
        g = eo.hw_align(v["query"], v["target"], stats=stats)
        assert (g["distance"], g["end"]) == (v["distance"], v["end"])
        if v["end"] >= 0:
            assert g["start"] == v["start"]
        if v["cigar"] != "*":
            assert eo.cigar(g["ops"]) == v["cigar"], (len(v["query"]), len(v["target"]))
    assert stats["splits"] > 20 and stats["leaves"] > 40      # both of obtainAlignment's branches were taken


def test_col_file_with_a_read_index_outside_the_read_list_is_refused(built):
    """A damaged .col whose SNPS line names a read beyond the contig's READ lines: the reference indexes out of bounds, the kernels
    would too -- hs::parse_col refuses the file (found by running mutated inputs through the AddressSanitizer build of the harness,
    tools/host_asan.py)."""
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack("simple_mock", td)
        p = os.path.join(td, "variants.col")
        lines = open(p).read().split("\n")
        k = next(i for i, l in enumerate(lines) if l.startswith("SNPS"))
        f = lines[k].split("\t")
        idx = f[4].split(",")
        idx[0] = "999999"
        f[4] = ",".join(idx)
        lines[k] = "\t".join(f)
        open(p, "w").write("\n".join(lines))
        r = subprocess.run([built["harness"], "separate_reads", p, "1", meta["error_rate_arg"], os.path.join(td, "absent_ploidy.txt"), "0", "0.01", "0",
                            os.path.join(td, "o.gro"), "0"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode != 0 and b"read index outside" in r.stdout, r.stdout[-500:]


def test_c_abi_exports_every_declared_symbol(built):
    from hairsplitter_amd import api
    hdr = open(os.path.join(ROOT, "include", "hairsplitter_hip.h")).read()
    declared = set(re.findall(r"\b(hs_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"hs_colstat", "hs_cv_batch", "hs_cv_result", "hs_sr_contig", "hs_sr_result"}
    assert declared == set(api.SYMBOLS), declared ^ set(api.SYMBOLS)
    lib = api.load()
    for s in declared:
        assert hasattr(lib, s), s
    assert b"gfx950" in lib.hs_version()


def test_product_refuses_to_run_without_gpu(built):
    """No CPU fallback: on a box without a HIP device the drop-in executables fail loudly."""
    from hairsplitter_amd import api
    if api.load().hs_device_count() > 0:
        pytest.skip("a GPU is present")
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack("dip20k", td)
        r = subprocess.run([built["cv"], os.path.join(td, "assembly.gfa"), os.path.join(td, "reads.fasta"), os.path.join(td, "aln.sam"),
                            "1", td, os.path.join(td, "e.txt"), "0", "0", os.path.join(td, "o.col"), os.path.join(td, "o.vcf"), "0.33"],
                           stdout=subprocess.PIPE)
        assert r.returncode != 0 and b"no HIP device" in r.stdout
        with pytest.raises(api.HsError):
            api.require_gpu()


def test_probe_calls_of_the_orchestrator(built):
    """hairsplitter.py:229-252 probes `HS_call_variants --version` and `HS_separate_reads --help`, expecting 0."""
    assert subprocess.run([built["cv"], "--version"], stdout=subprocess.DEVNULL).returncode == 0
    assert subprocess.run([built["sr"], "--help"], stdout=subprocess.DEVNULL).returncode == 0
    assert subprocess.run([built["sr"], "a", "b"], stdout=subprocess.DEVNULL).returncode == 1


@pytest.mark.parametrize("which", ["empty_sam", "empty_gfa"])
def test_degenerate_files_match_reference_binary(built, which):
    """inputs without alignments / without contigs: the product's host glue writes what the compiled reference writes"""
    if not os.path.exists(built["ref_cv"]):
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    from hairsplitter_amd import synth
    with tempfile.TemporaryDirectory() as td:
        f = synth.write_files([synth.make_contig(9, 0, 8000, 2, 0.01, 30, "ont")], td)
        if which == "empty_sam":
            open(f["sam"], "w").write("@HD\tVN:1.6\n")
        else:
            open(f["gfa"], "w").write("H\tVN:Z:1.0\n")
        outs = {}
        for tag, cmd in (("ref", [built["ref_cv"]]), ("hs", [built["harness"], "call_variants"])):
            col, vcf, err = (os.path.join(td, tag + x) for x in (".col", ".vcf", ".err"))
            r = subprocess.run(cmd + [f["gfa"], f["reads"], f["sam"], "1", td, err, "0", "0", col, vcf, "0.33"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            assert r.returncode == 0
            outs[tag] = tuple(open(p, "rb").read() for p in (col, vcf, err))
        assert outs["ref"] == outs["hs"]


def test_host_glue_deep_coverage_equals_oracle(built):
    """coverage > 1000 (low-memory graphs per contig, separate_reads.cpp:1515-1518; finalize on the never-filled matrix,
    :1708): the product's host glue against the oracle restatement on the same files"""
    from hairsplitter_amd import synth, canon
    contigs = [synth.make_contig(77, 0, 4000, 2, 0.01, 1200, "ont", read_len_override=(1000, 2500))]
    with tempfile.TemporaryDirectory() as td:
        f = synth.write_files(contigs, td)
        outs = {}
        for tag, cv, sr in (("hs", [built["harness"], "call_variants"], [built["harness"], "separate_reads"]),
                            ("orc", [built["oracle"], "call_variants"], [built["oracle"], "separate_reads"])):
            col, vcf, err, gro = (os.path.join(td, tag + x) for x in (".col", ".vcf", ".err", ".gro"))
            subprocess.run(cv + [f["gfa"], f["reads"], f["sam"], "2", td, err, "0", "0", col, vcf, "0.33"], check=True, stdout=subprocess.DEVNULL)
            e = min(float(open(err).read().strip()), 0.15)
            subprocess.run(sr + [col, "2", str(e), os.path.join(td, "no_ploidy"), "0", "0.01", "0", gro, "0"], check=True, stdout=subprocess.DEVNULL)
            outs[tag] = (col, vcf, err, gro)
        assert canon.split_blocks(outs["hs"][0]) == canon.split_blocks(outs["orc"][0])
        assert open(outs["hs"][2]).read() == open(outs["orc"][2]).read()
        assert canon.split_blocks(outs["hs"][3]) == canon.split_blocks(outs["orc"][3])


def _sr_through_harness(built, td, meta, col, gro, env):
    kw = meta.get("kwargs", {})
    ploidy = os.path.join(td, "ploidy.txt") if "ploidy_lines" in kw else os.path.join(td, "absent_ploidy.txt")
    r = subprocess.run([built["harness"], "separate_reads", col, "1", meta["error_rate_arg"], ploidy, str(kw.get("low_memory", 0)), "0.01",
                        str(kw.get("amplicon", 0)), gro, "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    return r.stderr.decode()


@pytest.mark.parametrize("case", ["multi", "penta30k", "clips"])
def test_binary_companion_of_the_col_file(built, case):
    """<out.col>.hsbin: stage 3 leaves the .col's content as flat arrays next to it; stage 4 takes them only
    while the .col still is that file, and the .gro is the same either way. (Host code: checked through the harness; the drop-in
    executables go through the same reader / writer.)"""
    if case not in gu.case_names():
        pytest.skip("golden case not present")
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack(case, td)
        outs = gu.run_stage_pair([built["harness"], "call_variants"], [built["harness"], "separate_reads"], td, meta)
        assert gu.compare(td, outs) == []
        col = outs["col"]
        assert os.path.exists(col + ".hsbin") and not os.path.exists(col + ".hsbin.tmp")
        rep = dict(os.environ, HS_SIDECAR_REPORT="1")
        g_bin, g_txt = os.path.join(td, "bin.gro"), os.path.join(td, "txt.gro")
        assert "from the binary companion" in _sr_through_harness(built, td, meta, col, g_bin, rep)
        assert "from the binary companion" not in _sr_through_harness(built, td, meta, col, g_txt, dict(rep, HS_NO_SIDECAR="1"))
        assert open(g_bin).read() == open(g_txt).read() == open(outs["gro"]).read()
        # the rarest-strain filter (separate_reads.cpp:151-167) from the companion's counts == from the text
        r1 = subprocess.run([built["harness"], "separate_reads", col, "1", meta["error_rate_arg"], os.path.join(td, "absent"), "0", "0.3", "0", g_bin + "2", "0"],
                            env=rep, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        r2 = subprocess.run([built["harness"], "separate_reads", col, "1", meta["error_rate_arg"], os.path.join(td, "absent"), "0", "0.3", "0", g_txt + "2", "0"],
                            env=dict(rep, HS_NO_SIDECAR="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r1.returncode == 0 and r2.returncode == 0 and b"from the binary companion" in r1.stderr
        assert open(g_bin + "2").read() == open(g_txt + "2").read()
        # an edited .col (one SNPS line gone: another size; one digit changed: the same size, another block hash) is parsed as text
        lines = open(col).read().split("\n")
        k = next(i for i, l in enumerate(lines) if l.startswith("SNPS"))
        edited = os.path.join(td, "edited.col")
        open(edited, "w").write("\n".join(lines[:k] + lines[k + 1:]))
        os.replace(col + ".hsbin", edited + ".hsbin")
        g_e1, g_e2 = os.path.join(td, "e1.gro"), os.path.join(td, "e2.gro")
        assert "from the binary companion" not in _sr_through_harness(built, td, meta, edited, g_e1, rep)
        _sr_through_harness(built, td, meta, edited, g_e2, dict(rep, HS_NO_SIDECAR="1"))
        assert open(g_e1).read() == open(g_e2).read()
        txt = open(col).read()
        pos = txt.index("SNPS\t") + 5
        flipped = txt[:pos] + ("1" if txt[pos] != "1" else "2") + txt[pos + 1:]
        same_size = os.path.join(td, "same_size.col")
        open(same_size, "w").write(flipped)
        assert len(flipped) == len(txt)
        os.replace(edited + ".hsbin", same_size + ".hsbin")
        assert "from the binary companion" not in _sr_through_harness(built, td, meta, same_size, os.path.join(td, "s.gro"), rep)


@pytest.mark.parametrize("case", ["multi", "penta30k", "linked"])
def test_checker_reads_the_reference_files_as_a_line_parser_does(built, case):
    """oracle/ref_outputs.read_blocks (the reader behind bench.py's parity gate and the full-size pipeline tests) against a plain
    line-by-line parse of the goldens' .col and .gro"""
    import sys
    sys.path.insert(0, os.path.join(gu.ROOT, "oracle"))
    import ref_outputs as ro
    with tempfile.TemporaryDirectory() as td:
        gu.unpack(case, td)
        for fname, tag, nfix in (("variants.col", "SNPS", 3), ("reads_haplo.gro", "GROUP", 2)):
            path = os.path.join(td, fname)
            b = ro.read_blocks(path, tag)
            names, extra, nread, recs = [], [], [], []
            for line in open(path):
                f = line.rstrip("\n").split("\t")
                if f[0] == "CONTIG":
                    names.append(f[1]); extra.append("\t".join(f[2:])); nread.append(0); recs.append([])
                elif f[0] == "READ":
                    nread[-1] += 1
                elif f[0] == tag:
                    nums = [int(x) for x in f[1:1 + nfix]] + [0] * (3 - nfix)
                    recs[-1].append((nums, [int(x) for x in f[1 + nfix].split(",") if x], [int(x) for x in f[2 + nfix].split(",") if x]))
            assert b["names"] == names and b["extra"] == extra and b["n_read_lines"].tolist() == nread
            assert b["rec_off"].tolist() == np.concatenate(([0], np.cumsum([len(r) for r in recs]))).tolist()
            k = 0
            for rs in recs:
                for nums, idx, val in rs:
                    assert [int(b["a"][k]), int(b["b"][k]), int(b["c"][k])] == nums
                    e0, e1 = int(b["ent_off"][k]), int(b["ent_off"][k + 1])
                    assert b["idx"][e0:e1].tolist() == idx and b["val"][e0:e1].tolist() == val
                    k += 1
            assert k == len(b["a"]) and k > 0
