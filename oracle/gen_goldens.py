#!/usr/bin/env python3
"""Generates tests/golden/* from the compiled REFERENCE (oracle/_ref, built by `make -C oracle ref`).

Run in the build container only (it needs oracle/_ref, i.e. /root/reference at build time):
    python oracle/gen_goldens.py [--check-oracle]

For every case it (1) generates build-owned synthetic inputs with hairsplitter_amd.synth (seeded),
(2) runs the reference HS_call_variants (-t 1) and the seed-pinned HS_separate_reads exactly as
hairsplitter.py:668-669,686-692,725-726 would, then the first half of the reference's HS_create_new_contigs on the
resulting .gro (it writes the .gaf before it needs external tools), (3) stores inputs + reference outputs, gzip'ed, under
tests/golden/<case>/. Nothing of the reference's source text is stored: fixtures are data only.

Also written: robin_hood_order.json (iteration orders observed from the reference's vendored header),
shuffle_perms.json (libstdc++ mt19937(12345)+std::shuffle permutations), edlib_vectors.json
(editDistance/endLocation from the reference's bundled edlib).
"""
import argparse
import gzip
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hairsplitter_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")


def py_error_rate_arg(path):
    """hairsplitter.py:686-692,725: float(line) capped at 0.15, re-stringified with str()."""
    with open(path) as f:
        e = float(f.readline().strip())
    if e > 0.15:
        e = 0.15
    return str(e)


def run_ref(workdir, files, low_memory=0, rsa="0.01", amplicon=0, ploidy_lines=None, thr="0.33", tag="", seed=None, fastq=False):
    col = os.path.join(workdir, tag + "variants.col")
    vcf = os.path.join(workdir, tag + "variants.vcf")
    err = os.path.join(workdir, tag + "error_rate.txt")
    gro = os.path.join(workdir, tag + "reads_haplo.gro")
    subprocess.run([os.path.join(REF, "HS_call_variants"), files["gfa"], files["reads"], files["sam"], "1", workdir,
                    err, str(amplicon), "0", col, vcf, thr], check=True, stdout=subprocess.DEVNULL)
    earg = py_error_rate_arg(err)
    ploidy = os.path.join(workdir, "hs_tmp_ploidy_absent.txt")
    if ploidy_lines is not None:
        ploidy = os.path.join(workdir, tag + "ploidy.txt")
        with open(ploidy, "w") as f:
            f.write("".join(ploidy_lines))
    # seed: a second build of the reference with another constant behind std::random_device (oracle/Makefile)
    subprocess.run([os.path.join(REF, "HS_separate_reads_seeded" + (f"_{seed}" if seed else "")), col, "1", earg, ploidy, str(low_memory), rsa,
                    str(amplicon), gro, "0"], check=True, stdout=subprocess.DEVNULL)
    gaf = run_ref_gaf(workdir, files, gro, amplicon, tag)
    return {"col": col, "vcf": vcf, "err": err, "gro": gro, "gaf": gaf, "error_rate_arg": earg,
            "ploidy": ploidy if ploidy_lines is not None else None}


def run_ref_gaf(workdir, files, gro, amplicon=0, tag=""):
    """The .gro consumer of the reference's next stage (create_new_contigs.cpp:1582-1590: parse_split_file, merge_intervals,
    output_GAF). HS_create_new_contigs writes the .gaf and only then shells out to minimap2 / racon / medaka, which do not
    exist here: it is given paths that do not exist, fails there, and the .gaf it wrote before is what is kept."""
    gaf = os.path.join(workdir, tag + "reads_haplo.gaf")
    tmp = os.path.join(workdir, tag + "cnc_tmp")
    os.makedirs(tmp, exist_ok=True)
    if os.path.exists(gaf):
        os.remove(gaf)
    subprocess.run([os.path.join(REF, "HS_create_new_contigs"), files["gfa"], files["reads"], "0.05", gro, files["sam"], tmp + "/", "1",
                    "ont", os.path.join(tmp, "out.gfa"), gaf, "racon", "0", str(amplicon), "/nonexistent/minimap2",
                    "/nonexistent/racon", "/nonexistent/medaka", "/nonexistent/samtools", "/nonexistent/python", "0"],
                   cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert os.path.exists(gaf), "the reference did not get as far as output_GAF"
    shutil.rmtree(tmp, ignore_errors=True)
    return gaf


def store(case, files, outs, meta):
    d = os.path.join(GOLD, case)
    os.makedirs(d, exist_ok=True)
    for k, p in list(files.items()) + [(k, v) for k, v in outs.items() if k in ("col", "vcf", "err", "gro", "gaf", "ploidy") and v]:
        with open(p, "rb") as fi, gzip.GzipFile(os.path.join(d, os.path.basename(p) + ".gz"), "wb", mtime=0) as fo:
            shutil.copyfileobj(fi, fo)
    meta = dict(meta)
    meta["error_rate_arg"] = outs["error_rate_arg"]
    with open(os.path.join(d, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def cases():
    """(name, contigs, sam_extra, run kwargs)"""
    out = []
    out.append(("dip20k", [synth.make_contig(101, 0, 20_000, 2, 0.01, 40, "ont")], None, {}))
    out.append(("penta30k", [synth.make_contig(102, 0, 30_000, 5, 0.01, 60, "ont")], None, {}))
    out.append(("hifi30k", [synth.make_contig(103, 0, 30_000, 2, 0.005, 30, "hifi")], None, {}))
    multi = [synth.make_contig(104, 0, 15_000, 2, 0.01, 40, "ont", name="ctgA"),
             synth.make_contig(104, 1, 10_000, 1, 0.0, 30, "ont", name="ctgB_noSNP"),
             synth.make_contig(104, 2, 8_000, 1, 0.0, 0, "ont", name="ctgC_noreads"),
             synth.make_contig(104, 3, 12_000, 3, 0.015, 45, "ont", name="ctgD")]
    extra = ["ctgA_r0\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*\tLN:i:100",
             "ctgA_r1\t256\tctgD\t100\t0\t50M\t*\t0\t0\t*\t*\tNM:i:0\tLN:i:50",
             "ghost_read\t0\tctgA\t10\t60\t20M\t*\t0\t0\t*\t*\tNM:i:0\tLN:i:20",
             "ghost_read\t0\tctgA\t30\t60\t20M\t*\t0\t0\t*\t*\tNM:i:0\tLN:i:20",
             "ctgA_r2\t0\tctgA\t5\t60\t10M"]
    out.append(("multi", multi, extra, {}))
    out.append(("lowdepth", [synth.make_contig(105, 0, 20_000, 2, 0.01, 14, "ont")], None, {}))
    out.append(("clips", [synth.make_contig(109, 0, 15_000, 2, 0.01, 40, "ont", clip_prob=0.5)], None, {}))
    out.append(("edge_ops", [synth.make_contig(110, 0, 12_000, 2, 0.01, 40, "ont", eqx=True, overhang_prob=0.2, inert_ops_prob=0.3,
                                                 clip_prob=0.2, read_len_override=(800, 4000))], None, {}))
    out.append(("tetra25k_lowmem", [synth.make_contig(106, 0, 25_000, 4, 0.01, 50, "ont")], None, {"low_memory": 1}))
    out.append(("tetra25k_ploidy2", [synth.make_contig(106, 0, 25_000, 4, 0.01, 50, "ont")], None,
                {"ploidy_lines": ["ctg0\t2\n"]}))
    out.append(("dip12k_amplicon", [synth.make_contig(107, 0, 12_000, 2, 0.01, 60, "ont")], None, {"amplicon": 1}))
    out.append(("short_reads_w500", [synth.make_contig(108, 0, 15_000, 2, 0.01, 40, "ont")], None, {}))
    out.append(linked_case())
    # the same inputs as dip20k / penta30k against a reference whose random_device returns 777 (the product takes HS_SEED=777)
    out.append(("penta30k_seed777", [synth.make_contig(102, 0, 30_000, 5, 0.01, 60, "ont")], None, {"seed": 777}))
    # reads as FASTQ (input_output.cpp:41-44,64: everything that is not .fasta / .fa; '@' and '+' may start a quality line)
    out.append(("dip10k_fastq", [synth.make_contig(112, 0, 10_000, 2, 0.01, 40, "ont")], None, {"fastq": True}))
    out.append(simple_mock_case())
    return out


def simple_mock_case():
    """BASELINE config C1: the reference's own test/simple_mock/assembly.gfa (a data file of the reference's test, stored as
    is) with reads simulated at 30x ONT from the three haplotypes of test/simple_mock/mock_reference.fasta (the
    mock_reads.fasta the README names is not in the repository). The assembly's contigs are substitution-only consensus
    pieces of the haplotypes (consensus@0 = [0, 100000), consensus@1 = [100000, 190000), consensus@2 = [189999, 199999);
    consensus_2 matches nothing and gets no reads), so the truth alignment of a read is known without an aligner."""
    import numpy as np
    mock = os.path.join(os.environ.get("HS_REFERENCE", "/root/reference"), "test", "simple_mock")
    code = np.full(256, 3, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    haps, name = [], None
    for line in open(os.path.join(mock, "mock_reference.fasta")):
        if not line.startswith(">"):
            haps.append(code[np.frombuffer(line.strip().encode(), dtype=np.uint8)])
    seqs, links = {}, []
    order = []
    for line in open(os.path.join(mock, "assembly.gfa")):
        f = line.rstrip("\n").split("\t")
        if f[0] == "S":
            seqs[f[1]] = code[np.frombuffer(f[2].encode(), dtype=np.uint8)]
            order.append(f[1])
        else:
            links.append(line.rstrip("\n"))
    place = {"consensus@0": 0, "consensus@1": 100000, "consensus@2": 189999}
    cs = []
    for i, nm in enumerate(order):
        s = seqs[nm]
        if nm in place:
            a = place[nm]
            cs.append(synth.make_contig(113, i, len(s), 3, 0.0, 30, "ont", name=nm, haplotypes=[h[a:a + len(s)] for h in haps], contig_seq=s))
        else:
            cs.append(synth.ContigData(nm, s, [], [], [], np.zeros(0, dtype=np.int32)))
    return ("simple_mock", cs, None, {}, links)


def linked_case():
    """Five contigs joined by 'L' lines, with reads that continue from one contig onto another (supplementary records):
    exercises the graph walk of the .gaf writer (find_paths: direct link, through a short contig, ambiguity, cycle, no
    path), reverse-strand continuation, a contig without SNPs, a read with two records on the same contig."""
    cs = [synth.make_contig(111, 0, 15_000, 2, 0.01, 40, "ont", name="ctgX"),
          synth.make_contig(111, 1, 600, 1, 0.0, 0, "ont", name="ctgY_short"),
          synth.make_contig(111, 2, 12_000, 2, 0.01, 40, "ont", name="ctgZ"),
          synth.make_contig(111, 3, 9_000, 3, 0.015, 45, "ont", name="ctgW"),
          synth.make_contig(111, 4, 6_000, 1, 0.0, 25, "ont", name="ctgV_noSNP")]
    X, Y, Z, W, V = cs
    links = ["L\tctgX\t+\tctgY_short\t+\t0M", "L\tctgY_short\t+\tctgZ\t+\t0M", "L\tctgX\t+\tctgW\t-\t0M",
             "L\tctgZ\t+\tctgV_noSNP\t+\t0M", "L\tctgV_noSNP\t+\tctgX\t+\t0M", "L\tctgW\t+\tctgW\t+\t0M"]
    extra = []

    def supp(src, k, dst, pos, m_len, lead_h, trail_h, reverse=False):
        """record k of contig `src` continues on `dst`: <lead_h>H <m_len>M <trail_h>H at 0-based `pos`"""
        a = src.alns[k]
        n = len(src.reads[a.read])
        assert lead_h + m_len + trail_h == n
        cig = (f"{lead_h}H" if lead_h else "") + f"{m_len}M" + (f"{trail_h}H" if trail_h else "")
        extra.append(f"{src.read_names[a.read]}\t{2064 if reverse else 2048}\t{dst.name}\t{pos + 1}\t60\t{cig}\t*\t0\t0\t*\t*"
                     f"\tNM:i:0\tLN:i:{n}")

    def n_of(src, k):
        return len(src.reads[src.alns[k].read])

    fwd_x = [k for k, a in enumerate(X.alns) if a.strand][:14]
    for k in fwd_x[0:5]:       # X -> (Y) -> Z
        supp(X, k, Z, 0, 300, n_of(X, k) - 300, 0)
    for k in fwd_x[5:8]:       # X -> Y
        supp(X, k, Y, 0, 200, n_of(X, k) - 200, 0)
    for k in fwd_x[8:11]:      # X -> W entered from its right end, read on the reverse strand of W
        supp(X, k, W, len(W.seq) - 250, 250, 0, n_of(X, k) - 250, reverse=True)
    k = fwd_x[11]              # X -> Z -> V, three records
    supp(X, k, Z, 0, 500, n_of(X, k) - 900, 400)
    supp(X, k, V, 0, 400, n_of(X, k) - 400, 0)
    k = fwd_x[12]              # twice on X itself
    supp(X, k, X, 100, 300, n_of(X, k) - 300, 0)
    fwd_z = [k for k, a in enumerate(Z.alns) if a.strand][:4]
    for k in fwd_z[0:2]:       # Z -> V (contig without SNPs)
        supp(Z, k, V, 0, 400, n_of(Z, k) - 400, 0)
    supp(Z, fwd_z[2], X, 0, 300, n_of(Z, fwd_z[2]) - 300, 0)   # Z -> X: only through V (6 kb: too long) -> not chained
    fwd_v = [k for k, a in enumerate(V.alns) if a.strand][:2]
    for k in fwd_v:            # V -> X closes the cycle
        supp(V, k, X, 0, 300, n_of(V, k) - 300, 0)
    rev_x = [k for k, a in enumerate(X.alns) if not a.strand][:2]
    for k in rev_x:            # reverse-strand read of X whose beginning lies on Z: the chain runs Z -> Y -> X backwards
        supp(X, k, Z, len(Z.seq) - 300, 300, 0, n_of(X, k) - 300, reverse=True)
    fwd_w = [k for k, a in enumerate(W.alns) if a.strand][:1]
    supp(W, fwd_w[0], Z, 50, 300, n_of(W, fwd_w[0]) - 300, 0)  # W -> Z: no link
    return ("linked", cs, extra, {}, links)


def gen_lib_vectors():
    rnd = random.Random(7)
    # robin_hood orders
    lines = ["u8 36 37 35 0 1 2", "u8 35 36 37 0 1 2", "u8 50 49 0 1 2", "u8 73 63 76 0 1 2",
             "u8 " + " ".join(map(str, range(33, 45))) + " 0 1 2", "int 0 1 2 3 4 5 6 7 8 9 -1 -2"]
    for _ in range(300):
        n = rnd.choice([1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 20, 30, 60, 125])
        ks = [rnd.randint(33, 157) for _ in range(n)]
        ks += rnd.choices(ks, k=rnd.randint(0, n))
        rnd.shuffle(ks)
        lines.append("u8 " + " ".join(map(str, ks)) + " 0 1 2")
    for _ in range(100):
        n = rnd.choice([1, 2, 5, 8, 13, 30, 100, 400])
        lines.append("int " + " ".join(str(rnd.randint(-2, 1500)) for _ in range(n)))
    res = subprocess.run([os.path.join(REF, "rh_probe")], input="\n".join(lines) + "\n", capture_output=True, text=True,
                         check=True).stdout.splitlines()
    vec = []
    for l, r in zip(lines, res):
        t, *ks = l.split()
        vec.append({"type": t, "keys": list(map(int, ks)), "order": list(map(int, r.split()))})
    with open(os.path.join(GOLD, "robin_hood_order.json"), "w") as f:
        json.dump(vec, f)
    # edlib
    lines = []
    for _ in range(240):
        mode = rnd.choice(["NW", "HW", "SHW"])
        qn = rnd.choice([1, 5, 17, 63, 64, 65, 100, 200, 300, 700])
        base = [rnd.choice("ACGT") for _ in range(qn)]
        t = []
        for c in base:
            u = rnd.random()
            if u < 0.05:
                t.append(rnd.choice("ACGT"))
            elif u < 0.10:
                continue
            elif u < 0.15:
                t.append(c)
                t.append(rnd.choice("ACGT"))
            else:
                t.append(c)
        if mode != "NW":
            t = [rnd.choice("ACGT") for _ in range(rnd.randint(0, 300))] + t + [rnd.choice("ACGT") for _ in range(rnd.randint(0, 300))]
        if not t:
            t = ["A"]
        lines.append(f"{mode} -1 {''.join(base)} {''.join(t)}")
    res = subprocess.run([os.path.join(REF, "edlib_driver")], input="\n".join(lines) + "\n", capture_output=True, text=True,
                         check=True).stdout.splitlines()
    vec = []
    for l, r in zip(lines, res):
        mode, k, q, t = l.split()
        d, nloc, sloc, eloc = map(int, r.split())
        vec.append({"mode": mode, "query": q, "target": t, "distance": d, "end": eloc})
    with open(os.path.join(GOLD, "edlib_vectors.json"), "w") as f:
        json.dump(vec, f)
    # shuffle permutations: produced by the reference itself is not observable directly; they are pinned through
    # the .gro goldens. The libstdc++ permutations below come from the oracle binary (system library, not reference code).


def gen_edlib_edge_vectors():
    """Distance + first end location where no column of the target beats the all-insertions score, for queries with and without
    padding rows in edlib's last block (multiples of 64 behave differently): tests/golden/edlib_edge_vectors.json"""
    rnd = random.Random(41)
    rs = lambda n, al="ACGT": "".join(rnd.choice(al) for _ in range(n))
    pairs = [("A" * 64, "C" * 100), ("A" * 128, "C" * 10), ("A" * 65, "C" * 100), ("AC" * 32, "G"), ("ACGT" * 16, "ACGT" * 16 + "T"), ("A" * 64, "A" * 10),
             ("A" * 63, "C" * 9), ("A" * 192, "CGT" * 50), ("A" * 256, "C"), ("AAAA", "CCCC"), ("A" * 64, "C" * 63 + "A")]
    for qn in (64, 128, 192, 63, 65, 127):
        for _ in range(4):
            pairs.append((rs(qn, "AC"), rs(rnd.randint(1, 150), "GT")))      # disjoint alphabets: nothing matches
            pairs.append((rs(qn), rs(rnd.randint(1, 40))))                   # target much shorter than the query
    lines = ["%s -1 %s %s" % (m, q, t) for q, t in pairs for m in ("NW", "SHW", "HW")]
    res = subprocess.run([os.path.join(REF, "edlib_driver")], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.splitlines()
    vec = []
    for l, r in zip(lines, res):
        mode, k, q, t = l.split()
        d, nloc, sloc, eloc = map(int, r.split())
        vec.append({"mode": mode, "query": q, "target": t, "distance": d, "end": eloc})
    with open(os.path.join(GOLD, "edlib_edge_vectors.json"), "w") as f:
        json.dump(vec, f)
    print("edlib edge vectors:", len(vec))


def gen_stage5_alphabet_cases():
    """Stage-5 call-site cases over other alphabets than ACGT (edlib compares bytes: lower case differs from upper case, N only
    matches N): tests/golden/stage5_alphabet_cases.json. Every query / target pair has at most four distinct bytes -- what the
    four-code kernel can represent exactly -- except the last case, which the library must refuse."""
    rnd = random.Random(31)

    def rs(n, alpha):
        return "".join(rnd.choice(alpha) for _ in range(n))

    def mutate(s, rate, alpha):
        out = []
        for c in s:
            u = rnd.random()
            if u < rate / 3:
                out.append(rnd.choice(alpha))
            elif u < 2 * rate / 3:
                continue
            elif u < rate:
                out.append(c); out.append(rnd.choice(alpha))
            else:
                out.append(c)
        return "".join(out)

    lines, cases = [], []
    for alpha in ("acgt", "ACGN", "ANTG", "Acgt", "AaCc", "NT", "ACG"):
        for _ in range(6):
            b = rs(rnd.randint(250, 1500), alpha)
            cut_l, cut_r = rnd.randint(0, 60), rnd.randint(0, 60)
            c = mutate(b[cut_l:len(b) - cut_r], rnd.choice([0.0, 0.02, 0.05]), alpha)
            lines.append("REATTACH 0 %s %s" % (b, c)); cases.append({"kind": "reattach", "backbone": b, "consensus": c})
        for _ in range(6):
            ol, orr = rnd.choice([0, 50, 150]), rnd.choice([0, 50, 150])
            tp = rs(ol, alpha) + rs(rnd.randint(400, 1500), alpha) + rs(orr, alpha)
            nc = mutate(tp, rnd.choice([0.0, 0.02, 0.05]), alpha)
            lines.append("TRIM %d,%d %s %s" % (ol, orr, tp, nc)); cases.append({"kind": "trim", "to_polish": tp, "newcontig": nc, "overhang_left": ol, "overhang_right": orr})
    res = subprocess.run([os.path.join(REF, "edlib_driver")], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.split("\n")
    for c, r in zip(cases, res):
        c["expected"] = r
    cases.append({"kind": "refused", "backbone": "NACGT" * 80, "consensus": "NACGT" * 76})      # five distinct bytes in one pair
    with open(os.path.join(GOLD, "stage5_alphabet_cases.json"), "w") as f:
        json.dump(cases, f)
    print("stage-5 alphabet cases:", len(cases))


def gen_edlib_path_vectors():
    """HW + PATH vectors from the reference's bundled edlib at the shapes of its stage-5 call sites (create_new_contigs.cpp:558-629,
    tools.cpp:515-534: a 200-300 bp query inside a target of a few hundred to a few thousand bases) plus edge cases: written to
    tests/golden/edlib_path_vectors.json"""
    rnd = random.Random(23)
    rs = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))

    def mutate(s, rate):
        out = []
        for c in s:
            u = rnd.random()
            if u < rate / 3:
                out.append(rnd.choice("ACGT"))
            elif u < 2 * rate / 3:
                continue
            elif u < rate:
                out.append(c); out.append(rnd.choice("ACGT"))
            else:
                out.append(c)
        return "".join(out)

    pairs = []
    for _ in range(120):      # stage-5 shapes: the end of a contig inside its polished version
        q = rs(rnd.choice([200, 250, 300, 300, 300]))
        t = rs(rnd.randint(0, 2500)) + mutate(q, rnd.choice([0.0, 0.02, 0.05, 0.12, 0.3])) + rs(rnd.randint(0, 800))
        pairs.append((q, t))
    for _ in range(40):       # tools.cpp:515-534: 200 bp of the consensus inside 300 bp of the backbone
        t = rs(300)
        a = rnd.randint(0, 100)
        pairs.append((mutate(t[a:a + 200], rnd.choice([0.0, 0.03, 0.1])), t))
    for _ in range(60):       # small and odd cases
        qn = rnd.choice([1, 2, 5, 17, 63, 64, 65, 127, 128, 129, 200])
        q = rs(qn)
        kind = rnd.randint(0, 5)
        if kind == 0:
            t = rs(rnd.randint(1, 90))                       # unrelated
        elif kind == 1:
            t = q                                            # identical
        elif kind == 2:
            t = q[:max(1, qn // 2)]                          # target shorter than the query
        elif kind == 3:
            t = rs(rnd.randint(0, 30)) + q + rs(rnd.randint(0, 30))
        elif kind == 4:
            t = (q * 3)[:rnd.randint(qn, 3 * qn)]            # repeats: several optimal placements
        else:
            t = mutate(q, 0.4) or "A"
        pairs.append((q, t))
    # no column beats the all-insertions score: with a query of k * 64 bases (no padding rows in edlib's last block) the first
    # column is the end location, otherwise the location "before the target" (-1)
    pairs += [("A" * 64, "C" * 100), ("A" * 128, "C" * 10), ("A" * 65, "C" * 100), ("A" * 63, "C" * 7), ("AC" * 32, "G"), ("A" * 192, "CCGT" * 90)]
    pairs += [("CCTT", "AAGG"), ("A", "C"), ("ACGT", "ACGT"), ("AAAA", "AAAAAAAAAAAA"), ("ACGTACGT", "TTTTACGTACGTTTTT"), ("G", "G"), ("-", "ACGT"), ("ACGT", "-")]
    lines = ["HWPATH -1 %s %s" % (q, t) for q, t in pairs]
    res = subprocess.run([os.path.join(REF, "edlib_driver")], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.splitlines()
    vec = []
    for (q, t), r in zip(pairs, res):
        d, st, en, cig = r.split()
        vec.append({"query": "" if q == "-" else q, "target": "" if t == "-" else t, "distance": int(d), "start": int(st), "end": int(en), "cigar": cig})
    with open(os.path.join(GOLD, "edlib_path_vectors.json"), "w") as f:
        json.dump(vec, f)
    print("edlib path vectors:", len(vec))
    # the two stage-5 computations around those calls (oracle/edlib_driver.cpp restates them on the reference's edlib)
    lines, cases = [], []
    for _ in range(40):      # tools.cpp:505-536: racon dropped some bases at both ends of the backbone
        b = rs(rnd.randint(250, 2500))
        cut_l, cut_r = rnd.randint(0, 60), rnd.randint(0, 60)
        c = mutate(b[cut_l:len(b) - cut_r], rnd.choice([0.0, 0.01, 0.04]))
        if len(c) < 10:
            continue
        lines.append("REATTACH 0 %s %s" % (b, c)); cases.append({"kind": "reattach", "backbone": b, "consensus": c})
    for _ in range(40):      # create_new_contigs.cpp:556-629: the piece was polished together with its overhangs
        ol, orr = rnd.choice([0, 50, 150, 400]), rnd.choice([0, 50, 150, 400])
        core = rs(rnd.randint(400, 2500))
        tp = rs(ol) + core + rs(orr)
        nc = mutate(tp, rnd.choice([0.0, 0.01, 0.04])) if rnd.random() < 0.85 else rs(len(tp))      # sometimes "reassembled": aligns badly
        lines.append("TRIM %d,%d %s %s" % (ol, orr, tp, nc)); cases.append({"kind": "trim", "to_polish": tp, "newcontig": nc, "overhang_left": ol, "overhang_right": orr})
    res = subprocess.run([os.path.join(REF, "edlib_driver")], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.split("\n")
    for c, r in zip(cases, res):
        c["expected"] = r
    with open(os.path.join(GOLD, "stage5_edlib_cases.json"), "w") as f:
        json.dump(cases, f)
    print("stage-5 call-site cases:", len(cases))


def gen_c5u():
    """The uncut stress variant of BASELINE C5: the reference needs about 12 minutes for it (one contig = one
    thread), so its OUTPUTS are stored (tests/golden_big/c5u, ~1.2 MB) and the inputs are regenerated from the seed by the
    test (tests/test_gpu_full_configs.py::test_c5_uncut_10mb_equals_reference_outputs), checked by their sha256."""
    import hashlib
    import time
    d = os.path.join(ROOT, "tests", "golden_big", "c5u")
    os.makedirs(d, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        files = synth.write_files(synth.config_contigs("C5U"), td)
        t0 = time.time()
        col, vcf, err, gro = (os.path.join(td, x) for x in ("variants.col", "variants.vcf", "error_rate.txt", "reads_haplo.gro"))
        subprocess.run([os.path.join(REF, "HS_call_variants"), files["gfa"], files["reads"], files["sam"], "1", td, err, "0", "0", col, vcf, "0.33"],
                       check=True, stdout=subprocess.DEVNULL)
        t1 = time.time()
        earg = py_error_rate_arg(err)
        subprocess.run([os.path.join(REF, "HS_separate_reads_seeded"), col, "1", earg, os.path.join(td, "absent"), "0", "0.01", "0", gro, "0"],
                       check=True, stdout=subprocess.DEVNULL)
        t2 = time.time()
        for p in (col, vcf, gro):
            with open(p, "rb") as fi, gzip.GzipFile(os.path.join(d, os.path.basename(p) + ".gz"), "wb", mtime=0) as fo:
                shutil.copyfileobj(fi, fo)
        shutil.copyfile(err, os.path.join(d, "error_rate.txt"))
        meta = {"case": "c5u", "error_rate_arg": earg, "n_snps": sum(1 for l in open(col) if l.startswith("SNPS")),
                "n_groups": sum(1 for l in open(gro) if l.startswith("GROUP")),
                "input_sha256": {os.path.basename(p): hashlib.sha256(open(p, "rb").read()).hexdigest() for p in files.values()},
                "reference_wall_s": {"HS_call_variants": round(t1 - t0, 1), "HS_separate_reads_seeded": round(t2 - t1, 1)},
                "generated_by": "python oracle/gen_goldens.py --c5u"}
        with open(os.path.join(d, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
    print("c5u:", meta["n_snps"], "SNPs,", meta["n_groups"], "windows")


def gen_edlib_long_path_vectors():
    """HW + PATH vectors from the reference's bundled edlib for queries of 1.5-60 kb: beyond 1 MB of its own bookkeeping edlib cuts
    the alignment in halves (Hirschberg, edlib.cpp:1166-1404), and which of the equally good alignments it returns depends on the
    cuts. Read-like errors, long insertions and deletions (a half that is all deletions: the boundary rows of :1335-1353),
    repeats (many optimal alignments), unrelated sequences: tests/golden/edlib_long_path_vectors.json.gz"""
    import gzip
    rnd = random.Random(31)
    rs = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))

    def mutate(s, rate):
        out = []
        for c in s:
            u = rnd.random()
            if u < rate / 3:
                out.append(rnd.choice("ACGT"))
            elif u < 2 * rate / 3:
                continue
            elif u < rate:
                out.append(c); out.append(rnd.choice("ACGT"))
            else:
                out.append(c)
        return "".join(out)

    pairs = []
    for qn in (1500, 1830, 1900, 2047, 2048, 3000, 4096, 4097, 5000, 8200, 12000):
        q = rs(qn)
        pairs.append((q, rs(rnd.randint(0, 1500)) + mutate(q, rnd.choice([0.0, 0.03, 0.12])) + rs(rnd.randint(0, 800))))
    for qn in (2500, 4500, 7000):
        q = rs(qn)
        pairs.append((q, mutate(q, 0.3)))                                        # a wide band
        unit = rs(rnd.randint(1, 7))
        r = (unit * (qn // len(unit) + 1))[:qn]
        pairs.append((r, mutate(r, 0.05) + rs(100)))                             # a repeat: many optimal alignments
        pairs.append((rs(qn), rs(qn // 3)))                                      # unrelated, target shorter than the query
        q = rs(qn); pairs.append((q, q[:qn // 2] + q[qn // 2 + 700:]))           # 700 query bases without a partner
        q = rs(qn); pairs.append((q, q[:qn // 3] + rs(900) + q[qn // 3:]))       # 900 target bases without a partner
    for _ in range(8):                                                           # several long gaps anywhere
        q = rs(rnd.randint(2000, 7000))
        t = q
        for _ in range(rnd.randint(1, 3)):
            p = rnd.randint(0, len(t))
            t = t[:p] + rs(rnd.randint(500, 5000)) + t[p:] if rnd.random() < 0.5 else t[:p] + t[p + rnd.randint(500, 3000):]
        pairs.append((q, t or "A"))
    q = rs(300); pairs.append((q, rs(4000) + mutate(q, 0.05) + rs(8000)))        # the stage-5 shape on a long target (no cut: 300 rows)
    q = rs(30000); pairs.append((q, rs(500) + mutate(q, 0.08) + rs(500)))
    q = rs(60000); pairs.append((q, mutate(q, 0.05)))
    lines = ["HWPATH -1 %s %s" % (q, t) for q, t in pairs]
    res = subprocess.run([os.path.join(REF, "edlib_driver")], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.splitlines()
    vec = []
    for (q, t), r in zip(pairs, res):
        d, st, en, cig = r.split()
        vec.append({"query": q, "target": t, "distance": int(d), "start": int(st), "end": int(en), "cigar": cig})
    with gzip.GzipFile(os.path.join(GOLD, "edlib_long_path_vectors.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(vec).encode())
    print("edlib long path vectors:", len(vec), "longest query", max(len(v["query"]) for v in vec))


def gen_edlib_mid_path_vectors():
    """HW + PATH vectors for queries of 330-2048 bases whose matrix edlib keeps whole (one leaf): the sizes at which
    k_myers_hw_path_grouped takes 8, 16 or 32 lanes per pair, mixed so that one call holds every class:
    tests/golden/edlib_mid_path_vectors.json.gz"""
    import gzip
    rnd = random.Random(47)
    rs = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))

    def mutate(s, rate):
        out = []
        for c in s:
            u = rnd.random()
            if u < rate / 3:
                out.append(rnd.choice("ACGT"))
            elif u < 2 * rate / 3:
                continue
            elif u < rate:
                out.append(c); out.append(rnd.choice("ACGT"))
            else:
                out.append(c)
        return "".join(out)

    pairs = []
    for qn in (330, 448, 511, 512, 513, 600, 777, 1000, 1023, 1024, 1025, 1300, 1536, 1800, 2047, 2048):
        nb = (qn + 63) // 64
        tmax = (1024 * 1024 - 1) // (20 * nb + 8)          # edlib.cpp:1192-1196
        for kind in range(4):
            q = rs(qn)
            body = mutate(q, [0.0, 0.04, 0.15, 0.4][kind])
            room = max(0, tmax - len(body) - 1)
            left = rnd.randint(0, room) if kind != 2 else 0
            t = (rs(left) + body + rs(rnd.randint(0, room - left)))[:tmax] or "A"
            pairs.append((q, t))
        pairs.append((rs(qn), rs(rnd.randint(1, min(tmax, qn // 2)))))       # unrelated, target shorter than the query
        unit = rs(rnd.randint(1, 5))
        r = (unit * (qn // len(unit) + 1))[:qn]
        pairs.append((r, (mutate(r, 0.05) + rs(50))[:tmax]))                  # repeats
    rnd.shuffle(pairs)
    lines = ["HWPATH -1 %s %s" % (q, t) for q, t in pairs]
    res = subprocess.run([os.path.join(REF, "edlib_driver")], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.splitlines()
    vec = []
    for (q, t), r in zip(pairs, res):
        d, st, en, cig = r.split()
        vec.append({"query": q, "target": t, "distance": int(d), "start": int(st), "end": int(en), "cigar": cig})
    with gzip.GzipFile(os.path.join(GOLD, "edlib_mid_path_vectors.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(vec).encode())
    print("edlib mid path vectors:", len(vec))


def gen_edlib_long_dist_vectors():
    """NW / SHW / HW distance + first end location from the reference's bundled edlib for the pairs of edlib_long_path_vectors and
    edlib_mid_path_vectors (0.3-60 kb): what hs_edit_distance returns. Stored by pair index: tests/golden/edlib_long_dist_vectors.json"""
    import gzip
    out = {}
    for name in ("edlib_long_path_vectors", "edlib_mid_path_vectors"):
        vec = json.loads(gzip.open(os.path.join(GOLD, name + ".json.gz")).read())
        lines = ["%s -1 %s %s" % (mode, v["query"], v["target"]) for v in vec for mode in ("NW", "SHW", "HW")]
        res = subprocess.run([os.path.join(REF, "edlib_driver")], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.splitlines()
        rows = []
        for i in range(len(vec)):
            row = {}
            for m, mode in enumerate(("NW", "SHW", "HW")):
                d, nloc, st, en = res[3 * i + m].split()
                row[mode] = [int(d), int(en)]
            rows.append(row)
        out[name] = rows
    with open(os.path.join(GOLD, "edlib_long_dist_vectors.json"), "w") as f:
        json.dump(out, f)
    print("edlib long distance vectors:", {k: len(v) for k, v in out.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-oracle", action="store_true", help="also run oracle/_build/hs_oracle and report parity")
    ap.add_argument("--only", default=None)
    ap.add_argument("--edlib-path", action="store_true", help="only tests/golden/edlib_path_vectors.json")
    ap.add_argument("--edlib-long", action="store_true", help="only tests/golden/edlib_long_path_vectors.json.gz")
    ap.add_argument("--edlib-long-dist", action="store_true", help="only tests/golden/edlib_long_dist_vectors.json")
    ap.add_argument("--edlib-mid", action="store_true", help="only tests/golden/edlib_mid_path_vectors.json.gz")
    ap.add_argument("--edlib-edge", action="store_true", help="only tests/golden/edlib_edge_vectors.json")
    ap.add_argument("--stage5-alphabet", action="store_true", help="only tests/golden/stage5_alphabet_cases.json")
    ap.add_argument("--c5u", action="store_true", help="only the uncut 10 Mb variant of C5: outputs into tests/golden_big/c5u (12 minutes of the reference)")
    args = ap.parse_args()
    if args.c5u:
        return gen_c5u()
    if args.edlib_path:
        return gen_edlib_path_vectors()
    if args.edlib_long:
        return gen_edlib_long_path_vectors()
    if args.edlib_mid:
        return gen_edlib_mid_path_vectors()
    if args.edlib_long_dist:
        return gen_edlib_long_dist_vectors()
    if args.stage5_alphabet:
        return gen_stage5_alphabet_cases()
    if args.edlib_edge:
        return gen_edlib_edge_vectors()
    os.makedirs(GOLD, exist_ok=True)
    if not args.only:
        gen_lib_vectors()
    from hairsplitter_amd import canon
    for case in cases():
        name, contigs, extra, kw = case[:4]
        gfa_extra = case[4] if len(case) > 4 else None
        if args.only and args.only != name:
            continue
        with tempfile.TemporaryDirectory() as td:
            if name == "short_reads_w500":
                # force the 500-bp window branch (separate_reads.cpp:1489): cut every read alignment to <= 1.5 kb
                contigs = [synth.make_contig(108, 0, 15_000, 2, 0.01, 40, "ont", read_len_override=(800, 1500))]
            files = synth.write_files(contigs, td, sam_extra=extra, gfa_extra=gfa_extra, fastq=bool(kw.get("fastq")))
            if name == "simple_mock":   # the reference's data file as it is (its S lines carry DP:f: tags and a trailing tab)
                shutil.copyfile("/root/reference/test/simple_mock/assembly.gfa", files["gfa"])
            outs = run_ref(td, files, **kw)
            meta = {"case": name, "kwargs": {k: v for k, v in kw.items()},
                    "aligned_bp": int(sum(c.aligned_bp for c in contigs)),
                    "n_snps": sum(1 for l in open(outs["col"]) if l.startswith("SNPS")),
                    "n_groups": sum(1 for l in open(outs["gro"]) if l.startswith("GROUP"))}
            store(name, files, outs, meta)
            msg = f"{name}: {meta['aligned_bp']} bp, {meta['n_snps']} SNPs, {meta['n_groups']} windows"
            if args.check_oracle:
                orc = os.path.join(ROOT, "oracle", "_build", "hs_oracle")
                ocol, ovcf, oerr, ogro = (os.path.join(td, "o_" + x) for x in ("v.col", "v.vcf", "e.txt", "r.gro"))
                subprocess.run([orc, "call_variants", files["gfa"], files["reads"], files["sam"], "1", td, oerr,
                                str(kw.get("amplicon", 0)), "0", ocol, ovcf, "0.33"], check=True, stdout=subprocess.DEVNULL)
                subprocess.run([orc, "separate_reads", outs["col"], "1", outs["error_rate_arg"],
                                outs["ploidy"] or os.path.join(td, "absent"), str(kw.get("low_memory", 0)), "0.01",
                                str(kw.get("amplicon", 0)), ogro, "0"], check=True, stdout=subprocess.DEVNULL,
                               env=dict(os.environ, HS_ORACLE_SEED=str(kw["seed"])) if kw.get("seed") else None)
                ok_col = canon.split_blocks(ocol) == canon.split_blocks(outs["col"])
                ok_vcf = canon.vcf_blocks(ovcf) == canon.vcf_blocks(outs["vcf"])
                ok_err = open(oerr).read() == open(outs["err"]).read()
                ok_gro = canon.split_blocks(ogro) == canon.split_blocks(outs["gro"])
                msg += f" | oracle col={ok_col} vcf={ok_vcf} err={ok_err} gro={ok_gro}"
                if not ok_gro:
                    msg += " " + "; ".join(canon.diff_blocks(canon.split_blocks(ogro), canon.split_blocks(outs["gro"])))
                if not ok_col:
                    msg += " " + "; ".join(canon.diff_blocks(canon.split_blocks(ocol), canon.split_blocks(outs["col"])))
            print(msg, flush=True)


if __name__ == "__main__":
    main()
