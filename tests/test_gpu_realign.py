"""GPU parity of the CIGAR-less input (hs_realign.cpp): a PAF file -> SAM through the device aligner (A1).
Checked two ways: (1) every aligned record against the reference's own bundled edlib (oracle/_ref/edlib_driver) on the same
(read segment, contig window): start, distance and the path move by move; (2) the drop-in executables on the .paf against the
compiled reference on the SAM the mode wrote (same .col / .vcf / error rate / .gro)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PAD = 100
OPS = re.compile(r"(\d+)([MIDNSHP=X])")


def _sam_to_paf(sam, paf, clip=0):
    """The PAF a mapper would print beside these SAM records (the truth alignment's intervals); clip > 0 narrows every read
    interval by that many bases on both sides (soft clips in the SAM the mode writes)"""
    n = 0
    with open(sam) as f, open(paf, "w") as o:
        for l in f:
            if l.startswith("@"):
                continue
            t = l.rstrip("\n").split("\t")
            qlen = int([x for x in t if x.startswith("LN:i:")][0][5:])
            ops = [(int(a), b) for a, b in OPS.findall(t[5])]
            ls = ops[0][0] if ops[0][1] in "SH" else 0
            rs = ops[-1][0] if ops[-1][1] in "SH" else 0
            refspan = sum(a for a, b in ops if b in "MD=X")
            minus = int(t[1]) & 16
            qs, qe = (rs, qlen - ls) if minus else (ls, qlen - rs)
            ts = int(t[3]) - 1
            te = ts + refspan
            if clip and qe - qs > 4 * clip and te - ts > 4 * clip:
                qs += clip; qe -= clip; ts += clip; te -= clip
            o.write("\t".join(map(str, [t[0], qlen, qs, qe, "-" if minus else "+", t[2], 0, ts, te, 0, te - ts, 60])) + "\n")
            n += 1
    return n


def _realign(built, gfa, reads, paf, out_sam):
    from hairsplitter_amd import api
    lib = api.load()

    class Stats(C.Structure):
        _fields_ = [("n_lines", C.c_int64), ("n_aligned", C.c_int64), ("query_bases", C.c_int64), ("ms_device", C.c_double), ("ms_total", C.c_double)]
    st = Stats()
    rc = lib.hs_realign_paf(gfa.encode(), reads.encode(), paf.encode(), out_sam.encode(), C.c_int32(4), C.byref(st))
    assert rc == 0, lib.hs_last_error()
    return st


def _seqs(gfa, reads):
    ctg, rd = {}, {}
    for l in open(gfa):
        if l.startswith("S\t"):
            t = l.rstrip("\n").split("\t"); ctg[t[1]] = t[2]
    name = None
    for l in open(reads):
        if l.startswith(">"):
            name = l[1:].split()[0]
        elif name is not None:
            rd[name] = l.strip(); name = None
    return ctg, rd


def _revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


@pytest.mark.parametrize("clip", [0, 25])
def test_realigned_records_equal_the_reference_edlib(built, clip):
    from hairsplitter_amd import synth
    with tempfile.TemporaryDirectory() as td:
        c = [synth.make_contig(41, 0, 14_000, 2, 0.01, 12, "ont"), synth.make_contig(41, 1, 9_000, 3, 0.01, 10, "ont")]
        f = synth.write_files(c, td)
        paf, sam = os.path.join(td, "aln.paf"), os.path.join(td, "re.sam")
        n = _sam_to_paf(f["sam"], paf, clip)
        st = _realign(built, f["gfa"], f["reads"], paf, sam)
        assert st.n_lines == n and st.n_aligned == n
        ctg, rd = _seqs(f["gfa"], f["reads"])
        pafs = [l.split("\t") for l in open(paf)]
        sams = [l.rstrip("\n").split("\t") for l in open(sam) if not l.startswith("@")]
        assert len(sams) == n
        lines = []
        for p in pafs:
            qs, qe, ts, te = int(p[2]), int(p[3]), int(p[7]), int(p[8])
            seg = rd[p[0]][qs:qe]
            if p[4] == "-":
                seg = _revcomp(seg)
            w0, w1 = max(0, ts - PAD), min(len(ctg[p[5]]), te + PAD)
            lines.append("HWPATH -1 %s %s" % (seg, ctg[p[5]][w0:w1]))
        ref = subprocess.run([os.path.join(os.path.dirname(built["ref_cv"]), "edlib_driver")], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.splitlines()
        assert len(ref) == n
        for p, s, r in zip(pafs, sams, ref):
            dist, start, end, ext = r.split()
            qlen, qs, qe, ts = int(p[1]), int(p[2]), int(p[3]), int(p[7])
            w0 = max(0, ts - PAD)
            assert s[0] == p[0] and s[2] == p[5] and int(s[1]) == (16 if p[4] == "-" else 0)
            assert int(s[3]) == w0 + int(start) + 1, (s[:6], r[:80])
            assert ("NM:i:" + dist) in s and ("LN:i:%d" % qlen) in s
            # edlib's extended CIGAR (= X I D) as the M / I / D runs of a SAM, with the clips of the read interval
            want, cur, run = [], None, 0
            for a, b in OPS.findall(ext):
                b = "M" if b in "=X" else b
                if b == cur:
                    run += int(a)
                else:
                    if cur: want.append("%d%s" % (run, cur))
                    cur, run = b, int(a)
            if cur: want.append("%d%s" % (run, cur))
            cl, cr = (qlen - qe, qs) if p[4] == "-" else (qs, qlen - qe)
            want = (["%dS" % cl] if cl else []) + want + (["%dS" % cr] if cr else [])
            assert s[5] == "".join(want), (p[0], s[5][:60], "".join(want)[:60])


def test_dropin_on_a_paf_equals_the_reference_on_the_sam_it_made(built):
    from hairsplitter_amd import synth, canon
    with tempfile.TemporaryDirectory() as td:
        c = [synth.make_contig(43, 0, 16_000, 2, 0.012, 30, "ont"), synth.make_contig(43, 1, 12_000, 3, 0.012, 32, "ont")]
        f = synth.write_files(c, td)
        paf = os.path.join(td, "aln.paf")
        _sam_to_paf(f["sam"], paf, 0)
        col, vcf, err, gro = (os.path.join(td, "hip." + x) for x in ("col", "vcf", "err", "gro"))
        cv = [built["cv"], f["gfa"], f["reads"], paf, "4", td, err, "0", "0", col, vcf, "0.33"]
        # without the opt-in the reference's refusal stands (call_variants.cpp:1256-1259)
        r = subprocess.run(cv, stdout=subprocess.PIPE, env={k: v for k, v in os.environ.items() if k != "HS_REALIGN"})
        assert r.returncode != 0 and b"please provide a .sam file" in r.stdout
        subprocess.run(cv, check=True, stdout=subprocess.DEVNULL, env=dict(os.environ, HS_REALIGN="1"))
        sam = os.path.join(td, "hs_realigned.sam")
        assert os.path.exists(sam)
        rcol, rvcf, rerr, rgro = (os.path.join(td, "ref." + x) for x in ("col", "vcf", "err", "gro"))
        subprocess.run([built["ref_cv"], f["gfa"], f["reads"], sam, "1", td, rerr, "0", "0", rcol, rvcf, "0.33"], check=True, stdout=subprocess.DEVNULL)
        assert canon.split_blocks(col) == canon.split_blocks(rcol)
        assert open(err).read() == open(rerr).read()
        assert canon.vcf_blocks(vcf) == canon.vcf_blocks(rvcf)
        e = str(min(float("%g" % float(open(err).read().strip())), 0.15))
        subprocess.run([built["sr"], col, "4", e, os.path.join(td, "none"), "0", "0.01", "0", gro, "0"], check=True, stdout=subprocess.DEVNULL)
        subprocess.run([built["ref_sr_seeded"], rcol, "1", e, os.path.join(td, "none"), "0", "0.01", "0", rgro, "0"], check=True, stdout=subprocess.DEVNULL)
        assert canon.split_blocks(gro) == canon.split_blocks(rgro)
        # the truth alignment itself gives nearly the same calls: the aligner's paths differ from the simulator's only where edits are equivalent
        tcol = os.path.join(td, "truth.col")
        subprocess.run([built["cv"], f["gfa"], f["reads"], f["sam"], "4", td, os.path.join(td, "t.err"), "0", "0", tcol, os.path.join(td, "t.vcf"), "0.33"], check=True, stdout=subprocess.DEVNULL)
        snps = lambda p: {(b.split("\n")[0].split("\t")[1], l.split("\t")[1]) for b in open(p).read().split("CONTIG")[1:] for l in b.split("\n") if l.startswith("SNPS")}
        a, b = snps(col), snps(tcol)
        assert len(a & b) >= 0.9 * max(1, len(b)), (len(a), len(b), len(a & b))
