"""Shared helpers for the golden-fixture tests: unpack a case, run a pair of stage binaries the way
hairsplitter.py does (hairsplitter.py:668-669,725-726), and compare canonically with the reference outputs."""
import gzip
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hairsplitter_amd import canon  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def case_names():
    return sorted(d for d in os.listdir(GOLD) if os.path.isdir(os.path.join(GOLD, d)))


def unpack(case, dst):
    src = os.path.join(GOLD, case)
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(src):
        if f.endswith(".gz"):
            with gzip.open(os.path.join(src, f), "rb") as fi, open(os.path.join(dst, f[:-3]), "wb") as fo:
                shutil.copyfileobj(fi, fo)
    with open(os.path.join(src, "meta.json")) as f:
        return json.load(f)


def run_stage_pair(cv_cmd, sr_cmd, workdir, meta, tag="t_", env=None, use_golden_col=False):
    """cv_cmd / sr_cmd: argv prefixes (e.g. [binary] or [binary, 'call_variants'])."""
    kw = meta.get("kwargs", {})
    amplicon = str(kw.get("amplicon", 0))
    col, vcf, err, gro = (os.path.join(workdir, tag + n) for n in ("variants.col", "variants.vcf", "error_rate.txt", "reads_haplo.gro"))
    r = subprocess.run(cv_cmd + [os.path.join(workdir, "assembly.gfa"), os.path.join(workdir, "reads.fasta"),
                                 os.path.join(workdir, "aln.sam"), "1", workdir, err, amplicon, "0", col, vcf, "0.33"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    ploidy = os.path.join(workdir, "ploidy.txt") if "ploidy_lines" in kw else os.path.join(workdir, "absent_ploidy.txt")
    col_in = os.path.join(workdir, "variants.col") if use_golden_col else col
    r = subprocess.run(sr_cmd + [col_in, "1", meta["error_rate_arg"], ploidy, str(kw.get("low_memory", 0)), "0.01", amplicon, gro, "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    return {"col": col, "vcf": vcf, "err": err, "gro": gro}


def compare(workdir, outs):
    """Returns a list of human-readable mismatches (empty == parity)."""
    bad = []
    a, b = canon.split_blocks(outs["col"]), canon.split_blocks(os.path.join(workdir, "variants.col"))
    if a != b:
        bad.append("col: " + "; ".join(canon.diff_blocks(a, b)))
    a, b = canon.vcf_blocks(outs["vcf"]), canon.vcf_blocks(os.path.join(workdir, "variants.vcf"))
    if a != b:
        bad.append("vcf differs")
    if open(outs["err"]).read() != open(os.path.join(workdir, "error_rate.txt")).read():
        bad.append("error_rate: %r != %r" % (open(outs["err"]).read(), open(os.path.join(workdir, "error_rate.txt")).read()))
    a, b = canon.split_blocks(outs["gro"]), canon.split_blocks(os.path.join(workdir, "reads_haplo.gro"))
    if a != b:
        bad.append("gro: " + "; ".join(canon.diff_blocks(a, b)))
    return bad
