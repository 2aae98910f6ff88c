#!/usr/bin/env python3
"""Fixtures of the two GFA text tools (tests/golden/gfa_tools): inputs made here, outputs of the REFERENCE itself --
src/cut_gfa.py run with this interpreter, src/gfa2fa.cpp compiled by oracle/Makefile into oracle/_ref/HS_gfa2fa.
Run in the build container (needs /root/reference): python oracle/gen_gfa_goldens.py"""
import gzip
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFSRC = os.path.join(os.environ.get("HS_REFERENCE", "/root/reference"), "src")
OUT = os.path.join(ROOT, "tests", "golden", "gfa_tools")


def synthetic(rnd):
    """S lines with and without tags, a trailing tab, an empty sequence, a two-field S line; L lines in all four orientations,
    with an overlap field and extra fields; a header and a P line that must be ignored"""
    seq = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))
    lines = ["H\tVN:Z:1.0"]
    names = {"a": 2500, "b": 1000, "c": 999, "d": 1, "e": 3001, "long_name.with-chars": 1234}
    for k, (n, ln) in enumerate(names.items()):
        tags = ["", "\tDP:f:%d" % (10 + k), "\tDP:f:3\tLN:i:%d" % ln, "\tdp:i:7\t"][k % 4]
        lines.append("S\t%s\t%s%s" % (n, seq(ln), tags))
    lines.append("S\tempty\t")
    lines.append("S\tnoseq")
    lines += ["L\ta\t+\tb\t+\t0M", "L\ta\t-\tc\t-\t12M", "L\tb\t+\te\t-\t0M\tRC:i:3", "L\te\t-\tlong_name.with-chars\t+\t5M", "L\td\t+\td\t+\t0M",
              "P\tpath1\ta+,b+\t*"]
    return "\n".join(lines) + "\n"


def main():
    os.makedirs(OUT, exist_ok=True)
    rnd = random.Random(17)
    cases = []
    mock = os.path.join(os.path.dirname(REFSRC), "test", "simple_mock", "assembly.gfa")
    with tempfile.TemporaryDirectory() as td:
        syn = os.path.join(td, "syn.gfa")
        open(syn, "w").write(synthetic(rnd))
        nonl = os.path.join(td, "nonl.gfa")
        open(nonl, "w").write("S\tx\tACGTACGTAC\nL\tx\t+\tx\t-\t0M")       # no newline at the end of the file
        for name, src, lengths in (("syn", syn, [1000, 999, 7, 100000]), ("nonl", nonl, [4]), ("simple_mock", mock, [300000, 30000])):
            for L in lengths:
                out = os.path.join(td, "cut.gfa")
                subprocess.run([sys.executable, os.path.join(REFSRC, "cut_gfa.py"), "-a", src, "-l", str(L), "-o", out], check=True)
                tag = "%s_cut%d" % (name, L)
                with open(out, "rb") as fi, gzip.GzipFile(os.path.join(OUT, tag + ".gfa.gz"), "wb", mtime=0) as fo:
                    shutil.copyfileobj(fi, fo)
                cases.append({"tool": "cut_gfa", "input": name + ".gfa", "length": L, "expected": tag + ".gfa.gz"})
            fa = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "HS_gfa2fa"), src], check=True, stdout=subprocess.PIPE).stdout
            with gzip.GzipFile(os.path.join(OUT, name + ".fa.gz"), "wb", mtime=0) as fo:
                fo.write(fa)
            cases.append({"tool": "gfa2fa", "input": name + ".gfa", "expected": name + ".fa.gz"})
            if name != "simple_mock":      # (that file is tests/golden/simple_mock/assembly.gfa.gz already)
                with open(src, "rb") as fi, gzip.GzipFile(os.path.join(OUT, name + ".gfa.gz"), "wb", mtime=0) as fo:
                    shutil.copyfileobj(fi, fo)
    json.dump(cases, open(os.path.join(OUT, "cases.json"), "w"), indent=1)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
