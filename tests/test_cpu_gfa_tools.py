"""The native GFA text tools (hs_cut_gfa == src/cut_gfa.py, HS_gfa2fa == src/gfa2fa.cpp) against outputs of the reference
itself (tests/golden/gfa_tools, written by oracle/gen_gfa_goldens.py). Host code: no GPU needed."""
import gzip
import json
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "gfa_tools")
CASES = json.load(open(os.path.join(GOLD, "cases.json")))


def _input(name, td):
    src = os.path.join(GOLD, name + ".gz") if name != "simple_mock.gfa" else os.path.join(ROOT, "tests", "golden", "simple_mock", "assembly.gfa.gz")
    dst = os.path.join(td, name)
    with gzip.open(src, "rb") as fi, open(dst, "wb") as fo:
        fo.write(fi.read())
    return dst


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["expected"])
def test_gfa_tool_matches_the_reference(built, case):
    want = gzip.open(os.path.join(GOLD, case["expected"]), "rb").read()
    with tempfile.TemporaryDirectory() as td:
        src = _input(case["input"], td)
        if case["tool"] == "cut_gfa":
            out = os.path.join(td, "out.gfa")
            r = subprocess.run([built["cut_gfa"], "--assembly", src, "-l", str(case["length"]), "--output", out], stdout=subprocess.PIPE)
            assert r.returncode == 0, r.stdout
            assert open(out, "rb").read() == want
        else:
            r = subprocess.run([built["gfa2fa"], src], stdout=subprocess.PIPE)
            assert r.returncode == 0
            assert r.stdout == want


def test_cut_gfa_refuses_what_the_reference_crashes_on(built):
    """an L line that names a contig without an S line raises KeyError in cut_gfa.py: non-zero exit here"""
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "bad.gfa")
        open(src, "w").write("S\ta\tACGT\nL\ta\t+\tghost\t-\t0M\n")
        r = subprocess.run([built["cut_gfa"], "-a", src, "-l", "2", "-o", os.path.join(td, "o.gfa")], stdout=subprocess.PIPE)
        assert r.returncode != 0
