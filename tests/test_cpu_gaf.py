"""The .gro consumer of the next stage (parse_split_file + merge_intervals + output_GAF of
create_new_contigs.cpp) against the reference: the golden .gaf files (written by the reference's own
HS_create_new_contigs, oracle/gen_goldens.py) and, where oracle/_ref holds that binary, the reference run live on
randomised .gro files. Host code only: runs without a GPU."""
import os
import random
import subprocess
import tempfile

import numpy as np
import pytest

import golden_util as gu


def _read(path):
    with open(path, "rb") as f:
        return f.read()


@pytest.mark.parametrize("case", gu.case_names())
def test_gaf_tool_matches_reference_goldens(built, case):
    """hs_gro_to_gaf on the reference's own .gro reproduces, byte for byte, the .gaf the reference wrote from it"""
    with tempfile.TemporaryDirectory() as td:
        meta = gu.unpack(case, td)
        amp = str(meta.get("kwargs", {}).get("amplicon", 0))
        out = os.path.join(td, "mine.gaf")
        r = subprocess.run([built["gaf"], os.path.join(td, "assembly.gfa"), gu.reads_path(td, meta), os.path.join(td, "aln.sam"),
                            os.path.join(td, "reads_haplo.gro"), amp, out, "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0, r.stdout.decode()[-2000:]
        assert _read(out) == _read(os.path.join(td, "reads_haplo.gaf"))


def _gro_to_arrays(td):
    """The golden .gro as the arrays of an hs_sr_result (contigs in GFA order; absent reads = -2)"""
    names = [l.split("\t")[1].split(" ")[0] for l in open(os.path.join(td, "assembly.gfa")) if l.startswith("S\t")]
    per = {}
    cur = None
    for l in open(os.path.join(td, "reads_haplo.gro")):
        f = l.rstrip("\n").split("\t")
        if f[0] == "CONTIG":
            cur = per.setdefault(f[1], {"n": 0, "win": []})
            cur["n"] = 0; cur["win"] = []
        elif f[0] == "READ":
            cur["n"] += 1
        elif f[0] == "GROUP":
            idx = [int(x) for x in f[3].split(",") if x] if len(f) > 3 else []
            lab = [int(x) for x in f[4].split(",") if x] if len(f) > 4 else []
            cur["win"].append((int(f[1]), int(f[2]), idx, lab))
    win_off, win_start, win_end, label_off, labels, has = [0], [], [], [0], [], []
    for nm in names:
        c = per.get(nm)
        has.append(1 if c is not None else 0)
        if c is not None:
            for s, e, idx, lab in c["win"]:
                v = np.full(c["n"], -2, np.int32)
                v[idx] = lab
                labels.append(v); win_start.append(s); win_end.append(e); label_off.append(label_off[-1] + c["n"])
        win_off.append(len(win_start))
    return ({"win_off": np.asarray(win_off, np.int64), "win_start": np.asarray(win_start, np.int32), "win_end": np.asarray(win_end, np.int32),
             "label_off": np.asarray(label_off, np.int64), "labels": np.concatenate(labels) if labels else np.zeros(0, np.int32)}, has)


@pytest.mark.parametrize("case", ["multi", "linked", "penta30k", "short_reads_w500"])
def test_gaf_from_labels_in_memory(built, case):
    """hs_gaf_from_labels (windows + labels as arrays, no .gro text) gives the same file"""
    from hairsplitter_amd import api
    with tempfile.TemporaryDirectory() as td:
        gu.unpack(case, td)
        sr, has = _gro_to_arrays(td)
        out = os.path.join(td, "mem.gaf")
        for h in (has, None):
            api.gaf_from_labels(os.path.join(td, "assembly.gfa"), os.path.join(td, "reads.fasta"), os.path.join(td, "aln.sam"), sr, out, h)
            assert _read(out) == _read(os.path.join(td, "reads_haplo.gaf"))
            os.remove(out)


def _random_gro(td, rng, style):
    """A .gro with the CONTIG / READ lines of the case's .col and made-up windows: cluster ids drift, split, merge and
    vanish from window to window, which is what merge_intervals / stitch have to sort out ("sparse": many tiny clusters with
    a third of the reads in a wrong one, so that hardly any junction is a one-to-one continuation)"""
    out = []
    blocks = []
    for l in open(os.path.join(td, "variants.col")):
        if l.startswith("CONTIG"):
            blocks.append([l.rstrip("\n"), []])
        elif l.startswith("READ"):
            blocks[-1][1].append(l.rstrip("\n"))
    for head, reads in blocks:
        if not reads or rng.random() < 0.15:
            continue   # contigs the writer skipped
        n = len(reads)
        L = int(head.split("\t")[2])
        out.append(head); out.extend(reads)
        w = rng.choice([500, 1000, 2000, 5000])
        k_clusters = rng.choice([8, 12, 20]) if style == "sparse" else rng.choice([1, 2, 2, 3, 4, 6])
        truth = [rng.randrange(k_clusters) for _ in range(n)]
        perm = list(range(k_clusters))
        pos = 0
        while pos < L:
            end = min(L, pos + w) - 1
            if rng.random() < 0.3:
                rng.shuffle(perm)                                   # relabelled: still a one-to-one continuation
            drop = rng.randrange(k_clusters) if (style == "rough" and rng.random() < 0.3) else -1
            fuse = rng.random() < (0.25 if style == "rough" else 0.05)
            idx, lab = [], []
            for r in range(n):
                u = rng.random()
                if u < 0.25:
                    continue                                        # absent (-2): not listed
                v = perm[truth[r]]
                if truth[r] == drop:
                    v = -1
                elif fuse and v == 1:
                    v = 0
                elif u < 0.32:
                    v = -1
                elif u < (0.65 if style == "sparse" else 0.36):
                    v = rng.randrange(k_clusters)                   # a read in the wrong cluster
                idx.append(r); lab.append(v)
            if style == "rough" and rng.random() < 0.08:
                idx, lab = [], []                                   # empty window
            out.append("GROUP\t%d\t%d\t%s\t%s" % (pos, end, "".join("%d," % i for i in idx), "".join("%d," % v for v in lab)))
            pos += w
    p = os.path.join(td, "random.gro")
    with open(p, "w") as f:
        f.write("\n".join(out) + "\n")
    return p


@pytest.mark.parametrize("case,style,seed", [("linked", "smooth", 1), ("linked", "rough", 2), ("multi", "rough", 3), ("penta30k", "rough", 4),
                                             ("linked", "rough", 5), ("short_reads_w500", "smooth", 6), ("clips", "rough", 7),
                                             ("penta30k", "sparse", 8), ("linked", "sparse", 9), ("lowdepth", "sparse", 10)])
def test_gaf_against_live_reference_on_random_partitions(built, case, style, seed):
    """Randomised windows/labels through the reference binary and through hs_gro_to_gaf: identical files. (The reference
    stops later, at its first external tool; the .gaf is complete by then.) Skipped where oracle/_ref was not built."""
    if not os.path.exists(built["ref_cnc"]):
        pytest.skip("oracle/_ref/HS_create_new_contigs not built here")
    rng = random.Random(seed)
    with tempfile.TemporaryDirectory() as td:
        gu.unpack(case, td)
        for rep in range(3):
            gro = _random_gro(td, rng, style)
            ref, mine, tmp = os.path.join(td, "ref.gaf"), os.path.join(td, "mine.gaf"), os.path.join(td, "cnc_tmp")
            os.makedirs(tmp, exist_ok=True)
            for p in (ref, mine):
                if os.path.exists(p):
                    os.remove(p)
            subprocess.run([built["ref_cnc"], os.path.join(td, "assembly.gfa"), os.path.join(td, "reads.fasta"), "0.05", gro,
                            os.path.join(td, "aln.sam"), tmp + "/", "1", "ont", os.path.join(tmp, "o.gfa"), ref, "racon", "0", "0",
                            "/nonexistent/minimap2", "/nonexistent/racon", "/nonexistent/medaka", "/nonexistent/samtools",
                            "/nonexistent/python", "0"], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            assert os.path.exists(ref), "the reference did not write a .gaf"
            r = subprocess.run([built["gaf"], os.path.join(td, "assembly.gfa"), os.path.join(td, "reads.fasta"), os.path.join(td, "aln.sam"),
                                gro, "0", mine], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            assert r.returncode == 0, r.stdout.decode()[-2000:]
            assert _read(mine) == _read(ref), "rep %d" % rep


def test_gaf_tool_usage_and_errors(built):
    assert subprocess.run([built["gaf"], "--help"], stdout=subprocess.DEVNULL).returncode == 0
    assert subprocess.run([built["gaf"]], stdout=subprocess.DEVNULL).returncode == 1
    with tempfile.TemporaryDirectory() as td:
        gu.unpack("dip20k", td)
        r = subprocess.run([built["gaf"], os.path.join(td, "assembly.gfa"), os.path.join(td, "reads.fasta"), os.path.join(td, "aln.sam"),
                            os.path.join(td, "missing.gro"), "0", os.path.join(td, "o.gaf")], stdout=subprocess.PIPE)
        assert r.returncode == 1 and b"could not open" in r.stdout
