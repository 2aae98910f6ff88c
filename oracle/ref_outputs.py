"""CHECKER (test infrastructure): the reference's output files (.col / .gro / error_rate.txt) next to the results an in-memory
pipeline call leaves with the host, contig by contig.

Used by bench.py's parity gate (after the timed region, on the results of the LAST TIMED STEP) and by tests/. The files are
read by oracle/_build/libhs_oracle.so (hso_parse_blocks); nothing here is on the product path, and nothing under
hairsplitter_amd/ imports it.

match: the writers of call_variants.cpp:1184-1211 (.col), :1310-1316,1377 (error rate) and separate_reads.cpp:1754-1786 (.gro).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


class _Blocks(C.Structure):
    _fields_ = [("n_contigs", C.c_int32), ("names_len", C.c_int64), ("names", C.POINTER(C.c_char)), ("extra_len", C.c_int64), ("extra", C.POINTER(C.c_char)),
                ("n_read_lines", C.POINTER(C.c_int32)), ("rec_off", C.POINTER(C.c_int64)), ("n_records", C.c_int64),
                ("a", C.POINTER(C.c_int32)), ("b", C.POINTER(C.c_int32)), ("c", C.POINTER(C.c_int32)),
                ("ent_off", C.POINTER(C.c_int64)), ("idx", C.POINTER(C.c_int32)), ("val", C.POINTER(C.c_int32))]


def _oracle():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libhs_oracle.so"))
        _lib.hso_parse_blocks.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.POINTER(C.POINTER(_Blocks))]
        _lib.hso_blocks_free.argtypes = [C.POINTER(_Blocks)]
        _lib.hso_blocks_free.restype = None
    return _lib


def _arr(p, n, dtype):
    return np.ctypeslib.as_array(p, (max(int(n), 1),))[:int(n)].astype(dtype, copy=True)


def read_blocks(path: str, tag: str) -> Dict:
    """A .col (tag "SNPS") or .gro (tag "GROUP") file as flat arrays, contigs in file order:
    names [C], extra [C] (the CONTIG line after the name), n_read_lines [C], rec_off [C+1], a / b / c [R], ent_off [R+1], idx, val."""
    lib = _oracle()
    out = C.POINTER(_Blocks)()
    rc = lib.hso_parse_blocks(path.encode(), tag.encode(), C.c_int32(3 if tag == "SNPS" else 2), C.byref(out))
    if rc != 0:
        raise RuntimeError(f"cannot read {path} as {tag} blocks (code {rc})")
    try:
        b = out.contents
        Cn, R = int(b.n_contigs), int(b.n_records)
        names = C.string_at(b.names, int(b.names_len)).decode().split("\n")[:Cn]
        extra = C.string_at(b.extra, int(b.extra_len)).decode().split("\n")[:Cn]
        ent_off = _arr(b.ent_off, R + 1, np.int64)
        E = int(ent_off[-1])
        return {"names": names, "extra": extra, "n_read_lines": _arr(b.n_read_lines, Cn, np.int32), "rec_off": _arr(b.rec_off, Cn + 1, np.int64),
                "a": _arr(b.a, R, np.int32), "b": _arr(b.b, R, np.int32), "c": _arr(b.c, R, np.int32), "ent_off": ent_off,
                "idx": _arr(b.idx, E, np.int32), "val": _arr(b.val, E, np.int32)}
    finally:
        lib.hso_blocks_free(out)


def mean_of_positive_f32(md) -> np.float32:
    """call_variants.cpp:1312-1315: float sum in contig order over the contigs with a positive distance / their number"""
    tot, n = np.float32(0), 0
    for x in np.asarray(md, np.float32):
        if x > 0:
            tot = np.float32(tot + x); n += 1
    return np.float32(tot / np.float32(n)) if n else np.float32(0)


def pipeline_snapshot(pg, cv: Dict, sr: Dict, names: List[str]) -> Dict:
    """Copies of everything one PipelineGroups call left with the host (the arrays of `sr["sparse"]` and the groups' stage-3
    results belong to the pipeline and die with its next call): per window bounds + the reads it lists + their labels, per contig
    mean distance, depth and the SNPs' positions / alleles / read counts -- and their entries when the call kept them
    (HS_PIPELINE_KEEP_COLUMNS)."""
    from hairsplitter_amd import api
    lib = api.load()
    lib.hs_pipeline_group_cv.restype = C.POINTER(api.CvResult)
    Cn = len(names)
    snap = {"names": list(names), "mean_distance": np.array(cv["mean_distance"], np.float32, copy=True),
            "win_off": np.array(sr["win_off"], np.int64, copy=True), "win_start": np.array(sr["win_start"], np.int32, copy=True),
            "win_end": np.array(sr["win_end"], np.int32, copy=True)}
    if "sparse" in sr:
        off, ids, lab = sr["sparse"]
        snap["row_off"], snap["ids"], snap["lab"] = np.array(off, np.int64, copy=True), np.array(ids, np.int32, copy=True), np.array(lab, np.int32, copy=True)
    else:      # the dense form: labels [window][reads of the contig], -2 = not in the window
        lo, lab = sr["label_off"], sr["labels"]
        seg = np.repeat(np.arange(len(lo) - 1), np.diff(lo))
        keep = lab != -2
        snap["row_off"] = np.concatenate(([0], np.cumsum(np.bincount(seg[keep], minlength=len(lo) - 1)))).astype(np.int64)
        snap["ids"] = (np.arange(len(lab)) - lo[seg])[keep].astype(np.int32)
        snap["lab"] = np.array(lab[keep], np.int32, copy=True)
    snp_n, depth, pos, ref, alt, nref, nalt, eoff, idx, code = [], [], [], [], [], [], [], [], [], []
    have_entries = True
    c_seen = 0
    ebase = 0
    for g in range(lib.hs_pipeline_groups(pg.handle)):
        c0, c1 = C.c_int32(0), C.c_int32(0)
        lib.hs_pipeline_group_range(pg.handle, C.c_int32(g), C.byref(c0), C.byref(c1))
        assert c0.value == c_seen, "contig groups are consecutive ranges"
        c_seen = c1.value
        rp = lib.hs_pipeline_group_cv(pg.handle, C.c_int32(g))
        assert bool(rp), "no stage-3 result of group %d" % g
        r = rp.contents
        n = int(r.n_contigs)
        assert n == c1.value - c0.value
        so = _arr(r.snp_off, n + 1, np.int64)
        S = int(so[-1])
        snp_n.append(np.diff(so)); depth.append(_arr(r.depth, n, np.float32))
        pos.append(_arr(r.snp_pos, S, np.int32)); ref.append(_arr(r.snp_ref, S, np.int32)); alt.append(_arr(r.snp_alt, S, np.int32))
        nref.append(_arr(r.snp_n_ref, S, np.int32)); nalt.append(_arr(r.snp_n_alt, S, np.int32))
        if bool(r.col_off) and bool(r.col_idx) and bool(r.col_code):
            co = _arr(r.col_off, S + 1, np.int64)
            E = int(co[-1])
            eoff.append(co[:-1] + ebase); ebase += E
            idx.append(_arr(r.col_idx, E, np.int32)); code.append(_arr(r.col_code, E, np.int32))
        else:
            have_entries = False
    assert c_seen == Cn
    cat = lambda v, dt: np.concatenate(v).astype(dt) if v else np.zeros(0, dt)
    snap["snp_off"] = np.concatenate(([0], np.cumsum(cat(snp_n, np.int64)))).astype(np.int64)
    snap["depth"] = cat(depth, np.float32)
    snap["snp_pos"], snap["snp_ref"], snap["snp_alt"] = cat(pos, np.int32), cat(ref, np.int32), cat(alt, np.int32)
    snap["snp_n_ref"], snap["snp_n_alt"] = cat(nref, np.int32), cat(nalt, np.int32)
    if have_entries:
        snap["col_off"] = np.concatenate((cat(eoff, np.int64), [ebase])).astype(np.int64)
        snap["col_idx"], snap["col_code"] = cat(idx, np.int32), cat(code, np.int32)
    return snap


def _seg_equal(off_a, val_a, a0, a1, off_b, val_b, b0, b1) -> bool:
    """records [a0, a1) of one CSR against [b0, b1) of another: same lengths and same values"""
    if a1 - a0 != b1 - b0:
        return False
    la, lb = np.diff(off_a[a0:a1 + 1]), np.diff(off_b[b0:b1 + 1])
    if not np.array_equal(la, lb):
        return False
    return np.array_equal(val_a[off_a[a0]:off_a[a1]], val_b[off_b[b0]:off_b[b1]])


def compare_with_reference(snap: Dict, col_path: str, gro_path: Optional[str], err_path: Optional[str] = None, max_diffs: int = 3, subset: bool = False) -> Dict:
    """The snapshot of a pipeline call against the reference's files of the same job, per contig NAME (the reference writes its
    contigs in hash-map / thread-completion order). .gro: the GROUP lines (bounds, reads, labels) of every contig, and which contigs
    have a block at all (contigs without a SNP have none, separate_reads.cpp:1522-1524). .col: depth, positions and alleles of the
    SNPS lines, the number of reads with either allele -- and the lines' entries where the snapshot has them. error_rate.txt: the text.
    `subset`: the files hold only some of the snapshot's contigs (a bounded sample of the job): those are compared, the error rate
    (a mean over the whole job) is not -- and neither is the .gro (pass gro_path=None): stage 4 of a sample ran with the sample's error rate."""
    out: Dict = {"checked": True, "contigs_compared": "all" if not subset else "those of the files"}
    names = snap["names"]
    idx_of = {n: i for i, n in enumerate(names)}
    diffs: List[str] = []

    # ---- .gro ----
    if gro_path is None:
        out["gro_identical"] = None
    else:
        g = read_blocks(gro_path, "GROUP")
        ok = True
        seen = set()
        n_groups = 0
        for k, name in enumerate(g["names"]):
            c = idx_of.get(name)
            if c is None:
                ok = False; diffs.append(f".gro: contig {name!r} is not in the job"); continue
            seen.add(c)
            w0, w1 = int(snap["win_off"][c]), int(snap["win_off"][c + 1])
            r0, r1 = int(g["rec_off"][k]), int(g["rec_off"][k + 1])
            n_groups += r1 - r0
            same = (w1 - w0 == r1 - r0 and np.array_equal(snap["win_start"][w0:w1], g["a"][r0:r1]) and np.array_equal(snap["win_end"][w0:w1], g["b"][r0:r1])
                    and _seg_equal(snap["row_off"], snap["ids"], w0, w1, g["ent_off"], g["idx"], r0, r1)
                    and _seg_equal(snap["row_off"], snap["lab"], w0, w1, g["ent_off"], g["val"], r0, r1))
            if not same:
                ok = False
                if len(diffs) < max_diffs:
                    diffs.append(f".gro: contig {name!r}: {w1 - w0} windows here, {r1 - r0} GROUP lines there, or their reads / labels differ")
        # contigs the reference wrote no block for must have no SNP (and no window) here
        for c in range(len(names) if not subset else 0):
            if c not in seen and int(snap["win_off"][c + 1]) - int(snap["win_off"][c]) > 0:
                ok = False
                if len(diffs) < max_diffs:
                    diffs.append(f".gro: contig {names[c]!r} has windows here and no block in the reference's file")
        out["gro_identical"] = bool(ok)
        out["gro_contigs"], out["gro_group_lines"] = len(g["names"]), int(n_groups)
        g = None

    # ---- .col ----
    col = read_blocks(col_path, "SNPS")
    ok = True
    ok_entries = "col_idx" in snap
    seg = np.repeat(np.arange(len(col["a"])), np.diff(col["ent_off"]))
    n_ref = np.bincount(seg, weights=(col["val"] == col["b"][seg]), minlength=len(col["a"])).astype(np.int64)
    n_alt = np.bincount(seg, weights=(col["val"] == col["c"][seg]), minlength=len(col["a"])).astype(np.int64)
    seg = None
    seen = set()
    for k, name in enumerate(col["names"]):
        c = idx_of.get(name)
        if c is None:
            ok = False; diffs.append(f".col: contig {name!r} is not in the job"); continue
        seen.add(c)
        s0, s1 = int(snap["snp_off"][c]), int(snap["snp_off"][c + 1])
        r0, r1 = int(col["rec_off"][k]), int(col["rec_off"][k + 1])
        depth_txt = col["extra"][k].split("\t")[-1]
        same = (s1 - s0 == r1 - r0 and np.array_equal(snap["snp_pos"][s0:s1], col["a"][r0:r1]) and np.array_equal(snap["snp_ref"][s0:s1], col["b"][r0:r1])
                and np.array_equal(snap["snp_alt"][s0:s1], col["c"][r0:r1]) and np.array_equal(snap["snp_n_ref"][s0:s1], n_ref[r0:r1])
                and np.array_equal(snap["snp_n_alt"][s0:s1], n_alt[r0:r1]) and ("%g" % snap["depth"][c]) == depth_txt)
        if not same:
            ok = False
            if len(diffs) < max_diffs:
                diffs.append(f".col: contig {name!r}: {s1 - s0} SNPs here, {r1 - r0} SNPS lines there (depth {'%g' % snap['depth'][c]} / {depth_txt}), or positions / alleles / counts differ")
        elif ok_entries:
            if not (_seg_equal(snap["col_off"], snap["col_idx"], s0, s1, col["ent_off"], col["idx"], r0, r1)
                    and _seg_equal(snap["col_off"], snap["col_code"], s0, s1, col["ent_off"], col["val"], r0, r1)):
                ok_entries = False
                if len(diffs) < max_diffs:
                    diffs.append(f".col: contig {name!r}: the entries of a SNPS line differ")
    if len(seen) != len(names) and not subset:
        ok = False
        diffs.append(f".col: {len(names) - len(seen)} contigs of the job have no block in the reference's file")
    out["col_snps_identical"] = bool(ok)
    out["col_entries_identical"] = bool(ok and ok_entries) if "col_idx" in snap else None
    out["col_snps_lines"] = int(len(col["a"]))
    col = None

    # ---- error rate ----
    if err_path is not None and not subset:
        want = open(err_path).read().strip()
        got = "%g" % mean_of_positive_f32(snap["mean_distance"])
        out["error_rate_identical"] = bool(got == want)
        if got != want:
            diffs.append(f"error rate: {got} here, {want} in the reference's file")
    # `identical` covers what was compared, and says what that was: a part that was not compared (the .gro and the error rate of a sample of
    # the job, the entries of a step that left them on the device) is named in identical_scope, never counted as equal
    compared = [k for k in ("gro_identical", "col_snps_identical", "col_entries_identical", "error_rate_identical") if out.get(k) is not None]
    out["identical"] = bool(compared and all(out[k] for k in compared))
    out["identical_scope"] = "compared: " + ", ".join(k[:-len("_identical")] for k in compared) + \
        ("; NOT compared: " + ", ".join(k[:-len("_identical")] for k in ("gro_identical", "col_snps_identical", "col_entries_identical", "error_rate_identical") if out.get(k) is None)
         if len(compared) < 4 else "")
    if diffs:
        out["diffs"] = diffs[:max_diffs + 2]
    return out
