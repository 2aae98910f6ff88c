"""Throughput of the A1 kernel at the shapes of the stage-5 call sites (create_new_contigs.cpp:558-629: the 300-base end of a
piece inside its polished version of a few kb; tools.cpp:515-534: 200 bases inside 300): pairs/s and DP cell updates/s of
hs_edlib_hw_align (HW + start location + path), next to the reference's edlib on one host core on a sample of the same pairs.
With a query length above 300 the pairs are READS against a contig window (the read with ~8 % substitutions, insertions and
deletions inside target_len bases): the banded sweeps and Hirschberg's cuts, one wavefront per read.
Usage: python tools/myers_bench.py [n_pairs=20000] [target_len=2000] [query_len=300]     (prints one JSON line)"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    tl = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    import torch
    from hairsplitter_amd import api
    api.require_gpu()
    lib = api.load()
    rng = np.random.default_rng(9)
    qn = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    reads = qn > 300
    q = rng.integers(0, 4, size=(n, qn), dtype=np.uint8)
    t = rng.integers(0, 4, size=(n, tl), dtype=np.uint8)
    pos = rng.integers(0, tl - qn - (qn >> 4), size=n)
    for i in range(n):                       # the query, with ~4 % substitutions, somewhere inside the target
        m = q[i].copy()
        e = rng.random(qn) < 0.04
        m[e] = (m[e] + 1) & 3
        if reads:                            # reads: 2 % deletions and 2 % insertions on top
            keep = rng.random(qn) >= 0.02
            m = m[keep]
            ins = np.flatnonzero(rng.random(len(m)) < 0.02)
            m = np.insert(m, ins, rng.integers(0, 4, size=len(ins), dtype=np.uint8))
        t[i, pos[i]:pos[i] + len(m)] = m
    qo = np.arange(n + 1, dtype=np.int64) * qn
    to = np.arange(n + 1, dtype=np.int64) * tl
    oo = np.arange(n + 1, dtype=np.int64) * (qn + tl)
    dev = "cuda:0"
    dq = torch.from_numpy(q.reshape(-1)).to(dev); dt = torch.from_numpy(t.reshape(-1)).to(dev)
    dd = torch.zeros(n, dtype=torch.int32, device=dev); ds = torch.zeros_like(dd); de = torch.zeros_like(dd); dl = torch.zeros_like(dd)
    dops = torch.zeros(int(oo[-1]), dtype=torch.uint8, device=dev)
    hp = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    out = {"pairs": n, "query_len": qn, "target_len": tl}
    for path in (True, False):
        times = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            api._check(lib.hs_edlib_hw_align(api._p(dq), hp(qo), api._p(dt), hp(to), C.c_int32(n), api._p(dd), api._p(ds), api._p(de),
                                             api._p(dops) if path else C.c_void_p(0), hp(oo), api._p(dl), C.c_void_p(0)))
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        best = min(times[1:])
        # cells: sweep 1 over the whole target, sweep 2 over the prefix up to the end, sweep 3 over the aligned part
        end = de.cpu().numpy().astype(np.int64); st = ds.cpu().numpy().astype(np.int64)
        cells = float((tl + (end + 1) + ((end - st + 1) if path else 0)).sum()) * qn
        out["path" if path else "locations_only"] = {"seconds": best, "pairs_per_s": n / best, "GCUPS": cells / best / 1e9}
    assert int((np.abs(ds.cpu().numpy() - pos) <= (8 if reads else 0)).sum()) > 0.95 * n       # the planted placement is found
    ref = os.path.join(ROOT, "oracle", "_ref", "edlib_driver")
    if os.path.exists(ref):
        k = min(n, 300 if not reads else 40)
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        lines = "".join("HWPATH -1 %s %s\n" % (acgt[q[i]].tobytes().decode(), acgt[t[i]].tobytes().decode()) for i in range(k))
        t0 = time.perf_counter()
        subprocess.run([ref], input=lines, capture_output=True, text=True, check=True)
        dt_ = time.perf_counter() - t0
        out["reference_edlib_one_core"] = {"pairs": k, "seconds": dt_, "pairs_per_s": k / dt_}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
