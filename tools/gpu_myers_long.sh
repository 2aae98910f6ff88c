#!/bin/bash
# A1 beyond one leaf matrix: the parity tests of the alignment path and the time of the long vectors (one wavefront per pair).
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "edlib or myers or stage5" > gpurun_out/myers_long_tests.log 2>&1
tail -5 gpurun_out/myers_long_tests.log
timeout 600 python - > gpurun_out/myers_long_time.log 2>&1 <<'P'
import gzip, json, time, sys
sys.path.insert(0, "tests")
from hairsplitter_amd import api
vec = json.loads(gzip.open("tests/golden/edlib_long_path_vectors.json.gz").read())
pairs = [(v["query"], v["target"]) for v in vec]
api.edlib_hw_align(pairs[:2])
for lo, hi in ((0, 11), (11, 34), (34, 35), (35, 36), (36, 37), (0, 37)):
    t0 = time.time(); api.edlib_hw_align(pairs[lo:hi]); t1 = time.time()
    api.edlib_hw_align(pairs[lo:hi], path=False); t2 = time.time()
    print("pairs %d..%d longest query %d: path %.3f s, locations only %.3f s" % (lo, hi, max(len(p[0]) for p in pairs[lo:hi]), t1 - t0, t2 - t1))
P
cat gpurun_out/myers_long_time.log
timeout 600 python - > gpurun_out/myers_dist_time.log 2>&1 <<'P'
import gzip, json, time, sys
import numpy as np
from hairsplitter_amd import api
vec = json.loads(gzip.open("tests/golden/edlib_long_path_vectors.json.gz").read())
code = np.full(256, 3, np.uint8)
for i, ch in enumerate(b"ACGT"): code[ch] = i
enc = lambda x: code[np.frombuffer(x.encode(), dtype=np.uint8)]
qs = [enc(v["query"]) for v in vec]; ts = [enc(v["target"]) for v in vec]
api.edit_distance(qs[:2], ts[:2], "NW")
for mode in ("NW", "SHW", "HW"):
    for lo, hi in ((0, 35), (35, 36), (36, 37)):
        t0 = time.time(); api.edit_distance(qs[lo:hi], ts[lo:hi], mode); t1 = time.time()
        print("%s pairs %d..%d longest query %d: %.3f s" % (mode, lo, hi, max(len(q) for q in qs[lo:hi]), t1 - t0))
P
cat gpurun_out/myers_dist_time.log
