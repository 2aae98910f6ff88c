"""diagnostic (library built with EXTRA=-DHS_K1R_PROFILE): cycles per phase of a K1 task, averaged over the tasks of a C4-like batch"""
import sys, os, ctypes as C, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hairsplitter_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
cs = [synth.make_contig(21, i, 100_000, 1 + i % 4, 0.01, 30, "ont") for i in range(n)]
flat = api.FlatBatch(cs)
t = api.device_tensors(flat)
lib = api.load()
for rep in range(2):
    api.pileup(t, flat)
    time.sleep(0.1)
# number of run tasks: two 64-op chunks each
nch = np.diff(flat.rec_cig_off)
ntask = int(((((nch + 63) // 64) + 1) // 2).sum())
ntask = min(ntask, 1 << 20)
buf = np.zeros(8 * ntask, np.uint32)
lib.hs_debug_k1_profile(buf.ctypes.data_as(C.c_void_p), C.c_longlong(buf.size))
v = buf.reshape(-1, 8).astype(np.float64)
v = v[v[:, 6] > 0]
names = ["metadata + chunk table", "cigar + scans + op records", "piece map", "first fetch", "rounds", "rounds per task", "task total", "pieces per task"]
print("tasks", len(v), "aligned bp", flat.aligned_bp)
for i, a in enumerate(names):
    print("  %-28s mean %10.1f  median %10.1f  p90 %10.1f" % (a, v[:, i].mean(), np.median(v[:, i]), np.percentile(v[:, i], 90)))
