"""K0 + K1 alone on a C4-like batch (hs_pileup: no stage behind it), a few launches; for rocprofv3 --kernel-trace --stats (tools/gpu_k1_alone.sh)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hairsplitter_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 160
cs = [synth.make_contig(21, i, 100_000, 1 + i % 4, 0.01, 30, "ont") for i in range(n)]
flat = api.FlatBatch(cs)
t = api.device_tensors(flat)
for rep in range(6):
    api.pileup(t, flat)
print("aligned bp", flat.aligned_bp)
