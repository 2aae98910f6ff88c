"""PCIe-inclusive rate (never bench.py's `value`): host buffers in (hs_cv_batch_create = H2D of contigs, reads, CIGARs and
the launch plan) + the whole hot path, per batch of 16 C2 contigs."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hairsplitter_amd import api, synth
torch.set_num_threads(1)
contigs = [synth.make_contig(2, i, 100_000, 2, 0.01, 50, "ont") for i in range(16)]
flat = api.FlatBatch(contigs)
for it in range(4):
    t0 = time.perf_counter()
    b = api.CvBatch(flat)
    t1 = time.perf_counter()
    cv, sr = b.run_pipeline(0.33, 64)
    t2 = time.perf_counter()
    b.close()
    print(f"create(H2D) {1e3*(t1-t0):.1f} ms, pipeline {1e3*(t2-t1):.1f} ms, PCIe-inclusive {flat.aligned_bp/(t2-t0)/1e9:.2f} Gbp/s, resident {flat.aligned_bp/(t2-t1)/1e9:.2f} Gbp/s")
