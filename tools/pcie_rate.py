"""PCIe-inclusive rate (never bench.py's `value`): host buffers in (hs_cv_batch_create = H2D of contigs, reads, CIGARs and
the launch plans) + the whole hot path, per batch of C2 contigs (default 64, as bench.py)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hairsplitter_amd import api, synth
torch.set_num_threads(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
contigs = [synth.make_contig(2, i, 100_000, 2, 0.01, 50, "ont") for i in range(n)]
for it in range(5):
    t0 = time.perf_counter()
    g = api.PipelineGroups(contigs, 8)      # FlatBatch (host packing, not counted below) + hs_cv_batch_create + pipeline threads
    t1 = time.perf_counter()
    t_pack = time.perf_counter()
    flat = g.flat
    # time the device-side creation alone on a second batch object
    b = api.CvBatch(flat)
    t2 = time.perf_counter()
    cv, sr = g.run(0.33, 64)
    t3 = time.perf_counter()
    b.close(); g.close()
    create = t2 - t_pack
    print(f"create(H2D + plans) {1e3*create:.1f} ms, pipeline {1e3*(t3-t2):.1f} ms, PCIe-inclusive {flat.aligned_bp/(create + t3 - t2)/1e9:.2f} Gbp/s, resident {flat.aligned_bp/(t3-t2)/1e9:.2f} Gbp/s")
