"""Throughput of the in-memory pipeline on other BASELINE shapes (diagnostic): C3 = tetraploid 200 kb at 40x ONT, C5 = 300 kb HiFi chunks."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hairsplitter_amd import api, synth
import bench

shape = sys.argv[1] if len(sys.argv) > 1 else "C3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if shape == "C3":
    contigs = [synth.make_contig(3, i, 200_000, 4, 0.01, 40, "ont") for i in range(n)]
elif shape == "C5":
    contigs = [synth.make_contig(5, i, 300_000, 2, 0.001, 30, "hifi") for i in range(n)]
else:
    contigs = [synth.make_contig(2, i, 100_000, 2, 0.01, 50, "ont") for i in range(n)]
torch.set_num_threads(1)
api.require_gpu()
pg = api.PipelineGroups(contigs, 8)
nt = min(64, 4 * bench.effective_cores())
for _ in range(2):
    cv, sr = pg.run(0.33, nt)
t0 = time.perf_counter(); K = 5
for _ in range(K):
    cv, sr = pg.run(0.33, nt)
dt = (time.perf_counter() - t0) / K
print(json.dumps({"shape": shape, "contigs": n, "aligned_bp": pg.aligned_bp, "ms_per_step": dt * 1e3, "Gbp_per_s": pg.aligned_bp / dt / 1e9,
                  "n_snps": cv["n_snps"], "n_cw": sr["n_cw_instances"], "rows_host": sr["n_graph_rows_host"], "wall": sr["wall_ms"]}))
