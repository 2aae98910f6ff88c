"""Known answers for bench.py's `labels_digest` (run on the GPU box, where the compiled reference is in oracle/_ref): for every listed job
`bench.py --gpus 1` runs WITH its reference gate, and the digest of the labels it held is kept only if that gate said "identical". The file
goes to tests/golden/bench_labels_digest.json; runs that cannot run the reference themselves (N > 1 ranks, --cpu-contigs 0) are checked
against it.   usage: python tools/make_bench_digests.py > gpurun_out/bench_labels_digest.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JOBS = [("C4", 0, ["--steps", "2", "--warmup", "1", "--f2f-runs", "1"]), ("C2", 6, ["--steps", "2", "--warmup", "1", "--f2f-runs", "1"])]
out = {}
for cfg, contigs, extra in JOBS:
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", cfg] + (["--contigs", str(contigs)] if contigs else []) + extra
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT)
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        sys.stderr.write("%s: bench failed (%d)\n%s\n" % (cfg, r.returncode, r.stderr.decode()[-1500:])); continue
    d = json.loads(lines[-1])
    p, g = d.get("parity") or {}, d.get("labels_digest")
    if not (p.get("checked") and p.get("identical") and p.get("kind", "reference") == "reference" and g):
        sys.stderr.write("%s: no reference verdict in this run: %r\n" % (cfg, p)); continue
    key = "%s:%d:%s" % (cfg, d["config"]["contigs"] if "contigs" in d["config"] else contigs, "default")
    out[key] = {"windows": g["windows"], "entries": g["entries"], "sum_crc32": g["sum_crc32"], "n_snps": g["n_snps"], "mean_distance_crc32": g["mean_distance_crc32"], "error_rate": g["error_rate"],
                "verified": "bench.py --gpus 1 on this job, reference gate: " + p.get("against", "")[:160]}
print(json.dumps(out, indent=1))
