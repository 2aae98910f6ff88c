# one rank of 8 on its shard with 16 threads of its own: the number of contig groups (bench.py's default follows the shard's size)
for g in 2 3 4 5 6; do for rep in 1 2; do python bench.py --as-rank-of 8 --threads 16 --cpu-contigs 0 --steps 40 --groups $g > gpurun_out/sw.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/sw.json").read().strip().split("\n")[-1])
print("rank of 8, groups $g run $rep: ms/step %.2f cpu %.1f waits %s" % (d["ms_per_step"], d["host"]["process_cpu_ms_per_step"], d["host"]["waits_per_step"]))
PY
done; done
