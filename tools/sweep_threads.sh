# host threads of the default job (bench.py --threads: the worker pools of the contig groups share them; default 3 x the cores)
for t in ${THREADS_LIST:-24 32 48 64 96}; do for rep in 1 2; do python bench.py --steps 20 --warmup 3 --cpu-contigs 0 --no-f2f-job --threads $t > gpurun_out/sw.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/sw.json").read().strip().split("\n")[-1])
print("threads $t run $rep: ms/step %.2f cpu %.1f" % (d["ms_per_step"], d["host"]["process_cpu_ms_per_step"]))
PY
done; done
