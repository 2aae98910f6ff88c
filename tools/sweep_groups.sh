for g in 6 8 10 12; do for rep in 1 2; do python bench.py --steps 20 --warmup 3 --cpu-contigs 0 --no-f2f-job --groups $g > gpurun_out/sw.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/sw.json").read().strip().split("\n")[-1])
print("groups $g run $rep: ms/step %.2f cpu %.1f" % (d["ms_per_step"], d["host"]["process_cpu_ms_per_step"]))
PY
done; done
for t in 0.3 0.7 1.0; do python bench.py --steps 20 --warmup 3 --cpu-contigs 0 --no-f2f-job > gpurun_out/sw.json 2>/dev/null; HS_GROUP_TAPER=$t python bench.py --steps 20 --warmup 3 --cpu-contigs 0 --no-f2f-job > gpurun_out/sw2.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/sw.json").read().strip().split("\n")[-1]); e=json.loads(open("gpurun_out/sw2.json").read().strip().split("\n")[-1])
print("taper default %.2f ms | taper $t: %.2f ms" % (d["ms_per_step"], e["ms_per_step"]))
PY
done
