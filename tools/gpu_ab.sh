# bench three times (noise), then the goldens
cd /root/repo
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 300 python bench.py --cpu-contigs 0 --steps 30 --warmup 3 > gpurun_out/ab_$i.json 2> gpurun_out/ab.err
  python - <<P
import json
j=json.load(open("gpurun_out/ab_$i.json"))
print(round(j["ms_per_step"],2), "ms", round(j["host"]["process_cpu_ms_per_step"],1), "CPU-ms", j["host"]["cfs_throttled_during_timed_steps"])
P
done
timeout 600 python -m pytest tests/test_gpu_dropin.py -m gpu -x -q -k "goldens or pipeline" 2>&1 | tail -2
