cd /root/repo
mkdir -p gpurun_out
for v in 0 1; do
  if [ $v = 1 ]; then export HS_NO_KERNEL_STATS=1; else unset HS_NO_KERNEL_STATS; fi
  timeout 300 python bench.py --cpu-contigs 0 --steps 20 --warmup 3 > gpurun_out/nostats_$v.json 2> gpurun_out/nostats.err
  python - <<P
import json
j=json.load(open("gpurun_out/nostats_$v.json"))
print("no kernel stats $v:", round(j["ms_per_step"],2), "ms", round(j["host"]["process_cpu_ms_per_step"],1), "CPU-ms")
P
done
