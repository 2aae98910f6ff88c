cd /root/repo
mkdir -p gpurun_out
for v in 0 1; do
  if [ $v = 1 ]; then export HS_LOOP_B_PAIRS_ON_DEVICE=1; else unset HS_LOOP_B_PAIRS_ON_DEVICE; fi
  HS_TIMING=1 timeout 600 python bench.py --steps 10 --warmup 2 --cpu-contigs 0 > gpurun_out/loopb_$v.json 2> gpurun_out/loopb_$v.err
  python - <<P
import json
j=json.load(open("gpurun_out/loopb_$v.json"))
print("pairs on device $v:", round(j["ms_per_step"],2), "ms", round(j["host"]["process_cpu_ms_per_step"],1), "CPU-ms", j["host"]["waits_per_step"], "waits")
P
  grep "loop B" gpurun_out/loopb_$v.err | tail -2
done
