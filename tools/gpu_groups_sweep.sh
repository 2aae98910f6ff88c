cd /root/repo
mkdir -p gpurun_out
for cfg in "8 48" "12 48" "16 48" "16 64" "10 40"; do
  set -- $cfg
  timeout 300 python bench.py --groups $1 --threads $2 --cpu-contigs 0 --steps 20 --warmup 3 > gpurun_out/groups_$1_$2.json 2> gpurun_out/groups.err
  python - <<P
import json
j=json.load(open("gpurun_out/groups_$1_$2.json"))
print("groups $1 threads $2:", round(j["ms_per_step"],2), "ms", round(j["host"]["process_cpu_ms_per_step"],1), "CPU-ms", j["host"]["cfs_throttled_during_timed_steps"], j["pipeline_wall_ms_per_step"])
P
done
