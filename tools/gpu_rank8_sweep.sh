cd /root/repo
mkdir -p gpurun_out
for cfg in "6 1" "12 2" "24 4" "48 8"; do
  set -- $cfg
  for spin in 0 1; do
    if [ $spin = 1 ]; then export HS_SPIN_WAIT=1; else unset HS_SPIN_WAIT; fi
    timeout 300 python bench.py --as-rank-of 8 --threads $1 --groups $2 --cpu-contigs 0 --steps 30 --warmup 3 > gpurun_out/rank8_t$1_g$2_s$spin.json 2> gpurun_out/rank8.err
    python - <<P
import json
j=json.load(open("gpurun_out/rank8_t$1_g$2_s$spin.json"))
print("threads $1 groups $2 spin $spin:", round(j["ms_per_step"],2), "ms", round(j["host"]["process_cpu_ms_per_step"],1), "CPU-ms", round(j["value"]/1e9,1), "Gbp/s per rank", j["host"]["waits_per_step"], "waits")
P
  done
done
