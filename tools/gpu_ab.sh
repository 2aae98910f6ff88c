# A/B of an environment switch on the default bench command, alternating, same box: bash tools/gpu_ab.sh VAR [rounds] [value]
cd /root/repo
mkdir -p gpurun_out
V=${1:-HS_K2_PLAIN}
N=${2:-3}
VAL=${3:-1}
for i in $(seq 1 $N); do
  for mode in off on; do
    if [ $mode = on ]; then export $V=$VAL; else unset $V; fi
    HS_BENCH_NO_PROBE=1 timeout 300 python bench.py --cpu-contigs 0 > gpurun_out/ab_${mode}.json 2> gpurun_out/ab_${mode}.err
    python - <<P
import json
j=json.load(open('gpurun_out/ab_${mode}.json'))
print('$V=$VAL', '$mode', round(j['ms_per_step'],2), 'ms/step', j['host']['process_cpu_ms_per_step'], 'CPU-ms', j['host']['waits_per_step'], 'waits')
P
  done
done
