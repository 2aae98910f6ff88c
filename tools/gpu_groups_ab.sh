# the default bench command at several group counts, alternating, same box: bash tools/gpu_groups_ab.sh "8 10 12" [rounds]
cd /root/repo
mkdir -p gpurun_out
GS=${1:-"8 10 12"}
N=${2:-2}
for i in $(seq 1 $N); do
  for g in $GS; do
    HS_BENCH_NO_PROBE=1 timeout 300 python bench.py --cpu-contigs 0 --groups $g > gpurun_out/gab_$g.json 2> gpurun_out/gab_$g.err
    python - <<P
import json
j=json.load(open('gpurun_out/gab_$g.json'))
print('groups', $g, round(j['ms_per_step'],2), 'ms/step', round(j['host']['process_cpu_ms_per_step'],1), 'CPU-ms', j['host']['waits_per_step'], 'waits')
P
  done
done
