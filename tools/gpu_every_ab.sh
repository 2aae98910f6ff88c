# the default bench command with timing events on every n-th step, alternating over a list of n, same box: bash tools/gpu_every_ab.sh "1 4 1000" [rounds]
cd /root/repo
mkdir -p gpurun_out
NS=${1:-"1 4 1000"}
N=${2:-3}
for i in $(seq 1 $N); do
  for n in $NS; do
    HS_BENCH_STATS_EVERY=$n HS_BENCH_NO_PROBE=1 timeout 300 python bench.py --cpu-contigs 0 --steps 24 > gpurun_out/every_$n.json 2> gpurun_out/every_$n.err
    python - <<P
import json
j=json.load(open('gpurun_out/every_$n.json'))
st=j['step_ms']
print('every', $n, 'mean %.2f' % (sum(st)/len(st)), 'median %.2f' % sorted(st)[len(st)//2], ' '.join('%.1f'%x for x in st))
P
  done
done
