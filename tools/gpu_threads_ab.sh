# the default bench command at several host-thread counts, alternating, same box: bash tools/gpu_threads_ab.sh "32 48 64" [rounds]
cd /root/repo
mkdir -p gpurun_out
TS=${1:-"32 48 64"}
N=${2:-3}
rm -f gpurun_out/tab.txt
for i in $(seq 1 $N); do
  for t in $TS; do
    HS_BENCH_NO_PROBE=1 timeout 300 python bench.py --cpu-contigs 0 --steps 30 --threads $t > gpurun_out/tab_$t.json 2> gpurun_out/tab_$t.err
    python - >> gpurun_out/tab.txt <<P
import json
j=json.load(open('gpurun_out/tab_$t.json'))
print($t, round(j['ms_per_step'],3), round(j['host']['process_cpu_ms_per_step'],1), j['config'].get('host_threads_per_rank'), j['config'].get('groups_per_gpu'))
P
  done
done
python - <<P
import statistics as st, collections
r=collections.defaultdict(list); c=collections.defaultdict(list)
for l in open('gpurun_out/tab.txt'):
    a,ms,cpu,th,g=l.split(); r[a].append(float(ms)); c[a].append(float(cpu))
for a in r:
    print('threads', a, 'ms/step mean %.2f min %.2f  CPU-ms mean %.1f' % (st.mean(r[a]), min(r[a]), st.mean(c[a])), ' '.join('%.2f'%x for x in r[a]))
P
