# A/B timing of a code change on one box: the default bench command alternating between two builds of the library.
# usage: bash tools/gpu_ab_lib.sh <lib A> <lib B> [rounds] [steps]   (copies of libhairsplitter_hip.so, e.g. gpurun_out/lib_a.so)
cd /root/repo
mkdir -p gpurun_out
A=$1; B=$2; N=${3:-5}; STEPS=${4:-30}
rm -f gpurun_out/ablib.txt
for i in $(seq 1 $N); do
  for arm in A B; do
    if [ $arm = A ]; then L=$A; else L=$B; fi
    HS_LIB_AB=$(readlink -f $L) HS_BENCH_NO_PROBE=1 timeout 300 python bench.py --cpu-contigs 0 --steps $STEPS > gpurun_out/ablib_$arm.json 2> gpurun_out/ablib_$arm.err
    python - >> gpurun_out/ablib.txt <<P
import json
j=json.load(open('gpurun_out/ablib_$arm.json'))
print('$arm', round(j['ms_per_step'],3), round(j['host']['process_cpu_ms_per_step'],1))
P
  done
done
python - <<P
import statistics as st
r={'A':[], 'B':[]}; c={'A':[], 'B':[]}
for l in open('gpurun_out/ablib.txt'):
    a,ms,cpu=l.split(); r[a].append(float(ms)); c[a].append(float(cpu))
for a in 'AB':
    print(a, 'ms/step mean %.2f median %.2f min %.2f (n=%d)  CPU-ms mean %.1f' % (st.mean(r[a]), st.median(r[a]), min(r[a]), len(r[a]), st.mean(c[a])), ' '.join('%.2f'%x for x in r[a]))
P
