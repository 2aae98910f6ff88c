# per-contig loop-A times of one step of the default job (HS_TIMING_AB): the distribution and where the long ones sit
cd /root/repo
mkdir -p gpurun_out
HS_TIMING_AB=1 HS_BENCH_NO_PROBE=1 timeout 600 python bench.py --steps 1 --warmup 0 --cpu-contigs 0 > gpurun_out/loopa_dist.json 2> gpurun_out/loopa_dist.err
python - <<'P'
import re
rows=[]
for l in open('gpurun_out/loopa_dist.err'):
    m=re.search(r'loop A: (\d+) candidates, (\d+) partitions, (\d+) comparisons, (\d+) augmentations; (\d+) us', l)
    if m: rows.append(tuple(int(x) for x in m.groups()))
n=500
last=rows[-n:]
us=[r[4] for r in last]
print(len(rows),'lines; last step:', len(last),'contigs, sum %.1f ms, max %.2f ms, top10 %s' % (sum(us)/1e3, max(us)/1e3, sorted(us)[-10:]))
big=sorted(last,key=lambda r:-r[4])[:5]
print('largest:', big)
P
