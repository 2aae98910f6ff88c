cd /root/repo
mkdir -p gpurun_out
HS_TIMING=cpu timeout 900 python bench.py --steps 3 --warmup 0 --cpu-contigs 0 "$@" > gpurun_out/laps_bench.json 2> gpurun_out/laps.err
python - <<'P'
import re, collections
tot=collections.defaultdict(lambda: [0.0,0.0,0])
for l in open('gpurun_out/laps.err'):
    m=re.match(r"\[hs timing\] (.+?) laps \(ms\):(.*)", l)
    if not m: continue
    tag=m.group(1)
    for name,w,c in re.findall(r" ([a-z0-9_ +]+?) ([0-9.]+)/([0-9.]+)", m.group(2)):
        t=tot[(tag,name.strip())]; t[0]+=float(w); t[1]+=float(c); t[2]+=1
steps=11
for (tag,name),(w,c,n) in sorted(tot.items(), key=lambda kv:-kv[1][0]):
    print("%-10s %-22s wall/step %8.2f  cpu/step %8.2f  (n=%d)"%(tag,name,w/steps,c/steps,n))
P
grep -c "laps" gpurun_out/laps.err
