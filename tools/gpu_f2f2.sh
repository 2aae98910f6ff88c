cd /root/repo
mkdir -p /tmp/c4job
python3 - <<'P'
import sys, time
sys.path.insert(0, '/root/repo')
from hairsplitter_amd import synth
cs, f = synth.generate_job("C4", range(500), workers=14, outdir="/tmp/c4job")
P
cd /tmp/c4job
python3 - <<'P'
import time, subprocess, os
cv=["/root/repo/hairsplitter_amd/bin/HS_call_variants","assembly.gfa","reads.fasta","aln.sam","16",".","err.txt","0","0","out.col","out.vcf","0.33"]
for rep in range(3):
    for fresh in (False, True):
        if fresh:
            for f in ("out.col","out.vcf","err.txt"):
                if os.path.exists(f): os.remove(f)
        t=time.time(); r=subprocess.run(cv,stdout=subprocess.DEVNULL,stderr=subprocess.PIPE,env=dict(os.environ,HS_TIMING="1")); t1=time.time()
        laps=[l for l in r.stderr.decode().splitlines() if "main:" in l]
        tot=sum(float(l.split()[-2]) for l in laps)
        print("fresh" if fresh else "overwrite", "wall %.3f laps %.3f" % (t1-t, tot/1e3), [l.split("main: ")[1] for l in laps][:1])
P
