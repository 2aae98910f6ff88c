# HS_TIMING breakdown of the two drop-ins on the whole C4 job
cd /root/repo
mkdir -p gpurun_out /tmp/f2f
python - <<P
import sys, subprocess, os, time
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import full_configs as fc
f,bp,n,t=fc.generate_files("C4","/tmp/f2f",None,workers=8)
cv=["hairsplitter_amd/bin/HS_call_variants",f["gfa"],f["reads"],f["sam"],"16","/tmp/f2f","/tmp/f2f/err.txt","0","0","/tmp/f2f/o.col","/tmp/f2f/o.vcf","0.33"]
e=dict(os.environ, HS_TIMING="1", HS_NO_DETACH="1", **({"HS_EXIT_PROBE": "1"} if os.environ.get("PROBE") else {}))
for tag in ("warm","timed"):
    t0=time.time(); subprocess.run(cv, env=e, stdout=subprocess.DEVNULL, stderr=open("gpurun_out/f2fc4_cv_%s.err"%tag,"w")); t1=time.time()
    er=min(float("%g" % float(open("/tmp/f2f/err.txt").read().strip())),0.15)
    sr=["hairsplitter_amd/bin/HS_separate_reads","/tmp/f2f/o.col","16",str(er),"/tmp/f2f/no_ploidy","0","0.01","0","/tmp/f2f/o.gro","0"]
    subprocess.run(sr, env=e, stdout=subprocess.DEVNULL, stderr=open("gpurun_out/f2fc4_sr_%s.err"%tag,"w")); t2=time.time()
    print(tag, "cv %.2f s sr %.2f s" % (t1-t0, t2-t1), flush=True)
P
grep "main:\|exit probe" gpurun_out/f2fc4_cv_timed.err | cut -c1-260
grep "main:\|exit probe" gpurun_out/f2fc4_sr_timed.err | cut -c1-260
