cd /root/repo
mkdir -p gpurun_out /tmp/f2f
python - <<P
import sys, subprocess, os, time
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import full_configs as fc
f,bp,n,t=fc.generate_files("C4","/tmp/f2f",None,workers=8)
cv=["hairsplitter_amd/bin/HS_call_variants",f["gfa"],f["reads"],f["sam"],"16","/tmp/f2f","/tmp/f2f/err.txt","0","0","/tmp/f2f/o.col","/tmp/f2f/o.vcf","0.33"]
base=dict(os.environ, HS_TIMING="1", HS_NO_DETACH="1")
subprocess.run(cv, env=base, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
er=min(float("%g" % float(open("/tmp/f2f/err.txt").read().strip())),0.15)
sr=["hairsplitter_amd/bin/HS_separate_reads","/tmp/f2f/o.col","16",str(er),"/tmp/f2f/no_ploidy","0","0.01","0","/tmp/f2f/o.gro","0"]
for devs in (None, "0,0", "0,0,0,0", "0,0,0,0,0,0,0,0", None, "0,0,0,0"):
    e=dict(base); 
    if devs: e["HS_DEVICES"]=devs
    t0=time.time(); r=subprocess.run(sr, env=e, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True); t1=time.time()
    m=[l for l in r.stderr.splitlines() if "main: hs_sr_run" in l or "entry to exit" in l]
    print("sr HS_DEVICES", devs, "wall %.2f" % (t1-t0), m, flush=True)
for devs in (None, "0,0", "0,0,0,0", None):
    e=dict(base); 
    if devs: e["HS_DEVICES"]=devs
    t0=time.time(); r=subprocess.run(cv, env=e, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True); t1=time.time()
    m=[l for l in r.stderr.splitlines() if "main: hs_cv_run_host" in l or "entry to exit" in l]
    print("cv HS_DEVICES", devs, "wall %.2f" % (t1-t0), m, flush=True)
P
