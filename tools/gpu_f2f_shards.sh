cd /root/repo
python3 - <<P
import sys, subprocess, os, time
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import full_configs as fc
os.makedirs("/tmp/f2f", exist_ok=True)
f,bp,n,t=fc.generate_files("C4","/tmp/f2f",None,workers=8)
cv=["hairsplitter_amd/bin/HS_call_variants",f["gfa"],f["reads"],f["sam"],"16","/tmp/f2f","/tmp/f2f/err.txt","0","0","/tmp/f2f/o.col","/tmp/f2f/o.vcf","0.33"]
for devs in (None, "0,0", "0,0,0", None, "0,0"):
    e=dict(os.environ, HS_NO_DETACH="1")
    if devs: e["HS_DEVICES"]=devs
    ts=[]
    for rep in range(3):
        t0=time.time(); subprocess.run(cv, env=e, stdout=subprocess.DEVNULL, check=True); ts.append(time.time()-t0)
    print("HS_DEVICES=%s: %s" % (devs or "one", " ".join("%.3f"%x for x in ts)), flush=True)
P
