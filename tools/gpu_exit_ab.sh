# the whole C4 job through HS_call_variants (HS_NO_DETACH=1), wall time to the full exit: as it is / with HS_EXIT_RESET=1
cd /root/repo
mkdir -p gpurun_out /tmp/f2f
python - <<P
import sys, subprocess, os, time
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import full_configs as fc
f,bp,n,t=fc.generate_files("C4","/tmp/f2f",None,workers=8)
cv=["hairsplitter_amd/bin/HS_call_variants",f["gfa"],f["reads"],f["sam"],"16","/tmp/f2f","/tmp/f2f/err.txt","0","0","/tmp/f2f/o.col","/tmp/f2f/o.vcf","0.33"]
for tag, extra in [("as it is", {})] * 3 + [("HS_EXIT_LEAK", {"HS_EXIT_LEAK": "1"})] * 3 + [("as it is", {})] * 3:
    for rep in range(1):
        e=dict(os.environ, HS_NO_DETACH="1", HS_TIMING="1", **extra)
        t0=time.time(); r=subprocess.run(cv, env=e, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE); t1=time.time()
        stamps=[float(l.split(" at ")[1].split()[0]) for l in r.stderr.decode().splitlines() if "stamp leaving" in l]
        lines=["exit took %.0f ms" % (t1*1e3 - stamps[-1])] if stamps else []
        lines+=[l for l in r.stderr.decode().splitlines() if "entry to exit" in l or "hipDeviceReset" in l or "main: load" in l or "exit:" in l]
        print("%-22s rc %d  %.3f s   %s" % (tag, r.returncode, t1-t0, " | ".join(x.replace("[hs timing] ","") for x in lines)), flush=True)
P
