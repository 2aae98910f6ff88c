# drop-ins on the whole C4 job with and without glibc's THP-backed malloc
cd /root/repo
mkdir -p gpurun_out /tmp/f2f
cat /sys/kernel/mm/transparent_hugepage/enabled
python - <<P
import sys, subprocess, os, time
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import full_configs as fc
f,bp,n,t=fc.generate_files("C4","/tmp/f2f",None,workers=8)
cv=["hairsplitter_amd/bin/HS_call_variants",f["gfa"],f["reads"],f["sam"],"16","/tmp/f2f","/tmp/f2f/err.txt","0","0","/tmp/f2f/o.col","/tmp/f2f/o.vcf","0.33"]
def pair(tag, env):
    t0=time.time(); subprocess.run(cv, env=env, stdout=subprocess.DEVNULL, stderr=open("gpurun_out/thp_cv_%s.err"%tag,"w")); t1=time.time()
    er=min(float("%g" % float(open("/tmp/f2f/err.txt").read().strip())),0.15)
    sr=["hairsplitter_amd/bin/HS_separate_reads","/tmp/f2f/o.col","16",str(er),"/tmp/f2f/no_ploidy","0","0.01","0","/tmp/f2f/o.gro","0"]
    subprocess.run(sr, env=env, stdout=subprocess.DEVNULL, stderr=open("gpurun_out/thp_sr_%s.err"%tag,"w")); t2=time.time()
    print(tag, "cv %.2f s sr %.2f s total %.2f" % (t1-t0, t2-t1, t2-t0), flush=True)
base=dict(os.environ, HS_TIMING="1", HS_NO_DETACH="1")
thp=dict(base, GLIBC_TUNABLES="glibc.malloc.hugetlb=1")
pair("warm", base)
for i in range(3):
    pair("base%d"%i, base); pair("thp%d"%i, thp)
P
grep "main:" gpurun_out/thp_cv_base2.err | cut -c1-120
grep "main:" gpurun_out/thp_cv_thp2.err | cut -c1-120
