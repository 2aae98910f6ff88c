cd /root/repo
mkdir -p gpurun_out /tmp/f2f
python - <<P
import sys, subprocess, os, time
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import full_configs as fc
f,bp,n,t=fc.generate_files("C4","/tmp/f2f",None,workers=8)
cv=["hairsplitter_amd/bin/HS_call_variants",f["gfa"],f["reads"],f["sam"],"16","/tmp/f2f","/tmp/f2f/err.txt","0","0","/tmp/f2f/o.col","/tmp/f2f/o.vcf","0.33"]
e=dict(os.environ, HS_NO_DETACH="1")
subprocess.run(cv, env=e, stdout=subprocess.DEVNULL)
subprocess.run(cv, env=dict(e, HS_CPU_PROFILE="/root/repo/gpurun_out/cv_prof.txt"), stdout=subprocess.DEVNULL)
er=min(float("%g" % float(open("/tmp/f2f/err.txt").read().strip())),0.15)
sr=["hairsplitter_amd/bin/HS_separate_reads","/tmp/f2f/o.col","16",str(er),"/tmp/f2f/no_ploidy","0","0.01","0","/tmp/f2f/o.gro","0"]
subprocess.run(sr, env=dict(e, HS_CPU_PROFILE="/root/repo/gpurun_out/sr_prof.txt"), stdout=subprocess.DEVNULL)
P
python tools/cpuprof_report.py gpurun_out/cv_prof.txt 40 > gpurun_out/cv_prof_top.txt 2>&1
python tools/cpuprof_report.py gpurun_out/sr_prof.txt 25 > gpurun_out/sr_prof_top.txt 2>&1
head -45 gpurun_out/cv_prof_top.txt
