cd /root/repo
mkdir -p gpurun_out /tmp/f2f
python - <<P
import sys, subprocess, os, time
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import full_configs as fc
f,bp,n,t=fc.generate_files("C3","/tmp/f2f",None,workers=8)
args=["hairsplitter_amd/bin/HS_call_variants","/tmp/f2f/assembly.gfa","/tmp/f2f/reads.fasta","/tmp/f2f/aln.sam","16","/tmp/f2f","/tmp/f2f/err.txt","0","0","/tmp/f2f/o.col","/tmp/f2f/o.vcf","0.33"]
def run(tag, env):
    t0=time.time()
    r=subprocess.run(args, env=env, stdout=subprocess.DEVNULL, stderr=open("gpurun_out/f2f3_%s.err"%tag,"w"))
    t1=time.time()
    print(tag, "start %.1f end %.1f wall %.1f ms" % (t0*1e3, t1*1e3, (t1-t0)*1e3), flush=True)
e=dict(os.environ, HS_TIMING="1")
run("warmup", e); run("detached", e); run("nodetach", dict(e, HS_NO_DETACH="1"))
P
for t in detached nodetach; do echo == $t; grep "stamp\|entry to exit" gpurun_out/f2f3_$t.err | cut -c1-200; done
