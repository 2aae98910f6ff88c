cd /root/repo
mkdir -p gpurun_out /tmp/f2f
python - <<P
import sys, subprocess, os, time
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import full_configs as fc
f,bp,n,t=fc.generate_files("C4","/tmp/f2f",None,workers=8)
args=["hairsplitter_amd/bin/HS_call_variants","/tmp/f2f/assembly.gfa","/tmp/f2f/reads.fasta","/tmp/f2f/aln.sam","16","/tmp/f2f","/tmp/f2f/err.txt","0","0","/tmp/f2f/o.col","/tmp/f2f/o.vcf","0.33"]
def run(tag, env):
    t0=time.perf_counter()
    r=subprocess.run(args, env=env, stdout=subprocess.DEVNULL, stderr=open("gpurun_out/f2f2_%s.err"%tag,"w"))
    print(tag, round(time.perf_counter()-t0,3), flush=True)
e=dict(os.environ, HS_TIMING="1")
run("cold", e); run("warm", e); run("warm_nodetach", dict(e, HS_NO_DETACH="1"))
from hairsplitter_amd import api, synth
api.require_gpu()
c=[synth.make_contig(3,k,20000,2,0.01,30,"ont") for k in range(8)]
pg=api.PipelineGroups(c,2); pg.run(); 
run("parent_has_gpu", e); run("parent_has_gpu_nodetach", dict(e, HS_NO_DETACH="1"))
import torch
torch.zeros(1,device="cuda")
run("parent_has_torch", e)
P
for t in warm parent_has_gpu parent_has_torch; do echo == $t; grep -v "^\[hs timing\]   " gpurun_out/f2f2_$t.err | grep "main:" | cut -c1-200; done
