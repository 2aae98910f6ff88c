cd /root/repo
python3 - <<'P'
import time, subprocess, os, tempfile, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import golden_util as gu
cv="/root/repo/hairsplitter_amd/bin/HS_call_variants"; sr="/root/repo/hairsplitter_amd/bin/HS_separate_reads"
def t(cmd, n=5, env=None):
    best=1e9
    for _ in range(n):
        a=time.time(); subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env); best=min(best,time.time()-a)
    return best
print("usage-only (ld.so + static init):", round(t([cv,"--version"]),3))
with tempfile.TemporaryDirectory() as td:
    meta=gu.unpack("dip20k", td)
    args=[cv, td+"/assembly.gfa", td+"/reads.fasta", td+"/aln.sam","4",td,td+"/e.txt","0","0",td+"/o.col",td+"/o.vcf","0.33"]
    print("tiny job cv:", round(t(args),3))
    print("tiny job sr:", round(t([sr, td+"/o.col","4","0.05",td+"/nop","0","0.01","0",td+"/o.gro","0"]),3))
    e=dict(os.environ, HS_TIMING="1")
    r=subprocess.run(args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=e); print(r.stderr.decode()[-900:])
P
