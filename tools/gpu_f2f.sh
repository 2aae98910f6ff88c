cd /root/repo
mkdir -p gpurun_out /tmp/c4job
python3 - <<'P'
import sys, time
sys.path.insert(0, '/root/repo')
from hairsplitter_amd import synth
t=time.time(); cs, f = synth.generate_job("C4", range(500), workers=14, outdir="/tmp/c4job"); print("gen+write", time.time()-t)
P
cd /tmp/c4job
for rep in 1 2; do
/usr/bin/env HS_TIMING=1 /root/repo/hairsplitter_amd/bin/HS_call_variants assembly.gfa reads.fasta aln.sam 16 . err.txt 0 0 out.col out.vcf 0.33 2> cv.err > /dev/null
E=$(python3 -c "print(min(float('%g' % float(open('err.txt').read())), 0.15))")
/usr/bin/env HS_TIMING=1 /root/repo/hairsplitter_amd/bin/HS_separate_reads out.col 16 $E nop 0 0.01 0 out.gro 0 2> sr.err > /dev/null
done
grep "main:\|load:" cv.err; grep "main:" sr.err
python3 - <<'P'
import time, subprocess
t=time.time(); subprocess.run(["/root/repo/hairsplitter_amd/bin/HS_call_variants","assembly.gfa","reads.fasta","aln.sam","16",".","err.txt","0","0","out.col","out.vcf","0.33"],stdout=subprocess.DEVNULL); t1=time.time()
subprocess.run(["/root/repo/hairsplitter_amd/bin/HS_separate_reads","out.col","16","0.0549","nop","0","0.01","0","out.gro","0"],stdout=subprocess.DEVNULL); t2=time.time()
print("wall cv %.3f sr %.3f total %.3f" % (t1-t, t2-t1, t2-t))
P
ls -la /tmp/c4job | head
