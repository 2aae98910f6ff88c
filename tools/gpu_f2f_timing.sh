cd /root/repo
mkdir -p gpurun_out /tmp/f2f
python - <<P
import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import full_configs as fc
f,bp,n,t=fc.generate_files("C4","/tmp/f2f",None,workers=8)
print(f,bp,n,t)
P
for i in 1 2; do
HS_TIMING=1 HS_NO_DETACH=1 hairsplitter_amd/bin/HS_call_variants /tmp/f2f/assembly.gfa /tmp/f2f/reads.fasta /tmp/f2f/aln.sam 16 /tmp/f2f /tmp/f2f/err.txt 0 0 /tmp/f2f/o.col /tmp/f2f/o.vcf 0.33 > /tmp/f2f/out.txt 2> gpurun_out/f2f_timing_cv_$i.err
done
grep -v "^\[hs timing\]   " gpurun_out/f2f_timing_cv_2.err | cut -c1-400 | tail -30
