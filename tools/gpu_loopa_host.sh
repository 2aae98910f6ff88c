cd /root/repo
mkdir -p gpurun_out
HS_TIMING_AB=1 timeout 300 python bench.py --steps 2 --warmup 1 --cpu-contigs 0 > gpurun_out/loopa_host.json 2> gpurun_out/loopa_host.err
grep "loop A:" gpurun_out/loopa_host.err | tail -500 | awk '{c+=$5; p+=$7; cmp+=$9; aug+=$11; t+=$13; b+=$16; gsub(")","",$18); a+=$18} END {print "contigs", NR, "candidates", c, "partitions", p, "comparisons", cmp, "augmentations", aug, "us total", t, "build", b, "augment", a}'
grep "loop A:" gpurun_out/loopa_host.err | tail -2
