cd /root/repo
mkdir -p gpurun_out
HS_COPY_STATS=1 timeout 600 python bench.py --steps 10 --warmup 2 --cpu-contigs 0 > gpurun_out/copy_stats.json 2> gpurun_out/copy_stats.err
grep "hs copies" gpurun_out/copy_stats.err
