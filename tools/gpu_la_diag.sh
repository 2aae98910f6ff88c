cd /root/repo
mkdir -p gpurun_out
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-contigs 0 --groups 1 > gpurun_out/la_diag.json 2> gpurun_out/la_diag.err
grep "hs la" gpurun_out/la_diag.err | tail -2
