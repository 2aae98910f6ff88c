cd /root/repo
mkdir -p gpurun_out
HS_TIMING=1 timeout 300 python bench.py --steps 3 --warmup 1 --cpu-contigs 0 > gpurun_out/laps2.json 2> gpurun_out/laps2.err
grep "cv range\|cv glue laps\|sr laps\|sr: planes" gpurun_out/laps2.err | tail -32 | cut -c1-260
