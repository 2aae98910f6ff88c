# bench variants (diagnostic): one line each
cd /root/repo
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --cpu-contigs 0 --steps 20 --warmup 3 $EXTRA > gpurun_out/var_$tag.json 2> gpurun_out/var_$tag.err; python - <<P
import json
try:
    j=json.load(open('gpurun_out/var_$tag.json'))
    print('$tag', round(j['ms_per_step'],2), 'ms', round(j['host']['process_cpu_ms_per_step'],1), 'CPU-ms', j['host']['waits_per_step'], 'waits', j['pipeline_wall_ms_per_step'])
except Exception as e: print('$tag failed', e)
P
}
run base HS_X=1
run fused HS_BENCH_FUSED=1
EXTRA="--groups 12" run g12 HS_X=1
EXTRA="--groups 16" run g16 HS_X=1
EXTRA="--groups 12" run g12fused HS_BENCH_FUSED=1
EXTRA="--groups 16" run g16fused HS_BENCH_FUSED=1
EXTRA="--groups 6" run g6 HS_X=1
EXTRA="--threads 32" run t32 HS_X=1
EXTRA="--threads 64" run t64 HS_X=1
