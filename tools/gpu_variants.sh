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
export HS_BENCH_NO_PROBE=1
run base1 HS_X=1
EXTRA="--threads 64" run t64 HS_X=1
EXTRA="--threads 96" run t96 HS_X=1
EXTRA="--threads 32" run t32 HS_X=1
run shared HS_SHARED_POOL=1
run shared24 HS_SHARED_POOL=1 HS_POOL_THREADS=24
run taper05 HS_GROUP_TAPER=0.5
run taper03 HS_GROUP_TAPER=0.3
run base2 HS_X=1
