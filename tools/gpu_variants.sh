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
timeout 600 python -m pytest tests/test_gpu_dropin.py -x -q -k "goldens or groups" 2>&1 | tail -2
run prio1 HS_X=1
run noprio1 HS_STREAM_PRIORITIES=0
run prio2 HS_X=1
run noprio2 HS_STREAM_PRIORITIES=0
EXTRA="--groups 10" run prio_g10 HS_X=1
EXTRA="--groups 12" run prio_g12 HS_X=1
EXTRA="--groups 6" run prio_g6 HS_X=1
run prio_shared HS_SHARED_POOL=1
