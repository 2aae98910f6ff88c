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
export HS_ORDER_SCOPE=k2 HS_GROUP_TAPER=1 HS_SHARED_POOL=0
timeout 900 python -m pytest tests/test_gpu_dropin.py -x -q 2>&1 | tail -2
run twocalls HS_X=1
run fused HS_BENCH_FUSED=1
run fused_hostpile HS_BENCH_FUSED=1 HS_FUSED_HOST_PILEUP=1
run fused_t05 HS_BENCH_FUSED=1 HS_GROUP_TAPER=0.5
run fused_shared HS_BENCH_FUSED=1 HS_SHARED_POOL=1
EXTRA="--groups 10" run fused_g10 HS_BENCH_FUSED=1
run fused_phase1 HS_BENCH_FUSED=1 HS_ORDER_SCOPE=phase1
