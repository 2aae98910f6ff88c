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
run sparse1 HS_X=1
run dense1 HS_BENCH_DENSE_LABELS=1
run sparse2 HS_X=1
run dense2 HS_BENCH_DENSE_LABELS=1
run twocalls HS_BENCH_TWO_CALLS=1
run phase1 HS_ORDER_SCOPE=phase1
run k2 HS_ORDER_SCOPE=k2
EXTRA="--groups 10" run g10 HS_X=1
EXTRA="--groups 6" run g6 HS_X=1
