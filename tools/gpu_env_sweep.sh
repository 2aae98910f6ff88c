# GPU suite (quick subset) + bench under a few runtime settings
cd /root/repo
mkdir -p gpurun_out
TAG=${1:-r03e}
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_full_configs.py -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1
tail -3 gpurun_out/${TAG}_pytest.log
run() {
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 40 --warmup 3 --cpu-contigs 0 > gpurun_out/${TAG}_bench_${name}.json 2> gpurun_out/${TAG}_bench_${name}.err
  python - <<P
import json
j=json.load(open('gpurun_out/${TAG}_bench_${name}.json'))
print('${name}', round(j['ms_per_step'],2), 'ms/step', round(j['host']['process_cpu_ms_per_step'],1), 'cpu-ms', j['host']['cfs_throttled_during_timed_steps'])
P
}
run default HS_X=1
run nosdma HSA_ENABLE_SDMA=0
run activewait0 ROC_ACTIVE_WAIT_TIMEOUT=0
run activewait200 ROC_ACTIVE_WAIT_TIMEOUT=200
run blocking HS_BLOCKING_WAIT=1
run spin HS_SPIN_WAIT=1
run nostats HS_NO_KERNEL_STATS=1
