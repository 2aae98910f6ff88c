cd /root/repo
mkdir -p gpurun_out
run() {
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 40 --warmup 3 --cpu-contigs 0 > gpurun_out/env2_${name}.json 2> gpurun_out/env2_${name}.err
  python - <<P
import json
try:
    j=json.load(open('gpurun_out/env2_${name}.json'))
    print('${name}', round(j['ms_per_step'],2), 'ms/step', round(j['host']['process_cpu_ms_per_step'],1), 'cpu-ms', j['host']['cfs_throttled_during_timed_steps'])
except Exception as e:
    print('${name}', 'failed', e)
P
}
run default HS_X=1
run activewait0 ROC_ACTIVE_WAIT_TIMEOUT=0
run nointerrupt HSA_ENABLE_INTERRUPT=0
run both ROC_ACTIVE_WAIT_TIMEOUT=0 HSA_ENABLE_INTERRUPT=0
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq8_aw0 GPU_MAX_HW_QUEUES=8 ROC_ACTIVE_WAIT_TIMEOUT=0
run default2 HS_X=1
run activewait0_2 ROC_ACTIVE_WAIT_TIMEOUT=0
