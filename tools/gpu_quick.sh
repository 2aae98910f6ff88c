cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -15
timeout 900 python bench.py --steps 10 --warmup 2 --cpu-contigs 0 > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err
tail -3 gpurun_out/quick_bench.err
python - <<'P'
import json
j=json.load(open('gpurun_out/quick_bench.json'))
print("G bp/s %.2f  ms/step %.2f  cpu ms/step %.1f" % (j['value']/1e9, j['ms_per_step'], j['host']['process_cpu_ms_per_step']))
print("roofline", j['roofline']['kernel'], round(j['roofline']['frac'],4), "whole", round(j['roofline']['whole_path']['frac_vs_kernel_time'],4), round(j['roofline']['whole_path']['frac_vs_step_time'],4))
for k,v in j['kernels'].items(): print("  %-28s %8.3f ms/step %5.1f launches  %8.1f GB/s" % (k, v['ms_per_step'], v['launches_per_step'], v['achieved_GBs']))
print(j['pipeline_wall_ms_per_step'])
P
