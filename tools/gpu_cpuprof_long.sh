# longer sampling profile of the host side (600 steps) for line-level reading: bash tools/gpu_cpuprof_long.sh
cd /root/repo
mkdir -p gpurun_out
HS_CPU_PROFILE=/root/repo/gpurun_out/cpu_prof_long.txt HS_BENCH_NO_PROBE=1 timeout 900 python bench.py --steps 600 --warmup 3 --cpu-contigs 0 > gpurun_out/cpu_prof_long_bench.json 2> gpurun_out/cpu_prof_long.err
python tools/cpuprof_report.py gpurun_out/cpu_prof_long.txt 120 > gpurun_out/cpu_profile_long_top.txt 2>&1
rm -f gpurun_out/cpu_prof_long.txt
head -5 gpurun_out/cpu_profile_long_top.txt
