cd /root/repo
mkdir -p gpurun_out
TAG=${1:-cp}
HS_CPU_PROFILE=/root/repo/gpurun_out/${TAG}_cpu_prof.txt timeout 900 python bench.py --steps 150 --warmup 3 --cpu-contigs 0 > gpurun_out/${TAG}_bench_prof.json 2> gpurun_out/${TAG}_bench_prof.err
python tools/cpuprof_report.py gpurun_out/${TAG}_cpu_prof.txt 70 > gpurun_out/${TAG}_cpu_profile_top.txt 2>&1
rm -f gpurun_out/${TAG}_cpu_prof.txt
head -90 gpurun_out/${TAG}_cpu_profile_top.txt
