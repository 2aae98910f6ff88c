# host CPU profile over many steps + intrinsic kernel times (one contig group) under rocprofv3
cd /root/repo
mkdir -p gpurun_out
TAG=${1:-r03d}
HS_CPU_PROFILE=/root/repo/gpurun_out/${TAG}_cpu_prof.txt timeout 900 python bench.py --steps 150 --warmup 3 --cpu-contigs 0 > gpurun_out/${TAG}_bench_prof.json 2> gpurun_out/${TAG}_bench_prof.err
python tools/cpuprof_report.py gpurun_out/${TAG}_cpu_prof.txt 70 > gpurun_out/${TAG}_cpu_profile_top.txt 2>&1
python - <<P
import json
j=json.load(open('gpurun_out/${TAG}_bench_prof.json'))
print(j['value']/1e9, j['ms_per_step'], j['host'])
P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${TAG}_prof -o ${TAG} -- python3 /root/repo/bench.py --steps 10 --warmup 2 --cpu-contigs 0 --groups 1 > /root/repo/gpurun_out/${TAG}_bench_g1_rocprof.json 2> /root/repo/gpurun_out/${TAG}_rocprof.err
find /root/repo/gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} /root/repo/gpurun_out/${TAG}_kernel_stats_g1.csv
find /root/repo/gpurun_out/${TAG}_prof -name "*kernel_trace.csv" -delete
head -40 /root/repo/gpurun_out/${TAG}_kernel_stats_g1.csv | cut -c1-160
