cd /tmp && export TMPDIR=/tmp
TAG=${1:-r03g}
mkdir -p /root/repo/gpurun_out
rocprofv3 --hip-trace --stats --output-format csv -d /root/repo/gpurun_out/${TAG}_hip -o ${TAG} -- python3 /root/repo/bench.py --steps 10 --warmup 2 --cpu-contigs 0 > /root/repo/gpurun_out/${TAG}_bench_hiptrace.json 2> /root/repo/gpurun_out/${TAG}_hiptrace.err
find /root/repo/gpurun_out/${TAG}_hip -name "*hip_api_stats.csv" | head -1 | xargs -I{} cp {} /root/repo/gpurun_out/${TAG}_hip_api_stats.csv
find /root/repo/gpurun_out/${TAG}_hip -name "*_trace.csv" -delete
head -30 /root/repo/gpurun_out/${TAG}_hip_api_stats.csv | cut -c1-140
ls /root/repo/gpurun_out/${TAG}_hip
