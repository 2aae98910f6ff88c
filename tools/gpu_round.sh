# end-of-round evidence: default bench (with the file-to-file leg), rocprofv3 kernel stats of the same command, A1 throughput
cd /root/repo
TAG=${1:-r02x}
mkdir -p gpurun_out
timeout 1500 python bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench_c4.json 2> gpurun_out/${TAG}_bench.err
tail -c 300 gpurun_out/${TAG}_bench.err
timeout 600 python tools/myers_bench.py > gpurun_out/${TAG}_myers_bench.json 2>> gpurun_out/${TAG}_bench.err
cat gpurun_out/${TAG}_myers_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${TAG}_prof -o ${TAG} -- python3 /root/repo/bench.py --steps 20 --warmup 3 --cpu-contigs 0 > /root/repo/gpurun_out/${TAG}_bench_c4_under_rocprof.json 2> /root/repo/gpurun_out/${TAG}_rocprof.err
find /root/repo/gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} /root/repo/gpurun_out/${TAG}_kernel_stats_bench_c4.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${TAG}_prof_myers -o ${TAG}m -- python3 /root/repo/tools/myers_bench.py 20000 2000 > /dev/null 2>&1
find /root/repo/gpurun_out/${TAG}_prof_myers -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} /root/repo/gpurun_out/${TAG}_kernel_stats_myers.csv
find /root/repo/gpurun_out/${TAG}_prof /root/repo/gpurun_out/${TAG}_prof_myers -name "*kernel_trace.csv" -delete
head -12 /root/repo/gpurun_out/${TAG}_kernel_stats_bench_c4.csv | cut -c1-150
head -4 /root/repo/gpurun_out/${TAG}_kernel_stats_myers.csv | cut -c1-150
