cd /root/repo
mkdir -p gpurun_out
TAG=${1:-prof}
shift
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "partition" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${TAG} -o ${TAG} -- python3 /root/repo/bench.py --steps 10 --warmup 2 --cpu-contigs 0 "$@" > /root/repo/gpurun_out/${TAG}_bench_under_rocprof.json 2> /root/repo/gpurun_out/${TAG}_rocprof.err
find /root/repo/gpurun_out/${TAG} -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} /root/repo/gpurun_out/${TAG}_kernel_stats.csv
head -30 /root/repo/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
find /root/repo/gpurun_out/${TAG} -name "*kernel_trace.csv" -size +30M -delete
