set -x
cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_full_configs.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r02b_full_tests.log
cat gpurun_out/r02b_full_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02b_prof -o r02b -- python3 /root/repo/bench.py --steps 10 --warmup 2 --cpu-contigs 0 > /root/repo/gpurun_out/r02b_bench_under_rocprof.json 2> /root/repo/gpurun_out/r02b_rocprof.err
ls -R /root/repo/gpurun_out/r02b_prof | head
