set -x
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r02a_tests.log
cat gpurun_out/r02a_tests.log
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-contigs 0 > gpurun_out/r02a_bench_c2x256.json 2> gpurun_out/r02a_bench.err
tail -c 1500 gpurun_out/r02a_bench_c2x256.json; tail -5 gpurun_out/r02a_bench.err
