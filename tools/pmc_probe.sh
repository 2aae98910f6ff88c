# one bounded --pmc pass of the default bench with progress on stderr (diagnostic)
cd /tmp && export TMPDIR=/tmp
export HS_BENCH_SERIAL_SETUP=1
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_probe
HS_TIMING=1 timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_probe -- python3 bench.py --steps 1 --warmup 0 --cpu-contigs 0 --no-f2f-job > gpurun_out/pmc_probe/bench.json 2> gpurun_out/pmc_probe/err.log
echo "rc $?"
grep -c "hs timing" gpurun_out/pmc_probe/err.log
grep "hs timing" gpurun_out/pmc_probe/err.log | tail -5 | cut -c1-200
find gpurun_out/pmc_probe -name "*.csv" -size +5M -delete
