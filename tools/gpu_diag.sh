cd /root/repo
bash tools/gpu_prof.sh r02f --config C2 --groups 1 > /dev/null 2>&1
python3 - <<'P'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/r02f_kernel_stats.csv')))
for r in rows[:6]:
    print("C2 %-34s calls %5s  ms/step %7.3f" % (r['Name'].split('(')[0][-34:], r['Calls'], float(r['TotalDurationNs'])/1e6/20))
P
cd /tmp && export TMPDIR=/tmp
export HS_BENCH_SERIAL_SETUP=1
cd "$GRAFT_REPO_ROOT"
i=0
for g in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  out=gpurun_out/pmc_diag/g$i
  mkdir -p $out
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d $out -- python3 bench.py --steps 1 --warmup 0 --cpu-contigs 0 --groups 1 > $out/bench.json 2> $out/err.log
  i=$((i+1))
done
python3 tools/pmc_summary.py gpurun_out/pmc_diag > gpurun_out/pmc_diag/summary.csv
grep -E "k_cw_seeded_lanes|k_cw_seed_sets|k_window_tail|k_column_partition_lanes|k_simdiff" gpurun_out/pmc_diag/summary.csv
find gpurun_out/pmc_diag -name "*.csv" -size +5M -delete
