#!/bin/bash
# PMC passes over the bench command (run ON the GPU box through gpurun): one rocprofv3 --pmc pass per counter group,
# kernel trace only (no sys/hip/hsa tracing with --pmc on this pool). Output: gpurun_out/pmc_<tag>/<group>/...
# usage: tools/pmc.sh <tag> [bench args]      (PMC_ONLY="2 3": only those counter groups; PMC_KERNEL=<name>: only that kernel in the print-out)
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
export HS_BENCH_SERIAL_SETUP=1   # no forked set-up workers under the profiler
cd "$GRAFT_REPO_ROOT"
groups=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE")      # (TCP_PENDING_STALL / TA_*_STALLED counters made rocprofv3 abort and hang on this pool: not collected)
i=0
for g in "${groups[@]}"; do
  if [ -n "$PMC_ONLY" ] && ! echo " $PMC_ONLY " | grep -q " $i "; then i=$((i+1)); continue; fi
  out=gpurun_out/pmc_${tag}/g$i
  mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --pmc $g ${PMC_KERNEL:+--kernel-include-regex $PMC_KERNEL} --output-format csv -d $out -- python3 bench.py --steps 2 --warmup 1 --cpu-contigs 0 --groups 1 "$@" > $out/bench.json 2> $out/err.log
  i=$((i+1))
done
python3 tools/pmc_summary.py gpurun_out/pmc_${tag} > gpurun_out/pmc_${tag}/summary.csv
if [ -n "$PMC_KERNEL" ]; then grep "$PMC_KERNEL" gpurun_out/pmc_${tag}/summary.csv; else head -80 gpurun_out/pmc_${tag}/summary.csv; fi
