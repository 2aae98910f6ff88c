#!/bin/bash
# PMC passes for K1 only (run ON the GPU box): SQ instruction / wait counters + vector-memory and L1 counters, one group per pass.
# usage: tools/gpu_k1_pmc.sh <tag> [env assignments for the bench, e.g. HS_K1_PACKED=1]
tag=${1:-k1}; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
export HS_BENCH_SERIAL_SETUP=1
cd "$GRAFT_REPO_ROOT"
ngroups=${HS_PMC_GROUPS:-4}
groups=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS")      # (a fifth group with TCP_PENDING_STALL / TA_*_STALLED counters made rocprofv3 abort and hang on this pool: not collected)
i=0
for g in "${groups[@]}"; do
  [ $i -ge $ngroups ] && break
  out=gpurun_out/pmc_${tag}/g$i
  mkdir -p $out
  timeout 240 rocprofv3 --kernel-trace --pmc $g --kernel-include-regex "k_pileup|k_cigar|k_column_stats" --output-format csv -d $out -- python3 bench.py --steps 2 --warmup 1 --cpu-contigs 0 --groups 1 > $out/bench.json 2> $out/err.log
  tail -2 $out/err.log
  i=$((i+1))
done
python3 tools/pmc_summary.py gpurun_out/pmc_${tag} > gpurun_out/pmc_${tag}/summary.csv
find gpurun_out/pmc_${tag} -name "*.csv" ! -name summary.csv -delete
grep -E "pileup|cigar|column_stats" gpurun_out/pmc_${tag}/summary.csv
