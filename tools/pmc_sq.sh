#!/bin/bash
# PMC passes over the bench command (run ON the GPU box through gpurun): one rocprofv3 --pmc pass per counter group,
# kernel trace only (no sys/hip/hsa tracing with --pmc on this pool). Output: gpurun_out/pmc_<tag>/<group>/...
# usage: tools/pmc.sh <tag> [bench args]
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
export HS_BENCH_SERIAL_SETUP=1   # no forked set-up workers under the profiler
cd "$GRAFT_REPO_ROOT"
groups=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES")
i=0
for g in "${groups[@]}"; do
  out=gpurun_out/pmc_${tag}/g$i
  mkdir -p $out
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d $out -- python3 bench.py --steps 2 --warmup 1 --cpu-contigs 0 --groups 1 "$@" > $out/bench.json 2> $out/err.log
  i=$((i+1))
done
python3 tools/pmc_summary.py gpurun_out/pmc_${tag} > gpurun_out/pmc_${tag}/summary.csv
cat gpurun_out/pmc_${tag}/summary.csv | head -80
