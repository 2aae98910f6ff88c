#!/bin/bash
# HBM traffic per launch of every kernel of the DEFAULT bench command (C4, 500 contigs, default groups), as the guide's
# HBM / rocprofv3 section prescribes: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (kernel trace only), bytes =
# counter x 1024; writes profiles/traffic_latest.json (read by bench.py for roofline.traffic) and the per-kernel CSV.
# One contig group by default (HS_PMC_GROUPS): every kernel one launch per step over the whole job, as in bench.py's one-group leg whose
# launch time `roofline.achieved` is quoted on.
# usage (on the GPU box, through gpurun): bash tools/pmc_traffic.sh <tag>
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
export HS_BENCH_SERIAL_SETUP=1
cd "$GRAFT_REPO_ROOT"
i=0
for g in "FETCH_SIZE" "WRITE_SIZE"; do
  out=gpurun_out/pmc_${tag}/g$i
  mkdir -p $out
  timeout 900 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $out -- python3 bench.py --steps 2 --warmup 0 --cpu-contigs 0 --groups ${HS_PMC_GROUPS:-1} > $out/bench.json 2> $out/err.log
  i=$((i+1))
done
python3 tools/pmc_summary.py gpurun_out/pmc_${tag} > gpurun_out/pmc_${tag}/summary.csv
python3 tools/pmc_traffic_json.py gpurun_out/pmc_${tag} > gpurun_out/pmc_${tag}/traffic.json
grep -E "k_pileup_runs|k_pileup_packed|k_column_stats_tiled|k_simdiff|k_cw_seeded_lanes|k_cw_seed_sets|k_window_tail|k_column_partition_lanes" gpurun_out/pmc_${tag}/traffic.json
find gpurun_out/pmc_${tag} -name "*.csv" -size +5M -delete
