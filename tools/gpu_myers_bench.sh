#!/bin/bash
# A1 throughput: the stage-5 shape and reads against a contig window; kernel trace of both.  usage: gpu_myers_bench.sh <tag>
tag=${1:-x}
mkdir -p gpurun_out
timeout 600 python tools/myers_bench.py 20000 2000 > gpurun_out/${tag}_myers_bench.json 2> gpurun_out/${tag}_myers_bench.err
timeout 900 python tools/myers_bench.py 4096 12000 10000 > gpurun_out/${tag}_myers_bench_reads.json 2> gpurun_out/${tag}_myers_bench_reads.err
cat gpurun_out/${tag}_myers_bench.json gpurun_out/${tag}_myers_bench_reads.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_my -o my -- python3 $GRAFT_REPO_ROOT/tools/myers_bench.py 4096 12000 10000 > /dev/null 2>&1
f=$(find /tmp/prof_my -name '*kernel_stats.csv' | head -1)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_my5 -o my -- python3 $GRAFT_REPO_ROOT/tools/myers_bench.py 20000 2000 > /dev/null 2>&1
g=$(find /tmp/prof_my5 -name '*kernel_stats.csv' | head -1)
[ -n "$g" ] && cp "$g" $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats_myers_stage5.csv && head -3 "$g"
if [ -n "$f" ]; then cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats_myers_reads.csv; head -4 "$f"; else ls -R /tmp/prof_my | head; fi
