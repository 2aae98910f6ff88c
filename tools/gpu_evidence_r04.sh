# The round's evidence in one call (run through gpurun; HS_COMMIT = the commit it is taken at): default bench line with the file-to-file leg,
# rocprofv3 kernel stats of the same command and at one contig group, HIP API stats (copy / fill calls), CPU profile, rank-of-8 runs,
# PMC traffic at one group, the summary table. Everything lands in gpurun_out/<tag>_*; what is judged is copied to profiles/.
TAG=${1:-r04}
R=/root/repo
cd $R
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/${TAG}_bench_c4.json 2> gpurun_out/${TAG}_bench_c4.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o ${TAG} -- python3 $R/bench.py --cpu-contigs 0 > $R/gpurun_out/${TAG}_bench_c4_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.err
find $R/gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/${TAG}_kernel_stats_bench_c4.csv
find $R/gpurun_out/${TAG}_prof -name "*_trace.csv" -delete
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof1 -o ${TAG} -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-contigs 0 --groups 1 > $R/gpurun_out/${TAG}_bench_c4_groups1_under_rocprof.json 2>> $R/gpurun_out/${TAG}_rocprof.err
find $R/gpurun_out/${TAG}_prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/${TAG}_kernel_stats_c4_groups1.csv
find $R/gpurun_out/${TAG}_prof1 -name "*_trace.csv" -delete
timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_hip -o ${TAG} -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-contigs 0 > $R/gpurun_out/${TAG}_bench_hiptrace.json 2> $R/gpurun_out/${TAG}_hiptrace.err
find $R/gpurun_out/${TAG}_hip -name "*hip_api_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/${TAG}_hip_api_stats.csv
find $R/gpurun_out/${TAG}_hip -name "*_trace.csv" -delete
cd $R
HS_CPU_PROFILE=$R/gpurun_out/${TAG}_cpu_prof.txt timeout 900 python bench.py --steps 150 --warmup 3 --cpu-contigs 0 > gpurun_out/${TAG}_bench_prof.json 2> gpurun_out/${TAG}_bench_prof.err
python tools/cpuprof_report.py gpurun_out/${TAG}_cpu_prof.txt 60 > gpurun_out/${TAG}_cpu_profile_top.txt 2>&1
rm -f gpurun_out/${TAG}_cpu_prof.txt
timeout 600 python bench.py --as-rank-of 8 --cpu-contigs 0 > gpurun_out/${TAG}_rank_of_8.json 2> gpurun_out/${TAG}_rank_of_8.err
timeout 600 python bench.py --as-rank-of 8 --cores 2 --cpu-contigs 0 > gpurun_out/${TAG}_rank_of_8_2cores.json 2>> gpurun_out/${TAG}_rank_of_8.err
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$R} bash tools/pmc_traffic.sh ${TAG} > gpurun_out/${TAG}_pmc_traffic.log 2>&1
cp gpurun_out/pmc_${TAG}/traffic.json gpurun_out/${TAG}_traffic_groups1.json 2>/dev/null
python tools/make_summary.py ${TAG} gpurun_out/${TAG}_kernel_stats_c4_groups1.csv gpurun_out/${TAG}_bench_c4_groups1_under_rocprof.json gpurun_out/${TAG}_bench_c4.json gpurun_out/${TAG}_traffic_groups1.json > gpurun_out/SUMMARY_${TAG}.md 2> gpurun_out/${TAG}_summary.err
python - <<P
import json
for n in ("bench_c4", "bench_c4_under_rocprof", "rank_of_8", "rank_of_8_2cores", "bench_prof"):
    try:
        j = json.load(open("gpurun_out/${TAG}_%s.json" % n))
        print(n, round(j["value"] / 1e9, 2), "Gbp/s", round(j["ms_per_step"], 2), "ms", round(j["host"]["process_cpu_ms_per_step"], 1), "CPU-ms", j["host"]["waits_per_step"], "waits", j["roofline"]["kernel"], round(j["roofline"]["frac"], 4))
    except Exception as e:
        print(n, "failed", e)
P
head -40 gpurun_out/SUMMARY_${TAG}.md
grep -iE "memcpy|memset" gpurun_out/${TAG}_hip_api_stats.csv | cut -c1-160
