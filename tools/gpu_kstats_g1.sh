# rocprofv3 kernel statistics at one contig group (every kernel one launch per step)
R=/root/repo
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstats_g1 -o k -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-contigs 0 --groups 1 > $R/gpurun_out/kstats_g1.json 2> $R/gpurun_out/kstats_g1.err
find $R/gpurun_out/kstats_g1 -name "*_trace.csv" -delete
f=$(find $R/gpurun_out/kstats_g1 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-44s calls %4s avg %.3f ms" % (r['Name'].split('(')[0].replace('void ','').replace('hsdev::','')[:44], r['Calls'], float(r['AverageNs'])/1e6))
P
