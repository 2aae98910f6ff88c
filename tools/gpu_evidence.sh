# The round's evidence in one call (run through gpurun): default bench line with the file-to-file leg and the parity gate, rocprofv3 kernel
# stats of the same command and at one contig group, CPU profile, rank-of-8 runs, the file-to-file job on eight shards of one device,
# PMC traffic at one group, the summary table. Everything lands in gpurun_out/<tag>_*; what is judged is copied to profiles/.
TAG=${1:-r06}
R=/root/repo
cd $R
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/${TAG}_bench_c4.json 2> gpurun_out/${TAG}_bench_c4.err
echo "bench rc $?" >> gpurun_out/${TAG}_bench_c4.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o ${TAG} -- python3 $R/bench.py --cpu-contigs 0 > $R/gpurun_out/${TAG}_bench_c4_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.err
find $R/gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/${TAG}_kernel_stats_bench_c4.csv
find $R/gpurun_out/${TAG}_prof -name "*_trace.csv" -delete
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof1 -o ${TAG} -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-contigs 0 --groups 1 > $R/gpurun_out/${TAG}_bench_c4_groups1_under_rocprof.json 2>> $R/gpurun_out/${TAG}_rocprof.err
find $R/gpurun_out/${TAG}_prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/${TAG}_kernel_stats_c4_groups1.csv
find $R/gpurun_out/${TAG}_prof1 -name "*_trace.csv" -delete
cd $R
HS_CPU_PROFILE=$R/gpurun_out/${TAG}_cpu_prof.txt timeout 600 python bench.py --steps 150 --warmup 3 --cpu-contigs 0 > gpurun_out/${TAG}_bench_prof.json 2> gpurun_out/${TAG}_bench_prof.err
python tools/cpuprof_report.py gpurun_out/${TAG}_cpu_prof.txt 60 > gpurun_out/${TAG}_cpu_profile_top.txt 2>&1
rm -f gpurun_out/${TAG}_cpu_prof.txt
# one rank of an N-rank job on this GPU: with the cores a rank has on an 8-GPU host (--threads 16) and with this box's 16-core quota shared by the N ranks
for N in 2 4 8; do
  timeout 400 python bench.py --as-rank-of $N --threads 16 --cpu-contigs 0 --steps 40 > gpurun_out/${TAG}_rank_of_${N}_16threads.json 2>> gpurun_out/${TAG}_rank_of.err
  timeout 400 python bench.py --as-rank-of $N --cpu-contigs 0 --steps 40 > gpurun_out/${TAG}_rank_of_${N}_shared_quota.json 2>> gpurun_out/${TAG}_rank_of.err
done
timeout 400 python bench.py --as-rank-of 8 --cores 2 --cpu-contigs 0 > gpurun_out/${TAG}_rank_of_8_2cores.json 2>> gpurun_out/${TAG}_rank_of.err
# loop A of every contig on the device (opt-in): the line, for the record
HS_LOOP_A_ON_DEVICE=1 timeout 400 python bench.py --cpu-contigs 0 --no-f2f-job > gpurun_out/${TAG}_bench_c4_loop_a_on_device.json 2> gpurun_out/${TAG}_loop_a_on_device.err
# the file-to-file job on one device and on eight shards of it (HS_DEVICES=0 x 8: the same sharding, batches and merge as eight GPUs)
python - > gpurun_out/${TAG}_f2f_devices.txt 2>&1 <<P
import sys, subprocess, os, time
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import full_configs as fc
os.makedirs("/tmp/f2f", exist_ok=True)
f,bp,n,t=fc.generate_files("C4","/tmp/f2f",None,workers=8)
for devs in (None, "0,0,0,0,0,0,0,0"):
    e=dict(os.environ, HS_NO_DETACH="1")
    if devs: e["HS_DEVICES"]=devs
    for rep in range(3):
        t0=time.time(); subprocess.run(["hairsplitter_amd/bin/HS_call_variants",f["gfa"],f["reads"],f["sam"],"16","/tmp/f2f","/tmp/f2f/err.txt","0","0","/tmp/f2f/o.col","/tmp/f2f/o.vcf","0.33"], env=e, stdout=subprocess.DEVNULL, check=True); t1=time.time()
        er=min(float("%g" % float(open("/tmp/f2f/err.txt").read().strip())),0.15)
        subprocess.run(["hairsplitter_amd/bin/HS_separate_reads","/tmp/f2f/o.col","16",str(er),"/tmp/f2f/no_ploidy","0","0.01","0","/tmp/f2f/o.gro","0"], env=e, stdout=subprocess.DEVNULL, check=True); t2=time.time()
        print("HS_DEVICES=%s run %d: call_variants %.2f s separate_reads %.2f s total %.2f s (no detach, to full exit)" % (devs or "one device", rep, t1-t0, t2-t1, t2-t0), flush=True)
P
bash tools/fetch_calib.sh ${TAG} > /dev/null 2>&1
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$R} timeout 900 bash tools/pmc_traffic.sh ${TAG} > gpurun_out/${TAG}_pmc_traffic.log 2>&1
cp gpurun_out/pmc_${TAG}/traffic.json gpurun_out/${TAG}_traffic_groups1.json 2>/dev/null
python tools/make_summary.py ${TAG} gpurun_out/${TAG}_kernel_stats_c4_groups1.csv gpurun_out/${TAG}_bench_c4_groups1_under_rocprof.json gpurun_out/${TAG}_bench_c4.json gpurun_out/${TAG}_traffic_groups1.json > gpurun_out/SUMMARY_${TAG}.md 2> gpurun_out/${TAG}_summary.err
python - <<P
import json
for n in ("bench_c4", "bench_c4_under_rocprof", "bench_prof", "rank_of_2_16threads", "rank_of_2_shared_quota", "rank_of_4_16threads", "rank_of_4_shared_quota", "rank_of_8_16threads", "rank_of_8_shared_quota", "rank_of_8_2cores", "bench_c4_loop_a_on_device"):
    try:
        j = json.loads(open("gpurun_out/${TAG}_%s.json" % n).read().strip().splitlines()[-1])
        print(n, round(j["value"] / 1e9, 2), "Gbp/s", round(j["ms_per_step"], 2), "ms", round(j["host"]["process_cpu_ms_per_step"], 1), "CPU-ms", j["host"]["waits_per_step"], "waits", j["roofline"]["kernel"], round(j["roofline"]["frac"], 4), j["roofline"].get("frac_probe_one_group"), (j.get("parity") or {}).get("identical"))
    except Exception as e:
        print(n, "failed", e)
P
cat gpurun_out/${TAG}_f2f_devices.txt
head -30 gpurun_out/SUMMARY_${TAG}.md
