# kernel trace of a few steps of the default job (diagnostic): per-dispatch begin / end, for the GPU's busy / idle picture
R=/root/repo
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
HS_ORDER_SCOPE=${HS_ORDER_SCOPE:-k2} HS_NO_KERNEL_STATS=1 HS_BENCH_NO_PROBE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ktrace -o kt -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-contigs 0 > $R/gpurun_out/ktrace.json 2> $R/gpurun_out/ktrace.err
f=$(find $R/gpurun_out/ktrace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), 'dispatches; columns', list(rows[0].keys()))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ','').replace('hsdev::',''), r.get('Queue_Id','')) for r in rows]
ev.sort()
# steps = runs separated by k_cigar_scan
starts = [e[0] for e in ev if e[2].startswith('k_cigar_scan')]
print('steps found', len(starts))
for si in range(max(0, len(starts) - 3), len(starts) - 0):
    t0 = starts[si]; t1 = starts[si + 1] if si + 1 < len(starts) else ev[-1][1]
    seg = [e for e in ev if t0 <= e[0] < t1]
    end = max(e[1] for e in seg)
    # union busy
    busy = 0; cur_s = None; cur_e = None
    for s, e, n, q in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print('step %d: span %.2f ms, GPU busy (any kernel) %.2f ms, sum of kernel durations %.2f ms, %d dispatches' % (si, (end - t0) / 1e6, busy / 1e6, sum(e[1] - e[0] for e in seg) / 1e6, len(seg)))
    # busy per ms bucket
    buckets = collections.defaultdict(float)
    cur_s = None; cur_e = None
    iv = []
    for s, e, n, q in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None: iv.append((cur_s, cur_e))
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    iv.append((cur_s, cur_e))
    for s, e in iv:
        b = int((s - t0) / 1e6)
        while s < e:
            lim = t0 + (b + 1) * 1000000
            x = min(e, lim)
            buckets[b] += (x - s) / 1e6
            s = x; b += 1
    print('  busy fraction per ms:', ' '.join('%.2f' % buckets[b] for b in range(int((end - t0) / 1e6) + 1)))
    # concurrency-weighted: per ms the sum of durations
    dur = collections.defaultdict(float)
    for s, e, n, q in seg:
        b = int((s - t0) / 1e6)
        while s < e:
            lim = t0 + (b + 1) * 1000000
            x = min(e, lim)
            dur[b] += (x - s) / 1e6
            s = x; b += 1
    print('  kernel-ms per ms     :', ' '.join('%.1f' % dur[b] for b in range(int((end - t0) / 1e6) + 1)))
    if si == len(starts) - 2:
        agg = collections.defaultdict(lambda: [0, 0.0])
        for s, e, n, q in seg: agg[n][0] += 1; agg[n][1] += (e - s) / 1e6
        for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]: print('   %-40s x%3d %.3f ms' % (n[:40], c, d))
P

python3 - "$f" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ','').replace('hsdev::',''), r.get('Stream_Id','') or r.get('Queue_Id','')) for r in rows]
ev.sort()
starts = [e[0] for e in ev if e[2].startswith('k_cigar_scan')]
si = len(starts) - 2
t0 = starts[si]; t1 = starts[si + 1]
seg = [e for e in ev if t0 <= e[0] < t1]
by = collections.defaultdict(list)
for e in seg: by[e[3]].append(e)
print('--- per stream, step', si)
for q, l in sorted(by.items(), key=lambda kv: kv[1][0][0]):
    l.sort()
    print('stream %s: %d kernels, first %.2f last end %.2f, sum of durations %.2f ms' % (q, len(l), (l[0][0]-t0)/1e6, (max(x[1] for x in l)-t0)/1e6, sum(x[1]-x[0] for x in l)/1e6))
# the stream that ends last: its chain in full
last = max(by.items(), key=lambda kv: max(x[1] for x in kv[1]))[0]
l = by[last]
print('--- chain of stream', last)
prev_end = None
for s_, e_, n, q in l:
    gap = (s_ - prev_end) / 1e6 if prev_end is not None else 0.0
    print('  %-34s start %7.3f dur %6.3f gap-before %6.3f' % (n[:34], (s_ - t0) / 1e6, (e_ - s_) / 1e6, gap))
    prev_end = max(prev_end, e_) if prev_end is not None else e_
P
python3 - "$f" > $R/gpurun_out/ktrace_rows.tsv <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print('\t'.join([r['Start_Timestamp'], r['End_Timestamp'], r['Thread_Id'], r['Queue_Id'], r['Kernel_Name'].split('(')[0].replace('void ','').replace('hsdev::','')[:40], r['Grid_Size_X'], r['Workgroup_Size_X']]))
P
gzip -f $R/gpurun_out/ktrace_rows.tsv
rm -rf $R/gpurun_out/ktrace
python3 $R/tools/ktrace_timeline.py 7 > $R/gpurun_out/ktrace_timeline.txt 2>&1
head -14 $R/gpurun_out/ktrace_timeline.txt
