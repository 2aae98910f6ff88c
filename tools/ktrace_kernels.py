#!/usr/bin/env python3
"""Per kernel, the launches of ONE default step (eight contig groups) from gpurun_out/ktrace_rows.tsv.gz (tools/gpu_ktrace.sh): count, sum,
min / median / max duration -- what a kernel costs a contig group's chain, as opposed to its own time over the whole job (SUMMARY_*.md)."""
import gzip, collections, sys
rows = [l.rstrip('\n').split('\t') for l in gzip.open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/ktrace_rows.tsv.gz', 'rt')]
ev = sorted((int(r[0]), int(r[1]), r[4]) for r in rows)
starts = [e[0] for e in ev if e[2].startswith('k_cigar_scan')]
steps = []
for s in starts:
    if not steps or s - steps[-1][-1] > 3_000_000: steps.append([s])
    else: steps[-1].append(s)
t0, t1 = steps[-2][0], steps[-1][0]
seg = [e for e in ev if t0 <= e[0] < t1]
agg = collections.defaultdict(list)
for s, e, n in seg: agg[n].append((e - s) / 1e3)
print("step span %.2f ms, %d dispatches" % ((max(e[1] for e in seg) - t0) / 1e6, len(seg)))
for n, l in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-34s x%3d sum %7.1f us  min %6.1f med %6.1f max %6.1f" % (n[:34], len(l), sum(l), min(l), sorted(l)[len(l) // 2], max(l)))
