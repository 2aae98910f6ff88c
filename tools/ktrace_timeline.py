"""Per contig group (= launching host thread) the start of its main kernels within one step of the default job, the end of its last kernel, and the
number of kernels running per half millisecond -- from the compact trace tools/gpu_ktrace.sh leaves (gpurun_out/ktrace_rows.tsv.gz).
usage: python tools/ktrace_timeline.py [index of the group whose whole chain is printed]"""
import gzip, collections, sys
rows=[l.rstrip('\n').split('\t') for l in gzip.open(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'gpurun_out', 'ktrace_rows.tsv.gz'),'rt')]
ev=sorted((int(r[0]),int(r[1]),r[2],r[4]) for r in rows)
cs=[e for e in ev if e[3].startswith('k_cigar_scan')]
# steps: chunks of 8 cigar scans (fused: one per group); take from the end
n=len(cs)//8
step_starts=[cs[len(cs)-8*(n-i)][0] for i in range(n)]
si=n-2
t0=step_starts[si]; t1=step_starts[si+1]
seg=[e for e in ev if t0<=e[0]<t1]
print('step span %.2f ms'%((max(e[1] for e in seg)-t0)/1e6), len(seg),'dispatches')
by=collections.defaultdict(list)
for e in seg: by[e[2]].append(e)
order=sorted(by, key=lambda k: by[k][0][0])
names=['k_cigar_scan','k_pileup_runs','k_column_stats_tiled_dw','k_gather_tiles','k_cand_bits','k_column_partition_lanes','k_snp_planes','k_simdiff','k_read_graph_rows','k_cw_seeded_lanes','k_window_tail']
print('thread      '+' '.join('%9s'%n[2:11] for n in names)+'   lastend  sumdur')
for t in order:
    l=by[t]
    def st(nm):
        x=[e for e in l if e[3].startswith(nm)]
        return '%9.2f'%((x[0][0]-t0)/1e6) if x else '        -'
    print('%-11s '%t[-6:]+' '.join(st(n) for n in names)+'  %7.2f %7.2f'%((max(e[1] for e in l)-t0)/1e6, sum(e[1]-e[0] for e in l)/1e6))
# busy per 0.5ms
end=max(e[1] for e in seg)
nb=int((end-t0)/5e5)+1
dur=[0.0]*nb
for s,e,_,_ in seg:
    b=int((s-t0)/5e5)
    while s<e:
        lim=t0+(b+1)*500000; x=min(e,lim); dur[b]+=(x-s)/5e5; s=x; b+=1
print('concurrency per 0.5 ms:',' '.join('%.1f'%d for d in dur))
if len(sys.argv)>1:
    t=order[int(sys.argv[1])]
    prev=None
    for s,e,_,n in by[t]:
        print('  %-32s start %7.3f dur %6.3f gap %6.3f'%(n,(s-t0)/1e6,(e-s)/1e6,0 if prev is None else (s-prev)/1e6)); prev=e
