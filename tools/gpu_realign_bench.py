"""Throughput of the PAF ingest (hs_realign_paf): the reads of a few synthetic contigs, 30x ONT, as PAF lines -> SAM through the device aligner.
usage (GPU box): python tools/gpu_realign_bench.py [contig_kb] [n_contigs]"""
import ctypes as C, os, re, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hairsplitter_amd import api, synth

OPS = re.compile(r"(\d+)([MIDNSHP=X])")


def sam_to_paf(sam, paf):
    n = 0
    with open(sam) as f, open(paf, "w") as o:
        for l in f:
            if l.startswith("@"):
                continue
            t = l.rstrip("\n").split("\t")
            qlen = int([x for x in t if x.startswith("LN:i:")][0][5:])
            ops = [(int(a), b) for a, b in OPS.findall(t[5])]
            ls = ops[0][0] if ops[0][1] in "SH" else 0
            rs = ops[-1][0] if ops[-1][1] in "SH" else 0
            refspan = sum(a for a, b in ops if b in "MD=X")
            minus = int(t[1]) & 16
            qs, qe = (rs, qlen - ls) if minus else (ls, qlen - rs)
            ts = int(t[3]) - 1
            o.write("\t".join(map(str, [t[0], qlen, qs, qe, "-" if minus else "+", t[2], 0, ts, ts + refspan, 0, refspan, 60])) + "\n")
            n += 1
    return n


class Stats(C.Structure):
    _fields_ = [("n_lines", C.c_int64), ("n_aligned", C.c_int64), ("query_bases", C.c_int64), ("ms_device", C.c_double), ("ms_total", C.c_double)]


def main():
    kb = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    nc = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    lib = api.load()
    with tempfile.TemporaryDirectory() as td:
        cs = [synth.make_contig(51, i, kb * 1000, 2, 0.01, 30, "ont") for i in range(nc)]
        f = synth.write_files(cs, td)
        paf, sam = os.path.join(td, "a.paf"), os.path.join(td, "re.sam")
        n = sam_to_paf(f["sam"], paf)
        for rep in range(2):
            st = Stats()
            t0 = time.time()
            rc = lib.hs_realign_paf(f["gfa"].encode(), f["reads"].encode(), paf.encode(), sam.encode(), C.c_int32(16), C.byref(st))
            dt = time.time() - t0
            assert rc == 0, lib.hs_last_error()
            print("run %d: %d records, %.1f M query bases, mean read %.0f b: device %.1f ms, total %.1f ms (wall %.2f s) -> %.1f k reads/s, %.2f G query bases/s on the device; %.1f k reads/s end to end"
                  % (rep, st.n_aligned, st.query_bases / 1e6, st.query_bases / max(1, st.n_aligned), st.ms_device, st.ms_total, dt, st.n_aligned / st.ms_device, st.query_bases / st.ms_device / 1e6, st.n_aligned / st.ms_total), flush=True)


main()
