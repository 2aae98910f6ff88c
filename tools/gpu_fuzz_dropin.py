"""Damaged SAM / reads / GFA files through the DROP-IN executables on the GPU (the kernels behind the file boundary): every run must
end with an exit status -- no signal, no hang, no device fault. CIGAR-targeted damage on top of random bytes: operations longer than
the read or the contig, zero-length and huge counts.  usage (GPU box): python tools/gpu_fuzz_dropin.py [seed] [runs per case]"""
import os
import random
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu

CV = os.path.join(ROOT, "hairsplitter_amd", "bin", "HS_call_variants")
SR = os.path.join(ROOT, "hairsplitter_amd", "bin", "HS_separate_reads")


def damage_cigar(sam, rnd):
    lines = sam.split("\n")
    recs = [i for i, l in enumerate(lines) if l and not l.startswith("@") and l.count("\t") >= 10]
    for i in rnd.sample(recs, min(len(recs), rnd.randint(1, 4))):
        f = lines[i].split("\t")
        ops = re.findall(r"(\d+)([MIDNSHP=X])", f[5])
        if not ops:
            continue
        k = rnd.randrange(len(ops))
        n, o = ops[k]
        kind = rnd.randint(0, 5)
        if kind == 0: ops[k] = (str(int(n) * 1000 + 7), o)
        elif kind == 1: ops[k] = ("0", o)
        elif kind == 2: ops[k] = ("2147483647", o)
        elif kind == 3: ops[k] = (n, rnd.choice("MIDSH=X"))
        elif kind == 4: ops = ops[:k + 1]
        else: f[3] = str(rnd.choice([0, 1, 10 ** 9, int(f[3]) + 10 ** 6]))
        f[5] = "".join(a + b for a, b in ops)
        lines[i] = "\t".join(f)
    return "\n".join(lines)


def main():
    rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    env = dict(os.environ, HS_NO_DETACH="1")
    rcs, bad = {}, 0
    for case in ("simple_mock", "edge_ops", "clips"):
        for it in range(runs):
            with tempfile.TemporaryDirectory() as td:
                meta = gu.unpack(case, td)
                sam = os.path.join(td, "aln.sam")
                if it % 3 != 2:
                    open(sam, "w").write(damage_cigar(open(sam).read(), rnd))
                else:
                    p = rnd.choice([sam, os.path.join(td, "assembly.gfa"), gu.reads_path(td, meta)])
                    s = bytearray(open(p, "rb").read())
                    for _ in range(rnd.randint(1, 4)):
                        k = rnd.randrange(len(s))
                        if rnd.random() < 0.5: del s[k:k + rnd.randint(1, 50)]
                        else: s[k] = rnd.choice(b"\t\n0123456789MIDS*@\x00")
                    open(p, "wb").write(bytes(s))
                kw = meta.get("kwargs", {})
                col, vcf, err, gro = (os.path.join(td, "f_" + n) for n in ("variants.col", "variants.vcf", "error_rate.txt", "reads_haplo.gro"))
                try:
                    r = subprocess.run([CV, os.path.join(td, "assembly.gfa"), gu.reads_path(td, meta), sam, "2", td, err, str(kw.get("amplicon", 0)), "0", col, vcf, "0.33"],
                                       capture_output=True, env=env, timeout=240)
                    rc = r.returncode
                    if rc == 0 and os.path.exists(col):
                        r2 = subprocess.run([SR, col, "2", meta["error_rate_arg"], os.path.join(td, "absent_ploidy.txt"), "0", "0.01", str(kw.get("amplicon", 0)), gro, "0"],
                                            capture_output=True, env=env, timeout=240)
                        if r2.returncode < 0:
                            rc = r2.returncode
                            r = r2
                except subprocess.TimeoutExpired:
                    rc = "timeout"
                    r = None
                rcs[rc] = rcs.get(rc, 0) + 1
                if rc == "timeout" or (isinstance(rc, int) and rc < 0):
                    bad += 1
                    print("FINDING", case, it, rc, (r.stdout + r.stderr)[-600:].decode(errors="replace") if r else "", flush=True)
    print("exit codes:", rcs, "findings:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
