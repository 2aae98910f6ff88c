"""Resolves the samples of HS_CPU_PROFILE (hs_cpuprof.cpp) to function names with addr2line and prints the top of the profile."""
import collections
import subprocess
import sys

mods = collections.defaultdict(list)
total = 0
for l in open(sys.argv[1]):
    if l.startswith("#"):
        print(l.strip()); continue
    m, off, c = l.rsplit(" ", 2)
    mods[m].append((off, int(c))); total += int(c)
fn = collections.Counter()
for m, lst in mods.items():
    if m == "?":
        fn["?"] += sum(c for _, c in lst); continue
    try:
        out = subprocess.run(["addr2line", "-f", "-C", "-e", m] + ["0x" + o for o, _ in lst], capture_output=True, text=True).stdout.splitlines()
        names = out[0::2]
    except Exception:
        names = ["?"] * len(lst)
    short = m.split("/")[-1]
    for (o, c), n in zip(lst, names):
        fn[(short, n.split("(")[0][:70])] += c
for (k, c) in fn.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 45):
    print("%6.2f%%  %-28s %s" % (100.0 * c / max(total, 1), k[0] if isinstance(k, tuple) else k, k[1] if isinstance(k, tuple) else ""))
