"""Resolves the samples of HS_CPU_PROFILE (hs_cpuprof.cpp) to function names with addr2line and prints (1) the top of the
profile by the function that was running and (2) by the function of libhairsplitter_hip.so that was (most likely) on the
stack when libc / the HIP runtime were running."""
import collections
import subprocess
import sys

rows = []
self_lib = None
for l in open(sys.argv[1]):
    if l.startswith("# self"):
        self_lib = l.split()[2]; continue
    if l.startswith("#"):
        print(l.strip()); continue
    m, off, via, c = l.rsplit(" ", 3)
    rows.append((m, off, via, int(c)))
total = sum(r[3] for r in rows)


def resolve(mod, offs, with_line=False):
    import os
    if mod and not os.path.exists(mod):      # a profile taken on another box: the same library of this tree
        here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hairsplitter_amd", "lib", os.path.basename(mod))
        if os.path.exists(here):
            mod = here
    if mod == "?" or not offs:
        return {o: "?" for o in offs}
    try:
        out = subprocess.run(["addr2line", "-f", "-C", "-e", mod] + ["0x" + o for o in offs], capture_output=True, text=True).stdout.splitlines()
        return {o: n.split("(")[0][:60] + (" @" + w.split("/")[-1].split(" ")[0] if with_line and not w.startswith("??") else "") for o, n, w in zip(offs, out[0::2], out[1::2])}
    except Exception:
        return {o: "?" for o in offs}


by_mod = collections.defaultdict(set)
for m, off, via, c in rows:
    by_mod[m].add(off)
names = {m: resolve(m, sorted(offs)) for m, offs in by_mod.items()}
via_names = resolve(self_lib, sorted({r[2] for r in rows if r[2] != "0"}), with_line=True) if self_lib else {}
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
fn = collections.Counter()
owner = collections.Counter()
owner_leaf = collections.defaultdict(collections.Counter)
for m, off, via, c in rows:
    short = m.split("/")[-1]
    leaf = (short, names[m].get(off, "?"))
    fn[leaf] += c
    o = via_names.get(via, "(not under this library)") if via != "0" else "(not under this library)"
    owner[o] += c
    owner_leaf[o][short if short != (self_lib or "").split("/")[-1] else "self"] += c
print("== by the function running ==")
for k, c in fn.most_common(top):
    print("%6.2f%%  %-28s %s" % (100.0 * c / max(total, 1), k[0], k[1]))
print("== by the function of this library on the stack ==")
for k, c in owner.most_common(top):
    parts = ", ".join("%s %.1f%%" % (kk, 100.0 * cc / max(total, 1)) for kk, cc in owner_leaf[k].most_common(4))
    print("%6.2f%%  %-90s [%s]" % (100.0 * c / max(total, 1), k, parts))
