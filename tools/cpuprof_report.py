"""Resolves the samples of HS_CPU_PROFILE (hs_cpuprof.cpp) with llvm-symbolizer (the library carries host line tables) and prints
(1) the top of the profile by the innermost (possibly inlined) function that was running, (2) by source line, and (3) by the
function of libhairsplitter_hip.so that was on the stack when libc / the HIP runtime were running.
usage: cpuprof_report.py <profile> [top]"""
import collections
import os
import subprocess
import sys

SYMBOLIZER = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"

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


def local_path(mod):
    if mod and not os.path.exists(mod):      # a profile taken on another box: the same library of this tree
        here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hairsplitter_amd", "lib", os.path.basename(mod))
        if os.path.exists(here):
            return here
    return mod


def resolve(mod, offs):
    """offset -> (innermost function, file:line, outermost function)"""
    mod = local_path(mod)
    if mod == "?" or not offs or not os.path.exists(mod):
        return {o: ("?", "?", "?") for o in offs}
    inp = "\n".join("0x" + o for o in offs) + "\n"
    try:
        out = subprocess.run([SYMBOLIZER, "--obj=" + mod, "--inlines", "--demangle", "--output-style=LLVM"], input=inp, capture_output=True, text=True).stdout
    except Exception:
        return {o: ("?", "?", "?") for o in offs}
    res = {}
    blocks = out.split("\n\n")
    for o, b in zip(offs, blocks):
        ls = [x for x in b.strip().splitlines() if x]
        if len(ls) < 2:
            res[o] = ("?", "?", "?"); continue
        short = lambda n: n.split("(")[0][-70:]
        where = ls[1].split("/")[-1]
        res[o] = (short(ls[0]), where.rsplit(":", 1)[0], short(ls[-2]))
    return res


by_mod = collections.defaultdict(set)
for m, off, via, c in rows:
    by_mod[m].add(off)
names = {m: resolve(m, sorted(offs)) for m, offs in by_mod.items()}
via_names = resolve(self_lib, sorted({r[2] for r in rows if r[2] != "0"})) if self_lib else {}
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
fn = collections.Counter(); line = collections.Counter(); outer = collections.Counter()
owner = collections.Counter()
owner_leaf = collections.defaultdict(collections.Counter)
self_short = (self_lib or "").split("/")[-1]
for m, off, via, c in rows:
    short = m.split("/")[-1]
    inner, where, out_fn = names[m].get(off, ("?", "?", "?"))
    fn[(short, inner)] += c
    if short == self_short:
        line[where] += c
        outer[out_fn] += c
    o = via_names.get(via, ("?", "?", "(not under this library)")) if via != "0" else ("?", "?", "(not under this library)")
    owner[(o[2], o[1])] += c
    owner_leaf[(o[2], o[1])][short if short != self_short else "self"] += c
print("== by the (innermost, possibly inlined) function running ==")
for k, c in fn.most_common(top):
    print("%6.2f%%  %-28s %s" % (100.0 * c / max(total, 1), k[0], k[1]))
print("== this library by outermost (non-inlined) function ==")
for k, c in outer.most_common(top):
    print("%6.2f%%  %s" % (100.0 * c / max(total, 1), k))
print("== this library by source line ==")
for k, c in line.most_common(top):
    print("%6.2f%%  %s" % (100.0 * c / max(total, 1), k))
print("== by the call site of this library on the stack (function, line) ==")
for k, c in owner.most_common(top):
    parts = ", ".join("%s %.1f%%" % (kk, 100.0 * cc / max(total, 1)) for kk, cc in owner_leaf[k].most_common(4))
    print("%6.2f%%  %-60s %-28s [%s]" % (100.0 * c / max(total, 1), k[0], k[1], parts))
