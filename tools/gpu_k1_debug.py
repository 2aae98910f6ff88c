"""diagnostic: where the pileup kernel's bytes differ from the oracle's (run on the GPU box)"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hairsplitter_amd import api, synth
import oracle_lib as ol
kind = sys.argv[1] if len(sys.argv) > 1 else "dip"
cs = [synth.make_contig(11, 0, 30_000, 2, 0.01, 40, "ont")] if kind == "dip" else [synth.make_contig(13, 0, 40_000, 2, 0.003, 25, "hifi")]
flat = api.FlatBatch(cs)
t = api.device_tensors(flat)
pile, stats = api.pileup(t, flat)
o_pile, o_stats, _ = ol.pileup(flat)
p = pile.cpu().numpy()
bad = np.flatnonzero(p != o_pile)
print("bytes", p.size, "differ", bad.size)
if bad.size:
    rec = np.searchsorted(flat.pile_off, bad, side="right") - 1
    off = bad - flat.pile_off[rec]
    span = (flat.pile_off[rec + 1] - flat.pile_off[rec])
    print("records with differences", np.unique(rec).size, "of", flat.n_rec, "strands of those", np.bincount(flat.rec_strand[np.unique(rec)], minlength=2))
    print("first 20:", [(int(r), int(o), int(sp), int(p[b]), int(o_pile[b])) for r, o, sp, b in zip(rec[:20], off[:20], span[:20], bad[:20])])
    print("distance from record end, histogram:", np.bincount(np.minimum(span - 1 - off, 40))[:41])
    # runs of consecutive bad bytes
    runs = np.split(bad, np.flatnonzero(np.diff(bad) != 1) + 1)
    print("runs", len(runs), "lengths", np.bincount(np.minimum([len(x) for x in runs], 40))[:41])
    print("got == 0 :", int((p[bad] == 0).sum()))
s = stats.cpu().numpy()
print("stats differ", int((s[:, :3] != o_stats[:, :3]).any(axis=1).sum()))
