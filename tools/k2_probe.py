"""K1 / K2 kernel time against depth (diagnostic): 8 contigs of 100 kb at 25x / 50x / 100x, hipEvents on the library's stream."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hairsplitter_amd import api, synth

lib = api.load()
for depth in (25, 50, 100):
    contigs = [synth.make_contig(5, i, 100_000, 2, 0.01, depth, "ont") for i in range(8)]
    flat = api.FlatBatch(contigs)
    t = api.device_tensors(flat)
    pile, _ = api.pileup(t, flat)
    plan = api.tile_plan(flat)
    total = int(flat.contig_off[-1])
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda:0"); gpos = torch.zeros(total, dtype=torch.int64, device="cuda:0"); dep = torch.zeros(total, dtype=torch.int32, device="cuda:0")
    a = C.c_void_p(); b = C.c_void_p()
    lib.hs_event_create(C.byref(a)); lib.hs_event_create(C.byref(b))
    def run():
        cnt.zero_(); torch.cuda.synchronize()
        api._check(lib.hs_column_stats_tiled(api._p(pile), api._p(plan["off"]), api._p(plan["ent"]), C.c_int64(total), C.c_void_p(0), C.c_int32(4), api._p(cnt), api._p(gpos), api._p(dep), C.c_int32(total), C.c_int32(255 if depth < 100 else 0), C.c_void_p(0)))
    run(); lib.hs_device_synchronize()
    ms = []
    for _ in range(5):
        cnt.zero_(); torch.cuda.synchronize()
        lib.hs_event_record(a, C.c_void_p(0)); 
        api._check(lib.hs_column_stats_tiled(api._p(pile), api._p(plan["off"]), api._p(plan["ent"]), C.c_int64(total), C.c_void_p(0), C.c_int32(4), api._p(cnt), api._p(gpos), api._p(dep), C.c_int32(total), C.c_int32(255 if depth < 100 else 0), C.c_void_p(0)))
        lib.hs_event_record(b, C.c_void_p(0)); lib.hs_device_synchronize()
        m = C.c_float(0); lib.hs_event_elapsed_ms(a, b, C.byref(m)); ms.append(m.value)
    print(f"depth {depth}: aligned bp {flat.aligned_bp}, positions {total}, K2 {min(ms)*1e3:.1f} us (median {sorted(ms)[2]*1e3:.1f})", flush=True)
