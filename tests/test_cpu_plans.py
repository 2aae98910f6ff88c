"""Host-side launch plans of the C ABI against plain restatements (no GPU: these entry points only read host arrays)."""
import ctypes as C

import numpy as np

from hairsplitter_amd import api


def _tile_plan_ref(contig_off, contig_rec_off, rec_pos, rec_qend, pile_off):
    """hs_tile_plan as its header describes it: per 256-position tile of the concatenated contigs the records that overlap it, ascending
    record index, entry = (first position inside the tile, reference span, pileup byte of the tile's position 0)"""
    n_tiles = (int(contig_off[-1]) + 255) // 256
    per = [[] for _ in range(n_tiles)]
    for c in range(len(contig_off) - 1):
        for r in range(contig_rec_off[c], contig_rec_off[c + 1]):
            if rec_qend[r] <= rec_pos[r]:
                continue
            gs, ge = int(contig_off[c]) + int(rec_pos[r]), int(contig_off[c]) + int(rec_qend[r])
            for t in range(gs >> 8, ((ge - 1) >> 8) + 1):
                first = gs - t * 256
                per[t].append((r, first, ge - gs, int(pile_off[r]) - first))
    off = np.zeros(n_tiles + 1, np.int64)
    off[1:] = np.cumsum([len(x) for x in per])
    flat = [e for x in per for e in x]
    return off, flat


def _run(contig_len, recs):
    """recs: per contig a list of (pos, qend)"""
    contig_off = np.concatenate(([0], np.cumsum(contig_len))).astype(np.int64)
    contig_rec_off = np.concatenate(([0], np.cumsum([len(r) for r in recs]))).astype(np.int32)
    rec_pos = np.array([p for r in recs for p, _ in r], np.int32)
    rec_qend = np.array([q for r in recs for _, q in r], np.int32)
    span = np.maximum(rec_qend - rec_pos, 0).astype(np.int64)
    pile_off = np.concatenate(([0], np.cumsum(span))).astype(np.int64)
    lib = api.load()
    p_off = C.POINTER(C.c_int64)(); p_ent = C.c_void_p(); p_rec = C.POINTER(C.c_int32)(); n_tiles = C.c_int64(0)
    hp = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    assert lib.hs_tile_plan(hp(contig_off, C.c_int64), C.c_int32(len(contig_len)), hp(contig_rec_off, C.c_int32), hp(rec_pos, C.c_int32), hp(rec_qend, C.c_int32),
                            hp(pile_off, C.c_int64), C.byref(p_off), C.byref(p_ent), C.byref(p_rec), C.byref(n_tiles)) == 0
    nt = int(n_tiles.value)
    off = np.ctypeslib.as_array(p_off, shape=(nt + 1,)).copy()
    ne = int(off[-1])
    ent = np.ctypeslib.as_array(C.cast(p_ent, C.POINTER(C.c_int32)), shape=(max(ne, 1) * 4,)).copy().reshape(-1, 4)[:ne]
    rec = np.ctypeslib.as_array(p_rec, shape=(max(ne, 1),)).copy()[:ne]
    lib.hs_free_host(p_off); lib.hs_free_host(p_ent); lib.hs_free_host(p_rec)
    r_off, r_flat = _tile_plan_ref(contig_off, contig_rec_off, rec_pos, rec_qend, pile_off)
    assert np.array_equal(off, r_off)
    got = [(int(rec[k]), int(ent[k, 0]), int(ent[k, 1]), int(ent[k, 2].astype(np.uint32)) | (int(ent[k, 3]) << 32)) for k in range(ne)]
    assert got == r_flat


def test_tile_plan_equals_the_plain_walk_on_many_contigs():
    rng = np.random.default_rng(5)
    lens = rng.integers(300, 40_000, size=37)
    recs = []
    for L in lens:
        n = int(rng.integers(0, 120))
        pos = rng.integers(0, L, size=n)
        end = np.minimum(pos + rng.integers(0, 9000, size=n), L)
        recs.append(list(zip(pos.tolist(), end.tolist())))      # (unsorted starts, empty spans and contigs without records included)
    _run(lens, recs)


def test_tile_plan_with_contigs_that_share_one_tile():
    """three contigs inside one 256-position tile, a contig that ends on a tile boundary, records that cover their whole contig"""
    lens = [100, 40, 60, 56, 256, 1000, 3, 509]
    recs = [[(0, 100), (10, 90)], [(0, 40)], [(5, 60), (0, 1)], [], [(0, 256), (255, 256)], [(0, 1000), (300, 700), (999, 1000)], [(0, 3)], [(0, 509), (200, 300)]]
    _run(lens, recs)
