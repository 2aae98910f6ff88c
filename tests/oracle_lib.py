"""ctypes access to oracle/_build/libhs_oracle.so (CPU restatement; test infrastructure only)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libhs_oracle.so"))
    return _lib


def _hp(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def pileup(flat):
    pile = np.zeros(max(flat.aligned_bp, 1), np.uint8)
    stats = np.zeros((max(flat.n_rec, 1), 4), np.int32)
    md = np.zeros(max(flat.n_contigs, 1), np.float32)
    lib().hso_pileup(_hp(flat.contig_seq, C.c_uint8), _hp(flat.contig_off, C.c_int64), C.c_int32(flat.n_contigs),
                     _hp(flat.read_seq, C.c_uint8), _hp(flat.read_off, C.c_int64), C.c_int32(flat.n_reads),
                     _hp(flat.rec_read, C.c_int32), _hp(flat.rec_pos, C.c_int32), _hp(flat.rec_strand, C.c_uint8),
                     _hp(flat.rec_cig_off, C.c_int64), _hp(flat.cigar, C.c_uint32), _hp(flat.contig_rec_off, C.c_int32),
                     _hp(flat.pile_off, C.c_int64), _hp(pile, C.c_uint8), _hp(stats, C.c_int32), _hp(md, C.c_float))
    return pile[:flat.aligned_bp], stats[:flat.n_rec], md[:flat.n_contigs]


def column_top3(flat, pile, c):
    r0, r1 = int(flat.contig_rec_off[c]), int(flat.contig_rec_off[c + 1])
    L = int(flat.contig_off[c + 1] - flat.contig_off[c])
    k0 = np.zeros(L, np.uint8); k1 = np.zeros(L, np.uint8)
    c0 = np.zeros(L, np.int32); c1 = np.zeros(L, np.int32); c2 = np.zeros(L, np.int32); d = np.zeros(L, np.int32)
    pile = np.ascontiguousarray(pile)
    lib().hso_column_top3(_hp(pile, C.c_uint8), _hp(flat.pile_off, C.c_int64), _hp(flat.rec_pos, C.c_int32), _hp(flat.rec_qend, C.c_int32),
                          C.c_int32(r0), C.c_int32(r1), C.c_int64(L), _hp(k0, C.c_uint8), _hp(k1, C.c_uint8), _hp(c0, C.c_int32),
                          _hp(c1, C.c_int32), _hp(c2, C.c_int32), _hp(d, C.c_int32))
    return k0, k1, c0, c1, c2, d


def call_variants_flags(flat, pile, c, mean_error, automatic_snp_threshold=0.33):
    """per position of contig c: bit 0 = candidate SNP (call_variants.cpp:525-529 with the spacing rule), bit 1 = automatic (:531)"""
    r0, r1 = int(flat.contig_rec_off[c]), int(flat.contig_rec_off[c + 1])
    L = int(flat.contig_off[c + 1] - flat.contig_off[c])
    flags = np.zeros(max(L, 1), np.uint8)
    pile = np.ascontiguousarray(pile)
    lib().hso_call_variants_flags(_hp(pile, C.c_uint8), _hp(flat.pile_off, C.c_int64), _hp(flat.rec_pos, C.c_int32), _hp(flat.rec_qend, C.c_int32),
                                  C.c_int32(r0), C.c_int32(r1), C.c_int64(L), C.c_float(mean_error), C.c_float(automatic_snp_threshold), _hp(flags, C.c_uint8))
    return flags[:L]


class _SrTapsO(C.Structure):
    _fields_ = [("n_tap", C.c_int32), ("tap_start", C.POINTER(C.c_int32)), ("tap_row0", C.POINTER(C.c_int64)), ("mask_ids", C.POINTER(C.c_int32)),
                ("run_begin", C.POINTER(C.c_int64)), ("run_snp", C.POINTER(C.c_int32)), ("run_labels", C.POINTER(C.c_int32)), ("third", C.POINTER(C.c_int32)),
                ("n_windows", C.c_int32), ("win_start", C.POINTER(C.c_int32)), ("win_end", C.POINTER(C.c_int32)), ("win_labels", C.POINTER(C.c_int32))]


def sr_contig_taps(ctg, window, error_rate, low_memory=False, seed=12345):
    """One contig (dict of numpy arrays as hairsplitter_amd.api.separate_reads(taps=True) leaves them in out["contigs"]) through the oracle's
    separate_reads_on_contig with its test taps on: per graph window its start, reads, per-SNP run labels (read ids), the labels of the run behind the
    small-cluster filter; and the finished labels of every window."""
    L = lib()
    L.hso_sr_contig_taps.argtypes = [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.c_float, C.c_int32, C.c_uint32, C.POINTER(C.POINTER(_SrTapsO))]
    L.hso_sr_taps_free.argtypes = [C.POINTER(_SrTapsO)]; L.hso_sr_taps_free.restype = None
    N, S = len(ctg["read_start"]), len(ctg["snp_pos"])
    tp = C.POINTER(_SrTapsO)()
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    keep = [np.ascontiguousarray(ctg[k]) for k in ("read_start", "read_end", "snp_pos", "snp_ref", "snp_alt", "col_off", "col_idx", "col_code")]
    rc = L.hso_sr_contig_taps(C.c_int32(N), C.c_int64(ctg["length"]), ptr(keep[0]), ptr(keep[1]), C.c_int32(S), ptr(keep[2]), ptr(keep[3]), ptr(keep[4]), ptr(keep[5]),
                              ptr(keep[6]), ptr(keep[7]), C.c_int32(window), C.c_float(error_rate), C.c_int32(1 if low_memory else 0), C.c_uint32(seed), C.byref(tp))
    assert rc == 0
    t = tp.contents
    a = lambda p, n, dt: np.ctypeslib.as_array(p, (max(n, 1),))[:n].astype(dt).copy()
    nt, nw = int(t.n_tap), int(t.n_windows)
    row0 = a(t.tap_row0, nt + 1, np.int64); rb = a(t.run_begin, nt + 1, np.int64)
    m = np.diff(row0)
    n_lab = int(sum(int(m[k]) * int(rb[k + 1] - rb[k]) for k in range(nt)))
    out = {"tap_start": a(t.tap_start, nt, np.int32), "tap_row0": row0, "mask_ids": a(t.mask_ids, int(row0[-1]), np.int32), "run_begin": rb,
           "run_snp": a(t.run_snp, int(rb[-1]), np.int32), "run_labels": a(t.run_labels, n_lab, np.int32), "third": a(t.third, int(row0[-1]), np.int32),
           "win_start": a(t.win_start, nw, np.int32), "win_end": a(t.win_end, nw, np.int32), "win_labels": a(t.win_labels, nw * N, np.int32).reshape(nw, N) if nw else np.zeros((0, N), np.int32)}
    L.hso_sr_taps_free(tp)
    return out


def column_partition_test(col_off, col_idx, col_code, col_contig, col_k0, col_k1, col_c1, col_is_cand, part_off, part_state_off, part_state,
                          n_reads_of_contig):
    n = len(col_contig)
    a = lambda x, dt: np.ascontiguousarray(x if len(x) else np.zeros(1), dt)
    col_off = a(col_off, np.int64); col_idx = a(col_idx, np.int32); col_code = a(col_code, np.uint8); col_contig = a(col_contig, np.int32)
    col_k0 = a(col_k0, np.uint8); col_k1 = a(col_k1, np.uint8); col_c1 = a(col_c1, np.int32); col_is_cand = a(col_is_cand, np.uint8)
    part_off = a(part_off, np.int32); part_state_off = a(part_state_off, np.int64); part_state = a(part_state, np.int8)
    nr = a(n_reads_of_contig, np.int32)
    keep = np.zeros(max(n, 1), np.uint8); chi = np.zeros(max(n, 1), np.float32); tab = np.zeros((max(n, 1), 4), np.int32)
    lib().hso_column_partition_test(_hp(col_off, C.c_int64), _hp(col_idx, C.c_int32), _hp(col_code, C.c_uint8), _hp(col_contig, C.c_int32),
                                    _hp(col_k0, C.c_uint8), _hp(col_k1, C.c_uint8), _hp(col_c1, C.c_int32), _hp(col_is_cand, C.c_uint8),
                                    C.c_int32(n), _hp(part_off, C.c_int32), _hp(part_state_off, C.c_int64), _hp(part_state, C.c_int8),
                                    _hp(nr, C.c_int32), C.c_int32(len(n_reads_of_contig)), _hp(keep, C.c_uint8), _hp(chi, C.c_float), _hp(tab, C.c_int32))
    return keep[:n], chi[:n], tab[:n]


def partition_pair_distance(state, more, less, part_off, part_n, pair_a, pair_b, threshold_p=2):
    """distance(Partition, Partition, threshold_p) of the oracle for every pair: [n_pairs, 6] = n00, n01, n10, n11, phased, augmented"""
    n = len(pair_a)
    a = lambda x, dt: np.ascontiguousarray(x if len(x) else np.zeros(1), dt)
    state = a(state, np.int8); more = a(more, np.int32); less = a(less, np.int32); part_off = a(part_off, np.int64); part_n = a(part_n, np.int32)
    pair_a = a(pair_a, np.int32); pair_b = a(pair_b, np.int32)
    out = np.zeros((max(n, 1), 6), np.int32)
    lib().hso_partition_pair_distance(_hp(state, C.c_int8), _hp(more, C.c_int32), _hp(less, C.c_int32), _hp(part_off, C.c_int64), _hp(part_n, C.c_int32),
                                      _hp(pair_a, C.c_int32), _hp(pair_b, C.c_int32), C.c_int32(n), C.c_int32(threshold_p), _hp(out, C.c_int32))
    return out[:n]


def simdiff(n_reads, snp_ref, snp_alt, col_off, col_idx, col_code):
    sim = np.zeros((n_reads, n_reads), np.int32); diff = np.zeros((n_reads, n_reads), np.int32)
    snp_ref = np.ascontiguousarray(snp_ref, np.uint8); snp_alt = np.ascontiguousarray(snp_alt, np.uint8)
    col_off = np.ascontiguousarray(col_off, np.int64); col_idx = np.ascontiguousarray(col_idx, np.int32); col_code = np.ascontiguousarray(col_code, np.uint8)
    lib().hso_simdiff(C.c_int32(n_reads), C.c_int32(len(snp_ref)), _hp(snp_ref, C.c_uint8), _hp(snp_alt, C.c_uint8), _hp(col_off, C.c_int64),
                      _hp(col_idx, C.c_int32), _hp(col_code, C.c_uint8), _hp(sim, C.c_int32), _hp(diff, C.c_int32))
    return sim, diff


def read_graph(sim, diff, mask, error_rate):
    """create_read_graph_matrix for one window -> list of sorted neighbour lists"""
    n = sim.shape[0]
    sim = np.ascontiguousarray(sim, np.int32); diff = np.ascontiguousarray(diff, np.int32); mask = np.ascontiguousarray(mask, np.uint8)
    off = np.zeros(n + 1, np.int32); adj = np.zeros(max(n * n, 1), np.int32)
    lib().hso_read_graph(C.c_int32(n), _hp(sim, C.c_int32), _hp(diff, C.c_int32), _hp(mask, C.c_uint8), C.c_float(error_rate), _hp(off, C.c_int32), _hp(adj, C.c_int32))
    return [adj[off[i]:off[i + 1]].tolist() for i in range(n)]


def chinese_whispers(adj_lists, mask, init, seed=12345):
    n = len(adj_lists)
    off = np.zeros(n + 1, np.int32); off[1:] = np.cumsum([len(a) for a in adj_lists])
    adj = np.ascontiguousarray(np.concatenate([np.asarray(a, np.int32) for a in adj_lists]) if off[-1] else np.zeros(1), np.int32)
    mask = np.ascontiguousarray(mask, np.uint8); init = np.ascontiguousarray(init, np.int32)
    out = np.zeros(n, np.int32); sw = C.c_int32(0)
    lib().hso_chinese_whispers(C.c_int32(n), _hp(off, C.c_int32), _hp(adj, C.c_int32), _hp(mask, C.c_uint8), _hp(init, C.c_int32),
                               C.c_uint32(seed), _hp(out, C.c_int32), C.byref(sw))
    return out, sw.value


def shuffled_order(n, seed=12345):
    out = np.zeros(max(n, 1), np.int32)
    lib().hso_shuffled_order(C.c_int32(n), C.c_uint32(seed), _hp(out, C.c_int32))
    return out[:n]


def edit_distance(q, t, mode):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    end = C.c_int32(0)
    d = lib().hso_edit_distance(_hp(q, C.c_uint8), C.c_int32(len(q)), _hp(t, C.c_uint8), C.c_int32(len(t)), C.c_int32(mode), C.byref(end))
    return d, end.value


def rh_order_u8(keys):
    k = np.ascontiguousarray(keys, np.uint8); out = np.zeros(300, np.uint8)
    n = lib().hso_rh_order_u8(_hp(k, C.c_uint8), C.c_int32(len(k)), _hp(out, C.c_uint8))
    return out[:n].tolist()


def rh_order_int(keys):
    k = np.ascontiguousarray(keys, np.int32); out = np.zeros(len(k) + 4, np.int32)
    n = lib().hso_rh_order_int(_hp(k, C.c_int32), C.c_int32(len(k)), _hp(out, C.c_int32))
    return out[:n].tolist()
