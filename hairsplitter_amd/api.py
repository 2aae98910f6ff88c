"""ctypes binding of include/hairsplitter_hip.h (the C ABI of the MI355X HairSplitter hot path).

There is no Python/CPU implementation behind these calls: if the HIP library is missing, or there is no GPU,
they raise. Device buffers are torch CUDA tensors (PyTorch-ROCm is only the allocator / stream owner here);
their `data_ptr()` is what crosses the ABI.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

_ROOT = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_ROOT, "lib", "libhairsplitter_hip.so")
if os.environ.get("HS_LIB_AB"):      # (tools/gpu_ab_lib.sh: another build of the same library, for A/B timing of a code change on one box)
    LIB_PATH = os.environ["HS_LIB_AB"]

# every symbol declared in include/hairsplitter_hip.h
SYMBOLS = [
    "hs_cv_batch_set_ploidy", "hs_version", "hs_last_error", "hs_device_count", "hs_warmup", "hs_set_device", "hs_device_synchronize", "hs_malloc", "hs_free",
    "hs_memcpy_h2d", "hs_memcpy_d2h", "hs_memset", "hs_event_create", "hs_event_destroy", "hs_event_record",
    "hs_event_elapsed_ms", "hs_pileup", "hs_pileup_plan", "hs_free_host", "hs_tile_plan", "hs_column_stats_tiled", "hs_cv_column_pass_taps", "hs_cv_taps_destroy", "hs_sr_run_taps", "hs_sr_taps_destroy", "hs_exclusive_scan_i32", "hs_gaf_from_files", "hs_gaf_from_labels", "hs_gro_to_gaf_main", "hs_column_partition_test", "hs_column_partition_last_counts", "hs_partition_pair_distance", "hs_snp_planes", "hs_simdiff", "hs_read_graphs",
    "hs_edit_distance", "hs_cv_batch_create", "hs_cv_batch_destroy", "hs_cv_batch_aligned_bp", "hs_cv_run",
    "hs_cv_result_destroy", "hs_cv_select", "hs_cv_run_range", "hs_cv_selection_destroy", "hs_sr_run", "hs_sr_run_cv", "hs_sr_run_cv_range", "hs_pipeline_create", "hs_pipeline_select", "hs_pipeline_run", "hs_pipeline_destroy", "hs_pipeline_thread_devices", "hs_cv_batch_device", "hs_sr_result_destroy", "hs_sr_window_size", "hs_call_variants_main", "hs_call_variants_epilogue",
    "hs_pipeline_run_fused", "hs_realign_paf", "hs_pipeline_set_option", "hs_pipeline_groups", "hs_pipeline_group_range", "hs_pipeline_group_cv", "hs_pipeline_sparse_labels", "hs_separate_reads_main", "hs_main_process_exits", "hs_kernel_name", "hs_kernel_stats_reset", "hs_kernel_stats_get", "hs_kernel_stats_every", "hs_host_wait_stats", "hs_devices", "hs_cv_run_host", "hs_edlib_hw_align", "hs_reattach_ends", "hs_trim_polished", "hs_free_strings", "hs_cut_gfa", "hs_gfa_to_fasta", "hs_cut_gfa_main", "hs_gfa2fa_main",
]

HS_NKERNELS = 28


class HsError(RuntimeError):
    pass


class CvResult(C.Structure):
    _fields_ = [("n_contigs", C.c_int32), ("mean_distance", C.POINTER(C.c_float)), ("depth", C.POINTER(C.c_float)),
                ("snp_off", C.POINTER(C.c_int64)), ("snp_pos", C.POINTER(C.c_int32)), ("snp_ref", C.POINTER(C.c_uint8)),
                ("snp_alt", C.POINTER(C.c_uint8)), ("snp_n_ref", C.POINTER(C.c_int32)), ("snp_n_alt", C.POINTER(C.c_int32)),
                ("col_off", C.POINTER(C.c_int64)), ("col_idx", C.POINTER(C.c_int32)),
                ("col_code", C.POINTER(C.c_uint8)), ("error_rate", C.c_float), ("n_contigs_with_error_rate", C.c_int32),
                ("t_device_ms", C.c_double), ("t_host_ms", C.c_double), ("t_kernel_ms", C.c_float * 4), ("t_kernel_k4_ms", C.c_float),
                ("n_columns_extracted", C.c_int64), ("n_columns_downloaded", C.c_int64), ("n_columns_downloaded_late", C.c_int64),
                ("entries_borrowed", C.c_int32)]


class KernelStats(C.Structure):
    _fields_ = [("ms", C.c_double * HS_NKERNELS), ("launches", C.c_int64 * HS_NKERNELS), ("bytes", C.c_int64 * HS_NKERNELS)]


class SrContig(C.Structure):
    _fields_ = [("length", C.c_int64), ("n_reads", C.c_int32), ("read_start", C.POINTER(C.c_int32)),
                ("read_end", C.POINTER(C.c_int32)), ("n_snps", C.c_int32), ("snp_pos", C.POINTER(C.c_int32)),
                ("snp_ref", C.POINTER(C.c_uint8)), ("snp_alt", C.POINTER(C.c_uint8)), ("col_off", C.POINTER(C.c_int64)),
                ("col_idx", C.POINTER(C.c_int32)), ("col_code", C.POINTER(C.c_uint8)), ("ploidy", C.c_int32)]


class SrResult(C.Structure):
    _fields_ = [("n_contigs", C.c_int32), ("win_off", C.POINTER(C.c_int64)), ("win_start", C.POINTER(C.c_int32)),
                ("win_end", C.POINTER(C.c_int32)), ("label_off", C.POINTER(C.c_int64)), ("labels", C.POINTER(C.c_int32)),
                ("t_device_ms", C.c_double), ("t_host_ms", C.c_double), ("n_cw_instances", C.c_int64),
                ("t_kernel_ms", C.c_float * 4), ("t_kernel_graph_ms", C.c_float), ("n_graph_rows_host", C.c_int64), ("n_windows_finished_on_host", C.c_int64),
                ("n_cw_sweeps", C.c_int64), ("cw_bytes", C.c_int64), ("graph_nnz", C.c_int64), ("n_graph_rows", C.c_int64), ("simdiff_bytes", C.c_int64)]


_lib = None


def load() -> C.CDLL:
    """Loads the in-tree HIP library; raises if it is missing or lacks a declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64.so.7; whichever HIP runtime is mapped first serves the whole process, and
    # torch only finds the GPU through its own copy. Import torch first so that this library binds to the same runtime.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    if not os.path.exists(LIB_PATH):
        raise HsError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950); "
                      "there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise HsError("libhairsplitter_hip.so lacks symbols: " + ", ".join(missing))
    lib.hs_version.restype = C.c_char_p
    lib.hs_last_error.restype = C.c_char_p
    lib.hs_cv_batch_aligned_bp.restype = C.c_int64
    lib.hs_cv_batch_aligned_bp.argtypes = [C.c_void_p]
    lib.hs_cv_batch_destroy.argtypes = [C.c_void_p]
    lib.hs_cv_batch_destroy.restype = None
    lib.hs_cv_result_destroy.argtypes = [C.c_void_p]
    lib.hs_cv_result_destroy.restype = None
    lib.hs_pipeline_destroy.argtypes = [C.c_void_p]
    lib.hs_pipeline_set_option.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
    lib.hs_pipeline_destroy.restype = None
    lib.hs_cv_selection_destroy.argtypes = [C.c_void_p]
    lib.hs_cv_selection_destroy.restype = None
    lib.hs_free_host.argtypes = [C.c_void_p]
    lib.hs_free_host.restype = None
    lib.hs_sr_result_destroy.argtypes = [C.c_void_p]
    lib.hs_sr_result_destroy.restype = None
    lib.hs_cv_run.argtypes = [C.c_void_p, C.c_float, C.c_int32, C.POINTER(C.POINTER(CvResult))]
    lib.hs_sr_run.argtypes = [C.POINTER(SrContig), C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_uint32, C.c_int32,
                              C.POINTER(C.POINTER(SrResult))]
    lib.hs_sr_run_cv.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_uint32, C.c_int32, C.c_int32,
                                 C.POINTER(C.POINTER(SrResult))]
    lib.hs_sr_window_size.argtypes = [C.POINTER(SrContig), C.c_int32, C.c_int32]
    lib.hs_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    lib.hs_event_record.argtypes = [C.c_void_p, C.c_void_p]
    lib.hs_event_destroy.argtypes = [C.c_void_p]
    lib.hs_kernel_name.restype = C.c_char_p
    lib.hs_kernel_name.argtypes = [C.c_int]
    lib.hs_kernel_stats_reset.restype = None
    lib.hs_kernel_stats_get.restype = None
    lib.hs_kernel_stats_get.argtypes = [C.POINTER(KernelStats)]
    _lib = lib
    return lib


def host_waits():
    """number of host waits for the device since the library was loaded"""
    n = C.c_int64(0); ms = C.c_double(0)
    lib = load()
    lib.hs_host_wait_stats.restype = None
    lib.hs_host_wait_stats(C.byref(n), C.byref(ms))
    return int(n.value)


def kernel_stats_reset():
    load().hs_kernel_stats_reset()


def kernel_stats_every(n: int):
    """Timing events on every n-th pipeline step only (counted from this call; 1 = every step, the default): hs_kernel_stats_every"""
    load().hs_kernel_stats_every(C.c_int32(int(n)))


def kernel_stats():
    """{kernel name: {"ms": summed launch durations (HIP events on the launch stream), "launches": n, "bytes": summed algorithmic bytes}}
    of everything the stage drivers launched since kernel_stats_reset()"""
    lib = load()
    st = KernelStats()
    lib.hs_kernel_stats_get(C.byref(st))
    out = {}
    for k in range(HS_NKERNELS):
        if st.launches[k]:
            out[lib.hs_kernel_name(k).decode()] = {"ms": float(st.ms[k]), "launches": int(st.launches[k]), "bytes": int(st.bytes[k])}
    return out


def _check(rc: int):
    if rc != 0:
        raise HsError(f"hairsplitter_hip error {rc}: {load().hs_last_error().decode()}")


def require_gpu():
    lib = load()
    if lib.hs_device_count() <= 0:
        raise HsError("no HIP device: the HairSplitter MI355X path has no CPU fallback")


def _p(t):
    """device pointer of a torch tensor (or None)"""
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _hp(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


# ------------------------------------------------------------------------------------------------
# flat batch (the layout the C ABI takes) from synth.ContigData
# ------------------------------------------------------------------------------------------------
class FlatBatch:
    """Host-side flattening of contigs + reads + alignment records (what parse_reads / parse_assembly / parse_SAM
    of the reference produce, input_output.cpp:39-536), in the C-ABI layout."""

    def __init__(self, contigs):
        cseq, coff = [], [0]
        rseq, roff = [], [0]
        rec_read, rec_pos, rec_strand, cig, cig_off = [], [], [], [], [0]
        contig_rec_off = [0]
        n_reads = 0
        for c in contigs:
            cseq.append(c.seq)
            coff.append(coff[-1] + len(c.seq))
            for r in c.reads:
                rseq.append(r)
                roff.append(roff[-1] + len(r))
            for a in c.alns:
                rec_read.append(n_reads + a.read)
                rec_pos.append(a.pos)
                rec_strand.append(1 if a.strand else 0)
                cig.append(a.cigar)
                cig_off.append(cig_off[-1] + len(a.cigar))
            n_reads += len(c.reads)
            contig_rec_off.append(len(rec_read))
        self.contig_seq = _np(np.concatenate(cseq) if cseq else np.zeros(0), np.uint8)
        self.contig_off = _np(coff, np.int64)
        self.read_seq = _np(np.concatenate(rseq) if rseq else np.zeros(0), np.uint8)
        self.read_off = _np(roff, np.int64)
        self.rec_read = _np(rec_read, np.int32)
        self.rec_pos = _np(rec_pos, np.int32)
        self.rec_strand = _np(rec_strand, np.uint8)
        self.rec_cig_off = _np(cig_off, np.int64)
        self.cigar = _np(np.concatenate(cig) if cig else np.zeros(0), np.uint32)
        self.contig_rec_off = _np(contig_rec_off, np.int32)
        self.n_contigs = len(contigs)
        self.n_reads = n_reads
        self.n_rec = len(rec_read)
        # derived: contig of each record, reference span, pileup offsets (same rule as hs_cv_batch_create)
        self.rec_contig = _np(np.repeat(np.arange(self.n_contigs), np.diff(self.contig_rec_off)), np.int32)
        ops = self.cigar & 0xF
        lens = (self.cigar >> 4).astype(np.int64)
        refc = np.where((ops == 0) | (ops == 2) | (ops == 7) | (ops == 8), lens, 0)
        csum = np.concatenate(([0], np.cumsum(refc)))
        span = csum[self.rec_cig_off[1:]] - csum[self.rec_cig_off[:-1]]
        L = np.diff(self.contig_off)[self.rec_contig] if self.n_rec else np.zeros(0, np.int64)
        pos = self.rec_pos.astype(np.int64)
        self.rec_refspan = _np(span, np.int64)
        qend = np.where(pos >= L, pos, np.minimum(pos + span, L))
        self.rec_qend = _np(qend, np.int32)
        self.pile_off = _np(np.concatenate(([0], np.cumsum(qend - pos))), np.int64)
        self.aligned_bp = int(self.pile_off[-1])


    _ARRAYS = ("contig_seq", "contig_off", "read_seq", "read_off", "rec_read", "rec_pos", "rec_strand", "rec_cig_off", "cigar", "contig_rec_off",
               "rec_contig", "rec_refspan", "rec_qend", "pile_off")

    def save(self, path: str):
        """the flat arrays as one .npz (a job generated in one process, run in another)"""
        np.savez(path, **{k: getattr(self, k) for k in self._ARRAYS})

    @classmethod
    def load(cls, path: str) -> "FlatBatch":
        o = object.__new__(cls)
        with np.load(path) as z:
            for k in cls._ARRAYS:
                setattr(o, k, np.ascontiguousarray(z[k]))
        o.n_contigs = len(o.contig_off) - 1
        o.n_reads = len(o.read_off) - 1
        o.n_rec = len(o.rec_read)
        o.aligned_bp = int(o.pile_off[-1])
        return o


class CvBatch:
    """hs_cv_batch: a FlatBatch resident in HBM."""

    def __init__(self, flat: FlatBatch):
        require_gpu()
        lib = load()
        self.flat = flat
        h = C.c_void_p()
        import time
        t0 = time.perf_counter()
        _check(lib.hs_cv_batch_create(_hp(flat.contig_seq, C.c_uint8), _hp(flat.contig_off, C.c_int64), C.c_int32(flat.n_contigs),
                                      _hp(flat.read_seq, C.c_uint8), _hp(flat.read_off, C.c_int64), C.c_int32(flat.n_reads),
                                      _hp(flat.rec_read, C.c_int32), _hp(flat.rec_pos, C.c_int32), _hp(flat.rec_strand, C.c_uint8),
                                      _hp(flat.rec_cig_off, C.c_int64), _hp(flat.cigar, C.c_uint32),
                                      _hp(flat.contig_rec_off, C.c_int32), C.byref(h)))
        self.create_s = time.perf_counter() - t0      # host buffers -> HBM + the launch plans (CIGAR spans, tile plan, pileup tasks)
        self.handle = h

    @property
    def aligned_bp(self) -> int:
        return int(load().hs_cv_batch_aligned_bp(self.handle))

    def set_ploidy(self, ploidy: Optional[Sequence[int]]):
        """Ploidy of every contig (0 = none) for the in-memory stage 3 -> 4 paths (run_pipeline, PipelineGroups.run): what the
        <ploidy_of_contigs> file is to HS_separate_reads. None clears it."""
        if ploidy is None:
            _check(load().hs_cv_batch_set_ploidy(self.handle, None))
            return
        p = np.ascontiguousarray(ploidy, dtype=np.int32)
        assert p.size == self.flat.n_contigs
        _check(load().hs_cv_batch_set_ploidy(self.handle, _hp(p, C.c_int32)))

    def run(self, automatic_snp_threshold: float = 0.33, n_threads: int = 0) -> Dict:
        """Stage 3 on the resident batch == HS_call_variants without the file I/O (call_variants.cpp:1276-1381)."""
        lib = load()
        res = C.POINTER(CvResult)()
        _check(lib.hs_cv_run(self.handle, C.c_float(automatic_snp_threshold), C.c_int32(n_threads), C.byref(res)))
        r = res.contents
        Cn = r.n_contigs
        snp_off = np.ctypeslib.as_array(r.snp_off, (Cn + 1,)).copy()
        S = int(snp_off[-1])
        col_off = np.ctypeslib.as_array(r.col_off, (S + 1,)).copy()
        E = int(col_off[-1])
        out = {
            "mean_distance": np.ctypeslib.as_array(r.mean_distance, (Cn,)).copy() if Cn else np.zeros(0, np.float32),
            "depth": np.ctypeslib.as_array(r.depth, (Cn,)).copy() if Cn else np.zeros(0, np.float32),
            "snp_off": snp_off,
            "snp_pos": np.ctypeslib.as_array(r.snp_pos, (max(S, 1),))[:S].copy(),
            "snp_ref": np.ctypeslib.as_array(r.snp_ref, (max(S, 1),))[:S].copy(),
            "snp_alt": np.ctypeslib.as_array(r.snp_alt, (max(S, 1),))[:S].copy(),
            "col_off": col_off,
            "col_idx": np.ctypeslib.as_array(r.col_idx, (max(E, 1),))[:E].copy(),
            "col_code": np.ctypeslib.as_array(r.col_code, (max(E, 1),))[:E].copy(),
            "error_rate": float(r.error_rate),
            "t_device_ms": float(r.t_device_ms), "t_host_ms": float(r.t_host_ms),
            "t_kernel_ms": [float(x) for x in r.t_kernel_ms],
            "n_columns_extracted": int(r.n_columns_extracted), "n_columns_downloaded": int(r.n_columns_downloaded),
            "n_columns_downloaded_late": int(r.n_columns_downloaded_late),
        }
        lib.hs_cv_result_destroy(res)
        return out

    def run_pipeline(self, automatic_snp_threshold: float = 0.33, n_threads: int = 0, error_rate_fn=None,
                     rarest_strain_abundance: float = 0.01, low_memory: bool = False, amplicon: bool = False, seed: int = 12345,
                     window_size: int = 0):
        """Stage 3 then stage 4 in-process on the resident batch (hs_cv_run -> hs_sr_run_cv): no .col text round trip.
        `error_rate_fn(cv_dict)` maps the stage-3 result to the error rate handed to stage 4 (default: what
        hairsplitter.py does with error_rate.txt: 6 significant digits, capped at 0.15)."""
        lib = load()
        res = C.POINTER(CvResult)()
        _check(lib.hs_cv_run(self.handle, C.c_float(automatic_snp_threshold), C.c_int32(n_threads), C.byref(res)))
        try:
            r = res.contents
            Cn = r.n_contigs
            cv = {"mean_distance": np.ctypeslib.as_array(r.mean_distance, (max(Cn, 1),))[:Cn].copy(), "error_rate": float(r.error_rate),
                  "n_snps": int(np.ctypeslib.as_array(r.snp_off, (Cn + 1,))[-1]),
                  "t_device_ms": float(r.t_device_ms), "t_host_ms": float(r.t_host_ms), "t_kernel_ms": [float(x) for x in r.t_kernel_ms], "t_kernel_k4_ms": float(r.t_kernel_k4_ms),
                  "n_columns_extracted": int(r.n_columns_extracted), "n_columns_downloaded": int(r.n_columns_downloaded),
                  "n_columns_downloaded_late": int(r.n_columns_downloaded_late)}
            e = error_rate_fn(cv) if error_rate_fn is not None else min(float("%g" % cv["error_rate"]), 0.15)
            sres = C.POINTER(SrResult)()
            _check(lib.hs_sr_run_cv(self.handle, res, C.c_float(e), C.c_float(rarest_strain_abundance), C.c_int32(1 if low_memory else 0),
                                    C.c_int32(1 if amplicon else 0), C.c_uint32(seed), C.c_int32(n_threads), C.c_int32(window_size),
                                    C.byref(sres)))
            sr = _sr_result_to_dict(sres, Cn)
            lib.hs_sr_result_destroy(sres)
        finally:
            lib.hs_cv_result_destroy(res)
        return cv, sr

    # -- the two halves of run_pipeline, for callers that drive several batches concurrently (run_pipeline_groups) --
    def _cv(self, automatic_snp_threshold, n_threads):
        lib = load()
        res = C.POINTER(CvResult)()
        _check(lib.hs_cv_run(self.handle, C.c_float(automatic_snp_threshold), C.c_int32(n_threads), C.byref(res)))
        r = res.contents
        Cn = r.n_contigs
        cv = {"mean_distance": np.ctypeslib.as_array(r.mean_distance, (max(Cn, 1),))[:Cn].copy(), "error_rate": float(r.error_rate),
              "n_snps": int(np.ctypeslib.as_array(r.snp_off, (Cn + 1,))[-1]),
              "t_device_ms": float(r.t_device_ms), "t_host_ms": float(r.t_host_ms), "t_kernel_ms": [float(x) for x in r.t_kernel_ms], "t_kernel_k4_ms": float(r.t_kernel_k4_ms),
                  "n_columns_extracted": int(r.n_columns_extracted), "n_columns_downloaded": int(r.n_columns_downloaded),
                  "n_columns_downloaded_late": int(r.n_columns_downloaded_late)}
        return res, cv

    def _sr(self, res, e, n_threads, rarest_strain_abundance, low_memory, amplicon, seed, window_size):
        lib = load()
        try:
            sres = C.POINTER(SrResult)()
            _check(lib.hs_sr_run_cv(self.handle, res, C.c_float(e), C.c_float(rarest_strain_abundance), C.c_int32(1 if low_memory else 0),
                                    C.c_int32(1 if amplicon else 0), C.c_uint32(seed), C.c_int32(n_threads), C.c_int32(window_size),
                                    C.byref(sres)))
            sr = _sr_result_to_dict(sres, res.contents.n_contigs)
            lib.hs_sr_result_destroy(sres)
        finally:
            lib.hs_cv_result_destroy(res)
        return sr

    def close(self):
        if self.handle:
            load().hs_cv_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CvSelection(C.Structure):
    _fields_ = [("n_selected", C.c_int64), ("t_kernel_ms", C.c_float * 4), ("t_device_ms", C.c_double), ("t_host_ms", C.c_double),
                ("impl", C.c_void_p)]


class PipelineStats(C.Structure):
    _fields_ = [("n_snps", C.c_int64), ("n_cw_instances", C.c_int64), ("n_graph_rows_host", C.c_int64),
                ("t_device_ms", C.c_double), ("t_host_ms", C.c_double), ("t_kernel_cv_ms", C.c_float * 4), ("t_kernel_k4_ms", C.c_float),
                ("t_kernel_sr_ms", C.c_float * 4), ("t_kernel_graph_ms", C.c_float),
                ("n_columns_extracted", C.c_int64), ("n_columns_downloaded", C.c_int64), ("n_columns_downloaded_late", C.c_int64),
                ("n_cw_sweeps", C.c_int64), ("cw_bytes", C.c_int64), ("graph_nnz", C.c_int64), ("n_graph_rows", C.c_int64), ("simdiff_bytes", C.c_int64)]


class PipelineGroups:
    """One resident batch whose contigs are processed as G consecutive groups (hs_pipeline_*). The streaming kernels (CIGAR scan,
    pileup, column statistics) run ONCE over the whole batch; everything after them runs per group, one persistent host
    thread and one HIP stream per group inside the library, so that the sequential host sections and device waits of one group
    overlap the work of the others. Contigs are independent in both stages (call_variants.cpp:1280, separate_reads.cpp:1508);
    the one cross-contig quantity, the error rate, is formed between the stages over ALL contigs in contig order, exactly as
    for a single call."""

    def __init__(self, contigs, n_groups):
        self.flat = contigs if isinstance(contigs, FlatBatch) else FlatBatch(contigs)
        self.batch = CvBatch(self.flat)
        self.handle = C.c_void_p()
        _check(load().hs_pipeline_create(self.batch.handle, C.c_int32(n_groups), C.byref(self.handle)))
        self.aligned_bp = self.flat.aligned_bp
        self.total_len = int(self.flat.contig_off[-1])
        self._owns_batch = True

    def keep_columns(self, on=True):
        """HS_PIPELINE_KEEP_COLUMNS: the entries of the SNP columns (.col's payload) are brought to the host inside every call too"""
        _check(load().hs_pipeline_set_option(self.handle, C.c_int32(1), C.c_int64(1 if on else 0)))

    def sparse_labels(self, on=True):
        """HS_PIPELINE_SPARSE_LABELS: results carry sr["sparse"] = (win_row_off, ids, labels) -- per window the reads it holds and their
        labels, what the .gro lists -- instead of the dense array sr["labels"] (None then). The arrays view the pipeline's memory and
        stay valid until its next call."""
        _check(load().hs_pipeline_set_option(self.handle, C.c_int32(2), C.c_int64(1 if on else 0)))
        self._sparse = bool(on)

    def _attach_sparse(self, sr):
        if not getattr(self, "_sparse", False):
            return sr
        off, ids, lab = C.POINTER(C.c_int64)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        nw, nr = C.c_int64(0), C.c_int64(0)
        _check(load().hs_pipeline_sparse_labels(self.handle, C.byref(off), C.byref(ids), C.byref(lab), C.byref(nw), C.byref(nr)))
        W, R = int(nw.value), int(nr.value)
        if R == 0:      # (an empty std::vector's data() may be NULL)
            sr["sparse"] = (np.ctypeslib.as_array(off, (W + 1,)), np.zeros(0, np.int32), np.zeros(0, np.int32))
            return sr
        sr["sparse"] = (np.ctypeslib.as_array(off, (W + 1,)), np.ctypeslib.as_array(ids, (R,)), np.ctypeslib.as_array(lab, (R,)))
        return sr

    def sibling(self, n_groups):
        """Another pipeline over the SAME resident batch with its own number of contig groups (bench.py: one group, where every
        kernel runs alone, to read the kernels' own durations next to those of the default run)."""
        o = object.__new__(PipelineGroups)
        o.flat, o.batch, o.aligned_bp, o.total_len = self.flat, self.batch, self.aligned_bp, self.total_len
        o.handle = C.c_void_p()
        o._owns_batch = False      # (the batch belongs to the pipeline it was made from: close() of the sibling leaves it alone)
        _check(load().hs_pipeline_create(self.batch.handle, C.c_int32(n_groups), C.byref(o.handle)))
        return o

    def run(self, automatic_snp_threshold=0.33, n_threads=0, error_rate_fn=None, rarest_strain_abundance=0.01, low_memory=False,
            amplicon=False, seed=12345, window_size=0):
        import time
        lib = load()
        Cn = self.flat.n_contigs
        md = np.zeros(max(Cn, 1), np.float32)
        st = PipelineStats()
        t_0 = time.perf_counter()
        _check(lib.hs_pipeline_select(self.handle, _hp(md, C.c_float), C.byref(st)))
        t_1 = time.perf_counter()
        md = md[:Cn]
        # call_variants.cpp:1312-1315,1377: float sum in contig order / number of contigs with a positive distance
        from .dist import mean_of_positive_f32
        cv = {"mean_distance": md, "error_rate": mean_of_positive_f32(md)}
        e = error_rate_fn(cv) if error_rate_fn is not None else min(float("%g" % cv["error_rate"]), 0.15)
        sres = C.POINTER(SrResult)()   # window_size <= 0: hs_pipeline_run chooses it over the whole batch (same rule as self.window_size())
        t_2 = time.perf_counter()
        _check(lib.hs_pipeline_run(self.handle, C.c_float(automatic_snp_threshold), C.c_float(e), C.c_float(rarest_strain_abundance),
                                   C.c_int32(1 if low_memory else 0), C.c_int32(1 if amplicon else 0), C.c_uint32(seed), C.c_int32(n_threads),
                                   C.c_int32(window_size), C.byref(sres), C.byref(st)))
        t_3 = time.perf_counter()
        cv.update({"n_snps": int(st.n_snps), "t_device_ms": float(st.t_device_ms), "t_host_ms": float(st.t_host_ms),
                   "t_kernel_ms": [float(x) for x in st.t_kernel_cv_ms], "t_kernel_k4_ms": float(st.t_kernel_k4_ms),
                   "n_columns_extracted": int(st.n_columns_extracted), "n_columns_downloaded": int(st.n_columns_downloaded),
                   "n_columns_downloaded_late": int(st.n_columns_downloaded_late)})
        sr = self._attach_sparse(_sr_result_to_dict(sres, Cn, take_ownership=True))   # the labels stay where the library put them
        sr["wall_ms"] = {"select": (t_1 - t_0) * 1e3, "between": (t_2 - t_1) * 1e3, "groups": (t_3 - t_2) * 1e3, "collect": (time.perf_counter() - t_3) * 1e3}
        return cv, sr

    def run_fused(self, automatic_snp_threshold=0.33, n_threads=0, rarest_strain_abundance=0.01, low_memory=False, amplicon=False, seed=12345,
                  window_size=0):
        """hs_pipeline_run_fused: the same job in one library call (single process: the error rate is formed inside)"""
        import time
        lib = load()
        Cn = self.flat.n_contigs
        md = np.zeros(max(Cn, 1), np.float32)
        st = PipelineStats()
        er = C.c_float(0)
        sres = C.POINTER(SrResult)()
        t_0 = time.perf_counter()
        _check(lib.hs_pipeline_run_fused(self.handle, C.c_float(automatic_snp_threshold), C.c_float(rarest_strain_abundance), C.c_int32(1 if low_memory else 0),
                                         C.c_int32(1 if amplicon else 0), C.c_uint32(seed), C.c_int32(n_threads), C.c_int32(window_size), _hp(md, C.c_float),
                                         C.byref(er), C.byref(sres), C.byref(st)))
        t_1 = time.perf_counter()
        cv = {"mean_distance": md[:Cn], "error_rate": float(er.value), "error_rate_is_final": True, "n_snps": int(st.n_snps), "t_device_ms": float(st.t_device_ms),
              "t_host_ms": float(st.t_host_ms), "t_kernel_ms": [float(x) for x in st.t_kernel_cv_ms], "t_kernel_k4_ms": float(st.t_kernel_k4_ms),
              "n_columns_extracted": int(st.n_columns_extracted), "n_columns_downloaded": int(st.n_columns_downloaded),
              "n_columns_downloaded_late": int(st.n_columns_downloaded_late)}
        sr = self._attach_sparse(_sr_result_to_dict(sres, Cn, take_ownership=True))
        sr["wall_ms"] = {"select": 0.0, "between": 0.0, "groups": (t_1 - t_0) * 1e3, "collect": (time.perf_counter() - t_1) * 1e3}
        return cv, sr

    def window_size(self, amplicon=False):
        """choose_window_size over the whole job (separate_reads.cpp:1466-1498): it depends on every read of every contig.
        READ limits are (POS-1, POS + reference span) (input_output.cpp:503-511), the length sum is a wrapping int."""
        f = self.flat
        if amplicon:
            return int(np.diff(f.contig_off).max()) if f.n_contigs else 0
        lens = f.rec_refspan + 2
        if lens.size == 0:
            return 2000
        total = int(lens.sum()) & 0xFFFFFFFF
        if total >= 1 << 31:
            total -= 1 << 32
        mean = total / float(lens.size)
        above = int((lens > 4000).sum())
        if above < 20 and 2000 < mean < 4000:
            return 1000
        if above < 20 and mean < 2000:
            return 500
        return 2000

    def close(self):
        if self.handle:
            load().hs_pipeline_destroy(self.handle)
            self.handle = None
        if getattr(self, "_owns_batch", True):
            self.batch.close()


class _SrResultOwner:
    """Frees an hs_sr_result when the last array that views its memory is gone"""

    def __init__(self, res):
        self.res = res

    def __del__(self):
        try:
            load().hs_sr_result_destroy(self.res)
        except Exception:
            pass


class _SrTaps(C.Structure):
    _fields_ = [("n_windows", C.c_int32), ("win_contig", C.POINTER(C.c_int32)), ("win_start", C.POINTER(C.c_int32)), ("win_row0", C.POINTER(C.c_int64)),
                ("mask_ids", C.POINTER(C.c_int32)), ("run_begin", C.POINTER(C.c_int64)), ("run_snp", C.POINTER(C.c_int32)), ("run_off", C.POINTER(C.c_int64)),
                ("run_labels", C.POINTER(C.c_int32)), ("third", C.POINTER(C.c_int32))]


def _sr_result_to_dict(res, Cn, take_ownership=False):
    """take_ownership: the labels (tens of MB per batch) are returned as a view of the C result instead of a copy; the
    result is destroyed when that array (and every view of it) is gone, and the caller must not destroy it."""
    r = res.contents
    win_off = np.ctypeslib.as_array(r.win_off, (Cn + 1,)).copy()
    W = int(win_off[-1])
    label_off = np.ctypeslib.as_array(r.label_off, (W + 1,)).copy()
    NL = int(label_off[-1])
    return {
        "win_off": win_off,
        "win_start": np.ctypeslib.as_array(r.win_start, (max(W, 1),))[:W].copy(),
        "win_end": np.ctypeslib.as_array(r.win_end, (max(W, 1),))[:W].copy(),
        "label_off": label_off,
        "labels": (None if not r.labels else (_owned_view_i32(res, r.labels, NL) if take_ownership else np.ctypeslib.as_array(r.labels, (max(NL, 1),))[:NL].copy())),
        "_owner": (_SrResultOwner(res) if (take_ownership and not r.labels) else None),      # (no label array to carry the ownership: the dict does)
        "t_device_ms": float(r.t_device_ms), "t_host_ms": float(r.t_host_ms), "n_cw_instances": int(r.n_cw_instances),
        "t_kernel_ms": [float(x) for x in r.t_kernel_ms], "t_kernel_graph_ms": float(r.t_kernel_graph_ms),
        "n_graph_rows_host": int(r.n_graph_rows_host), "n_windows_finished_on_host": int(r.n_windows_finished_on_host),
        "n_cw_sweeps": int(r.n_cw_sweeps), "cw_bytes": int(r.cw_bytes), "graph_nnz": int(r.graph_nnz), "n_graph_rows": int(r.n_graph_rows),
        "simdiff_bytes": int(r.simdiff_bytes),
    }


def _owned_view_i32(res, ptr, n):
    buf = (C.c_int32 * max(n, 1)).from_address(C.addressof(ptr.contents))
    buf._hs_owner = _SrResultOwner(res)      # the ndarray's base is this ctypes array, which keeps the owner alive
    return np.ctypeslib.as_array(buf)[:n]


def separate_reads(cv_out: Dict, flat: FlatBatch, error_rate: float, low_memory: bool = False, amplicon: bool = False,
                   seed: int = 12345, n_threads: int = 0, ploidy: Optional[Sequence[int]] = None,
                   rarest_strain_abundance: float = 0.0, window_size: Optional[int] = None, taps: bool = False) -> Dict:
    """(taps=True: through hs_sr_run_taps -- out["taps"] holds what the kernels of the clustering chain left, out["contigs"] the per-contig
    inputs as numpy arrays, for the tests.)
    Stage 4 on the in-memory result of stage 3 == HS_separate_reads without the .col round trip
    (separate_reads.cpp:1440-1739). READ limits are (position_2_1, position_2_2) of the records
    (call_variants.cpp:1186-1189 -> separate_reads.cpp:176-179)."""
    require_gpu()
    lib = load()
    Cn = flat.n_contigs
    arr = (SrContig * Cn)()
    keep = []   # keep numpy buffers alive
    per_contig = []
    ops = flat.cigar & 0xF
    lens = (flat.cigar >> 4).astype(np.int64)
    refc = np.where((ops == 0) | (ops == 2) | (ops == 7) | (ops == 8), lens, 0)
    csum = np.concatenate(([0], np.cumsum(refc)))
    span = csum[flat.rec_cig_off[1:]] - csum[flat.rec_cig_off[:-1]]
    for c in range(Cn):
        r0, r1 = int(flat.contig_rec_off[c]), int(flat.contig_rec_off[c + 1])
        rs = _np(flat.rec_pos[r0:r1], np.int32)
        re = _np(flat.rec_pos[r0:r1].astype(np.int64) + 1 + span[r0:r1], np.int32)   # pos2_1 + length_contig
        s0, s1 = int(cv_out["snp_off"][c]), int(cv_out["snp_off"][c + 1])
        e0 = int(cv_out["col_off"][s0])
        snp_pos = _np(cv_out["snp_pos"][s0:s1], np.int32)
        snp_ref = _np(cv_out["snp_ref"][s0:s1], np.uint8)
        snp_alt = _np(cv_out["snp_alt"][s0:s1], np.uint8)
        col_off = _np(cv_out["col_off"][s0:s1 + 1] - e0, np.int64)
        e1 = e0 + int(col_off[-1])
        col_idx = _np(cv_out["col_idx"][e0:e1], np.int32)
        col_code = _np(cv_out["col_code"][e0:e1], np.uint8)
        if rarest_strain_abundance > 0 and s1 > s0:
            # parse_column_file drops SNPs whose second base is rarer than the threshold (separate_reads.cpp:167)
            seg = np.repeat(np.arange(s1 - s0), np.diff(col_off))
            n_ref = np.bincount(seg, weights=(col_code == snp_ref[seg]), minlength=s1 - s0)
            n_alt = np.bincount(seg, weights=(col_code == snp_alt[seg]) & (col_code != snp_ref[seg]), minlength=s1 - s0)
            keep_snp = n_alt.astype(np.float32) >= np.float32(rarest_strain_abundance) * (n_ref + n_alt).astype(np.float32)
            if not keep_snp.all():
                km = keep_snp[seg]
                lens = np.diff(col_off)[keep_snp]
                snp_pos, snp_ref, snp_alt = snp_pos[keep_snp], snp_ref[keep_snp], snp_alt[keep_snp]
                col_idx, col_code = _np(col_idx[km], np.int32), _np(col_code[km], np.uint8)
                col_off = _np(np.concatenate(([0], np.cumsum(lens))), np.int64)
                snp_pos, snp_ref, snp_alt = _np(snp_pos, np.int32), _np(snp_ref, np.uint8), _np(snp_alt, np.uint8)
                s1 = s0 + len(snp_pos)
        keep += [rs, re, snp_pos, snp_ref, snp_alt, col_off, col_idx, col_code]
        per_contig.append({"read_start": rs, "read_end": re, "snp_pos": snp_pos, "snp_ref": snp_ref, "snp_alt": snp_alt, "col_off": col_off, "col_idx": col_idx,
                           "col_code": col_code, "length": int(flat.contig_off[c + 1] - flat.contig_off[c])})
        a = arr[c]
        a.length = int(flat.contig_off[c + 1] - flat.contig_off[c])
        a.n_reads = r1 - r0
        a.read_start = _hp(rs, C.c_int32); a.read_end = _hp(re, C.c_int32)
        a.n_snps = s1 - s0
        a.snp_pos = _hp(snp_pos, C.c_int32); a.snp_ref = _hp(snp_ref, C.c_uint8); a.snp_alt = _hp(snp_alt, C.c_uint8)
        a.col_off = _hp(col_off, C.c_int64); a.col_idx = _hp(col_idx, C.c_int32); a.col_code = _hp(col_code, C.c_uint8)
        a.ploidy = int(ploidy[c]) if ploidy is not None else 0
    w = lib.hs_sr_window_size(arr, C.c_int32(Cn), C.c_int32(1 if amplicon else 0)) if window_size is None else int(window_size)
    res = C.POINTER(SrResult)()
    tp = C.POINTER(_SrTaps)()
    if taps:
        lib.hs_sr_run_taps.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_uint32, C.c_int32, C.POINTER(C.POINTER(SrResult)), C.POINTER(C.POINTER(_SrTaps))]
        lib.hs_sr_taps_destroy.argtypes = [C.POINTER(_SrTaps)]; lib.hs_sr_taps_destroy.restype = None
        _check(lib.hs_sr_run_taps(C.cast(arr, C.c_void_p), C.c_int32(Cn), C.c_int32(w), C.c_float(error_rate), C.c_int32(1 if low_memory else 0),
                                  C.c_uint32(seed), C.c_int32(n_threads), C.byref(res), C.byref(tp)))
    else:
        _check(lib.hs_sr_run(arr, C.c_int32(Cn), C.c_int32(w), C.c_float(error_rate), C.c_int32(1 if low_memory else 0),
                             C.c_uint32(seed), C.c_int32(n_threads), C.byref(res)))
    r = res.contents
    win_off = np.ctypeslib.as_array(r.win_off, (Cn + 1,)).copy()
    W = int(win_off[-1])
    label_off = np.ctypeslib.as_array(r.label_off, (W + 1,)).copy()
    NL = int(label_off[-1])
    out = {
        "window_size": int(w), "win_off": win_off,
        "win_start": np.ctypeslib.as_array(r.win_start, (max(W, 1),))[:W].copy(),
        "win_end": np.ctypeslib.as_array(r.win_end, (max(W, 1),))[:W].copy(),
        "label_off": label_off,
        "labels": np.ctypeslib.as_array(r.labels, (max(NL, 1),))[:NL].copy(),
        "t_device_ms": float(r.t_device_ms), "t_host_ms": float(r.t_host_ms), "n_cw_instances": int(r.n_cw_instances),
        "t_kernel_ms": [float(x) for x in r.t_kernel_ms], "t_kernel_graph_ms": float(r.t_kernel_graph_ms),
        "n_graph_rows_host": int(r.n_graph_rows_host), "n_windows_finished_on_host": int(r.n_windows_finished_on_host),
        "n_cw_sweeps": int(r.n_cw_sweeps), "cw_bytes": int(r.cw_bytes), "graph_nnz": int(r.graph_nnz), "n_graph_rows": int(r.n_graph_rows),
        "simdiff_bytes": int(r.simdiff_bytes),
    }
    lib.hs_sr_result_destroy(res)
    if taps:
        t = tp.contents
        Wc = int(t.n_windows)
        a = lambda ptr, n, dt: np.ctypeslib.as_array(ptr, (max(n, 1),))[:n].astype(dt).copy()
        row0 = a(t.win_row0, Wc + 1, np.int64); rb = a(t.run_begin, Wc + 1, np.int64)
        n_runs = int(rb[-1]); ro = a(t.run_off, n_runs + 1, np.int64)
        out["taps"] = {"win_contig": a(t.win_contig, Wc, np.int32), "win_start": a(t.win_start, Wc, np.int32), "win_row0": row0, "mask_ids": a(t.mask_ids, int(row0[-1]), np.int32),
                       "run_begin": rb, "run_snp": a(t.run_snp, n_runs, np.int32), "run_off": ro, "run_labels": a(t.run_labels, int(ro[-1]), np.int32),
                       "third": a(t.third, int(row0[-1]), np.int32)}
        out["contigs"] = per_contig
        lib.hs_sr_taps_destroy(tp)
    return out


# ------------------------------------------------------------------------------------------------
# kernel-level wrappers on torch CUDA tensors (used by the parity tests and bench.py's roofline leg)
# ------------------------------------------------------------------------------------------------
def device_tensors(flat: FlatBatch, device="cuda:0"):
    import torch
    t = {}
    for k in ("contig_seq", "contig_off", "read_seq", "read_off", "rec_read", "rec_contig", "rec_pos", "rec_strand",
              "rec_cig_off", "pile_off", "contig_rec_off", "rec_qend"):
        t[k] = torch.from_numpy(getattr(flat, k)).to(device)
    t["cigar"] = torch.from_numpy(flat.cigar.view(np.int32)).to(device)
    return t


def pileup_plan(flat: FlatBatch, ev_per_task: int = 4096):
    """hs_pileup_plan: (rec_chunk_off int64[n_rec+1], task_rec int32[], task_ev0 int32[])."""
    lib = load()
    chunk_off = np.zeros(flat.n_rec + 1, np.int64)
    n = C.c_int32(0)
    tr = C.POINTER(C.c_int32)(); te = C.POINTER(C.c_int32)()
    _check(lib.hs_pileup_plan(_hp(flat.rec_cig_off, C.c_int64), _hp(flat.cigar, C.c_uint32), C.c_int32(flat.n_rec), C.c_int32(ev_per_task),
                              _hp(chunk_off, C.c_int64), C.byref(n), C.byref(tr), C.byref(te)))
    nt = n.value
    task_rec = np.ctypeslib.as_array(tr, (max(nt, 1),))[:nt].copy()
    task_ev0 = np.ctypeslib.as_array(te, (max(nt, 1),))[:nt].copy()
    lib.hs_free_host(tr); lib.hs_free_host(te)
    return chunk_off, task_rec, task_ev0


def pileup(t, flat: FlatBatch, ev_per_task: int = 4096):
    """K0+K1 on device tensors; returns (pile u8[aligned_bp], rec_stats i32[n_rec,4])."""
    import torch
    require_gpu()
    dev = t["contig_seq"].device
    chunk_off, task_rec, task_ev0 = pileup_plan(flat, ev_per_task)
    d_co = torch.from_numpy(chunk_off).to(dev)
    d_tr = torch.from_numpy(task_rec if len(task_rec) else np.zeros(1, np.int32)).to(dev)
    d_te = torch.from_numpy(task_ev0 if len(task_ev0) else np.zeros(1, np.int32)).to(dev)
    scratch = torch.zeros(max(4 * int(chunk_off[-1]), 4), dtype=torch.int32, device=dev)
    pile = torch.zeros(max(flat.aligned_bp, 1), dtype=torch.uint8, device=dev)
    stats = torch.zeros((max(flat.n_rec, 1), 4), dtype=torch.int32, device=dev)
    _check(load().hs_pileup(_p(t["contig_seq"]), _p(t["contig_off"]), _p(t["read_seq"]), _p(t["read_off"]), _p(t["rec_read"]),
                            _p(t["rec_contig"]), _p(t["rec_pos"]), _p(t["rec_strand"]), _p(t["rec_cig_off"]), _p(t["cigar"]),
                            _p(t["pile_off"]), C.c_int32(flat.n_rec), _p(d_co), _p(scratch), _p(d_tr), _p(d_te),
                            C.c_int32(len(task_rec)), C.c_int32(ev_per_task), _p(pile), _p(stats), C.c_void_p(0)))
    torch.cuda.synchronize()
    return pile[:flat.aligned_bp], stats[:flat.n_rec]


def tile_plan(flat: FlatBatch, device="cuda:0"):
    """hs_tile_plan on the host arrays of a FlatBatch, uploaded: dict of device tensors (off int64, ent int64 [n,2] view of the 16-B entries, rec int32)"""
    import torch
    p_off = C.POINTER(C.c_int64)(); p_ent = C.c_void_p(); p_rec = C.POINTER(C.c_int32)(); n_tiles = C.c_int64(0)
    _check(load().hs_tile_plan(_hp(flat.contig_off, C.c_int64), C.c_int32(flat.n_contigs), _hp(flat.contig_rec_off, C.c_int32),
                               _hp(flat.rec_pos, C.c_int32), _hp(flat.rec_qend, C.c_int32), _hp(flat.pile_off, C.c_int64),
                               C.byref(p_off), C.byref(p_ent), C.byref(p_rec), C.byref(n_tiles)))
    nt = int(n_tiles.value)
    off = np.ctypeslib.as_array(p_off, shape=(nt + 1,)).copy()
    ne = int(off[-1])
    ent = np.ctypeslib.as_array(C.cast(p_ent, C.POINTER(C.c_int64)), shape=(max(ne, 1) * 2,)).copy()
    rec = np.ctypeslib.as_array(p_rec, shape=(max(ne, 1),)).copy()
    load().hs_free_host(p_off); load().hs_free_host(p_ent); load().hs_free_host(p_rec)
    return {"off": torch.from_numpy(off).to(device), "ent": torch.from_numpy(ent).to(device), "rec": torch.from_numpy(rec).to(device),
            "h_off": off, "h_ent": ent.view(np.int32).reshape(-1, 4), "h_rec": rec}


def column_stats(t, flat: FlatBatch, pile, min_second: int = 0, max_depth: int = 0, plan=None):
    """K2 in its full-statistics form on the tile plan (hs_column_stats_tiled); returns a numpy structured view: key u8[4], cnt u16[5],
    depth u16 per position (and, when min_second > 0, the compact selection -- second count > min_second, or == min_second with no
    third allele -- as sorted global positions + depths)."""
    import torch
    require_gpu()
    total = int(flat.contig_off[-1])
    dev = pile.device
    if plan is None:
        plan = tile_plan(flat, dev)
    out = torch.zeros((max(total, 1), 16), dtype=torch.uint8, device=dev)
    if min_second > 0:
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        gpos = torch.zeros(max(total, 1), dtype=torch.int64, device=dev)
        dep = torch.zeros(max(total, 1), dtype=torch.int32, device=dev)
        args = (C.c_int32(min_second), _p(cnt), _p(gpos), _p(dep), C.c_int32(total))
    else:
        args = (C.c_int32(0), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0), C.c_int32(0))
    _check(load().hs_column_stats_tiled(_p(pile), _p(plan["off"]), _p(plan["ent"]), C.c_int64(total), _p(out), *args, C.c_int32(max_depth), C.c_void_p(0)))
    torch.cuda.synchronize()
    dt = np.dtype([("key", np.uint8, 4), ("cnt", np.uint16, 5), ("depth", np.uint16)])
    st = out[:total].cpu().numpy().view(dt).reshape(-1)
    if min_second > 0:
        n = int(cnt.item())
        g = gpos[:n].cpu().numpy(); d = dep[:n].cpu().numpy()
        o = np.argsort(g, kind="stable")
        return st, g[o], d[o]
    return st


class _CandBits(C.Structure):
    _fields_ = [("wlo", C.c_int32), ("n_words", C.c_uint16), ("n_slots", C.c_uint16), ("idx_min", C.c_int32), ("idx_max", C.c_int32), ("reach", C.c_int32),
                ("n_entries", C.c_int32), ("word_off", C.c_int64)]


class _CvTaps(C.Structure):
    _fields_ = [("n_contigs", C.c_int32), ("n_cols", C.c_int64), ("n_entries", C.c_int64), ("col_gpos", C.POINTER(C.c_int64)), ("col_rec", C.c_void_p),
                ("col_off", C.POINTER(C.c_int64)), ("col_idx", C.POINTER(C.c_int32)), ("col_code", C.POINTER(C.c_uint8)), ("n_cand", C.c_int64),
                ("cand_rec", C.c_void_p), ("cand_col", C.POINTER(C.c_int32)), ("cand_bits", C.POINTER(_CandBits)), ("cand_words", C.POINTER(C.c_uint64)),
                ("n_cand_words", C.c_int64), ("contig_n_cand", C.POINTER(C.c_int32)), ("contig_mean_distance", C.POINTER(C.c_float))]


COLREC_DTYPE = np.dtype([("pos", np.int32), ("contig", np.int32), ("c0", np.uint16), ("c1", np.uint16), ("k0", np.uint8), ("k1", np.uint8), ("flags", np.uint8), ("c2", np.uint8)])
CANDBITS_DTYPE = np.dtype([("wlo", np.int32), ("n_words", np.uint16), ("n_slots", np.uint16), ("idx_min", np.int32), ("idx_max", np.int32), ("reach", np.int32),
                           ("n_entries", np.int32), ("word_off", np.int64)])


def cv_column_pass_taps(batch: "CvBatch", c0: int, c1: int, automatic_snp_threshold: float = 0.33) -> Dict:
    """The column pass of stage 3 as the pipeline queues it (hs_cv_column_pass_taps), with what its kernels left on the device: the extracted
    columns (positions, records, CSR), the packed candidates, their bit sets, the contigs' candidate counts and mean distances."""
    lib = load()
    lib.hs_cv_column_pass_taps.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.POINTER(C.POINTER(_CvTaps))]
    lib.hs_cv_taps_destroy.argtypes = [C.POINTER(_CvTaps)]
    lib.hs_cv_taps_destroy.restype = None
    tp = C.POINTER(_CvTaps)()
    _check(lib.hs_cv_column_pass_taps(batch.handle, C.c_int32(c0), C.c_int32(c1), C.c_float(automatic_snp_threshold), C.byref(tp)))
    t = tp.contents
    n, e, nc, Cn = int(t.n_cols), int(t.n_entries), int(t.n_cand), int(t.n_contigs)

    def arr(ptr, count, dtype):
        if count == 0:
            return np.zeros(0, dtype)
        return np.frombuffer(C.string_at(ptr, count * np.dtype(dtype).itemsize), dtype=dtype).copy()
    out = {
        "col_gpos": arr(t.col_gpos, n, np.int64), "col_rec": arr(t.col_rec, n, COLREC_DTYPE), "col_off": arr(t.col_off, n + 1, np.int64),
        "col_idx": arr(t.col_idx, e, np.int32), "col_code": arr(t.col_code, e, np.uint8),
        "cand_rec": arr(t.cand_rec, nc, COLREC_DTYPE), "cand_col": arr(t.cand_col, nc, np.int32), "cand_bits": arr(t.cand_bits, nc, CANDBITS_DTYPE),
        "cand_words": arr(t.cand_words, int(t.n_cand_words), np.uint64),
        "contig_n_cand": arr(t.contig_n_cand, Cn, np.int32), "contig_mean_distance": arr(t.contig_mean_distance, Cn, np.float32),
    }
    lib.hs_cv_taps_destroy(tp)
    return out


def exclusive_scan(values):
    """n ints -> n + 1 offsets on the device (the scan behind the graph CSR and the selection list)"""
    import torch
    require_gpu()
    v = _np(values, np.int32)
    d_in = torch.from_numpy(v if v.size else np.zeros(1, np.int32)).to("cuda:0")
    d_out = torch.full((len(v) + 1,), -1, dtype=torch.int64, device="cuda:0")
    _check(load().hs_exclusive_scan_i32(_p(d_in), C.c_int32(len(v)), _p(d_out), C.c_void_p(0)))
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


def column_partition_test(col_off, col_idx, col_code, col_contig, col_k0, col_k1, col_c1, col_is_cand, part_off, part_state_off, part_state, n_reads):
    """K4: loops C/D of keep_only_robust_variants (call_variants.cpp:721-764); returns keep uint8 [n_cols]."""
    import torch
    require_gpu()
    dev = "cuda:0"
    n = len(col_contig)
    def up(a, dt):
        a = _np(a, dt)
        return torch.from_numpy(a if a.size else np.zeros(1, dt)).to(dev)
    d = [up(col_off, np.int64), up(col_idx, np.int32), up(col_code, np.uint8), up(col_contig, np.int32), up(col_k0, np.uint8), up(col_k1, np.uint8),
         up(col_c1, np.int32), up(col_is_cand, np.uint8)]
    q = [up(part_off, np.int32), up(part_state_off, np.int64), up(part_state, np.int8)]
    keep = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)
    nr = _np(n_reads, np.int32)
    _check(load().hs_column_partition_test(*[_p(x) for x in d], C.c_int32(n), *[_p(x) for x in q], _hp(nr, C.c_int32), C.c_int32(len(nr)), _p(keep), C.c_void_p(0)))
    torch.cuda.synchronize()
    return keep[:n].cpu().numpy()


def column_partition_last_counts():
    """what the kernels of this thread's last column_partition_test passed on: columns to k_column_partition_grouped, whole columns to
    k_column_partition_test, (column, partition) pairs to k_column_partition_pairs"""
    out = (C.c_int32 * 3)()
    load().hs_column_partition_last_counts(out)
    return {"to_grouped": int(out[0]), "whole_columns_to_exact": int(out[1]), "pairs_to_exact": int(out[2])}


def partition_pair_distance(state, more, less, part_off, part_n, pair_a, pair_b, threshold_p=2):
    """V5 for a list of partition pairs (dense arrays, state 2 = absent): [n_pairs, 8] = n00, n01, n10, n11, phased, augmented, valid, comparable"""
    import torch
    require_gpu()
    dev = "cuda:0"
    n = len(pair_a)
    up = lambda a, dt: torch.from_numpy(_np(a, dt) if len(a) else np.zeros(1, dt)).to(dev)
    sigma3 = np.array([np.float32(0.5 * k + 3 * np.sqrt(k * 0.5 * (1 - 0.5))) for k in range(4096)], np.float32)
    d = [up(state, np.int8), up(more, np.int32), up(less, np.int32), up(part_off, np.int64), up(part_n, np.int32), up(pair_a, np.int32), up(pair_b, np.int32)]
    sg = up(sigma3, np.float32)
    out = torch.zeros((max(n, 1), 8), dtype=torch.int32, device=dev)
    _check(load().hs_partition_pair_distance(*[_p(x) for x in d], C.c_int32(n), C.c_int32(threshold_p), _p(sg), _p(out), C.c_void_p(0)))
    torch.cuda.synchronize()
    return out[:n].cpu().numpy()


def snp_planes(n_reads, snp_ref, snp_alt, col_off, col_idx, col_code):
    """K5a for one contig: returns (alt, ref) uint64 [N, words] bit-planes"""
    import torch
    require_gpu()
    dev = "cuda:0"
    S = len(snp_ref)
    W = (S + 63) // 64
    up = lambda a, dt: torch.from_numpy(_np(a, dt) if len(a) else np.zeros(1, dt)).to(dev)
    d = [up(col_off, np.int64), up(col_idx, np.int32), up(col_code, np.uint8), up(snp_ref, np.uint8), up(snp_alt, np.uint8),
         up(np.zeros(S, np.int32), np.int32), up(np.zeros(1, np.int64), np.int64), up(np.zeros(1, np.int64), np.int64), up(np.array([W], np.int32), np.int32)]
    # (not zeroed on purpose: the kernel writes every word of the rows)
    alt = torch.full((n_reads, max(W, 1)), -1, dtype=torch.int64, device=dev); ref = torch.full_like(alt, -1)
    hw = np.array([W], np.int32)
    _check(load().hs_snp_planes(*[_p(x) for x in d], _p(up(np.array([n_reads], np.int32), np.int32)), _hp(hw, C.c_int32), C.c_int32(1), C.c_int32(S), _p(alt), _p(ref), C.c_void_p(0)))
    torch.cuda.synchronize()
    return alt.cpu().numpy().view(np.uint64)[:, :W], ref.cpu().numpy().view(np.uint64)[:, :W]


def simdiff(alt_planes: np.ndarray, ref_planes: np.ndarray):
    """K5 for one contig: planes are uint64 [N, words]; returns (sim, diff) int32 [N, N]."""
    import torch
    require_gpu()
    N, W = alt_planes.shape
    dev = "cuda:0"
    a = torch.from_numpy(alt_planes.view(np.int64)).to(dev); r = torch.from_numpy(ref_planes.view(np.int64)).to(dev)
    po = torch.zeros(1, dtype=torch.int64, device=dev); oo = torch.zeros(1, dtype=torch.int64, device=dev)
    n = torch.tensor([N], dtype=torch.int32, device=dev); w = torch.tensor([W], dtype=torch.int32, device=dev)
    sim = torch.zeros((N, N), dtype=torch.int32, device=dev); diff = torch.zeros((N, N), dtype=torch.int32, device=dev)
    _check(load().hs_simdiff(_p(a), _p(r), _p(po), _p(n), _p(w), _p(oo), C.c_int32(1), _p(sim), _p(diff), C.c_void_p(0)))
    torch.cuda.synchronize()
    return sim.cpu().numpy(), diff.cpu().numpy()


def read_graphs(sim_list, diff_list, windows, error_rate):
    """K6: sim_list/diff_list = per-contig int32 [N, N] matrices (uploaded here); windows = [(contig, ascending masked read ids)].
    Returns (per window: dict read id -> sorted neighbour list, rows resolved on the host)."""
    import torch
    require_gpu()
    dev = "cuda:0"
    n = np.array([m.shape[0] for m in sim_list], np.int32)
    off = np.zeros(len(n), np.int64)
    if len(n) > 1:
        off[1:] = np.cumsum(n[:-1].astype(np.int64) ** 2)
    d_sim = torch.from_numpy(np.concatenate([np.ascontiguousarray(m, np.int32).ravel() for m in sim_list])).to(dev)
    d_diff = torch.from_numpy(np.concatenate([np.ascontiguousarray(m, np.int32).ravel() for m in diff_list])).to(dev)
    wc = np.array([w[0] for w in windows], np.int32)
    moff = np.zeros(len(windows) + 1, np.int64); moff[1:] = np.cumsum([len(w[1]) for w in windows])
    ids = _np(np.concatenate([np.asarray(w[1], np.int32) for w in windows]) if moff[-1] else np.zeros(1, np.int32), np.int32)
    p_off = C.POINTER(C.c_int64)(); p_nbr = C.POINTER(C.c_int32)(); n_host = C.c_int64(0)
    _check(load().hs_read_graphs(_p(d_sim), _p(d_diff), off.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p), C.c_int32(len(n)),
                                 wc.ctypes.data_as(C.c_void_p), moff.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), C.c_int32(len(windows)),
                                 C.c_float(error_rate), C.byref(p_off), C.byref(p_nbr), C.byref(n_host), C.c_void_p(0)))
    rows = int(moff[-1])
    noff = np.ctypeslib.as_array(p_off, shape=(rows + 1,)).copy()
    nbr = np.ctypeslib.as_array(p_nbr, shape=(max(int(noff[-1]), 1),)).copy()
    load().hs_free_host(p_off); load().hs_free_host(p_nbr)
    out = []
    for w in range(len(windows)):
        out.append({int(ids[r]): nbr[noff[r]:noff[r + 1]].tolist() for r in range(int(moff[w]), int(moff[w + 1]))})
    return out, int(n_host.value)


def edit_distance(queries: Sequence[np.ndarray], targets: Sequence[np.ndarray], mode: str = "NW"):
    """A1 Myers bit-vector kernel; codes 0..3; mode NW | SHW | HW. Returns (distance, end) int32 arrays."""
    import torch
    require_gpu()
    dev = "cuda:0"
    m = {"NW": 0, "SHW": 1, "HW": 2}[mode]
    qo = np.zeros(len(queries) + 1, np.int64); qo[1:] = np.cumsum([len(q) for q in queries])
    to = np.zeros(len(targets) + 1, np.int64); to[1:] = np.cumsum([len(x) for x in targets])
    q = _np(np.concatenate(queries) if qo[-1] else np.zeros(1), np.uint8)
    tt = _np(np.concatenate(targets) if to[-1] else np.zeros(1), np.uint8)
    T = lambda a: torch.from_numpy(a).to(dev)
    d_q, d_qo, d_t, d_to = T(q), T(qo), T(tt), T(to)
    n = len(queries)
    dist = torch.zeros(n, dtype=torch.int32, device=dev); end = torch.zeros(n, dtype=torch.int32, device=dev)
    _check(load().hs_edit_distance(_p(d_q), _p(d_qo), _p(d_t), _p(d_to), C.c_int32(n), C.c_int32(m), _p(dist), _p(end), C.c_void_p(0)))
    torch.cuda.synchronize()
    return dist.cpu().numpy(), end.cpu().numpy()


# ---- next stage: the .gro consumer (host code; needs no device) -----------------------------------------
def gaf_from_files(gfa: str, reads: str, sam: str, gro: str, out_gaf: str, amplicon: bool = False, n_threads: int = 1) -> None:
    """parse_split_file + merge_intervals + output_GAF of the reference's stage 5 (create_new_contigs.cpp:1582-1590)."""
    _check(load().hs_gaf_from_files(gfa.encode(), reads.encode(), sam.encode(), gro.encode(), C.c_int32(1 if amplicon else 0),
                                    out_gaf.encode(), C.c_int32(n_threads)))


def gaf_from_labels(gfa: str, reads: str, sam: str, sr: Dict, out_gaf: str, contig_has_snps=None, amplicon: bool = False,
                    n_threads: int = 1) -> None:
    """Same, from a stage-4 result in memory (`sr` as returned by separate_reads / run_pipeline: win_off, win_start, win_end,
    label_off, labels) instead of the .gro text; contig_has_snps[c] = False for contigs the .gro writer would skip (default:
    the contigs without windows)."""
    win_off = _np(sr["win_off"], np.int64); win_start = _np(sr["win_start"], np.int32); win_end = _np(sr["win_end"], np.int32)
    label_off = _np(sr["label_off"], np.int64); labels = _np(sr["labels"], np.int32)
    has = None if contig_has_snps is None else _np(np.asarray(contig_has_snps).astype(np.uint8), np.uint8)

    def ptr(a):
        return a.ctypes.data_as(C.c_void_p)
    _check(load().hs_gaf_from_labels(gfa.encode(), reads.encode(), sam.encode(), C.c_int32(1 if amplicon else 0), C.c_int32(len(win_off) - 1),
                                     ptr(win_off), ptr(win_start), ptr(win_end), ptr(label_off), ptr(labels), None if has is None else ptr(has),
                                     out_gaf.encode(), C.c_int32(n_threads)))


def edlib_hw_align(pairs, path=True):
    """A1 as stage 5 uses edlib (HW, k = -1, TASK_PATH): list of (query, target) strings over ACGT -> list of dicts
    {distance, start, end, ops (numpy uint8 of edlib move codes) or None}"""
    import torch
    require_gpu()
    dev = "cuda:0"
    code = np.full(256, 3, np.uint8)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    enc = lambda x: code[np.frombuffer(x.encode(), dtype=np.uint8)] if x else np.zeros(0, np.uint8)
    qs = [enc(q) for q, _ in pairs]; ts = [enc(t) for _, t in pairs]
    n = len(pairs)
    qo = np.zeros(n + 1, np.int64); to = np.zeros(n + 1, np.int64); oo = np.zeros(n + 1, np.int64)
    np.cumsum([len(x) for x in qs], out=qo[1:]); np.cumsum([len(x) for x in ts], out=to[1:]); np.cumsum([len(a) + len(b) for a, b in zip(qs, ts)], out=oo[1:])
    cat = lambda xs: np.concatenate(xs) if xs and sum(len(x) for x in xs) else np.zeros(1, np.uint8)
    dq = torch.from_numpy(cat(qs)).to(dev); dt = torch.from_numpy(cat(ts)).to(dev)
    dd = torch.zeros(max(n, 1), dtype=torch.int32, device=dev); ds = torch.zeros_like(dd); de = torch.zeros_like(dd); dl = torch.zeros_like(dd)
    dops = torch.zeros(max(int(oo[-1]), 1), dtype=torch.uint8, device=dev)
    _check(load().hs_edlib_hw_align(_p(dq), _hp(qo, C.c_int64), _p(dt), _hp(to, C.c_int64), C.c_int32(n), _p(dd), _p(ds), _p(de),
                                    _p(dops) if path else C.c_void_p(0), _hp(oo, C.c_int64), _p(dl), C.c_void_p(0)))
    torch.cuda.synchronize()
    dd, ds, de, dl, ops = dd.cpu().numpy(), ds.cpu().numpy(), de.cpu().numpy(), dl.cpu().numpy(), dops.cpu().numpy()
    out = []
    for i in range(n):
        out.append({"distance": int(dd[i]), "start": int(ds[i]), "end": int(de[i]),
                    "ops": ops[oo[i]:oo[i] + dl[i]].copy() if path and dl[i] >= 0 else None})
    return out


def _string_batch(fn, lists, ints=()):
    lib = load()
    require_gpu()
    n = len(lists[0])
    arrs = [(C.c_char_p * max(n, 1))(*[x.encode() for x in l]) for l in lists]
    iarrs = [np.ascontiguousarray(v, np.int32) for v in ints]
    out = C.POINTER(C.c_char_p)()
    fn.argtypes = [C.POINTER(C.c_char_p)] * len(arrs) + [C.POINTER(C.c_int32)] * len(iarrs) + [C.c_int32, C.POINTER(C.POINTER(C.c_char_p))]
    _check(fn(*arrs, *[_hp(v, C.c_int32) for v in iarrs], C.c_int32(n), C.byref(out)))
    res = [out[i].decode() for i in range(n)]
    lib.hs_free_strings.argtypes = [C.POINTER(C.c_char_p), C.c_int32]
    lib.hs_free_strings.restype = None
    lib.hs_free_strings(out, C.c_int32(n))
    return res


def reattach_ends(backbones, consensuses):
    """tools.cpp:505-536, batched"""
    return _string_batch(load().hs_reattach_ends, [backbones, consensuses])


def trim_polished(to_polish, newcontigs, overhang_left, overhang_right):
    """create_new_contigs.cpp:556-629, batched"""
    return _string_batch(load().hs_trim_polished, [to_polish, newcontigs], [overhang_left, overhang_right])
