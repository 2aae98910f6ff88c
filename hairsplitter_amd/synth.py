"""Build-owned synthetic workload generator.

Produces what stage 3 of the pipeline is handed (`hairsplitter.py:623-653` of the reference):
a GFA of contigs, a single-line FASTA of reads and the *truth* SAM (SEQ/QUAL replaced by `*`,
trailing `LN:i:<read length>` tag, exactly as the awk step at `hairsplitter.py:629-630` leaves it).

Model: contig = haplotype 0 (iid uniform ACGT); other haplotypes = haplotype 0 + iid substitutions at
rate `div`; reads are sampled uniformly over the contig from a uniformly chosen haplotype, 50/50 strand,
with iid errors split 1:1:1 substitution / insertion / deletion. ONT: lognormal(median 8 kb, sigma .6)
clipped to [1 kb, 60 kb], 5 % errors. HiFi: N(15 kb, 2 kb) clipped to [5 kb, 25 kb], 0.2 % errors.

Everything is driven by numpy's PCG64 seeded with (seed, contig index) so a contig's data does not depend
on which other contigs are generated with it (needed for per-contig sharding over ranks).
"""
from __future__ import annotations

import dataclasses
import io
import os
from typing import List, Optional, Sequence

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.array([3, 2, 1, 0], dtype=np.uint8)

# BAM-style CIGAR op codes used across the C-ABI (include/hairsplitter_hip.h)
OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X = range(9)
_OPCHAR = "MIDNSHP=X"


@dataclasses.dataclass
class Alignment:
    read: int          # index into ContigData.reads
    pos: int           # 0-based leftmost reference position
    strand: bool       # True = forward
    cigar: np.ndarray  # uint32 BAM-style ops (len << 4 | op)
    nm: int


@dataclasses.dataclass
class ContigData:
    name: str
    seq: np.ndarray                 # uint8 codes 0..3
    reads: List[np.ndarray]         # as sequenced (i.e. reverse-complemented when strand is False)
    read_names: List[str]
    alns: List[Alignment]
    haplotype_of_read: np.ndarray   # truth labels (not used by the path; kept for sanity checks)

    @property
    def aligned_bp(self) -> int:
        tot = 0
        for a in self.alns:
            ops = a.cigar & 0xF
            lens = a.cigar >> 4
            tot += int(lens[(ops == OP_M) | (ops == OP_D) | (ops == OP_EQ) | (ops == OP_X)].sum())
        return tot


def _read_lengths(rng, n, tech):
    if tech == "ont":
        l = np.exp(rng.normal(np.log(8000.0), 0.6, size=n))
        return np.clip(l, 1000, 60000).astype(np.int64)
    if tech == "hifi":
        l = rng.normal(15000.0, 2000.0, size=n)
        return np.clip(l, 5000, 25000).astype(np.int64)
    raise ValueError(tech)


def _simulate_read(rng, hap_seq: np.ndarray, start: int, span: int, err: float, eqx: bool = False):
    """Returns (aligned-orientation read codes, cigar uint32 array, NM)."""
    ref = hap_seq[start:start + span]
    u = rng.random(span)
    sub = u < err / 3.0
    dele = (u >= err / 3.0) & (u < 2.0 * err / 3.0)
    ins = rng.random(span) < err / 3.0
    # alignment must start and end with a match column
    sub[0] = sub[-1] = False
    dele[0] = dele[-1] = False
    ins[-1] = False
    bases = ref.copy()
    nsub = int(sub.sum())
    if nsub:
        bases[sub] = (bases[sub] + rng.integers(1, 4, size=nsub).astype(np.uint8)) & 3
    nins = int(ins.sum())
    ev_index = np.arange(span) + np.cumsum(ins) - ins  # event slot of each reference position
    nev = span + nins
    ev_op = np.empty(nev, dtype=np.uint8)
    ev_base = np.zeros(nev, dtype=np.uint8)
    ev_op[ev_index] = np.where(dele, OP_D, (np.where(sub, OP_X, OP_EQ) if eqx else OP_M))
    ev_base[ev_index] = bases
    if nins:
        ins_slots = ev_index[ins] + 1
        ev_op[ins_slots] = OP_I
        ev_base[ins_slots] = rng.integers(0, 4, size=nins).astype(np.uint8)
    read = ev_base[ev_op != OP_D]
    # run-length encode
    change = np.flatnonzero(np.diff(ev_op)) + 1
    starts = np.concatenate(([0], change))
    lens = np.diff(np.concatenate((starts, [nev])))
    cigar = (lens.astype(np.uint32) << 4) | ev_op[starts].astype(np.uint32)
    nm = nsub + nins + int(dele.sum())
    return read, cigar, nm


def make_contig(seed: int, index: int, length: int, n_hap: int, div: float, depth: float,
                tech: str = "ont", name: Optional[str] = None, err: Optional[float] = None,
                hap_weights: Optional[Sequence[float]] = None,
                read_len_override: Optional[Sequence[int]] = None,
                clip_prob: float = 0.0, eqx: bool = False, overhang_prob: float = 0.0,
                inert_ops_prob: float = 0.0, haplotypes: Optional[Sequence[np.ndarray]] = None,
                contig_seq: Optional[np.ndarray] = None) -> ContigData:
    """haplotypes / contig_seq: see below. eqx: write matches/mismatches as '=' / 'X' instead of 'M'. overhang_prob: with this probability a read near the
    contig end is given a CIGAR that runs past the end of the contig (the reference stops at `indexQuery < L`,
    call_variants.cpp:217)."""
    rng = np.random.default_rng([seed, index])
    if err is None:
        err = 0.05 if tech == "ont" else 0.002
    if haplotypes is not None:
        # given haplotypes (same coordinates as the contig, substitutions only) and a given contig sequence, e.g. a consensus
        # assembly of them: reads are sampled from the haplotypes, the truth alignment is expressed on the contig
        haps = [np.ascontiguousarray(h, dtype=np.uint8) for h in haplotypes]
        hap0 = np.ascontiguousarray(contig_seq if contig_seq is not None else haps[0], dtype=np.uint8)
        n_hap, length = len(haps), len(hap0)
        assert all(len(h) == length for h in haps)
    else:
        hap0 = rng.integers(0, 4, size=length).astype(np.uint8)
        haps = [hap0]
        for _ in range(1, n_hap):
            h = hap0.copy()
            m = rng.random(length) < div
            k = int(m.sum())
            if k:
                h[m] = (h[m] + rng.integers(1, 4, size=k).astype(np.uint8)) & 3
            haps.append(h)
    reads, names, alns, truth = [], [], [], []
    target_bp = depth * length
    tot = 0
    while tot < target_bp:
        rl = int(_read_lengths(rng, 1, tech)[0])
        if read_len_override is not None:
            rl = int(rng.integers(read_len_override[0], read_len_override[1] + 1))
        span = min(rl, length)
        start = int(rng.integers(0, length - span + 1))
        if hap_weights is None:
            h = int(rng.integers(0, n_hap))
        else:
            h = int(rng.choice(n_hap, p=np.asarray(hap_weights) / np.sum(hap_weights)))
        strand = bool(rng.integers(0, 2))
        if span < 3:
            continue
        over = 0
        if overhang_prob > 0.0 and rng.random() < overhang_prob:
            over = int(rng.integers(1, 300))
            start = max(0, length - span)          # touches the contig end; the alignment is then extended past it
        src = haps[h] if not over else np.concatenate((haps[h], rng.integers(0, 4, size=over).astype(np.uint8)))
        read, cigar, nm = _simulate_read(rng, src, start, span + over, err, eqx=eqx)
        if inert_ops_prob > 0.0 and rng.random() < inert_ops_prob and len(cigar) > 4:
            # 'N' and 'P' runs: the reference expands them but no branch of its CIGAR walk handles them
            # (call_variants.cpp:226-342), so they consume nothing -- not even reference bases
            k = int(rng.integers(1, len(cigar) - 1))
            extra = np.array([(int(rng.integers(1, 50)) << 4) | (OP_N if rng.random() < 0.5 else OP_P)], dtype=np.uint32)
            cigar = np.concatenate((cigar[:k], extra, cigar[k:]))
        if clip_prob > 0.0:
            # soft/hard clips: the FASTA read carries the clipped bases either way (the reference reloads reads
            # from the reads file, input_output.cpp:546-569, and steps over S and H alike, call_variants.cpp:269-273)
            for side in (0, 1):
                if rng.random() < clip_prob:
                    k = int(rng.integers(1, 200))
                    op = OP_S if rng.random() < 0.7 else OP_H
                    junk = rng.integers(0, 4, size=k).astype(np.uint8)
                    tok = np.array([(k << 4) | op], dtype=np.uint32)
                    if side == 0:
                        read = np.concatenate((junk, read)); cigar = np.concatenate((tok, cigar))
                    else:
                        read = np.concatenate((read, junk)); cigar = np.concatenate((cigar, tok))
        if not strand:
            read_out = _COMP[read[::-1]]
        else:
            read_out = read
        ridx = len(reads)
        reads.append(np.ascontiguousarray(read_out))
        names.append(f"{name or ('ctg%d' % index)}_r{ridx}")
        alns.append(Alignment(ridx, start, strand, cigar, nm))
        truth.append(h)
        tot += span
    return ContigData(name or f"ctg{index}", hap0, reads, names, alns, np.asarray(truth, dtype=np.int32))


def cigar_string(cigar: np.ndarray) -> str:
    return "".join(f"{int(c) >> 4}{_OPCHAR[int(c) & 0xF]}" for c in cigar)


def write_files(contigs: Sequence[ContigData], outdir: str, prefix: str = "",
                sam_extra: Optional[List[str]] = None, gfa_extra: Optional[List[str]] = None, fastq: bool = False,
                sam_header: bool = True) -> dict:
    """Writes assembly.gfa / reads.fasta / aln.sam. Returns the paths. `sam_extra` / `gfa_extra`: verbatim lines appended
    to the SAM / the GFA (e.g. supplementary records, 'L' lines)."""
    os.makedirs(outdir, exist_ok=True)
    gfa = os.path.join(outdir, prefix + "assembly.gfa")
    fa = os.path.join(outdir, prefix + ("reads.fastq" if fastq else "reads.fasta"))   # anything but .fasta / .fa is read as FASTQ (input_output.cpp:41-44)
    sam = os.path.join(outdir, prefix + "aln.sam")
    with open(gfa, "w") as g:
        for c in contigs:
            g.write(f"S\t{c.name}\t{_ACGT[c.seq].tobytes().decode()}\n")
        for line in (gfa_extra or []):
            g.write(line.rstrip("\n") + "\n")
    with open(fa, "w") as f:
        for c in contigs:
            for nm, r in zip(c.read_names, c.reads):
                if fastq:
                    # quality strings that begin with '@' or '+' exercise the record test of input_output.cpp:64
                    q = ("@" if len(r) % 3 == 0 else "+" if len(r) % 3 == 1 else "I") + "I" * (len(r) - 1)
                    f.write(f"@{nm} some description\n{_ACGT[r].tobytes().decode()}\n+\n{q}\n")
                else:
                    f.write(f">{nm}\n{_ACGT[r].tobytes().decode()}\n")
    with open(sam, "w") as s:
        if sam_header:
            s.write("@HD\tVN:1.6\tSO:unsorted\n")
            for c in contigs:
                s.write(f"@SQ\tSN:{c.name}\tLN:{len(c.seq)}\n")
        for c in contigs:
            for a in c.alns:
                flag = 0 if a.strand else 16
                s.write(f"{c.read_names[a.read]}\t{flag}\t{c.name}\t{a.pos + 1}\t60\t{cigar_string(a.cigar)}"
                        f"\t*\t0\t0\t*\t*\tNM:i:{a.nm}\tLN:i:{len(c.reads[a.read])}\n")
        for line in (sam_extra or []):
            s.write(line.rstrip("\n") + "\n")
    return {"gfa": gfa, "reads": fa, "sam": sam}


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs. `scale` < 1 shrinks lengths for CPU-sized parity cases.
# ---------------------------------------------------------------------------------------------

def config_contigs(cfg: str, seed: Optional[int] = None, first: int = 0, count: Optional[int] = None,
                   scale: float = 1.0) -> List[ContigData]:
    cfg = cfg.upper()
    if cfg == "C2":
        seed = 2 if seed is None else seed
        n = 1 if count is None else count
        return [make_contig(seed, first + i, int(100_000 * scale), 2, 0.01, 50, "ont") for i in range(n)]
    if cfg == "C3":
        seed = 3 if seed is None else seed
        n = 50 if count is None else count
        return [make_contig(seed, first + i, int(200_000 * scale), 4, 0.01, 40, "ont") for i in range(n)]
    if cfg == "C4":
        seed = 4 if seed is None else seed
        n = 500 if count is None else count
        out = []
        for i in range(first, first + n):
            r = np.random.default_rng([seed, 1_000_000 + i])
            L = int(np.clip(np.exp(r.normal(np.log(100_000.0), 0.5)), 20_000, 300_000) * scale)
            ploidy = int(r.integers(1, 9))
            out.append(make_contig(seed, i, L, ploidy, 0.01, 30, "ont"))
        return out
    if cfg == "C5":
        seed = 5 if seed is None else seed
        n = 34 if count is None else count  # 10 Mb cut in <=300 kb chunks (hairsplitter.py:583)
        out = []
        for i in range(first, first + n):
            L = 300_000 if i < 33 else 100_000
            out.append(make_contig(seed, i, int(L * scale), 2, 0.001, 30, "hifi"))
        return out
    if cfg == "C5U":   # the uncut stress variant of C5: one 10 Mb contig, about 20 000 HiFi reads
        seed = 5 if seed is None else seed
        return [make_contig(seed, 100 + first, int(10_000_000 * scale), 2, 0.001, 30, "hifi", name="c5uncut")]
    raise ValueError(f"unknown config {cfg}")


def _config_chunk(a):
    cfg, seed, first, count, scale = a
    return config_contigs(cfg, seed=seed, first=first, count=count, scale=scale)


def config_contigs_parallel(cfg: str, seed: Optional[int] = None, count: Optional[int] = None, scale: float = 1.0,
                            workers: Optional[int] = None) -> List[ContigData]:
    """config_contigs on forked workers (a contig's data depends only on (seed, index), so the result is the same list).
    Only call it from a process that has not initialised the GPU."""
    n = {"C2": 1, "C3": 50, "C4": 500, "C5": 34, "C5U": 1}[cfg.upper()] if count is None else count
    workers = max(1, min(workers or (os.cpu_count() or 1), n // 4))
    if workers <= 1:
        return config_contigs(cfg, seed=seed, count=count, scale=scale)
    import multiprocessing as mp
    step = max(1, n // (4 * workers))
    jobs = [(cfg, seed, i, min(step, n - i), scale) for i in range(0, n, step)]
    with mp.get_context("fork").Pool(workers) as pool:
        parts = pool.map(_config_chunk, jobs)
    return [c for part in parts for c in part]


def config_shapes(cfg: str, seed: Optional[int] = None, count: Optional[int] = None):
    """(length, ploidy, depth) of every contig of a configuration WITHOUT generating it: what a scheduler knows up front
    (contig lengths come with the assembly; aligned bp ~ depth x length). Used to shard a job over ranks."""
    cfg = cfg.upper()
    if cfg == "C2":
        return [(100_000, 2, 50)] * (1 if count is None else count)
    if cfg == "C3":
        return [(200_000, 4, 40)] * (50 if count is None else count)
    if cfg == "C4":
        seed = 4 if seed is None else seed
        out = []
        for i in range(500 if count is None else count):
            r = np.random.default_rng([seed, 1_000_000 + i])
            L = int(np.clip(np.exp(r.normal(np.log(100_000.0), 0.5)), 20_000, 300_000))
            out.append((L, int(r.integers(1, 9)), 30))
        return out
    if cfg == "C5":
        n = 34 if count is None else count
        return [(300_000 if i < 33 else 100_000, 2, 30) for i in range(n)]
    if cfg == "C5U":
        return [(10_000_000, 2, 30)]
    raise ValueError(f"unknown config {cfg}")


def _gen_ids_chunk(a):
    cfg, seed, ids, outdir, part = a
    cs = [config_contigs(cfg, seed=seed, first=i, count=1)[0] for i in ids]
    if outdir is not None:
        write_files(cs, outdir, prefix=f"part{part:05d}_", sam_header=False)
    return cs


def generate_job(cfg: str, ids: Sequence[int], seed: Optional[int] = None, workers: int = 1, outdir: Optional[str] = None):
    """The contigs `ids` of a configuration on forked workers (a contig depends only on (seed, id)); with `outdir` the three
    input files of the job are written too (every worker writes its part, the parts are concatenated). Only call it from a
    process that has not initialised the GPU. Returns (contigs, paths or None)."""
    ids = list(ids)
    workers = max(1, min(workers, (len(ids) + 3) // 4))
    step = max(1, (len(ids) + 4 * workers - 1) // (4 * workers))
    jobs = [(cfg, seed, ids[i:i + step], outdir, k) for k, i in enumerate(range(0, len(ids), step))]
    if workers <= 1:
        parts = [_gen_ids_chunk(j) for j in jobs]
    else:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            parts = pool.map(_gen_ids_chunk, jobs)
    contigs = [c for p in parts for c in p]
    files = None
    if outdir is not None:
        import shutil
        files = {"gfa": os.path.join(outdir, "assembly.gfa"), "reads": os.path.join(outdir, "reads.fasta"), "sam": os.path.join(outdir, "aln.sam")}
        with open(files["sam"], "wb") as s:
            s.write(b"@HD\tVN:1.6\tSO:unsorted\n")
            for c in contigs:
                s.write(f"@SQ\tSN:{c.name}\tLN:{len(c.seq)}\n".encode())
            for k in range(len(jobs)):
                with open(os.path.join(outdir, f"part{k:05d}_aln.sam"), "rb") as f:
                    shutil.copyfileobj(f, s, 16 << 20)
                os.remove(os.path.join(outdir, f"part{k:05d}_aln.sam"))
        for key, name in (("gfa", "assembly.gfa"), ("reads", "reads.fasta")):
            with open(files[key], "wb") as o:
                for k in range(len(jobs)):
                    with open(os.path.join(outdir, f"part{k:05d}_{name}"), "rb") as f:
                        shutil.copyfileobj(f, o, 16 << 20)
                    os.remove(os.path.join(outdir, f"part{k:05d}_{name}"))
    return contigs, files
