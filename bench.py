#!/usr/bin/env python3
"""bench.py -- aligned read-bp/s through call_variants + separate_reads on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path (stage 3 + stage 4, device kernels and host glue) over the job's synthetic
contigs, inputs already resident in HBM. Default job = BASELINE.json configs[3] (C4): the 500-contig synthetic metagenome
(lengths lognormal around 100 kb, ploidy 1-8, 30x ONT), the configuration the headline metric and the >= 20x target are
quoted on; it fits one GPU (1.7 G aligned bp). `--config C2|C3|C5` select the other configurations (C2: `--contigs` copies
of the 100 kb diploid contig). With N ranks the SAME job is sharded by contig (longest-processing-time on the contig
lengths), one process per GPU: strong scaling. The only collectives are the per-contig error-rate exchange and ONE gather
of the partition labels to rank 0 per step (RCCL).

Launched by the driver as
    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import shutil
import statistics
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
SETUP_STEPS = 8         # untimed passes before the W warm-up steps: they size the block pools of the library and the runtime

WORKLOADS = {
    "C2": "C2 (BASELINE.json configs[1]): 100 kb contig, 2 haplotypes @1% divergence, 50x ONT-error reads; {n} such contigs",
    "C3": "C3 (BASELINE.json configs[2]): {n} contigs x 200 kb, tetraploid (4 haplotypes @1%), 40x ONT-error reads",
    "C4": "C4 (BASELINE.json configs[3]): {n}-contig synthetic metagenome, lengths lognormal(median 100 kb) in [20 kb, 300 kb], ploidy 1-8 @1%, 30x ONT-error reads",
    "C5": "C5 (BASELINE.json configs[4], pipeline-faithful form): 10 Mb diploid @0.1%, 30x HiFi, cut in {n} chunks of <= 300 kb",
}


def effective_cores() -> int:
    """CPUs this process can actually use: min(online, affinity mask, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def py_error_rate(er32: float) -> float:
    """What hairsplitter.py hands to stage 4: the float printed by stage 3 (6 significant digits), capped at 0.15
    (hairsplitter.py:686-692,725)."""
    e = float("%g" % er32)
    return min(e, 0.15)


def throttle_stats():
    """(nr_throttled, throttled_usec) of this process's cgroup (CFS quota), or None"""
    for p in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            d = dict(l.split() for l in open(p).read().strip().splitlines())
            return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", d.get("throttled_time", 0)))
        except Exception:
            pass
    return None


def run_stage_pair(cv, sr, files, td, tag, threads, env=None):
    """The two executables as hairsplitter.py:668-669,725-726 runs them; returns (seconds stage 3, seconds stage 4)"""
    col, vcf, err, gro = (os.path.join(td, f"{tag}.{x}") for x in ("col", "vcf", "err", "gro"))
    t0 = time.perf_counter()
    subprocess.run(cv + [files["gfa"], files["reads"], files["sam"], str(threads), td, err, "0", "0", col, vcf, "0.33"], check=True,
                   stdout=subprocess.DEVNULL, env=env, timeout=900)
    t1 = time.perf_counter()
    e = py_error_rate(float(open(err).read().strip()))
    subprocess.run(sr + [col, str(threads), str(e), os.path.join(td, "no_ploidy"), "0", "0.01", "0", gro, "0"], check=True,
                   stdout=subprocess.DEVNULL, env=env, timeout=900)
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1


def run_stage4_on(sr, col, err_file, td, tag, threads):
    """stage 4 alone on an existing .col (the second half of run_stage_pair); returns (seconds, .gro path)"""
    gro = os.path.join(td, f"{tag}.gro")
    e = py_error_rate(float(open(err_file).read().strip()))
    t0 = time.perf_counter()
    subprocess.run(sr + [col, str(threads), str(e), os.path.join(td, "no_ploidy"), "0", "0.01", "0", gro, "0"], check=True, stdout=subprocess.DEVNULL, timeout=900)
    return time.perf_counter() - t0, gro


def file_to_file(cfg, n_job, sample_ids, sample_files, job_files, reps, reference_on_full_job, parity_fn=None):
    """SURVEY.md 8(d) metric (ii): wall clock of the two drop-in executables next to the compiled reference (oracle/_ref,
    built from /root/reference by oracle/Makefile) on the SAME files. The drop-ins are timed both ways: as a caller sees them
    (the started process exits when the outputs are complete, the worker's teardown goes on in the background: hs_dropin_main.h)
    and with HS_NO_DETACH=1 (one process, timed to its full exit) -- speed-ups are quoted from the second, conservative number.
    The reference runs ONCE on the files of the whole job (C4: about 40 s with 16 threads) -- that run is the `cpu_baseline` of
    the bench line; without the job's files (or with --no-f2f-reference-full) it runs on a bounded sample instead.
    PARITY GATE: `parity_fn(col, gro, err, against, subset)` is called while the reference's output files exist -- the .col / error rate
    of the reference's HS_call_variants and the .gro of its HS_separate_reads built with std::random_device pinned (oracle/_ref/
    HS_separate_reads_seeded: the stock binary re-seeds from the hardware at every Chinese-Whispers sweep, its labels are not reproducible,
    cluster_graph.cpp:175-177) on the SAME files -- and compares them with what the timed steps of this run returned."""
    import __graft_entry__ as ge
    p = ge.paths()
    cores = effective_cores()
    have_ref = os.path.exists(p["ref_cv"]) and os.path.exists(p["ref_sr"])
    have_seeded = have_ref and os.path.exists(p["ref_sr_seeded"])
    parity = [None]
    if not have_ref and not os.path.exists(p["oracle"]):
        return None, None
    out = {"threads": cores, "runs": reps}
    med = lambda v: statistics.median(v)
    no_detach = dict(os.environ, HS_NO_DETACH="1")

    def leg(files, n_contigs, with_reference, tag, subset):
        bp = int(files["aligned_bp"])
        with tempfile.TemporaryDirectory() as td:
            det = [run_stage_pair([p["cv"]], [p["sr"]], files, td, "hip", cores) for _ in range(reps)]
            one = [run_stage_pair([p["cv"]], [p["sr"]], files, td, "hip1", cores, env=no_detach) for _ in range(reps)]
            d = {"contigs": n_contigs, "aligned_bp": bp,
                 "dropin_s": {"call_variants": med([a for a, _ in one]), "separate_reads": med([b for _, b in one]), "total": med([a + b for a, b in one]),
                              "note": "HS_NO_DETACH=1: one process per stage, timed to its exit"},
                 "dropin_detached_s": {"call_variants": med([a for a, _ in det]), "separate_reads": med([b for _, b in det]), "total": med([a + b for a, b in det]),
                                       "note": "as started by hairsplitter.py: the process exits when the outputs are complete, teardown in the background"}}
            d["dropin_bp_per_s"] = bp / d["dropin_s"]["total"]
            base = None
            if with_reference:
                if have_ref:
                    a, b = run_stage_pair([p["ref_cv"]], [p["ref_sr"]], files, td, "ref", cores)
                    kind, threads = "reference", cores
                else:
                    a, b = run_stage_pair([p["oracle"], "call_variants"], [p["oracle"], "separate_reads"], files, td, "ref", 1)
                    kind, threads = "port", 1
                d["reference_s"] = {"call_variants": a, "separate_reads": b, "total": a + b}
                d["speedup"] = (a + b) / d["dropin_s"]["total"]
                d["speedup_detached"] = (a + b) / d["dropin_detached_s"]["total"]
                d["reference_kind"] = kind
                gro_ref, against = os.path.join(td, "ref.gro"), "oracle restatement (oracle/_build/hs_oracle), seed 12345"
                if kind == "reference":
                    gro_ref, against = None, "reference HS_call_variants (.col, error rate); no seeded HS_separate_reads on this box: .gro not compared"
                    if have_seeded:      # the reference's stage 4 with std::random_device pinned, on the reference's own .col: what the labels are compared with
                        b_seeded, gro_ref = run_stage4_on([p["ref_sr_seeded"]], os.path.join(td, "ref.col"), os.path.join(td, "ref.err"), td, "refseed", cores)
                        against = "reference binaries built from /root/reference (oracle/_ref): HS_call_variants + HS_separate_reads_seeded (std::random_device pinned to 12345)"
                        d["reference_seeded_s"] = {"call_variants": a, "separate_reads": b_seeded, "total": a + b_seeded,
                                                   "note": "the same reference with std::random_device pinned (no entropy read per Chinese-Whispers sweep): the deterministic build the parity gate compares with"}
                        d["speedup_vs_seeded_reference"] = (a + b_seeded) / d["dropin_s"]["total"]
                if parity_fn is not None and parity[0] is None:
                    parity[0] = parity_fn(os.path.join(td, "ref.col"), None if subset else gro_ref, os.path.join(td, "ref.err"), against, subset)
                base = {"value": bp / (a + b), "unit": "aligned read-bp/s", "cores": threads, "kind": kind,
                        "sample": f"{tag} ({bp} aligned bp), stage 3+4 file to file, one run: {a + b:.2f} s wall"
                                  + (f", -t {threads} (contig-level OpenMP only)" if kind == "reference" else ", single thread") + "; ONE run, not a median",
                        "note": "file to file (parsing 2 GB of text included): compare with file_to_file.*.dropin_bp_per_s, NOT with `value` (HBM-resident steps)"}
            return d, base

    base = None
    if job_files is not None:
        out["job"], base = leg(job_files, n_job, reference_on_full_job, f"the whole job: the {n_job} contigs of {cfg}", False)
    if base is None and sample_files is not None:
        out["sample"], base = leg(sample_files, len(sample_ids), True, f"the first {len(sample_ids)} of the {n_job} contigs of {cfg}", True)
    return out, base, parity[0]


def main():
    # a run that stops making progress leaves the stacks of its threads on stderr (SIGUSR1 at any time; on its own after
    # HS_BENCH_WATCHDOG_S seconds, default 1500, and exits) instead of sitting there until somebody's timeout
    import faulthandler
    import signal
    faulthandler.enable()
    try:
        faulthandler.register(signal.SIGUSR1, all_threads=True)
    except (AttributeError, ValueError):
        pass
    # re-armed at every step and phase (a slow but progressing run -- a bigger configuration, the reference on the whole job -- is not killed)
    watchdog_s = float(os.environ.get("HS_BENCH_WATCHDOG_S", "900"))
    def pet():
        faulthandler.cancel_dump_traceback_later()
        faulthandler.dump_traceback_later(watchdog_s, exit=True)
    pet()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C4", choices=sorted(WORKLOADS))
    ap.add_argument("--contigs", type=int, default=0, help="number of contigs of the configuration (0 = its own: C2 256, C3 50, C4 500, C5 34)")
    ap.add_argument("--groups", type=int, default=0, help="contig groups (host thread + HIP stream each) per GPU; 0 = min(8, host threads / 3)")
    ap.add_argument("--threads", type=int, default=0, help="host threads for the sequential glue (0 = all cores / ranks)")
    ap.add_argument("--cpu-contigs", type=int, default=-1, help="contigs of the CPU-baseline / file-to-file sample (0 disables both; -1 = about 80 M aligned bp)")
    ap.add_argument("--no-f2f-job", action="store_true", help="skip the file-to-file run of the drop-ins on the whole job")
    ap.add_argument("--no-f2f-reference-full", action="store_true", help="do not time the reference on the files of the whole job (C4: ~40 s); a sample of the job instead")
    ap.add_argument("--f2f-runs", type=int, default=2)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--as-rank-of", type=int, default=0, help="readiness check without a node: run rank 0's LPT shard of an N-rank job on this one GPU with the "
                    "host threads / groups a rank of N gets (the line says so in config.emulated_rank_of; `value` is this ONE rank's rate)")
    ap.add_argument("--cores", type=int, default=0, help="pin the process to this many CPUs before anything starts (with --as-rank-of: the share of the host one rank has)")
    args = ap.parse_args()

    rank_env, world_env = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if args.cores > 0:
        os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:args.cores])
    emulated = args.as_rank_of if (args.as_rank_of > 1 and world_env == 1) else 0
    if emulated:
        world_env = emulated      # shard, threads and groups as rank 0 of that many; one process, no collective
    cfg = args.config
    n_job = args.contigs or {"C2": 256, "C3": 50, "C4": 500, "C5": 34}[cfg]
    from hairsplitter_amd import synth, dist as hdist
    shapes = synth.config_shapes(cfg, seed=args.seed, count=n_job)
    # ---- this rank's shard of the job: longest-processing-time over the contig lengths (depth is constant inside a
    # configuration, so length ~ aligned bp; a scheduler knows the lengths from the assembly). Generated first, on forked
    # workers, while this process has not touched the GPU (or loaded torch) yet; rank 0 of a single-GPU run also writes the
    # job's three input files for the file-to-file leg ----
    shards = hdist.lpt_shards([float(s[0] * s[2]) for s in shapes], world_env)
    my_ids = shards[rank_env]
    want_f2f = world_env == 1 and args.cpu_contigs != 0      # (never for an emulated rank)
    job_dir = tempfile.mkdtemp(prefix="hs_bench_job_") if (want_f2f and not args.no_f2f_job) else None
    t_gen = time.perf_counter()
    # HS_BENCH_SERIAL_SETUP=1: no forked workers (under `rocprofv3 --pmc` the profiler has initialised the GPU before this program
    # starts, and a fork from such a process hangs on this pool: tools/pmc_traffic.sh sets it)
    gen_workers = 1 if os.environ.get("HS_BENCH_SERIAL_SETUP") else max(1, min(8, effective_cores() // (1 if emulated else world_env)))
    contigs, job_files = synth.generate_job(cfg, my_ids, seed=args.seed, workers=gen_workers, outdir=job_dir)
    t_gen = time.perf_counter() - t_gen
    # The contigs of the resident batch in the order of their lengths, longest first (a scheduler knows the lengths from the assembly:
    # the same knowledge the LPT shards use). The contig groups of the pipeline are consecutive ranges cut by aligned bases, so the
    # first groups hold the few long contigs and the last one -- whose chain ends the step -- many short ones: its sequential per-contig
    # walks (loop A) are short and spread over its threads. Results are per contig; the parity gate compares by contig NAME.
    # OPT-IN (HS_BENCH_CONTIG_ORDER=length): alternating runs on one box gave 16.2 / 14.6 ms against 16.5 / 14.5 in the order of the job's files -- inside the box's own spread.
    if os.environ.get("HS_BENCH_CONTIG_ORDER", "job") == "length" and len(contigs) > 1:
        order = sorted(range(len(contigs)), key=lambda i: (-len(contigs[i].seq), i))
        contigs = [contigs[i] for i in order]
        my_ids = [my_ids[i] for i in order]
    # the files of the file-to-file sample too, NOW: forking workers from a process that has initialised the GPU (runtime threads,
    # their locks copied mid-flight into the child) hangs now and then
    sample_dir, sample_files, n_sample = None, None, 0
    if want_f2f and (args.no_f2f_reference_full or job_dir is None):
        if args.cpu_contigs > 0:
            n_sample = min(args.cpu_contigs, n_job)
        else:   # about 80 M aligned bp: ~2-4 s of the reference on 16 cores
            acc = 0.0
            while n_sample < n_job and acc < 80e6:
                acc += shapes[n_sample][0] * shapes[n_sample][2]; n_sample += 1
            n_sample = max(n_sample, min(n_job, effective_cores()))
        sample_dir = tempfile.mkdtemp(prefix="hs_bench_sample_")
        sample_contigs, sample_files = synth.generate_job(cfg, list(range(n_sample)), seed=args.seed, workers=gen_workers, outdir=sample_dir)
        sample_files["aligned_bp"] = int(sum(c.aligned_bp for c in sample_contigs))
        del sample_contigs

    import torch
    import torch.distributed as dist
    from hairsplitter_amd import api
    # PyTorch is only the allocator / collective layer here: keep its CPU thread pools out of the way of the host glue
    torch.set_num_threads(1)
    try:
        torch.set_num_interop_threads(1)
    except RuntimeError:
        pass

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = emulated or int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ   # launched by torch.distributed.run
    # HS_BENCH_SHARE_DEVICE=1 (a test of the N > 1 path on a box with ONE GPU): every rank on device 0, every exchange over gloo -- RCCL
    # does not take two ranks on one device; sharding, per-rank pipelines, the exchanges and rank 0's line are what a node runs
    share_device = use_dist and os.environ.get("HS_BENCH_SHARE_DEVICE") == "1"
    if share_device:
        local_rank = 0
        torch.cuda.set_device(0)
        dist.init_process_group(backend="gloo")
    elif use_dist:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
    else:
        local_rank = 0
        torch.cuda.set_device(0)
    # tiny host-side exchanges (error rate, window size) go over gloo; RCCL is used for the one gather of labels per step
    cpu_group = dist.new_group(backend="gloo") if use_dist else None
    api.require_gpu()
    api.load().hs_set_device(local_rank)
    # host threads for the sequential glue: the parallel sections are short, and waking hundreds of workers on a busy
    # box costs more than it buys (measured: 256 threads -> 15-60 ms steps, 64 threads -> 11.6 ms steps)
    n_threads = args.threads or max(1, min(64, (3 * effective_cores()) // (1 if args.cores > 0 else world)))      # (16 usable cores: 32 threads 46 ms per step, 48: 44, 64: 47, 128: 51)

    B = len(contigs)
    contig_names = [c.name for c in contigs]
    # contig groups of this process's pipeline: as many as its threads feed (three per group) and as its share of the job is worth -- a group's
    # chain has a fixed part (five waits, ~60 launches), so a 1/8 shard wants 3 groups, a 1/4 shard 4, the whole 500-contig job 8 (measured
    # with 16-48 threads: rank of 8 3.7-4.0 ms at 3 groups against 4.1-4.9 at 4-8; rank of 4 6.6 ms at 4 against 7.1-7.4 at 6-8): sqrt(aligned bp / 24 M)
    shard_bp = float(sum(shapes[i][0] * shapes[i][2] for i in my_ids))      # (length x depth of the shard's contigs: what a scheduler knows before anything is parsed)
    G_size = max(2, min(8, int(round((shard_bp / 24e6) ** 0.5))))
    G = max(1, min(args.groups if args.groups > 0 else min(G_size, max(1, n_threads // 3)), max(B, 1)))      # (a rank of 8 with 6 threads: 2 groups, 7.0 ms per step on its shard against 10.3 with one)
    pet()
    t_up = time.perf_counter()
    batch = api.PipelineGroups(contigs, G)   # inputs now resident in HBM; the streaming kernels run once per step over all of them
    t_upload = batch.batch.create_s           # hs_cv_batch_create: H2D of the job's flat arrays + the launch plans (CIGAR spans, tile plan, pileup tasks), once per job
    t_flatten = time.perf_counter() - t_up - t_upload   # (Python: the synthetic contigs flattened into those arrays -- not part of the path)
    local_bp = batch.aligned_bp
    if job_files is not None:
        job_files["aligned_bp"] = int(local_bp)
    # window size of stage 4: chosen over the reads of the WHOLE job (separate_reads.cpp:1466-1498), i.e. across the ranks
    window_size = hdist.global_window_size(batch.flat.rec_refspan, group=cpu_group) if use_dist else 0
    contigs = None

    # A job that lives in ONE process goes through hs_pipeline_run_fused: every contig group brings up its own share of the pileup at
    # the head of its chain on the device and the job's error rate is formed inside the library (round 4, no host wait before the
    # candidates: 18.5 ms per C4 step against 21.0 with the two calls on the same box). With several ranks the per-contig distances
    # cross the processes between hs_pipeline_select and hs_pipeline_run. HS_BENCH_TWO_CALLS=1: the two calls in one process too.
    fused = (not use_dist) and not emulated and not os.environ.get("HS_BENCH_TWO_CALLS")
    py_ms = {"pipeline_call": 0.0, "error_rate": 0.0, "gather": 0.0}
    no_coll = bool(os.environ.get("HS_BENCH_NO_COLLECTIVES"))   # diagnostic only
    cap = [None]

    my_ids_np = np.asarray(my_ids, dtype=np.int32)

    def error_rate_fn(cv):
        # the one cross-contig quantity of the path: mean of the per-contig distances over the WHOLE job, in contig order
        t = time.perf_counter()
        er = hdist.global_error_rate(my_ids, cv["mean_distance"], n_job, group=cpu_group) if not no_coll else float(cv["error_rate"])
        py_ms["error_rate"] += (time.perf_counter() - t) * 1e3
        return py_error_rate(er)

    def step():
        t = time.perf_counter()
        if fused:      # one process = the whole job: pileup per contig group, the error rate formed inside the library
            cv, sr = batch.run_fused(0.33, n_threads, rarest_strain_abundance=0.01, window_size=window_size)
        else:
            cv, sr = batch.run(0.33, n_threads, error_rate_fn, rarest_strain_abundance=0.01, window_size=window_size)
        t3 = time.perf_counter(); py_ms["pipeline_call"] += (t3 - t) * 1e3
        # the labels arrive on rank 0 (which would write the .gro) as the lists its GROUP lines are made of: per window the reads it holds
        # and their labels; decoding them into arrays is the consumer's business. cap[0] is made during set-up (below), not here.
        gathered = None
        if cap[0] is not None and not no_coll:
            # ... and END THE STEP ON RANK 0'S HOST, as the single process's step does: the gathered payloads are copied into one pinned block and
            # that copy is waited for inside the step (to_host). Every window carries the contig of the JOB it belongs to.
            off, ids, lab = sr["sparse"]
            gathered = cap[0].gather(off, ids, lab, decode=False, win_contig=np.repeat(my_ids_np, np.diff(sr["win_off"])), to_host=True)
        py_ms["gather"] += (time.perf_counter() - t3) * 1e3
        return cv, sr, gathered

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # the synthetic inputs are a few hundred thousand Python / numpy objects that live for the whole run: park them in the
    # permanent generation so that a full garbage collection in the middle of a step does not walk them (25-30 ms)
    import gc
    gc.collect()
    gc.freeze()
    # set-up, not measurement (reported as setup_steps): a few passes that size the device / pinned block pools, the
    # per-thread scratch and the HIP runtime's own pools (first use of every buffer size goes to hipMalloc / hipHostMalloc),
    # then the W warm-up steps of the contract
    if not os.environ.get("HS_BENCH_DENSE_LABELS"):      # (diagnostic switch: the dense [window][reads of the contig] array of the C result instead)
        batch.sparse_labels(True)      # a step returns what the .gro lists: per window its reads and their labels (config.outputs)
    pet()
    _cv0, _sr0, _ = step()      # set-up: the first pass sizes the library's pools -- and the per-step collective (largest payload of any rank, once)
    if use_dist and "sparse" in _sr0:      # (one rank: its lists are the job's, nothing to gather)
        cap[0] = hdist.SparseLabelGatherer(hdist.SparseLabelGatherer.job_capacity(hdist.sparse_payload_bytes(int(_sr0["sparse"][0].size) - 1, int(_sr0["sparse"][1].size), with_contigs=True)))
    _cv0 = _sr0 = None
    for _ in range(SETUP_STEPS - 1):
        pet(); step()
    sync()
    for _ in range(args.warmup):
        pet(); step()
    sync()
    for k in py_ms:
        py_ms[k] = 0.0
    thr0 = throttle_stats()
    if os.environ.get("HS_CPU_PROFILE"):      # diagnostic: sampling profile of the host side over the timed steps (tools/cpuprof_report.py)
        api.load().hs_cpuprof_start(os.environ["HS_CPU_PROFILE"].encode())
    # The library's timing events (two per kernel launch) are recorded on every STATS_EVERY-th step of the timed region (default: every tenth, at
    # least two steps): ~1200 event records
    # per C4 step cost the step 1.3 ms when every step is instrumented (alternating runs on one box: 15.6 against 16.9 ms per step). The
    # per-kernel rows below are averages over the instrumented steps; `value` is the mean over ALL timed steps, instrumented or not.
    STATS_EVERY = max(1, int(os.environ.get("HS_BENCH_STATS_EVERY", str(max(1, min(10, args.steps // 2))))))      # (two instrumented steps of the default twenty)
    api.kernel_stats_every(STATS_EVERY)
    K_TIMED = (args.steps + STATS_EVERY - 1) // STATS_EVERY
    api.kernel_stats_reset()
    waits0 = api.host_waits()
    def thread_cpu():      # HS_BENCH_THREAD_CPU=1: CPU time per thread NAME over the timed region (the library names its own threads; the rest is Python + the ROCm runtime)
        acc = {}
        for tid in os.listdir("/proc/self/task"):
            try:
                name = open("/proc/self/task/%s/comm" % tid).read().strip()
                f = open("/proc/self/task/%s/stat" % tid).read().rsplit(")", 1)[1].split()
                acc.setdefault(name, [0, 0, 0]); acc[name][0] += int(f[11]); acc[name][1] += int(f[12]); acc[name][2] += 1
            except Exception:
                pass
        return acc
    tcpu0 = thread_cpu() if os.environ.get("HS_BENCH_THREAD_CPU") else None
    t0 = time.perf_counter(); cpu0 = time.process_time()
    t_dev = 0.0; t_host = 0.0
    last = None
    step_ms = []
    wall = {}
    last_results = None
    last_gathered = None
    for it in range(args.steps):
        pet()
        ts = time.perf_counter()
        cv, sr, gathered = step()
        step_ms.append((time.perf_counter() - ts) * 1e3)
        if it == args.steps - 1:
            last_results = (cv, sr)      # the LAST TIMED STEP's results: what the parity gate compares with the reference (after the clock stops)
            last_gathered = gathered
        for kk, vv in sr.get("wall_ms", {}).items():
            wall[kk] = wall.get(kk, 0.0) + vv
        t_dev += cv["t_device_ms"] + sr["t_device_ms"]; t_host += cv["t_host_ms"] + sr["t_host_ms"]
        last = {k: sr[k] for k in ("n_cw_instances", "n_cw_sweeps", "cw_bytes", "graph_nnz", "n_graph_rows", "simdiff_bytes", "n_graph_rows_host",
                                   "n_windows_finished_on_host")}
        last["n_snps"] = cv["n_snps"]; last["n_windows"] = int(sr["win_off"][-1])
        cv = sr = gathered = None   # no references to the step's arrays: they die here, as in the warm-up
    sync()
    dt = time.perf_counter() - t0
    cpu_ms_per_step = (time.process_time() - cpu0) / args.steps * 1e3
    if tcpu0 is not None:
        tcpu1 = thread_cpu(); tick = 1e3 / os.sysconf("SC_CLK_TCK")
        rows = sorted(((n, (v[0] - tcpu0.get(n, [0, 0, 0])[0]) * tick / args.steps, (v[1] - tcpu0.get(n, [0, 0, 0])[1]) * tick / args.steps, v[2]) for n, v in tcpu1.items()), key=lambda r: -(r[1] + r[2]))
        sys.stderr.write("[thread cpu] per step, by thread name: " + "; ".join("%s x%d user %.1f sys %.1f ms" % (n, k, u, sy) for n, u, sy, k in rows if u + sy > 0.05) + "\n")
    # ---- parity gate, part 1: copies of what the last timed step left with the host (its arrays belong to the pipeline and die with the
    # next call). Compared with the reference's files in the file-to-file leg below; one rank = the whole job only ----
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    snap_timed = snap_col = None
    if rank == 0 and world == 1 and not emulated and last_results is not None:
        import ref_outputs
        snap_timed = ref_outputs.pipeline_snapshot(batch, last_results[0], last_results[1], contig_names)
    # what rank 0 holds of the WHOLE job after the last timed step, as an order-independent digest: one process and N ranks must agree on it
    labels_digest = None
    try:
        if rank == 0 and last_results is not None:
            if use_dist and last_gathered is not None:
                labels_digest = dict(hdist.sparse_digest([hdist.decode_sparse(np.asarray(o)) for o in last_gathered]), ranks=len(last_gathered))
            elif not use_dist and "sparse" in last_results[1]:
                labels_digest = dict(hdist.sparse_digest([tuple(np.array(x) for x in last_results[1]["sparse"]) + (np.repeat(my_ids_np, np.diff(last_results[1]["win_off"])),)]), ranks=1)
    except Exception as e:
        sys.stderr.write("labels digest failed: %r\n" % (e,))
    # ... and what stage 3 leaves beside the labels: the job's SNP count, the contigs' mean distances (error_rate.txt is their mean) -- summed
    # / put together over the ranks after the clock has stopped (two tiny host-side exchanges every rank takes part in)
    try:
        if last_results is not None and not emulated:
            import zlib
            md_full = np.zeros(n_job, np.float32); md_full[my_ids_np] = np.asarray(last_results[0]["mean_distance"], np.float32)[:len(my_ids)]
            snps = np.array([int(last_results[0].get("n_snps", 0))], np.int64)
            if use_dist:
                t_md = torch.from_numpy(md_full); t_sn = torch.from_numpy(snps)
                dist.all_reduce(t_md, op=dist.ReduceOp.SUM, group=cpu_group); dist.all_reduce(t_sn, op=dist.ReduceOp.SUM, group=cpu_group)
            if labels_digest is not None:
                labels_digest["n_snps"] = int(snps[0])
                labels_digest["mean_distance_crc32"] = int(zlib.crc32(md_full.tobytes()))
                labels_digest["error_rate"] = "%g" % hdist.mean_of_positive_f32(md_full)
    except Exception as e:
        sys.stderr.write("digest of the stage-3 results failed: %r\n" % (e,))
    last_results = None
    last_gathered = None
    waits_per_step = (api.host_waits() - waits0) / args.steps
    if os.environ.get("HS_CPU_PROFILE"):
        api.load().hs_cpuprof_stop()
    kstats = api.kernel_stats()
    api.kernel_stats_every(1)      # (the one-group probe below times every step)
    thr1 = throttle_stats()
    # After the timed region: the same job with ONE contig group (every kernel alone on the GPU), a few steps, to put the kernels' own
    # durations next to the ones above -- with G groups a launch shares the GPU with the other groups' kernels and lasts longer.
    kstats_alone = None
    PROBE_STEPS = 5
    # (not under a profiler: rocprofv3's per-kernel averages of this command stay those of the timed configuration)
    profiled = any(os.environ.get(k) for k in ("ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD")) or "rocprof" in os.environ.get("LD_PRELOAD", "")

    def any_step(pl):
        pet()
        if fused:
            cv1, sr1 = pl.run_fused(0.33, n_threads, rarest_strain_abundance=0.01, window_size=window_size)
        else:
            cv1, sr1 = pl.run(0.33, n_threads, error_rate_fn, rarest_strain_abundance=0.01, window_size=window_size)
        return cv1, sr1

    if G > 1 and not use_dist and not profiled and not os.environ.get("HS_BENCH_NO_PROBE"):
        probe = None
        try:
            probe = batch.sibling(1)
            any_step(probe); any_step(probe); sync()      # (its first pass sizes its arrays the careful way)
            api.kernel_stats_reset()
            for _ in range(PROBE_STEPS):
                any_step(probe)
            sync()
            kstats_alone = api.kernel_stats()
        except Exception as e:      # (a diagnostic leg: the line goes out without it)
            kstats_alone = None
            sys.stderr.write("one-group probe failed: %r\n" % (e,))
        finally:
            if probe is not None:
                probe.close()      # (a sibling leaves the batch alone)
    # After the timed region too: the same steps with .col's payload (the entries of every SNP column) brought to the host inside
    # the call -- the timed steps hand stage 3 -> 4 over on the device and return the SNPs' positions, alleles and counts only
    ms_with_col = None
    if (not use_dist or world == 1) and not profiled and not os.environ.get("HS_BENCH_NO_PROBE"):
        try:
            batch.keep_columns(True)
            for _ in range(4):      # (the pinned blocks of the entries are sized in these)
                any_step(batch)
            sync()
            tc0 = time.perf_counter()
            for _ in range(10):
                res_col = any_step(batch)
            sync()
            ms_with_col = (time.perf_counter() - tc0) / 10 * 1e3
            if snap_timed is not None:      # (with the SNP columns' entries: the .col's SNPS lines are compared whole)
                snap_col = ref_outputs.pipeline_snapshot(batch, res_col[0], res_col[1], contig_names)
            res_col = None
        except Exception as e:
            sys.stderr.write("column-download leg failed: %r\n" % (e,))
        finally:
            try:
                batch.keep_columns(False)
            except Exception:
                pass
    throttled = None if thr0 is None or thr1 is None else {"periods": thr1[0] - thr0[0], "usec": thr1[1] - thr0[1]}
    if use_dist:
        red_dev = "cpu" if share_device else "cuda"
        tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([local_bp], dtype=torch.int64, device=red_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_bp = int(tot.item())
    else:
        total_bp = local_bp

    parity_failed = False
    if rank == 0:
        K = args.steps
        # ---- roofline of the kernel family with the largest time per step (all kernels of the path compete) ----
        per_step = {k: {"ms_per_step": v["ms"] / K_TIMED, "launches_per_step": v["launches"] / K_TIMED, "avg_launch_ms": v["ms"] / max(1, v["launches"]),
                        "algorithmic_bytes_per_launch": v["bytes"] / max(1, v["launches"]),
                        "achieved_GBs": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else 0.0} for k, v in kstats.items()}
        if not per_step:       # HS_NO_KERNEL_STATS=1 (diagnostic): no roofline leg
            per_step = {"none": {"ms_per_step": 0.0, "launches_per_step": 0.0, "avg_launch_ms": 0.0, "algorithmic_bytes_per_launch": 0.0, "achieved_GBs": 0.0}}
        # the dominant kernel = the one with the largest time per step when every kernel runs ALONE on the GPU (the one-group leg after the
        # timed region); in the timed region the contig groups' kernels share the GPU and a launch lasts longer for reasons that are not its own
        alone_ms = {k: v["ms"] / PROBE_STEPS for k, v in (kstats_alone or {}).items() if k not in ("k_ship", "other")}
        timed_ms = {k: v["ms_per_step"] for k, v in per_step.items() if k not in ("k_ship", "other")}
        dom = max(alone_ms, key=alone_ms.get) if alone_ms else max(timed_ms or {"none": 0.0}, key=(timed_ms or {"none": 0.0}).get)
        if alone_ms:      # K1 and K2 are within a per cent of each other: kernels within 3 % of the longest count as tied, and a tie goes to the one that
            top = alone_ms[dom]      # moves more algorithmic bytes per launch -- the line then names the same kernel run after run; the other one is `runner_up`
            tied = [k for k, v in alone_ms.items() if v >= 0.97 * top]
            dom = max(tied, key=lambda k: (kstats_alone[k]["bytes"] / max(1, kstats_alone[k]["launches"]), alone_ms[k]))
        d = per_step.get(dom, {"ms_per_step": 0.0, "launches_per_step": 0.0, "avg_launch_ms": 0.0, "algorithmic_bytes_per_launch": 0.0, "achieved_GBs": 0.0})
        traffic = None
        traffic_src = None
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("_config") == cfg and tj.get("_contigs") == n_job and world == 1 and tj.get("_groups", tj.get("_groups_per_gpu")) == 1:      # (per launch of the whole job, like `achieved`)
                    traffic = tj.get(dom)      # HBM bytes per launch from the PMC passes at THIS size (tools/pmc_traffic.sh)
                    traffic_src = {"file": "profiles/traffic_latest.json", "measured_at_commit": tj.get("_commit"), "groups": tj.get("_groups"),
                                   "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this command (tools/pmc_traffic.sh), stored, not measured in this run"}
            except Exception:
                traffic = None
        # ---- whole path by SURVEY.md 8(d)'s formula with the counts the run itself emitted ----
        t_kernels_ms = sum(v["ms_per_step"] for v in per_step.values())
        whole_bytes = 4.0 * local_bp + float(last["simdiff_bytes"]) + float(last["cw_bytes"])
        whole = {"formula": "4 B x aligned bp + sum_contigs(N*S/4 + 16*N^2) + sum_CW-runs sweeps*(4*nnz + 8*m)",
                 "bytes_per_step": whole_bytes, "terms": {"4_bytes_per_aligned_bp": 4.0 * local_bp, "simdiff": float(last["simdiff_bytes"]), "chinese_whispers": float(last["cw_bytes"])},
                 "counts": {"aligned_bp": int(local_bp), "cw_runs": int(last["n_cw_instances"]), "cw_sweeps": int(last["n_cw_sweeps"]), "graph_nnz": int(last["graph_nnz"]),
                            "graph_rows": int(last["n_graph_rows"]), "windows": int(last["n_windows"]), "snps": int(last["n_snps"])},
                 "sum_of_kernel_ms_per_step": t_kernels_ms,
                 "frac_vs_kernel_time": whole_bytes / (t_kernels_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if t_kernels_ms > 0 else None,
                 "frac_vs_step_time": whole_bytes / (dt / K) / 1e9 / HBM_PEAK_GBS,
                 "note": "rank 0's shard; m = masked reads of the window (the kernels work in the window's local index space), not the N reads of the contig"}
        alone = None
        if kstats_alone and dom in kstats_alone and kstats_alone[dom]["ms"] > 0:
            a = kstats_alone[dom]
            a_gbs = a["bytes"] / (a["ms"] * 1e-3) / 1e9
            alone = {"groups": 1, "steps": PROBE_STEPS, "launches_per_step": a["launches"] / PROBE_STEPS, "avg_launch_ms": a["ms"] / max(1, a["launches"]),
                     "ms_per_step": a["ms"] / PROBE_STEPS, "algorithmic_bytes_per_launch": a["bytes"] / max(1, a["launches"]), "achieved": a_gbs, "frac": a_gbs / HBM_PEAK_GBS,
                     "kernels_ms_per_step": {k: round(v["ms"] / PROBE_STEPS, 4) for k, v in sorted(kstats_alone.items(), key=lambda kv: -kv[1]["ms"])[:14]}}
            # the runner-up beside it (K1 and K2 are within a few per cent of each other: which of them leads can change from run to run)
            second = sorted((k for k in alone_ms if k != dom), key=lambda k: -alone_ms[k])[:1]
            if second and kstats_alone[second[0]]["ms"] > 0:
                b2 = kstats_alone[second[0]]
                b_gbs = b2["bytes"] / (b2["ms"] * 1e-3) / 1e9
                alone["runner_up"] = {"kernel": second[0], "avg_launch_ms": b2["ms"] / max(1, b2["launches"]), "ms_per_step": b2["ms"] / PROBE_STEPS,
                                      "algorithmic_bytes_per_launch": b2["bytes"] / max(1, b2["launches"]), "achieved": b_gbs, "frac": b_gbs / HBM_PEAK_GBS}
        out = {
            "metric": "aligned read-bp/sec through call_variants+separate_reads",
            "value": total_bp * K / dt, "unit": "aligned read-bp/s", "n_gpus": 1 if emulated else world, "steps": K, "warmup": args.warmup, "setup_steps": SETUP_STEPS,
            "ms_per_step": dt / K * 1e3, "ms_per_step_with_col_download": ms_with_col,
            "value_with_col_download": (total_bp / (ms_with_col * 1e-3)) if ms_with_col else None,
            "labels_digest": labels_digest,      # windows / entries / sum of per-window CRC-32 of (reads, labels) rank 0 holds for the whole job
            "value_excludes_col_payload": True,      # the timed steps hand the SNP columns' entries (.col's payload) from stage 3 to stage 4 ON THE DEVICE (config.outputs)
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "host": {"cpus_online": os.cpu_count(), "cpus_effective": effective_cores(), "process_cpu_ms_per_step": cpu_ms_per_step,
                     "cfs_throttled_during_timed_steps": throttled, "input_generation_s": t_gen, "waits_per_step": waits_per_step,
                     "note": "waits = host round trips to the device summed over the contig groups"},
            "host_fallbacks_per_step": {"windows_finished_on_host": int(last["n_windows_finished_on_host"]), "of_windows": int(last["n_windows"]),
                                        "graph_rows_resolved_by_std_sort_on_host": int(last["n_graph_rows_host"]), "of_graph_rows": int(last["n_graph_rows"])},
            "with_h2d_and_plans": {"h2d_plus_launch_plans_s": t_upload, "python_flattening_s_not_counted": t_flatten,
                                   "bp_per_s_one_job_from_host_buffers": total_bp / (t_upload + dt / K) if world == 1 else None,
                                   "note": "host buffers in (hs_cv_batch_create: PCIe upload + launch plans, once per job) + one step; never `value`"},
            "config": {"workload": WORKLOADS[cfg].format(n=n_job) + "; the whole job per step, inputs resident in HBM",
                       "config": cfg, "contigs": n_job, "aligned_bp": total_bp, "contigs_rank0": B, "aligned_bp_rank0": int(local_bp),
                       "parallelism": f"contigs sharded over {world} GPU(s) by LPT on contig length", "groups_per_gpu": G,
                       **({"emulated_rank_of": emulated, "cores_pinned": args.cores or None,
                           "note": "ONE rank of an %d-rank job on one GPU (its LPT shard, its threads and groups): a readiness check, not a scaling measurement" % emulated} if emulated else {}),
                       "host_threads_per_rank": n_threads, "pipeline": "hs_pipeline_run_fused" if fused else "hs_pipeline_select + hs_pipeline_run",
                       "outputs": "per step, on the host: for every clustering window its bounds, the reads it holds and their partition labels (hs_pipeline_sparse_labels: the "
                                  "content of the GROUP lines of the .gro; every other read of the contig has the label -2), per contig the mean distance (error_rate.txt) and the SNPs' positions, "
                                  "alleles and read counts. NOT in the timed steps: the per-read entries of the SNP columns (.col's payload) -- stage 3 hands them to stage 4 on "
                                  "the device; `ms_per_step_with_col_download` is the same step with them brought to the host as well"},
            "roofline": {"bound": "hbm", "kernel": dom,
                         # PRIMARY: the dominant kernel INSIDE THE TIMED REGION -- HIP events on each launch stream over the instrumented timed steps; the
                         # contig groups overlap there, so a launch shares the GPU with the other groups' kernels (the kernel on its own: `probe_one_group`)
                         "achieved": d["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["achieved_GBs"] / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": d["avg_launch_ms"], "launches_per_step": d["launches_per_step"], "ms_per_step": d["ms_per_step"],
                         "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
                         "measured": "HIP events on the launch streams over the instrumented steps of the timed region (%d contig groups overlap)" % G,
                         "selection": "the kernel with the largest time per step when it runs alone (one-group probe; all kernels of the path compete; transfers excluded; kernels within 3 % of the longest are tied and the tie goes to the one with more algorithmic bytes per launch: see probe_one_group.runner_up)"
                                      if alone else "the kernel with the largest time per step in the timed region (transfers excluded)",
                         # PROBE (not the timed region): the same resident job through a ONE-group pipeline right after the timed region, every kernel alone
                         # on the GPU, one launch per step over the whole job -- the kernel's own rate, reproducible from profiles/*_groups1.csv
                         "frac_probe_one_group": alone["frac"] if alone else None,
                         "probe_one_group": alone,
                         "whole_path": whole},
            "kernel_timing": {"events_on_every_nth_step": STATS_EVERY, "instrumented_steps": K_TIMED, "of_steps": args.steps,
                              "note": "per-kernel rows and roofline.timed_region: HIP events on the launch streams over the instrumented steps of the timed region; `value` and ms_per_step: all timed steps"},
            "kernels": {k: {kk: (round(vv, 6) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in sorted(per_step.items(), key=lambda kv: -kv[1]["ms_per_step"])},
            "step_ms": [round(x, 2) for x in step_ms],
            "pipeline_wall_ms_per_step": {k: v / K for k, v in wall.items()},
            "phase_ms_per_step": {"device_phases_summed_over_groups": t_dev / K, "host_glue_summed_over_groups": t_host / K, **{"py_" + k: v / K for k, v in py_ms.items()}},
        }
        batch.close()
        batch = None
        if want_f2f:
            try:
                pet()
                faulthandler.cancel_dump_traceback_later()      # (the reference on the whole job takes its time; every subprocess has its own limit)
                def parity_fn(col, gro, err, against, subset):
                    if snap_timed is None:
                        return None
                    t_p = time.perf_counter()
                    r = ref_outputs.compare_with_reference(snap_timed, col, gro, err, subset=subset)
                    r["against"] = against
                    r["what"] = "the results of the LAST TIMED STEP (%s, %d contig groups, sparse labels, size hints of the step before)" % ("hs_pipeline_run_fused" if fused else "hs_pipeline_select + hs_pipeline_run", G)
                    if snap_col is not None:      # the same job one leg later, with the SNP columns' entries on the host: the SNPS lines whole
                        rc_ = ref_outputs.compare_with_reference(snap_col, col, gro, err, subset=subset)
                        r["col_entries_identical"] = rc_["col_entries_identical"]
                        r["col_download_step_identical"] = rc_["identical"]
                        if not rc_["identical"]:
                            r["identical"] = False
                            r.setdefault("diffs", []).extend(rc_.get("diffs", []))
                    keys = ("gro_identical", "col_snps_identical", "col_entries_identical", "error_rate_identical")
                    r["identical_scope"] = "compared: " + ", ".join(k[:-len("_identical")] for k in keys if r.get(k) is not None) + \
                        ("; NOT compared: " + ", ".join(k[:-len("_identical")] for k in keys if r.get(k) is None) if any(r.get(k) is None for k in keys) else "")
                    r["seconds"] = round(time.perf_counter() - t_p, 2)
                    return r
                f2f, base, par = file_to_file(cfg, n_job, list(range(n_sample)), sample_files, job_files, max(1, args.f2f_runs), not args.no_f2f_reference_full, parity_fn)
                if f2f is not None:
                    out["file_to_file"] = f2f
                    out["cpu_baseline"] = base
                if par is not None:
                    out["parity"] = par
            except Exception as e:  # the baseline is informational; never fail the bench on it
                out["cpu_baseline"] = {"error": repr(e)}
        # a run that could not put the reference beside itself (several ranks, --cpu-contigs 0) still has a known answer for the jobs listed in
        # tests/golden/bench_labels_digest.json: the digest of the labels a single process held when ITS reference gate said "identical"
        # (tools/make_bench_digests.py). Only the default seed, and only when rank 0 holds the whole job's labels.
        if "parity" not in out and labels_digest is not None and args.seed is None and not emulated:
            try:
                known = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_labels_digest.json"))).get("%s:%d:default" % (cfg, n_job))
            except Exception:
                known = None
            if known:
                keys = [k for k in ("windows", "entries", "sum_crc32", "n_snps", "mean_distance_crc32", "error_rate") if k in known]
                same = all(labels_digest.get(k) == known[k] for k in keys)
                # (a digest, not the reference's files beside the run: `identical` says so through its scope; the file-level flags stay null)
                out["parity"] = {"checked": True, "kind": "labels digest", "identical": bool(same), "identical_scope": "digest of " + ", ".join(keys),
                                 "gro_identical": None, "col_snps_identical": None, "error_rate_identical": None,
                                 "against": "tests/golden/bench_labels_digest.json: windows / entries / sum of per-window CRC-32 of (contig, reads, labels), the job's SNP count, CRC-32 of the "
                                            "contigs' mean distances and the error rate of the single-process run of this job whose reference gate passed (" + known.get("verified", "")[:120] + ")",
                                 "what": "what rank 0 holds for the whole job after the LAST TIMED STEP (%d rank(s))" % labels_digest.get("ranks", 1),
                                 "not_compared": "the SNPS lines of the .col (positions, alleles, entries): the reference gate of a --gpus 1 run with the file-to-file leg compares them",
                                 "diffs": None if same else {"got": {k: labels_digest.get(k) for k in keys}, "expected": {k: known[k] for k in keys}}}
        if "parity" not in out:
            out["parity"] = {"checked": False, "why": "several ranks (each holds a shard; the full-job comparison runs at --gpus 1)" if (use_dist or emulated) else
                             ("no file-to-file leg in this run (--cpu-contigs 0) or no reference / oracle binaries on this box" if "error" not in (out.get("cpu_baseline") or {}) else "the file-to-file leg failed")}
        print(json.dumps(out), flush=True)
        if out["parity"].get("checked") and not out["parity"].get("identical"):
            sys.stderr.write("PARITY GATE FAILED: the timed path's results differ from the reference's: %r\n" % (out["parity"].get("diffs"),))
            parity_failed = True
    if batch is not None:
        batch.close()
    if job_dir is not None:
        shutil.rmtree(job_dir, ignore_errors=True)
    if sample_dir is not None:
        shutil.rmtree(sample_dir, ignore_errors=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
