#!/usr/bin/env python3
"""bench.py -- aligned read-bp/s through call_variants + separate_reads on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path (stage 3 + stage 4, device kernels and host glue) over one batch of
synthetic contigs whose inputs are already resident in HBM. Workload at any N: BASELINE.json configs[1] (C2:
100 kb contig, 2 haplotypes @1 %, 50x ONT-error reads), `--contigs` independent contigs of that shape per GPU
(weak scaling: per-GPU work is fixed as N grows; contigs are sharded by index over the ranks, the only
collectives are the tiny error-rate exchange and ONE gather of the partition labels to rank 0 per step).

Launched by the driver as
    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


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


def thread_cpu():
    """{tid: (comm, utime + stime in seconds)} of this process's threads"""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    for t in os.listdir("/proc/self/task"):
        try:
            st = open(f"/proc/self/task/{t}/stat").read()
            comm = st[st.index("(") + 1:st.rindex(")")]
            f = st[st.rindex(")") + 2:].split()
            out[int(t)] = (comm, (int(f[11]) + int(f[12])) / tick)
        except Exception:
            pass
    return out


def cpu_baseline(n_contigs: int, seed: int):
    """The compiled reference (oracle/_ref, built from /root/reference by oracle/Makefile) timed file-to-file on this
    box's host cores on a bounded sample of the same workload. Falls back to the oracle restatement ("port")."""
    from hairsplitter_amd import synth
    import __graft_entry__ as ge
    p = ge.paths()
    cores = effective_cores()
    kind = "reference" if os.path.exists(p["ref_cv"]) and os.path.exists(p["ref_sr"]) else "port"
    if kind == "port" and not os.path.exists(p["oracle"]):
        return None
    contigs = [synth.make_contig(seed, 10_000 + i, 100_000, 2, 0.01, 50, "ont") for i in range(n_contigs)]
    bp = sum(c.aligned_bp for c in contigs)
    with tempfile.TemporaryDirectory() as td:
        f = synth.write_files(contigs, td)
        col, vcf, err, gro = (os.path.join(td, x) for x in ("v.col", "v.vcf", "e.txt", "r.gro"))
        if kind == "reference":
            cv, sr, threads = [p["ref_cv"]], [p["ref_sr"]], cores
        else:
            cv, sr, threads = [p["oracle"], "call_variants"], [p["oracle"], "separate_reads"], 1
        t0 = time.perf_counter()
        subprocess.run(cv + [f["gfa"], f["reads"], f["sam"], str(threads), td, err, "0", "0", col, vcf, "0.33"], check=True,
                       stdout=subprocess.DEVNULL)
        e = py_error_rate(float(open(err).read().strip()))
        subprocess.run(sr + [col, str(threads), str(e), os.path.join(td, "no_ploidy"), "0", "0.01", "0", gro, "0"], check=True,
                       stdout=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
    return {"value": bp / dt, "unit": "aligned read-bp/s", "cores": threads if kind == "reference" else 1, "kind": kind,
            "sample": f"{n_contigs} contigs of the C2 shape ({bp} aligned bp), stage 3+4 file-to-file, {dt:.2f} s wall"
                      + (f", -t {threads}; contig-level OpenMP only" if kind == "reference" else ", single thread")}


def _make_contig(a):
    from hairsplitter_amd import synth
    return synth.make_contig(a[0], a[1], 100_000, 2, 0.01, 50, "ont")


def make_contigs(seed, ids, workers):
    """The synthetic C2-shaped contigs `ids` (deterministic per (seed, id)); forked workers when there are many"""
    jobs = [(seed, i) for i in ids]
    if workers <= 1 or len(jobs) < 16 or os.environ.get("HS_BENCH_SERIAL_SETUP"):
        return [_make_contig(j) for j in jobs]
    import multiprocessing as mp
    with mp.get_context("fork").Pool(workers) as pool:
        return pool.map(_make_contig, jobs, chunksize=max(1, len(jobs) // (4 * workers)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--contigs", type=int, default=256, help="C2-shaped contigs per GPU in one batch (256 ~ half of the 500-contig config; the step is a chain of short device calls and host sections per contig group, so small batches are latency-bound)")
    ap.add_argument("--groups", type=int, default=0, help="contig groups (host thread + HIP stream each) per GPU; 0 = min(8, host threads / 4)")
    ap.add_argument("--threads", type=int, default=0, help="host threads for the sequential glue (0 = all cores / ranks)")
    ap.add_argument("--cpu-contigs", type=int, default=8, help="size of the CPU-baseline sample (0 disables)")
    ap.add_argument("--seed", type=int, default=2)
    args = ap.parse_args()

    # ---- this rank's shard: contigs [rank*B, (rank+1)*B) of the job (weak scaling). Generated first, on a few forked
    # workers, while this process has not touched the GPU (or loaded torch) yet ----
    rank_env, world_env = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    my_ids = list(range(rank_env * args.contigs, (rank_env + 1) * args.contigs))
    contigs = make_contigs(args.seed, my_ids, max(1, min(8, effective_cores() // world_env)))

    import torch
    import torch.distributed as dist
    from hairsplitter_amd import api, synth, dist as hdist
    # PyTorch is only the allocator / collective layer here: keep its CPU thread pools out of the way of the host glue
    torch.set_num_threads(1)
    try:
        torch.set_num_interop_threads(1)
    except RuntimeError:
        pass

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ   # launched by torch.distributed.run
    if use_dist:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
    else:
        local_rank = 0
        torch.cuda.set_device(0)
    # tiny host-side exchanges (error rate) go over gloo; RCCL is used for the one gather of labels per step
    cpu_group = dist.new_group(backend="gloo") if use_dist else None
    api.require_gpu()
    api.load().hs_set_device(local_rank)
    # host threads for the sequential glue: the parallel sections are short, and waking hundreds of workers on a busy
    # box costs more than it buys (measured: 256 threads -> 15-60 ms steps, 64 threads -> 11.6 ms steps)
    n_threads = args.threads or max(1, min(64, (4 * effective_cores()) // world))

    B = args.contigs
    G = max(1, min(args.groups if args.groups > 0 else min(8, max(1, n_threads // 4)), B))
    batch = api.PipelineGroups(contigs, G)   # inputs now resident in HBM; the streaming kernels run once per step over all of them
    local_bp = batch.aligned_bp

    py_ms = {"pipeline_call": 0.0, "error_rate": 0.0, "gather": 0.0}
    no_coll = bool(os.environ.get("HS_BENCH_NO_COLLECTIVES"))   # diagnostic only
    cap = [None]

    def error_rate_fn(cv):
        # the one cross-contig quantity of the path: mean of the per-contig distances over the WHOLE job, in contig order
        t = time.perf_counter()
        er = hdist.global_error_rate(my_ids, cv["mean_distance"], world * B, group=cpu_group) if not no_coll else float(cv["error_rate"])
        py_ms["error_rate"] += (time.perf_counter() - t) * 1e3
        return py_error_rate(er)

    def step():
        t = time.perf_counter()
        # window size 2000 for every rank when sharded: it depends on the read lengths of the whole job
        # (separate_reads.cpp:1466-1498) and all shards of this workload have the same read-length distribution
        cv, sr = batch.run(0.33, n_threads, error_rate_fn, rarest_strain_abundance=0.01, window_size=2000 if world > 1 else 0)
        t3 = time.perf_counter(); py_ms["pipeline_call"] += (t3 - t) * 1e3
        if cap[0] is None:   # first (warm-up) step only: fix the size of the per-step collective, allocate its buffers
            cap[0] = hdist.LabelGatherer(hdist.gather_capacity(int(sr["labels"].size)))
        # the labels arrive on rank 0 (which would write the .gro); decoding them into arrays is the consumer's business
        gathered = cap[0].gather(sr["labels"], decode=False) if not no_coll else None
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
    # set-up, not measurement: a few passes that size the device / pinned block pools, the per-thread scratch and the HIP
    # runtime's own pools (first use of every buffer size goes to hipMalloc / hipHostMalloc; the runtime stalls once for
    # ~30 ms around the eighth pass of a process), then the W warm-up steps of the contract
    for _ in range(8):
        step()
    sync()
    for _ in range(args.warmup):
        step()
    sync()
    for k in py_ms:
        py_ms[k] = 0.0
    plain = bool(os.environ.get("HS_BENCH_PLAIN"))
    thr0 = None if plain else throttle_stats()
    tcpu0 = thread_cpu() if os.environ.get("HS_BENCH_THREADS") else None   # diagnostic: CPU time by thread over the timed steps
    t0 = time.perf_counter(); cpu0 = 0.0 if plain else time.process_time()
    k_cv = np.zeros(4); k_sr = np.zeros(4); t_dev = 0.0; t_host = 0.0; k4 = 0.0; k6 = 0.0
    last = None
    step_ms = []
    wall = {}
    for _ in range(args.steps):
        ts = time.perf_counter()
        cv, sr, gathered = step()
        step_ms.append((time.perf_counter() - ts) * 1e3)
        for kk, vv in sr.get("wall_ms", {}).items():
            wall[kk] = wall.get(kk, 0.0) + vv
        if os.environ.get("HS_BENCH_OUTLIERS") and step_ms[-1] > 15:
            sys.stderr.write(f"outlier step {len(step_ms)}: {step_ms[-1]:.1f} ms {sr.get('wall_ms')}\n")
        k_cv += np.asarray(cv["t_kernel_ms"]); k_sr += np.asarray(sr["t_kernel_ms"]); k4 += cv.get("t_kernel_k4_ms", 0.0); k6 += sr.get("t_kernel_graph_ms", 0.0)
        t_dev += cv["t_device_ms"] + sr["t_device_ms"]; t_host += cv["t_host_ms"] + sr["t_host_ms"]
        last = ({"n_snps": cv["n_snps"]}, {"n_cw_instances": sr["n_cw_instances"]})
        cv = sr = gathered = None   # no references to the step's arrays: they die here, as in the warm-up
    sync()
    dt = time.perf_counter() - t0
    cpu_ms_per_step = (time.process_time() - cpu0) / args.steps * 1e3
    if tcpu0 is not None:
        tcpu1 = thread_cpu()
        rows = sorted(((tcpu1[t][1] - tcpu0.get(t, (None, 0.0))[1], tcpu1[t][0], t) for t in tcpu1), reverse=True)
        tot = sum(r[0] for r in rows)
        sys.stderr.write(f"thread CPU over {args.steps} steps: {tot * 1e3 / args.steps:.1f} ms/step in {len(rows)} threads\n")
        for d, comm, t in rows[:40]:
            sys.stderr.write(f"  tid {t} {comm:16s} {d * 1e3 / args.steps:8.2f} ms/step\n")
    thr1 = throttle_stats()
    throttled = None if thr0 is None or thr1 is None else {"periods": thr1[0] - thr0[0], "usec": thr1[1] - thr0[1]}
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([local_bp], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_bp = int(tot.item())
    else:
        total_bp = local_bp

    if rank == 0 and os.environ.get("HS_BENCH_THREAD_CPU"):   # diagnostic: CPU seconds per thread of this process
        rows = []
        for t in os.listdir("/proc/self/task"):
            try:
                f = open(f"/proc/self/task/{t}/stat").read()
                comm = f[f.index("(") + 1:f.rindex(")")]
                rest = f[f.rindex(")") + 2:].split()
                rows.append(((int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"), comm, t))
            except Exception:
                pass
        rows.sort(reverse=True)
        sys.stderr.write("thread cpu_s: " + ", ".join(f"{c}:{u:.2f}" for u, c, t in rows[:30]) + f" (threads={len(rows)})\n")
    if rank == 0:
        K = args.steps
        cv, sr = last
        kernels = {"k_cigar_scan": k_cv[3] / K, "k_pileup": k_cv[0] / K, "k_column_stats": k_cv[1] / K, "k_gather_columns": k_cv[2] / K, "k_column_partition_test": k4 / K,
                   "k_simdiff": k_sr[0] / K, "k_read_graph_rows": k6 / K, "k_chinese_whispers": (k_sr[1] + k_sr[2] + k_sr[3]) / K}
        # algorithmic bytes per launch: pileup = read base in + code out = 2 B / aligned bp;
        # column_stats = 1 B / aligned bp in + 16 B / position out; chinese_whispers: see DESIGN.md
        alg_bytes = {"k_pileup": 2.0 * local_bp, "k_column_stats": 1.0 * local_bp + 16.0 * float(batch.total_len)}
        dom = max(("k_pileup", "k_column_stats"), key=lambda k: kernels[k])
        achieved = alg_bytes[dom] / (kernels[dom] * 1e-3) / 1e9 if kernels[dom] > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                traffic = tj.get(dom)
                if traffic is not None and tj.get("_aligned_bp"):
                    traffic = traffic * local_bp / float(tj["_aligned_bp"])   # measured on a 16-contig batch; streaming kernels scale with bp
            except Exception:
                traffic = None
        out = {
            "metric": "aligned read-bp/sec through call_variants+separate_reads",
            "value": total_bp * K / dt, "unit": "aligned read-bp/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "host": {"cpus_online": os.cpu_count(), "cpus_effective": effective_cores(), "process_cpu_ms_per_step": cpu_ms_per_step,
                     "cfs_throttled_during_timed_steps": throttled},
            "config": {"workload": f"C2 (BASELINE.json configs[1]): 100 kb contig, 2 haplotypes @1% divergence, 50x ONT-error reads; "
                                   f"{B} such contigs per GPU per step, inputs resident in HBM",
                       "contigs_per_gpu": B, "aligned_bp_per_gpu": local_bp, "parallelism": f"contig-sharded x{world}", "groups_per_gpu": G,
                       "host_threads_per_rank": n_threads, "snps_rank0": int(cv["n_snps"]), "cw_instances_rank0": sr["n_cw_instances"]},
            "roofline": {"bound": "hbm", "kernel": ("k_pileup_packed" if dom == "k_pileup" and not os.environ.get("HS_K1_PER_EVENT") else dom), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": kernels[dom],
                         "algorithmic_bytes_per_launch": alg_bytes[dom]},
            "kernel_ms_per_step": kernels, "step_ms": [round(x, 2) for x in step_ms],
            "pipeline_wall_ms_per_step": {k: v / K for k, v in wall.items()},
            "phase_ms_per_step": {"device_phases": t_dev / K, "host_glue": t_host / K, **{"py_" + k: v / K for k, v in py_ms.items()}},
        }
        if world == 1 and args.cpu_contigs > 0:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_contigs, args.seed)
            except Exception as e:  # the baseline is informational; never fail the bench on it
                out["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(out), flush=True)
    batch.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
