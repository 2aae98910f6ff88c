"""bench.py itself under the launchers the driver uses, on a small job: the single-process line (hs_pipeline_run_fused) and one rank under
torch.distributed.run with the nccl (= RCCL) backend -- the two-call pipeline, the gloo exchanges of the error rate / window size and the
RCCL gather of the labels all execute, with one rank. Both lines must carry the parity gate's verdict against the compiled reference."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert lines, out.decode()[-2000:]
    return json.loads(lines[-1])


def _check(d, pipeline):
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0
    assert d["config"]["pipeline"] == pipeline
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    p = d["parity"]
    assert p["checked"] and p["identical"], p
    assert p["gro_identical"] and p["col_snps_identical"] and p["error_rate_identical"] and p["col_entries_identical"]
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["value"] > 0


def test_bench_single_process_line_with_parity_gate(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--config", "C2", "--contigs", "6", "--f2f-runs", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    _check(_line(r.stdout), "hs_pipeline_run_fused")


def test_bench_one_rank_under_torch_distributed_run_with_rccl(built):
    """world size 1, backend nccl: what the driver launches for N > 1, with one rank (this box has one GPU)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--config", "C2", "--contigs", "6", "--f2f-runs", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = _line(r.stdout)
    _check(d, "hs_pipeline_select + hs_pipeline_run")
    assert d["phase_ms_per_step"]["py_gather"] > 0      # the collective ran


def test_bench_two_ranks_hold_what_one_process_holds(built):
    """N = 2 as the driver launches it (torch.distributed.run, one rank per GPU) on a box with one GPU: HS_BENCH_SHARE_DEVICE=1 puts both
    ranks on device 0 and every exchange on gloo. The job is sharded by LPT, each rank runs its own pipeline, the error rate and window
    size are formed across the ranks, rank 0 gathers the labels: its digest of the whole job's windows equals the single process's."""
    common = ["--steps", "2", "--warmup", "1", "--config", "C2", "--contigs", "6", "--cpu-contigs", "0"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, timeout=900)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    d1 = _line(one.stdout)
    env = dict(os.environ, HS_BENCH_SHARE_DEVICE="1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, env=env, timeout=900)
    assert two.returncode == 0, two.stderr.decode()[-3000:]
    d2 = _line(two.stdout)
    assert d2["n_gpus"] == 2 and d2["scaling"] == "strong" and d2["value"] > 0
    g1, g2 = d1["labels_digest"], d2["labels_digest"]
    assert g1["ranks"] == 1 and g2["ranks"] == 2
    assert g1["windows"] > 0 and g1["entries"] > 0
    assert {k: g1[k] for k in ("windows", "entries", "sum_crc32")} == {k: g2[k] for k in ("windows", "entries", "sum_crc32")}
    # neither run had the reference beside it (--cpu-contigs 0): both are checked against the known answer of this job (tests/golden/bench_labels_digest.json,
    # the digest of a single-process run whose reference gate passed: tools/make_bench_digests.py)
    for d in (d1, d2):
        assert d["parity"]["checked"] and d["parity"]["kind"] == "labels digest" and d["parity"]["identical"], d["parity"]


def test_bench_default_job_against_its_known_answer(built):
    """The default job (C4) without the file-to-file leg: the labels of the timed path against the digest kept from the run that passed the reference gate"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-contigs", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = _line(r.stdout)
    assert d["config"]["contigs"] == 500 and d["labels_digest"]["windows"] == 25594
    assert d["parity"]["checked"] and d["parity"]["kind"] == "labels digest" and d["parity"]["identical"], d["parity"]


def test_bench_default_job_with_loop_a_on_the_device(built):
    """The same job with loop A of keep_only_robust_variants walked by k_loop_a for every contig (HS_LOOP_A_ON_DEVICE=1: 500 chains of up to
    3 300 candidate columns, one wavefront each; the host imports the partitions for loop B): the timed path's labels against the same digest"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-contigs", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT,
                       env=dict(os.environ, HS_LOOP_A_ON_DEVICE="1"), timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = _line(r.stdout)
    assert d["kernels"]["k_loop_a"]["launches_per_step"] > 0
    assert d["parity"]["checked"] and d["parity"]["kind"] == "labels digest" and d["parity"]["identical"], d["parity"]
