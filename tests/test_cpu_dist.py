"""Multi-process path on CPU (gloo, world_size 2): contig sharding, the error-rate exchange and the single
gather of partition labels (hairsplitter_amd/dist.py). The same code runs over RCCL ("nccl") in bench.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hairsplitter_amd import dist as hdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    weights = [5.0, 1.0, 3.0, 3.0, 2.0, 8.0, 1.0]
    shards = hdist.lpt_shards(weights, world)
    mine = shards[rank]
    md = np.array([0.05 + 0.001 * i if i != 1 else 0.0 for i in mine], np.float32)   # contig 1 has no reads
    er = hdist.global_error_rate(mine, md, len(weights))
    labels = np.concatenate([np.full(3 + i, i % 3 - 2, np.int32) for i in mine]) if mine else np.zeros(0, np.int32)
    g = hdist.gather_labels(labels)
    # the preallocated form bench.py uses per step must deliver the same thing, step after step
    lg = hdist.LabelGatherer(hdist.gather_capacity(int(labels.size)))
    for _ in range(2):
        g2 = lg.gather(labels)
        assert (g is None) == (g2 is None)
        if g is not None:
            assert all(np.array_equal(a, b) for a, b in zip(g, g2))
    # window size of stage 4: every rank derives the job-wide choice from its own reads + one all-reduce
    spans = np.array([1500, 2500, 3500], np.int64) if rank == 0 else np.array([2600] * 5, np.int64)
    ws = hdist.global_window_size(spans)
    q.put((rank, mine, er, None if g is None else [x.tolist() for x in g], ws))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_exchange_gloo_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    shards = [r[1] for r in res]
    assert sorted(shards[0] + shards[1]) == list(range(7)) and not set(shards[0]) & set(shards[1])
    # LPT balance: heaviest contig alone-ish, loads within the largest weight of each other
    w = [5.0, 1.0, 3.0, 3.0, 2.0, 8.0, 1.0]
    loads = [sum(w[i] for i in s) for s in shards]
    assert abs(loads[0] - loads[1]) <= 8.0
    # both ranks agree on the global error rate = float32 sum in contig order over contigs with distance > 0
    vals = [np.float32(0.05 + 0.001 * i) for i in range(7) if i != 1]
    tot = np.float32(0)
    for v in vals:
        tot = np.float32(tot + v)
    exp = float(np.float32(tot / np.float32(len(vals))))
    assert res[0][2] == exp and res[1][2] == exp
    # mean READ length of the whole job = (sum + 2 per read) / 8 reads in (2000, 4000), fewer than 20 reads above 4 kb -> 1000
    assert res[0][4] == 1000 and res[1][4] == 1000
    # rank 0 received every rank's labels unchanged; rank 1 nothing
    assert res[1][3] is None
    for r in range(world):
        exp_l = np.concatenate([np.full(3 + i, i % 3 - 2, np.int32) for i in shards[r]]).tolist()
        assert res[0][3][r] == exp_l


def test_lpt_shards_deterministic():
    w = list(np.random.default_rng(0).integers(1, 100, 50))
    a = hdist.lpt_shards(w, 8)
    b = hdist.lpt_shards(w, 8)
    assert a == b and sorted(sum(a, [])) == list(range(50))
    loads = [sum(w[i] for i in s) for s in a]
    assert max(loads) - min(loads) <= max(w)


def test_bench_inputs_are_the_same_from_forked_workers():
    """bench.py generates its shard of the job on forked workers before it touches the GPU (and writes the job's files from
    the workers' parts): same contigs, same files as one by one"""
    import tempfile
    from hairsplitter_amd import synth
    ids = [3, 4, 7, 8, 9, 15, 16, 20, 21]
    with tempfile.TemporaryDirectory() as a, tempfile.TemporaryDirectory() as b:
        pooled, files = synth.generate_job("C4", ids, workers=3, outdir=a)
        serial = [synth.config_contigs("C4", first=i, count=1)[0] for i in ids]
        for c, d in zip(pooled, serial):
            assert c.name == d.name and np.array_equal(c.seq, d.seq) and len(c.reads) == len(d.reads)
            assert all(np.array_equal(x, y) for x, y in zip(c.reads, d.reads))
            assert all(x.pos == y.pos and x.strand == y.strand and np.array_equal(x.cigar, y.cigar) for x, y in zip(c.alns, d.alns))
        f2 = synth.write_files(serial, b)
        for k in f2:
            assert open(files[k], "rb").read() == open(f2[k], "rb").read()
    # what the sharding knows up front (lengths) is what the generator produces
    shapes = synth.config_shapes("C4", count=22)
    assert [shapes[i][0] for i in ids] == [len(c.seq) for c in serial]


def test_strong_scaling_shards_cover_the_job():
    """bench.py --gpus N: the SAME job, contigs assigned by LPT on (length x depth)"""
    from hairsplitter_amd import synth
    shapes = synth.config_shapes("C4")
    for world in (1, 2, 4, 8):
        shards = hdist.lpt_shards([float(s[0] * s[2]) for s in shapes], world)
        assert sorted(sum(shards, [])) == list(range(500))
        loads = [sum(shapes[i][0] for i in s) for s in shards]
        assert max(loads) - min(loads) <= 300_000


# ---------------------------------------------------------------------------------------------
# bench.py's step logic across two ranks, on the CPU: the job's contigs sharded by LPT, every rank runs both stages on ITS contigs
# (the product's host glue with the oracle-backed device interface: tests/harness), the per-contig distances cross the ranks for
# the job's error rate, the windows' (read, label) lists are gathered to rank 0 in ONE collective -- and what rank 0 then holds is
# what a single process computes for the whole job.
# ---------------------------------------------------------------------------------------------
def _gro_lists(path):
    """per contig name: [(start, end, ids, labels)] of its GROUP lines"""
    out, name = {}, None
    for l in open(path):
        t = l.rstrip("\n").split("\t")
        if t[0] == "CONTIG":
            name = t[1]; out[name] = []
        elif t[0] == "GROUP":
            ids = [int(x) for x in t[3].split(",") if x != ""]
            lab = [int(x) for x in t[4].split(",") if x != ""]
            out[name].append((int(t[1]), int(t[2]), ids, lab))
    return out


def _rank_job(rank, world, port, td, names, q):
    import subprocess
    import __graft_entry__ as ge
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = ge.paths()
    mine = names[rank]
    d = os.path.join(td, "rank%d" % rank)
    # stage 3 on this rank's contigs; its per-contig distances are what crosses the ranks
    col, vcf, err, gro = (os.path.join(d, x) for x in ("o.col", "o.vcf", "o.err", "o.gro"))
    subprocess.run([p["harness"], "call_variants", os.path.join(d, "assembly.gfa"), os.path.join(d, "reads.fasta"), os.path.join(d, "aln.sam"), "1", d, err, "0", "0", col, vcf, "0.33"],
                   check=True, stdout=subprocess.DEVNULL, env=dict(os.environ, HS_HARNESS_DUMP_MD=os.path.join(d, "md.txt")))
    md = np.array([float.fromhex(x) for x in open(os.path.join(d, "md.txt")).read().split()], np.float32)
    er = hdist.global_error_rate(mine["ids"], md, mine["n_total"])
    e = min(float("%g" % er), 0.15)
    ws = hdist.global_window_size(np.array(mine["refspans"], np.int64))
    env = dict(os.environ, HS_HARNESS_WINDOW_SIZE=str(ws))
    subprocess.run([p["harness"], "separate_reads", col, "1", str(e), os.path.join(d, "none"), "0", "0.01", "0", gro, "0"], check=True, stdout=subprocess.DEVNULL, env=env)
    lists = _gro_lists(gro)
    off, ids, lab = [0], [], []
    order = []
    for name in sorted(lists):
        for (a, b, i, l) in lists[name]:
            order.append((name, a, b)); ids += i; lab += l; off.append(len(ids))
    off, ids, lab = np.array(off, np.int64), np.array(ids, np.int32), np.array(lab, np.int32)
    g = hdist.SparseLabelGatherer(hdist.SparseLabelGatherer.job_capacity(hdist.sparse_payload_bytes(off.size - 1, ids.size)))
    got = None
    for _ in range(2):      # (step after step through the same buffers)
        got = g.gather(off, ids, lab)
    all_orders = [None] * world
    dist.all_gather_object(all_orders, order)
    if rank == 0:
        merged = {}
        for r in range(world):
            o, i, l = got[r]
            for w, (name, a, b) in enumerate(all_orders[r]):
                merged.setdefault(name, []).append((a, b, i[o[w]:o[w + 1]].tolist(), l[o[w]:o[w + 1]].tolist()))
        q.put((e, ws, merged))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gather_what_one_process_computes():
    import subprocess
    import tempfile
    import __graft_entry__ as ge
    from hairsplitter_amd import synth
    ge.build()
    p = ge.paths()
    contigs = [synth.make_contig(77, i, L, h, 0.012, 28, "ont") for i, (L, h) in enumerate([(9000, 2), (7000, 3), (6000, 1), (8000, 2), (5000, 2)])]
    world = 2
    shards = hdist.lpt_shards([float(len(c.seq)) for c in contigs], world)
    with tempfile.TemporaryDirectory() as td:
        full = synth.write_files(contigs, os.path.join(td, "full"))
        col, vcf, err, gro = (os.path.join(td, "full", x) for x in ("o.col", "o.vcf", "o.err", "o.gro"))
        subprocess.run([p["harness"], "call_variants", full["gfa"], full["reads"], full["sam"], "1", td, err, "0", "0", col, vcf, "0.33"], check=True, stdout=subprocess.DEVNULL)
        e_full = min(float("%g" % float(open(err).read().strip())), 0.15)
        subprocess.run([p["harness"], "separate_reads", col, "1", str(e_full), os.path.join(td, "none"), "0", "0.01", "0", gro, "0"], check=True, stdout=subprocess.DEVNULL)
        want = _gro_lists(gro)
        names = []
        for r in range(world):
            d = os.path.join(td, "rank%d" % r)
            mine = [contigs[i] for i in shards[r]]
            synth.write_files(mine, d)
            spans = [int(sum(int(x) >> 4 for x in a.cigar if (int(x) & 15) in (0, 2, 7, 8))) for c in mine for a in c.alns]
            names.append({"ids": shards[r], "n_total": len(contigs), "refspans": spans})
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_rank_job, args=(r, world, port, td, names, q)) for r in range(world)]
        for pr in procs:
            pr.start()
        e, ws, merged = q.get(timeout=300)
        for pr in procs:
            pr.join(120)
            assert pr.exitcode == 0
        assert e == e_full                     # the job's error rate from the ranks' distances == the single process's
        assert merged == want                  # and every window's reads and labels as one process computes them


def test_sparse_digest_does_not_depend_on_how_the_windows_are_sharded():
    """hdist.sparse_digest: the same windows give the same digest in one payload, split over ranks in any order, and a changed label changes it"""
    import numpy as np
    from hairsplitter_amd import dist as hdist
    rng = np.random.default_rng(3)
    wins = []
    for _ in range(200):
        n = int(rng.integers(0, 40))
        wins.append((np.sort(rng.choice(500, size=n, replace=False)).astype(np.int32), rng.integers(-1, 6, size=n).astype(np.int32)))

    def payload(ws):
        off = np.zeros(len(ws) + 1, np.int64)
        off[1:] = np.cumsum([len(w[0]) for w in ws])
        ids = np.concatenate([w[0] for w in ws]) if ws else np.zeros(0, np.int32)
        lab = np.concatenate([w[1] for w in ws]) if ws else np.zeros(0, np.int32)
        return off, ids, lab

    one = hdist.sparse_digest([payload(wins)])
    order = rng.permutation(len(wins))
    parts = [[wins[i] for i in order[k::3]] for k in range(3)]
    three = hdist.sparse_digest([payload(p) for p in parts])
    assert one == three and one["windows"] == 200 and one["entries"] == sum(len(w[0]) for w in wins)
    # round trip through the gather's encoding (labels travel as int16)
    enc = []
    for p in parts:
        off, ids, lab = payload(p)
        buf = np.zeros(hdist.sparse_payload_bytes(len(off) - 1, len(ids)) + 64, np.uint8)
        n = hdist.encode_sparse(off, ids, lab, buf)
        enc.append(hdist.decode_sparse(buf[:n]))
    assert hdist.sparse_digest(enc) == one
    changed = [(w[0], w[1].copy()) for w in wins]
    k = next(i for i, w in enumerate(changed) if len(w[0]))
    changed[k][1][0] += 1
    assert hdist.sparse_digest([payload(changed)]) != one


def test_sparse_digest_with_contigs_notices_windows_that_changed_places():
    """With the contig of every window in the payload (what bench.py gathers) the digest still does not depend on the sharding, survives the
    gather's encoding, and two windows of different contigs that swap their contents change it (the plain sum of CRCs would not)"""
    import numpy as np
    from hairsplitter_amd import dist as hdist
    rng = np.random.default_rng(5)
    wins = []
    for k in range(120):
        n = int(rng.integers(1, 30))
        wins.append((np.sort(rng.choice(300, size=n, replace=False)).astype(np.int32), rng.integers(-1, 5, size=n).astype(np.int32), k // 4))

    def payload(ws):
        off = np.zeros(len(ws) + 1, np.int64)
        off[1:] = np.cumsum([len(w[0]) for w in ws])
        return off, np.concatenate([w[0] for w in ws]), np.concatenate([w[1] for w in ws]), np.array([w[2] for w in ws], np.int32)

    one = hdist.sparse_digest([payload(wins)])
    order = rng.permutation(len(wins))
    parts = [[wins[i] for i in order[k::2]] for k in range(2)]
    assert hdist.sparse_digest([payload(p) for p in parts]) == one
    enc = []
    for p in parts:
        off, ids, lab, wc = payload(p)
        buf = np.zeros(hdist.sparse_payload_bytes(len(off) - 1, len(ids), with_contigs=True) + 64, np.uint8)
        n = hdist.encode_sparse(off, ids, lab, buf, wc)
        assert n == hdist.sparse_payload_bytes(len(off) - 1, len(ids), with_contigs=True)
        dec = hdist.decode_sparse(buf[:n])
        assert len(dec) == 4 and (dec[3] == wc).all()
        enc.append(dec)
    assert hdist.sparse_digest(enc) == one
    swapped = list(wins)
    a, b = 0, 117      # windows of contig 0 and contig 29
    swapped[a], swapped[b] = (wins[b][0], wins[b][1], wins[a][2]), (wins[a][0], wins[a][1], wins[b][2])
    assert hdist.sparse_digest([payload(swapped)]) != one
    assert hdist.sparse_digest([payload(swapped)[:3]]) == hdist.sparse_digest([payload(wins)[:3]])      # (without the contigs the swap goes unnoticed)
