"""CPU, world_size 2, gloo: the N>1 path -- shard -> trace -> pack -> variable-length gather to rank 0 -- through
parallel.trace_sharded, the one implementation bench.py --gpus N and tests/test_gpu_multirank.py also run.  Here
the trace is a stand-in with ragged row counts and the packer is the plain-torch reference (the HIP packer needs a
GPU); everything else is the product code."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stanford_raytracer_amd import parallel

SLOTS, OUTPUTPER = 4, 3


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 9, 1000003):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            assert max(hi - lo for lo, hi in spans) <= (n + world - 1) // world


def _fake_trace(lo, hi):
    idx = torch.arange(lo, hi, dtype=torch.float64)
    rows = idx[:, None, None] + torch.arange(SLOTS * 20, dtype=torch.float64).reshape(1, SLOTS, 20) * 1e-3
    nrows = (idx % (SLOTS * OUTPUTPER) + 1).to(torch.int32)  # 1 .. maxsteps rows: 1 .. SLOTS kept
    nrows[(idx % 7) == 3] = 0  # a ray that produced nothing keeps nothing
    stop = (idx % 3).to(torch.int32)
    return rows, nrows, stop


def _pack(rows, nrows):
    return parallel.pack_rows_torch(rows, nrows, OUTPUTPER)


def test_pack_unpack_roundtrip():
    rows, nrows, _ = _fake_trace(0, 50)
    packed, off = _pack(rows, nrows)
    kept = parallel.kept_rows(nrows.to(torch.int64), OUTPUTPER, SLOTS)
    assert off[-1] == kept.sum() == packed.shape[0]
    back = parallel.unpack_rows(packed, nrows, OUTPUTPER, SLOTS)
    mask = torch.arange(SLOTS)[None, :] < kept[:, None]
    assert torch.equal(back[mask], rows[mask]) and float(back[~mask].abs().sum()) == 0.0
    # ray i's rows sit at offsets[i] .. offsets[i+1]
    for i in (0, 3, 17, 49):
        assert torch.equal(packed[off[i]:off[i + 1]], rows[i, :kept[i]])


def _worker(rank, world, port, nrays, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tm = {}
    out = parallel.trace_sharded(dist, nrays, _fake_trace, _pack, dst=0, timings=tm)
    assert tm["shard"] == parallel.shard_bounds(nrays, rank, world)
    if rank == 0:
        packed, nrows, stop = out
        er, en, es = _fake_trace(0, nrays)
        ep, _ = _pack(er, en)
        lo, hi = parallel.shard_bounds(nrays, 0, world)
        own = int(parallel.kept_rows(en[lo:hi].to(torch.int64), OUTPUTPER, SLOTS).sum())
        expect_bytes = (ep.shape[0] - own) * 160 + (nrays - (hi - lo)) * 8
        q.put(bool(torch.equal(packed, ep) and torch.equal(nrows, en) and torch.equal(stop, es)
                   and tm["gather_bytes"] == expect_bytes))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gather_world2_gloo():
    for nrays in (11, 8, 1):  # ragged last shard, even split, fewer rays than ranks (rank 1 sends nothing)
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, nrays, q)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert q.get(timeout=10) is True


def _worker_pipelined(rank, world, port, nrays, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = parallel.shard_bounds(nrays, rank, world)

    def launch(k):  # step k produces different rows (k added), so a mixed-up step would show
        rows, nrows, stop = _fake_trace(lo, hi)
        return rows + 1000.0 * k, nrows, stop

    outs = parallel.trace_sharded_pipelined(dist, nrays, 3, launch, lambda rows, nrows, slot: _pack(rows, nrows), dst=0,
                                            keep_last_only=False)
    if rank == 0:
        er, en, es = _fake_trace(0, nrays)
        ok = len(outs) == 3
        for k, (packed, nrows, stop) in enumerate(outs):
            ep, _ = _pack(er + 1000.0 * k, en)
            ok = ok and bool(torch.equal(packed, ep) and torch.equal(nrows, en) and torch.equal(stop, es))
        q.put(ok)
    else:
        assert outs is None
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_gather_world2_gloo():
    """trace_sharded_pipelined (gather of step k behind the trace of step k + 1): every step's rows arrive, in step order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipelined, args=(r, 2, port, 11, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def test_gather_world8_gloo():
    """The shape the driver's N = 8 run takes: eight ranks, a launch set that does not divide (shards of 13 and 12 rays), and
    fewer rays than ranks (five ranks with empty shards send nothing and still take part in the counts exchange)."""
    for nrays in (101, 3):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 8, port, nrays, q)) for r in range(8)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        assert q.get(timeout=10) is True
