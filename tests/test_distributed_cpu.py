"""CPU, world_size 2, gloo: the N>1 path (shard -> trace -> gather to rank 0) with a stand-in trace."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stanford_raytracer_amd import parallel


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 9, 1000003):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            assert max(hi - lo for lo, hi in spans) <= (n + world - 1) // world


def _fake_trace(lo, hi, slots=3):
    idx = torch.arange(lo, hi, dtype=torch.float64)
    rows = idx[:, None, None] + torch.arange(slots * 20, dtype=torch.float64).reshape(1, slots, 20) * 1e-3
    nrows = (idx % 5 + 1).to(torch.int32)
    stop = (idx % 3).to(torch.int32)
    return rows, nrows, stop


def _worker(rank, world, port, nrays, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = parallel.trace_sharded(dist, nrays, _fake_trace, dst=0)
    if rank == 0:
        rows, nrows, stop = out
        er, en, es = _fake_trace(0, nrays)
        q.put(bool(torch.equal(rows, er) and torch.equal(nrows, en) and torch.equal(stop, es)))
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
    for nrays in (11, 8, 1):  # ragged last shard, even split, fewer rays than ranks
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
