"""The N>1 path with the REAL kernel: 2 ranks (gloo rendezvous, both on cuda:0 -- the GPU box has one card and RCCL
refuses two ranks on one device) each trace their contiguous shard of one launch set with srt_trace_batch_device,
pack the kept rows with srt_pack_rows_device and gather them to rank 0 through parallel.trace_sharded -- the code
bench.py --gpus N runs.  The gathered rows must be bit-identical to a single-rank trace of the whole set
(rays are independent: raytracer_driver.f95:1144-1232 is a serial loop with no carried state)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

NRAYS, GRID, OUTPUTPER, MAXSTEPS = 3001, 24, 4, 96  # ragged: shards of 1501 and 1500 rays


def _setup(lo, hi):
    import torch

    from stanford_raytracer_amd import api, workloads as wl
    from stanford_raytracer_amd.device_batch import DeviceBatch

    api.init(0)
    dev = torch.device("cuda", 0)
    F, b = wl.make_grid(GRID, half_width=10.0 * wl.R_E)
    model = api.Model.interp(F, b, wl.QS, wl.MS)
    p = api.make_params(fixedstep=0, dt0=1e-3, dtmax=0.1, maxerr=5e-4, tmax=0.5, root=2, minalt=wl.MINALT,
                        maxsteps=MAXSTEPS, outputper=OUTPUTPER, del_=1e-6, ray_order=1)
    pos0, dir0, w0 = wl.launch_set(NRAYS, 4)
    return model, p, DeviceBatch(model, p, pos0[lo:hi], dir0[lo:hi], w0[lo:hi], dev)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from stanford_raytracer_amd import parallel

    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = parallel.shard_bounds(NRAYS, rank, world)
    model, p, batch = _setup(lo, hi)
    tm = {}
    out = parallel.trace_sharded(dist, NRAYS, lambda a, b: batch.trace(),
                                 lambda rows, nrows: parallel.pack_rows_device(rows, nrows, OUTPUTPER), dst=0, timings=tm)
    if rank == 0:
        packed, nrows, stop = out
        q.put((packed.cpu().numpy(), nrows.cpu().numpy(), stop.cpu().numpy(), tm["gather_bytes"]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gather_is_bit_identical_to_one_rank():
    import torch
    import torch.multiprocessing as mp

    from stanford_raytracer_amd import parallel

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    packed2, nrows2, stop2, nbytes = q.get(timeout=300)
    for pr in procs:
        pr.join(120)
        assert pr.exitcode == 0
    # single rank, whole set, same library
    model, p, batch = _setup(0, NRAYS)
    rows, nrows, stop = batch.trace()
    torch.cuda.synchronize()
    buf, off = parallel.pack_rows_device(rows, nrows, OUTPUTPER)
    assert buf.shape[0] == NRAYS * batch.slots  # the reused worst-case buffer; its valid rows are the first off[-1]
    packed1 = buf[:int(off[-1])]
    pt, offt = parallel.pack_rows_torch(rows, nrows, OUTPUTPER)  # the HIP packer against the plain-torch one
    assert torch.equal(off, offt) and torch.equal(packed1, pt)
    assert np.array_equal(nrows2, nrows.cpu().numpy()) and np.array_equal(stop2, stop.cpu().numpy())
    assert packed2.shape == tuple(packed1.shape)
    assert np.array_equal(packed2.view(np.uint64), packed1.cpu().numpy().view(np.uint64)), "gathered rows differ bitwise"
    lo, hi = parallel.shard_bounds(NRAYS, 0, 2)
    assert nbytes == int(off[-1] - off[hi]) * 160 + (NRAYS - hi) * 8
    assert int(off[-1]) < NRAYS * batch.slots  # the gather moved fewer rows than the padded buffer holds
    assert nrows.max() > 1


def test_pack_rows_edge_cases():
    """srt_pack_rows_device: zero rays, zero kept rows, a capacity that is too small."""
    import ctypes as C

    import torch

    from stanford_raytracer_amd import api, parallel

    api.init(0)
    dev = torch.device("cuda", 0)
    slots, per = 5, 3
    nrows = torch.tensor([0, 1, 3, 4, 15, 16, 200], dtype=torch.int32, device=dev)  # 200: clamped to the slots
    rows = torch.arange(7 * slots * 20, dtype=torch.float64, device=dev).reshape(7, slots, 20)
    packed, off = parallel.pack_rows_device(rows, nrows, per)
    assert off.cpu().tolist() == [0, 0, 1, 2, 4, 9, 14, 19]
    pt, offt = parallel.pack_rows_torch(rows, nrows, per)
    assert torch.equal(off, offt) and torch.equal(packed[:19], pt)
    fresh, _ = parallel.pack_rows_device(rows, nrows, per)
    assert fresh.data_ptr() != packed.data_ptr()  # default: a buffer of the caller's own per call
    first, _ = parallel.pack_rows_device(rows, nrows, per, slot=0)
    again, _ = parallel.pack_rows_device(rows, nrows, per, slot=0)
    assert again.data_ptr() == first.data_ptr()  # opt-in: one buffer per (device, size, slot), reused from step to step
    other, _ = parallel.pack_rows_device(rows, nrows, per, slot=1)
    assert other.data_ptr() != first.data_ptr() and torch.equal(other[:19], pt)
    parallel.release_pack_buffers()
    assert not parallel._PACK_BUFFERS
    e = torch.zeros((0, slots, 20), dtype=torch.float64, device=dev)
    packed, off = parallel.pack_rows_device(e, torch.zeros(0, dtype=torch.int32, device=dev), per)
    assert off.cpu().tolist() == [0]
    # capacity 3 rows: rays 0..2 fit, the total still reports 19
    offs = torch.empty(8, dtype=torch.int64, device=dev)
    small = torch.full((3, 20), -1.0, dtype=torch.float64, device=dev)
    api._check(api.lib().srt_pack_rows_device(slots, per, 7, rows.data_ptr(), nrows.data_ptr(), offs.data_ptr(),
                                              small.data_ptr(), 3, None))
    torch.cuda.synchronize()
    assert int(offs[7]) == 19 and torch.equal(small[:2], pt[:2]) and float(small[2, 0]) == -1.0


def test_pack_rows_rejects_host_pointers_and_keeps_the_callers_device():
    """srt_pack_rows_device works on the device that OWNS the buffers, whatever device the calling thread is bound to, and
    leaves that binding alone; pointers that are not device memory are refused (round-2 advice)."""
    import torch

    from stanford_raytracer_amd import api

    api.init(0)
    dev = torch.device("cuda", 0)
    slots, per = 2, 1
    nrows = torch.tensor([1, 2], dtype=torch.int32, device=dev)
    rows = torch.ones((2, slots, 20), dtype=torch.float64, device=dev)
    offs = torch.empty(3, dtype=torch.int64, device=dev)
    out = torch.empty((4, 20), dtype=torch.float64, device=dev)
    before = torch.cuda.current_device()
    api._check(api.lib().srt_pack_rows_device(slots, per, 2, rows.data_ptr(), nrows.data_ptr(), offs.data_ptr(), out.data_ptr(), 4, None))
    torch.cuda.synchronize()
    assert offs.cpu().tolist() == [0, 1, 3] and torch.cuda.current_device() == before
    host = torch.empty(3, dtype=torch.int64)  # a host buffer where device memory is required
    rc = api.lib().srt_pack_rows_device(slots, per, 2, rows.data_ptr(), nrows.data_ptr(), host.data_ptr(), out.data_ptr(), 4, None)
    assert rc == api.SRT_EINVAL and b"not device memory" in api.lib().srt_last_error()


def _nccl_world1(port, q, pipelined):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from stanford_raytracer_amd import parallel

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    model, p, batch = _setup(0, NRAYS)
    if pipelined:
        from stanford_raytracer_amd.device_batch import DeviceBatch
        from stanford_raytracer_amd import workloads as wl

        pos0, dir0, w0 = wl.launch_set(NRAYS, 4)
        batch = DeviceBatch(model, p, pos0, dir0, w0, torch.device("cuda", 0), nbuf=2)
        out = parallel.trace_sharded_pipelined(dist, NRAYS, 3, lambda k: batch.trace(k % 2),
                                               lambda rows, nrows, slot: parallel.pack_rows_device(rows, nrows, OUTPUTPER, slot=slot))
        tm = {"gather_bytes": 0}
    else:
        tm = {}
        out = parallel.trace_sharded(dist, NRAYS, lambda a, b: batch.trace(),
                                     lambda rows, nrows: parallel.pack_rows_device(rows, nrows, OUTPUTPER), dst=0, timings=tm)
    packed, nrows, stop = out
    assert packed.is_cuda and nrows.is_cuda  # the device-tensor branch: nothing was staged through the host
    q.put((packed.cpu().numpy(), nrows.cpu().numpy(), stop.cpu().numpy(), tm["gather_bytes"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("pipelined", [False, True])
def test_rccl_branch_world_size_one(pipelined):
    """Backend "nccl" (= RCCL) with ONE rank: parallel.trace_sharded / trace_sharded_pipelined end to end through the branch
    an 8-GPU run takes -- counts all_gather on device int64 tensors, device-resident packed rows, no host staging -- before
    any multi-GPU run exists (RCCL refuses two ranks on one device, so the 2-rank test above has to use gloo)."""
    import torch
    import torch.multiprocessing as mp

    from stanford_raytracer_amd import parallel

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_nccl_world1, args=(_free_port(), q, pipelined))
    pr.start()
    packed2, nrows2, stop2, nbytes = q.get(timeout=300)
    pr.join(120)
    assert pr.exitcode == 0
    model, p, batch = _setup(0, NRAYS)
    rows, nrows, stop = batch.trace()
    torch.cuda.synchronize()
    pt, off = parallel.pack_rows_torch(rows, nrows, OUTPUTPER)
    assert nbytes == 0  # nothing travels to the only rank
    assert np.array_equal(nrows2, nrows.cpu().numpy()) and np.array_equal(stop2, stop.cpu().numpy())
    assert np.array_equal(packed2.view(np.uint64), pt.cpu().numpy().view(np.uint64))


def _run_bench(nranks, port, rays=6007):
    import json
    import subprocess

    args = ["--gpus", str(nranks), "--steps", "2", "--warmup", "1", "--workload", "interp4m", "--rays", str(rays), "--grid", "24",
            "--maxsteps", "96", "--traffic", "off", "--other-configs", "0", "--cpu-seconds", "0", "--damping-rays", "0"]
    cmd = [sys.executable]
    if nranks > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    env = dict(os.environ, SRT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line, the other ranks none: %r" % r.stdout[-500:]
    return json.loads(lines[0])


def test_three_rank_rehearsal_through_bench_py():
    """`bench.py --gpus N` as the driver launches it (python -m torch.distributed.run, one process per rank, rendezvous on
    127.0.0.1), rehearsed on the one card: 3 ranks (the box's process guard kills a run with more than 6 processes on the GPU:
    this test runner and the launcher's agent are two of them -- a 5-rank version of this test was killed at 7) with the gloo
    backend standing in for RCCL (SRT_BENCH_BACKEND=gloo folds the ranks onto cuda:0; RCCL itself refuses two ranks on one
    device), a tiny grid, a launch set that does not divide (6007 rays: shards of 2003, 2002, 2002).
    Through bench.py's own code: parallel.trace_sharded (counts all_gather, grouped send / recv of the packed rows to rank 0),
    the barrier-bracketed timed region, the max-over-ranks reduction, the pipelined pass.  The printed line must name 3 GPUs,
    carry the multi_gpu fields, and its rows_checksum -- a position-weighted sum over the bits of every gathered row, row count
    and stop code -- must equal the one-GPU run's of the same launch set."""
    one = _run_bench(1, 0)
    many = _run_bench(3, _free_port())
    assert one["n_gpus"] == 1 and many["n_gpus"] == 3 and many["scaling"] == "strong"
    assert many["config"]["total_rays"] == 6007 and many["config"]["rays_per_gpu"] == 2003
    mg = many["multi_gpu"]
    for key in ("gather_ms", "pack_ms", "kernel_ms_max_over_ranks", "gather_bytes_into_rank0", "gather_GBs", "pipelined"):
        assert key in mg, key
    assert mg["gather_ms"] > 0 and mg["pack_ms"] > 0 and mg["kernel_ms_max_over_ranks"] > 0
    assert mg["pipelined"]["ms_per_step"] is not None, mg["pipelined"]
    acc1, acc5 = one["detail"]["accepted_steps_per_launch_all_ranks"], many["detail"]["accepted_steps_per_launch_all_ranks"]
    assert acc1 == acc5 and acc1 > 100000
    assert one["detail"]["packed_rows"] == many["detail"]["packed_rows"]
    assert one["detail"]["rows_checksum"] == many["detail"]["rows_checksum"]
    # rank 0 received every other shard's packed rows (160 B each) + nrows and stop codes (8 B per ray)
    rows0 = mg["gather_bytes_into_rank0"]
    assert 0 < rows0 < many["detail"]["packed_rows"] * 160 + 6007 * 8
    assert many["value"] > 0 and many["steps"] == 2 and many["warmup"] == 1
