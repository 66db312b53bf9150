"""Multi-GPU layer: rays are independent, so the launch set is cut into contiguous shards (one per rank,
one rank per GPU, one model replica per rank) and the only communication is the final gather of the
trajectory rows to rank 0 (RCCL over xGMI with backend "nccl"; "gloo" on CPU for tests).

The reference has no parallel mode at all (serial `do` over rays, raytracer_driver.f95:1144-1232);
shards reproduce exactly what running it on disjoint ray files would.

The gather is variable-length: a ray keeps (nrows-1)/outputper + 1 of its slots, so every
rank packs its kept rows back to back (srt_pack_rows_device on the GPU), the ranks exchange their counts
(one all_gather of two int64 per rank), and every peer sends exactly its packed rows, nrows and stop codes
to the root in ONE group of point-to-point transfers (RCCL has no gatherv; 7 concurrent transfers use the 7
xGMI links into the root).  bench.py --gpus N, the 2-rank GPU test and the gloo CPU test all run
`trace_sharded` below -- there is no second implementation.
"""
import time

import numpy as np

ROW = 20


def shard_bounds(nrays, rank, world):
    """Contiguous block of ceil(n/world) rays for this rank (last ranks may be short or empty)."""
    per = (nrays + world - 1) // world
    lo = min(rank * per, nrays)
    hi = min(lo + per, nrays)
    return lo, hi


def kept_rows(nrows, outputper, slots):
    """Rows of a ray that exist in its slots: 0, outputper, 2*outputper, .. < nrows (torch or numpy)."""
    k = (nrows - 1) // outputper + 1
    k = k * (nrows > 0)
    return k.clip(0, slots) if isinstance(k, np.ndarray) else k.clamp(0, slots)


def pack_rows_torch(rows, nrows, outputper):
    """Reference packer in plain torch (what srt_pack_rows_device does on the GPU): rows[n,slots,20], nrows[n]
    -> (packed[total,20], offsets[n+1] int64).  Used by the CPU tests and to cross-check the HIP packer."""
    import torch

    n, slots = rows.shape[0], rows.shape[1]
    kept = kept_rows(nrows.to(torch.int64), outputper, slots)
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=rows.device)
    if n:
        offsets[1:] = torch.cumsum(kept, 0)
        mask = torch.arange(slots, device=rows.device)[None, :] < kept[:, None]
        packed = rows[mask]
    else:
        packed = rows.reshape(0, ROW)
    return packed.contiguous(), offsets


_PACK_BUFFERS = {}  # (device, rows capacity, slot) -> worst-case packed buffer, reused from step to step (opt-in: slot=...)


def release_pack_buffers(device=None):
    """Drop the reused pack buffers (of one device, or all): 2.7 GB per slot at BASELINE config[2]."""
    for k in [k for k in _PACK_BUFFERS if device is None or k[0] == device]:
        del _PACK_BUFFERS[k]


def pack_rows_device(rows, nrows, outputper, stream=None, slot=None):
    """srt_pack_rows_device on torch CUDA tensors -> (packed[n * slots, 20], offsets[n + 1]).  `packed` is a worst-case
    buffer; its valid rows are the first offsets[n] -- a number that stays ON THE DEVICE: nothing here waits for the GPU
    (the gather reads it together with the other ranks' counts).
    slot=None (default): a fresh buffer per call, the caller's to keep.  slot=k: ONE buffer per (device, size, k) that is
    reused from call to call -- what a step loop wants (no allocation per step); a `packed` kept from an earlier call with
    the same k is overwritten by the next.  Steps that are in flight together use different k (trace_sharded_pipelined);
    release_pack_buffers() frees them."""
    import torch

    from . import api

    n, slots = rows.shape[0], rows.shape[1]
    cap = max(n * slots, 1)
    if slot is None:
        packed = torch.empty((cap, ROW), dtype=torch.float64, device=rows.device)
    else:
        key = (rows.device, cap, slot)
        packed = _PACK_BUFFERS.get(key)
        if packed is None:
            for k in [k for k in _PACK_BUFFERS if k[0] == rows.device and k[2] == slot]:
                del _PACK_BUFFERS[k]  # (a launch set of another size on this device: one buffer per slot is kept)
            packed = _PACK_BUFFERS[key] = torch.empty((cap, ROW), dtype=torch.float64, device=rows.device)
    offsets = torch.empty(n + 1, dtype=torch.int64, device=rows.device)
    st = stream if stream is not None else torch.cuda.current_stream(rows.device)
    api._check(api.lib().srt_pack_rows_device(slots, outputper, n, rows.data_ptr(), nrows.data_ptr(), offsets.data_ptr(),
                                              packed.data_ptr(), n * slots, st.cuda_stream))
    return packed, offsets


def _sync(t):
    if t.is_cuda:
        import torch

        torch.cuda.synchronize(t.device)


def gather_packed(dist, packed, nrows, stop, nrays, dst=0, total=None):
    """Variable-length gather to `dst`: every rank contributes packed[total_r,20], nrows[n_r], stop[n_r] of its
    contiguous shard.  `total` (optional): one-element int64 tensor on packed's device holding the number of valid rows of
    `packed` (a worst-case buffer, pack_rows_device); None = all of packed.  The counts exchange is the one place where
    the host waits for the device.  Returns (packed_all[total,20], nrows_all[nrays], stop_all[nrays], bytes_received) on
    dst, None elsewhere.  gloo cannot move device memory: there the buffers are staged through the host."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    dev = packed.device
    stage = dist.get_backend() == "gloo" and packed.is_cuda
    cdev = torch.device("cpu") if stage else dev
    # 1. counts: [rays, kept rows] of every rank
    if total is None:
        mine = torch.tensor([nrows.shape[0], packed.shape[0]], dtype=torch.int64, device=cdev)
    else:
        mine = torch.cat([torch.tensor([nrows.shape[0]], dtype=torch.int64, device=dev), total.reshape(1).to(torch.int64)]).to(cdev)
    allc = [torch.zeros(2, dtype=torch.int64, device=cdev) for _ in range(world)]
    dist.all_gather(allc, mine)
    counts = torch.stack(allc).cpu().numpy()  # (the host waits here: the sizes of the transfers below)
    nray_r, nrow_r = counts[:, 0], counts[:, 1]
    if int(nray_r.sum()) != nrays:
        raise RuntimeError("shards cover %d rays, expected %d" % (int(nray_r.sum()), nrays))
    if int(nrow_r[rank]) > packed.shape[0]:
        raise RuntimeError("rank %d: %d packed rows reported, the buffer holds %d" % (rank, int(nrow_r[rank]), packed.shape[0]))
    packed = packed[:int(nrow_r[rank])]
    ray0 = np.concatenate([[0], np.cumsum(nray_r)])
    row0 = np.concatenate([[0], np.cumsum(nrow_r)])
    src = [t.cpu() if stage else t for t in (packed.contiguous(), nrows.contiguous(), stop.contiguous())]
    ops = []
    out = None
    if rank == dst:
        out = (torch.empty((int(row0[-1]), ROW), dtype=packed.dtype, device=cdev),
               torch.empty(nrays, dtype=nrows.dtype, device=cdev), torch.empty(nrays, dtype=stop.dtype, device=cdev))
        for r in range(world):
            sl = [out[0][row0[r]:row0[r + 1]], out[1][ray0[r]:ray0[r + 1]], out[2][ray0[r]:ray0[r + 1]]]
            if r == dst:
                for d, s in zip(sl, src):
                    d.copy_(s)
            else:
                ops += [dist.P2POp(dist.irecv, t, r) for t in sl if t.numel()]
    else:
        ops += [dist.P2POp(dist.isend, t, dst) for t in src if t.numel()]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != dst:
        return None
    recv = int((row0[-1] - nrow_r[dst]) * ROW * 8 + (nrays - nray_r[dst]) * 8)
    if stage:
        out = tuple(t.to(dev) for t in out)
    return out + (recv,)


def trace_sharded(dist, nrays, trace_fn, pack_fn, dst=0, timings=None):
    """Run trace_fn(lo, hi) -> (rows[n,slots,20], nrows[n], stop[n]) (torch tensors) on this rank's shard, pack the
    kept rows with pack_fn(rows, nrows) -> (packed, offsets) and gather them to rank `dst` in ray order.
    Returns (packed_all, nrows_all, stop_all) on dst, else None.  timings (dict, optional) receives this rank's
    trace_s, pack_s, gather_s (each closed by a device synchronisation) and gather_bytes on dst."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(nrays, rank, world)
    t0 = time.perf_counter()
    rows, nrows, stop = trace_fn(lo, hi)
    _sync(rows)
    t1 = time.perf_counter()
    packed, offsets = pack_fn(rows, nrows)
    _sync(packed)
    t2 = time.perf_counter()
    res = gather_packed(dist, packed, nrows, stop, nrays, dst, total=offsets[-1:])
    _sync(packed)
    t3 = time.perf_counter()
    if timings is not None:
        timings.update(trace_s=t1 - t0, pack_s=t2 - t1, gather_s=t3 - t2, gather_bytes=res[3] if res is not None else 0,
                       shard=(lo, hi))
    return None if res is None else res[:3]


def trace_sharded_pipelined(dist, nrays, steps, launch_fn, pack_fn, dst=0, keep_last_only=True):
    """`steps` steps of trace -> pack -> gather with the pack + gather of step k overlapped with the trace of step k + 1:
    launch_fn(k) enqueues the trace of this rank's shard on the CURRENT stream into output buffer set k % 2 and returns
    (rows, nrows, stop); the pack and the gather of step k run on a side stream that waits for that trace only, while the
    current stream already carries the trace of step k + 1.  (The persistent trace kernel takes every CU's registers, so
    the pack / RCCL kernels of step k get onto the chip as the waves of step k + 1 thin out: what is hidden is the
    gather behind the next launch's tail.)  pack_fn(rows, nrows, slot) -> (packed, offsets).  CPU tensors (gloo tests):
    the same order of operations without streams.  Returns the gathered (packed, nrows, stop) of the last step on dst
    (of every step if keep_last_only is False), None elsewhere."""
    import torch

    results, pending = [], None
    cuda = None
    side = None

    def finish(item):
        k, rows, nrows, stop, ev = item
        if cuda:
            side.wait_event(ev)
            with torch.cuda.stream(side):
                packed, offsets = pack_fn(rows, nrows, k % 2)
                res = gather_packed(dist, packed, nrows, stop, nrays, dst, total=offsets[-1:])
        else:
            packed, offsets = pack_fn(rows, nrows, k % 2)
            res = gather_packed(dist, packed, nrows, stop, nrays, dst, total=offsets[-1:])
        if res is not None:
            if keep_last_only:
                results.clear()
            results.append(res[:3])

    for k in range(steps):
        rows, nrows, stop = launch_fn(k)
        if cuda is None:
            cuda = rows.is_cuda
            if cuda:
                side = torch.cuda.Stream(rows.device)
        ev = None
        if cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(rows.device))
        if pending is not None:
            finish(pending)  # (the host waits inside for the counts of step k - 1 while the device traces step k)
        pending = (k, rows, nrows, stop, ev)
    if pending is not None:
        finish(pending)
    if cuda:
        torch.cuda.current_stream().wait_stream(side)
    if not results:
        return None
    return results[-1] if keep_last_only else results


def unpack_rows(packed, nrows, outputper, slots):
    """Inverse of the packing on the root: -> rows[n, slots, 20] (zero beyond a ray's kept rows)."""
    import torch

    n = nrows.shape[0]
    kept = kept_rows(nrows.to(torch.int64), outputper, slots)
    rows = torch.zeros((n, slots, ROW), dtype=packed.dtype, device=packed.device)
    mask = torch.arange(slots, device=packed.device)[None, :] < kept[:, None]
    rows[mask] = packed
    return rows
