"""Multi-GPU layer: rays are independent, so the launch set is cut into contiguous shards (one per rank,
one rank per GPU, one model replica per rank) and the only communication is the final gather of the
trajectory buffers to rank 0 (RCCL over xGMI with backend "nccl"; "gloo" on CPU for tests).

The reference has no parallel mode at all (serial `do` over rays, raytracer_driver.f95:1144-1232);
shards reproduce exactly what running it on disjoint ray files would.
"""
import numpy as np


def shard_bounds(nrays, rank, world):
    """Contiguous block of ceil(n/world) rays for this rank (last ranks may be short or empty)."""
    per = (nrays + world - 1) // world
    lo = min(rank * per, nrays)
    hi = min(lo + per, nrays)
    return lo, hi


def gather_to_root(dist, local, per, dst=0):
    """Gather a per-ray tensor [n_local, ...] from every rank to `dst`, padded to `per` rows per rank.

    Returns the stacked [world*per, ...] tensor on dst (caller trims to nrays), None elsewhere."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat(bufs, dim=0)


def trace_sharded(dist, nrays, trace_fn, dst=0):
    """Run trace_fn(lo, hi) -> (rows[n,slots,20], nrows[n], stop[n]) (torch tensors) on this rank's shard
    and gather everything to rank `dst` in ray order.  Returns (rows, nrows, stop) on dst, else None."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(nrays, rank, world)
    rows, nrows, stop = trace_fn(lo, hi)
    per = (nrays + world - 1) // world
    out = [gather_to_root(dist, t, per, dst) for t in (rows, nrows, stop)]
    if rank != dst:
        return None
    return tuple(t[:nrays] for t in out)
