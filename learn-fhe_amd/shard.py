"""Data-parallel sharding of independent ring operations across the GPUs of one node.

Batches of independent polynomials / ciphertexts split contiguously across ranks; twiddle tables and
keys are replicated.  No collective is needed on the data path; the only
collective is the optional final gather of results."""
from __future__ import annotations


def shard_range(total: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of `total` independent units owned by `rank`; sizes differ by at most one
    and every unit is owned exactly once (ragged totals allowed, empty shards allowed)."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_results(local, group=None):
    """Final gather of per-rank results (torch tensors of equal shape) onto every rank: one all_gather
    (RCCL over xGMI on GPUs, gloo on CPU).  Not part of the timed hot path."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(out, local, group=group)
    return torch.cat(out, dim=0)
