"""Batch sharding over the GPUs of one node: independent tensors, one gather at the end.

The sweeps of one tensor train are sequential dependency chains (core k+1 needs R from
core k), so a single tensor does not shard; a BATCH of independent tensors (the reference's
``batch=True`` dimension, tensor.py:163) shards embarrassingly.  One process per GPU
(``torch.distributed``, backend ``nccl`` = RCCL over xGMI on ROCm, ``gloo`` on CPU):
rank g owns the contiguous block ``[lo, hi)`` of the batch, runs the sweeps with no
communication, and the rounded cores are collected with ONE gather of a packed buffer
(rmax mode => identical core shapes on every rank, so the buffer layout is static).
"""

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .tensor import Tensor

__all__ = ["shard_range", "pack_cores", "unpack_cores", "gather_batch", "round_tt_sharded"]


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition of ``total`` items over ``world`` ranks (first ranks get the remainder)."""
    q, rem = divmod(total, world)
    lo = rank * q + min(rank, rem)
    return lo, lo + q + (1 if rank < rem else 0)


def pack_cores(cores: Sequence[torch.Tensor]) -> torch.Tensor:
    """Concatenate batched cores ``[B, r0, I, r1]`` into one flat buffer (core-major)."""
    return torch.cat([c.reshape(-1) for c in cores])


def unpack_cores(flat: torch.Tensor, shapes: Sequence[Sequence[int]]) -> List[torch.Tensor]:
    out, off = [], 0
    for shp in shapes:
        n = 1
        for s in shp:
            n *= int(s)
        out.append(flat[off : off + n].reshape(list(shp)))
        off += n
    return out


def gather_batch(t: Tensor, dst: int = 0, group=None) -> Optional[Tensor]:
    """Collect the batch-sharded tensor ``t`` (``batch=True``) on rank ``dst`` with a single gather.

    Every rank must hold cores of identical trailing shape (rmax-mode rounding); local batch
    sizes may differ by one (block partition) -- shorter shards are padded in the packed buffer.
    Returns the full-batch ``Tensor`` on ``dst`` and ``None`` elsewhere.
    """
    assert t.batch, "gather_batch needs a batch=True tensor"
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    Bl = t.cores[0].shape[0]
    dev = t.cores[0].device
    sizes = torch.tensor([Bl], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)  # 8 bytes per rank; part of the same exchange step
    all_B = [int(s.item()) for s in all_sizes]
    Bmax = max(all_B)
    cores = t.cores
    if Bl < Bmax:
        cores = [torch.cat([c, c.new_zeros((Bmax - Bl,) + tuple(c.shape[1:]))]) for c in cores]
    flat = pack_cores(cores)
    if rank == dst:
        bufs = [torch.empty_like(flat) for _ in range(world)]
        dist.gather(flat, gather_list=bufs, dst=dst, group=group)
        shapes = [[Bmax] + list(c.shape[1:]) for c in cores]
        per_rank = [unpack_cores(b, shapes) for b in bufs]
        merged = [torch.cat([per_rank[g][k][: all_B[g]] for g in range(world)]) for k in range(len(cores))]
        return Tensor(merged, batch=True)
    dist.gather(flat, gather_list=None, dst=dst, group=group)
    return None


def round_tt_sharded(cores: Sequence[torch.Tensor], rmax, algorithm: str = "svd", dst: int = 0, group=None):
    """Round the local shard of a batch and gather the result on ``dst``.

    ``cores``: this rank's block of the batch (``[B_local, r0, I, r1]`` per core).
    """
    t = Tensor(list(cores), batch=True)
    t.round_tt(rmax=rmax, algorithm=algorithm)
    return gather_batch(t, dst=dst, group=group)
