"""Batch sharding over the GPUs of one node: independent tensors, one gather at the end.

The sweeps of one tensor train are sequential dependency chains (core k+1 needs R from
core k), so a single tensor does not shard; a BATCH of independent tensors (the reference's
``batch=True`` dimension, tensor.py:163) shards embarrassingly.  One process per GPU
(``torch.distributed``, backend ``nccl`` = RCCL over xGMI on ROCm, ``gloo`` on CPU):
rank g owns the contiguous block ``[lo, hi)`` of the batch, runs the sweeps with no
communication, and the rounded cores are collected with ONE gather of a packed buffer
(rmax mode => identical core shapes on every rank, so the buffer layout is static).

xGMI is point-to-point: every peer reaches the root over its own link, so the gather is
bound by one link per peer (0.8 GB per rank for 512 rounded 64^8 trains).  The device ``round_tt``
writes the rounded cores of a large batch back to back into one buffer, so packing is a view.  ``gather_batch``
can therefore run asynchronously (``async_op=True``): RCCL executes it on its own stream
while the next batch is already being rounded, and the root receives views into the
per-rank receive buffers -- no concatenation copy.
"""

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .tensor import Tensor

__all__ = ["shard_range", "pack_cores", "unpack_cores", "gather_batch", "round_tt_sharded", "GatherHandle", "GatherSchedule"]


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition of ``total`` items over ``world`` ranks (first ranks get the remainder)."""
    q, rem = divmod(total, world)
    lo = rank * q + min(rank, rem)
    return lo, lo + q + (1 if rank < rem else 0)


def pack_cores(cores: Sequence[torch.Tensor]) -> torch.Tensor:
    """Batched cores ``[B, r0, I, r1]`` as one flat buffer (core-major).  Zero-copy when the cores already lie back
    to back in one storage -- which is how the device ``round_tt`` lays out the result of a large batch."""
    c0 = cores[0]
    off = c0.storage_offset()
    packed = all(c.is_contiguous() and c.dtype == c0.dtype for c in cores)
    if packed:
        for c in cores:
            if c.untyped_storage().data_ptr() != c0.untyped_storage().data_ptr() or c.storage_offset() != off:
                packed = False
                break
            off += c.numel()
    if packed:
        total = off - c0.storage_offset()
        return torch.empty(0, dtype=c0.dtype, device=c0.device).set_(c0.untyped_storage(), c0.storage_offset(), (total,), (1,))
    return torch.cat([c.reshape(-1) for c in cores])


def unpack_cores(flat: torch.Tensor, shapes: Sequence[Sequence[int]]) -> List[torch.Tensor]:
    """Views of ``flat`` with the given shapes (inverse of ``pack_cores``, zero-copy)."""
    out, off = [], 0
    for shp in shapes:
        n = 1
        for s in shp:
            n *= int(s)
        out.append(flat[off : off + n].reshape(list(shp)))
        off += n
    return out


class GatherHandle:
    """Result of ``gather_batch``: ``wait()`` returns, on the destination rank, one ``Tensor`` per source
    rank (cores are views into the receive buffers) -- ``None`` on the other ranks."""

    def __init__(self, work, bufs, shapes, sizes, keep):
        self._work, self._bufs, self._shapes, self._sizes, self._keep = work, bufs, shapes, sizes, keep
        self._done = work is None

    def wait(self) -> Optional[List[Tensor]]:
        if not self._done:
            self._work.wait()  # stream-level wait for NCCL, blocking for gloo
            self._done = True
        if self._bufs is None:
            return None
        out = []
        for g, buf in enumerate(self._bufs):
            cores = unpack_cores(buf, self._shapes)
            out.append(Tensor([c[: self._sizes[g]] for c in cores], batch=True))
        return out

    def merged(self) -> Optional[Tensor]:
        """Concatenate the per-rank pieces into one full-batch ``Tensor`` (copies)."""
        parts = self.wait()
        if parts is None:
            return None
        if len(parts) == 1:
            return parts[0]
        n = parts[0].dim()
        return Tensor([torch.cat([p.cores[k] for p in parts]) for k in range(n)], batch=True)


def gather_batch(
    t: Tensor,
    dst: int = 0,
    group=None,
    sizes: Optional[Sequence[int]] = None,
    async_op: bool = False,
    local_shortcut: bool = True,
) -> GatherHandle:
    """Collect the batch-sharded tensor ``t`` (``batch=True``) on rank ``dst`` with a single gather.

    Every rank must hold cores of identical trailing shape (rmax-mode rounding).  ``sizes``: local batch
    size of every rank when known up front (skips the 8-byte size exchange and its host sync); shorter
    shards are padded in the packed buffer.  ``async_op=True`` returns immediately; call ``wait()``.
    ``local_shortcut=False``: a process group of ONE rank still goes through the collective (tests: the code an
    N-rank job runs -- packed view, receive buffers, ``dist.gather``, the work handle -- on a single GPU).
    """
    assert t.batch, "gather_batch needs a batch=True tensor"
    Bl = t.cores[0].shape[0]
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and local_shortcut):
        flat = pack_cores(t.cores)
        return GatherHandle(None, [flat], [list(c.shape) for c in t.cores], [Bl], None)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = t.cores[0].device
    if sizes is None:
        mine = torch.tensor([Bl], dtype=torch.int64, device=dev)
        allsz = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allsz, mine, group=group)
        sizes = [int(s.item()) for s in allsz]
    sizes = [int(s) for s in sizes]
    assert sizes[rank] == Bl, "sizes[rank] does not match the local batch"
    Bmax = max(sizes)
    cores = t.cores
    if Bl < Bmax:
        cores = [torch.cat([c, c.new_zeros((Bmax - Bl,) + tuple(c.shape[1:]))]) for c in cores]
    flat = pack_cores(cores)
    shapes = [[Bmax] + list(c.shape[1:]) for c in cores]
    if rank == dst:
        bufs = [torch.empty_like(flat) for _ in range(world)]
        work = dist.gather(flat, gather_list=bufs, dst=dst, group=group, async_op=True)
        h = GatherHandle(work, bufs, shapes, sizes, flat)
    else:
        work = dist.gather(flat, gather_list=None, dst=dst, group=group, async_op=True)
        h = GatherHandle(work, None, shapes, sizes, flat)
    if not async_op:
        h.wait()
    return h


def round_tt_sharded(cores: Sequence[torch.Tensor], rmax, algorithm: str = "svd", dst: int = 0, group=None,
                     sizes: Optional[Sequence[int]] = None):
    """Round the local shard of a batch and gather the result on ``dst``.

    ``cores``: this rank's block of the batch (``[B_local, r0, I, r1]`` per core).  Returns the full-batch
    ``Tensor`` on ``dst`` and ``None`` elsewhere.
    """
    t = Tensor(list(cores), batch=True)
    t.round_tt(rmax=rmax, algorithm=algorithm)
    return gather_batch(t, dst=dst, group=group, sizes=sizes).merged()


class GatherSchedule:
    """When the rounded cores of a stream of steps travel to the root (the policy `bench.py --gather` selects):

    ``end``   north_star's reading -- "a single RCCL gather ... at the end": nothing moves while the steps run, ``drain()``
              gathers the LAST step's result (blocking) -- one collective per job.
    ``step``  every step's result is delivered: ``after_step`` starts an asynchronous gather (RCCL's own stream, under the next
              step's compute) after waiting for the previous one -- at most one in flight; ``drain()`` completes the last.
    ``none``  no gather (compute-only scaling).

    ``drain()`` returns, on ``dst``, the list of per-rank ``Tensor`` parts of the last gathered step (``None`` elsewhere / for
    ``none``).  ``gathers`` counts the collectives issued."""

    MODES = ("end", "step", "none")

    def __init__(self, mode: str, sizes: Optional[Sequence[int]] = None, dst: int = 0, group=None, local_shortcut: bool = True):
        if mode not in self.MODES:
            raise ValueError(f"gather mode must be one of {self.MODES}, got {mode!r}")
        self.mode, self.sizes, self.dst, self.group, self.local_shortcut = mode, sizes, dst, group, local_shortcut
        self._pending: Optional[GatherHandle] = None
        self._last: Optional[Tensor] = None
        self.gathers = 0

    def _gather(self, t: Tensor, async_op: bool) -> GatherHandle:
        self.gathers += 1
        return gather_batch(t, dst=self.dst, group=self.group, sizes=self.sizes, async_op=async_op, local_shortcut=self.local_shortcut)

    def after_step(self, t: Tensor) -> None:
        self._last = t
        if self.mode == "step":
            if self._pending is not None:
                self._pending.wait()
            self._pending = self._gather(t, True)

    def drain(self) -> Optional[List[Tensor]]:
        if self.mode == "end" and self._last is not None:
            t, self._last = self._last, None
            return self._gather(t, False).wait()
        if self._pending is not None:
            res = self._pending.wait()
            self._pending = None
            return res
        return None
