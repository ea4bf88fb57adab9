"""Unfoldings of TT cores / dense tensors -- the layout contract of the hot path -- and the rounding tree.

Mirror of ``tntorch/tools.py:211-258`` (same names, arguments and results).  All three are
pure ``reshape``/``permute`` views: a core ``[r0, I, r1]`` is row-major, so its left
unfolding has row index ``r0*I + i`` and its right unfolding column index ``i*R1 + r1`` --
exactly the addressing the HIP kernels use (no data movement on either side).
"""

import time

import numpy as np
import torch

__all__ = ["unfolding", "right_unfolding", "left_unfolding", "reduce"]


def unfolding(data: torch.Tensor, n: int, batch: bool = False) -> torch.Tensor:
    """Mode-``n`` unfolding of a dense tensor (tools.py:211-228)."""
    if batch:
        order = [0, n + 1] + list(range(1, n + 1)) + list(range(n + 2, data.dim()))
        return data.permute(order).reshape([data.shape[0], data.shape[n + 1], -1])
    order = [n] + list(range(n)) + list(range(n + 1, data.dim()))
    return data.permute(order).reshape([data.shape[n], -1])


def right_unfolding(core: torch.Tensor, batch: bool = False) -> torch.Tensor:
    """``[r0, I, r1] -> [r0, I*r1]`` (tools.py:231-243)."""
    if batch:
        return core.reshape([core.shape[0], core.shape[1], -1])
    return core.reshape([core.shape[0], -1])


def left_unfolding(core: torch.Tensor, batch: bool = False) -> torch.Tensor:
    """``[r0, I, r1] -> [r0*I, r1]`` (tools.py:246-258)."""
    if batch:
        return core.reshape([core.shape[0], -1, core.shape[-1]])
    return core.reshape([-1, core.shape[-1]])


def reduce(ts, function, eps=0, rmax=np.iinfo(np.int32).max, algorithm="svd", verbose=False, **kwargs):
    """Fold a sequence (or generator) of tensors with ``function`` (``operator.add``, ``tn.cat`` ...), rounding after
    every combination so that no intermediate exceeds ``rmax`` / ``eps`` -- same contract as tools.py:460-512.

    The combinations form a balanced tree built on the fly like a binary counter: ``slots[h]`` holds the rounded result
    of a complete group of 2**h consecutive elements (or nothing); a new element enters at height 0 and, like a carry,
    merges upwards while the slot at its height is occupied.  Operands of a merge therefore always cover equally many
    elements (similar ranks meet), at most log2(len) partial results are alive, and ``function`` always receives the
    EARLIER group first (order matters for e.g. concatenation).  At the end the remaining groups are merged from the
    oldest (highest) to the newest.
    """
    import operator

    from .round import round as _round
    from .tensor import Tensor

    def merge(older, newer):
        if function is operator.add and not kwargs:
            # add-then-round fused: the concatenated cores of the sum are never materialised (device TT tensors)
            fused = Tensor._round_of_sum(older, newer, eps=eps, rmax=rmax, algorithm=algorithm)
            if fused is not None:
                return fused
        return _round(function(older, newer, **kwargs), eps=eps, rmax=rmax, algorithm=algorithm)

    slots = []
    started = time.time()
    for count, item in enumerate(ts):
        if verbose and count % 100 == 0:
            print("reduce: element {}, time={:g}".format(count, time.time() - started))
        height = 0
        while height < len(slots) and slots[height] is not None:
            item = merge(slots[height], item)
            slots[height] = None
            height += 1
        if height == len(slots):
            slots.append(item)
        else:
            slots[height] = item
    groups = [g for g in reversed(slots) if g is not None]
    if not groups:
        raise ValueError("reduce() needs at least one tensor")
    result = groups[0]
    for g in groups[1:]:
        result = merge(result, g)
    return result
