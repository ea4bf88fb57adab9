"""Unfoldings of TT cores / dense tensors -- the layout contract of the hot path -- and the rounding tree.

Mirror of ``tntorch/tools.py:211-258`` (same names, arguments and results).  All three are
pure ``reshape``/``permute`` views: a core ``[r0, I, r1]`` is row-major, so its left
unfolding has row index ``r0*I + i`` and its right unfolding column index ``i*R1 + r1`` --
exactly the addressing the HIP kernels use (no data movement on either side).
"""

import time

import numpy as np
import torch

__all__ = ["unfolding", "right_unfolding", "left_unfolding", "reduce", "shift_mode"]


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


def shift_mode(t, n, shift, eps=1e-3):
    """Move mode ``n`` of a tensor train ``shift`` positions to the right (``shift > 0``) or left, in place on the core list
    -- same contract as tools.py:650-697: the train is made ``n``-orthogonal, then every exchange of two neighbouring
    modes contracts the two cores, swaps the mode axes and splits the result again by ``truncated_svd`` (relative error
    ``eps / sqrt(|shift|)`` per exchange, or ``eps='same'``: the old bond rank is the cap).  On device tensors the
    contraction is one MFMA GEMM and the split the Gram / eigensolver / projection kernels of the rounding sweep.
    """
    from ._dispatch import ops_for
    from .round import truncated_svd

    N = t.dim()
    assert 0 <= n + shift < N
    if shift == 0:
        return t
    if t.batch:
        raise NotImplementedError("shift_mode: batched tensors are not supported (the reference indexes core.shape[0] as a rank)")
    if any(U is not None for U in t.Us):
        t = t.decompress_tucker_factors(_clone=False)
    if isinstance(eps, str):
        if eps != "same":
            raise ValueError("Relative error '{}' not recognized".format(eps))
    elif not eps >= 0:
        raise ValueError("Relative error '{}' not recognized".format(eps))
    t.orthogonalize(n)
    cores = t._norm4()  # [1, r0, I, r1] views (CP factors become TT cores, as the reference's orthogonalize does)
    sign = 1 if shift > 0 else -1
    for i in range(n, n + shift, sign):
        c1, c2, left_ortho = (i, i + 1, True) if sign == 1 else (i - 1, i, False)
        _, R1, I1, R2 = cores[c1].shape
        _, _, I2, R3 = cores[c2].shape
        sc = ops_for(cores[c1]).merge_swap(cores[c1], cores[c2])[0]  # [R1*I2, I1*R3]
        if isinstance(eps, str):
            left, right = truncated_svd(sc, eps=0, rmax=R2, left_ortho=left_ortho)
        else:
            left, right = truncated_svd(sc, eps=eps / np.sqrt(np.abs(shift)), left_ortho=left_ortho)
        newR2 = left.shape[1]
        cores[c1] = left.reshape(1, R1, I2, newR2)
        cores[c2] = right.reshape(1, newR2, I1, R3)
    t.cores = t._denorm(cores)
    return t
