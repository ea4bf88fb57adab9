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
    """Combine a sequence of tensors with ``function`` (e.g. ``operator.add``), rounding every intermediate
    result (tools.py:460-512): a binary-counter tree, so that operands of similar rank meet."""
    from .round import round as _round

    d = dict()
    start = time.time()
    for i, elem in enumerate(ts):
        if verbose and i % 100 == 0:
            print("reduce: element {}, time={:g}".format(i, time.time() - start))
        climb = 0  # for going up the tree
        while climb in d:
            elem = _round(function(d[climb], elem, **kwargs), eps=eps, rmax=rmax, algorithm=algorithm)
            d.pop(climb)
            climb += 1
        d[climb] = elem
    keys = list(d.keys())
    result = d[keys[0]]
    for key in keys[1:]:
        result = _round(function(result, d[key], **kwargs), eps=eps, rmax=rmax, algorithm=algorithm)
    return result
