"""Unfoldings of TT cores / dense tensors -- the layout contract of the hot path.

Mirror of ``tntorch/tools.py:211-258`` (same names, arguments and results).  All three are
pure ``reshape``/``permute`` views: a core ``[r0, I, r1]`` is row-major, so its left
unfolding has row index ``r0*I + i`` and its right unfolding column index ``i*R1 + r1`` --
exactly the addressing the HIP kernels use (no data movement on either side).
"""

import torch

__all__ = ["unfolding", "right_unfolding", "left_unfolding"]


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
