"""Inner products / error metrics on tensor trains used by the parity tests.

Mirror of the TT subset of ``tntorch/metrics.py`` (``dot`` 28-116, ``dist`` 119-132,
``relative_error`` 135-151, ``normsq`` 457-466, ``norm`` 469-478): the consumers of the hot path
(SURVEY 8f-4).  The contraction runs on the HIP GEMM for device tensors and on torch for CPU tensors.
"""

import torch

from ._dispatch import ops_for
from .tensor import Tensor

__all__ = ["dot", "dist", "relative_error", "normsq", "norm"]


def _dense(t):
    return t.torch() if isinstance(t, Tensor) else t


def dot(t1, t2):
    """Full inner product <t1, t2> (metrics.py:28-116 with k = N, no Tucker factors)."""
    if not isinstance(t1, Tensor) or not isinstance(t2, Tensor):
        a, b = _dense(t1), _dense(t2)
        return ops_for(a).dense_dot(a, b)
    if t1.batch or t2.batch:
        raise ValueError("Batched tensors are not supproted.")
    if t1.shape != t2.shape:
        raise ValueError("Dot product requires leading dimensions to be equal, but they are {} and {}".format(t1.shape, t2.shape))
    c1 = t1._absorbed4()  # Tucker factors contracted in (the reference contracts U1^T U2 instead: same value)
    return ops_for(c1[0]).dot(c1, t2._absorbed4())  # device tensors: Lprod chain on ttr_gemm


def dist(t1, t2):
    """Euclidean distance (metrics.py:119-132)."""
    if not isinstance(t1, Tensor) or not isinstance(t2, Tensor):
        a, b = _dense(t1), _dense(t2)
        return ops_for(a).dense_dist(a, b)
    return torch.sqrt((dot(t1, t1) + dot(t2, t2) - 2 * dot(t1, t2)).clamp(0))


def relative_error(gt, approx):
    """||gt - approx|| / ||gt|| (metrics.py:135-151).

    Between two compressed tensors this uses the reference's <a,a>+<b,b>-2<a,b> formula and
    is therefore limited to ~sqrt(machine eps) by cancellation.
    """
    if not isinstance(gt, Tensor) or not isinstance(approx, Tensor):
        a, b = _dense(gt), _dense(approx)
        ops = ops_for(a)
        return ops.dense_dist(a, b) / ops.dense_norm(a)
    dotgt = dot(gt, gt)
    return torch.sqrt((dotgt + dot(approx, approx) - 2 * dot(gt, approx)).clamp(0)) / torch.sqrt(dotgt.clamp(0))


def normsq(t):
    return dot(t, t)


def norm(t):
    return torch.sqrt(torch.clamp(normsq(t), min=0))
