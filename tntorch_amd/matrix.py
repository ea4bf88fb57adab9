"""``TTMatrix``: a matrix stored as a tensor train of ``[r, i_k, o_k, r']`` cores -- the construction path of
``tntorch/matrix.py:12-160`` (the consumer that sits directly on the dense TT-SVD).

A matrix ``M`` of shape ``prod(input_dims) x prod(output_dims)`` is reshaped to ``i_0 x ... x i_{d-1} x o_0 x ... x o_{d-1}``,
its axes interleaved to ``(i_0 o_0) x ... x (i_{d-1} o_{d-1})``, decomposed by ``Tensor(..., ranks_tt=ranks)`` -- on device
tensors the streaming right-to-left TT-SVD kernels -- and every core ``[r, i_k o_k, r']`` is viewed as ``[r, i_k, o_k, r']``.
``torch()`` contracts the train (a chain of MFMA GEMMs on the device) and undoes the interleaving.
"""

from __future__ import annotations

from typing import List, Optional, Sequence, Union

import torch

from ._dispatch import ops_for
from .tensor import Tensor

__all__ = ["TTMatrix"]


class TTMatrix:
    """Same constructor contract as ``tntorch.TTMatrix`` (matrix.py:23-111)."""

    def __init__(self, t: Union[torch.Tensor, Sequence[torch.Tensor]], ranks: Optional[List[int]], input_dims: Sequence[int],
                 output_dims: Sequence[int]):
        assert len(input_dims) == len(output_dims)
        assert len(input_dims) > 0
        self.input_dims = torch.tensor(list(input_dims))
        self.output_dims = torch.tensor(list(output_dims))
        self.d = len(input_dims)
        idims, odims = [int(x) for x in input_dims], [int(x) for x in output_dims]

        if isinstance(t, (list, tuple)):  # pre-processed cores (matrix.py:48-57)
            core_dims = len(t[0].shape)
            assert core_dims in [4, 5]
            self.batch = core_dims == 5  # b x r_{i-1} x input_i x output_i x r_i
            self.cores = list(t)
            self.ranks = torch.tensor([c.shape[-1] for c in t[:-1]])
            return

        assert isinstance(ranks, list) and len(ranks) == self.d - 1
        M = t
        assert len(M.shape) in [2, 3]
        self.batch = len(M.shape) == 3
        nb = 1 if self.batch else 0
        rows, cols = 1, 1
        for a, b in zip(idims, odims):
            rows, cols = rows * a, cols * b
        assert rows == M.shape[-2]
        assert cols == M.shape[-1]

        lead = [M.shape[0]] if self.batch else []
        tensor = M.reshape(lead + idims + odims)
        # i_0 .. i_{d-1} o_0 .. o_{d-1}  ->  i_0 o_0 i_1 o_1 ...  (matrix.py:69-92); a layout copy
        perm = list(range(nb)) + [nb + k + off for k in range(self.d) for off in (0, self.d)]
        tensor = tensor.permute(perm).reshape(lead + [idims[k] * odims[k] for k in range(self.d)])
        if not tensor.is_contiguous():
            tensor = tensor.contiguous()
        tt = Tensor(tensor, ranks_tt=ranks, batch=self.batch)
        self.ranks = tt.ranks_tt[1:-1]
        self.cores = [
            c.reshape((list(c.shape[:nb + 1])) + [idims[k], odims[k], c.shape[-1]]) for k, c in enumerate(tt.cores)
        ]

    # ------------------------------------------------------------------ conversions (matrix.py:113-158)
    def flatten(self) -> Tensor:
        """The train with every core's input and output mode merged: a ``Tensor`` of shape ``(i_k o_k)_k`` (matrix.py:177-201)."""
        nb = 1 if self.batch else 0
        return Tensor([c.reshape(list(c.shape[:nb + 1]) + [c.shape[nb + 1] * c.shape[nb + 2], c.shape[-1]]) for c in self.cores],
                      batch=self.batch)

    def torch(self) -> torch.Tensor:
        """Decompress into a dense ``rows x cols`` matrix (batch: ``b x rows x cols``)."""
        idims, odims = self.input_dims.tolist(), self.output_dims.tolist()
        nb = 1 if self.batch else 0
        dense = self.flatten().torch()
        lead = [dense.shape[0]] if self.batch else []
        shape = [x for k in range(self.d) for x in (idims[k], odims[k])]
        dense = dense.reshape(lead + shape)
        perm = list(range(nb)) + [nb + 2 * k for k in range(self.d)] + [nb + 2 * k + 1 for k in range(self.d)]
        rows, cols = int(torch.prod(self.input_dims)), int(torch.prod(self.output_dims))
        return dense.permute(perm).reshape(lead + [rows, cols])

    def to(self, device):
        self.cores = [c.to(device) for c in self.cores]
        return self

    def numpy(self):
        return self.torch().detach().cpu().numpy()

    def trace(self) -> torch.Tensor:
        """Trace of the matrix (matrix.py:160-175): the diagonal slices of every core summed, then a chain of small
        products; a scalar (batch: one per item)."""
        c5 = self.cores if self.batch else [c[None] for c in self.cores]
        ops = ops_for(c5[0])
        acc = None
        for c in c5:
            D = ops.diag_sum(c)  # [B, r0, r1]
            if acc is None:
                acc = D
            elif D.is_cuda:
                from . import _hip

                acc = _hip.gemm(acc, D)
            else:
                acc = torch.bmm(acc, D)
        if acc.shape[1] > 1:
            # a leading boundary rank > 1 is summed away: the reference starts from factor = ones(1), which einsum broadcasts
            # over the first rank index (matrix.py:160-175)
            if acc.is_cuda:
                from . import _hip

                acc = _hip.gemm(torch.ones((acc.shape[0], 1, acc.shape[1]), dtype=acc.dtype, device=acc.device), acc)
            else:
                acc = acc.sum(dim=1, keepdim=True)
        out = acc[:, 0, 0]
        return out if self.batch else out[0]
