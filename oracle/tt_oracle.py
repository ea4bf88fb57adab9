"""CPU oracle for the TT orthogonalisation / rounding hot path of tntorch.

TEST INFRASTRUCTURE.  This file restates, function by function, the algorithm
of the reference (``/root/reference/tntorch``) on plain lists of torch CPU
tensors (no ``Tensor`` class), calling the same LAPACK entry points the
reference reaches through ``torch.linalg`` (geqrf/orgqr, gesdd, syevd).  It is
pinned against the reference itself: ``oracle/gen_golden.py`` imports the
unmodified reference in the build container, records its outputs on seeded
inputs under ``tests/golden/`` and ``tests/test_oracle_golden.py`` replays
them against this file.  Parity status: PINNED (golden vectors + the
known-answer values of ``docs/tutorials/decompositions.ipynb`` cells 3/18).

A TT tensor is a ``list`` of cores ``[R_k, I_k, R_{k+1}]`` (``batch=True``
prepends a batch dim to every core), as in ``tensor.py:111-117``.

Every function names the reference lines it follows.
"""

from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple, Union

import torch

__all__ = [
    "left_unfolding",
    "right_unfolding",
    "unfolding",
    "truncated_svd",
    "left_orthogonalize",
    "right_orthogonalize",
    "orthogonalize",
    "round_tt",
    "round_tucker",
    "round_general",
    "tucker_absorb",
    "tucker_to_dense",
    "dense_to_tucker_tt",
    "relative_error_tt",
    "cp_hosvd_init",
    "cp_to_dense",
    "cp_als",
    "cp_als_batch",
    "cp_on_tucker_core",
    "full_rank_tt",
    "dense_to_tt",
    "tt_to_dense",
    "tt_add",
    "tt_mul",
    "reduce_sum",
    "tt_scale",
    "tt_randn",
    "tt_rand",
    "tt_ranks",
    "tt_dot",
    "tt_norm",
    "bond_singular_values",
    "gauge_align",
    "shift_mode",
    "ttmatrix_cores",
    "ttmatrix_to_dense",
    "ttmatrix_trace",
]

Cores = List[torch.Tensor]


# --------------------------------------------------------------------------
# unfoldings -- tools.py:211-258 (pure reshape views)
# --------------------------------------------------------------------------
def right_unfolding(core: torch.Tensor, batch: bool = False) -> torch.Tensor:
    """tools.py:231-243: [r0, I, r1] -> [r0, I*r1]."""
    if batch:
        return core.reshape(core.shape[0], core.shape[1], -1)
    return core.reshape(core.shape[0], -1)


def left_unfolding(core: torch.Tensor, batch: bool = False) -> torch.Tensor:
    """tools.py:246-258: [r0, I, r1] -> [r0*I, r1]."""
    if batch:
        return core.reshape(core.shape[0], -1, core.shape[-1])
    return core.reshape(-1, core.shape[-1])


def unfolding(data: torch.Tensor, n: int, batch: bool = False) -> torch.Tensor:
    """tools.py:211-228: mode-n unfolding (mode n first, the rest flattened)."""
    if batch:
        order = [0, n + 1] + [d for d in range(1, data.dim()) if d != n + 1]
        return data.permute(order).reshape(data.shape[0], data.shape[n + 1], -1)
    order = [n] + [d for d in range(data.dim()) if d != n]
    return data.permute(order).reshape(data.shape[n], -1)


def _t(M: torch.Tensor) -> torch.Tensor:
    return M.transpose(-1, -2)


# --------------------------------------------------------------------------
# truncated SVD -- round.py:52-187
# --------------------------------------------------------------------------
def truncated_svd(
    M: torch.Tensor,
    delta: Optional[float] = None,
    eps: Optional[float] = None,
    rmax: Optional[int] = None,
    left_ortho: bool = True,
    algorithm: str = "svd",
    batch: bool = False,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """round.py:52-187.  Returns ``left (m x r)``, ``M2 (r x n)``, left@M2 ~= M."""
    if delta is not None and eps is not None:  # round.py:77-78
        raise ValueError("Provide either `delta` or `eps`")
    if delta is None and eps is not None:  # round.py:79-80
        delta = eps * torch.norm(M).item()
    if delta is None:  # round.py:81-82
        delta = 0
    if rmax is None:  # round.py:83-84
        rmax = torch.iinfo(torch.int32).max
    assert rmax >= 1
    assert algorithm in ("svd", "eig")

    if algorithm == "svd":  # round.py:94-100 (full_matrices=True, Vh dropped)
        U, sig = torch.linalg.svd(M)[:2]
        side = "left"
    else:  # round.py:101-135
        if M.shape[-2] <= M.shape[-1]:
            gram, side = M @ _t(M), "left"
        else:
            gram, side = _t(M) @ M, "right"
        w, U = torch.linalg.eigh(gram)
        w = torch.where(w < 0, torch.zeros_like(w) + 1e-8, w)  # quirk: -> 1e-8
        sig = torch.sqrt(w)
        # argsort ascending then reverse (round.py:121-135)
        n = sig.shape[-1]
        rev = torch.arange(n - 1, -1, -1)
        if batch:
            idx = torch.argsort(sig)[:, rev]
            U = torch.stack([U[b][..., idx[b]] for b in range(len(idx))])
            sig = torch.stack([sig[b][idx[b]] for b in range(len(idx))])
        else:
            idx = torch.argsort(sig)[rev]
            U, sig = U[..., idx], sig[idx]

    # zero guard (round.py:137-145); the reference allocates CPU/default-dtype
    # zeros -- the oracle keeps M's dtype.
    if batch:
        if sig.max() < 1e-13:
            B = M.shape[0]
            return (
                torch.zeros(B, M.shape[1], 1, dtype=M.dtype),
                torch.zeros(B, 1, M.shape[2], dtype=M.dtype),
            )
    elif sig[0] < 1e-13:
        return torch.zeros(M.shape[0], 1, dtype=M.dtype), torch.zeros(1, M.shape[1], dtype=M.dtype)

    S = sig**2  # round.py:147
    if batch:  # round.py:149-150 (eps/delta ignored)
        rank = max(1, int(min(rmax, S.shape[-1])))
    else:  # round.py:152-158
        tail = torch.cumsum(torch.flip(S, [0]), dim=0) <= delta**2
        where = torch.where(tail)[0]
        if len(where) == 0:
            rank = max(1, int(min(rmax, len(S))))
        else:
            rank = max(1, int(min(rmax, len(S) - 1 - int(where[-1]))))

    left = U[..., :rank]
    sr = sig[..., :rank]
    if side == "left":  # round.py:164-172
        if left_ortho:
            M2 = _t(left) @ M
        else:
            M2 = (1.0 / sr)[..., None] * _t(left) @ M
            left = left * sr[..., None, :]
    else:  # round.py:173-182
        if left_ortho:
            newleft = M @ (left * (1.0 / sr)[..., None, :])
            M2 = _t(left * sr[..., None, :])
            left = newleft
        else:
            newleft = M @ left
            M2 = _t(left)
            left = newleft
    return left, M2


# --------------------------------------------------------------------------
# orthogonalisation sweeps -- tensor.py:1800-1909 (pure TT cores, Us = None)
# --------------------------------------------------------------------------
def factor_orthogonalize(cores: Cores, Us, mu: int, batch: bool = False) -> None:
    """tensor.py:1771-1798: QR of the Tucker factor, R pushed into the core (no-op without a factor)."""
    if Us is None or Us[mu] is None:
        return
    Q, R = torch.linalg.qr(Us[mu])
    Us[mu] = Q
    if batch:
        cores[mu] = torch.einsum("bijk,baj->biak", cores[mu], R)
    else:
        cores[mu] = torch.einsum("ijk,aj->iak", cores[mu], R)


def left_orthogonalize(cores: Cores, mu: int, batch: bool = False, Us=None) -> torch.Tensor:
    """tensor.py:1800-1833: reduced QR of the left unfolding; R is pushed right."""
    assert 0 <= mu < len(cores) - 1
    factor_orthogonalize(cores, Us, mu, batch)  # tensor.py:1815
    core = cores[mu]
    Q, R = torch.linalg.qr(left_unfolding(core, batch))
    cores[mu] = Q.reshape(core.shape[:-1] + (Q.shape[-1],))
    nxt = cores[mu + 1]
    pushed = R @ right_unfolding(nxt, batch)
    if batch:
        cores[mu + 1] = pushed.reshape((R.shape[0], R.shape[1]) + nxt.shape[2:])
    else:
        cores[mu + 1] = pushed.reshape((R.shape[0],) + nxt.shape[1:])
    return R


def right_orthogonalize(cores: Cores, mu: int, batch: bool = False, Us=None) -> torch.Tensor:
    """tensor.py:1835-1879: QR of the transposed right unfolding; L is pushed left."""
    assert 1 <= mu < len(cores)
    factor_orthogonalize(cores, Us, mu, batch)  # tensor.py:1850
    core = cores[mu]
    Q, L = torch.linalg.qr(_t(right_unfolding(core, batch)))
    Q, L = _t(Q), _t(L)
    if batch:
        cores[mu] = Q.reshape(Q.shape[:2] + core.shape[2:])
    else:
        cores[mu] = Q.reshape((Q.shape[0],) + core.shape[1:])
    prev = cores[mu - 1]
    cores[mu - 1] = (left_unfolding(prev, batch) @ L).reshape(prev.shape[:-1] + (L.shape[-1],))
    return L


def orthogonalize(cores: Cores, mu: int, batch: bool = False, Us=None):
    """tensor.py:1881-1909: make the train mu-orthogonal (in place on the list, and on ``Us`` if given)."""
    N = len(cores)
    if mu < 0:
        mu += N
    R = L = None
    for i in range(mu):
        R = left_orthogonalize(cores, i, batch, Us)
    for i in range(N - 1, mu, -1):
        L = right_orthogonalize(cores, i, batch, Us)
    return R, L


# --------------------------------------------------------------------------
# TT rounding -- tensor.py:2008-2083
# --------------------------------------------------------------------------
def round_tt(
    cores: Sequence[torch.Tensor],
    eps: float = 1e-14,
    rmax: Union[None, int, Sequence[Optional[int]]] = None,
    algorithm: str = "svd",
    batch: bool = False,
    Us=None,
) -> Cores:
    """tensor.py:2008-2083.  Returns the rounded cores (input list untouched).  ``Us`` (optional list of
    Tucker factors) is modified IN PLACE by the factor orthogonalisations of the L2R sweep."""
    cores = [c.clone() for c in cores]
    N = len(cores)
    if not hasattr(rmax, "__len__"):
        rmax = [rmax] * (N - 1)
    assert len(rmax) == N - 1  # tensor.py:2029

    orthogonalize(cores, N - 1, batch, Us)  # tensor.py:2033
    if batch:  # tensor.py:2036-2037
        delta = None
    else:  # tensor.py:2039-2051 (float64 factor times the core's norm)
        delta = (eps / max(1.0, math.sqrt(N - 1))) * torch.norm(cores[-1]).double()
        delta = delta.item()

    for mu in range(N - 1, 0, -1):  # tensor.py:2053-2083
        core = cores[mu]
        left, right = truncated_svd(
            right_unfolding(core, batch),
            delta=delta,
            rmax=rmax[mu - 1],
            left_ortho=False,
            algorithm=algorithm,
            batch=batch,
        )
        left, right = left.to(core.dtype), right.to(core.dtype)
        if batch:
            cores[mu] = right.reshape(core.shape[0], -1, core.shape[2], core.shape[3])
        else:
            cores[mu] = right.reshape(-1, core.shape[1], core.shape[2])
        cores[mu - 1] = cores[mu - 1] @ left if not batch else torch.matmul(cores[mu - 1], left[:, None])
    return cores


# --------------------------------------------------------------------------
# Tucker rounding and the general round() -- tensor.py:1911-2006, 2085-2098
# A TT-Tucker tensor is (cores, Us): cores [R_k, S_k, R_{k+1}], Us[k] = None or [I_k, S_k]
# (batch=True prepends B to both).
# --------------------------------------------------------------------------
def round_tucker(
    cores: Sequence[torch.Tensor],
    Us: Optional[Sequence[Optional[torch.Tensor]]] = None,
    eps: float = 1e-14,
    rmax: Union[None, int, Sequence[Optional[int]]] = None,
    dim="all",
    algorithm: str = "svd",
    batch: bool = False,
):
    """tensor.py:1911-2006.  Returns (cores, Us) (inputs untouched)."""
    cores = [c.clone() for c in cores]
    N = len(cores)
    Us = [None] * N if Us is None else [None if U is None else U.clone() for U in Us]
    if not hasattr(rmax, "__len__"):
        rmax = [rmax] * N
    assert len(rmax) == N  # tensor.py:1933
    if dim == "all":
        dim = range(N)
    if not hasattr(dim, "__len__"):
        dim = [dim] * N
    orthogonalize(cores, N - 1, batch, Us)  # tensor.py:1944
    for mu in range(N - 1, -1, -1):  # tensor.py:1945
        core = cores[mu]
        if Us[mu] is None:  # tensor.py:1946-1958
            if batch:
                Us[mu] = torch.eye(core.shape[2], dtype=core.dtype).repeat(core.shape[0], 1, 1)
            else:
                Us[mu] = torch.eye(core.shape[1], dtype=core.dtype)
        # send non-orthogonality to the factor (tensor.py:1960-1984)
        if batch:
            Q, R = torch.linalg.qr(core.permute(0, 1, 3, 2).reshape(core.shape[0], -1, core.shape[2]))
            cores[mu] = Q.reshape(core.shape[0], core.shape[1], core.shape[3], -1).permute(0, 1, 3, 2)
        else:
            Q, R = torch.linalg.qr(core.permute(0, 2, 1).reshape(-1, core.shape[1]))
            cores[mu] = Q.reshape(core.shape[0], core.shape[2], -1).permute(0, 2, 1)
        Us[mu] = Us[mu] @ _t(R)  # tensor.py:1986
        left, right = truncated_svd(  # tensor.py:1989-1996
            Us[mu], eps=eps / math.sqrt(len(dim)), rmax=rmax[mu], left_ortho=True, algorithm=algorithm, batch=batch
        )
        Us[mu] = left.to(core.dtype)
        right = right.to(core.dtype)
        if batch:  # tensor.py:1999-2002
            cores[mu] = torch.einsum("bijk,baj->biak", cores[mu], right)
        else:
            cores[mu] = torch.einsum("ijk,aj->iak", cores[mu], right)
        if mu > 0:  # tensor.py:2005-2006
            right_orthogonalize(cores, mu, batch, Us)
    return cores, Us


def tucker_absorb(cores: Sequence[torch.Tensor], Us: Optional[Sequence[Optional[torch.Tensor]]], batch: bool = False) -> Cores:
    """Contract every factor into its core (what ``Tensor.torch()`` does per mode, tensor.py:1639-1687)."""
    out = []
    for k, c in enumerate(cores):
        U = None if Us is None else Us[k]
        if U is None:
            out.append(c)
        elif batch:
            out.append(torch.einsum("biak,bja->bijk", c, U))
        else:
            out.append(torch.einsum("iak,ja->ijk", c, U))
    return out


def tucker_to_dense(cores, Us, batch: bool = False) -> torch.Tensor:
    return tt_to_dense(tucker_absorb(cores, Us, batch), batch)


def relative_error_tt(gt: Sequence[torch.Tensor], approx: Sequence[torch.Tensor]) -> torch.Tensor:
    """metrics.py:135-151 between two compressed tensors (the <a,a>+<b,b>-2<a,b> formula)."""
    dotgt = tt_dot(gt, gt)
    return torch.sqrt((dotgt + tt_dot(approx, approx) - 2 * tt_dot(gt, approx)).clamp(0)) / torch.sqrt(dotgt.clamp(0))


def round_general(cores: Sequence[torch.Tensor], eps: float = 1e-14, algorithm: str = "svd"):
    """tensor.py:2085-2098 ``Tensor.round`` for a pure-TT input: TT rounding, then Tucker rounding with the
    remaining error budget.  Returns (cores, Us)."""
    copy = [c.clone() for c in cores]
    out = round_tt(cores, eps=eps, algorithm=algorithm)  # pure TT input: no factors to orthogonalise
    reached = relative_error_tt(copy, out)
    Us = [None] * len(out)
    if reached < eps:
        out, Us = round_tucker(out, None, (1 + eps) / (1 + reached.item()) - 1, algorithm=algorithm)
    return out, Us


def dense_to_tucker_tt(data: torch.Tensor, ranks_tucker=None, ranks_tt=None, eps=None, algorithm: str = "svd",
                       batch: bool = False):
    """tensor.py:401-408, 436-439: ``tn.Tensor(data, ranks_tucker=, ranks_tt=)`` / ``tn.Tensor(data, eps=)``."""
    cores = full_rank_tt(data, batch)
    Us = [None] * len(cores)
    if eps is not None:
        assert not batch
        return round_general(cores, eps, algorithm)
    if ranks_tucker is not None:
        cores, Us = round_tucker(cores, None, rmax=ranks_tucker, algorithm=algorithm, batch=batch)
    if ranks_tt is not None:
        cores = round_tt(cores, rmax=ranks_tt, algorithm=algorithm, batch=batch, Us=Us)
    return cores, Us


# --------------------------------------------------------------------------
# CP-ALS -- tensor.py:210-400, ``tn.Tensor(data, ranks_cp=R)``, non-batch,
# ranks_tucker=None.  A CP tensor is a list of 2-D factors [I_n, R].
# --------------------------------------------------------------------------
def cp_hosvd_init(data: torch.Tensor, R: int) -> Cores:
    """tensor.py:228-277: leading R eigenvectors of every mode Gram matrix (eigh ascending -> reversed);
    random completion when I_n < R (consumes the global torch RNG exactly like the reference)."""
    cores = []
    for n in range(data.dim()):
        gram = unfolding(data, n)
        gram = gram @ _t(gram)
        eigvals, eigvecs = torch.linalg.eigh(gram)
        reverse = torch.arange(len(eigvals) - 1, -1, -1)
        idx = torch.argsort(eigvals)[reverse[:R]]
        c = eigvecs[:, idx]
        if c.shape[1] < R:
            c = torch.cat((c, torch.randn(c.shape[0], R - c.shape[1], dtype=c.dtype)), dim=1)
        cores.append(c)
    return cores


def cp_to_dense(cores: Sequence[torch.Tensor]) -> torch.Tensor:
    """Dense tensor of a CP decomposition: sum_r prod_n cores[n][i_n, r] (tensor.py:1639-1687 on 2-D cores)."""
    acc = cores[0]  # [I_0, R]
    for c in cores[1:]:
        acc = torch.einsum("ar,ir->air", acc, c).reshape(-1, c.shape[1])
    return acc.sum(dim=1).reshape([c.shape[0] for c in cores])


def cp_als(data: torch.Tensor, R: int, max_iter: int = 25, tol: float = 1e-4, init: Optional[Cores] = None):
    """tensor.py:279-394.  Returns (cores, errors) -- ``errors[k]`` = relative error after sweep k."""
    N = data.dim()
    cores = cp_hosvd_init(data, R) if init is None else [c.clone() for c in init]
    data_norm = torch.norm(data)
    grams = [None] + [_t(cores[n]) @ cores[n] for n in range(1, N)]  # tensor.py:279-282
    errors = []
    for _ in range(max_iter):  # tensor.py:295
        for n in range(N):
            khatri = torch.ones(1, R, dtype=data.dtype)
            prod = torch.ones(R, R, dtype=data.dtype)
            for m in range(N - 1, -1, -1):  # tensor.py:328-334
                if m != n:
                    prod = prod * grams[m]
                    khatri = torch.einsum("ir,jr->ijr", cores[m], khatri).reshape(-1, R)
            unf = unfolding(data, n)
            unf_khatri_t = _t(unf @ khatri)  # MTTKRP, tensor.py:336-338
            cores[n] = _t(torch.linalg.lstsq(prod, unf_khatri_t).solution)  # tensor.py:339-341
            grams[n] = _t(cores[n]) @ cores[n]
        errors.append(torch.norm(data - cp_to_dense(cores)) / data_norm)  # tensor.py:373-379
        if len(errors) >= 2 and errors[-2] - errors[-1] < tol:  # tensor.py:380-381
            break
    return cores, errors


def cp_als_batch(data: torch.Tensor, R: int, max_iter: int = 25, tol: float = 1e-4, init: Optional[Cores] = None):
    """The ``batch=True`` branches of tensor.py:214-400 on ``data [B, I_1..I_N]``: HOSVD init per item (tensor.py:228-258),
    bmm sweeps, error of an iteration = MEAN of the items' relative errors (tensor.py:362-372), one convergence decision."""
    B, N = data.shape[0], data.dim() - 1
    if init is None:
        cores = []
        for n in range(N):
            gram = unfolding(data, n, batch=True)
            gram = gram @ _t(gram)
            eigvals, eigvecs = torch.linalg.eigh(gram)
            reverse = torch.arange(eigvals.shape[1] - 1, -1, -1)
            idx = torch.argsort(eigvals)[:, reverse[:R]]
            c = eigvecs[[[i] for i in range(len(idx))], :, idx].transpose(-1, -2)
            if c.shape[2] < R:
                c = torch.cat((c, torch.randn(c.shape[0], c.shape[1], R - c.shape[2], dtype=c.dtype)), dim=2)
            cores.append(c)
    else:
        cores = [c.clone() for c in init]
    norms = torch.sqrt(torch.sum(data**2, dim=list(range(1, data.dim()))))
    grams = [None] + [_t(cores[n]) @ cores[n] for n in range(1, N)]
    errors = []
    for _ in range(max_iter):
        for n in range(N):
            khatri = torch.ones(B, 1, R, dtype=data.dtype)
            prod = torch.ones(B, R, R, dtype=data.dtype)
            for m in range(N - 1, -1, -1):
                if m != n:
                    prod = prod * grams[m]
                    khatri = torch.einsum("bir,bjr->bijr", cores[m], khatri).reshape(B, -1, R)
            unf = unfolding(data, n, batch=True)
            cores[n] = _t(torch.linalg.lstsq(prod, _t(unf @ khatri)).solution)
            grams[n] = _t(cores[n]) @ cores[n]
        rec = torch.stack([cp_to_dense([c[i] for c in cores]) for i in range(B)])
        err = data - rec
        errors.append((torch.sqrt(torch.sum(err**2, dim=list(range(1, err.dim())))) / norms).mean())
        if len(errors) >= 2 and errors[-2] - errors[-1] < tol:
            break
    return cores, errors


def cp_on_tucker_core(data: torch.Tensor, R: int, ranks_tucker, init: Cores, max_iter: int = 25, tol: float = 1e-4,
                      algorithm: str = "svd"):
    """tensor.py:278-300 (non-batch): full-rank TT, ``round_tucker(rmax=ranks_tucker)``, ALS on the dense Tucker core
    (``tucker_core()``, tensor.py:1702-1715) from the given start (the reference draws ``randn(S_n, R)`` per mode).
    Returns (CP factors of the core, Tucker factors, errors)."""
    N = data.dim()
    rtk = list(ranks_tucker) if hasattr(ranks_tucker, "__len__") else [ranks_tucker] * N
    cores, Us = round_tucker(full_rank_tt(data), None, rmax=rtk, algorithm=algorithm)
    core = tt_to_dense(cores)
    fac, errors = cp_als(core, R, max_iter=max_iter, tol=tol, init=init)
    return fac, Us, errors


# --------------------------------------------------------------------------
# dense -> TT entry -- tensor.py:10-104 + ctor dense branch 401-408
# --------------------------------------------------------------------------
def full_rank_tt(data: torch.Tensor, batch: bool = False) -> Cores:
    """tensor.py:10-104: exact TT padded with identity cores (no compression)."""
    shape = list(data.shape[1:]) if batch else list(data.shape)
    B = data.shape[0] if batch else None
    N = len(shape)
    lead = (B,) if batch else ()

    def eye(n):
        I = torch.eye(n, dtype=data.dtype)
        return I.repeat(B, 1, 1) if batch else I

    resh = data.reshape(lead + (shape[0], -1))
    out: Cores = []
    for n in range(1, N):
        rows, cols = resh.shape[-2], resh.shape[-1]
        if rows < cols:  # tensor.py:32-64: emit an identity core
            out.append(eye(rows).reshape(lead + (rows // shape[n - 1], shape[n - 1], rows)))
            resh = resh.reshape(lead + (rows * shape[n], cols // shape[n]))
        else:  # tensor.py:65-96: emit the data, continue with an identity
            out.append(resh.reshape(lead + (rows // shape[n - 1], shape[n - 1], cols)))
            resh = eye(cols).reshape(lead + (cols * shape[n], cols // shape[n]))
    rows = resh.shape[-2]
    out.append(resh.reshape(lead + (rows // shape[N - 1], shape[N - 1], 1)))
    return out


def dense_to_tt(
    data: torch.Tensor,
    ranks_tt: Union[None, int, Sequence[int]] = None,
    algorithm: str = "svd",
    batch: bool = False,
) -> Cores:
    """tensor.py:401-408: ``tn.Tensor(data, ranks_tt=r)`` = full_rank_tt + round_tt(rmax=r)."""
    cores = full_rank_tt(data, batch)
    if ranks_tt is not None:
        cores = round_tt(cores, rmax=ranks_tt, algorithm=algorithm, batch=batch)
    return cores


# --------------------------------------------------------------------------
# helpers used by the parity harness (thin restatements)
# --------------------------------------------------------------------------
def tt_to_dense(cores: Sequence[torch.Tensor], batch: bool = False) -> torch.Tensor:
    """tensor.py:1639-1687 for pure TT cores: chain the cores left to right."""
    if batch:
        B = cores[0].shape[0]
        acc = cores[0].reshape(B, -1, cores[0].shape[-1])
        for c in cores[1:]:
            acc = torch.bmm(acc, c.reshape(B, c.shape[1], -1)).reshape(B, -1, c.shape[-1])
        return acc.sum(-1).reshape([B] + [c.shape[2] for c in cores])
    acc = cores[0].reshape(-1, cores[0].shape[-1])
    for c in cores[1:]:
        acc = (acc @ c.reshape(c.shape[0], -1)).reshape(-1, c.shape[-1])
    return acc.sum(-1).reshape([c.shape[1] for c in cores])


def tt_add(a: Sequence[torch.Tensor], b: Sequence[torch.Tensor], batch: bool = False) -> Cores:
    """tensor.py:445-668 for TT+TT: first core [a b], middle blockdiag, last [a; b]."""
    N = len(a)
    if N == 1:
        return [a[0] + b[0]]
    out = []
    for n in range(N):
        ca, cb = a[n], b[n]
        if n == 0:
            out.append(torch.cat([ca, cb], dim=-1))
        elif n == N - 1:
            out.append(torch.cat([ca, cb], dim=-3))
        else:
            za = torch.zeros(ca.shape[:-1] + (cb.shape[-1],), dtype=ca.dtype)
            zb = torch.zeros(cb.shape[:-1] + (ca.shape[-1],), dtype=ca.dtype)
            out.append(torch.cat([torch.cat([ca, za], -1), torch.cat([zb, cb], -1)], dim=-3))
    return out


def tt_mul(a: Sequence[torch.Tensor], b: Sequence[torch.Tensor]) -> Cores:
    """tensor.py:687-773 for two pure TT tensors: slice-wise Kronecker product ``_core_kron`` (tensor.py:2309-2320)."""
    out = []
    for ca, cb in zip(a, b):
        c = ca[:, None, :, :, None] * cb[None, :, :, None, :]
        out.append(c.reshape([ca.shape[0] * cb.shape[0], -1, ca.shape[-1] * cb.shape[-1]]))
    return out


def reduce_sum(ts: Sequence[Sequence[torch.Tensor]], eps: float = 0, rmax=None, algorithm: str = "svd"):
    """tools.py:460-512 ``tn.reduce(ts, operator.add, eps, rmax)`` for pure TT inputs: binary-counter tree, every
    intermediate sum goes through ``tn.round`` (TT rounding, then Tucker rounding with the left-over budget).
    Works on (cores, Us) pairs; returns (cores, Us)."""
    def add(x, y):
        return tt_add(tucker_absorb(*x), tucker_absorb(*y)), None

    def rnd(x):
        cores, _ = x
        copy = [c.clone() for c in cores]
        out = round_tt(cores, eps=eps, rmax=rmax, algorithm=algorithm)
        reached = relative_error_tt(copy, out)
        Us = [None] * len(out)
        if reached < eps:
            rm = None if rmax is None else rmax
            out, Us = round_tucker(out, None, (1 + eps) / (1 + reached.item()) - 1, rmax=rm, algorithm=algorithm)
        return out, Us

    d = dict()
    for elem in ts:
        elem = (list(elem), None)
        climb = 0
        while climb in d:
            elem = rnd(add(d[climb], elem))
            d.pop(climb)
            climb += 1
        d[climb] = elem
    keys = list(d.keys())
    result = d[keys[0]]
    for key in keys[1:]:
        result = rnd(add(result, d[key]))
    return result


def tt_scale(cores: Sequence[torch.Tensor], s: float) -> Cores:
    out = [c.clone() for c in cores]
    out[0] = out[0] * s
    return out


def _tt_random(fn, shape, ranks, dtype, batch_size=None):
    N = len(shape)
    if not hasattr(ranks, "__len__"):
        ranks = [ranks] * (N - 1)
    r = [1] + list(ranks) + [1]
    lead = () if batch_size is None else (batch_size,)
    return [fn(lead + (r[n], shape[n], r[n + 1]), dtype=dtype) for n in range(N)]


def tt_randn(shape, ranks, dtype=torch.float64, batch_size=None) -> Cores:
    """create.py:60-65 / 210-357 for pure TT: i.i.d. N(0,1) cores, core by core."""
    return _tt_random(torch.randn, tuple(shape), ranks, dtype, batch_size)


def tt_rand(shape, ranks, dtype=torch.float64, batch_size=None) -> Cores:
    """create.py (rand): i.i.d. U[0,1) cores."""
    return _tt_random(torch.rand, tuple(shape), ranks, dtype, batch_size)


def tt_ranks(cores: Sequence[torch.Tensor]) -> List[int]:
    """tensor.py:861-883."""
    return [cores[0].shape[-3]] + [c.shape[-1] for c in cores]


def tt_dot(a: Sequence[torch.Tensor], b: Sequence[torch.Tensor]) -> torch.Tensor:
    """metrics.py:28-116 (k = N, no Tucker factors): running Lprod contraction."""
    L = torch.ones(b[0].shape[0], a[0].shape[0], dtype=a[0].dtype)
    for ca, cb in zip(a, b):
        U = torch.einsum("sr,rai->sai", L, ca)
        L = left_unfolding(cb).t() @ left_unfolding(U)
    return L.sum()


def tt_norm(a: Sequence[torch.Tensor]) -> torch.Tensor:
    """metrics.py:469-478."""
    return torch.sqrt(torch.clamp(tt_dot(a, a), min=0))


def bond_singular_values(cores: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """Gauge-invariant fingerprint of a TT: singular values of every bond.

    Left-orthogonalises a float64 copy (tensor.py:1881-1909) and reads the
    singular values of the right unfolding while sweeping back (the quantities
    round.py:96 computes), without truncating.
    """
    cs = [c.double().clone() for c in cores]
    N = len(cs)
    orthogonalize(cs, N - 1)
    out = []
    for mu in range(N - 1, 0, -1):
        M = right_unfolding(cs[mu])
        U, s, Vh = torch.linalg.svd(M, full_matrices=False)
        out.append(s)
        cs[mu] = Vh.reshape(-1, cs[mu].shape[1], cs[mu].shape[2])
        cs[mu - 1] = cs[mu - 1] @ (U * s)
    return out[::-1]


def gauge_align(ref: Sequence[torch.Tensor], ours: Sequence[torch.Tensor]) -> Cores:
    """Fix the per-bond +-1 sign gauge of ``ours`` against ``ref``.

    For bond k the rows of the right unfolding of core k (= right singular
    vectors) are compared; our core k rows and core k-1 columns are multiplied
    by ``sign(<row_j(ref), row_j(ours)>)``.  Valid when both trains have the
    same ranks and separated spectra.
    """
    out = [c.clone() for c in ours]
    for k in range(len(out) - 1, 0, -1):
        a = right_unfolding(ref[k]).double()
        b = right_unfolding(out[k]).double()
        s = torch.sign((a * b).sum(-1))
        s[s == 0] = 1
        s = s.to(out[k].dtype)
        out[k] = out[k] * s[:, None, None]
        out[k - 1] = out[k - 1] * s[None, None, :]
    return out


# --------------------------------------------------------------------------
# consumers next to the path: tools.shift_mode, matrix.TTMatrix
# --------------------------------------------------------------------------
def shift_mode(cores: Sequence[torch.Tensor], n: int, shift: int, eps=1e-3, algorithm: str = "svd") -> Cores:
    """tools.py:650-697 on pure TT cores (non-batch): orthogonalise at ``n`` (tools.py:667), then per exchange contract
    the two cores with swapped modes (``einsum("iaj,jbk->ibak")``, tools.py:680-681) and split by truncated_svd with
    ``eps / sqrt(|shift|)`` or, for ``eps == 'same'``, ``eps=0, rmax=R2`` (tools.py:682-692)."""
    cores = [c.clone() for c in cores]
    N = len(cores)
    assert 0 <= n + shift < N
    if shift == 0:
        return cores
    orthogonalize(cores, n)
    sign = 1 if shift > 0 else -1
    for i in range(n, n + shift, sign):
        c1, c2, left_ortho = (i, i + 1, True) if sign == 1 else (i - 1, i, False)
        R1, I1, R2 = cores[c1].shape
        _, I2, R3 = cores[c2].shape
        sc = torch.einsum("iaj,jbk->ibak", cores[c1], cores[c2]).reshape(R1 * I2, I1 * R3)
        if isinstance(eps, str):
            if eps != "same":
                raise ValueError("Relative error '{}' not recognized".format(eps))
            left, right = truncated_svd(sc, eps=0, rmax=R2, left_ortho=left_ortho, algorithm=algorithm)
        elif eps >= 0:
            left, right = truncated_svd(sc, eps=eps / math.sqrt(abs(shift)), left_ortho=left_ortho, algorithm=algorithm)
        else:
            raise ValueError("Relative error '{}' not recognized".format(eps))
        r = left.shape[1]
        cores[c1] = left.reshape(R1, I2, r)
        cores[c2] = right.reshape(r, I1, R3)
    return cores


def ttmatrix_cores(M: torch.Tensor, ranks: Sequence[int], input_dims: Sequence[int], output_dims: Sequence[int],
                   algorithm: str = "svd") -> Cores:
    """matrix.py:59-111 (non-batch): reshape to i_0..i_{d-1} o_0..o_{d-1}, interleave to (i_k o_k)_k, decompose with
    ``tn.Tensor(tensor, ranks_tt=ranks)`` and view every core as ``[r, i_k, o_k, r']``."""
    d = len(input_dims)
    tensor = M.reshape(list(input_dims) + list(output_dims))
    perm = [k + off for k in range(d) for off in (0, d)]
    tensor = tensor.permute(perm).reshape([input_dims[k] * output_dims[k] for k in range(d)])
    cores = dense_to_tt(tensor, list(ranks), algorithm=algorithm)
    return [c.reshape(c.shape[0], input_dims[k], output_dims[k], c.shape[-1]) for k, c in enumerate(cores)]


def ttmatrix_to_dense(cores4: Sequence[torch.Tensor]) -> torch.Tensor:
    """matrix.py:113-151: contract the flattened train and undo the interleaving."""
    d = len(cores4)
    idims, odims = [c.shape[1] for c in cores4], [c.shape[2] for c in cores4]
    dense = tt_to_dense([c.reshape(c.shape[0], -1, c.shape[-1]) for c in cores4])
    dense = dense.reshape([x for k in range(d) for x in (idims[k], odims[k])])
    rows, cols = math.prod(idims), math.prod(odims)
    return dense.permute([2 * k for k in range(d)] + [2 * k + 1 for k in range(d)]).reshape(rows, cols)


def ttmatrix_trace(cores4: Sequence[torch.Tensor]) -> torch.Tensor:
    """matrix.py:160-175: ``factor = einsum("i,iaaj->j", factor, core)`` from ``ones(1)``."""
    factor = torch.ones(1, dtype=cores4[0].dtype)
    for c in cores4:
        factor = torch.einsum("i,iaaj->j", factor, c)
    return factor[0]
