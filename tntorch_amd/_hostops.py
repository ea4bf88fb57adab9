"""Host (CPU tensor) mirror of the hot path, on batch-normalised ``[B, ...]`` tensors.

This is the "plumbing" path of BASELINE config C0 (``tn.Tensor(torch.randn(16,16,16,16))
.round_tt(rmax=4)`` on CPU PyTorch): same operator sequence as the reference
(torch.linalg.qr / svd / eigh on the CPU), so CPU results agree with the reference to
round-off.  It is selected ONLY for CPU tensors; device tensors never come here
(``_dispatch.ops_for`` raises instead of falling back).
"""

from __future__ import annotations

import math
import time
from typing import List, Sequence, Tuple

import torch

INT32_MAX = 2**31 - 1
VERBOSE = False   # set by the API layer around a call with verbose=True: the reference's per-stage timing lines (round.py:95-117, 163-185; tensor.py:2032-2035)


def _t(M):
    return M.transpose(-1, -2)


def qr(A3: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """tensor.py:1816 (reduced QR).  A single matrix goes through the 2-D LAPACK path exactly as in
    the reference's non-batch mode, so CPU results match the reference bit for bit."""
    if A3.shape[0] == 1:
        Q, R = torch.linalg.qr(A3[0])
        return Q[None], R[None]
    return torch.linalg.qr(A3)


def _mm(A, B):
    if A.dim() == 3 and A.shape[0] == 1:
        return (A[0] @ B[0])[None]
    return A @ B


def truncated_svd(M3, delta, eps, rmax, left_ortho, algorithm, batch):
    """round.py:52-187 with the batch dim always present ([B, m, n]; B == 1 when not batch)."""
    if not batch:  # same 2-D operator calls as the reference's non-batch mode
        left, M2 = _truncated_svd(M3, delta, eps, rmax, left_ortho, algorithm, False, squeeze=True)
        return left, M2
    return _truncated_svd(M3, delta, eps, rmax, left_ortho, algorithm, True, squeeze=False)


def _truncated_svd(M3, delta, eps, rmax, left_ortho, algorithm, batch, squeeze):
    if squeeze:
        M3 = M3[0]
        lead = ()
    else:
        lead = (M3.shape[0],)
    if delta is None and eps is not None:  # round.py:79-80
        delta = eps * torch.norm(M3).item()
    if delta is None:
        delta = 0
    if rmax is None:
        rmax = INT32_MAX
    m, n = M3.shape[-2], M3.shape[-1]

    start = time.time()
    if algorithm == "svd":  # round.py:94-100
        U, sig = torch.linalg.svd(M3)[:2]
        side = "left"
        if VERBOSE:
            print("Time (SVD):", time.time() - start)
    else:  # round.py:101-135
        if m <= n:
            gram, side = M3 @ _t(M3), "left"
        else:
            gram, side = _t(M3) @ M3, "right"
        if VERBOSE:
            print("Time (gram):", time.time() - start)
        start = time.time()
        w, U = torch.linalg.eigh(gram)
        if VERBOSE:
            print("Time (symmetric EIG):", time.time() - start)
        w = torch.where(w < 0, torch.zeros_like(w) + 1e-8, w)
        sig = torch.sqrt(w)
        sig, idx = torch.sort(sig, dim=-1, descending=True)
        U = torch.gather(U, -1, idx[..., None, :].expand(U.shape))

    if sig.max() < 1e-13:  # round.py:137-145 (kept on M's device/dtype)
        return M3.new_zeros((1,) * squeeze + lead + (m, 1)), M3.new_zeros((1,) * squeeze + lead + (1, n))

    S = sig**2
    k = S.shape[-1]
    if batch:  # round.py:149-150
        rank = max(1, int(min(rmax, k)))
    else:  # round.py:152-158
        tail = torch.cumsum(torch.flip(S, [0]), dim=0) <= delta**2
        where = torch.where(tail)[0]
        if len(where) == 0:
            rank = max(1, int(min(rmax, k)))
        else:
            rank = max(1, int(min(rmax, k - 1 - int(where[-1]))))

    left = U[..., :rank]
    sr = sig[..., :rank].to(M3.dtype)
    start = time.time()
    if side == "left":  # round.py:164-172
        if left_ortho:
            M2 = _t(left) @ M3
        else:
            M2 = (1.0 / sr)[..., None] * _t(left) @ M3
            left = left * sr[..., None, :]
    else:  # round.py:173-182
        if left_ortho:
            newleft = M3 @ (left * (1.0 / sr)[..., None, :])
            M2 = _t(left * sr[..., None, :])
            left = newleft
        else:
            newleft = M3 @ left
            M2 = _t(left)
            left = newleft
    if VERBOSE:
        print("Time (product):", time.time() - start)
    if squeeze:
        return left[None], M2[None]
    return left, M2


def mode_mul(core4: torch.Tensor, M3: torch.Tensor) -> torch.Tensor:
    """[B, r0, S, r1] x_2 [B, a, S] -> [B, r0, a, r1] (the einsum of tensor.py:1790-1798, 1999-2002)."""
    if core4.shape[0] == 1:
        return torch.einsum("ijk,aj->iak", core4[0], M3[0])[None]
    return torch.einsum("bijk,baj->biak", core4, M3)


def merge_swap(c1: torch.Tensor, c2: torch.Tensor) -> torch.Tensor:
    """Two neighbouring cores [B, R1, I1, R2], [B, R2, I2, R3] contracted over the bond with their modes exchanged:
    ``einsum("iaj,jbk->ibak")`` flattened to [B, R1*I2, I1*R3] (tools.py:680-681)."""
    sc = torch.einsum("ziaj,zjbk->zibak", c1, c2)
    return sc.reshape(sc.shape[0], sc.shape[1] * sc.shape[2], sc.shape[3] * sc.shape[4])


def diag_sum(core5: torch.Tensor) -> torch.Tensor:
    """[B, r0, a, a, r1] -> [B, r0, r1]: sum of the slices on the diagonal of the two mode axes (matrix.py:160-175)."""
    return torch.einsum("ziaaj->zij", core5)


def factor_orthogonalize(c: List[torch.Tensor], Us, mu: int) -> None:
    """tensor.py:1771-1798: QR of the Tucker factor [B, I, S], R pushed into the core."""
    if Us is None or Us[mu] is None:
        return
    Q, R = qr(Us[mu])
    Us[mu] = Q
    c[mu] = mode_mul(c[mu], R)


def left_orthogonalize(c: List[torch.Tensor], mu: int, Us=None) -> torch.Tensor:
    """tensor.py:1800-1833 on [B, r0, I, r1] cores."""
    factor_orthogonalize(c, Us, mu)
    Bt, r0, I, r1 = c[mu].shape
    Q, R = qr(c[mu].reshape(Bt, r0 * I, r1))
    k = Q.shape[2]
    c[mu] = Q.reshape(Bt, r0, I, k)
    nxt = c[mu + 1]
    c[mu + 1] = _mm(R, nxt.reshape(Bt, nxt.shape[1], -1)).reshape(Bt, k, nxt.shape[2], nxt.shape[3])
    return R


def right_orthogonalize(c: List[torch.Tensor], mu: int, Us=None) -> torch.Tensor:
    """tensor.py:1835-1879."""
    factor_orthogonalize(c, Us, mu)
    Bt, r0, I, r1 = c[mu].shape
    Q, Lt = qr(_t(c[mu].reshape(Bt, r0, I * r1)))
    Q, L = _t(Q), _t(Lt)
    k = Q.shape[1]
    c[mu] = Q.reshape(Bt, k, I, r1)
    prev = c[mu - 1]
    c[mu - 1] = _mm(prev.reshape(Bt, prev.shape[1] * prev.shape[2], r0), L).reshape(Bt, prev.shape[1], prev.shape[2], k)
    return L


def round_tt(cores4: Sequence[torch.Tensor], eps, rmax, algorithm, batch, Us=None) -> List[torch.Tensor]:
    """tensor.py:2008-2083 (``Us``: Tucker factors, orthogonalised in place by the L2R sweep)."""
    c = list(cores4)
    N = len(c)
    start = time.time()
    for mu in range(N - 1):
        left_orthogonalize(c, mu, Us)
    if VERBOSE:
        print("Orthogonalization time:", time.time() - start)
    if batch:
        delta = None
    else:
        delta = (eps / max(1.0, math.sqrt(N - 1))) * float(torch.norm(c[-1]).double().item())
    for mu in range(N - 1, 0, -1):
        Bt, R, I, rn = c[mu].shape
        left, right = truncated_svd(c[mu].reshape(Bt, R, I * rn), delta, None, rmax[mu - 1], False, algorithm, batch)
        left, right = left.to(c[mu].dtype), right.to(c[mu].dtype)
        r = right.shape[1]
        c[mu] = right.reshape(Bt, r, I, rn)
        prev = c[mu - 1]
        c[mu - 1] = _mm(prev.reshape(Bt, prev.shape[1] * prev.shape[2], R), left).reshape(Bt, prev.shape[1], prev.shape[2], r)
    return c


def round_tucker(cores4: Sequence[torch.Tensor], Us, eps, rmax, ndims, algorithm, batch):
    """tensor.py:1911-2006 on [B, r0, S, r1] cores and [B, I, S] factors; returns (cores, Us)."""
    c = list(cores4)
    N = len(c)
    Us = [None] * N if Us is None else list(Us)
    for i in range(N - 1):  # orthogonalize(-1), tensor.py:1944
        left_orthogonalize(c, i, Us)
    for mu in range(N - 1, -1, -1):
        Bt, r0, S, r1 = c[mu].shape
        if Us[mu] is None:  # tensor.py:1946-1958
            Us[mu] = torch.eye(S, dtype=c[mu].dtype, device=c[mu].device).repeat(Bt, 1, 1)
        Q, R = qr(c[mu].permute(0, 1, 3, 2).reshape(Bt, r0 * r1, S))  # tensor.py:1960-1984
        c[mu] = Q.reshape(Bt, r0, r1, Q.shape[2]).permute(0, 1, 3, 2)
        Us[mu] = _mm(Us[mu], _t(R))  # tensor.py:1986
        left, right = truncated_svd(Us[mu], None, eps / math.sqrt(ndims), rmax[mu], True, algorithm, batch)
        Us[mu] = left.to(c[mu].dtype)
        c[mu] = mode_mul(c[mu], right.to(c[mu].dtype))  # tensor.py:1999-2002
        if mu > 0:
            right_orthogonalize(c, mu, Us)
    return c, Us


def absorb_factors(cores4: Sequence[torch.Tensor], Us) -> List[torch.Tensor]:
    """Contract every Tucker factor into its core (what tensor.py:1639-1687 does per mode)."""
    return [c if U is None else mode_mul(c, U) for c, U in zip(cores4, Us)]


def dense_tucker_tt(X: torch.Tensor, ranks_tucker, ranks_tt, algorithm, batch):
    """tensor.py:401-408: full-rank TT, ``round_tucker(rmax=ranks_tucker)``, ``round_tt(rmax=ranks_tt)``."""
    c = full_rank_tt(X)
    N = len(c)
    c, Us = round_tucker(c, None, 1e-14, ranks_tucker, N, algorithm, batch)
    if ranks_tt is not None:
        c = round_tt(c, 1e-14, ranks_tt, algorithm, batch, Us)
    return c, Us


def full_rank_tt(X: torch.Tensor) -> List[torch.Tensor]:
    """tensor.py:10-104 on a batch-normalised dense tensor [B, I_1..I_N] (views/eye only)."""
    Bt = X.shape[0]
    shape = list(X.shape[1:])
    N = len(shape)

    def eye(n):
        return torch.eye(n, dtype=X.dtype, device=X.device).repeat(Bt, 1, 1)

    resh = X.reshape(Bt, shape[0], -1)
    out = []
    for n in range(1, N):
        rows, cols = resh.shape[1], resh.shape[2]
        if rows < cols:
            out.append(eye(rows).reshape(Bt, rows // shape[n - 1], shape[n - 1], rows))
            resh = resh.reshape(Bt, rows * shape[n], cols // shape[n])
        else:
            out.append(resh.reshape(Bt, rows // shape[n - 1], shape[n - 1], cols))
            resh = eye(cols).reshape(Bt, cols * shape[n], cols // shape[n])
    rows = resh.shape[1]
    out.append(resh.reshape(Bt, rows // shape[N - 1], shape[N - 1], 1))
    return out


def dense_tt_svd(X: torch.Tensor, eps, rmax, algorithm, batch) -> List[torch.Tensor]:
    """tensor.py:401-408 exactly as the reference does it: full-rank TT, then round_tt."""
    return round_tt(full_rank_tt(X), eps, rmax, algorithm, batch)


# ---------------------------------------------------------------------------------------------- consumers
def decompress(c: Sequence[torch.Tensor]) -> torch.Tensor:
    """tensor.py:1639-1687 for TT cores [B, r0, I, r1] -> [B, I_1, ..., I_N]."""
    Bt = c[0].shape[0]
    acc = c[0].reshape(Bt, -1, c[0].shape[-1])
    for core in c[1:]:
        acc = torch.bmm(acc, core.reshape(Bt, core.shape[1], -1)).reshape(Bt, -1, core.shape[-1])
    # ranks_tt[0] and ranks_tt[-1] may exceed 1: the reference sums the boundary indices away
    r0 = c[0].shape[1]
    acc = acc.reshape(Bt, r0, -1, c[-1].shape[-1]).sum(dim=(1, 3))
    return acc.reshape([Bt] + [core.shape[2] for core in c])


def dot(c1: Sequence[torch.Tensor], c2: Sequence[torch.Tensor]) -> torch.Tensor:
    """metrics.py:28-116 (k = N, TT cores only) on cores [1, r, I, r']."""
    a0, b0 = c1[0][0], c2[0][0]
    L = torch.ones([b0.shape[0], a0.shape[0]], device=a0.device, dtype=a0.dtype)
    for a, b in zip(c1, c2):
        a, b = a[0], b[0]
        U = torch.einsum("sr,rai->sai", L, a)
        L = b.reshape(-1, b.shape[-1]).t() @ U.reshape(-1, U.shape[-1])
    return torch.sum(L)


def dense_dot(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return a.flatten().dot(b.flatten())


def dense_norm(a: torch.Tensor) -> torch.Tensor:
    return torch.norm(a)


def dense_dist(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return torch.dist(a, b)


# ---------------------------------------------------------------------------------------------- CP-ALS
def cp_als(X: torch.Tensor, R: int, max_iter: int, tol: float, verbose: bool = False, batch: bool = False, init=None):
    """tensor.py:210-400, the reference's operator sequence on the CPU.  ``init=None``: HOSVD initialisation
    (tensor.py:228-277); otherwise the given factors (CP on a Tucker core starts from ``randn``, tensor.py:282-300).
    ``batch``: X is [B, I_1..I_N], factors are [B, I_n, R], ONE convergence decision on the batch-mean error."""
    if batch:
        return _cp_als_batch(X, R, max_iter, tol, verbose, init)
    N = X.dim()

    def unf(n):
        return X.permute([n] + list(range(n)) + list(range(n + 1, N))).reshape(X.shape[n], -1)

    if init is not None:
        A = list(init)
    else:
        A = []
        for n in range(N):  # tensor.py:228-277
            g = unf(n)
            g = g @ _t(g)
            w, V = torch.linalg.eigh(g)
            reverse = torch.arange(len(w) - 1, -1, -1)
            idx = torch.argsort(w)[reverse[:R]]
            c = V[:, idx]
            if c.shape[1] < R:
                c = torch.cat((c, torch.randn(c.shape[0], R - c.shape[1], dtype=c.dtype, device=c.device)), dim=1)
            A.append(c)
    xnorm = torch.norm(X)
    grams = [None] + [_t(A[n]) @ A[n] for n in range(1, N)]
    errors = []
    for it in range(max_iter):
        for n in range(N):
            khatri = torch.ones(1, R, dtype=X.dtype, device=X.device)
            prod = torch.ones(R, R, dtype=X.dtype, device=X.device)
            for m in range(N - 1, -1, -1):
                if m != n:
                    prod *= grams[m]
                    khatri = torch.reshape(torch.einsum("ir,jr->ijr", (A[m], khatri)), [-1, R])
            A[n] = _t(torch.linalg.lstsq(prod, _t(unf(n) @ khatri)).solution)
            grams[n] = _t(A[n]) @ A[n]
        acc = A[0]
        for c in A[1:]:
            acc = torch.einsum("ar,ir->air", acc, c).reshape(-1, R)
        errors.append(float(torch.norm(X - acc.sum(dim=1).reshape(X.shape)) / xnorm))
        if verbose:
            print("iter: {} | eps: {:.8f}".format(it, errors[-1]))
        if len(errors) >= 2 and errors[-2] - errors[-1] < tol:
            break
    return A, errors


def _cp_als_batch(X: torch.Tensor, R: int, max_iter: int, tol: float, verbose: bool, init):
    """The ``batch=True`` branches of tensor.py:214-400: every product is a ``bmm`` over the leading axis; the error of an
    iteration is the MEAN of the items' relative errors (tensor.py:362-372) and convergence is decided once for all."""
    Bt, N = X.shape[0], X.dim() - 1

    def unf(n):
        return X.permute([0, n + 1] + list(range(1, n + 1)) + list(range(n + 2, N + 1))).reshape(Bt, X.shape[n + 1], -1)

    if init is not None:
        A = list(init)
    else:
        A = []
        for n in range(N):  # tensor.py:228-258
            g = unf(n)
            g = g @ _t(g)
            w, V = torch.linalg.eigh(g)
            reverse = torch.arange(w.shape[1] - 1, -1, -1)
            idx = torch.argsort(w)[:, reverse[:R]]
            c = V[[[i] for i in range(len(idx))], :, idx].transpose(-1, -2)
            if c.shape[2] < R:
                c = torch.cat((c, torch.randn(c.shape[0], c.shape[1], R - c.shape[2], dtype=c.dtype, device=c.device)), dim=2)
            A.append(c)
    xnorms = torch.sqrt(torch.sum(X**2, dim=list(range(1, X.dim()))))
    grams = [None] + [_t(A[n]) @ A[n] for n in range(1, N)]
    errors = []
    for it in range(max_iter):
        for n in range(N):
            khatri = torch.ones(Bt, 1, R, dtype=X.dtype, device=X.device)
            prod = torch.ones(Bt, R, R, dtype=X.dtype, device=X.device)
            for m in range(N - 1, -1, -1):
                if m != n:
                    prod *= grams[m]
                    khatri = torch.reshape(torch.einsum("bir,bjr->bijr", (A[m], khatri)), [Bt, -1, R])
            A[n] = _t(torch.linalg.lstsq(prod, _t(unf(n) @ khatri)).solution)
            grams[n] = _t(A[n]) @ A[n]
        acc = A[0]
        for c in A[1:]:
            acc = torch.einsum("bar,bir->bair", acc, c).reshape(Bt, -1, R)
        err = X - acc.sum(dim=2).reshape(X.shape)
        errors.append(float((torch.sqrt(torch.sum(err**2, dim=list(range(1, err.dim())))) / xnorms).mean()))
        if verbose:
            print("iter: {} | eps: {:.8f}".format(it, errors[-1]))
        if len(errors) >= 2 and errors[-2] - errors[-1] < tol:
            break
    return A, errors


# ---------------------------------------------------------------------------------------------- producers
def core_kron(a4: torch.Tensor, b4: torch.Tensor) -> torch.Tensor:
    """tensor.py:2309-2320 ``_core_kron`` (batch form) on [B, r, I, r'] cores."""
    c = a4[:, :, None, :, :, None] * b4[:, None, :, :, None, :]
    return c.reshape([a4.shape[0], a4.shape[1] * b4.shape[1], -1, a4.shape[-1] * b4.shape[-1]])


def scale(x: torch.Tensor, value) -> torch.Tensor:
    return x * value
