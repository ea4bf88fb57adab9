"""Device-side (HIP) implementation of the TT sweeps: orchestration of the C-ABI kernels.

Everything here works on batch-normalised tensors (``[B, ...]``; the non-batch API adds
a leading 1).  Arithmetic on tensor data is done by ``libttround_hip.so``; torch allocates, reshapes,
slices, copies and fills (identity / zero blocks, seeded noise for the blocked QR), and reads back the
few scalars that steer control flow (ranks in eps mode, convergence flags of the block-Jacobi driver).

Algorithms (reference call sites in brackets):

* ``qr``            Householder TSQR; block Gram-Schmidt around it above 64 columns  [tensor.py:1816]
* ``truncate``      truncated SVD of ``M`` [round.py:52-187]:
    - ``algorithm='eig'``: Gram -> tridiagonal-QL eigh (with the reference's 1e-8 clamp) ->
      rank rule -> projection; one pass, Gram accuracy (sigma resolved down to
      ~sqrt(eps)*sigma_max), exactly what round.py:101-135 computes.
    - ``algorithm='svd'``: two Gram passes (tridiagonal QL, then Jacobi).  Pass 1 rotates ``M`` into rows
      (columns) that are orthogonal down to the accuracy of a float Gram matrix; pass 2
      re-computes the Gram matrix of the ROTATED matrix, whose small rows are now formed
      from small numbers (no cancellation against sigma_max^2), so every sigma comes out
      with absolute error O(eps * sigma_max) -- the accuracy class of LAPACK gesdd
      [round.py:96] -- instead of the O(sqrt(eps) * sigma_max) of a single Gram pass, and
      without ever forming the (I r)x(I r) right factor the reference throws away.
* ``round_tt``      L2R QR sweep + R2L truncation sweep            [tensor.py:2008-2083]
* ``dense_tt_svd``  right-to-left TT-SVD on the dense unfoldings; mathematically equal to
                    the reference's identity-padded ``_full_rank_tt`` + ``round_tt``
                    [tensor.py:10-104, 401-408], but feasible at scale.
* ``eigh_block_jacobi``  eigenproblems above one workgroup (n up to 4096 fp32 / 2048 fp64)  [round.py:96, 115]
* ``round_tucker`` / ``dense_tucker_tt``  Tucker rounding, ST-HOSVD entry   [tensor.py:1911-2006, 401-408]
* ``cp_als``        CP-ALS with a fused MTTKRP                                [tensor.py:210-400]
* ``decompress`` / ``dot`` / ``core_kron``  consumers and producers     [tensor.py:1639-1687, metrics.py:28-116,
                    tensor.py:2309-2320]
"""

from __future__ import annotations

import math
import os
import time
from typing import List, Optional, Sequence, Tuple

import torch

from . import _hip

INT32_MAX = 2**31 - 1
# set by the API layer around a call with verbose=True: the reference's per-stage timing lines (round.py:95-117, 163-185;
# tensor.py:2032-2035).  Every stage boundary then synchronises the device (a diagnostic mode), and a batch runs on one stream.
VERBOSE = False


class _Stage:
    """Stage timer of the verbose mode (no-op otherwise)."""

    def __init__(self):
        self.t = None
        if VERBOSE:
            torch.cuda.synchronize()
            self.t = time.time()

    def lap(self, label: str) -> None:
        if self.t is not None:
            torch.cuda.synchronize()
            now = time.time()
            print(label, now - self.t)
            self.t = now



def _rank_cap(rmax: Optional[int], k: int) -> int:
    if rmax is None:
        rmax = INT32_MAX
    return max(1, int(min(int(rmax), k)))


def qr(A3: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Reduced QR of [B, m, n]: the TSQR kernel up to its column limit, block Gram-Schmidt around it above."""
    if A3.shape[2] <= _hip.max_qr_cols(A3.dtype):
        return _hip.qr(A3)
    return _qr_blocked(A3)


def _qr_blocked(A: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """QR for more columns than one TSQR panel holds (TT ranks > 64): block classical Gram-Schmidt with
    re-orthogonalisation (BCGS2) over 64-column panels -- every panel is projected against the finished
    Q (two MFMA GEMMs), factored by the Householder TSQR kernel, and the normalised panel is projected and
    factored once more ("twice is enough"; also keeps Q orthonormal when the panel is rank deficient).
    Wide inputs (m < n): Q from the leading m x m block, R = [R_L | Q^T A_R] (what geqrf returns)."""
    Bt, m, n = A.shape
    if m < n:
        Q, RL = qr(A[:, :, :m].contiguous())
        return Q, torch.cat([RL, _hip.gemm(Q, A[:, :, m:], transA=True)], dim=2)
    pw = _hip.max_qr_cols(A.dtype)
    # Exactly dependent / exactly zero columns (block-structured sums such as t + t have them) leave an exactly
    # zero remainder after the projection; the Householder kernel then completes the panel with unit vectors
    # e_0, e_1, ... -- the SAME ones in every such panel, and already inside span(Q).  An item with a panel whose
    # remainder COLLAPSES (some |R_ii| of the projected panel <= 32 eps x that column's norm before the projection) is
    # therefore factored again with a perturbation of EVERY panel at the rounding level of the item (8 eps * rms(A[b]),
    # seeded, deterministic): zero remainders become generic directions, which the two projection passes make orthogonal
    # to the finished Q; A = Q R still holds to O(eps ||A[b]||).  (All panels, not only the collapsing ones: with an exactly
    # block-structured input the unperturbed leading panels have exactly zero rows, a later panel's rounding-level remainder
    # then lies entirely inside the span that is already taken, and Q loses orthogonality -- measured 2.7e-13 against 4.9e-15 in
    # fp64 on a 136 x 136 rank-68 unfolding.)  Items that keep their rank in every panel -- every full-rank input -- are
    # factored exactly as they are (rounds 1-2 perturbed everything); in a mixed batch they come out of the second run
    # bit-identical (their perturbation is scaled by zero).  Per item: every item is first brought to ||A[b]|| in
    # [0.5, 1) by an exact power of two (ttr_pow2_normalize; given back to R at the end), so one delta serves the whole
    # batch and an item 1e-6 times smaller than its neighbours is not swamped by their noise level.  The collapse test costs
    # ONE readback (control flow; this path only exists for TT ranks above the kernel's 64 columns).
    A, a_exp = _hip.pow2_normalize(A)
    eps = torch.finfo(A.dtype).eps
    delta = 8.0 * eps / math.sqrt(max(1, m * n))
    # (an all-zero item keeps exponent 0: its perturbation is scaled down by 1e-20 -- still generic directions for the
    # panels, the QR kernel factors every block at its own exponent, but R comes back at 1e-28 instead of 1e-8.  [Bt]-element
    # masks and the [Bt, w] collapse test: the only torch arithmetic here.)
    live = torch.sign(_hip.norm(A.reshape(Bt, -1))).clamp_min(1e-20)
    gen = torch.Generator(device=A.device)

    Bt_all, sel = Bt, None   # (``sel``: positions of the items being refactored inside the full batch)

    def factor(perturb):
        """BCGS2 over the panels; ``perturb``: None or the [Bt] scale of the items' perturbation.  -> Q, R, [Bt] collapse flags."""
        R = torch.zeros((Bt, n, n), dtype=A.dtype, device=A.device)
        Q = torch.empty((Bt, m, n), dtype=A.dtype, device=A.device)
        collapsed = torch.zeros(Bt, dtype=torch.bool, device=A.device)
        for j0 in range(0, n, pw):
            j1 = min(j0 + pw, n)
            w = j1 - j0
            W0 = A[:, :, j0:j1]
            W = W0.contiguous()
            if perturb is not None:
                gen.manual_seed(0x5EED + j0)
                # (drawn for the WHOLE batch and indexed by the items' positions: an item's perturbation does not depend on which
                # of its neighbours collapsed, so the same tensor rounds bit-identically in any batch)
                noise = torch.randn((Bt_all, m, w), dtype=A.dtype, device=A.device, generator=gen)
                if sel is not None:
                    noise = noise.index_select(0, sel)
                eye = _hip.scale_batch(torch.eye(w, dtype=A.dtype, device=A.device).expand(Bt, w, w).contiguous(), scale=perturb)
                _hip.gemm_axpby(noise, eye, W, delta, 1.0)         # W += delta * noise (zero items: += 0; rank-keeping items: untouched)
            else:
                # column norms [Bt, w]: ttr_norm over the columns of the panel (a transposed layout copy of m x w elements; round 3
                # formed the whole w x w Gram matrix on the matrix cores for its diagonal)
                cn = _hip.norm(W0.transpose(1, 2).reshape(Bt * w, m)).reshape(Bt, w)
            if j0 == 0:
                Qj, Rjj = _hip.qr(W)
                d = torch.diagonal(Rjj, dim1=1, dim2=2)
            else:
                Qp = Q[:, :, :j0]
                C1 = _hip.gemm(Qp, W, transA=True)                 # j0 x w
                _hip.gemm_axpby(Qp, C1, W, -1.0, 1.0)              # W -= Qp C1
                Q1, R1 = _hip.qr(W)
                d = torch.diagonal(R1, dim1=1, dim2=2)
                C2 = _hip.gemm(Qp, Q1, transA=True)
                _hip.gemm_axpby(Qp, C2, Q1, -1.0, 1.0)             # Q1 -= Qp C2
                Qj, R2 = _hip.qr(Q1)
                Rjj = _hip.gemm(R2, R1)
                _hip.gemm_axpby(C2, R1, C1, 1.0, 1.0)              # C1 += C2 R1
                R[:, :j0, j0:j1] = C1
            if perturb is None:
                collapsed |= (d.abs() <= (32.0 * eps) * cn).any(dim=1)
            Q[:, :, j0:j1] = Qj
            R[:, j0:j1, j0:j1] = Rjj
        return Q, R, collapsed

    Q, R, collapsed = factor(None)
    bad = torch.nonzero(collapsed).flatten()           # (readback of the item list: control flow only)
    if bad.numel() == Bt:
        Q, R, _ = factor(live)
    elif bad.numel() > 0:
        # only the collapsing items are factored again (gather / scatter of those items: layout copies)
        A_all, live_all = A, live
        A, live, Bt, sel = A_all.index_select(0, bad).contiguous(), live_all.index_select(0, bad), int(bad.numel()), bad
        Qb, Rb, _ = factor(live)
        A, live, Bt, sel = A_all, live_all, A_all.shape[0], None
        Q.index_copy_(0, bad, Qb)
        R.index_copy_(0, bad, Rb)
    return Q, _hip.scale_batch(R, expo=a_exp, expo_sign=+1)


# ----------------------------------------------------------------------------------------------
# Symmetric eigenproblems too large for one workgroup: block Jacobi over the whole GPU.
#
# G (n x n) is cut into blocks of b <= 32 columns; a round pairs the blocks (round-robin), the 2b x 2b
# diagonal pair problems are diagonalised by the LDS-resident Jacobi kernel (batched over pairs and matrices, diagonal-matched
# column order) and the pair rotations are applied on the matrix cores:  G <- W^T G W,  V <- V W.  Device-resident driver
# (`_hip.bj_sweeps`): the blocks are never moved, the pair problems are gathered through a device pair table; the host-driven
# loop below (odd block counts, cross-check in tests) keeps the pairs physically adjacent by permuting the block order between
# rounds (gather copies -- layout only).
_BJ_MAX_SWEEPS = 12
_BJ_MAX_SWEEPS_DEVICE = 24   # (launches after convergence return at once: a generous bound is free)


def _bj_block(n: int) -> Optional[int]:
    for b in range(32, 7, -1):
        if n % b == 0 and n // b >= 2:
            return b
    return None


def _pair_cols(X: torch.Tensor, W: torch.Tensor, npairs: int, w: int) -> torch.Tensor:
    """X[:, :, p*w:(p+1)*w] <- X[:, :, p*w:(p+1)*w] @ W[b, p]  for the first npairs*w columns."""
    Bt, rows, n = X.shape
    used = npairs * w
    A = X[:, :, :used].reshape(Bt, rows, npairs, w).permute(0, 2, 1, 3).reshape(Bt * npairs, rows, w)
    out = _hip.gemm(A.contiguous(), W)  # [Bt*npairs, rows, w]
    out = out.reshape(Bt, npairs, rows, w).permute(0, 2, 1, 3).reshape(Bt, rows, used)
    if used == n:
        return out.contiguous()
    return torch.cat([out, X[:, :, used:]], dim=2)


def _pair_rows(X: torch.Tensor, W: torch.Tensor, npairs: int, w: int) -> torch.Tensor:
    """X[:, p*w:(p+1)*w, :] <- W[b, p]^T @ X[:, p*w:(p+1)*w, :]."""
    Bt, n, cols = X.shape
    used = npairs * w
    A = X[:, :used, :].reshape(Bt * npairs, w, cols)
    out = _hip.gemm(W, A, transA=True).reshape(Bt, used, cols)
    if used == n:
        return out
    return torch.cat([out, X[:, used:, :]], dim=1)


def eigh_block_jacobi(G: torch.Tensor, relative: bool = False, prerotation: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Eigen-decomposition of symmetric [B, n, n] (n a multiple of a block size in 8..32).

    Returns (V [B, n, n] orthogonal, d [B, n] eigenvalues, unsorted).
    Pair problems run on the Jacobi kernel (relative rotation test + absolute floor).  ``relative=False``: stop
    when ||offdiag|| <= sqrt(n)/2 eps ||G|| (absolute accuracy O(eps ||G||): pass 1 / 'eig').  ``relative=True``:
    stop when a whole sweep found nothing to rotate (pass 2 of 'svd': G is an accurately formed, nearly diagonal,
    graded Gram matrix and small eigenvalues keep relative accuracy).  ``prerotation`` (pass 1 of 'svd'): V only has to
    bring the matrix close to diagonal -- pass 2 re-forms the Gram matrix of the rotated data and continues from there --
    so the sweeps stop at an off-diagonal ratio of 1e-3 instead of the rounding floor (same total number of sweeps over
    the two passes, none wasted on digits pass 2 recomputes).
    """
    Bt, n, _ = G.shape
    b = _bj_block(n)
    assert b is not None
    nbk = n // b
    w = 2 * b
    dev, dt = G.device, G.dtype
    G0 = G
    G = G.clone()
    V = torch.eye(n, dtype=dt, device=dev).repeat(Bt, 1, 1)
    if nbk % 2 == 0 and BLOCK_JACOBI_ON_DEVICE:
        # the whole sweep loop on the device: two launches per round, one control launch per sweep, no readback -- the
        # maximum number of sweeps is enqueued and everything after convergence returns at its first instruction
        tol_dev = 1e-3 if prerotation else 0.5 * math.sqrt(n) * torch.finfo(dt).eps
        _hip.bj_sweeps(G, V, b, relative, tol_dev, _BJ_MAX_SWEEPS_DEVICE)
        return _bj_finish(G0, G, V, relative)
    circle = list(range(nbk)) + ([-1] if nbk % 2 else [])  # -1: bye
    phys = list(range(nbk))
    ar = torch.arange(b, device=dev)
    eps = torch.finfo(dt).eps
    gnorm = _hip.norm(G.reshape(Bt, -1))
    tol = 0.5 * math.sqrt(n) * eps
    prev = None
    for sweep in range(_BJ_MAX_SWEEPS):
        worked = None
        for _ in range(len(circle) - 1):
            half = len(circle) // 2
            pairs = [(circle[i], circle[-1 - i]) for i in range(half)]
            bye = [x for pr in pairs if -1 in pr for x in pr if x != -1]
            pairs = [pr for pr in pairs if -1 not in pr]
            target = [x for pr in pairs for x in pr] + bye
            if target != phys:  # physical re-ordering of block rows / columns (gather: layout only)
                slot = {blk: i for i, blk in enumerate(phys)}
                idx = torch.cat([slot[blk] * b + ar for blk in target])
                G = G.index_select(1, idx).index_select(2, idx)
                V = V.index_select(2, idx)
                phys = target
            npairs = len(pairs)
            Gv = G[:, : npairs * w, : npairs * w].reshape(Bt, npairs, w, npairs, w)
            S = torch.stack([Gv[:, p, :, p, :] for p in range(npairs)], dim=1).reshape(Bt * npairs, w, w)
            # pair problems: the 4-wave Jacobi kernel with the diagonal-matched column order.  Measured against the
            # tridiagonal kernel (rounds 1-2: one wave per matrix) it is 1.2x (8192 pair problems per round) to 2.7x (16) faster here: its
            # pre-check makes converged pairs free, so every outer sweep is cheaper than the one before.
            nsw = torch.zeros(Bt * npairs, dtype=torch.int32, device=dev) if relative else None
            W, _, _ = _hip.eigh_trunc(S.contiguous(), _hip.EIG_MATCH_DIAG, False, 0.0, w,
                                      abs_floor=_hip.SOLVER_JACOBI_ABS, sweeps=nsw)
            if relative:
                worked = nsw.max() if worked is None else torch.maximum(worked, nsw.max())
            G = _pair_cols(G, W, npairs, w)
            G = _pair_rows(G, W, npairs, w)
            V = _pair_cols(V, W, npairs, w)
            circle = [circle[0], circle[-1]] + circle[1:-1]
        if relative:  # one int32 readback per sweep (control flow only)
            if int(worked.item()) == 0:
                break
            continue
        # off-diagonal mass per matrix (one small readback per sweep; control flow only)
        Goff = G.clone()
        torch.diagonal(Goff, dim1=1, dim2=2).zero_()
        ratio = float((_hip.norm(Goff.reshape(Bt, -1)) / gnorm.clamp_min(torch.finfo(dt).tiny)).max().item())
        if ratio <= tol or (prev is not None and sweep >= 3 and ratio > 0.7 * prev and ratio <= 64 * tol):  # (as bj_control_kernel)
            break
        prev = ratio
    return _bj_finish(G0, G, V, relative)


BLOCK_JACOBI_ON_DEVICE = True   # False: the round-2 host-driven loop (kept for odd block counts and as a cross-check in tests)


def _bj_finish(G0, G, V, relative):
    d = torch.diagonal(G, dim1=1, dim2=2).contiguous()
    # one Newton-Schulz step removes the orthogonality drift of the ~100 accumulated block rotations
    # (3e-6 in fp32, 1e-13 in fp64 -- the Rayleigh quotients below assume unit columns)
    Vn = V.clone()
    V = _hip.gemm_axpby(V, _hip.gemm(V, V, transA=True), Vn, -0.5, 1.5)   # 1.5 V - 0.5 V (V^T V)
    if not relative:
        # Rayleigh quotients against the ORIGINAL matrix: the ~100 two-sided fp updates of G accumulate
        # O(100 eps ||G||) in its diagonal, v_i^T G0 v_i is second order in the eigenvector error
        d = torch.diagonal(_hip.gemm(V, _hip.gemm(G0, V), transA=True), dim1=1, dim2=2).contiguous()
    return V, d


def _eigh_any(G: torch.Tensor, eig_mode: int, use_delta: bool, delta2: float, cap: int, solver: int, prerotation: bool = False):
    """Dispatch on the problem size: n <= LDS limit -> one workgroup per matrix (tridiagonal QL for n <= 64 when
    requested, Jacobi otherwise); larger -> block Jacobi over the GPU, then one (rotation-free) pass of the
    Jacobi kernel for its epilogue (clamp, sqrt, sort, rank rule).  Sizes no block width in 8..32 divides are
    padded with decoupled rows/columns whose eigenvalue -||G||_F lies below every real one."""
    Bt, n, _ = G.shape
    lds_limit = _hip.lib().ttr_eigh_max_n_lds(_hip.dtype_code(G.dtype))
    if n <= lds_limit:
        return _hip.eigh_trunc(G, eig_mode, use_delta, delta2, cap, abs_floor=solver)
    relative = solver != _hip.SOLVER_TRIDIAG
    if _bj_block(n) is not None:
        Vb, d = eigh_block_jacobi(G, relative=relative, prerotation=prerotation)
        P, sig, info = _hip.eigh_trunc(torch.diag_embed(d), eig_mode, use_delta, delta2, cap, abs_floor=_hip.SOLVER_JACOBI_ABS)
        return _hip.gemm(Vb, P), sig, info
    nblk = -(-n // 32)
    b = -(-n // nblk)
    npad = nblk * b
    if npad > _hip.max_eigh_n(G.dtype):
        return _hip.eigh_trunc(G, eig_mode, use_delta, delta2, cap, abs_floor=solver)
    gn = _hip.norm(G.reshape(Bt, -1))                                     # >= |lambda|_max
    Gp = G.new_zeros((Bt, npad, npad))
    Gp[:, :n, :n] = G
    pad_idx = torch.arange(n, npad, device=G.device)
    Gp[:, pad_idx, pad_idx] = gn[:, None].neg().expand(Bt, npad - n)      # sentinel eigenvalues (fill: layout)
    Vb, d = eigh_block_jacobi(Gp, relative=relative)
    # order by d + ||G|| (pads become exactly the smallest): ttr_gemm_axpby adds the shift, the epilogue sorts
    shifted = d.reshape(Bt, npad, 1).clone()
    _hip.gemm_axpby(torch.ones((Bt, npad, 1), dtype=G.dtype, device=G.device), gn.reshape(Bt, 1, 1), shifted, 1.0, 1.0)
    P1, _, _ = _hip.eigh_trunc(torch.diag_embed(shifted[:, :, 0]), _hip.EIG_RAW, False, 0.0, npad,
                               abs_floor=_hip.SOLVER_JACOBI_ABS)
    Vs = _hip.gemm(Vb, P1)[:, :n, :n].contiguous()                        # real eigenvectors, decreasing eigenvalue
    ds = torch.diagonal(_hip.gemm(P1, _hip.gemm(torch.diag_embed(d), P1), transA=True), dim1=1, dim2=2)[:, :n]
    P2, sig, info = _hip.eigh_trunc(torch.diag_embed(ds.contiguous()), eig_mode, use_delta, delta2, cap,
                                    abs_floor=_hip.SOLVER_JACOBI_ABS)
    return _hip.gemm(Vs, P2), sig, info


# Selected eigenpairs instead of a full decomposition (csrc/ttr_eigsel.hip): batch mode with a rank cap far below the size of the
# bond's Gram matrix -- only the top of the spectrum is ever looked at (BASELINE config C3: n = 256, rmax = 8; the block-Jacobi
# driver needs ~170 launches per such bond).  TTR_EIGH_TOPK=0 switches it off.
EIGH_TOPK_ENABLED = os.environ.get("TTR_EIGH_TOPK", "1") != "0"
# batch-mode bonds with <= 64 rows and a rank cap <= 32: pass 1 by the top-r solver (ttr_eigh_top), the QL solver only for the
# items that one declines
EIGH_TOP_ENABLED = os.environ.get("TTR_EIGH_TOP", "1") != "0"
_TOPK_MAX_RANK = 32


def _topk_one_pass(G: torch.Tensor, r: int, use_delta: bool = False, delta2: float = 0.0):
    """Pass 1 of an 'svd' truncation as the answer, from the r largest eigenpairs of the Gram matrix alone: returns
    (V [B, n, r], sigma [B, r], info [B]) when EVERY item's kept singular values are flat (sigma_r >= FLAT_SPECTRUM_THR sigma_1:
    the first Gram pass carries them to a few eps, see ``truncate``), the solver resolved them (no collapsed vector: clustered
    or multiple eigenvalues are the block-Jacobi driver's job) and -- in eps mode -- the rank cap provably binds: the tail
    energy beyond the cap, trace(G) - sum of the r largest eigenvalues, exceeds delta^2 by more than the error margin
    E = 64 n eps sigma_1^2 of a pass-1 tail energy (the criterion of ttr_spectrum_flat with use_delta).  Else None.  One flag
    readback, like the full-decomposition variant of the same decision."""
    Bt, n, _ = G.shape
    if not EIGH_TOPK_ENABLED or n <= 64 or n > _hip.lib().ttr_eigsel_max_n() or r > _TOPK_MAX_RANK or 4 * r > n:
        return None
    Gn, ex = _hip.pow2_normalize(G)                    # ||G[b]|| in [0.5, 1): the reduction squares the entries
    X, lam_n, rmin = _hip.eigh_topk(Gn, r)
    lam = _hip.scale_batch(lam_n, expo=ex, expo_sign=+1)  # exact
    sig = lam.clamp_min(0).sqrt()                      # ([B, r]: epilogue of the solver, as the eigensolver kernels' own sqrt)
    ok = _hip.spectrum_flat(sig, r, FLAT_SPECTRUM_THR) * (rmin > 0.5).to(torch.int32)
    zero = sig[:, 0] < 1e-13                           # zero guard, round.py:137-145 (an all-zero item is "flat" as well)
    if use_delta and delta2 > 0.0:
        # (a few [B]-sized device scalars, in double: tail beyond the cap against delta^2 + E, everything in G's own units)
        tail = _gram_trace(G).double() - lam.double().sum(dim=1)
        margin = 64.0 * n * torch.finfo(G.dtype).eps * lam[:, 0].double().clamp_min(0)
        ok = ok * (tail > delta2 + margin).to(torch.int32)
    if int((ok + zero.to(torch.int32)).amin().item()) < 1:  # (readback: control flow only)
        return None
    info = torch.where(zero, 0, r).to(torch.int32)
    return X, sig, info


# Rank-capped truncation of a big bond (both dimensions above 64) whose energy is CONCENTRATED -- real data, low rank + noise,
# decaying spectra: SURVEY 8d's primary inputs for C1 / C3 -- through a randomised range finder on the bond's n x n Gram matrix
# (north_star: "randomised range-finder SVD"): subspace iteration Q <- orth(G Q) with l = max(32, 2 r) <= 64 columns (one small
# MFMA GEMM + one TSQR per step; G is already there, the iteration never touches M), then ONE more pass over M for the
# projection B = M Q and the fused <= 64-column / -row two-pass truncation of B (singular values of B to O(eps sigma_1), the
# class of gesdd) -- instead of an n x n eigen-decomposition, a rotation M V1 (as many flops as the Gram matrix), a second Gram
# matrix and a second n x n eigenproblem.  The basis is accepted on a CERTIFICATE, per item, decided from a handful of device
# scalars and one flag readback per round (control flow, like the flat-spectrum decisions next to it):
#   tau = trace(G) - trace(Q^T G Q) >= lambda_max of G on the complement of span(Q)  (G is PSD), so with theta_r > 2 tau the r kept
#   Ritz values are separated from everything outside the basis by gap >= theta_r - tau, and lambda_j - theta_j <=
#   ||G Q - Q (Q^T G Q)||_F^2 / gap  (the quadratic residual bound), which has to lie below 1e-5 sqrt(theta_r theta_1), i.e. the
#   kept singular values of B = M Q equal those of M to ~5e-6 sigma_1 (they can only be smaller: interlacing).
# A flat spectrum (randn: the participation ratio trace(G)^2 / ||G||_F^2 exceeds the basis size -- decided before any
# iteration, from two reductions) or a heavy tail fails the certificate and takes the paths below unchanged (selected
# eigenpairs when the kept spectrum is flat, the full decomposition otherwise).  eps mode: only when the rank rule, evaluated
# on B's singular values alone (the invisible tail can only ADD energy), already returns the cap.  TTR_SUBSPACE=0 switches it off.
SUBSPACE_ENABLED = os.environ.get("TTR_SUBSPACE", "1") != "0"
_SUBSPACE_ROUNDS = (2, 4, 8)      # subspace-iteration steps before the 1st / 2nd / 3rd look at the certificate
PATH_TRACE: Optional[list] = None  # bench / tests: set to a list and ``truncate`` appends (path, m, n, rank cap) for every big bond


def _trace(path: str, m: int, n: int, r: int) -> None:
    if PATH_TRACE is not None:
        PATH_TRACE.append((path, int(m), int(n), int(r)))


def _subspace_basis(G: torch.Tensor, r: int) -> Optional[torch.Tensor]:
    """Orthonormal Q [B, n, l] whose span carries the r leading eigenpairs of the PSD matrices G [B, n, n] to the certificate
    above for EVERY item, or None."""
    Bt, n, _ = G.shape
    l = min(64, max(32, 2 * r))
    if not SUBSPACE_ENABLED or r > _TOPK_MAX_RANK or n < 2 * l or l > _hip.max_qr_cols(G.dtype):
        return None
    Gn, _ = _hip.pow2_normalize(G)                       # ||G[b]||_F in [0.5, 1)
    tr = _gram_trace(Gn).double()                         # ([B]-sized device scalars from here on, in double)
    fro = _hip.norm(Gn.reshape(Bt, -1)).double()
    live = fro > 0
    eff = torch.where(live, tr * tr / (fro * fro).clamp_min(1e-300), torch.zeros_like(tr))
    if float(eff.amax().item()) > l + 1:                  # (readback: control flow only) energy spread over more directions than the basis holds
        return None
    gen = torch.Generator(device=G.device)
    gen.manual_seed(0x5AB5 + n)
    Q, _ = _hip.qr(_hip.gemm(Gn, torch.randn((Bt, n, l), dtype=G.dtype, device=G.device, generator=gen)))
    eps = torch.finfo(G.dtype).eps
    for steps in _SUBSPACE_ROUNDS:
        for _ in range(steps):
            Q, _ = _hip.qr(_hip.gemm(Gn, Q))
        Y = _hip.gemm(Gn, Q)
        H = _hip.gemm(Q, Y, transA=True)                  # l x l
        _hip.gemm_axpby(Q, H, Y, -1.0, 1.0)               # Y <- G Q - Q H
        res2 = _hip.norm(Y.reshape(Bt, -1)).double().square()
        _, sg, _ = _hip.eigh_trunc(H, _hip.EIG_RAW, False, 0.0, l, abs_floor=_hip.SOLVER_TRIDIAG)
        th = sg.double().square()                         # Ritz values, decreasing
        tau = (tr - th.sum(dim=1)).clamp_min(0) + 4.0 * (l + math.sqrt(n)) * eps * tr   # (+ the rounding level of the subtraction)
        th_r, th_1 = th[:, r - 1], th[:, 0]
        ok = (th_r > 2.0 * tau) & (res2 <= (th_r - tau) * (th_r * th_1).sqrt() * 1e-5)
        if bool((ok | ~live).all().item()):               # (readback: control flow only)
            return Q
        if bool(((th_r <= 2.0 * tau) & live).any().item()) and steps >= 4:
            break                                          # the tail outside the basis is not small: more steps will not change that
    return None


def _subspace_truncate(M, G, r, delta, rmax, left_ortho, algorithm, batch, right_alloc, gtr):
    """``truncate`` of a big bond through the range finder above; None when the basis is not certified (or, in eps mode, the
    rank rule does not return the cap on the visible spectrum)."""
    Bt, m, n = M.shape
    left_side = m <= n
    Q = _subspace_basis(G, r)
    if Q is None:
        return None
    if left_side:
        Bm = _hip.gemm(Q, M, transA=True)                 # l x n: the fused row kernels take it from here
        t = truncate(Bm, delta, rmax, left_ortho, algorithm, batch, right_alloc)
    else:
        Bm = _hip.gemm(M, Q)                              # m x l: ... the fused column kernels
        t = truncate(Bm, delta, rmax, left_ortho, algorithm, batch)
    if t.zero:
        zl, zr = _zero_factors(M)
        return Truncation(zl, None, zr, 1, zero=True, gtrace=gtr)
    if not batch and t.rank < r:
        return None                                        # eps mode: the cap does not bind on the visible spectrum -- the full path decides
    if left_side:
        left = _hip.gemm(Q, t.left_scaled())
        return Truncation(left, None, t.right, t.rank, info=t.info, gtrace=gtr)
    right = _hip.gemm(t.right, Q, transB=True)
    return Truncation(t.left, t.colscale, right, t.rank, info=t.info, gtrace=gtr)


class Truncation:
    """Result of ``truncate``: ``left_core`` (m x r), optional column scale, ``right`` (r x n)."""

    __slots__ = ("left", "colscale", "right", "rank", "zero", "info", "gtrace")

    def __init__(self, left, colscale, right, rank, zero=False, info=None, gtrace=None):
        self.left, self.colscale, self.right, self.rank, self.zero = left, colscale, right, rank, zero
        self.info = info  # [B] int32 from the eigensolver epilogue: the item's rank by the rank rule, 0 = zero guard (round.py:137-145)
        self.gtrace = gtrace  # [B] trace of the first Gram matrix = ||M[b]||_F^2 (device; ``want_trace``): the fp32 range guard without a pass over M

    def left_scaled(self) -> torch.Tensor:
        if self.colscale is None:
            return self.left
        return _hip.scale_cols(self.left, self.colscale, _hip.SCALE_MUL)


_ZF_PENDING = -(2 ** 31)   # (never a rank-rule result: those are >= 0)


def _pinned_flag_wait(host: torch.Tensor):
    """Callable that waits until the device has written the pinned host word ``host`` (initialised to ``_ZF_PENDING``; written by a
    kernel of the current stream) and hands the tensor back.  Polls the word; after 2 s without a write it synchronises the
    stream instead (the kernel's write is visible at the latest when the kernel has completed)."""
    stream = torch.cuda.current_stream()

    def wait() -> torch.Tensor:
        import time

        t0 = time.perf_counter()
        while int(host.min()) == _ZF_PENDING:      # (every word: the device's stores become visible in no particular order)
            if time.perf_counter() - t0 > 2.0:
                stream.synchronize()
                break
        return host

    return wait


def _deferred_readback(x: torch.Tensor):
    """Start an asynchronous copy of a small device tensor to pinned host memory on the current stream and return a
    callable that waits for THAT copy (an event -- not a stream or device synchronisation) and hands back the host
    tensor.  For control-flow scalars that are only needed once everything else has been enqueued: the host blocks,
    the device never idles."""
    host = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
    host.copy_(x, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()

    def wait() -> torch.Tensor:
        ev.synchronize()
        return host

    return wait


# sigma_keep >= FLAT_SPECTRUM_THR * sigma_1: the second Gram pass of the 'svd' truncation is skipped for that item (batch
# mode); 0 disables the shortcut (TTR_FLAT_SPECTRUM_THR).  1/8: one pass leaves right-orthonormality 8e-7 and relative sigma
# errors 4e-7 at sigma_keep = sigma_1 / 8 against 4e-7 / 2e-7 after two passes (fp32, emulation with the kernels' split
# accumulation), and the metric's bonds (sigma_32 / sigma_1 = 0.19 .. 0.73) all qualify: 626 k cores/s
# against 610 k at 1/4 (five of the six large bonds) and 549 k without the shortcut, parity vs the oracle 7.4e-6 / 8.4e-6 / 8.7e-6.
FLAT_SPECTRUM_THR = float(os.environ.get("TTR_FLAT_SPECTRUM_THR", "0.125"))


def _gram_trace(G: torch.Tensor) -> torch.Tensor:
    """[B] trace of a Gram matrix [B, n, n] or of its split partials [B, parts, n, n] (a few hundred numbers: the only arithmetic)."""
    return torch.diagonal(G, dim1=-2, dim2=-1).reshape(G.shape[0], -1).sum(dim=1)


def _select_rank(info: torch.Tensor, batch: bool, rmax: Optional[int], k: int) -> int:
    if batch:  # round.py:149-150: no eps truncation, no device->host sync
        return _rank_cap(rmax, k)
    return int(info[0].item())  # one readback per truncation, as round.py:150-158


_INPLACE_ROTATE_BYTES = 1 << 32   # carries above 4 GiB are rotated in place ...
_INPLACE_CHUNK_BYTES = 1 << 30    # ... through a 1 GiB buffer


def truncate(
    M: torch.Tensor,
    delta: Optional[float],
    rmax: Optional[int],
    left_ortho: bool,
    algorithm: str,
    batch: bool,
    right_alloc=None,
    scratch_ok: bool = False,
    gram: Optional[torch.Tensor] = None,
    delta2_dev: Optional[torch.Tensor] = None,
    want_trace: bool = False,
    consume: bool = False,
    rows32: Optional[torch.Tensor] = None,
) -> Truncation:
    """Truncated SVD of ``M`` [B, m, n]; semantics of round.py:52-187.
    ``right_alloc(r)``: optional callable returning the contiguous [B, r, n] tensor ``right`` is written into.
    ``scratch_ok``: M is a temporary of the caller (the carry of a dense TT-SVD) and may be overwritten: a tall M with
    more than 64 columns is then rotated IN PLACE (row chunks) instead of into a second tensor of its size.
    ``gram``: the first Gram matrix, already formed -- split partials of M M^T accumulated by the kernel that produced M
    (``qr_apply(want_gram=True)``), or what ``first_gram`` returns (a dense TT-SVD takes ||X|| from its trace).
    ``delta2_dev`` (device double [1], fused <= 64-row path only): eps mode WITHOUT a readback -- the rank rule takes its bound
    from device memory, the factors are computed at the rank cap ``min(rmax, k)``, the selected rank stays on the device
    (``Truncation.info``) and the columns of ``left`` beyond it are zeroed there (``ttr_mask_cols``): the caller slices the
    cores once, at the end of its sweep, after ONE readback of all ranks.

    ``rows32`` (fused <= 64-row path; int32 [B] on the device): items flagged != 0 have exactly zero rows 32.. in M (the carry of
    a bond whose QR packed its rows, ``_hip.QrFactors.rows32``): the Gram and projection kernels do not load them.

    ``consume`` (tall fused path, one contiguous matrix): M's storage may be overwritten -- ``left`` is produced in place over
    its front (``_colproject_inplace``).

    ``want_trace``: also return ``Truncation.gtrace`` = trace of the first Gram matrix (the tall column sweep and the GEMM
    path: the shapes of a dense TT-SVD's first steps).

    ``left_ortho=False`` (the branch round_tt uses): ``right`` has orthonormal rows,
    ``left * colscale`` carries the singular values.  ``left_ortho=True``: ``left`` is
    orthonormal, ``right`` carries them.
    """
    Bt, m, n = M.shape
    k = min(m, n)
    use_delta = not batch
    delta2 = float(delta) ** 2 if (delta is not None and use_delta) else 0.0
    cap = INT32_MAX if rmax is None else int(rmax)
    left_side = m <= n  # round.py:104-109; the 'svd' path is orientation-free
    ref_clamp = algorithm == "eig"

    if left_side and _hip.sweep_fused_ok(M):
        # Up to 64 rows (every bond of a train with TT ranks <= 64): the fused sweep kernels.  M is streamed three
        # times ('eig': twice) and nothing of its size is written except the result: row Gram -> pass-1 eigenvectors V1
        # -> Gram of the ROTATED rows (ttr_rotgram: the rotated matrix only exists 16 columns at a time in registers)
        # -> Jacobi -> projection with U = V1 V2 formed in the kernel's prologue, which also emits left = U sigma.
        V1 = None
        st = _Stage()
        G = gram if gram is not None else _hip.rowgram(M, rows32=rows32)
        gtr = _gram_trace(G) if want_trace else None
        if algorithm == "eig":
            st.lap("Time (gram):")
        if algorithm == "svd":
            # r of the top-r launch: the rank cap in batch mode; in eps mode (any non-batch call) min(cap, 32) -- with `need_all`
            # the top-r path then only serves the 32 x 32 zero-tail problems of packed bonds, whose whole spectrum that is
            r_top = _rank_cap(rmax, k) if batch else min(_rank_cap(rmax, k), 32)
            if ((rmax is not None or not batch) and EIGH_TOP_ENABLED and FLAT_SPECTRUM_THR > 0 and _hip.eigh_top_ok(k, r_top)):
                # only the rmax largest eigenpairs matter when the kept spectrum is flat: multisection + twisted factorisations
                # instead of the QL iteration over the whole spectrum, decided per item inside the launch (flat kept spectrum
                # without close pairs; the others fall through to the QL phase).  Such items carry zeros beyond column / entry
                # rmax, which nothing below looks at in batch mode.
                # (the deferred eps-mode sweep, round 5: its rank rule needs EVERY sigma, so the top-r path only takes the items of
                # which it computes every eigenpair -- the 32 x 32 zero-tail problems of a packed bond under a cap >= 32: 47 instead
                # of ~100 us per bond of one 64^8 train; its flags are not the pass-through flags there)
                V1, sig1, _, top_flat = _hip.eigh_top(G, r_top, FLAT_SPECTRUM_THR, need_all=not batch)
                if not batch:
                    top_flat = None
            else:
                V1, sig1, _ = _hip.eigh_trunc(G, _hip.EIG_RAW, False, 0.0, k, abs_floor=_hip.SOLVER_TRIDIAG)
                top_flat = None
            flat = None
            if top_flat is not None and not use_delta:
                # batch mode: the launch's flags are the pass-through flags already (1: top-r path; 2: declined there, but the
                # full decomposition's kept sigma pass the same test) -- include/ttround_hip.h: ttr_eigh_top
                flat = top_flat
            elif FLAT_SPECTRUM_THR > 0:
                # items whose KEPT singular values lie within a factor 8 of each other do not need the second pass (the first
                # Gram matrix already carries them to a few eps; include/ttround_hip.h: ttr_spectrum_flat): their rotated
                # Gram matrix is not formed and the pass-2 solver hands pass 1's result through.  Decided per item, on the
                # device.  Batch mode: the rank does not depend on the small singular values.  eps mode: the rank rule is
                # evaluated on pass 1's sigma and the item only qualifies when pass 2 provably selects the same rank (fp64
                # trains: config C2; in fp32 the error margin of pass 1's tail energies exceeds a delta of 1e-4 ||T||).
                flat = _hip.spectrum_flat(sig1, _rank_cap(rmax, k), FLAT_SPECTRUM_THR, use_delta, delta2, delta2_dev, rows32=rows32)
                if top_flat is not None:
                    # an item the top-r kernel decided on eigenvalues carries ONLY its r leading eigenpairs: it must pass through,
                    # also when the same test on sigma = sqrt(lambda) rounds to the other side of the threshold ([B] int32 flags)
                    flat = torch.maximum(flat, top_flat)
            V, sig, info = _hip.eigh_trunc(_hip.rowgram(M, V1, skip=flat, rows32=rows32), _hip.EIG_RAW, use_delta, delta2, cap,
                                           abs_floor=_hip.SOLVER_JACOBI_LIVE, delta2_dev=delta2_dev,
                                           skip_items=flat, sigma_in=sig1 if flat is not None else None)
        else:
            V, sig, info = _hip.eigh_trunc(G, _hip.EIG_REF, use_delta, delta2, cap,
                                           abs_floor=_hip.SOLVER_TRIDIAG, delta2_dev=delta2_dev)
        st.lap("Time (SVD):" if algorithm == "svd" else "Time (symmetric EIG):")   # ('svd' here: both Gram passes + both solvers)
        if delta2_dev is not None:  # rank on the device: factors at the cap, the columns beyond info[b] zeroed in place
            r = _rank_cap(rmax, k)
            right, left = _hip.project(M, V1, V, sig, r, scale_right=not left_ortho, rows32=rows32)
            if algorithm == "svd" and not left_ortho:
                _hip.orth_fixup(right, sig, r, k * torch.finfo(M.dtype).eps, rank_dev=info)  # (not the rows that are cut away)
            _hip.mask_cols(left, info)
            st.lap("Time (product):")
            return Truncation(left, None, right, r, info=info, gtrace=gtr)
        r = _select_rank(info, batch, rmax, k)
        if r == 0:  # zero guard, round.py:137-145 (kept on M's device/dtype)
            return Truncation(torch.zeros((Bt, m, 1), dtype=M.dtype, device=M.device), None,
                              torch.zeros((Bt, 1, n), dtype=M.dtype, device=M.device), 1, zero=True, gtrace=gtr)
        dst = right_alloc(r) if right_alloc is not None else None
        right, left = _hip.project(M, V1, V, sig, r, scale_right=not left_ortho, out=dst, rows32=rows32)
        if algorithm == "svd" and not left_ortho:
            _hip.orth_fixup(right, sig, r, k * torch.finfo(M.dtype).eps)  # see below
        st.lap("Time (product):")
        return Truncation(left, None, right, r, info=info, gtrace=gtr)

    if not left_side and _hip.colsweep_fused_ok(M):
        # Tall matrix with up to 64 columns (the first, largest steps of a dense right-to-left TT-SVD): the same fused
        # kernels with the contraction over the rows.  The unfolding is read three times ('eig': twice) and only the
        # carry (r / n of its size) is written -- no rotated copy of the input.
        V1 = None
        st = _Stage()
        G0 = gram if gram is not None else _hip.colgram(M)
        gtr = _gram_trace(G0) if want_trace else None
        if algorithm == "eig":
            st.lap("Time (gram):")
        if algorithm == "svd":
            V1, sig1, _ = _hip.eigh_trunc(G0, _hip.EIG_RAW, False, 0.0, k, abs_floor=_hip.SOLVER_TRIDIAG)
            # (batch mode, or eps mode with the rank decision certified on pass 1's sigma: see the row sweep above)
            flat = _hip.spectrum_flat(sig1, _rank_cap(rmax, k), FLAT_SPECTRUM_THR, use_delta, delta2) if FLAT_SPECTRUM_THR > 0 else None
            V, sig, info = _hip.eigh_trunc(_hip.colgram(M, V1, skip=flat), _hip.EIG_RAW, use_delta, delta2, cap,
                                           abs_floor=_hip.SOLVER_JACOBI_LIVE,
                                           skip_items=flat, sigma_in=sig1 if flat is not None else None)  # (as the row sweep above)
        else:
            V, sig, info = _hip.eigh_trunc(G0, _hip.EIG_REF, use_delta, delta2, cap,
                                           abs_floor=_hip.SOLVER_TRIDIAG)
        st.lap("Time (SVD):" if algorithm == "svd" else "Time (symmetric EIG):")
        r = _select_rank(info, batch, rmax, k)
        if r == 0:
            return Truncation(torch.zeros((Bt, m, 1), dtype=M.dtype, device=M.device), None,
                              torch.zeros((Bt, 1, n), dtype=M.dtype, device=M.device), 1, zero=True, gtrace=gtr)
        if consume and Bt == 1 and M.is_contiguous() and 2 * r <= n:
            left, right = _colproject_inplace(M, V1, V, sig, r, left_ortho)
        else:
            left, right = _hip.colproject(M, V1, V, sig, r, left_ortho)
        if algorithm == "svd" and left_ortho:
            _hip.orth_fixup(left, sig, r, k * torch.finfo(M.dtype).eps, columns=True)
        st.lap("Time (product):")
        return Truncation(left, None, right, r, info=info, gtrace=gtr)

    one_pass = None
    st = _Stage()
    if algorithm == "svd":
        # ---- pass 1: rotate into (nearly) orthogonal rows / columns
        G = gram if gram is not None else (_hip.gemm(M, M, transB=True) if left_side else _hip.gemm(M, M, transA=True))
        gtr = _gram_trace(G) if want_trace else None
        # Batch mode: pass 1 is run to full accuracy and, when EVERY item's kept singular values lie within 1 / FLAT_SPECTRUM_THR
        # of each other, it is the answer (see the fused path above) -- the rotation GEMM, the second Gram matrix and the
        # second eigenproblem, i.e. two of the three passes over M, are not enqueued at all.  The decision concerns launches
        # on the host, hence one flag readback per such bond (dense batches: BASELINE config C3; the bonds of a TT-to-TT
        # rounding have <= 64 rows and decide per item on the device).  Otherwise pass 1 is a pre-rotation.
        try_flat = batch and FLAT_SPECTRUM_THR > 0
        if (batch or rmax is not None) and min(m, n) > 64:
            # concentrated spectra (low rank + noise, decaying): certified range finder + the fused small truncation
            sub = _subspace_truncate(M, G, _rank_cap(rmax, k), delta, rmax, left_ortho, algorithm, batch, right_alloc, gtr)
            if sub is not None:
                _trace("subspace", m, n, _rank_cap(rmax, k))
                return sub
        if FLAT_SPECTRUM_THR > 0 and (batch or rmax is not None):
            # (eps mode with a rank cap -- a single dense tensor to given ranks, BASELINE config C1: only when the cap provably binds)
            one_pass = _topk_one_pass(G, _rank_cap(rmax, k), use_delta, delta2)
        if one_pass is not None:
            V1, Mw = None, M
            _trace("topk_one_pass", m, n, _rank_cap(rmax, k))
        else:
            V1, sig1, info1 = _eigh_any(G, _hip.EIG_RAW, False, 0.0, cap if try_flat else k, _hip.SOLVER_TRIDIAG,
                                        prerotation=not try_flat)
        if one_pass is not None:
            pass
        elif try_flat and int(_hip.spectrum_flat(sig1, _rank_cap(rmax, k), FLAT_SPECTRUM_THR).amin().item()) == 1:
            one_pass = (V1, sig1, info1)
            V1, Mw = None, M
            _trace("full_one_pass", m, n, _rank_cap(rmax, k))
        elif left_side:
            _trace("full_two_pass", m, n, _rank_cap(rmax, k))
            Mw = _hip.gemm(V1, M, transA=True)           # V1^T M
            G = _hip.gemm(Mw, Mw, transB=True)
        else:
            _trace("full_two_pass", m, n, _rank_cap(rmax, k))
            if scratch_ok and M.is_contiguous() and M.numel() * M.element_size() > _INPLACE_ROTATE_BYTES:
                # config-scale carries (C1 class: tens of GiB): every row of M V1 depends on the same row of M only, so
                # the rotation runs chunk by chunk into a bounded buffer that is copied back over its source rows
                rows_per = max(1, _INPLACE_CHUNK_BYTES // (n * M.element_size() * Bt))
                for r0 in range(0, m, rows_per):
                    blk = M[:, r0:r0 + rows_per]
                    blk.copy_(_hip.gemm(blk, V1))
                Mw = M
            else:
                Mw = _hip.gemm(M, V1)                    # M V1
            G = _hip.gemm(Mw, Mw, transA=True)
    else:
        V1 = None
        Mw = M
        G = gram if gram is not None else (_hip.gemm(M, M, transB=True) if left_side else _hip.gemm(M, M, transA=True))
        gtr = _gram_trace(G) if want_trace else None

    # 'eig' (and pass 1 above): absolute accuracy is all a plain Gram matrix carries -> tridiagonal QL solver;
    # pass 2 of 'svd': graded, accurately formed Gram matrix -> Jacobi (relative accuracy of the small sigmas)
    # (problems above the LDS limit go through the block-Jacobi driver in either pass: absolute accuracy)
    if one_pass is not None:
        V, sig, info = one_pass
    else:
        V, sig, info = _eigh_any(G, _hip.EIG_REF if ref_clamp else _hip.EIG_RAW, use_delta, delta2, cap,
                                 _hip.SOLVER_TRIDIAG if algorithm == "eig" else _hip.SOLVER_JACOBI_LIVE)
    # 'svd': kept directions whose sigma lies below the resolution of the input (k eps sigma_max) carry rounding
    # noise only; LAPACK's V is orthonormal there too (round.py:96), so they get an orthonormal completion
    # (ttr_orth_fixup; a per-item early exit when there are none -- the normal case)
    dead_rel = k * torch.finfo(M.dtype).eps if algorithm == "svd" else None
    st.lap("Time (SVD):" if algorithm == "svd" else "Time (gram + symmetric EIG):")
    r = _select_rank(info, batch, rmax, k)
    if r == 0:  # zero guard, round.py:137-145 (kept on M's device/dtype)
        z_l = torch.zeros((Bt, m, 1), dtype=M.dtype, device=M.device)
        z_r = torch.zeros((Bt, 1, n), dtype=M.dtype, device=M.device)
        return Truncation(z_l, None, z_r, 1, zero=True, gtrace=gtr)
    Vr = V[:, :, :r]

    if left_side:
        # right = diag(1/sigma) Vr^T Mw   (or Vr^T Mw when left_ortho)
        dst = right_alloc(r) if right_alloc is not None else None
        if left_ortho:
            right = _hip.gemm(Vr, Mw, transA=True, out=dst)
        else:
            right = _hip.gemm(Vr, Mw, transA=True, rowscale=sig, rowscale_mode=_hip.SCALE_DIV, out=dst)
            if dead_rel is not None:
                _hip.orth_fixup(right, sig, r, dead_rel)
        U = _hip.gemm(V1, Vr) if V1 is not None else Vr
        st.lap("Time (product):")
        return Truncation(U, None if left_ortho else sig, right, r, info=info, gtrace=gtr)
    # right side: left = Mw Vr (= U sigma); right = (V1 Vr)^T
    if left_ortho:
        left = _hip.gemm(Mw, Vr, colscale=sig, colscale_mode=_hip.SCALE_DIV)
        if dead_rel is not None:
            _hip.orth_fixup(left, sig, r, dead_rel, columns=True)
        if V1 is not None:
            right = _hip.gemm(Vr, V1, transA=True, transB=True, rowscale=sig, rowscale_mode=_hip.SCALE_MUL)
        else:
            right = _hip.scale_cols(Vr, sig, _hip.SCALE_MUL).transpose(1, 2).contiguous()
    else:
        left = _hip.gemm(Mw, Vr)
        if V1 is not None:
            right = _hip.gemm(Vr, V1, transA=True, transB=True)
        else:
            right = Vr.transpose(1, 2).contiguous()
    st.lap("Time (product):")
    return Truncation(left, None, right, r, info=info, gtrace=gtr)


_INPLACE_FIRST_ROWS = 1 << 22   # rows of the in-place projection that go through a bounded temporary (256 MB at 16 kept columns)


def _colproject_inplace(M: torch.Tensor, V1, V, sig, r: int, left_ortho: bool):
    """``_hip.colproject`` for ONE tall contiguous matrix [1, rows, n] whose storage may be consumed: ``left`` (rows x r, r <= n / 2)
    is written over the FRONT of M's own storage -- output row i (r values at element offset i r) lies inside input row i r / n --
    so a dense tensor that fills the GPU (BASELINE config C1 at its stated 64^6 = 256 GiB, whose 64 GiB carry does not fit next
    to it) is decomposed without a second buffer.  Order: the rows are processed in ranges [a, b) with b <= a n / r (one launch
    each; the range's output lands in input rows < a, which earlier launches have consumed -- the kernel boundary is the
    ordering), the first range through a bounded temporary that is copied to the front last.  Returns (left view, right)."""
    Bt, rows, n = M.shape
    assert Bt == 1 and M.is_contiguous() and 2 * r <= n
    flat = M.reshape(-1)
    U = _hip.gemm(V1, V[:, :, :r]) if V1 is not None else V[:, :, :r].contiguous()
    a = min(rows, _INPLACE_FIRST_ROWS)
    first, right = _hip.colproject(M[:, :a], None, U, sig, r, left_ortho)
    while a < rows:
        b = min(rows, (a * n) // r)
        _hip.colproject(M[:, a:b], None, U, sig, r, left_ortho, left_out=flat[a * r:b * r].view(1, b - a, r))
        a = b
    flat[:first.numel()].view(first.shape).copy_(first)
    return flat[:rows * r].view(1, rows, r), right


def _scale_batch(X: torch.Tensor, e: torch.Tensor, sign: int) -> torch.Tensor:
    """X[b] * 2^(sign * e[b]) (exact) for a [B, ...] tensor (ttr_scale_batch)."""
    return _hip.scale_batch(X, expo=e, expo_sign=sign)


def _range_guard(X: torch.Tensor):
    """fp32 only: binary exponents (per batch item) to take out of ``X`` before Gram matrices are formed from it,
    or None when every item lies within 2^+-40 (squares far from the fp32 range limits).  The rounding sweep
    normalises its own operands; this is for the entries that receive user data directly."""
    if X.dtype != torch.float32:
        return None
    return _range_guard_from_norms(_hip.norm(X.reshape(X.shape[0], -1)))


def _range_guard_from_norms(nr: torch.Tensor):
    """The exponents of ``_range_guard`` from the items' Frobenius norms [B] (ttr_norm streams a long item with the
    whole chip; ttr_pow2_normalize is one workgroup per item -- on a 192 GiB tensor that was 80 s)."""
    if nr.dtype != torch.float32:
        return None
    _, e = _hip.pow2_normalize(nr.reshape(-1, 1), exponent_only=True)
    if int(e.abs().max().item()) < 40:  # (readback: control flow only)
        return None
    return e


def _zero_factors(M3):
    Bt, m, n = M3.shape
    return (torch.zeros((Bt, m, 1), dtype=M3.dtype, device=M3.device), torch.zeros((Bt, 1, n), dtype=M3.dtype, device=M3.device))


def _truncated_svd_batch(M3, rmax, left_ortho, algorithm):
    """Batch mode of ``truncated_svd``: the rank is not data dependent (round.py:149-150), so the kernels are enqueued
    optimistically and the two data-dependent exceptions -- an fp32 item outside the 2^+-40 exponent window (the range
    guard: redo with scaled inputs) and an all-zero batch (round.py:138-141: rank-1 zeros) -- are decided from flags read
    back ONCE, after everything has been enqueued (no host wait in front of any kernel)."""
    Bt = M3.shape[0]
    range_flag = None
    if M3.dtype == torch.float32:
        _, e = _hip.pow2_normalize(_hip.norm(M3.reshape(Bt, -1)).reshape(-1, 1), exponent_only=True)
        range_flag = _deferred_readback(e.abs().amax())
    t = truncate(M3, None, rmax, left_ortho, algorithm, True)
    zero_flag = _deferred_readback(t.info.amax()) if t.info is not None else None
    if range_flag is not None and int(range_flag().item()) >= 40:
        return None  # (rare) the caller takes the scaled path
    if zero_flag is not None and int(zero_flag().item()) == 0:
        return _zero_factors(M3)
    left = t.left_scaled()
    return (left if left.is_contiguous() else left.contiguous()), t.right


def truncated_svd(M3, delta, eps, rmax, left_ortho, algorithm, batch):
    """round.py:52-187 on a [B, m, n] device tensor -> (left [B, m, r], M2 [B, r, n])."""
    if batch:
        res = _truncated_svd_batch(M3, rmax, left_ortho, algorithm)
        if res is not None:
            return res
    e = _range_guard(M3)
    if e is not None:  # out-of-range fp32 scale: work on M 2^-e, give the exponent back to the non-orthonormal factor
        M3 = _scale_batch(M3, e, -1)
        if delta is not None:
            delta = delta * 2.0 ** (-int(e.max().item()))  # (absolute bound: only meaningful for a single matrix)
    if delta is None and eps is not None:  # round.py:79-80
        delta = eps * float(_hip.norm(M3.reshape(1, -1))[0].item())
    if delta is None:
        delta = 0.0
    t = truncate(M3, delta, rmax, left_ortho, algorithm, batch)
    if batch and t.info is not None and int(t.info.amax().item()) == 0:  # round.py:138-141 (scaled path: exact zeros only)
        return _zero_factors(M3)
    left = t.left_scaled()
    if not left.is_contiguous():
        left = left.contiguous()
    right = t.right
    if e is not None:
        if left_ortho:
            right = _scale_batch(right, e, +1)
        else:
            left = _scale_batch(left, e, +1)
    return left, right


# ----------------------------------------------------------------------------------------------
def mode_mul(core4: torch.Tensor, M3: torch.Tensor) -> torch.Tensor:
    """[B, r0, S, r1] x_2 [B, a, S] -> [B, r0, a, r1] (the einsum of tensor.py:1790-1798, 1999-2002) as ONE batched GEMM
    over the B * r0 slices ``core[b, r0]`` (S x r1, contiguous where they lie) with the small matrix as the left
    operand: out[b, r0] = M[b] @ core[b, r0].  Neither the core nor the result is permuted (round 2 moved both through
    layout copies); the a x S matrix is the only thing replicated (r0 times, KB)."""
    Bt, r0, S, r1 = core4.shape
    a = M3.shape[1]
    core4 = core4 if core4.is_contiguous() else core4.contiguous()
    Mrep = M3.reshape(Bt, 1, a, S).expand(Bt, r0, a, S).reshape(Bt * r0, a, S)
    out = _hip.gemm(Mrep, core4.reshape(Bt * r0, S, r1))  # [B * r0, a, r1]
    return out.reshape(Bt, r0, a, r1)


def merge_swap(c1: torch.Tensor, c2: torch.Tensor) -> torch.Tensor:
    """Two neighbouring cores [B, R1, I1, R2], [B, R2, I2, R3] contracted over the bond with their modes exchanged
    (``einsum("iaj,jbk->ibak")``, tools.py:680-681) -> [B, R1*I2, I1*R3]: one batched MFMA GEMM of the left unfolding
    with the right unfolding; the mode exchange is a layout copy of the product."""
    Bt, R1, I1, R2 = c1.shape
    _, _, I2, R3 = c2.shape
    prod = _hip.gemm(c1.reshape(Bt, R1 * I1, R2), c2.reshape(Bt, R2, I2 * R3))  # [B, R1*I1, I2*R3]
    return prod.reshape(Bt, R1, I1, I2, R3).permute(0, 1, 3, 2, 4).reshape(Bt, R1 * I2, I1 * R3)


def diag_sum(core5: torch.Tensor) -> torch.Tensor:
    """[B, r0, a, a, r1] -> [B, r0, r1]: sum over the diagonal of the two mode axes (matrix.py:160-175) as a GEMM of the
    gathered diagonal slices with a ones vector."""
    Bt, r0, a, _, r1 = core5.shape
    d = torch.diagonal(core5, dim1=2, dim2=3).contiguous()  # [B, r0, r1, a] (layout copy)
    return _sum_last(d.reshape(Bt, r0 * r1, a)).reshape(Bt, r0, r1)


def factor_orthogonalize(c: List[torch.Tensor], Us, mu: int) -> None:
    """tensor.py:1771-1798: QR of the Tucker factor [B, I, S], R pushed into the core."""
    if Us is None or Us[mu] is None:
        return
    Q, R = qr(Us[mu].contiguous())
    Us[mu] = Q
    c[mu] = mode_mul(c[mu], R)


def left_orthogonalize(c: List[torch.Tensor], mu: int, Us=None) -> torch.Tensor:
    """tensor.py:1800-1833 on [B, r0, I, r1] cores (in place on the list); returns R [B, k, r1]."""
    factor_orthogonalize(c, Us, mu)
    Bt, r0, I, r1 = c[mu].shape
    Q, R = qr(c[mu].reshape(Bt, r0 * I, r1))
    k = Q.shape[2]
    c[mu] = Q.reshape(Bt, r0, I, k)
    nxt = c[mu + 1]
    pushed = _hip.gemm(R, nxt.reshape(Bt, nxt.shape[1], nxt.shape[2] * nxt.shape[3]))
    c[mu + 1] = pushed.reshape(Bt, k, nxt.shape[2], nxt.shape[3])
    return R


def right_orthogonalize(c: List[torch.Tensor], mu: int, Us=None) -> torch.Tensor:
    """tensor.py:1835-1879: QR of the transposed right unfolding; returns L [B, r0, k]."""
    factor_orthogonalize(c, Us, mu)
    Bt, r0, I, r1 = c[mu].shape
    if r0 <= _hip.max_qr_cols(c[mu].dtype):
        # the right unfolding is the TRANSPOSE of the matrix to factor: the QR kernels address it (and Q^T, which is the new
        # core as it lies) through strides -- no transposed copies of the core (round 2: three layout copies)
        Qt, Lt = _hip.qr_t(c[mu].reshape(Bt, r0, I * r1))  # unfolding^T (I r1 x r0) = Q Lt;  Qt = Q^T (k x I r1)
        k = Qt.shape[1]
        c[mu] = Qt.reshape(Bt, k, I, r1)
    else:
        Mt = c[mu].reshape(Bt, r0, I * r1).transpose(1, 2).contiguous()  # more than one TSQR panel of columns: blocked QR on a copy
        Q, Lt = qr(Mt)  # Mt (I r1 x r0) = Q (I r1 x k) Lt (k x r0)
        k = Q.shape[2]
        c[mu] = Q.transpose(1, 2).contiguous().reshape(Bt, k, I, r1)
    L = Lt.transpose(1, 2).contiguous()  # r0 x k (a rank-sized matrix)
    prev = c[mu - 1]
    pushed = _hip.gemm(prev.reshape(Bt, prev.shape[1] * prev.shape[2], r0), L)
    c[mu - 1] = pushed.reshape(Bt, prev.shape[1], prev.shape[2], k)
    return L


class SumCore:
    """Middle core of a TT sum a + b (tensor.py:445-668) kept as its two diagonal blocks: ``blockdiag(a, b)`` with
    a [B, ra, I, ca], b [B, rb, I, cb] is what ``__add__`` would materialise (a core (ra+rb)^2 / (ra^2+rb^2) times larger,
    half of it zeros).  The L2R sweep feeds the blocks straight to ``ttr_qr_factor_pushed_sum``."""

    __slots__ = ("a", "b")

    def __init__(self, a: torch.Tensor, b: torch.Tensor):
        assert a.dim() == 4 and b.dim() == 4 and a.shape[0] == b.shape[0] and a.shape[2] == b.shape[2]
        self.a, self.b = a, b

    @property
    def shape(self):
        return torch.Size((self.a.shape[0], self.a.shape[1] + self.b.shape[1], self.a.shape[2], self.a.shape[3] + self.b.shape[3]))

    @property
    def dtype(self):
        return self.a.dtype

    @property
    def device(self):
        return self.a.device

    def __getitem__(self, idx):
        return SumCore(self.a[idx], self.b[idx])

    def dense(self) -> torch.Tensor:
        a, b = self.a, self.b
        za = a.new_zeros(a.shape[:-1] + (b.shape[-1],))
        zb = b.new_zeros(b.shape[:-1] + (a.shape[-1],))
        return torch.cat([torch.cat([a, za], dim=-1), torch.cat([zb, b], dim=-1)], dim=1)


def sum_cores(ca: Sequence[torch.Tensor], cb: Sequence[torch.Tensor]) -> List:
    """Cores of a + b for [B, r0, I, r1] trains: first / last cores concatenated (small), middle cores lazy."""
    N = len(ca)
    if N == 1:
        return [ca[0] + cb[0]]  # (only reached for one-mode tensors; a plain vector sum)
    out: List = [torch.cat([ca[0], cb[0]], dim=-1)]
    out += [SumCore(a, b) for a, b in zip(ca[1:-1], cb[1:-1])]
    out.append(torch.cat([ca[-1], cb[-1]], dim=1))
    return out


class _ExplicitQ:
    """Stand-in for ``_hip.QrFactors`` when Q had to be formed (blocked QR above the TSQR column limit)."""

    def __init__(self, Q: torch.Tensor, R: torch.Tensor):
        self.Q, self.R, self.batch = Q, R, Q.shape[0]


# The apply kernel can accumulate the row Gram matrix of its output itself (ttr_qr_apply_pushed_gram: one pass over the new
# carry less, -6 GB of HBM traffic per metric step).  Measured on MI355X it is 1.6 % SLOWER end to end (32.1 vs 31.6 ms/step:
# the +140 us of MFMA work per apply launch do not overlap, while the stand-alone HBM-bound rowgram launch overlaps with the
# other sub-batch's eigensolver), so it is opt-in.
_FUSE_APPLY_GRAM = os.environ.get("TTR_FUSE_APPLY_GRAM", "0") == "1"


if _FUSE_APPLY_GRAM:   # (its epilogue assumes the unpacked row map of the level-0 blocks)
    _hip.set_knob(_hip.KNOB_QR_PACK, 0)


def _apply_q(f, C: torch.Tensor, out: Optional[torch.Tensor] = None, want_gram: bool = False, skip_zero_rows: bool = False):
    if isinstance(f, _ExplicitQ):
        Q = _hip.gemm(f.Q, C, out=out)
        return (Q, None) if want_gram else Q
    return _hip.qr_apply(f, C, out=out, want_gram=want_gram, skip_zero_rows=skip_zero_rows)


# Sub-batch streams.  Within one tensor train the sweeps are a dependency chain, and some of its kernels are
# latency-bound at any batch size the chip can hold (the one-wave-per-matrix eigensolver keeps 2 waves per SIMD
# busy at B = 2048).  A batch is therefore cut into sub-batches that run the whole sweep on their own HIP
# streams: the latency-bound kernels of one sub-batch overlap the bandwidth-bound ones of the others (measured
# on the metric workload: B = 2048 50.5 -> 48.0 ms/step with 2 streams, B = 512 18.2 -> 16.1 ms).  The
# results are written straight into full-batch tensors (no concatenation pass).
STREAM_CHUNKS_ENABLED = True
_SIDE_STREAMS: dict = {}


def _stream_chunks(Bt: int, batch: bool) -> int:
    if not (batch and STREAM_CHUNKS_ENABLED) or VERBOSE:
        return 1
    forced = os.environ.get("TTR_STREAM_CHUNKS")  # (experiments)
    if forced:
        return max(1, min(int(forced), Bt))
    return 2 if Bt >= 128 else 1  # (4 streams measured between -7 % and +7 % from box to box, 2 streams -5 % always)


def _side_streams(dev: torch.device, n: int):
    """The sub-batch streams that belong to the CALLER's current stream: two calls issued from different streams (a
    program that keeps several independent batches in flight) get different side streams and overlap freely -- the
    latency-bound eigensolver periods of one call are filled by the QR kernels of the other."""
    di = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (di, n, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _SIDE_STREAMS[key]


class _OutArena:
    """Result tensors of a chunked sweep: ONE flat full-batch buffer holding the rounded cores back to back
    (core-major, each core [B, r, I, r'] contiguous), every sub-batch writes its own batch slice of every core.
    The layout is exactly what ``dist_batch.pack_cores`` produces, so the multi-GPU gather sends it without a
    packing copy."""

    def __init__(self, total: int, bounds, main_stream, tails, like: torch.Tensor):
        self.total, self.bounds, self.tails = total, bounds, tails
        sizes = [total * t[0] * t[1] for t in tails]
        with torch.cuda.stream(main_stream):  # owned by the caller's stream, like any other result
            self.flat = torch.empty(sum(sizes), dtype=like.dtype, device=like.device)
        self.full, off = [], 0
        for t, sz in zip(tails, sizes):
            self.full.append(self.flat[off:off + sz].view(total, t[0], t[1]))
            off += sz

    def slice(self, key: int, chunk: int, tail) -> torch.Tensor:
        assert tuple(tail) == tuple(self.tails[key]), (key, tail, self.tails[key])
        lo, hi = self.bounds[chunk]
        return self.full[key][lo:hi]


def _rounded_tails(shapes, rmax):
    """Shapes of the cores round_tt produces in batch mode (no data-dependent ranks, round.py:149-150), as the
    2-D tails the arena stores: core 0 -> (r0*I, r1), core mu > 0 -> (r, I*r')."""
    N = len(shapes)
    lr = [shapes[0][0]]
    for mu in range(N - 1):
        lr.append(min(lr[mu] * shapes[mu][1], shapes[mu][2]))
    out_r = [None] * (N + 1)
    out_r[N] = shapes[N - 1][2]
    for mu in range(N - 1, 0, -1):
        k = min(lr[mu], shapes[mu][1] * out_r[mu + 1])
        out_r[mu] = _rank_cap(rmax[mu - 1], k)
    out_r[0] = lr[0]
    tails = [(out_r[0] * shapes[0][1], out_r[1])]
    tails += [(out_r[mu], shapes[mu][1] * out_r[mu + 1]) for mu in range(1, N)]
    return tails, out_r


def round_tt(
    cores4: Sequence[torch.Tensor],
    eps: float,
    rmax: Sequence[Optional[int]],
    algorithm: str,
    batch: bool,
    Us=None,
) -> List[torch.Tensor]:
    """tensor.py:2008-2083 on [B, r0, I, r1] cores.  Returns new cores (inputs untouched).
    ``Us``: Tucker factors; those of modes 0..N-2 are orthogonalised in place first (tensor.py:1815 inside the
    L2R sweep; independent of the core QRs, so hoisting them keeps the fused sweep).

    Same two sweeps as the reference, with one fusion: the left-orthogonal cores Q_mu of the L2R sweep
    are never materialised.  Each QR leaves its reflectors in a workspace; in the R2L sweep the core
    that the reference obtains as ``einsum(Q_mu, U sigma)`` (tensor.py:2081-2083) is produced directly
    by applying the reflectors to ``[U sigma; 0]`` (``ttr_qr_apply``): half the columns, no Q round
    trip through HBM, no separate push-left GEMM.  Large batches run as sub-batches on separate HIP streams
    (see ``_stream_chunks``).
    """
    c = list(cores4)
    N = len(c)
    for i in range(N - 1):
        factor_orthogonalize(c, Us, i)
    Bt = c[0].shape[0]
    nchunk = _stream_chunks(Bt, batch)
    # Batch mode, zero guard of round.py:137-141: when EVERY item's sigma_max lies below 1e-13 at the first truncation the
    # reference returns rank-1 zeros for the whole batch (and then for every further bond: the carry is zero).  The ranks
    # of a batch are otherwise not data dependent, so the sweep is enqueued for them without waiting; the per-item flags
    # the eigensolver epilogue leaves behind (info == 0) are reduced on the device, copied to pinned host memory and only
    # looked at after the last kernel of the sweep has been enqueued.
    zflags: Optional[list] = [] if batch else None
    shapes = [tuple(x.shape[1:]) for x in c]

    def all_zero_batch() -> bool:
        # (every flag is waited for, also when the first one already decides: a pinned word must not be handed back to the
        # allocator before the device has written it)
        vals = [int(w().reshape(-1)[0].item()) for w in (zflags or ())]   # (non-batch calls: no flags at all)
        return bool(vals) and all(v == 0 for v in vals)

    def zero_train():
        return [torch.zeros((Bt, shapes[0][0] if mu == 0 else 1, shapes[mu][1], shapes[N - 1][2] if mu == N - 1 else 1),
                            dtype=c[0].dtype, device=c[0].device) for mu in range(N)]

    if nchunk == 1:
        out = _round_tt_sweep(c, eps, rmax, algorithm, batch, None, 0, zflags)
        return zero_train() if all_zero_batch() else out
    dev = c[0].device
    main = torch.cuda.current_stream(dev)
    streams = _side_streams(dev, nchunk)
    q, rem = divmod(Bt, nchunk)
    bounds, lo = [], 0
    for ci in range(nchunk):
        hi = lo + q + (1 if ci < rem else 0)
        bounds.append((lo, hi))
        lo = hi
    tails, out_r = _rounded_tails(shapes, rmax)
    arena = _OutArena(Bt, bounds, main, tails, c[0])
    for ci, st in enumerate(streams):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            lo, hi = bounds[ci]
            _round_tt_sweep([x[lo:hi] for x in c], eps, rmax, algorithm, batch, arena, ci, zflags)
    for st in streams:
        main.wait_stream(st)
    if all_zero_batch():
        return zero_train()
    return [arena.full[mu].view(Bt, out_r[mu], shapes[mu][1], out_r[mu + 1]) for mu in range(N)]


# The whole sweep behind ONE library call (ttr_round_tt, csrc/ttr_roundtt.hip): same kernels, same order, bit-identical results --
# the ~80 launches of a train are enqueued from C++ instead of one ctypes call each (the host cost of a call: 1.2 ms -> what the
# device needs).  TTR_SWEEP_C=0: always the Python loop below (A/B; tests cross-check the two).
SWEEP_C_ENABLED = os.environ.get("TTR_SWEEP_C", "1") != "0"
# The host loop's fp32 factorisations normalise their R factors themselves (ttr_qr_factor_expo / ttr_qr_factor_pushed_expo, ABI 11)
# instead of one ttr_pow2_normalize launch per core; TTR_FUSE_QR_NORM=0: the separate launches (A/B; the results are bit-identical
# either way -- the scalings are exact powers of two -- and the one-call entry always uses the fused form).
FUSE_QR_NORM = os.environ.get("TTR_FUSE_QR_NORM", "1") != "0"
SWEEP_C_CALLS = 0   # (tests: how many sweeps went through ttr_round_tt)


def _round_tt_sweep_c(c, eps, rmax, algorithm, batch, arena, chunk, zflags) -> Optional[List[torch.Tensor]]:
    """``_round_tt_sweep`` through ttr_round_tt, or None when the train lies outside that entry's envelope (TT ranks above 64,
    cores outside the fused push, bonds with more rows than columns, fused sums, the eps-mode cases the Python loop reads back
    bond by bond)."""
    global SWEEP_C_CALLS
    N = len(c)
    if N < 2 or any(not torch.is_tensor(x) for x in c) or c[0].dtype not in (torch.float32, torch.float64):
        return None
    Bt = c[0].shape[0]
    if Bt < 1 or any(x.dtype != c[0].dtype or x.shape[0] != Bt for x in c):
        return None
    shapes = [tuple(x.shape[1:]) for x in c]
    eps_mode = not batch
    rcap = [_hip.RANK_NONE if r is None else max(1, min(int(r), _hip.RANK_NONE)) for r in rmax]
    if eps_mode:
        # the reference's non-batch rule with the ranks kept on the device (see ``_eps_deferred_ok``: same policy)
        mode = os.environ.get("TTR_EPS_DEFERRED", "auto")
        if mode == "0" or Bt != 1:
            return None
        k, elems = shapes[0][0], 0
        for mu in range(N - 1):
            elems += k * shapes[mu][1] * shapes[mu][2]
            k = min(k * shapes[mu][1], shapes[mu][2])
        elems += k * shapes[N - 1][1] * shapes[N - 1][2]
        if mode != "1" and (elems > _EPS_DEFERRED_MAX_ELEMS or any(r is None for r in rmax)):
            return None
    wsb = _hip.round_tt_plan(c[0].dtype, shapes, rcap, Bt, eps_mode)
    if wsb < 0:
        return None
    tails, out_r = _rounded_tails(shapes, rmax)
    dev, dt = c[0].device, c[0].dtype
    if arena is not None:
        outs = [arena.slice(mu, chunk, tails[mu]) for mu in range(N)]
    else:
        outs = [torch.empty((Bt,) + tuple(tails[mu]), dtype=dt, device=dev) for mu in range(N)]
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    ranks_dev = torch.empty(N - 1, dtype=torch.int32, device=dev) if eps_mode else None
    # the zero-guard flag (round.py:138-141 for the whole batch): the sweep's reduction kernel writes it STRAIGHT into pinned host
    # memory right after the first truncation (ttr_round_tt accepts any device-accessible address), the host polls the word.  As a
    # device word copied back after the call (rounds 5: a copy + event at the END of the stream) the caller waited for the whole
    # sweep before it could return -- one 64^8 train: 1.27 ms per call where the host loop, whose flag travels early, took 1.18.
    zf = None
    if zflags is not None and not eps_mode:
        zf = torch.empty(1, dtype=torch.int32, pin_memory=True)
        zf[0] = _ZF_PENDING
    elif eps_mode and N - 1 <= 64:
        # (eps mode: the same word pattern for the selected ranks -- an early copy of ranks_dev in pinned host memory)
        zf = torch.full((N - 1,), _ZF_PENDING, dtype=torch.int32).pin_memory()
    use_top = EIGH_TOP_ENABLED and FLAT_SPECTRUM_THR > 0
    try:
        _hip.round_tt_sweep([x.contiguous() for x in c], rcap, algorithm, eps_mode, eps if eps is not None else 0.0,
                            max(FLAT_SPECTRUM_THR, 0.0), use_top, outs, ranks_dev, zf, ws)
    except NotImplementedError:
        # a per-kernel limit the planner does not mirror (TTR_E_UNSUPPORTED from inside the sweep): ttr_round_tt never writes its
        # inputs, so the loop over the per-kernel entries -- which asks every kernel's own predicate -- starts from the same state
        return None
    SWEEP_C_CALLS += 1
    if zf is not None and not eps_mode:
        zflags.append(_pinned_flag_wait(zf))
    out = [outs[mu].view(Bt, out_r[mu], shapes[mu][1], out_r[mu + 1]) for mu in range(N)]
    if eps_mode:
        # the ONE host synchronisation of the sweep: the selected ranks; the cores -- computed at their caps, zero beyond the
        # selected ranks -- are cut to size (layout copies)
        # (polled from the pinned copy: available before the sweep's last kernels have run)
        if zf is not None:
            _pinned_flag_wait(zf)()
            ranks = zf.tolist()                         # ranks[mu - 1] = rank of bond mu
        else:
            ranks = ranks_dev.tolist()
        if min(ranks) == 0:  # zero guard (round.py:137-145): the carry was zero from the first bond on
            return [torch.zeros((1, shapes[0][0] if mu == 0 else 1, shapes[mu][1], shapes[N - 1][2] if mu == N - 1 else 1),
                                dtype=dt, device=dev) for mu in range(N)]
        bond = [shapes[0][0]] + ranks + [shapes[N - 1][2]]
        out = [x[:, :bond[mu], :, :bond[mu + 1]].contiguous() for mu, x in enumerate(out)]
    return out


def _round_tt_sweep(c, eps, rmax, algorithm, batch, arena, chunk, zflags=None) -> List[torch.Tensor]:
    """The two sweeps on one (sub-)batch; with an ``arena`` the resulting cores are written into its slices.
    ``zflags``: list that receives the deferred readback of "largest rank-rule result of the first truncation" (0 = every
    item of this sub-batch hit the zero guard); see ``round_tt``."""
    N = len(c)
    if SWEEP_C_ENABLED and not VERBOSE and not _FUSE_APPLY_GRAM:
        done = _round_tt_sweep_c(c, eps, rmax, algorithm, batch, arena, chunk, zflags)
        if done is not None:
            return done
    facs = []
    st = _Stage()
    Rprev = None  # R factor still to be pushed into the current core
    expo = None   # fp32: accumulated binary exponent taken out of the R factors (per batch item)
    for mu in range(N - 1):  # L2R: tensor.py:1905-1906 (Q implicit, push fused into the next QR)
        Bt, r0, I, r1 = c[mu].shape
        rows_k = r0 if Rprev is None else Rprev.shape[1]
        if isinstance(c[mu], SumCore):
            if Rprev is not None and _hip.pushed_supported(Rprev.shape[1], r0, I, r1, c[mu].dtype):
                f = _hip.qr_factor_pushed_sum(Rprev, c[mu].a, c[mu].b)  # blockdiag(a, b) is never materialised
                facs.append((f, rows_k, I))
                Rprev = f.R
                if Rprev.dtype == torch.float32:
                    if expo is None:
                        expo = torch.zeros(Bt, dtype=torch.int32, device=Rprev.device)
                    Rprev, _ = _hip.pow2_normalize(Rprev, expo_acc=expo)
                c[mu] = None
                continue
            c[mu] = c[mu].dense()  # ranks above the fused kernel's 64 columns: the padded core after all
        # fp32: ||R_mu|| is the norm of the partially contracted tensor and grows like (I r)^(mu/2): the squared
        # column norms / Gram entries of a high-order train overflow fp32 (LAPACK rescales internally).  Every R
        # is therefore brought back to O(1) by an exact power of two per batch item -- by the factor kernel itself
        # (``expo_acc``: ttr_qr_factor_expo; rounds 1 - 4 and the explicit-Q path: one ttr_pow2_normalize launch) -- ; the
        # exponents are summed on the device and returned to core 0 at the end, so the result is bit-identical whenever
        # nothing overflowed.
        f32 = c[mu].dtype == torch.float32
        if f32 and expo is None:
            expo = torch.zeros(Bt, dtype=torch.int32, device=c[mu].device)
        ex = expo if (f32 and FUSE_QR_NORM) else None
        if r1 > _hip.max_qr_cols(c[mu].dtype):
            # more columns than a TSQR panel holds (TT rank > 64): explicit Q from the blocked QR
            A = c[mu] if Rprev is None else _hip.gemm(Rprev, c[mu].reshape(Bt, r0, I * r1))
            f = _ExplicitQ(*qr(A.reshape(Bt, rows_k * I, r1)))
            ex = None
        elif Rprev is None:
            f = _hip.qr_factor(c[mu].reshape(Bt, r0 * I, r1), expo_acc=ex)
        elif _hip.pushed_supported(Rprev.shape[1], r0, I, r1, c[mu].dtype):
            f = _hip.qr_factor_pushed(Rprev, c[mu], expo_acc=ex)  # QR of (Rprev @ core) without materialising it
        else:
            pushed = _hip.gemm(Rprev, c[mu].reshape(Bt, r0, I * r1)).reshape(Bt, Rprev.shape[1], I, r1)
            f = _hip.qr_factor(pushed.reshape(Bt, Rprev.shape[1] * I, r1), expo_acc=ex)
        facs.append((f, rows_k, I))
        Rprev = f.R
        if f32 and ex is None:
            Rprev, _ = _hip.pow2_normalize(Rprev, expo_acc=expo)
        c[mu] = None
    last = c[N - 1]
    # the first truncation's carry M = R x (last core): rows 32.. of M are negligible whenever those of R are -- the flags the
    # packed push computes for every other bond (rank-inflated trains: half the rows of the first Gram / projection passes, and
    # the first eigenproblem shrinks to 32 x 32 like the others)
    r32_last = None
    c[N - 1] = _hip.gemm(Rprev, last.reshape(last.shape[0], last.shape[1], -1)).reshape(
        last.shape[0], Rprev.shape[1], last.shape[2], last.shape[3])
    if (not _FUSE_APPLY_GRAM and Rprev is not None and Rprev.dim() == 3 and Rprev.shape[1] == 64 and Rprev.shape[0] == last.shape[0]
            and last.shape[2] * last.shape[3] >= 64):
        # (the flag is computed on the carry M itself: what is dropped is below c eps ||M||_F whatever the conditioning of the last
        # core -- round 4 tested R, whose small rows bound those of M only up to ||last||_2)
        r32_last = _hip.carry_rows32(c[N - 1].reshape(last.shape[0], Rprev.shape[1], -1))
    if expo is not None:  # the last core carries ||X|| / 2^expo: bring it to O(1) as well (see above)
        c[N - 1], _ = _hip.pow2_normalize(c[N - 1], expo_acc=expo)
    st.lap("Orthogonalization time:")   # tensor.py:2032-2035 (here: the factorisations; Q stays implicit)
    d2dev, infos = None, []
    if batch:  # tensor.py:2036-2037
        delta = None
    elif _eps_deferred_ok(c, facs, rmax):
        # tensor.py:2039-2051 without the `.item()`: delta^2 = (eps / max(1, sqrt(N - 1)))^2 ||last core||^2 stays on the
        # device (three scalar ops), every bond is enqueued at its rank cap, ONE readback of the N - 1 ranks at the end
        delta = 0.0
        d2dev = _hip.norm(c[-1].reshape(1, -1)).double().square_().mul_((eps / max(1.0, math.sqrt(N - 1))) ** 2)
    else:  # tensor.py:2039-2051
        nrm = float(_hip.norm(c[-1].reshape(1, -1))[0].item())
        delta = eps / max(1.0, math.sqrt(N - 1)) * nrm
    left = None  # (U sigma) of the bond to the right, to be absorbed by the current core
    for mu in range(N - 1, 0, -1):  # R2L: tensor.py:2053-2083
        gram = None
        if mu == N - 1:
            M4 = c[mu]
        else:
            f, r0, I = facs[mu]
            if _FUSE_APPLY_GRAM:
                M4, gram = _apply_q(f, left, want_gram=True)  # the apply kernel also accumulates M M^T where it can
            else:
                # (the carry only goes to the rows32-aware Gram / projection kernels below: the exactly-zero rows kk >= 32 of a
                # packed item stay unwritten)
                M4 = _apply_q(f, left, skip_zero_rows=(r0 == 64 and I * left.shape[2] >= 64))   # (>= 64 columns: the fused row kernels)
            M4 = M4.reshape(f.batch, r0, I, left.shape[2])
        Bt, R, I, rn = M4.shape
        # rows kk >= 32 of the carry are exactly zero for the items whose QR of this bond packed its rows
        if mu < N - 1:
            r32 = getattr(facs[mu][0], "rows32", None) if (R == 64 and I * rn >= 64) else None
        else:
            r32 = r32_last if (R == 64 and I * rn >= 64) else None
        alloc = None
        if arena is not None:
            def alloc(r, mu=mu, n=I * rn):
                return arena.slice(mu, chunk, (r, n))
        t = truncate(M4.reshape(Bt, R, I * rn), delta, rmax[mu - 1], False, algorithm, batch, alloc, gram=gram, delta2_dev=d2dev,
                     rows32=r32)
        if d2dev is not None:
            infos.append(t.info)
        if zflags is not None and mu == N - 1 and t.info is not None:
            zflags.append(_deferred_readback(t.info.amax()))  # ([B] int32 -> one scalar: control flow only)
        right = t.right
        if arena is not None:
            dst = alloc(t.rank)
            if right.data_ptr() != dst.data_ptr():  # branch of truncate() that does not write in place
                dst.copy_(right)
                right = dst
        c[mu] = right.reshape(Bt, t.rank, I, rn)
        left = t.left_scaled()
    f, r0, I = facs[0]
    if expo is not None:  # give the exponents back (exact); a norm beyond the fp32 range overflows here, as it must
        left = _hip.scale_batch(left, expo=expo, expo_sign=+1)
    dst = None
    if arena is not None:
        dst = arena.slice(0, chunk, (r0 * I, left.shape[2]))
    c[0] = _apply_q(f, left, dst).reshape(f.batch, r0, I, left.shape[2])
    if d2dev is not None:
        # the ONE host synchronisation of the sweep: the selected ranks (bond N-1 first), then the cores -- computed at their
        # caps, zero beyond the selected ranks -- are cut to size (layout copies)
        ranks = torch.cat(infos).tolist()[::-1]          # ranks[mu - 1] = rank of bond mu
        if min(ranks) == 0:  # zero guard (round.py:137-145): the carry was zero from the first bond on
            return [torch.zeros((1, c[0].shape[1] if mu == 0 else 1, c[mu].shape[2], c[N - 1].shape[3] if mu == N - 1 else 1),
                                dtype=c[0].dtype, device=c[0].device) for mu in range(N)]
        bond = [c[0].shape[1]] + ranks + [c[N - 1].shape[3]]
        c = [x[:, :bond[mu], :, :bond[mu + 1]].contiguous() for mu, x in enumerate(c)]
    return c


_EPS_DEFERRED_MAX_ELEMS = 1 << 24   # eps-mode sweeps of trains up to this many core elements keep the ranks on the device


def _eps_deferred_ok(c, facs, rmax) -> bool:
    """Non-batch (eps-mode) sweep: can every bond be enqueued without reading its rank back?  Only worth it where the
    host synchronisations dominate (a small train: every kernel is latency-bound) AND the caller gave a rank cap (the
    factors are computed at the cap: `round_tt(rmax=r)` on one tensor -- eps defaults to 1e-14, tensor.py:2008-2014 --
    is the case this is for); only on the fused <= 64-row truncation kernels.  TTR_EPS_DEFERRED=1 / 0 forces / forbids it."""
    mode = os.environ.get("TTR_EPS_DEFERRED", "auto")
    if mode == "0":
        return False
    N = len(c)
    last = c[N - 1]
    if last.shape[0] != 1 or any(isinstance(f[0], _ExplicitQ) for f in facs):
        return False
    elems = last.numel() + sum(f[0].m * f[0].n for f in facs)
    if mode != "1" and (elems > _EPS_DEFERRED_MAX_ELEMS or any(r is None for r in rmax)):
        # (without a rank cap every bond would be computed at its FULL rank: measured on config C2 -- rank 64 in, 32 out, no
        # rmax -- 9.3 ms deferred against 8.3 ms with one readback per bond; with a cap the widths are those of the result)
        return False
    rn = last.shape[3]
    for mu in range(N - 1, 0, -1):
        m = last.shape[1] if mu == N - 1 else facs[mu][1]
        I = last.shape[2] if mu == N - 1 else facs[mu][2]
        n = I * rn
        if m > 64 or m > n:
            return False
        rn = _rank_cap(rmax[mu - 1], min(m, n))
    return True


def round_tucker(cores4: Sequence[torch.Tensor], Us, eps, rmax, ndims, algorithm, batch):
    """tensor.py:1911-2006 on [B, r0, S, r1] cores and [B, I, S] factors; returns (cores, Us).
    Same kernel sequence as the TT sweeps: TSQR of the mode unfolding, truncated SVD of the small factor,
    mode products as batched GEMMs."""
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
        c[mu] = Q.reshape(Bt, r0, r1, Q.shape[2]).permute(0, 1, 3, 2).contiguous()
        Um = _hip.gemm(Us[mu], R, transB=True)  # tensor.py:1986
        left, right = truncated_svd(Um, None, eps / math.sqrt(ndims), rmax[mu], True, algorithm, batch)
        Us[mu] = left
        c[mu] = mode_mul(c[mu], right)  # tensor.py:1999-2002
        if mu > 0:
            right_orthogonalize(c, mu, Us)
    return c, Us


def absorb_factors(cores4: Sequence[torch.Tensor], Us) -> List[torch.Tensor]:
    """Contract every Tucker factor into its core (what tensor.py:1639-1687 does per mode)."""
    return [c if U is None else mode_mul(c, U.contiguous()) for c, U in zip(cores4, Us)]


def dense_tucker_tt(X: torch.Tensor, ranks_tucker, ranks_tt, algorithm, batch):
    """``tn.Tensor(X, ranks_tucker=, ranks_tt=)`` for a dense device tensor (tensor.py:401-408).

    The reference builds the full-rank TT, rounds the Tucker ranks mode N-1 .. 0 and then the TT ranks.  On the
    orthogonalised full-rank train the factor of mode mu has the singular values / left singular vectors of the
    mode-mu unfolding of the (already partly truncated) tensor, so this is the sequentially truncated HOSVD:
    per mode, Gram + eigensolver + projection on the dense unfolding (same kernels as ``dense_tt_svd``),
    followed by the TT-SVD of the small Tucker core."""
    from . import _hostops

    Bt = X.shape[0]
    N = X.dim() - 1
    Us: List[Optional[torch.Tensor]] = [None] * N
    for mu in range(N - 1, -1, -1):
        shp = list(X.shape)
        Xp = X.movedim(mu + 1, 1).reshape(Bt, shp[mu + 1], -1)
        if not Xp.is_contiguous():
            Xp = Xp.contiguous()
        left, right = truncated_svd(Xp, None, 1e-14 / math.sqrt(N), ranks_tucker[mu], True, algorithm, batch)
        Us[mu] = left
        S = left.shape[2]
        rest = shp[1:mu + 1] + shp[mu + 2:]
        X = right.reshape([Bt, S] + rest).movedim(1, mu + 1).contiguous()
    if ranks_tt is None:
        return _hostops.full_rank_tt(X), Us  # views / identity cores only (layout)
    return dense_tt_svd(X, 1e-14, list(ranks_tt), algorithm, batch), Us


def first_gram(M: torch.Tensor) -> torch.Tensor:
    """The Gram matrix ``truncate`` forms first for M [B, m, n], by the kernel its path selection would use (``gram=``)."""
    Bt, m, n = M.shape
    if m <= n and _hip.sweep_fused_ok(M):
        return _hip.rowgram(M)
    if m > n and _hip.colsweep_fused_ok(M):
        return _hip.colgram(M)
    return _hip.gemm(M, M, transB=True) if m <= n else _hip.gemm(M, M, transA=True)


_LAZY_GUARD_BYTES = 1 << 28   # dense inputs from 256 MB: the fp32 range guard of a batch-mode TT-SVD comes from the first Gram matrix


def dense_tt_svd(
    X: torch.Tensor,
    eps: float,
    rmax: Sequence[Optional[int]],
    algorithm: str,
    batch: bool,
    _guard_scaled: Optional[torch.Tensor] = None,
    consume_input: bool = False,
) -> List[torch.Tensor]:
    """Dense [B, I_1..I_N] -> TT cores [B, r, I, r'] by a right-to-left TT-SVD on the unfoldings.
    ``consume_input`` (one tensor, fp32 in range / fp64): X's storage is overwritten by the first carry (see
    ``_colproject_inplace``) -- for inputs that leave no room for a carry next to them.

    Equivalent to ``_full_rank_tt`` + ``round_tt(eps, rmax)`` of the reference
    (tensor.py:10-104, 401-408): after the full left orthogonalisation the right unfolding
    of the last core has the singular values / right singular vectors of the dense
    unfolding, and ||last core|| = ||X|| (so delta is identical).
    """
    Bt = X.shape[0]
    shape = list(X.shape[1:])
    N = len(shape)
    if N == 1:
        return [X.reshape(Bt, 1, shape[0], 1).clone()]
    # Batch mode needs no delta, and the fp32 range guard can be read off the first bond's Gram matrix (its trace is ||X[b]||^2):
    # a config-scale input is then not read a third time for its norm (C3: 1.2 of 15 ms, C1: 33 of 730).  The first truncation is
    # enqueued optimistically; an out-of-range trace (rare) restarts the sweep on the scaled input.
    big = X.numel() * X.element_size() >= _LAZY_GUARD_BYTES
    lazy_guard = _guard_scaled is None and batch and X.dtype == torch.float32 and big
    e = _guard_scaled
    delta = None
    gram0 = None
    if _guard_scaled is None and not batch and big:
        # a single config-scale tensor (BASELINE C1: 128-192 GiB): ||X||^2 -- delta and the range guard -- is the trace of the first
        # bond's Gram matrix, which the first truncation needs anyway: formed here, handed on (``gram=``), no pass for the norm
        gram0 = first_gram(X.reshape(Bt, -1, shape[-1]))
        tr = float(_gram_trace(gram0)[0].item())      # (readback: delta, as tensor.py:2039-2051)
        if not math.isfinite(tr) or (X.dtype == torch.float32 and (tr >= 2.0 ** 80 or 0.0 < tr <= 2.0 ** -80)):
            gram0 = None                               # out of range: the norm pass and the scaled input (below)
        else:
            delta = eps / max(1.0, math.sqrt(N - 1)) * math.sqrt(max(tr, 0.0))
    if gram0 is None and not lazy_guard and _guard_scaled is None and (not batch or X.dtype == torch.float32):
        nr = _hip.norm(X.reshape(Bt, -1))  # ONE pass over the input: delta and the fp32 range guard both come from it
        if not batch:
            delta = eps / max(1.0, math.sqrt(N - 1)) * float(nr[0].item())
        e = _range_guard_from_norms(nr)
        if e is not None:  # ||X|| outside 2^+-40 in fp32: every bond's Gram matrix would leave the range
            X = _scale_batch(X, e, -1)
            if delta is not None:
                delta = delta * 2.0 ** (-int(e[0].item()))
    cores: List[Optional[torch.Tensor]] = [None] * N
    C = X.reshape(Bt, -1, shape[-1])
    rn = 1
    for kdim in range(N - 1, 0, -1):
        Mk = C.reshape(Bt, -1, shape[kdim] * rn)
        first = lazy_guard and kdim == N - 1
        t = truncate(Mk, delta, rmax[kdim - 1], False, algorithm, batch, scratch_ok=kdim < N - 1, want_trace=first,
                     gram=gram0 if kdim == N - 1 else None,  # C is our own carry
                     consume=consume_input and kdim == N - 1 and e is None)
        if first:
            tr = t.gtrace
            bad = tr is None or bool(((~torch.isfinite(tr)) | (tr >= 2.0 ** 80) | ((tr > 0) & (tr <= 2.0 ** -80))).any().item())
            if bad:  # (readback: control flow only) redo with the norm pass and the scaled input
                nr = _hip.norm(X.reshape(Bt, -1))
                e2 = _range_guard_from_norms(nr)
                if e2 is None:  # (the trace overflowed by accumulation only: the input itself is in range -- no scaled copy)
                    return dense_tt_svd(X, eps, rmax, algorithm, batch, _guard_scaled=torch.zeros(Bt, dtype=torch.int32, device=X.device))
                return dense_tt_svd(_scale_batch(X, e2, -1), eps, rmax, algorithm, batch, _guard_scaled=e2)
        cores[kdim] = t.right.reshape(Bt, t.rank, shape[kdim], rn)
        C = t.left_scaled()
        rn = t.rank
    if e is not None:
        C = _scale_batch(C, e, +1)
    c0 = C.reshape(Bt, 1, shape[0], rn).contiguous()
    if consume_input and c0.untyped_storage().data_ptr() == X.untyped_storage().data_ptr():
        c0 = c0.clone()   # (N == 2: the in-place carry IS core 0 -- the result must not live in, nor pin, the consumed input)
    cores[0] = c0
    return cores  # type: ignore[return-value]


# ---------------------------------------------------------------------------------------------- consumers
def _sum_last(x3: torch.Tensor) -> torch.Tensor:
    """[B, r, c] -> [B, r, 1]: sum over the last axis as a GEMM with a ones vector."""
    ones = torch.ones((x3.shape[0], x3.shape[2], 1), dtype=x3.dtype, device=x3.device)
    return _hip.gemm(x3, ones)


def decompress(c: Sequence[torch.Tensor]) -> torch.Tensor:
    """tensor.py:1639-1687 for TT cores [B, r0, I, r1]: chain of unfolding GEMMs, left to right.
    Returns [B, I_1, ..., I_N] (boundary ranks > 1 are summed away, as the reference does)."""
    Bt = c[0].shape[0]
    acc = c[0].reshape(Bt, -1, c[0].shape[-1])
    for core in c[1:]:
        acc = _hip.gemm(acc, core.reshape(Bt, core.shape[1], -1)).reshape(Bt, -1, core.shape[-1])
    r0, rN = c[0].shape[1], c[-1].shape[-1]
    if rN > 1:
        acc = _sum_last(acc)
    if r0 > 1:
        acc = _sum_last(acc.reshape(Bt, r0, -1).transpose(1, 2).contiguous())
    return acc.reshape([Bt] + [core.shape[2] for core in c])


def dot(c1: Sequence[torch.Tensor], c2: Sequence[torch.Tensor]) -> torch.Tensor:
    """metrics.py:28-116 (k = N, TT cores only) for cores [1, r, I, r']: the left-to-right ``Lprod``
    contraction, two GEMMs per core; returns a 0-d tensor on the device."""
    a0, b0 = c1[0], c2[0]
    L = torch.ones((1, b0.shape[1], a0.shape[1]), dtype=a0.dtype, device=a0.device)
    for a, b in zip(c1, c2):
        _, r, I, r1 = a.shape
        _, s_, _, s1 = b.shape
        U = _hip.gemm(L, a.reshape(1, r, I * r1)).reshape(1, s_ * I, r1)       # einsum("sr,rai->sai")
        L = _hip.gemm(b.reshape(1, s_ * I, s1), U, transA=True)                # left_unf(b)^T left_unf(U)
    if L.numel() > 1:
        L = _sum_last(_sum_last(L).transpose(1, 2).contiguous())
    return L.reshape(())


def dense_dot(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """<a, b> of two dense device tensors: a 1 x n x 1 GEMM (split-K over the CUs)."""
    n = a.numel()
    return _hip.gemm(a.reshape(1, 1, n), b.reshape(1, n, 1)).reshape(())


def dense_norm(a: torch.Tensor) -> torch.Tensor:
    return _hip.norm(a.reshape(1, -1))[0]


def dense_dist(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """||a - b||: the difference through ttr_gemm_axpby (b as an n x 1 times 1 x 1 product), then ttr_norm."""
    n = a.numel()
    d = a.reshape(1, n, 1).clone()
    one = torch.ones((1, 1, 1), dtype=a.dtype, device=a.device)
    _hip.gemm_axpby(b.reshape(1, n, 1), one, d, -1.0, 1.0)
    return _hip.norm(d.reshape(1, -1))[0]


# ---------------------------------------------------------------------------------------------- CP-ALS
def _sum_all(x: torch.Tensor) -> torch.Tensor:
    """Sum of all entries as a 1 x n x 1 GEMM with a ones vector (split-K); returns a 0-d tensor."""
    n = x.numel()
    ones = torch.ones((1, n, 1), dtype=x.dtype, device=x.device)
    return _hip.gemm(x.reshape(1, 1, n), ones).reshape(())


def cp_hosvd_init(X: torch.Tensor, R: int) -> List[torch.Tensor]:
    """tensor.py:228-277: leading R eigenvectors of every mode Gram matrix X_(n) X_(n)^T (split-K MFMA GEMM on
    the dense unfolding; first / last mode without a copy, middle modes through one permuted copy)."""
    N = X.dim()
    cores = []
    for n in range(N):
        I = X.shape[n]
        if n == N - 1 and N > 1:
            A = X.reshape(1, -1, I)
            G = _hip.gemm(A, A, transA=True)
        else:
            A = (X if n == 0 else X.movedim(n, 0).contiguous()).reshape(1, I, -1)
            G = _hip.gemm(A, A, transB=True)
        c = None
        if EIGH_TOPK_ENABLED and 64 < I <= _hip.lib().ttr_eigsel_max_n() and R <= 64 and 4 * R <= I:
            # only the R leading eigenvectors are looked at (tensor.py:262): selected eigenpairs (tridiagonalisation, multisection,
            # twisted factorisations) instead of the full block-Jacobi decomposition -- C4: four 256 x 256 problems, 113 -> 79 ms of
            # init, of which 58 ms are the four 2.2e12-flop Gram matrices at ~150 TF (fp32 MFMA peak 157.3) and 14 the two permuted copies.  Vectors of a cluster the solver cannot separate (min |R_jj| of its orthonormalisation <= 0.5): the full solver.
            Gn, _ = _hip.pow2_normalize(G)
            Xk, _, rmin = _hip.eigh_topk(Gn, R)
            if float(rmin[0].item()) > 0.5:   # (readback: control flow only)
                c = Xk[0].contiguous()
        if c is None:
            V, _, _ = _eigh_any(G, _hip.EIG_RAW, False, 0.0, I, _hip.SOLVER_TRIDIAG)  # columns by decreasing eigenvalue
            c = V[0][:, :R].contiguous()
        if c.shape[1] < R:  # complete with random entries (tensor.py:262-277)
            c = torch.cat((c, torch.randn(I, R - c.shape[1], dtype=c.dtype, device=c.device)), dim=1)
        cores.append(c)
    return cores


def _cp_solve(prod: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
    """A = Y prod^+ for the symmetric PSD R x R Hadamard-of-Grams matrix (``torch.linalg.lstsq(prod, Y^T)``,
    tensor.py:339-341): eigen-decomposition prod = V diag(s^2) V^T, A = ((Y V) / s / s) V^T."""
    R = prod.shape[-1]
    # The Hadamard product of the Gram matrices is badly conditioned during the first sweeps after the HOSVD start (measured
    # 2.4e6 for R = 32 on 48^4): an fp32 eigen-decomposition resolves its small eigenvalues to ~15 %, and the ALS trajectory
    # then leaves the reference's (lstsq, tensor.py:339-341) by 1e-3 .. 4e-3 in the error after five sweeps -- LAPACK's
    # fp32 eigh does the same (measured on the CPU), so it is the method, not the kernel.  The R x R system and the
    # I x R right-hand side are tiny: fp32 inputs are solved in fp64 (conversion copies + the fp64 kernels).
    out_dt = prod.dtype
    if out_dt == torch.float32:
        prod, Y = prod.double(), Y.double()
    V, sg, _ = _eigh_any(prod.reshape(1, R, R), _hip.EIG_RAW, False, 0.0, R, _hip.SOLVER_TRIDIAG)
    Z = _hip.gemm(Y[None], V, colscale=sg, colscale_mode=_hip.SCALE_DIV)
    Z = _hip.scale_cols(Z, sg, _hip.SCALE_DIV)
    A = _hip.gemm(Z, V, transB=True)[0]
    return A if A.dtype == out_dt else A.to(out_dt)


class _CpState:
    """One dense device tensor in the middle of CP-ALS: factors, Gram matrices and one Gauss-Seidel sweep (see cp_als)."""

    def __init__(self, X: torch.Tensor, R: int, init=None):
        self.X, self.R, self.N = X, R, X.dim()
        if self.N < 2:
            raise NotImplementedError("tntorch_amd: CP-ALS needs at least 2 modes")
        self.shape = list(X.shape)
        self.A = list(init) if init is not None else cp_hosvd_init(X, R)
        self.grams = [None] + [_hip.gemm(self.A[n][None], self.A[n][None], transA=True)[0] for n in range(1, self.N)]
        self.xnorm = float(_hip.norm(X.reshape(1, -1))[0].item())

    def _had(self, skip):
        out = None
        for m in range(self.N - 1, -1, -1):
            if m != skip:
                out = self.grams[m] if out is None else _hip.hadamard(out, self.grams[m])
        return out

    def sweep(self) -> float:
        """Modes 0 .. N-1 once; returns the relative error ||X - T|| / ||X|| after the sweep."""
        X, A, grams, shape, N, R = self.X, self.A, self.grams, self.shape, self.N, self.R
        P_last = _hip.gemm(X.reshape(1, -1, shape[-1]), A[N - 1][None])[0]  # [I_0 * .. * I_{N-2}, R]
        for n in range(N - 1):
            T = P_last
            for m in range(N - 2, n, -1):  # trailing modes
                Pm = 1
                for d in shape[:m]:
                    Pm *= d
                T = _hip.krp_contract(T.reshape(Pm, shape[m], 1, R), A[m])
            for m in range(0, n):  # leading modes
                Qm = 1
                for d in shape[m + 1:n + 1]:
                    Qm *= d
                T = _hip.krp_contract(T.reshape(1, shape[m], Qm, R), A[m])
            A[n] = _cp_solve(self._had(n), T.reshape(shape[n], R))
            grams[n] = _hip.gemm(A[n][None], A[n][None], transA=True)[0]
        T = _hip.gemm(X.reshape(1, shape[0], -1), A[0][None], transA=True)[0]  # [I_1 * .. * I_{N-1}, R]
        for m in range(1, N - 1):
            Qm = 1
            for d in shape[m + 1:]:
                Qm *= d
            T = _hip.krp_contract(T.reshape(1, shape[m], Qm, R), A[m])
        Y = T.reshape(shape[N - 1], R)
        A[N - 1] = _cp_solve(self._had(N - 1), Y)
        grams[N - 1] = _hip.gemm(A[N - 1][None], A[N - 1][None], transA=True)[0]
        xt = float(dense_dot(Y, A[N - 1]).item())
        tt = float(_sum_all(_hip.hadamard(self._had(N - 1), grams[N - 1])).item())
        if self.xnorm == 0.0:
            return 0.0
        return math.sqrt(max(self.xnorm * self.xnorm - 2.0 * xt + tt, 0.0)) / self.xnorm


def cp_als(X: torch.Tensor, R: int, max_iter: int, tol: float, verbose: bool = False, batch: bool = False, init=None):
    """``tn.Tensor(X, ranks_cp=R)`` for a dense device tensor (tensor.py:210-400; HOSVD init, or the given ``init``
    factors for CP on a Tucker core, tensor.py:282-300).

    Same alternating sweep as the reference (mode 0 .. N-1, Gauss-Seidel), restructured so that the dense
    tensor is read TWICE per sweep instead of N times and nothing of size I^(N-1) x R or I^N is ever written:
      * modes 0..N-2 start from  P = X x_{N-1} A_{N-1}  (one ttr_gemm over X; A_{N-1} only changes at the end
        of the sweep) and fold the other factors in with ttr_krp_contract (trailing modes, then leading modes);
      * mode N-1 starts from  X_(0)^T A_0  (second ttr_gemm over X) and folds modes 1..N-2 in;
      * the R x R normal equations use the Hadamard product of the Gram matrices (ttr_hadamard) and the
        tridiagonal eigensolver; the relative error comes from ||X||^2 - 2<X,T> + ||T||^2 with <X,T> taken from
        the last MTTKRP (no dense reconstruction, tensor.py:373-379) -- resolved down to ~sqrt(eps).
    ``batch``: X is [B, I_1..I_N]; the items run in lock step (one sweep of every item per iteration) because the
    reference decides convergence ONCE on the batch-mean error (tensor.py:362-381); factors come back as [B, I_n, R].
    Returns (factors, errors)."""
    if batch:
        Bt = X.shape[0]
        states = [_CpState(X[i], R, None if init is None else [f[i] for f in init]) for i in range(Bt)]
    else:
        states = [_CpState(X, R, init)]
    errors: List[float] = []
    for it in range(max_iter):
        err = sum(st.sweep() for st in states) / len(states)
        errors.append(err)
        if verbose:
            print("iter: {} | eps: {:.8f}".format(it, err))
        if len(errors) >= 2 and errors[-2] - errors[-1] < tol:  # tensor.py:380-381
            break
    if batch:
        N = states[0].N
        return [torch.stack([st.A[n] for st in states]) for n in range(N)], errors
    return states[0].A, errors


# ---------------------------------------------------------------------------------------------- producers
def core_kron(a4: torch.Tensor, b4: torch.Tensor) -> torch.Tensor:
    """tensor.py:2309-2320 ``_core_kron`` on [B, r, I, r'] cores."""
    return _hip.core_kron(a4, b4)


def scale(x: torch.Tensor, value: float) -> torch.Tensor:
    """x * value for a device tensor of any shape (ttr_scale_batch with one broadcast scalar)."""
    return _hip.scale_batch(x.reshape(1, -1), scale=float(value)).reshape(x.shape)
