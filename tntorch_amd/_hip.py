"""ctypes binding of ``libttround_hip.so`` (the C ABI declared in ``include/ttround_hip.h``).

There is no fallback: a CUDA/HIP tensor reaching the hot path without the built
library raises ``RuntimeError``.  Torch is used here only for device memory
(``torch.empty``) and to obtain the current HIP stream.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_int64, c_void_p
from typing import Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# TTR_LIB_PATH: load another build of the same library (kernel experiments: several variants side by side)
LIB_PATH = os.environ.get("TTR_LIB_PATH") or os.path.join(_HERE, "libttround_hip.so")

F32, F64 = 0, 1
ABI_VERSION = 11  # include/ttround_hip.h: TTR_ABI_VERSION
SCALE_NONE, SCALE_MUL, SCALE_DIV = 0, 1, 2
EIG_RAW, EIG_REF, EIG_MATCH_DIAG = 0, 1, 2
SOLVER_JACOBI_REL, SOLVER_JACOBI_ABS, SOLVER_TRIDIAG, SOLVER_JACOBI_LIVE = 0, 1, 2, 3  # `abs_floor` argument of ttr_eigh_trunc
PROF_KINDS = ("gemm", "qr_factor", "qr_apply", "eigh", "misc", "rotgram", "project", "rowgram")

_lib = None

# name -> (restype, argtypes); mirrors include/ttround_hip.h one to one
_SIGNATURES = {
    "ttr_version": (c_int, []),
    "ttr_last_error": (c_char_p, []),
    "ttr_qr_max_cols": (c_int, [c_int]),
    "ttr_eigh_max_n_lds": (c_int, [c_int]),
    "ttr_eigh_max_n": (c_int, [c_int]),
    "ttr_gemm_workspace_bytes": (c_int64, [c_int, c_int64, c_int64, c_int64, c_int64]),
    "ttr_gemm": (
        c_int,
        [c_int, c_int, c_int, c_int64, c_int64, c_int64,
         c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
         c_void_p, c_int64, c_int, c_void_p, c_int64, c_int,
         c_int64, c_void_p, c_int64, c_void_p],
    ),
    "ttr_gemm_axpby": (
        c_int,
        [c_int, c_int, c_int, c_int64, c_int64, c_int64,
         c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
         c_double, c_double, c_int64, c_void_p, c_int64, c_void_p],
    ),
    "ttr_qr_workspace_bytes": (c_int64, [c_int, c_int64, c_int64, c_int64]),
    "ttr_qr": (
        c_int,
        [c_int, c_int64, c_int64, c_int64,
         c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
         c_void_p, c_int64, c_void_p],
    ),
    "ttr_qr_t": (
        c_int,
        [c_int, c_int64, c_int64, c_int64,
         c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
         c_void_p, c_int64, c_void_p],
    ),
    "ttr_qr_factor": (
        c_int,
        [c_int, c_int64, c_int64, c_int64,
         c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
         c_void_p, c_int64, c_void_p],
    ),
    "ttr_qr_factor_expo": (
        c_int,
        [c_int, c_int64, c_int64, c_int64,
         c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
         c_void_p, c_int64, c_void_p, c_void_p],
    ),
    "ttr_qr_apply": (
        c_int,
        [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64,
         c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p],
    ),
    "ttr_qr_pushed_workspace_bytes": (c_int64, [c_int, c_int64, c_int64, c_int64]),
    "ttr_qr_factor_pushed": (
        c_int,
        [c_int, c_int64, c_int64, c_int64, c_int64, c_int64,
         c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64,
         c_void_p, c_int64, c_void_p],
    ),
    "ttr_qr_factor_pushed_expo": (
        c_int,
        [c_int, c_int64, c_int64, c_int64, c_int64, c_int64,
         c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64,
         c_void_p, c_int64, c_void_p, c_void_p],
    ),
    "ttr_qr_factor_pushed_sum": (
        c_int,
        [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64,
         c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p],
    ),
    "ttr_qr_apply_pushed": (
        c_int,
        [c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64,
         c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p],
    ),
    "ttr_qr_apply_pushed_gram_parts": (c_int64, [c_int, c_int64, c_int64, c_int64, c_int64]),
    "ttr_qr_apply_pushed_gram": (
        c_int,
        [c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64,
         c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p],
    ),
    "ttr_eigh_workspace_bytes": (c_int64, [c_int, c_int64, c_int64]),
    "ttr_eigh_trunc": (
        c_int,
        [c_int, c_int64, c_int64,
         c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p,
         c_int, c_int, c_double, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p],
    ),
    "ttr_carry_rows32": (c_int, [c_int, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "ttr_spectrum_flat": (c_int, [c_int, c_int64, c_int64, c_void_p, c_int64, c_int64, c_double, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ttr_eigh_top_ok": (c_int, [c_int64, c_int64]),
    "ttr_eigh_top": (
        c_int,
        [c_int, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64,
         c_void_p, c_int64, c_double, c_void_p, c_int, c_void_p],
    ),
    "ttr_eigsel_max_n": (c_int, []),
    "ttr_eigsel_scratch_bytes": (c_int64, [c_int, c_int64, c_int64]),
    "ttr_tridiag_workspace_bytes": (c_int64, [c_int, c_int64, c_int64]),
    "ttr_tridiag": (c_int, [c_int, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "ttr_tri_eigsel": (c_int, [c_int, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "ttr_tridiag_back": (c_int, [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "ttr_bj_scratch_bytes": (c_int64, [c_int, c_int64, c_int64, c_int64]),
    "ttr_bj_solve": (c_int, [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ttr_bj_apply": (
        c_int,
        [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "ttr_bj_control": (c_int, [c_int, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_double, c_void_p]),
    "ttr_sweep_gram_parts": (c_int64, [c_int64, c_int64]),
    "ttr_rowgram": (c_int, [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "ttr_qr_pushed_flag_offset": (c_int64, [c_int, c_int64, c_int64, c_int64]),
    "ttr_rotgram": (
        c_int,
        [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p],
    ),
    "ttr_project": (
        c_int,
        [c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
         c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p],
    ),
    "ttr_colgram_workspace_bytes": (c_int64, [c_int, c_int64, c_int64, c_int64]),
    "ttr_colgram": (
        c_int,
        [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p],
    ),
    "ttr_colproject": (
        c_int,
        [c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
         c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p],
    ),
    "ttr_pow2_normalize": (c_int, [c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "ttr_scale_batch": (
        c_int,
        [c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64, c_void_p],
    ),
    "ttr_orth_fixup_workspace_bytes": (c_int64, [c_int, c_int64, c_int64, c_int64, c_int64]),
    "ttr_orth_fixup": (
        c_int,
        [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_double, c_void_p,
         c_void_p, c_int64, c_void_p],
    ),
    "ttr_norm": (c_int, [c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "ttr_scale_cols": (
        c_int,
        [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int,
         c_void_p, c_int64, c_int64, c_void_p],
    ),
    "ttr_mask_cols": (c_int, [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "ttr_krp_contract": (c_int, [c_int, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "ttr_hadamard": (c_int, [c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ttr_core_kron": (c_int, [c_int] + [c_int64] * 6 + [c_void_p, c_void_p, c_void_p, c_void_p]),
    "ttr_round_tt_workspace_bytes": (c_int64, [c_int, c_int64, c_void_p, c_void_p, c_int64, c_int]),
    "ttr_round_tt": (
        c_int,
        [c_int, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_double, c_double, c_int, c_void_p,
         c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    ),
    "ttr_debug_set_qr_stamps": (c_int, [c_void_p]),
    "ttr_debug_set_knob": (c_int, [c_int, c_int]),
    "ttr_prof_enable": (c_int, [c_int]),
    "ttr_prof_collect": (c_int, [ctypes.POINTER(c_double), ctypes.POINTER(c_int64)]),
    "ttr_prof_collect_work": (c_int, [ctypes.POINTER(c_double), ctypes.POINTER(c_double)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def available() -> bool:
    return os.path.exists(LIB_PATH)


def lib():
    """Load the shared library (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP kernels are not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C tntorch_amd/csrc`). "
                "tntorch_amd has no CPU/torch fallback for GPU tensors."
            )
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.ttr_version() != ABI_VERSION:
            raise RuntimeError(
                f"{LIB_PATH} was built for ABI version {L.ttr_version()}, this binding expects {ABI_VERSION} "
                "(include/ttround_hip.h: TTR_ABI_VERSION): rebuild with `python __graft_entry__.py --force`.")
        # A/B measurements without code changes: TTR_KNOBS="6=0,0=1" calls ttr_debug_set_knob(6, 0), (0, 1) once after loading
        for item in filter(None, os.environ.get("TTR_KNOBS", "").split(",")):
            k, v = item.split("=")
            if L.ttr_debug_set_knob(int(k), int(v)) != 0:
                raise ValueError(f"TTR_KNOBS: ttr_debug_set_knob({k}, {v}) rejected: " + L.ttr_last_error().decode(errors="replace"))
        # The eps-mode rank rule sees null directions at LAPACK's noise floor (header: TTR_KNOB_RANK_NOISE_FLOOR = 1, the library's
        # default since round 6: the reference's ranks).  TTR_STRICT_RANKS=0 switches the floor off: exact zeros are cut
        # (INTEGRATION.md "Ranks of rank-deficient trains")
        if os.environ.get("TTR_STRICT_RANKS", "1") == "0":
            L.ttr_debug_set_knob(9, 0)
        # TTR_ORTH_SPLIT=<batch size>: ttr_orth_fixup's three-launch rounds from that batch size on (header: TTR_KNOB_ORTH_SPLIT)
        if os.environ.get("TTR_ORTH_SPLIT", "") != "":
            L.ttr_debug_set_knob(15, int(os.environ["TTR_ORTH_SPLIT"]))
        _lib = L
    return _lib


def _check(code: int, what: str):
    if code != 0:
        msg = lib().ttr_last_error().decode(errors="replace")
        if code == -2:
            raise NotImplementedError(f"{what}: {msg}")
        if code == -1:
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what} failed ({code}): {msg}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.float64:
        return F64
    raise TypeError(f"tntorch_amd HIP path supports float32/float64 tensors only, got {dt}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _first_cuda_tensor(args, kwargs):
    for a in args:
        if isinstance(a, torch.Tensor) and a.is_cuda:
            return a
        if isinstance(a, QrFactors):
            return a.ws
    for a in kwargs.values():
        if isinstance(a, torch.Tensor) and a.is_cuda:
            return a
    return None


def _on_device(fn):
    """Run a binding with the device of its first device operand current: workspaces / results are allocated there
    and ``_stream()`` is that device's current stream (a tensor on cuda:1 while cuda:0 is current would otherwise be
    processed on cuda:0's stream -- a cross-device launch, unordered against torch's work on cuda:1)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        t = _first_cuda_tensor(args, kwargs)
        if t is None or t.device.index is None or t.device.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(t.device):
            return fn(*args, **kwargs)

    return wrapped


def _mat(t: torch.Tensor) -> Tuple[torch.Tensor, int, int]:
    """Return (tensor, ld, batch_stride) of a [B, r, c] tensor whose rows are contiguous."""
    assert t.dim() == 3 and t.is_cuda
    B, r, c = t.shape
    ok = (c == 1 or t.stride(2) == 1) and (r == 1 or t.stride(1) >= max(c, 1)) and (B == 1 or t.stride(0) >= 0)
    if not ok:
        t = t.contiguous()
    ld = t.stride(1) if r > 1 else max(c, 1)
    bs = t.stride(0) if B > 1 else r * ld
    return t, int(ld), int(bs)


def max_qr_cols(dt: torch.dtype) -> int:
    return lib().ttr_qr_max_cols(dtype_code(dt))


def max_eigh_n(dt: torch.dtype = torch.float32) -> int:
    return lib().ttr_eigh_max_n(dtype_code(dt))


# ----------------------------------------------------------------------------------------------
@_on_device
def gemm(
    A: torch.Tensor,
    B: torch.Tensor,
    transA: bool = False,
    transB: bool = False,
    rowscale: Optional[torch.Tensor] = None,
    rowscale_mode: int = SCALE_NONE,
    colscale: Optional[torch.Tensor] = None,
    colscale_mode: int = SCALE_NONE,
    out: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """C[b] = scale(op(A[b]) @ op(B[b])) for [batch, ., .] tensors; returns a fresh tensor (or ``out``, a
    contiguous [batch, M, N] tensor -- e.g. a batch slice of a larger result -- that is overwritten)."""
    L = lib()
    dt = dtype_code(A.dtype)
    assert A.dtype == B.dtype and A.shape[0] == B.shape[0]
    A, lda, sA = _mat(A)
    B, ldb, sB = _mat(B)
    batch = A.shape[0]
    M, K = (A.shape[2], A.shape[1]) if transA else (A.shape[1], A.shape[2])
    K2, N = (B.shape[2], B.shape[1]) if transB else (B.shape[1], B.shape[2])
    if K != K2:
        raise ValueError(f"gemm: inner dimensions differ ({K} vs {K2})")
    if out is not None:
        assert tuple(out.shape) == (batch, M, N) and out.is_contiguous() and out.dtype == A.dtype
        C = out
    else:
        C = torch.empty((batch, M, N), dtype=A.dtype, device=A.device)
    if M == 0 or N == 0 or batch == 0:
        return C
    rs_ptr, rs_stride = None, 0
    if rowscale is not None:
        rowscale = rowscale.contiguous()
        rs_ptr, rs_stride = rowscale.data_ptr(), rowscale.shape[-1]
    cs_ptr, cs_stride = None, 0
    if colscale is not None:
        colscale = colscale.contiguous()
        cs_ptr, cs_stride = colscale.data_ptr(), colscale.shape[-1]
    wsb = L.ttr_gemm_workspace_bytes(dt, M, N, K, batch)
    ws = torch.empty(wsb, dtype=torch.uint8, device=A.device) if wsb > 0 else None
    code = L.ttr_gemm(
        dt, int(transA), int(transB), M, N, K,
        A.data_ptr(), lda, sA, B.data_ptr(), ldb, sB, C.data_ptr(), N, M * N,
        rs_ptr, rs_stride, rowscale_mode if rowscale is not None else SCALE_NONE,
        cs_ptr, cs_stride, colscale_mode if colscale is not None else SCALE_NONE,
        batch, ws.data_ptr() if ws is not None else None, wsb, _stream(),
    )
    _check(code, "ttr_gemm")
    return C


@_on_device
def gemm_axpby(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, alpha: float, beta: float,
               transA: bool = False, transB: bool = False) -> torch.Tensor:
    """In place: C[b] <- beta * C[b] + alpha * op(A[b]) @ op(B[b]); C must be contiguous [batch, M, N]."""
    L = lib()
    dt = dtype_code(A.dtype)
    assert A.dtype == B.dtype == C.dtype and A.shape[0] == B.shape[0] == C.shape[0]
    assert C.is_contiguous()
    A, lda, sA = _mat(A)
    B, ldb, sB = _mat(B)
    batch = A.shape[0]
    M, K = (A.shape[2], A.shape[1]) if transA else (A.shape[1], A.shape[2])
    K2, N = (B.shape[2], B.shape[1]) if transB else (B.shape[1], B.shape[2])
    if K != K2 or tuple(C.shape) != (batch, M, N):
        raise ValueError(f"gemm_axpby: shapes do not match ({K} vs {K2}, C {tuple(C.shape)})")
    if M == 0 or N == 0 or batch == 0:
        return C
    wsb = L.ttr_gemm_workspace_bytes(dt, M, N, K, batch)
    ws = torch.empty(wsb, dtype=torch.uint8, device=A.device) if wsb > 0 else None
    code = L.ttr_gemm_axpby(
        dt, int(transA), int(transB), M, N, K,
        A.data_ptr(), lda, sA, B.data_ptr(), ldb, sB, C.data_ptr(), N, M * N,
        float(alpha), float(beta), batch, ws.data_ptr() if ws is not None else None, wsb, _stream(),
    )
    _check(code, "ttr_gemm_axpby")
    return C


@_on_device
def qr(A: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Reduced Householder QR of [batch, m, n] -> Q [batch, m, k], R [batch, k, n]."""
    L = lib()
    dt = dtype_code(A.dtype)
    A, lda, sA = _mat(A)
    batch, m, n = A.shape
    k = min(m, n)
    Q = torch.empty((batch, m, k), dtype=A.dtype, device=A.device)
    R = torch.empty((batch, k, n), dtype=A.dtype, device=A.device)
    if batch == 0:
        return Q, R
    wsb = L.ttr_qr_workspace_bytes(dt, m, n, batch)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=A.device)
    code = L.ttr_qr(dt, m, n, batch, A.data_ptr(), lda, sA, Q.data_ptr(), k, m * k, R.data_ptr(), n, k * n,
                    ws.data_ptr(), wsb, _stream())
    _check(code, "ttr_qr")
    return Q, R


@_on_device
def qr_t(At: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """QR of the TRANSPOSE of ``At`` [batch, n, m] without transposing it: returns (Qt [batch, k, m] = Q^T, R [batch, k, n])
    (ttr_qr_t: the kernels address the operand and the result through strides)."""
    L = lib()
    dt = dtype_code(At.dtype)
    At, ldat, sAt = _mat(At)
    batch, n, m = At.shape
    k = min(m, n)
    Qt = torch.empty((batch, k, m), dtype=At.dtype, device=At.device)
    R = torch.empty((batch, k, n), dtype=At.dtype, device=At.device)
    if batch == 0:
        return Qt, R
    wsb = L.ttr_qr_workspace_bytes(dt, m, n, batch)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=At.device)
    _check(L.ttr_qr_t(dt, m, n, batch, At.data_ptr(), ldat, sAt, Qt.data_ptr(), m, k * m, R.data_ptr(), n, k * n,
                      ws.data_ptr(), wsb, _stream()), "ttr_qr_t")
    return Qt, R


class QrFactors:
    """Handle on a factored batch (reflectors + T factors live in ``ws``) for later ``qr_apply`` calls."""

    __slots__ = ("ws", "wsb", "m", "n", "batch", "dtype", "R", "pushed", "rows32")

    def __init__(self, ws, wsb, m, n, batch, dtype, R, pushed=None, rows32=None):
        self.ws, self.wsb, self.m, self.n, self.batch, self.dtype, self.R = ws, wsb, m, n, batch, dtype, R
        self.pushed = pushed  # (k, I) when the factorisation came from qr_factor_pushed
        # int32 [batch] view into ``ws`` (pushed factorisations): != 0 for items whose level-0 blocks packed their rows -- rows
        # kk >= 32 of what qr_apply produces from this handle are exactly zero (ttr_rowgram / ttr_project: ``rows32``)
        self.rows32 = rows32

    @property
    def k(self):
        return min(self.m, self.n)


@_on_device
def qr_factor(A: torch.Tensor, expo_acc: Optional[torch.Tensor] = None) -> QrFactors:
    """Factor [batch, m, n]; returns a handle holding R [batch, k, n] and the implicit Q.

    ``expo_acc`` (fp32; device int32 [batch]): R comes back as R 2^-e, e the exponent of the top block's largest entry, and e is
    added to expo_acc[b] -- the sweep's per-core power-of-two normalisation without a launch of its own (ttr_qr_factor_expo)."""
    L = lib()
    dt = dtype_code(A.dtype)
    A, lda, sA = _mat(A)
    batch, m, n = A.shape
    k = min(m, n)
    R = torch.empty((batch, k, n), dtype=A.dtype, device=A.device)
    wsb = L.ttr_qr_workspace_bytes(dt, m, n, max(batch, 1))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=A.device)
    if batch > 0 and expo_acc is not None:
        assert expo_acc.dtype == torch.int32 and expo_acc.numel() == batch and expo_acc.is_contiguous()
        code = L.ttr_qr_factor_expo(dt, m, n, batch, A.data_ptr(), lda, sA, R.data_ptr(), n, k * n, ws.data_ptr(), wsb,
                                    expo_acc.data_ptr(), _stream())
        _check(code, "ttr_qr_factor_expo")
    elif batch > 0:
        code = L.ttr_qr_factor(dt, m, n, batch, A.data_ptr(), lda, sA, R.data_ptr(), n, k * n, ws.data_ptr(), wsb, _stream())
        _check(code, "ttr_qr_factor")
    return QrFactors(ws, wsb, m, n, batch, A.dtype, R)


def pushed_supported(k: int, Rin: int, I: int, n: int, dt: torch.dtype) -> bool:
    return k <= 64 and Rin <= 64 and n <= max_qr_cols(dt) and k * I >= n


@_on_device
def qr_factor_pushed(Rm: torch.Tensor, core4: torch.Tensor, expo_acc: Optional[torch.Tensor] = None) -> QrFactors:
    """Factor the left unfolding of ``Rm @ core`` (Rm [batch, k, Rin], core [batch, Rin, I, n]) without forming it.
    ``expo_acc``: as for ``qr_factor`` (ttr_qr_factor_pushed_expo)."""
    L = lib()
    dt = dtype_code(core4.dtype)
    Rm, ldrm, sRm = _mat(Rm)
    core4 = core4.contiguous()
    batch, Rin, I, n = core4.shape
    k = Rm.shape[1]
    assert Rm.shape[2] == Rin and Rm.shape[0] == batch
    kq = min(k * I, n)
    R = torch.empty((batch, kq, n), dtype=core4.dtype, device=core4.device)
    wsb = L.ttr_qr_pushed_workspace_bytes(dt, I, n, max(batch, 1))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=core4.device)
    if batch > 0 and expo_acc is not None:
        assert expo_acc.dtype == torch.int32 and expo_acc.numel() == batch and expo_acc.is_contiguous()
        code = L.ttr_qr_factor_pushed_expo(dt, k, Rin, I, n, batch, Rm.data_ptr(), ldrm, sRm, core4.data_ptr(), Rin * I * n,
                                           R.data_ptr(), n, kq * n, ws.data_ptr(), wsb, expo_acc.data_ptr(), _stream())
        _check(code, "ttr_qr_factor_pushed_expo")
    elif batch > 0:
        code = L.ttr_qr_factor_pushed(dt, k, Rin, I, n, batch, Rm.data_ptr(), ldrm, sRm, core4.data_ptr(), Rin * I * n,
                                      R.data_ptr(), n, kq * n, ws.data_ptr(), wsb, _stream())
        _check(code, "ttr_qr_factor_pushed")
    flags = None
    if batch > 0 and k == 64:
        off = int(L.ttr_qr_pushed_flag_offset(dt, I, n, batch))
        if off >= 0:
            flags = ws[off:off + 4 * batch].view(torch.int32)
    return QrFactors(ws, wsb, k * I, n, batch, core4.dtype, R, pushed=(k, I), rows32=flags)


@_on_device
def qr_factor_pushed_sum(Rm: torch.Tensor, a4: torch.Tensor, b4: torch.Tensor) -> QrFactors:
    """Factor the left unfolding of ``Rm @ blockdiag(a, b)`` (Rm [batch, k, ra + rb], a [batch, ra, I, ca],
    b [batch, rb, I, cb]) without forming the block-diagonal core of the TT sum (ttr_qr_factor_pushed_sum)."""
    L = lib()
    dt = dtype_code(a4.dtype)
    Rm, ldrm, sRm = _mat(Rm)
    a4, b4 = a4.contiguous(), b4.contiguous()
    batch, ra, I, ca = a4.shape
    _, rb, _, cb = b4.shape
    k, n = Rm.shape[1], ca + cb
    assert Rm.shape[2] == ra + rb and Rm.shape[0] == batch and b4.shape[0] == batch and b4.shape[2] == I
    kq = min(k * I, n)
    R = torch.empty((batch, kq, n), dtype=a4.dtype, device=a4.device)
    wsb = L.ttr_qr_pushed_workspace_bytes(dt, I, n, max(batch, 1))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=a4.device)
    if batch > 0:
        code = L.ttr_qr_factor_pushed_sum(dt, k, I, batch, Rm.data_ptr(), ldrm, sRm, a4.data_ptr(), ra, ca, ra * I * ca,
                                          b4.data_ptr(), rb, cb, rb * I * cb, R.data_ptr(), n, kq * n, ws.data_ptr(), wsb,
                                          _stream())
        _check(code, "ttr_qr_factor_pushed_sum")
    return QrFactors(ws, wsb, k * I, n, batch, a4.dtype, R, pushed=(k, I))


@_on_device
def qr_apply(f: QrFactors, C: Optional[torch.Tensor] = None, kcols: Optional[int] = None,
             out: Optional[torch.Tensor] = None, want_gram: bool = False, skip_zero_rows: bool = False):
    """Out [batch, m, kcols] = Q @ C  (C: [batch, k, kcols]; None -> first ``kcols`` columns of Q).
    ``out``: optional contiguous destination (e.g. a batch slice of a larger result).
    ``want_gram``: return ``(Out, G)``; G = split partials [batch, parts, k, k] of the row Gram matrix of Out's
    k x (I kcols) right unfolding, accumulated by the apply kernel itself (ttr_qr_apply_pushed_gram), or None when
    the shape is not covered (the caller then runs ``rowgram``)."""
    L = lib()
    dt = dtype_code(f.dtype)
    if C is not None:
        C, ldc, sC = _mat(C)
        assert C.shape[0] == f.batch and C.shape[1] == f.k
        kcols = C.shape[2]
        cptr = C.data_ptr()
    else:
        kcols = f.k if kcols is None else kcols
        ldc, sC, cptr = 0, 0, None
    if out is not None:
        assert tuple(out.shape) == (f.batch, f.m, kcols) and out.is_contiguous() and out.dtype == f.dtype
        Out = out
    else:
        Out = torch.empty((f.batch, f.m, kcols), dtype=f.dtype, device=f.ws.device)
    G = None
    if f.batch == 0:
        return (Out, G) if want_gram else Out
    if f.pushed is not None:
        k, I = f.pushed
        parts = int(L.ttr_qr_apply_pushed_gram_parts(dt, k, I, f.n, kcols)) if want_gram else 0
        if parts > 0:
            G = torch.empty((f.batch, parts, k, k), dtype=f.dtype, device=f.ws.device)
            code = L.ttr_qr_apply_pushed_gram(dt, k, I, f.n, f.batch, f.ws.data_ptr(), f.wsb, cptr, ldc, sC, kcols,
                                              Out.data_ptr(), kcols, f.m * kcols, G.data_ptr(), _stream())
            _check(code, "ttr_qr_apply_pushed_gram")
        else:
            # skip_zero_rows: the rows kk >= 32 of packed items (f.rows32) stay unwritten -- only for results that are read
            # through the rows32-aware kernels (rowgram / rotgram / project)
            code = L.ttr_qr_apply_pushed(dt, k, I, f.n, f.batch, f.ws.data_ptr(), f.wsb, cptr, ldc, sC, kcols,
                                         Out.data_ptr(), kcols, f.m * kcols, int(bool(skip_zero_rows and f.rows32 is not None)),
                                         _stream())
            _check(code, "ttr_qr_apply_pushed")
        return (Out, G) if want_gram else Out
    code = L.ttr_qr_apply(dt, f.m, f.n, f.batch, f.ws.data_ptr(), f.wsb, cptr, ldc, sC, kcols,
                          Out.data_ptr(), kcols, f.m * kcols, _stream())
    _check(code, "ttr_qr_apply")
    return (Out, G) if want_gram else Out


@_on_device
def eigh_trunc(
    G: torch.Tensor, eig_mode: int, use_delta: bool, delta2: float, rmax: int, abs_floor: int = 1,
    sweeps: Optional[torch.Tensor] = None, delta2_dev: Optional[torch.Tensor] = None,
    skip_items: Optional[torch.Tensor] = None, sigma_in: Optional[torch.Tensor] = None,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Eigen-decomposition of symmetric [batch, n, n] + rank rule.  Returns V (columns sorted by
    decreasing sigma), sigma [batch, n], info [batch] int32 (rank, or 0 for the zero guard).
    ``G`` may also be [batch, parts, n, n] (contiguous): split-K partials of a Gram kernel, summed on load."""
    L = lib()
    dt = dtype_code(G.dtype)
    gparts, sGp = 1, 0
    if G.dim() == 4:
        G = G.contiguous()
        batch, gparts, n, _ = G.shape
        ldg, sG, sGp = n, gparts * n * n, n * n
    else:
        G, ldg, sG = _mat(G)
        batch, n, _ = G.shape
    V = torch.empty((batch, n, n), dtype=G.dtype, device=G.device)
    sigma = torch.empty((batch, n), dtype=G.dtype, device=G.device)
    info = torch.empty((batch,), dtype=torch.int32, device=G.device)
    if batch == 0:
        return V, sigma, info
    wsb = L.ttr_eigh_workspace_bytes(dt, n, batch)
    ws = torch.empty(wsb, dtype=torch.uint8, device=G.device) if wsb > 0 else None
    rmax = int(min(max(int(rmax), 1), 2**31 - 1))
    code = L.ttr_eigh_trunc(
        dt, n, batch, G.data_ptr(), ldg, sG, gparts, sGp, V.data_ptr(), n, n * n, sigma.data_ptr(), n, info.data_ptr(),
        eig_mode, int(bool(use_delta)), float(delta2),
        delta2_dev.data_ptr() if delta2_dev is not None else None, rmax, int(abs_floor),
        sweeps.data_ptr() if sweeps is not None else None,
        skip_items.data_ptr() if skip_items is not None else None,
        sigma_in.data_ptr() if sigma_in is not None else None, sigma_in.shape[-1] if sigma_in is not None else 0,
        ws.data_ptr() if ws is not None else None, wsb, _stream(),
    )
    _check(code, "ttr_eigh_trunc")
    return V, sigma, info


def eigh_top_ok(n: int, r: int) -> bool:
    return bool(lib().ttr_eigh_top_ok(int(n), int(r)))


@_on_device
def eigh_top(G: torch.Tensor, r: int, thr: float, need_all: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Pass 1 of a batch-mode bond: symmetric [batch, n, n] (or split partials [batch, parts, n, n]), 40 <= n <= 64, rank cap
    r <= 32 (ttr_eigh_top).  Returns (V, sigma, info, flat): items with flat[b] = 1 carry their r largest eigenpairs (V[b][:, :r],
    sigma[b][:r], zeros beyond), the others the full decomposition of ``eigh_trunc(G, EIG_RAW, ..., abs_floor=SOLVER_TRIDIAG)``
    (flat[b] = 2 when its kept sigma pass ``spectrum_flat``'s batch-mode test, else 0).  ``need_all``: eps mode -- the top-r path only
    takes items whose every eigenpair it computes (zero-tail items under a cap >= 32); sigma / V are complete for every item."""
    L = lib()
    dt = dtype_code(G.dtype)
    gparts, sGp = 1, 0
    if G.dim() == 4:
        G = G.contiguous()
        batch, gparts, n, _ = G.shape
        ldg, sG, sGp = n, gparts * n * n, n * n
    else:
        G, ldg, sG = _mat(G)
        batch, n, _ = G.shape
    V = torch.empty((batch, n, n), dtype=G.dtype, device=G.device)
    sigma = torch.empty((batch, n), dtype=G.dtype, device=G.device)
    info = torch.empty((batch,), dtype=torch.int32, device=G.device)
    flat = torch.empty((batch,), dtype=torch.int32, device=G.device)
    if batch:
        _check(L.ttr_eigh_top(dt, n, batch, G.data_ptr(), ldg, sG, gparts, sGp, V.data_ptr(), n, n * n, sigma.data_ptr(), n,
                              info.data_ptr(), int(r), float(thr), flat.data_ptr(), int(bool(need_all)), _stream()), "ttr_eigh_top")
    return V, sigma, info, flat


@_on_device
def eigh_topk(G: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """The k largest eigenpairs of symmetric [batch, n, n] (64 < n <= ttr_eigsel_max_n(), k <= 64): tridiagonalisation,
    multisection + twisted factorisation on the tridiagonal matrix, TSQR of the k vectors, back-transformation (ttr_tridiag,
    ttr_tri_eigsel, ttr_qr, ttr_tridiag_back).  Returns (X [batch, n, k] orthonormal, lam [batch, k] descending,
    rmin [batch] = the smallest |R_jj| of the orthonormalisation: well below 1 when two of the k vectors nearly coincided --
    clustered or multiple eigenvalues, which this solver does not resolve; the caller decides)."""
    L = lib()
    dt = dtype_code(G.dtype)
    batch, n, _ = G.shape
    A = G.contiguous().clone()  # destroyed by the reduction
    d = torch.empty((batch, n), dtype=G.dtype, device=G.device)
    e = torch.empty_like(d)
    tau = torch.empty_like(d)
    lam = torch.empty((batch, k), dtype=G.dtype, device=G.device)
    Z = torch.empty((batch, n, k), dtype=G.dtype, device=G.device)
    if batch == 0:
        return Z, lam, torch.empty((0,), dtype=G.dtype, device=G.device)
    wsb = L.ttr_tridiag_workspace_bytes(dt, n, batch)
    ws = torch.empty(wsb, dtype=torch.uint8, device=G.device)
    _check(L.ttr_tridiag(dt, n, batch, A.data_ptr(), n, n * n, d.data_ptr(), e.data_ptr(), tau.data_ptr(), ws.data_ptr(), wsb, _stream()),
           "ttr_tridiag")
    sb = L.ttr_eigsel_scratch_bytes(dt, n, batch)
    scratch = torch.empty(sb, dtype=torch.uint8, device=G.device)
    _check(L.ttr_tri_eigsel(dt, n, batch, int(k), d.data_ptr(), e.data_ptr(), lam.data_ptr(), Z.data_ptr(), scratch.data_ptr(), sb,
                            _stream()), "ttr_tri_eigsel")
    Zq, R = qr(Z)
    rmin = torch.diagonal(R, dim1=1, dim2=2).abs().amin(dim=1)
    Zq = Zq.contiguous()
    _check(L.ttr_tridiag_back(dt, n, batch, int(k), A.data_ptr(), n, n * n, tau.data_ptr(), Zq.data_ptr(), _stream()), "ttr_tridiag_back")
    return Zq, lam, rmin


_BJ_TABLES: dict = {}


def bj_pair_tables(nbk: int, device) -> torch.Tensor:
    """Round-robin tournament over an even number of blocks: int32 [nbk - 1, nbk / 2, 2] on the device (cached)."""
    key = (nbk, str(device))
    if key not in _BJ_TABLES:
        assert nbk % 2 == 0 and nbk >= 2
        circle = list(range(nbk))
        rounds = []
        for _ in range(nbk - 1):
            rounds.append([[circle[i], circle[-1 - i]] for i in range(nbk // 2)])
            circle = [circle[0], circle[-1]] + circle[1:-1]
        _BJ_TABLES[key] = torch.tensor(rounds, dtype=torch.int32).to(device)
    return _BJ_TABLES[key]


@_on_device
def bj_sweeps(G: torch.Tensor, V: torch.Tensor, b: int, relative: bool, tol: float, max_sweeps: int) -> torch.Tensor:
    """Block-Jacobi sweeps on G [B, n, n] / V [B, n, n] IN PLACE (ttr_bj_solve / ttr_bj_apply / ttr_bj_control).  The
    convergence word is read back once per tranche of 8 sweeps (control flow only: the launches of a converged driver return at
    their first instruction, but each still costs a launch).  Returns the device control block (int32 [4]: converged, -, sweeps
    performed, -) for diagnostics."""
    L = lib()
    dt = dtype_code(G.dtype)
    Bt, n, _ = G.shape
    nbk = n // b
    assert n == nbk * b and nbk % 2 == 0 and G.is_contiguous() and V.is_contiguous()
    npairs, w = nbk // 2, 2 * b
    tabs = bj_pair_tables(nbk, G.device)
    ctrl = torch.zeros(4, dtype=torch.int32, device=G.device)
    state = torch.zeros(Bt + 1, dtype=torch.float64, device=G.device)
    state[Bt:].fill_(-1.0)
    gn = norm(G.reshape(Bt, -1)) if not relative else None
    W = torch.empty((Bt * npairs, w, w), dtype=G.dtype, device=G.device)
    scratch = torch.empty(max(int(L.ttr_bj_scratch_bytes(dt, b, npairs, Bt)), 16), dtype=torch.uint8, device=G.device)
    st = _stream()
    rounds = nbk - 1
    tranche = 8   # sweeps enqueued before the convergence word is looked at (one readback: control flow only).  Convergence
                  # typically comes after 6 .. 10 sweeps; the launches of the remaining sweeps of max_sweeps would return at their
                  # first instruction, but each still costs a launch (n = 1024, b = 32: 63 launches per sweep)
    for sweep in range(max_sweeps):
        if sweep > 0 and sweep % tranche == 0 and int(ctrl[0].item()) != 0:
            break
        for r in range(rounds):
            tab = tabs[r].data_ptr()
            _check(L.ttr_bj_solve(dt, b, npairs, Bt, G.data_ptr(), n, n * n, tab, W.data_ptr(), scratch.data_ptr(),
                                  ctrl.data_ptr(), st), "ttr_bj_solve")
            off = state.data_ptr() if (not relative and r == rounds - 1) else None
            _check(L.ttr_bj_apply(dt, b, npairs, Bt, G.data_ptr(), n, n * n, V.data_ptr(), n, n * n, tab, W.data_ptr(),
                                  ctrl.data_ptr(), off, st), "ttr_bj_apply")
        _check(L.ttr_bj_control(dt, Bt, ctrl.data_ptr(), state.data_ptr(), gn.data_ptr() if gn is not None else None,
                                int(relative), float(tol), st), "ttr_bj_control")
    return ctrl


@_on_device
def norm(x: torch.Tensor) -> torch.Tensor:
    """Frobenius norm per batch item of a [batch, ...] tensor -> [batch]."""
    L = lib()
    dt = dtype_code(x.dtype)
    x = x.contiguous()
    batch = x.shape[0]
    count = x[0].numel() if batch > 0 else 0
    out = torch.empty((batch,), dtype=x.dtype, device=x.device)
    if batch == 0:
        return out
    if count >= (1 << 20) and batch < 512:
        # ttr_norm runs one workgroup per batch item: split long items into k chunks (norm of the chunk
        # norms is exact: the second stage squares and sums in double) so that the whole chip streams
        k = 4096
        while k > 1 and count % k:
            k //= 2
        if k > 1:
            part = norm(x.reshape(batch * k, count // k))
            return norm(part.reshape(batch, k))
    _check(L.ttr_norm(dt, count, batch, x.data_ptr(), count, out.data_ptr(), _stream()), "ttr_norm")
    return out


@_on_device
def scale_cols(X: torch.Tensor, s: torch.Tensor, mode: int) -> torch.Tensor:
    """out[b, i, j] = X[b, i, j] * s[b, j] (SCALE_MUL) or / s[b, j] (SCALE_DIV)."""
    L = lib()
    dt = dtype_code(X.dtype)
    X, ldx, sX = _mat(X)
    s = s.contiguous()
    batch, rows, cols = X.shape
    out = torch.empty((batch, rows, cols), dtype=X.dtype, device=X.device)
    if out.numel() == 0:
        return out
    code = L.ttr_scale_cols(dt, rows, cols, batch, X.data_ptr(), ldx, sX, s.data_ptr(), s.shape[-1], mode,
                            out.data_ptr(), cols, rows * cols, _stream())
    _check(code, "ttr_scale_cols")
    return out


@_on_device
def mask_cols(X: torch.Tensor, keep: torch.Tensor) -> torch.Tensor:
    """In place: X[b, :, j] = 0 for j >= keep[b] (keep: int32 [batch] on the device)."""
    X3, ldx, sX = _mat(X)
    assert X3.data_ptr() == X.data_ptr() and keep.dtype == torch.int32 and keep.shape[0] == X.shape[0]
    _check(lib().ttr_mask_cols(dtype_code(X.dtype), X.shape[1], X.shape[2], X.shape[0], X.data_ptr(), ldx, sX, keep.data_ptr(),
                               _stream()), "ttr_mask_cols")
    return X


def sweep_fused_ok(M: torch.Tensor) -> bool:
    """The fused row-Gram / rotate-Gram / projection kernels hold up to 64 rows."""
    return M.shape[1] <= 64


@_on_device
def spectrum_flat(sigma: torch.Tensor, keep: int, thr: float, use_delta: bool = False, delta2: float = 0.0,
                  delta2_dev: Optional[torch.Tensor] = None, rows32: Optional[torch.Tensor] = None) -> torch.Tensor:
    """int32 [batch]: 1 where sigma[b, r - 1] >= thr * sigma[b, 0] > 0 (sigma [batch, n] sorted decreasing), r = keep, or with
    ``use_delta`` the rank the tail-energy rule selects from these sigma provided that decision is robust (ttr_spectrum_flat)."""
    sigma = sigma.contiguous()
    batch, n = sigma.shape
    flat = torch.empty((batch,), dtype=torch.int32, device=sigma.device)
    if batch:
        _check(lib().ttr_spectrum_flat(dtype_code(sigma.dtype), n, batch, sigma.data_ptr(), n, int(keep), float(thr),
                                       int(bool(use_delta)), float(delta2),
                                       delta2_dev.data_ptr() if delta2_dev is not None else None,
                                       flat.data_ptr(), rows32.data_ptr() if rows32 is not None else None, _stream()),
               "ttr_spectrum_flat")
    return flat


@_on_device
def rowgram(M: torch.Tensor, V1: Optional[torch.Tensor] = None, skip: Optional[torch.Tensor] = None,
            rows32: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Split-K partials [batch, parts, R, R] of M M^T (V1 None) or of (V1^T M)(V1^T M)^T for M [batch, R, n], R <= 64
    (ttr_rowgram / ttr_rotgram); ``eigh_trunc`` sums the parts on load."""
    L = lib()
    dt = dtype_code(M.dtype)
    M, ldm, sM = _mat(M)
    batch, R, n = M.shape
    parts = int(L.ttr_sweep_gram_parts(n, max(batch, 1)))
    G = torch.empty((batch, parts, R, R), dtype=M.dtype, device=M.device)
    if batch == 0:
        return G
    if V1 is None:
        _check(L.ttr_rowgram(dt, R, n, batch, M.data_ptr(), ldm, sM, G.data_ptr(), parts,
                             rows32.data_ptr() if rows32 is not None else None, _stream()), "ttr_rowgram")
    else:
        V1, ldv, sV = _mat(V1)
        assert V1.shape == (batch, R, R)
        _check(L.ttr_rotgram(dt, R, n, batch, M.data_ptr(), ldm, sM, V1.data_ptr(), ldv, sV, G.data_ptr(), parts,
                             skip.data_ptr() if skip is not None else None,
                             rows32.data_ptr() if rows32 is not None else None, _stream()),
               "ttr_rotgram")
    return G


@_on_device
def project(M: torch.Tensor, V1: Optional[torch.Tensor], V2: torch.Tensor, sigma: Optional[torch.Tensor], ro: int,
            scale_right: bool, out: Optional[torch.Tensor] = None, want_left: bool = True,
            rows32: Optional[torch.Tensor] = None):
    """right [batch, ro, n] = diag(1/sigma) U^T M and left [batch, R, ro] = U diag(sigma) with U = V1 V2[:, :ro]
    (ttr_project; ``scale_right=False``: right = U^T M, left = U).  ``out``: optional contiguous destination of right."""
    L = lib()
    dt = dtype_code(M.dtype)
    M, ldm, sM = _mat(M)
    batch, R, n = M.shape
    V2, ldv2, sV2 = _mat(V2)
    v1p, ldv1, sV1 = None, 0, 0
    if V1 is not None:
        V1, ldv1, sV1 = _mat(V1)
        v1p = V1.data_ptr()
    if out is not None:
        assert tuple(out.shape) == (batch, ro, n) and out.is_contiguous() and out.dtype == M.dtype
        right = out
    else:
        right = torch.empty((batch, ro, n), dtype=M.dtype, device=M.device)
    left = torch.empty((batch, R, ro), dtype=M.dtype, device=M.device) if want_left else None
    if batch == 0:
        return right, left
    sp, ss = None, 0
    if sigma is not None:
        sigma = sigma.contiguous()
        sp, ss = sigma.data_ptr(), sigma.shape[-1]
    _check(L.ttr_project(dt, R, n, ro, batch, M.data_ptr(), ldm, sM, v1p, ldv1, sV1, V2.data_ptr(), ldv2, sV2, sp, ss,
                         int(bool(scale_right)), right.data_ptr(), n, ro * n,
                         left.data_ptr() if left is not None else None, ro, R * ro,
                         rows32.data_ptr() if rows32 is not None else None, _stream()), "ttr_project")
    return right, left


def colsweep_fused_ok(M: torch.Tensor) -> bool:
    """The fused tall-matrix kernels (column Gram / rotated Gram / projection) hold up to 64 columns."""
    return M.shape[2] <= 64


@_on_device
def colgram(M: torch.Tensor, V1: Optional[torch.Tensor] = None, skip: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[batch, n, n] = M^T M (V1 None) or (M V1)^T (M V1) for a tall M [batch, rows, n], n <= 64 (ttr_colgram)."""
    L = lib()
    dt = dtype_code(M.dtype)
    M, ldm, sM = _mat(M)
    batch, rows, n = M.shape
    G = torch.empty((batch, n, n), dtype=M.dtype, device=M.device)
    if batch == 0:
        return G
    wsb = L.ttr_colgram_workspace_bytes(dt, rows, n, batch)
    ws = torch.empty(wsb, dtype=torch.uint8, device=M.device) if wsb > 0 else None
    v1p, ldv, sV = None, 0, 0
    if V1 is not None:
        V1, ldv, sV = _mat(V1)
        assert V1.shape == (batch, n, n)
        v1p = V1.data_ptr()
    _check(L.ttr_colgram(dt, rows, n, batch, M.data_ptr(), ldm, sM, v1p, ldv, sV, G.data_ptr(),
                         ws.data_ptr() if ws is not None else None, wsb, skip.data_ptr() if skip is not None else None,
                         _stream()), "ttr_colgram")
    return G


@_on_device
def colproject(M: torch.Tensor, V1: Optional[torch.Tensor], V2: torch.Tensor, sigma: Optional[torch.Tensor], ro: int,
               left_ortho: bool, left_out: Optional[torch.Tensor] = None):
    """left [batch, rows, ro] = M U [/ sigma], right [batch, ro, n] = [sigma] U^T with U = V1 V2[:, :ro] (ttr_colproject).
    ``left_out``: optional contiguous [batch, rows, ro] destination of ``left`` (may lie in M's own storage BELOW the rows this
    call reads: the in-place first step of a config-scale dense TT-SVD, ``_hipops._colproject_inplace``)."""
    L = lib()
    dt = dtype_code(M.dtype)
    M, ldm, sM = _mat(M)
    batch, rows, n = M.shape
    V2, ldv2, sV2 = _mat(V2)
    v1p, ldv1, sV1 = None, 0, 0
    if V1 is not None and rows >= 4096:
        # tall inputs: U = V1 V2[:, :ro] as one small GEMM up front -- the kernel without the prologue product keeps a third of
        # the LDS and three times the workgroups per CU (its main loop covers the HBM latency with resident waves alone)
        V2, ldv2, sV2 = _mat(gemm(V1, V2[:, :, :ro]))
        V1 = None
    if V1 is not None:
        V1, ldv1, sV1 = _mat(V1)
        v1p = V1.data_ptr()
    if left_out is not None:
        assert tuple(left_out.shape) == (batch, rows, ro) and left_out.is_contiguous() and left_out.dtype == M.dtype
        left = left_out
    else:
        left = torch.empty((batch, rows, ro), dtype=M.dtype, device=M.device)
    right = torch.empty((batch, ro, n), dtype=M.dtype, device=M.device)
    if batch == 0:
        return left, right
    sp, ss = None, 0
    if sigma is not None:
        sigma = sigma.contiguous()
        sp, ss = sigma.data_ptr(), sigma.shape[-1]
    _check(L.ttr_colproject(dt, rows, n, ro, batch, M.data_ptr(), ldm, sM, v1p, ldv1, sV1, V2.data_ptr(), ldv2, sV2, sp, ss,
                            int(bool(left_ortho)), left.data_ptr(), ro, rows * ro, right.data_ptr(), n, ro * n, _stream()),
           "ttr_colproject")
    return left, right


@_on_device
def pow2_normalize(x: torch.Tensor, expo_acc: Optional[torch.Tensor] = None, exponent_only: bool = False):
    """[batch, ...] -> (x * 2^-e per batch item, e int32 [batch]) with e the binary exponent of ||x[b]||;
    ``expo_acc`` (int32 [batch]) is incremented by e in place.  ``exponent_only``: returns (None, e), x is only read."""
    L = lib()
    dt = dtype_code(x.dtype)
    x = x.contiguous()
    batch = x.shape[0]
    count = x[0].numel() if batch > 0 else 0
    out = None if exponent_only else torch.empty_like(x)
    e = torch.empty((batch,), dtype=torch.int32, device=x.device)   # (every entry is written by the kernel)
    if batch == 0:
        return out, e
    _check(L.ttr_pow2_normalize(dt, count, batch, x.data_ptr(), count, out.data_ptr() if out is not None else None, count,
                                e.data_ptr(), expo_acc.data_ptr() if expo_acc is not None else None, _stream()),
           "ttr_pow2_normalize")
    return out, e


@_on_device
def scale_batch(x: torch.Tensor, scale=None, expo: Optional[torch.Tensor] = None, expo_sign: int = 1) -> torch.Tensor:
    """out[b] = x[b] * scale[b] * 2^(expo_sign * expo[b]) for a [batch, ...] tensor.  ``scale``: None, a python
    number (one scalar for the whole batch) or a [batch] tensor; ``expo``: None or int32 [batch]."""
    L = lib()
    dt = dtype_code(x.dtype)
    x = x.contiguous()
    batch = x.shape[0]
    count = x[0].numel() if batch > 0 else 0
    out = torch.empty_like(x)
    if batch == 0 or count == 0:
        return out
    sp, ss = None, 0
    if scale is not None:
        if not isinstance(scale, torch.Tensor):
            scale = torch.full((1,), float(scale), dtype=x.dtype, device=x.device)  # (a fill, not arithmetic)
            ss = 0
        else:
            scale = scale.to(x.dtype).contiguous()
            ss = 1
        sp = scale.data_ptr()
    _check(L.ttr_scale_batch(dt, count, batch, x.data_ptr(), count, sp, ss,
                             expo.data_ptr() if expo is not None else None, int(expo_sign),
                             out.data_ptr(), count, _stream()), "ttr_scale_batch")
    return out


@_on_device
def carry_rows32(R: torch.Tensor) -> torch.Tensor:
    """[batch, 64, cols] -> int32 [batch]: 1 where rows 32.. of R[b] are negligible by the packing test of the fused push
    (ttr_carry_rows32): the `rows32` flags of a carry that no ``qr_factor_pushed`` follows (the sweep's last core)."""
    L = lib()
    R, ldr, sR = _mat(R)
    batch, rows, cols = R.shape
    assert rows == 64
    flag = torch.empty((batch,), dtype=torch.int32, device=R.device)
    if batch:
        _check(L.ttr_carry_rows32(dtype_code(R.dtype), cols, batch, R.data_ptr(), ldr, sR, flag.data_ptr(), _stream()),
               "ttr_carry_rows32")
    return flag


@_on_device
def orth_fixup(X: torch.Tensor, sigma: torch.Tensor, r: int, dead_rel: float, columns: bool = False,
               rank_dev: Optional[torch.Tensor] = None) -> None:
    """In place: orthonormal completion of the kept vectors whose sigma <= dead_rel * sigma_max (ttr_orth_fixup).
    ``X``: contiguous [batch, r, n] (rows are the vectors) or, with ``columns=True``, [batch, n, r]."""
    L = lib()
    dt = dtype_code(X.dtype)
    assert X.is_contiguous() and X.dim() == 3 and sigma.is_contiguous()
    batch = X.shape[0]
    if columns:
        n, vs, es = X.shape[1], 1, X.shape[2]
        assert X.shape[2] == r
    else:
        n, vs, es = X.shape[2], X.shape[2], 1
        assert X.shape[1] == r
    if batch == 0 or r == 0 or n == 0:
        return
    wsb = int(L.ttr_orth_fixup_workspace_bytes(dt, r, n, batch, es))   # > 0: a large batch, three launches per round
    ws = torch.empty(wsb, dtype=torch.uint8, device=X.device) if wsb > 0 else None
    _check(L.ttr_orth_fixup(dt, r, n, batch, X.data_ptr(), vs, es, X.shape[1] * X.shape[2], sigma.data_ptr(),
                            sigma.shape[-1], float(dead_rel), rank_dev.data_ptr() if rank_dev is not None else None,
                            ws.data_ptr() if ws is not None else None, wsb, _stream()),
           "ttr_orth_fixup")


@_on_device
def krp_contract(T: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """out[p, q, r] = sum_j T[p, j, q, r] * B[j, r] for contiguous T [P, J, Q, R], B [J, R]."""
    L = lib()
    dt = dtype_code(T.dtype)
    assert T.dim() == 4 and B.dim() == 2 and T.is_cuda and T.dtype == B.dtype
    P, J, Q, R = T.shape
    assert B.shape[0] == J and B.shape[1] == R
    T = T.contiguous()
    B = B.contiguous()
    out = torch.empty((P, Q, R), dtype=T.dtype, device=T.device)
    code = L.ttr_krp_contract(dt, P, J, Q, R, T.data_ptr(), B.data_ptr(), R, out.data_ptr(), _stream())
    _check(code, "ttr_krp_contract")
    return out


@_on_device
def hadamard(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    L = lib()
    dt = dtype_code(a.dtype)
    assert a.shape == b.shape and a.dtype == b.dtype and a.is_cuda
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    _check(L.ttr_hadamard(dt, a.numel(), a.data_ptr(), b.data_ptr(), out.data_ptr(), _stream()), "ttr_hadamard")
    return out


@_on_device
def core_kron(a: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """[B, R1, I, R2] (x) [B, S1, I, S2] -> [B, R1*S1, I, R2*S2] (slice-wise Kronecker product)."""
    L = lib()
    dt = dtype_code(a.dtype)
    assert a.dim() == 4 and c.dim() == 4 and a.dtype == c.dtype and a.shape[0] == c.shape[0] and a.shape[2] == c.shape[2]
    a, c = a.contiguous(), c.contiguous()
    B, R1, I, R2 = a.shape
    _, S1, _, S2 = c.shape
    out = torch.empty((B, R1 * S1, I, R2 * S2), dtype=a.dtype, device=a.device)
    _check(L.ttr_core_kron(dt, B, R1, S1, I, R2, S2, a.data_ptr(), c.data_ptr(), out.data_ptr(), _stream()), "ttr_core_kron")
    return out


KNOB_QR_PANEL = 0
KNOB_BJ_INNER_SWEEPS = 1
KNOB_GEMM_BIG = 2
KNOB_QR_STAMP_BX, KNOB_QR_STAMP_BY = 3, 4
KNOB_QR_F64_NW4 = 5
KNOB_QR_RANK_SKIP = 6
KNOB_QR_PACK = 7
KNOB_EIGH_SMALL = 8
KNOB_RANK_NOISE_FLOOR = 9
KNOB_QR_STAGGER = 16
KNOB_QR_PACK_PRE = 17
KNOB_EIGH_BIG_OCC = 18
KNOB_ORTH_ROUNDS = 10
KNOB_JACOBI_LIVE_WAVE = 11
KNOB_ORTH_V2 = 12
KNOB_QR_INTERLEAVE = 13
KNOB_SWEEP_STAGGER = 14
KNOB_ORTH_SPLIT = 15


ALG_SVD, ALG_EIG = 0, 1
RANK_NONE = 2**31 - 1   # rank cap meaning "none" (round.py:83-84)


def round_tt_plan(dt: torch.dtype, shapes, rcap, batch: int, eps_mode: bool) -> int:
    """Workspace bytes of ``round_tt_sweep`` for cores of ``shapes`` [(r0, I, r1), ...], or a negative status when the train
    lies outside the envelope of ttr_round_tt (the caller then runs its own loop over the per-kernel entries)."""
    N = len(shapes)
    sh = (c_int64 * (3 * N))(*[int(v) for s3 in shapes for v in s3])
    rc = (c_int64 * max(N - 1, 1))(*[int(r) for r in rcap])
    return int(lib().ttr_round_tt_workspace_bytes(dtype_code(dt), N, sh, rc, int(batch), int(bool(eps_mode))))


@_on_device
def round_tt_sweep(cores, rcap, algorithm: str, eps_mode: bool, eps: float, flat_thr: float, use_eigh_top: bool, outs,
                   ranks_dev: Optional[torch.Tensor], zero_flag: Optional[torch.Tensor], ws: torch.Tensor) -> None:
    """ttr_round_tt: both sweeps of tensor.py:2008-2083 on contiguous [B, r0, I, r1] ``cores`` in ONE library call; the
    rounded cores are written to ``outs`` (contiguous, at the rank caps).  See include/ttround_hip.h."""
    N = len(cores)
    B = cores[0].shape[0]
    sh = (c_int64 * (3 * N))(*[int(v) for c in cores for v in c.shape[1:]])
    rc = (c_int64 * max(N - 1, 1))(*[int(r) for r in rcap])
    cin = (c_void_p * N)(*[c.data_ptr() for c in cores])
    cout = (c_void_p * N)(*[o.data_ptr() for o in outs])
    code = lib().ttr_round_tt(dtype_code(cores[0].dtype), N, sh, B, cin, rc, ALG_SVD if algorithm == "svd" else ALG_EIG,
                              int(bool(eps_mode)), float(eps), float(flat_thr), int(bool(use_eigh_top)), cout,
                              ranks_dev.data_ptr() if ranks_dev is not None else None,
                              zero_flag.data_ptr() if zero_flag is not None else None, ws.data_ptr(), ws.numel(), _stream())
    _check(code, "ttr_round_tt")


def set_knob(knob: int, value: int):
    """Diagnostics: select a kernel variant (see ttr_debug_set_knob in the header)."""
    _check(lib().ttr_debug_set_knob(int(knob), int(value)), "ttr_debug_set_knob")


def prof_enable(on):
    """False / True: per-kind device times; 2: also the executed-work census (``prof_collect_work``)."""
    _check(lib().ttr_prof_enable(int(on)), "ttr_prof_enable")


def prof_collect_work():
    """{kind: {"flops": f, "bytes": b}} executed by the instrumented launches since the last collect (census mode)."""
    n = len(PROF_KINDS)
    fl = (c_double * n)()
    by = (c_double * n)()
    _check(lib().ttr_prof_collect_work(fl, by), "ttr_prof_collect_work")
    return {k: {"flops": fl[i], "bytes": by[i]} for i, k in enumerate(PROF_KINDS)}


def prof_collect():
    n = len(PROF_KINDS)
    ms = (c_double * n)()
    cnt = (c_int64 * n)()
    lib().ttr_prof_collect(ms, cnt)
    return {k: {"ms": ms[i], "launches": cnt[i]} for i, k in enumerate(PROF_KINDS)}
