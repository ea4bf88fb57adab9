"""Orthogonality / residual of the tridiagonal eigensolver's V on metric-like and full-rank Gram matrices (fp32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tntorch_amd import _hip
torch.manual_seed(0)
B = 256
for kind in ("rank32", "full", "graded"):
    M = torch.randn(B, 64, 2048, device="cuda")
    if kind == "rank32":
        M[:, 32:] = M[:, :32] + 1e-7 * torch.randn(B, 32, 2048, device="cuda")
    if kind == "graded":
        M = M * (0.8 ** torch.arange(64, device="cuda"))[None, :, None]
    G = _hip.gemm(M, M, transB=True)
    V, sig, _ = _hip.eigh_trunc(G, _hip.EIG_RAW, False, 0.0, 64, abs_floor=_hip.SOLVER_TRIDIAG)
    Vd, Gd = V.double(), G.double()
    orth = (Vd.transpose(1, 2) @ Vd - torch.eye(64, device="cuda", dtype=torch.float64)).abs().amax(dim=(1, 2))
    lam = (sig.double() ** 2)
    res = (Gd @ Vd - Vd * lam[:, None, :]).abs().amax(dim=(1, 2)) / lam[:, 0]
    ref = torch.linalg.eigvalsh(Gd).flip(-1).clamp_min(0)
    ev = ((lam - ref).abs().amax(dim=1) / ref[:, 0])
    print(f"{kind}: |V^T V - I| max {orth.max().item():.2e} mean {orth.mean().item():.2e}; residual max {res.max().item():.2e}; eigenvalue err max {ev.max().item():.2e}")
