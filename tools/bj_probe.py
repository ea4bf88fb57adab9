"""Block-Jacobi driver (large-n eigenproblems): accuracy and time per size.  GPU only."""
import sys, time, math
sys.path.insert(0, "/root/repo")
import torch
from tntorch_amd import _hip as h, _hipops

for thr in (0,):
  for dt in (torch.float32, torch.float64):
    for n, B, solver in [(288, 1, 2), (512, 1, 2), (1024, 1, 2), (256, 8, 2), (256, 64, 2), (128, 2, 1)]:
        g = torch.Generator().manual_seed(n)
        Mx = torch.randn(B, n, 2 * n + 1, generator=g, dtype=torch.float64)
        Mx = Mx * torch.logspace(0, -3, n, dtype=torch.float64)[None, :, None]
        G = (Mx @ Mx.transpose(1, 2)).to(dt).cuda()
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            V, sig, info = _hipops._eigh_any(G, h.EIG_RAW, False, 0.0, n, solver)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
        V, sig = V.cpu().double(), sig.cpu().double()
        wref = torch.linalg.eigvalsh(G.cpu().double()).flip(-1).clamp_min(0)
        e_eig = ((sig**2 - wref).abs().max(dim=1).values / wref[:, 0]).max()
        eye = torch.eye(n, dtype=torch.float64)
        e_orth = (V.transpose(1, 2) @ V - eye).abs().max()
        resid = (G.cpu().double() @ V - V * (sig**2)[:, None, :]).abs().max() / wref.max()
        print(f"{str(dt):14s} n={n:5d} B={B:3d} solver={solver}: {el*1e3:8.1f} ms  eig {e_eig:.2e} orth {e_orth:.2e} resid {resid:.2e}", flush=True)
