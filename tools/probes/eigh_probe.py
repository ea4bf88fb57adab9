"""Timing of the eigensolver launches of the metric sweep in isolation: tridiagonal QL (pass 1) and Jacobi LIVE (pass 2) on
Gram matrices like the metric's (rank 32 of 64) and on full-rank ones; batch 2048 and 1.   python tools/probes/eigh_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tntorch_amd import _hip

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

torch.manual_seed(0)
for B in (2048, 1):
    for kind in ("rank32", "full"):
        if kind == "rank32":
            M = torch.randn(B, 32, 2048, device="cuda")
            M = torch.cat([M, M + 1e-7 * torch.randn_like(M)], dim=1)
        else:
            M = torch.randn(B, 64, 2048, device="cuda")
        G = _hip.gemm(M, M, transB=True)
        sw = torch.zeros(B, dtype=torch.int32, device="cuda")
        V1, s1, _ = _hip.eigh_trunc(G, _hip.EIG_RAW, False, 0.0, 64, abs_floor=_hip.SOLVER_TRIDIAG, sweeps=sw)
        t1 = timeit(lambda: _hip.eigh_trunc(G, _hip.EIG_RAW, False, 0.0, 64, abs_floor=_hip.SOLVER_TRIDIAG))
        G2 = _hip.rowgram(M, V1)
        sw2 = torch.zeros(B, dtype=torch.int32, device="cuda")
        _hip.eigh_trunc(G2, _hip.EIG_RAW, False, 0.0, 32, abs_floor=_hip.SOLVER_JACOBI_LIVE, sweeps=sw2)
        t2 = timeit(lambda: _hip.eigh_trunc(G2, _hip.EIG_RAW, False, 0.0, 32, abs_floor=_hip.SOLVER_JACOBI_LIVE))
        t3 = timeit(lambda: _hip.eigh_trunc(G, _hip.EIG_RAW, False, 0.0, 64, abs_floor=_hip.SOLVER_JACOBI_ABS))
        print(f"B={B} {kind}: tridiag {t1:.0f} us (QL iterations min/mean/max {sw.min().item()}/{sw.float().mean().item():.0f}/{sw.max().item()}), "
              f"jacobi-live pass 2 {t2:.0f} us (sweeps max {sw2.max().item()}), full jacobi on G {t3:.0f} us")
