#!/usr/bin/env python3
"""Run each hot kernel of the metric workload in isolation (for rocprofv3 --pmc passes).
    python tools/kernel_probe.py [B] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tntorch_amd import _hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.manual_seed(0)
core = torch.randn(B, 64, 64, 64, device="cuda")
Rm = torch.randn(B, 64, 64, device="cuda")
M = torch.randn(B, 64, 2048, device="cuda")
L32 = torch.randn(B, 64, 32, device="cuda")
for _ in range(reps):
    f = _hip.qr_factor_pushed(Rm, core)       # fused push + TSQR factor (3 tree levels)
    Q = _hip.qr_apply(f, L32)                  # Q [U sigma; 0], 32 columns
    G = _hip.gemm(M, M, transB=True)           # Gram        (64x2048 @ 2048x64)
    V, s, info = _hip.eigh_trunc(G, _hip.EIG_RAW, False, 0.0, 64, abs_floor=_hip.SOLVER_TRIDIAG)   # pass 1
    M1 = _hip.gemm(V, M, transA=True)          # rotate      (64x64 @ 64x2048)
    G1 = _hip.gemm(M1, M1, transB=True)
    V2, s2, info = _hip.eigh_trunc(G1, _hip.EIG_RAW, False, 0.0, 32, abs_floor=_hip.SOLVER_JACOBI_ABS)  # pass 2
    M2 = _hip.gemm(V2[:, :, :32], M1, transA=True, rowscale=s2, rowscale_mode=_hip.SCALE_DIV)  # projection
torch.cuda.synchronize()
print("probe done", B, reps)
