#!/usr/bin/env python3
"""Run each hot kernel of the metric workload in isolation (for rocprofv3 --pmc passes).
    python tools/kernel_probe.py [B] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tntorch_amd import _hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.manual_seed(0)
A = torch.randn(B, 4096, 64, device="cuda")
M = torch.randn(B, 64, 2048, device="cuda")
R = torch.randn(B, 64, 64, device="cuda")
core = torch.randn(B, 64, 4096, device="cuda")
G = _hip.gemm(M, M, transB=True)
for _ in range(reps):
    Q, Rr = _hip.qr(A)                       # qr_factor + qr_apply (3 tree levels each)
    P = _hip.gemm(R, core)                   # push right  (64x64 @ 64x4096)
    G = _hip.gemm(M, M, transB=True)         # Gram        (64x2048 @ 2048x64)
    V, s, info = _hip.eigh_trunc(G, _hip.EIG_RAW, False, 0.0, 32)
    M2 = _hip.gemm(V[:, :, :32], M, transA=True, rowscale=s, rowscale_mode=_hip.SCALE_DIV)  # projection
    L = _hip.gemm(A, V[:, :, :32], colscale=s, colscale_mode=_hip.SCALE_MUL)               # push left
torch.cuda.synchronize()
print("probe done", B, reps)
