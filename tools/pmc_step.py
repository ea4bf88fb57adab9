#!/usr/bin/env python3
"""Identical single-stream steps of one workload for rocprofv3 --pmc / --kernel-trace passes (every kernel is dispatched the same
number of times per step, nothing else runs):
    python tools/pmc_step.py [B] [input] [steps]
input:  gg (default)   the metric: round_tt(rmax=32) of B resident 64^8 rank-64 trains, t = g + g
        decay0.5 / decay1.0   the same shapes with bond singular values ~ 2^(-j/2) / 2^-j (SURVEY 8d's second input: no shortcut fires)
        c3             BASELINE config C3's per-GPU share: B (default 64) dense 32^5 tensors -> TT, rmax 8
        c4             BASELINE config C4: CP-ALS R = 32 on a dense 256^4 tensor, `steps` sweeps after the HOSVD init (B is ignored)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
import tntorch_amd as tn
from tntorch_amd import _hipops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
kind = sys.argv[2] if len(sys.argv) > 2 else "gg"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
_hipops.STREAM_CHUNKS_ENABLED = False
dev = torch.device("cuda")
if kind in ("gg", "decay0.5", "decay1.0"):
    inp = bench.make_input(B, dev, seed=1234) if kind == "gg" else bench.make_decaying_input(B, dev, seed=777, decay=float(kind[5:]))
    for _ in range(steps):
        t = tn.Tensor(inp, batch=True)
        t.round_tt(rmax=32)
elif kind == "c3":
    gen = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn((B, 32, 32, 32, 32, 32), generator=gen, device=dev)
    for _ in range(steps):
        t = tn.Tensor(X, ranks_tt=8, batch=True)
elif kind == "c4":
    gen = torch.Generator(device=dev).manual_seed(0)
    I, R = 256, 32
    X = torch.randn((I, I, I, I), generator=gen, device=dev)
    _hipops.cp_als(X, R, max_iter=steps, tol=-1.0)
else:
    raise SystemExit(f"unknown input {kind}")
torch.cuda.synchronize()
print("pmc_step done", B, kind, steps)
