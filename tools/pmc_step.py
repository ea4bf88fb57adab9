#!/usr/bin/env python3
"""Two identical single-stream steps of the metric workload (for rocprofv3 --pmc / --kernel-trace passes): every
kernel of round_tt(rmax=32) on B resident 64^8 rank-64 trains is dispatched exactly twice.
    python tools/pmc_step.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
import tntorch_amd as tn
from tntorch_amd import _hipops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
_hipops.STREAM_CHUNKS_ENABLED = False
inp = bench.make_input(B, torch.device("cuda"), seed=1234)
for _ in range(2):
    t = tn.Tensor(inp, batch=True)
    t.round_tt(rmax=32)
torch.cuda.synchronize()
print("pmc_step done", B)
