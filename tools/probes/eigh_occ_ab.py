"""The 32-row top-r eigensolver instance at two (default), three and four waves per SIMD (TTR_KNOB_EIGH_SMALL = 1 / 2 / 3: builds
capped at 256 / 168 / 128 VGPRs, 0 / 44 / 124 spilled registers) on the headline input, alternating in one process.
    python tools/probes/eigh_occ_ab.py [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
import tntorch_amd as tn
from tntorch_amd import _hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
inp = bench.make_input(B, dev, seed=1234)


def step():
    t = tn.Tensor(inp, batch=True); t.round_tt(rmax=32); return t


ref = None
for rep in range(3):
    for v in (1, 2, 3):
        _hip.set_knob(_hip.KNOB_EIGH_SMALL, v)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        evs = []
        t0 = time.perf_counter()
        for _ in range(12):
            if len(evs) >= 2:
                evs.pop(0).synchronize()
            t = step()
            e = torch.cuda.Event(); e.record(); evs.append(e)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 12 * 1e3
        _hip.prof_enable(True); step(); torch.cuda.synchronize(); p = _hip.prof_collect(); _hip.prof_enable(False)
        same = ""
        if v == 1:
            ref = [c.clone() for c in t.cores]
        elif ref is not None:
            same = "; cores bit-identical to variant 1: " + str(all(torch.equal(a, b) for a, b in zip(ref, t.cores)))
        print(f"B={B} eigh_small={v}: {ms:.3f} ms/step = {B * 8 / ms * 1e3:.0f} cores/s; eigh {p['eigh']['ms']:.3f} ms / {p['eigh']['launches']} launches{same}", flush=True)
_hip.set_knob(_hip.KNOB_EIGH_SMALL, 2)
