"""A/B of TTR_KNOB_QR_INTERLEAVE (order of the block-major QR launches) and of the staggered chunk order of ttr_orth_fixup (V2):
per-kind device times (single stream) and the two-stream wall time of a step, on the headline input (t = g + g) and the two
decaying-spectrum variants.    python tools/probes/interleave_ab.py [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
import tntorch_amd as tn
from tntorch_amd import _hip, _hipops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda", 0)
inputs = {"g+g": bench.make_input(B, dev, seed=1234), "decay 1.0": bench.make_decaying_input(B, dev, seed=777, decay=1.0),
          "decay 0.5": bench.make_decaying_input(B, dev, seed=777, decay=0.5)}
for name, inp in inputs.items():
    def step():
        t = tn.Tensor(inp, batch=True); t.round_tt(rmax=32); return t
    for il, ov, sg, sp in ((1, 1, 1, 256), (1, 1, 1, 0), (0, 0, 0, 0)):
        _hip.set_knob(_hip.KNOB_QR_INTERLEAVE, il)
        _hip.set_knob(_hip.KNOB_ORTH_V2, ov)
        _hip.set_knob(_hip.KNOB_SWEEP_STAGGER, sg)
        _hip.set_knob(_hip.KNOB_ORTH_SPLIT, sp)
        _hipops.STREAM_CHUNKS_ENABLED = True
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 8 * 1e3
        _hipops.STREAM_CHUNKS_ENABLED = False
        step(); torch.cuda.synchronize()
        _hip.prof_enable(True)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        p = _hip.prof_collect()
        _hip.prof_enable(False)
        print(f"{name} B={B} interleave={il} orth_v2={ov} stagger={sg} orth_split={sp}: step {wall:.2f} ms (two streams) = {B * 8 / wall * 1e3:.0f} cores/s; per kind ms/step: " +
              ", ".join(f"{k} {v['ms'] / 3:.2f}" for k, v in p.items() if v["launches"]))
_hip.set_knob(_hip.KNOB_QR_INTERLEAVE, 1); _hip.set_knob(_hip.KNOB_ORTH_V2, 2); _hip.set_knob(_hip.KNOB_SWEEP_STAGGER, 1); _hip.set_knob(_hip.KNOB_ORTH_SPLIT, 2048)
