"""What the three-launch rounds of ttr_orth_fixup (TTR_KNOB_ORTH_SPLIT) cost on a batch WITHOUT dead directions (the headline
input, where all of their launches exit at once) and gain on the 2^-j input, alternating in one process.
    python tools/probes/orth_split_headline_ab.py [B]"""
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
for name, inp in (("g+g", bench.make_input(B, dev, seed=1234)), ("decay 1.0", bench.make_decaying_input(B, dev, seed=777, decay=1.0))):
    def step():
        t = tn.Tensor(inp, batch=True); t.round_tt(rmax=32); return t
    for rep in range(4):
        for sp in (0, 256):
            _hip.set_knob(_hip.KNOB_ORTH_SPLIT, sp)
            evs = []
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(12):
                if len(evs) >= 2:
                    evs.pop(0).synchronize()
                step()
                e = torch.cuda.Event(); e.record(); evs.append(e)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 12 * 1e3
            print(f"{name} B={B} orth_split={sp}: {ms:.3f} ms/step = {B * 8 / ms * 1e3:.0f} cores/s")
    del inp
    torch.cuda.empty_cache()
_hip.set_knob(_hip.KNOB_ORTH_SPLIT, 2048)
