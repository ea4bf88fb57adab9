"""The second round's Gram matrix of ttr_orth_fixup's three-launch rounds: left by the first round's product launch
(TTR_KNOB_ORTH_V2 = 2, the default) against a ttr_rowgram launch of its own (1), alternating in one process on the 2^-j input
(where every item has dead directions) and on the headline input (where these launches exit at once).
    python tools/probes/orth_fuse_ab.py [B]"""
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
for name, inp in (("decay 1.0", bench.make_decaying_input(B, dev, seed=777, decay=1.0)), ("g+g", bench.make_input(B, dev, seed=1234))):
    def step():
        t = tn.Tensor(inp, batch=True); t.round_tt(rmax=32); return t
    outs = {}
    for rep in range(3):
        for v2 in (1, 2):
            _hip.set_knob(_hip.KNOB_ORTH_V2, v2)
            evs = []
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(12):
                if len(evs) >= 2:
                    evs.pop(0).synchronize()
                t = step()
                e = torch.cuda.Event(); e.record(); evs.append(e)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 12 * 1e3
            outs[v2] = t
            print(f"{name} B={B} orth_v2={v2}: {ms:.3f} ms/step = {B * 8 / ms * 1e3:.0f} cores/s", flush=True)
    # the two variants' cores: orthonormal rows either way, the same vectors to rounding
    for k in (3, 6):
        a, b = outs[1].cores[k][:64].double(), outs[2].cores[k][:64].double()
        ra = a.reshape(a.shape[0], a.shape[1], -1)
        rb = b.reshape(b.shape[0], b.shape[1], -1)
        eye = torch.eye(ra.shape[1], dtype=torch.float64, device=dev)
        print(f"  core {k}: orthonormality defect {float((ra @ ra.transpose(1, 2) - eye).abs().max()):.2e} (1) "
              f"{float((rb @ rb.transpose(1, 2) - eye).abs().max()):.2e} (2); max |difference| {float((ra - rb).abs().max()):.2e}", flush=True)
    del inp, outs
    torch.cuda.empty_cache()
_hip.set_knob(_hip.KNOB_ORTH_V2, 2)
