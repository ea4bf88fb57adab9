"""Cycle stamps of ttr_orth_fixup's block kernel (item 0 of the last launch of a decaying-spectrum step, 2^-j): where a round's
time goes -- Gram pass / coefficient section / apply pass.  python tools/probes/orth_stamps.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
import tntorch_amd as tn
from tntorch_amd import _hip, _hipops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda", 0)
_hipops.STREAM_CHUNKS_ENABLED = False
inp = bench.make_decaying_input(B, dev, seed=777, decay=1.0)
buf = torch.zeros(64, dtype=torch.int64, device=dev)
for v2 in (1, 0):
    _hip.set_knob(_hip.KNOB_ORTH_V2, v2)
    t = tn.Tensor(inp, batch=True); t.round_tt(rmax=32)
    torch.cuda.synchronize()
    buf.zero_()
    _hip.lib().ttr_debug_set_qr_stamps(buf.data_ptr())
    _hip.set_knob(_hip.KNOB_QR_STAMP_BX, -1)
    t = tn.Tensor(inp, batch=True); t.round_tt(rmax=32)
    torch.cuda.synchronize()
    _hip.lib().ttr_debug_set_qr_stamps(None)
    _hip.set_knob(_hip.KNOB_QR_STAMP_BX, 0)
    s = [int(x) for x in buf.tolist() if x]
    d = [s[i + 1] - s[i] for i in range(len(s) - 1)]
    names = ["gram pass", "coefficients", "apply pass"]
    print(f"orth_v2={v2} B={B}: {len(s)} stamps; phases in cycles (clock64, 100 MHz-class counter: compare ratios): " +
          ", ".join(f"r{i // 3} {names[i % 3]} {x}" for i, x in enumerate(d)) + f"; total {s[-1] - s[0]}")
