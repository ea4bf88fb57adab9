"""Cycle stamps of level-0 blocks of the fused push + factor kernel at different places of the grid (B = 2048): block (0, 0)
starts together with every other first-wave block (all of them stream their core slices at once); blocks further into
the grid show the steady state.  Prints per block: total cycles and the deltas [push | per panel: transpose, phases,
T / W, update]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tntorch_amd import _hip as h

L = h.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.manual_seed(0)
Rm = torch.triu(torch.randn(B, 64, 64, device="cuda")); core = torch.randn(B, 64, 64, 64, device="cuda")
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
h.qr_factor_pushed(Rm, core); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); h.qr_factor_pushed(Rm, core); e1.record(); torch.cuda.synchronize()
print(f"launch (both levels): {e0.elapsed_time(e1):.3f} ms")
for bx, by in ((0, 0), (3, 40), (5, B // 4), (2, B // 2), (7, 3 * B // 4), (4, B - 3)):
    h.set_knob(h.KNOB_QR_STAMP_BX, bx); h.set_knob(h.KNOB_QR_STAMP_BY, by)
    L.ttr_debug_set_qr_stamps(buf.data_ptr()); buf.zero_()
    h.qr_factor_pushed(Rm, core); torch.cuda.synchronize()
    L.ttr_debug_set_qr_stamps(None)
    st = [x for x in buf.cpu().tolist() if x != 0]
    d = [st[i + 1] - st[i] for i in range(len(st) - 1)]
    print(f"block ({bx}, {by}): total {st[-1] - st[0]}  push {d[0]}  panels {[d[1 + 4 * k: 5 + 4 * k] for k in range(4)]}")
h.set_knob(h.KNOB_QR_STAMP_BX, 0); h.set_knob(h.KNOB_QR_STAMP_BY, 0)
