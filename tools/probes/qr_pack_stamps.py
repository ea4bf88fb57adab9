"""Cycle stamps of level-0 blocks of the fused push + factor kernel on a rank-inflated input (R of numerical rank 32 of 64, core =
blockdiag(g, g): the metric's structure), with and without row packing (TTR_KNOB_QR_PACK): per block total cycles and the deltas
[push | per panel: transpose + skip test, phases, T / W, update]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tntorch_amd import _hip as h

L = h.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.manual_seed(0)
top = torch.triu(torch.randn(B, 32, 64, device="cuda"))
Rm = torch.cat([top, 1e-8 * torch.triu(torch.randn(B, 32, 64, device="cuda"), diagonal=32)], dim=1)
g = torch.randn(B, 32, 64, 32, device="cuda")
z = torch.zeros_like(g)
core = torch.cat([torch.cat([g, z], dim=-1), torch.cat([z, g], dim=-1)], dim=1).contiguous()
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
for pack in (0, 1, 3):
    h.set_knob(h.KNOB_QR_PACK, pack)
    h.qr_factor_pushed(Rm, core); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); h.qr_factor_pushed(Rm, core); e1.record(); torch.cuda.synchronize()
    print(f"pack={pack}: launch (both levels): {e0.elapsed_time(e1):.3f} ms")
    for Bs in (256, 512, 1024):
        e0.record(); h.qr_factor_pushed(Rm[:Bs], core[:Bs]); e1.record(); torch.cuda.synchronize()
        print(f"    B={Bs}: {e0.elapsed_time(e1):.3f} ms")
    for bx, by in ((1, B // 4), (2, B // 2), (5, B // 2), (6, B // 4), (3, B // 2 + 1)):
        h.set_knob(h.KNOB_QR_STAMP_BX, bx); h.set_knob(h.KNOB_QR_STAMP_BY, by)
        L.ttr_debug_set_qr_stamps(buf.data_ptr()); buf.zero_()
        h.qr_factor_pushed(Rm, core); torch.cuda.synchronize()
        L.ttr_debug_set_qr_stamps(None)
        st = [x for x in buf.cpu().tolist() if x != 0]
        d = [st[i + 1] - st[i] for i in range(len(st) - 1)]
        print(f"  block ({bx}, {by}): stamps {len(st)} total {st[-1] - st[0] if st else 0}  push {d[0] if d else 0}  rest {d[1:]}")
h.set_knob(h.KNOB_QR_STAMP_BX, 0); h.set_knob(h.KNOB_QR_STAMP_BY, 0); h.set_knob(h.KNOB_QR_PACK, 1)
