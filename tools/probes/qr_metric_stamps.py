"""Cycle stamps of level-0 blocks of the fused push + factor kernel on the METRIC input (t = g + g: second core pushed with the
first core's R factor), B = 2048: per block total cycles and the deltas [push | per panel: transpose + skip test, phases, T / W,
update], with TTR_KNOB_QR_PACK = 3 (default) and 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from tntorch_amd import _hip as h

L = h.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda", 0)
inp = bench.make_input(B, dev, seed=1)
c0 = inp[0]
R = h.qr_factor(c0.reshape(B, -1, c0.shape[-1])).R
Rn, _ = h.pow2_normalize(R)
f = h.qr_factor_pushed(Rn, inp[1]); R2, _ = h.pow2_normalize(f.R)
core = inp[2]
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
for pack in (3, 0):
    h.set_knob(h.KNOB_QR_PACK, pack)
    h.qr_factor_pushed(R2, core); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f = h.qr_factor_pushed(R2, core); e1.record(); torch.cuda.synchronize()
    print(f"pack={pack}: launch (both levels): {e0.elapsed_time(e1):.3f} ms; packed items: {int((f.rows32 != 0).sum()) if f.rows32 is not None else None}")
    for bx, by in ((0, 0), (1, B // 4), (2, B // 2), (3, B // 2 + 1), (0, 3 * B // 4), (6, B // 2)):
        h.set_knob(h.KNOB_QR_STAMP_BX, bx); h.set_knob(h.KNOB_QR_STAMP_BY, by)
        L.ttr_debug_set_qr_stamps(buf.data_ptr()); buf.zero_()
        h.qr_factor_pushed(R2, core); torch.cuda.synchronize()
        L.ttr_debug_set_qr_stamps(None)
        st = [x for x in buf.cpu().tolist() if x != 0]
        d = [st[i + 1] - st[i] for i in range(len(st) - 1)]
        print(f"  block ({bx}, {by}): stamps {len(st)} total {st[-1] - st[0] if st else 0}  push {d[0] if d else 0}  rest {d[1:]}")
h.set_knob(h.KNOB_QR_STAMP_BX, 0); h.set_knob(h.KNOB_QR_STAMP_BY, 0); h.set_knob(h.KNOB_QR_PACK, 3)
