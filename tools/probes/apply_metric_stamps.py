"""Cycle stamps inside the level-0 qr_apply kernel on the METRIC's packed factorisation (library built with -DTTR_QR_WSTAMPS,
TTR_LIB_PATH): block (0, 0)'s deltas [prologue + first stage | per live panel: W partials, partial sum, W2 + update | store]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from tntorch_amd import _hip as h

L = h.lib()
for B in [int(a) for a in sys.argv[1:]] or [64, 4096]:
    dev = torch.device("cuda", 0)
    inp = bench.make_input(B, dev, seed=1)
    c0 = inp[0]
    R = h.qr_factor(c0.reshape(B, -1, c0.shape[-1])).R
    Rn, _ = h.pow2_normalize(R)
    f = h.qr_factor_pushed(Rn, inp[1]); R2, _ = h.pow2_normalize(f.R)
    f2 = h.qr_factor_pushed(R2, inp[2])
    C = torch.randn(B, 64, 32, device="cuda")
    h.qr_apply(f2, C); torch.cuda.synchronize()
    buf = torch.zeros(64 + 40 * 8, dtype=torch.int64, device="cuda")
    L.ttr_debug_set_qr_stamps(buf.data_ptr())
    h.qr_apply(f2, C); torch.cuda.synchronize()
    L.ttr_debug_set_qr_stamps(None)
    st = [x for x in buf.cpu().tolist()[:64] if x]
    print(f"B={B}: stamps {len(st)} total {st[-1]-st[0] if st else 0}; deltas:", [st[i+1]-st[i] for i in range(len(st)-1)])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = h.qr_apply(f2, C)
    e1.record(); torch.cuda.synchronize()
    print(f"   apply (both levels) {e0.elapsed_time(e1) / 10 * 1e3:.1f} us/call")
