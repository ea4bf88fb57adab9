"""Cycle stamps inside the level-0 qr_apply kernel (library built with -DTTR_QR_WSTAMPS, TTR_LIB_PATH)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tntorch_amd import _hip
L = _hip.lib()
for B in [int(a) for a in sys.argv[1:]] or [1, 2048]:
    torch.manual_seed(0)
    Rm = torch.triu(torch.randn(B, 64, 64, device="cuda")); core = torch.randn(B, 64, 64, 64, device="cuda")
    C = torch.randn(B, 64, 32, device="cuda")
    f = _hip.qr_factor_pushed(Rm, core)
    _hip.qr_apply(f, C); torch.cuda.synchronize()
    buf = torch.zeros(64 + 40 * 8, dtype=torch.int64, device="cuda")
    L.ttr_debug_set_qr_stamps(buf.data_ptr())
    _hip.qr_apply(f, C); torch.cuda.synchronize()
    L.ttr_debug_set_qr_stamps(None)
    st = [x for x in buf.cpu().tolist()[:64] if x]
    print(f"B={B}: total {st[-1]-st[0]}; deltas (C init+first stage | per panel: W, W2, update, restage ... | store):", [st[i+1]-st[i] for i in range(len(st)-1)])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = _hip.qr_apply(f, C)
    e1.record(); torch.cuda.synchronize()
    print(f"   apply {e0.elapsed_time(e1) / 20 * 1e3:.1f} us/call; checksum {out.double().abs().sum().item():.9e} {out.double().flatten()[::99991][:4].tolist()}")
