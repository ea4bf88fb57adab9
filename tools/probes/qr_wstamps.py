"""Per-wave cycle stamps of the first panel's pair phases of the level-0 fused push+factor kernel (needs a library
built with -DTTR_QR_WSTAMPS, pointed to by TTR_LIB_PATH).  Prints, for every phase, what each wave did between the
phase entry, the pre-barrier point, the barrier release and the end of its apply section."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tntorch_amd import _hip

L = _hip.lib()
for B in [int(a) for a in sys.argv[1:]] or [1, 2048]:
    torch.manual_seed(0)
    Rm = torch.triu(torch.randn(B, 64, 64, device="cuda")); core = torch.randn(B, 64, 64, 64, device="cuda")
    buf = torch.zeros(64 + 40 * 8, dtype=torch.int64, device="cuda")
    _hip.qr_factor_pushed(Rm, core); torch.cuda.synchronize()
    L.ttr_debug_set_qr_stamps(buf.data_ptr()); buf.zero_()
    _hip.qr_factor_pushed(Rm, core); torch.cuda.synchronize()
    L.ttr_debug_set_qr_stamps(None)
    st = buf.cpu().tolist()
    coarse = [x for x in st[:64] if x != 0]
    print(f"B={B}: coarse total {coarse[-1]-coarse[0]}, deltas {[coarse[i+1]-coarse[i] for i in range(len(coarse)-1)]}")
    t0 = min(x for x in st[64:] if x != 0)
    for ph in range(8):
        row = []
        for w in range(8):
            e = [st[64 + 40 * w + 4 * ph + k] for k in range(4)]
            row.append("/".join(str(x - t0) if x else "-" for x in e))
        print(f"  phase {ph}: " + "  ".join(f"w{w}:{r}" for w, r in enumerate(row)))
