import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tntorch_amd import _hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
A = torch.randn(B, 4096, 64, device="cuda")
Rm = torch.randn(B, 64, 64, device="cuda"); core = torch.randn(B, 64, 64, 64, device="cuda")
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
L = _hip.lib()
for name, fn in (("plain", lambda: _hip.qr_factor(A)), ("pushed", lambda: _hip.qr_factor_pushed(Rm, core))):
    fn(); torch.cuda.synchronize()
    L.ttr_debug_set_qr_stamps(buf.data_ptr()); buf.zero_()
    fn(); torch.cuda.synchronize()
    L.ttr_debug_set_qr_stamps(None)
    st = buf.cpu().tolist(); st = [x for x in st if x != 0]
    d = [st[i+1]-st[i] for i in range(len(st)-1)]
    print(name, "stamps", len(st), "total", st[-1]-st[0])
    print("  deltas (load | per panel: transpose, 16 steps, S+T, W/update):", d)
