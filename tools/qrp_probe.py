import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tntorch_amd import _hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
Rm = torch.randn(B, 64, 64, device="cuda")
core = torch.randn(B, 64, 64, 64, device="cuda")
A = torch.randn(B, 4096, 64, device="cuda")
for _ in range(2):
    f = _hip.qr_factor_pushed(Rm, core)
    f2 = _hip.qr_factor(A)
torch.cuda.synchronize()
