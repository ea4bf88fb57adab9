"""Two launches of the level-0 fused push+factor kernel at the metric shape (B = 2048) -- for `rocprofv3 --pmc` passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tntorch_amd import _hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.manual_seed(0)
Rm = torch.triu(torch.randn(B, 64, 64, device="cuda")); core = torch.randn(B, 64, 64, 64, device="cuda")
for _ in range(2):
    f = _hip.qr_factor_pushed(Rm, core)
torch.cuda.synchronize()
