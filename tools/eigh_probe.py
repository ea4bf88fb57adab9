import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tntorch_amd import _hip
B = 512
torch.manual_seed(0)
M = torch.randn(B, 64, 2048, device="cuda")
G = _hip.gemm(M, M, transB=True)
sw = torch.zeros(B, dtype=torch.int32, device="cuda")
for _ in range(2):
    V, s, info = _hip.eigh_trunc(G, _hip.EIG_RAW, False, 0.0, 32, sweeps=sw)
torch.cuda.synchronize()
print("sweeps", sw.min().item(), sw.max().item())
