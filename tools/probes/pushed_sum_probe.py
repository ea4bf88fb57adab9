import sys; sys.path.insert(0, "/root/repo")
import torch
from tntorch_amd import _hip as h
B=1024
torch.manual_seed(0)
Rm = torch.triu(torch.randn(B,64,64,device="cuda"))
a = torch.randn(B,32,64,32,device="cuda"); b = torch.randn(B,32,64,32,device="cuda")
za, zb = torch.zeros(B,32,64,32,device="cuda"), torch.zeros(B,32,64,32,device="cuda")
core = torch.cat([torch.cat([a, za], dim=-1), torch.cat([zb, b], dim=-1)], dim=1).contiguous()
def t(fn):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); fn(); fn(); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/3
print("pushed_sum", round(t(lambda: h.qr_factor_pushed_sum(Rm, a, b)),3), "ms; pushed (materialised core)", round(t(lambda: h.qr_factor_pushed(Rm, core)),3), "ms")
