"""Where the time of the CP HOSVD init (tensor.py:228-277) of BASELINE C4 goes: per mode, Gram / eigensolver, cold and warm."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tntorch_amd import _hip, _hipops  # noqa: E402

I, R = 256, 32
dev = torch.device("cuda", 0)
X = torch.randn(I, I, I, I, device=dev)
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, r
for rep in range(2):
    ms, _ = t(lambda: _hipops.cp_hosvd_init(X, R))
    print("init total", rep, round(ms, 2))
for n in range(4):
    if n == 3:
        ms_c, A = 0.0, X.reshape(1, -1, I)
        ms_g, G = t(lambda: _hip.gemm(A, A, transA=True))
    else:
        ms_c, A = t(lambda: (X if n == 0 else X.movedim(n, 0).contiguous()).reshape(1, I, -1))
        ms_g, G = t(lambda: _hip.gemm(A, A, transB=True))
    Gn, _ = _hip.pow2_normalize(G)
    ms_e, _ = t(lambda: _hip.eigh_topk(Gn, R))
    print("mode", n, "copy", round(ms_c, 2), "gram", round(ms_g, 2), "eigh_topk", round(ms_e, 2))
