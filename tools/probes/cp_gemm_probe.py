"""The two X-sized products of one CP-ALS sweep at config C4's shape (X = 256^4 fp32, R = 32): X (I^3 x I) times a factor, and
X^T ((I x I^3)^T) times a factor -- launch time and HBM rate.   python tools/probes/cp_gemm_probe.py [I]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tntorch_amd import _hip

I = int(sys.argv[1]) if len(sys.argv) > 1 else 256
R = 32
torch.manual_seed(0)
X = torch.randn(I ** 3, I, device="cuda")
A = torch.randn(I, R, device="cuda")


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


gb = (X.numel() + I ** 3 * R) * 4 / 1e9
t = timeit(lambda: _hip.gemm(X.reshape(1, -1, I), A[None]))
print(f"X (I^3 x I) @ A: {t:.2f} ms = {gb / t:.2f} TB/s")
t = timeit(lambda: _hip.gemm(X.reshape(1, I, -1), A[None], transA=True))
print(f"X^T ((I x I^3)^T) @ A: {t:.2f} ms = {gb / t:.2f} TB/s")
