"""Launch times of the tall-matrix kernels of a dense TT-SVD's first steps (ttr_colgram, ttr_colproject) at C3's and C1's
shapes, with the HBM rates their algorithmic bytes give.   python tools/probes/col_kernel_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tntorch_amd import _hip

torch.manual_seed(0)


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


for B, rows, n, ro in ((64, 1 << 20, 32, 8), (1, 1 << 27, 64, 16)):
    M = torch.randn(B, rows, n, device="cuda")
    V = torch.linalg.qr(torch.randn(B, n, n, device="cuda"))[0].contiguous()
    sig = torch.rand(B, n, device="cuda") + 0.5
    gb = M.numel() * 4 / 1e9
    t = timeit(lambda: _hip.colgram(M))
    print(f"B={B} rows={rows} n={n}: colgram {t:.2f} ms = {gb / t:.2f} TB/s", end="; ")
    t = timeit(lambda: _hip.colgram(M, V))
    print(f"rotated colgram {t:.2f} ms = {gb / t:.2f} TB/s", end="; ")
    t = timeit(lambda: _hip.colproject(M, V, V, sig, ro, False))
    print(f"colproject -> {ro} {t:.2f} ms = {gb * (1 + ro / n) / t:.2f} TB/s")
    del M
