"""Launch times of the R2L sweep's streaming kernels at the metric's shapes (B items of a 64 x 2048 right unfolding, fp32):
ttr_rowgram (reads M), ttr_rotgram (reads M, MFMA-bound), ttr_project to rank 32 (reads M, writes half of it), with the
HBM rates their algorithmic bytes give.   python tools/probes/sweep_kernel_probe.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tntorch_amd import _hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.manual_seed(0)
M = torch.randn(B, 64, 2048, device="cuda")
V = torch.linalg.qr(torch.randn(B, 64, 64, device="cuda"))[0].contiguous()
sig = torch.rand(B, 64, device="cuda") + 0.5
out = torch.empty(B, 32, 2048, device="cuda")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


mb = M.numel() * 4
for name, fn, nbytes in (
    ("rowgram", lambda: _hip.rowgram(M), mb),
    ("rotgram", lambda: _hip.rowgram(M, V), mb),
    ("project V1 V2 -> 32", lambda: _hip.project(M, V, V, sig, 32, True, out=out), mb + out.numel() * 4),
    ("project V2 -> 32", lambda: _hip.project(M, None, V, sig, 32, True, out=out), mb + out.numel() * 4),
):
    us = timeit(fn)
    print(f"B={B} {name}: {us:.1f} us, {nbytes / us / 1e6:.2f} TB/s algorithmic")
