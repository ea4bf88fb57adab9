"""Cost of the fused row-Gram epilogue of qr_apply at the metric's shape: apply alone, apply + Gram, rowgram alone (us per launch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tntorch_amd import _hip

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for B in [int(a) for a in sys.argv[1:]] or [2048, 64, 1]:
    torch.manual_seed(0)
    Rm = torch.triu(torch.randn(B, 64, 64, device="cuda")); core = torch.randn(B, 64, 64, 64, device="cuda")
    C = torch.randn(B, 64, 32, device="cuda")
    f = _hip.qr_factor_pushed(Rm, core)
    out = torch.empty(B, 4096, 32, device="cuda")
    t0 = timeit(lambda: _hip.qr_apply(f, C, out=out))
    t1 = timeit(lambda: _hip.qr_apply(f, C, out=out, want_gram=True))
    M = out.reshape(B, 64, 2048)
    t2 = timeit(lambda: _hip.rowgram(M))
    _, G = _hip.qr_apply(f, C, out=out, want_gram=True)
    Gr = _hip.rowgram(M)
    t3 = timeit(lambda: _hip.eigh_trunc(G, _hip.EIG_RAW, False, 0.0, 64, abs_floor=_hip.SOLVER_TRIDIAG), 5)
    t4 = timeit(lambda: _hip.eigh_trunc(Gr, _hip.EIG_RAW, False, 0.0, 64, abs_floor=_hip.SOLVER_TRIDIAG), 5)
    d = (G.sum(1) - Gr.sum(1)).abs().max().item() / Gr.sum(1).abs().max().item()
    print(f"B={B}: apply {t0:.0f} us, apply+gram {t1:.0f} us, rowgram {t2:.0f} us ({Gr.shape[1]} parts); "
          f"eigh on {G.shape[1]} parts {t3:.0f} us vs {t4:.0f} us; gram rel diff {d:.1e}")
