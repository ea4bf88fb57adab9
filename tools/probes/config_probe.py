#!/usr/bin/env python3
"""Time the non-headline BASELINE configs (C2, C3 unit, C1 proxy) through the public API on one GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tntorch_amd as tn

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts)//2], r

torch.manual_seed(0)
# C2: round_tt(eps=1e-4) of rank-64 TT, 10 cores x 128, fp64
g = tn.randn([128]*10, ranks_tt=32, dtype=torch.float64, device="cuda")
t = g + g
for alg in ("svd", "eig"):
    dt, r = timeit(lambda: tn.round_tt(t, eps=1e-4, algorithm=alg))
    print(f"C2 fp64 round_tt(eps=1e-4) alg={alg}: {dt*1e3:.1f} ms  ranks {r.ranks_tt.tolist()}")
# C3 unit: dense 32^5 -> rmax 8, fp32, batch
for B in (1, 8, 64):
    X = torch.randn(B, 32, 32, 32, 32, 32, device="cuda")
    for alg in ("svd", "eig"):
        dt, r = timeit(lambda: tn.Tensor(X, ranks_tt=8, batch=True, algorithm=alg), reps=2)
        print(f"C3 dense 32^5 -> 8, B={B}, alg={alg}: {dt*1e3:.1f} ms ({dt/B*1e3:.1f} ms/tensor) ranks {r.ranks_tt.tolist()}")
# C1 proxy: dense 64^4 -> 16 fp32
X = torch.randn(64, 64, 64, 64, device="cuda")
for alg in ("svd", "eig"):
    dt, r = timeit(lambda: tn.Tensor(X, ranks_tt=16, algorithm=alg), reps=2)
    print(f"C1 proxy dense 64^4 -> 16 alg={alg}: {dt*1e3:.1f} ms ranks {r.ranks_tt.tolist()} relerr {tn.relative_error(X, r).item():.4f}")
del X
torch.cuda.empty_cache()
# C1 larger proxy: dense 64^5 (4.3 GB) -> 16 fp32, low rank + noise so that the truncation is meaningful
if len(sys.argv) > 1 and sys.argv[1] == "big":
    cores = [torch.randn(1 if k == 0 else 16, 64, 1 if k == 4 else 16, device="cuda") / 8 for k in range(5)]
    X = tn.Tensor(cores).torch()
    X += 1e-3 * X.std() * torch.randn_like(X)
    for alg in ("svd", "eig"):
        dt, r = timeit(lambda: tn.Tensor(X, ranks_tt=16, algorithm=alg), reps=1)
        print(f"C1 proxy dense 64^5 -> 16 alg={alg}: {dt*1e3:.1f} ms ranks {r.ranks_tt.tolist()} relerr {tn.relative_error(X, r).item():.2e}")
