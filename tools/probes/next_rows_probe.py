#!/usr/bin/env python3
"""Device timings of the SURVEY 8f rows other than CP-ALS (tools/cp_probe.py): Tucker rounding, consumers, producers."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tntorch_amd as tn

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], r

torch.manual_seed(0)
dev = "cuda"
# f4 consumers on the metric's train (64^8, rank 32, fp32): norm / dot = 2 GEMMs per core
g = tn.randn([64] * 8, ranks_tt=32, device=dev)
h = tn.randn([64] * 8, ranks_tt=32, device=dev)
dt, _ = timeit(lambda: tn.norm(g)); print(f"f4 norm(64^8 rank 32): {dt*1e3:.2f} ms")
dt, _ = timeit(lambda: tn.dot(g, h)); print(f"f4 dot(64^8 rank 32): {dt*1e3:.2f} ms")
s = tn.randn([32] * 5, ranks_tt=16, device=dev)
dt, X = timeit(lambda: s.torch()); print(f"f4 torch() 32^5 rank 16 -> 134 MB dense: {dt*1e3:.2f} ms ({X.numel()*4/dt/1e12:.2f} TB/s written)")
# f3 producers: Hadamard product of two rank-32 trains (rank 1024 cores: 64 x 1024 x 1024 fp32 = 268 MB each)
a = tn.randn([64] * 4, ranks_tt=32, device=dev); b = tn.randn([64] * 4, ranks_tt=32, device=dev)
dt, p = timeit(lambda: a * b, reps=2)
nbytes = sum(c.numel() for c in p.cores) * 4
print(f"f3 a*b (64^4, ranks 32 x 32 -> 1024): {dt*1e3:.2f} ms, {nbytes/1e9:.2f} GB written, {nbytes/dt/1e12:.2f} TB/s")
a = tn.randn([64] * 6, ranks_tt=8, device=dev); b = tn.randn([64] * 6, ranks_tt=8, device=dev)
dt, r = timeit(lambda: tn.round_tt(a * b, eps=1e-6), reps=2)
print(f"f3 round_tt(a*b) (64^6, ranks 8 x 8 -> 64, eps 1e-6): {dt*1e3:.2f} ms, ranks {r.ranks_tt.tolist()}")
# f2 Tucker rounding: batch of 256 trains 64^5 rank 16 whose modes have Tucker rank 12 (rmax mode)
cores = [torch.randn(256, 1 if k == 0 else 16, 12, 1 if k == 4 else 16, device=dev) for k in range(5)]
Us = [torch.randn(256, 64, 12, device=dev) for _ in range(5)]
t = tn.Tensor(cores, Us=Us, batch=True)
full = tn.Tensor(t._denorm(t._absorbed4()), batch=True)          # TT cores 64-wide, Tucker structure hidden inside
dt, r = timeit(lambda: tn.round_tucker(full, rmax=12), reps=2)
print(f"f2 round_tucker(rmax=12), 256 x 64^5 rank 16: {dt*1e3:.2f} ms ({dt/256*1e3:.3f} ms/tensor), Tucker ranks {r.ranks_tucker.tolist()}")
one = tn.Tensor([c[0] for c in full.cores])
dt, r = timeit(lambda: tn.round(one, eps=1e-5), reps=2)
print(f"f2 round(eps=1e-5), one 64^5 rank-16 train: {dt*1e3:.2f} ms, TT ranks {r.ranks_tt.tolist()}, Tucker ranks {r.ranks_tucker.tolist()}")
