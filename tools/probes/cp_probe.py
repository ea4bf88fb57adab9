"""BASELINE config C4: CP-ALS R=32 on a dense I^4 fp32 tensor resident in HBM (default I=256: 17.2 GB).
Prints init time, per-sweep time and the algorithmic HBM rate (2 reads of X per sweep).  GPU only."""
import sys, time, math
sys.path.insert(0, "/root/repo")
import torch
from tntorch_amd import _hip as h, _hipops

I = int(sys.argv[1]) if len(sys.argv) > 1 else 256
R, N = 32, 4
torch.manual_seed(0)
dev = "cuda"
fac = [torch.randn(I, R, device=dev) for _ in range(N)]
# X = rank-32 CP + noise, built on device with the library's own GEMM chain (I^3 x R times R x I)
T = fac[0]
for f in fac[1:-1]:
    T = (T[:, None, :] * f[None, :, :]).reshape(-1, R)
X = h.gemm(T[None], fac[-1][None], transB=True)[0].reshape([I] * N)
del T
X += 0.01 * X.std() * torch.randn_like(X)
torch.cuda.synchronize()
print(f"X: {list(X.shape)} {X.numel()*4/1e9:.2f} GB", flush=True)
t0 = time.perf_counter(); A0 = _hipops.cp_hosvd_init(X, R); torch.cuda.synchronize(); t_init = time.perf_counter() - t0
print(f"HOSVD init: {t_init*1e3:.1f} ms", flush=True)
def run(iters):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    A, errs = _hipops.cp_als(X, R, max_iter=iters, tol=-1.0)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, errs

t1, _ = run(1)
h.prof_enable(True)
t6, errs = run(6)
prof = h.prof_collect(); h.prof_enable(False)
sweep = (t6 - t1) / 5
print("errors per sweep:", ["%.5f" % e for e in errs])
print(f"init + 1 sweep: {t1*1e3:.1f} ms ; init + 6 sweeps: {t6*1e3:.1f} ms ; per sweep {sweep*1e3:.2f} ms")
print("per-kind (init + 6 sweeps):", {k: (round(v["ms"], 1), v["launches"]) for k, v in prof.items()})
alg_bytes = 2 * X.numel() * 4
print(f"algorithmic HBM rate of a sweep (2 reads of X = {alg_bytes/1e9:.1f} GB): {alg_bytes/sweep/1e12:.2f} TB/s "
      f"({alg_bytes/sweep/8e12*100:.0f} % of 8 TB/s); reference formulation (N reads of X + KR matrices) would need "
      f"{(N*X.numel()*4)/8e12*1e3:.1f} ms at peak")
if len(sys.argv) > 2:  # CPU baseline: the oracle (same operator sequence as the reference) on the host cores
    sys.path.insert(0, "/root/repo")
    from oracle import tt_oracle
    Xc = X.cpu()
    for th in (8, 32):
        torch.set_num_threads(th)
        init = tt_oracle.cp_hosvd_init(Xc, R)
        t0 = time.perf_counter(); tt_oracle.cp_als(Xc, R, max_iter=2, tol=-1.0, init=init); el = time.perf_counter() - t0
        print(f"CPU oracle ({th} threads): {el/2*1e3:.1f} ms per sweep")
