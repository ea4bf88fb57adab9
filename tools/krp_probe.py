"""Time ttr_krp_contract / the CP-ALS GEMM shapes in isolation.  GPU only."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from tntorch_amd import _hip as h

def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

for (P, J, Q, R) in [(65536, 256, 1, 32), (1, 256, 65536, 32), (256, 256, 1, 32), (1, 256, 256, 32), (4096, 256, 16, 32)]:
    T = torch.randn(P, J, Q, R, device="cuda"); B = torch.randn(J, R, device="cuda")
    dt = timeit(lambda: h.krp_contract(T, B))
    print(f"krp P={P} J={J} Q={Q} R={R}: {dt*1e3:8.2f} ms  {T.numel()*4/dt/1e12:.2f} TB/s", flush=True)
I = 256
X = torch.randn(I**3, I, device="cuda"); A = torch.randn(I, 32, device="cuda")
dt = timeit(lambda: h.gemm(X[None], A[None])); print(f"gemm X[I^3,I] @ A: {dt*1e3:.2f} ms {X.numel()*4/dt/1e12:.2f} TB/s")
X0 = X.reshape(I, -1)
dt = timeit(lambda: h.gemm(X0[None], A[None], transA=True)); print(f"gemm X_(0)^T @ A: {dt*1e3:.2f} ms {X.numel()*4/dt/1e12:.2f} TB/s")
dt = timeit(lambda: h.gemm(X0[None], X0[None], transB=True), 1); print(f"gram mode 0: {dt*1e3:.2f} ms {2*I*I*I**3*I/dt/1e12:.1f} TF")
dt = timeit(lambda: h.gemm(X[None], X[None], transA=True), 1); print(f"gram mode N-1: {dt*1e3:.2f} ms {2*I*I*I**3*I/dt/1e12:.1f} TF")
