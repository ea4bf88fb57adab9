"""Selected-eigenpair solver (ttr_tridiag -> ttr_tri_eigsel -> ttr_qr -> ttr_tridiag_back) against the full block-Jacobi
decomposition on Gram matrices of C3's shape: per-stage kernel time.   python tools/probes/eigsel_probe.py [B] [n] [k]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tntorch_amd import _hip, _hipops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
k = int(sys.argv[3]) if len(sys.argv) > 3 else 8
torch.manual_seed(0)
M = torch.randn(B, n, 4096, device="cuda")
G = _hip.gemm(M, M, transB=True)
Gn, _ = _hip.pow2_normalize(G)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def kinds(fn):
    _hip.prof_enable(True); fn(); torch.cuda.synchronize(); p = _hip.prof_collect(); _hip.prof_enable(False)
    return ", ".join(f"{a} {b['ms']:.3f}/{b['launches']}" for a, b in p.items() if b["launches"])


print(f"B={B} n={n} k={k}: eigh_topk {timeit(lambda: _hip.eigh_topk(Gn, k)):.3f} ms ({kinds(lambda: _hip.eigh_topk(Gn, k))})")
print(f"   block-Jacobi full decomposition {timeit(lambda: _hipops._eigh_any(G, _hip.EIG_RAW, False, 0.0, k, _hip.SOLVER_TRIDIAG), 2):.3f} ms")

# per-stage times (direct calls)
L = _hip.lib()
dt = _hip.dtype_code(Gn.dtype)
A = Gn.clone(); d = torch.empty(B, n, device="cuda"); e = torch.empty_like(d); tau = torch.empty_like(d)
lam = torch.empty(B, k, device="cuda"); Z = torch.empty(B, n, k, device="cuda")
sb = L.ttr_eigsel_scratch_bytes(dt, n, B); scratch = torch.empty(sb, dtype=torch.uint8, device="cuda")
st = _hip._stream()
twsb = L.ttr_tridiag_workspace_bytes(dt, n, B); tws = torch.empty(twsb, dtype=torch.uint8, device="cuda")
def s1():
    A.copy_(Gn)
    assert L.ttr_tridiag(dt, n, B, A.data_ptr(), n, n * n, d.data_ptr(), e.data_ptr(), tau.data_ptr(), tws.data_ptr(), twsb, st) == 0
def s2():
    assert L.ttr_tri_eigsel(dt, n, B, k, d.data_ptr(), e.data_ptr(), lam.data_ptr(), Z.data_ptr(), scratch.data_ptr(), sb, st) == 0
def s3():
    assert L.ttr_tridiag_back(dt, n, B, k, A.data_ptr(), n, n * n, tau.data_ptr(), Z.data_ptr(), st) == 0
t_copy = timeit(lambda: A.copy_(Gn))
print(f"   stages: tridiag {timeit(s1) - t_copy:.3f} ms, multisection + twisted vectors {timeit(s2):.3f} ms, back-transformation {timeit(s3):.3f} ms")
