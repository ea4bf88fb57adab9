"""How much work does pass 2 (Jacobi on the re-computed Gram matrix) do?  Sweep histogram for rank-deficient and
full-rank 64 x 2048 unfoldings (metric shapes)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from tntorch_amd import _hip as h
B = 2048
torch.manual_seed(0)
for name, M in (("rank 32 (g+g like)", torch.randn(B, 64, 32, device="cuda") @ torch.randn(B, 32, 2048, device="cuda")),
                ("full rank randn", torch.randn(B, 64, 2048, device="cuda")),
                ("graded 2^-j", torch.randn(B, 64, 2048, device="cuda") * (0.5 ** torch.arange(64, device="cuda"))[None, :, None])):
    G = h.gemm(M, M, transB=True)
    V1, _, _ = h.eigh_trunc(G, h.EIG_RAW, False, 0.0, 64, abs_floor=h.SOLVER_TRIDIAG)
    Mw = h.gemm(V1, M, transA=True)
    G2 = h.gemm(Mw, Mw, transB=True)
    sw = torch.zeros(B, dtype=torch.int32, device="cuda")
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        V2, s, info = h.eigh_trunc(G2, h.EIG_RAW, False, 0.0, 32, abs_floor=h.SOLVER_JACOBI_ABS, sweeps=sw)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
    hist = torch.bincount(sw.cpu(), minlength=6).tolist()
    print(f"{name}: pass-2 kernel {el*1e3:.3f} ms; sweeps histogram {hist}")

# ---- the real metric input: histogram per eigensolver call inside round_tt
import bench, tntorch_amd as tn
from tntorch_amd import _hipops
_hipops.STREAM_CHUNKS_ENABLED = False
inp = bench.make_input(512, torch.device("cuda", 0), 1234)
orig = h.eigh_trunc
log = []
def spy(G, eig_mode, use_delta, delta2, rmax, abs_floor=1, sweeps=None):
    sw = torch.zeros(G.shape[0], dtype=torch.int32, device=G.device)
    out = orig(G, eig_mode, use_delta, delta2, rmax, abs_floor=abs_floor, sweeps=sw)
    log.append((abs_floor, G.shape[1], sw))
    return out
h.eigh_trunc = spy
t = tn.Tensor(inp, batch=True); t.round_tt(rmax=32)
torch.cuda.synchronize()
for (solver, n, sw) in log:
    s = sw.cpu()
    print(f"solver {solver} n={n}: mean {s.float().mean():.2f} max {int(s.max())} hist(0..4) {torch.bincount(s.clamp(max=5), minlength=6).tolist()}")
