import sys, time, torch
sys.path.insert(0, '.')
import oracle
from tntorch_amd import _hip, _hipops
B = 64
torch.manual_seed(1)
g = oracle.tt_randn([64]*8, 32, dtype=torch.float32, batch_size=B)
inp = oracle.tt_add(g, g, batch=True)
c = [x.cuda() for x in inp]
for mu in range(7):
    _hipops.left_orthogonalize(c, mu)
M = c[7].reshape(B, 64, 64)
def run(G, mode, absf):
    sw = torch.zeros(G.shape[0], dtype=torch.int32, device='cuda')
    torch.cuda.synchronize(); t0=time.perf_counter()
    V, s, info = _hip.eigh_trunc(G, mode, False, 0.0, 64, abs_floor=absf, sweeps=sw)
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    return V, s, sw, dt
G = _hip.gemm(M, M, transB=True)
for rep in range(2):
    V1, s1, sw, dt = run(G, _hip.EIG_RAW, True)
    print('pass1 abs_floor sweeps', sw.min().item(), sw.max().item(), f'{dt*1e3:.2f} ms')
V1b, _, sw, dt = run(G, _hip.EIG_RAW, False); print('pass1 NO floor sweeps', sw.min().item(), sw.max().item(), f'{dt*1e3:.2f} ms')
Mw = _hip.gemm(V1, M, transA=True)
G1 = _hip.gemm(Mw, Mw, transB=True)
V2, s2, sw, dt = run(G1, _hip.EIG_RAW, False); print('pass2 rel-only sweeps', sw.min().item(), sw.max().item(), f'{dt*1e3:.2f} ms')
V2, s2b, sw, dt = run(G1, _hip.EIG_RAW, True); print('pass2 abs floor sweeps', sw.min().item(), sw.max().item(), f'{dt*1e3:.2f} ms')
print('sigma rel-only', s2[0,:2].tolist(), s2[0,30:34].tolist()); print('sigma absfloor', s2b[0,:2].tolist(), s2b[0,30:34].tolist())
# a full-rank well-conditioned case
Mr = torch.randn(B, 64, 2048, device='cuda')
Gr = _hip.gemm(Mr, Mr, transB=True)
_, _, sw, dt = run(Gr, _hip.EIG_RAW, True); print('randn gram sweeps', sw.min().item(), sw.max().item(), f'{dt*1e3:.2f} ms')
Gd = Gr.double()
_, _, sw, dt = run(Gd, _hip.EIG_RAW, True); print('randn gram f64 sweeps', sw.min().item(), sw.max().item(), f'{dt*1e3:.2f} ms')
