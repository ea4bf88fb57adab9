"""Block-Jacobi driver in isolation: sweeps performed (device control block) and time per eigenproblem for C3's (n = 256,
64 items) and C1's (n = 1024, one item) bond sizes, absolute (pass 1) and relative (pass 2) mode, per inner-sweep setting."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, math
from tntorch_amd import _hip as h, _hipops

dev = torch.device("cuda")
for n, B in ((256, 64), (1024, 1)):
    g = torch.Generator(device=dev).manual_seed(n)
    M = torch.randn(B, 4 * n, n, generator=g, device=dev)
    G1 = h.gemm(M, M, transA=True)
    V1, d1 = _hipops.eigh_block_jacobi(G1)
    Mw = h.gemm(M, V1)
    G2 = h.gemm(Mw, Mw, transA=True)
    b = _hipops._bj_block(n)
    for relative, G in ((False, G1), (True, G2)):
        for inner in (1, 2, 0):
            h.set_knob(h.KNOB_BJ_INNER_SWEEPS, inner)
            Gc = G.clone(); V = torch.eye(n, device=dev).repeat(B, 1, 1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctrl = h.bj_sweeps(Gc, V, b, relative, 0.5 * math.sqrt(n) * torch.finfo(G.dtype).eps, 20)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
            c = ctrl.cpu().tolist()
            off = Gc.clone(); torch.diagonal(off, dim1=1, dim2=2).zero_()
            ratio = float((off.reshape(B, -1).norm(dim=1) / G.reshape(B, -1).norm(dim=1)).max())
            print(json.dumps({"n": n, "B": B, "relative": relative, "inner": inner, "ms": round(el * 1e3, 2), "converged": c[0],
                              "sweeps": c[2], "offdiag_ratio": ratio}), flush=True)
h.set_knob(h.KNOB_BJ_INNER_SWEEPS, 1)
