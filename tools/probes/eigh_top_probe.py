"""ttr_eigh_top (pass 1 of a batch-mode bond: r largest eigenpairs, or the full QL decomposition for the items the top-r path
declines) against torch.linalg.eigh in fp64, and against the plain QL kernel's time."""
import sys
import time
import torch
sys.path.insert(0, ".")
from tntorch_amd import _hip

dev = torch.device("cuda:0")
B = 64


def gram(n, cols, dtype, spectrum=None, seed=0, batch=B):
    """Random Gram matrices M M^T (M [n, cols], on the device) or Q diag(spectrum^2) Q^T (Q from a CPU QR), exact in fp64 first."""
    if spectrum is None:
        M = torch.randn(batch, n, cols, generator=torch.Generator(device=dev).manual_seed(seed), dtype=torch.float64, device=dev)
        return (M @ M.transpose(1, 2)).to(dtype)
    g = torch.Generator(device="cpu").manual_seed(seed)
    Q, _ = torch.linalg.qr(torch.randn(batch, n, n, generator=g, dtype=torch.float64))
    lam = torch.as_tensor(spectrum, dtype=torch.float64) ** 2
    return ((Q * lam) @ Q.transpose(1, 2)).to(dtype).to(dev)


def errors(G, V, sig, cols):
    Gd, Vd, s = G.double(), V[:, :, :cols].double(), sig[:, :cols].double()
    lam = torch.linalg.eigvalsh(Gd.cpu()).flip(-1)[:, :cols].to(dev)
    eps = torch.finfo(G.dtype).eps
    orth = (Vd.transpose(1, 2) @ Vd - torch.eye(cols, device=dev, dtype=torch.float64)).abs().amax().item()
    res = ((Gd @ Vd - Vd * (s * s)[:, None, :]).norm(dim=(1, 2)) / Gd.norm(dim=(1, 2)).clamp_min(1e-300)).amax().item()
    serr = ((s - lam.clamp_min(0).sqrt()).abs().amax(dim=1) / s[:, 0].clamp_min(1e-300)).amax().item()
    return f"orth {orth / eps:.1f} eps  resid {res / eps:.1f} eps  sigma {serr / eps:.1f} eps"


def check(name, G, r, thr=0.125, expect=None):
    n = G.shape[-1]
    V, sig, info, flat = _hip.eigh_top(G, r, thr)
    torch.cuda.synchronize()
    nfl = int(flat.sum().item())
    msg = f"{name}: n={n} r={r} {str(G.dtype)[6:]} top {nfl}/{G.shape[0]}"
    if expect is not None and nfl != expect:
        msg += f"  UNEXPECTED (wanted {expect})"
    it, iq = flat.nonzero()[:, 0], (flat == 0).nonzero()[:, 0]
    if len(it):
        msg += "  | top: " + errors(G[it], V[it], sig[it], r)
        msg += f" tail {V[it][:, :, r:].abs().amax().item()} {sig[it][:, r:].abs().amax().item()} info {info[it].unique().tolist()}"
    if len(iq) and float(G[iq].abs().amax()) > 0:
        msg += "  | ql: " + errors(G[iq], V[iq], sig[iq], n) + f" info {info[iq].unique().tolist()}"
    print(msg, flush=True)


T0 = time.time()
for dt in (torch.float32, torch.float64):
    e = torch.finfo(dt).eps
    check("random", gram(64, 4096, dt, batch=512), 32)
    check("random r=16", gram(64, 4096, dt), 16, expect=B)
    check("random r=20", gram(64, 512, dt), 20)
    check("random r=3", gram(64, 512, dt), 3)
    check("random r=1", gram(64, 512, dt), 1)
    check("n=48 r=24", gram(48, 512, dt), 24)
    check("n=40 r=32", gram(40, 512, dt), 32)
    check("lowrank32", gram(64, 0, dt, spectrum=[1.0 + 0.05 * (i % 7) + 0.01 * i for i in range(32)] + [0.0] * 32), 32)
    check("lowrank32+noise", gram(64, 0, dt, spectrum=[2.0 - 0.03 * i for i in range(32)] + [1e-3] * 32), 32)
    check("graded", gram(64, 0, dt, spectrum=[0.5 ** i for i in range(64)]), 32, expect=0)
    check("pairs", gram(64, 0, dt, spectrum=[1.0, 1.0] + [0.9 - 0.01 * i for i in range(62)]), 32, expect=0)
    check("pairs at 300 eps", gram(64, 0, dt, spectrum=[(1.0 - 0.02 * (i // 2)) * (1.0 + 300 * e * (i % 2)) for i in range(64)]), 32)
    check("pairs at 1000 eps", gram(64, 0, dt, spectrum=[(1.0 - 0.02 * (i // 2)) * (1.0 + 1000 * e * (i % 2)) for i in range(64)]), 32)
    check("triples at 600 eps", gram(64, 0, dt, spectrum=[(1.0 - 0.03 * (i // 3)) * (1.0 + 600 * e * (i % 3)) for i in range(64)]), 32)
    check("identity", torch.eye(64, dtype=dt, device=dev).repeat(8, 1, 1), 32, expect=0)
    check("zero", torch.zeros(8, 64, 64, dtype=dt, device=dev), 32, expect=0)
    print(f"[{time.time() - T0:.1f} s]", flush=True)

for dt in (torch.float32, torch.float64):
    for batch in (1, 64, 4096):
        G = gram(64, 256, dt, batch=batch)
        for name, fn in (("top32", lambda: _hip.eigh_top(G, 32, 0.125)), ("top16", lambda: _hip.eigh_top(G, 16, 0.125)),
                         ("ql", lambda: _hip.eigh_trunc(G, _hip.EIG_RAW, False, 0.0, 64, abs_floor=_hip.SOLVER_TRIDIAG))):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            print(f"{str(dt)[6:]} B={batch} {name}: {e0.elapsed_time(e1) / 5 * 1e3:.0f} us", flush=True)
