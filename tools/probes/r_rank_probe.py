"""Numerical rank structure of the R factors of the L2R sweep on the metric input (t = g + g: unfoldings of rank 32 in 64 columns):
||R[32:, :]||_F / ||R||_F per core, max over a small batch -- the quantity the rank-revealing shortcuts of ttr_qr.hip test."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tntorch_amd import _hip  # noqa: E402

B = 64
dev = torch.device("cuda", 0)
for name, inp in (("g+g", bench.make_input(B, dev, seed=1)), ("decay0.5", bench.make_decaying_input(B, dev, 3, 0.5)),
                  ("decay1.0", bench.make_decaying_input(B, dev, 3, 1.0))):
    c0 = inp[0]
    f = _hip.qr_factor(c0.reshape(B, -1, c0.shape[-1]))
    R = f.R
    out = []
    for mu in range(1, len(inp) - 1):
        Rn, _ = _hip.pow2_normalize(R)
        lo = Rn[:, 32:, :].reshape(B, -1).norm(dim=1) / Rn.reshape(B, -1).norm(dim=1)
        out.append((float(lo.min()), float(lo.max())))
        f = _hip.qr_factor_pushed(Rn, inp[mu])
        R = f.R
    print(name, ["%.2e..%.2e" % o for o in out])
