"""Decaying-spectrum variant of the metric input (bond sigma_j ~ 2^(-decay j)): per-kind kernel time of one step and a
small-batch check against the float64 oracle.  usage: python tools/decay_probe.py [decay] [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import tntorch_amd as tn  # noqa: E402
from tntorch_amd import _hip, _hipops  # noqa: E402

decay = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device("cuda", 0)
inp = bench.make_decaying_input(B, dev, seed=777, decay=decay)


def step():
    t = tn.Tensor(inp, batch=True)
    t.round_tt(rmax=32)
    return t


sweep_log = []
if os.environ.get("PROBE_SWEEPS") == "1":  # Jacobi sweeps of every pass-2 launch (ttr_eigh_trunc's `sweeps` output)
    orig = _hip.eigh_trunc

    def spy(G, eig_mode, use_delta, delta2, rmax, abs_floor=1, sweeps=None, **kw):
        if abs_floor == _hip.SOLVER_JACOBI_LIVE and sweeps is None:
            sweeps = torch.zeros(G.shape[0], dtype=torch.int32, device=G.device)
            sweep_log.append(sweeps)
        return orig(G, eig_mode, use_delta, delta2, rmax, abs_floor=abs_floor, sweeps=sweeps, **kw)

    _hip.eigh_trunc = spy
for _ in range(2):
    out = step()
torch.cuda.synchronize()
_hipops.STREAM_CHUNKS_ENABLED = False
step()   # (untimed: the kernels' first run on the single stream's queue -- scratch sizing, see bench.py)
torch.cuda.synchronize()
_hip.prof_enable(True)
out = step()
torch.cuda.synchronize()
pk = _hip.prof_collect()
_hip.prof_enable(False)
print("decay", decay, "B", B, {k: (round(v["ms"], 3), v["launches"]) for k, v in pk.items() if v["launches"]})
print("ranks", out.ranks_tt.tolist(), "norm core0", float(out.cores[0][0].norm()), "finite", bool(torch.isfinite(out.cores[0]).all()))
chk = bench.decaying_parity(inp, out, 0)
print(chk)
if sweep_log:
    last = sweep_log[-7:]
    print("jacobi sweeps per bond (min / mean / max over the batch):", [(int(x.min()), round(float(x.float().mean()), 2), int(x.max())) for x in last])
