"""Round-5 probe of the general-rank (decaying-spectrum) sweep: per-kind times + the executed-work census, how many rounds the
orthonormal completion takes (TTR_KNOB_ORTH_ROUNDS A/B), and the pass-2 Jacobi launches on their own (sweeps used, time).
    python tools/probes/decay_r05.py [B] > gpurun_out/decay_probe.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
import tntorch_amd as tn
from tntorch_amd import _hip, _hipops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda", 0)
_hipops.STREAM_CHUNKS_ENABLED = False


def kinds(fn):
    fn(); torch.cuda.synchronize()
    _hip.prof_enable(2)
    fn(); torch.cuda.synchronize()
    p, w = _hip.prof_collect(), _hip.prof_collect_work()
    _hip.prof_enable(False)
    return p, w


for decay in (1.0, 0.5):
    inp = bench.make_decaying_input(B, dev, seed=777, decay=decay)

    def step():
        t = tn.Tensor(inp, batch=True); t.round_tt(rmax=32); return t
    for rounds, jw, ov in ((4, 1, 1), (4, 0, 1), (4, 1, 0), (1, 1, 1)):
        _hip.set_knob(_hip.KNOB_ORTH_ROUNDS, rounds)
        _hip.set_knob(_hip.KNOB_JACOBI_LIVE_WAVE, jw)
        _hip.set_knob(_hip.KNOB_ORTH_V2, ov)
        p, w = kinds(step)
        tot = sum(v["ms"] for v in p.values())
        print(f"decay {decay} B={B} orth rounds<={rounds} jacobi_one_wave={jw} orth_v2={ov}: total {tot:.2f} ms; " +
              ", ".join(f"{k} {v['ms']:.2f}/{v['launches']}" for k, v in p.items() if v["launches"]) +
              f"; orth_fixup: {w['misc']['bytes']:.0f} items with dead rows, {w['misc']['flops']:.0f} rounds"
              f" ({w['misc']['flops'] / max(w['misc']['bytes'], 1):.2f} per item)")
    _hip.set_knob(_hip.KNOB_ORTH_ROUNDS, 4); _hip.set_knob(_hip.KNOB_JACOBI_LIVE_WAVE, 1); _hip.set_knob(_hip.KNOB_ORTH_V2, 2)
    # the pass-2 Jacobi launches alone: capture the arguments of the host loop's calls
    _hipops.SWEEP_C_ENABLED = False
    cap = []
    orig = _hip.eigh_trunc

    def spy(G, eig_mode, use_delta, delta2, rmax, abs_floor=1, **kw):
        if abs_floor == _hip.SOLVER_JACOBI_LIVE:
            cap.append((G.clone(), kw.get("skip_items"), kw.get("sigma_in")))
        return orig(G, eig_mode, use_delta, delta2, rmax, abs_floor=abs_floor, **kw)
    _hip.eigh_trunc = spy
    step(); torch.cuda.synchronize()
    _hip.eigh_trunc = orig
    _hipops.SWEEP_C_ENABLED = True
    for i, (G, skip, sin) in enumerate(cap):
        sw = torch.zeros(G.shape[0], dtype=torch.int32, device=dev)
        orig(G, _hip.EIG_RAW, False, 0.0, 32, abs_floor=_hip.SOLVER_JACOBI_LIVE, sweeps=sw, skip_items=skip, sigma_in=sin)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            orig(G, _hip.EIG_RAW, False, 0.0, 32, abs_floor=_hip.SOLVER_JACOBI_LIVE, skip_items=skip, sigma_in=sin)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        Gs = G.sum(dim=1) if G.dim() == 4 else G
        d = torch.diagonal(Gs, dim1=1, dim2=2)
        live = (d > (64 * 1.19e-7) ** 2 * d.amax(dim=1, keepdim=True)).sum(dim=1).float()
        off = (Gs - torch.diag_embed(d)).norm(dim=(1, 2)) / Gs.norm(dim=(1, 2))
        print(f"  decay {decay} pass-2 Jacobi of bond call {i}: {ms:.3f} ms per launch of {G.shape[0]}; sweeps mean {sw.float().mean():.2f} max {int(sw.max())};"
              f" skipped {int(skip.sum()) if skip is not None else 0}; live prefix mean {live.mean():.1f}; offdiag/||G|| mean {off.mean():.2e}")
