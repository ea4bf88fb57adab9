"""fp64 TSQR block size A/B on config C2's resident batch: 256-row (4-wave) blocks vs 512-row (8-wave) blocks."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tntorch_amd as tn
from tntorch_amd import _hip as h
dev = torch.device("cuda")
B, N, I, r = 256, 10, 128, 32
gen = torch.Generator(device=dev).manual_seed(5)
rr = [1] + [r] * (N - 1) + [1]
cores = []
for k in range(N):
    g = torch.randn((B, rr[k], I, rr[k + 1]), generator=gen, device=dev, dtype=torch.float64)
    c = torch.cat([g, g], dim=-1) if k == 0 else (torch.cat([g, g], dim=-3) if k == N - 1 else
        torch.cat([torch.cat([g, torch.zeros_like(g)], dim=-1), torch.cat([torch.zeros_like(g), g], dim=-1)], dim=-3))
    cores.append(c.contiguous())
t = tn.Tensor(cores, batch=True)
one = tn.Tensor([c[:1].clone() for c in cores], batch=True)
for nw4 in (1, 0, 1, 0):
    h.set_knob(h.KNOB_QR_F64_NW4, nw4)
    out = tn.round_tt(t, rmax=r); torch.cuda.synchronize()
    h.prof_enable(True); t0 = time.perf_counter()
    for _ in range(3): out = tn.round_tt(t, rmax=r)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 3
    prof = h.prof_collect(); h.prof_enable(False)
    o1 = tn.round_tt(one, rmax=r); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): o1 = tn.round_tt(one, rmax=r)
    torch.cuda.synchronize(); el1 = (time.perf_counter() - t0) / 5
    print(json.dumps({"f64_nw4": nw4, "batch256_ms": round(el * 1e3, 2), "single_ms": round(el1 * 1e3, 2),
                      "kinds": {k: round(v["ms"] / 3, 2) for k, v in prof.items() if v["launches"]}}), flush=True)
h.set_knob(h.KNOB_QR_F64_NW4, 1)
