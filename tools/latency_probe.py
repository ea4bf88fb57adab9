"""Single-tensor latency of round_tt (launch-bound regime): metric shape, batch of 1 (no host syncs) and
non-batch (one rank readback per bond)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tntorch_amd as tn
from tntorch_amd import _hip
import bench

def timeit(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for B in (1, 4, 16, 64):
    inp = bench.make_input(B, torch.device("cuda", 0), 7)
    def f():
        t = tn.Tensor(inp, batch=True); t.round_tt(rmax=32); return t
    _hip.prof_enable(True); f(); torch.cuda.synchronize(); prof = _hip.prof_collect(); _hip.prof_enable(False)
    nl = sum(v["launches"] for v in prof.values()); kms = sum(v["ms"] for v in prof.values())
    print(f"batch=True B={B}: {timeit(f):.3f} ms per call ({nl} library launches, {kms:.3f} ms of kernel time; per kind ms/launches: "
          + ", ".join(f"{k} {v['ms']:.3f}/{v['launches']}" for k, v in prof.items() if v["launches"]) + ")")
one = [c[0] for c in bench.make_input(1, torch.device('cuda', 0), 7)]
def g():
    t = tn.Tensor(one); t.round_tt(rmax=32); return t
print(f"non-batch rmax=32: {timeit(g):.3f} ms per call")
def h():
    t = tn.Tensor(one); t.round_tt(eps=1e-4); return t
print(f"non-batch eps=1e-4: {timeit(h):.3f} ms per call")
