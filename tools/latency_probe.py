"""Small-batch latency of round_tt (launch-bound regime) on the metric's shape: batch of 1 .. 64, the reference's own NON-batch
call signatures (`t.round_tt(rmax=32)`, `t.round_tt(eps=1e-4)`) and one BASELINE-C2 train -- through ttr_round_tt (the whole sweep
behind one library call) and through the host loop over the per-kernel entries (TTR_SWEEP_C=0), side by side.
    python tools/latency_probe.py > gpurun_out/latency.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import tntorch_amd as tn
from tntorch_amd import _hip, _hipops


def timeit(fn, reps=30):
    fn(); fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


dev = torch.device("cuda", 0)
for on in (True, False):
    _hipops.SWEEP_C_ENABLED = on
    tag = "ttr_round_tt (one call)" if on else "host loop (one call per kernel)"
    print(f"---- {tag}")
    for B in (1, 4, 16, 64):
        inp = bench.make_input(B, dev, 7)

        def f():
            t = tn.Tensor(inp, batch=True); t.round_tt(rmax=32); return t
        f(); torch.cuda.synchronize()   # (untimed: a kernel's first launch in a process loads its code object and sizes the queue's scratch -- the 7 ms of
        # "event-timed kernel time" at B = 1 in r05_latency.txt / the first r06 runs were that, not kernel time)
        _hip.prof_enable(True); f(); torch.cuda.synchronize(); prof = _hip.prof_collect(); _hip.prof_enable(False)
        nl = sum(v["launches"] for v in prof.values()); kms = sum(v["ms"] for v in prof.values())
        t0 = time.perf_counter(); f(); host = (time.perf_counter() - t0) * 1e3; torch.cuda.synchronize()
        print(f"batch=True B={B}: {timeit(f):.3f} ms per call (host enqueue {host:.3f} ms; {nl} library launches, {kms:.3f} ms of "
              "event-timed kernel time; per kind ms/launches: "
              + ", ".join(f"{k} {v['ms']:.3f}/{v['launches']}" for k, v in prof.items() if v["launches"]) + ")")
    one = [c[0] for c in bench.make_input(1, dev, 7)]
    for mode in ("auto", "1"):
        os.environ["TTR_EPS_DEFERRED"] = mode

        def g():
            t = tn.Tensor(one); t.round_tt(rmax=32); return t

        def h():
            t = tn.Tensor(one); t.round_tt(eps=1e-4); return t
        print(f"non-batch rmax=32 (TTR_EPS_DEFERRED={mode}): {timeit(g):.3f} ms per call")
        print(f"non-batch eps=1e-4 (TTR_EPS_DEFERRED={mode}): {timeit(h):.3f} ms per call, ranks {list(h().ranks_tt)}")
    torch.manual_seed(0)
    g2 = tn.randn([128] * 10, ranks_tt=32, dtype=torch.float64)
    c2 = [c.to(dev) for c in (g2 + g2).cores]
    for mode in ("auto", "1"):
        os.environ["TTR_EPS_DEFERRED"] = mode

        def c2f():
            t = tn.Tensor(c2); t.round_tt(eps=1e-4); return t
        print(f"C2 single train fp64 eps=1e-4 (TTR_EPS_DEFERRED={mode}): {timeit(c2f, 10):.3f} ms per call, ranks {list(c2f().ranks_tt)}")
    os.environ["TTR_EPS_DEFERRED"] = "auto"
