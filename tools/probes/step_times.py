"""Per-step wall times of the metric workload in a fresh process (are the first steps slower: clock ramp, first-touch of the
workspaces?).  python tools/probes/step_times.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import tntorch_amd as tn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
inp = bench.make_input(2048, dev, seed=1234)
ts = []
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = tn.Tensor(inp, batch=True); t.round_tt(rmax=32)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("ms per step:", " ".join(f"{x:.1f}" for x in ts))
print("reserved GB", torch.cuda.memory_reserved() / 2**30)
