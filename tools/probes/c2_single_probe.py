"""Config C2, ONE fp64 train (round_tt eps=1e-4; 10 cores x mode 128, rank 64 -> 32): wall time per call and the library's
per-kind kernel time / launch counts of one call.   python tools/probes/c2_single_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tools/
import torch
import tntorch_amd as tn
import oracle
import bench_configs as bc

dev = torch.device("cuda", 0)
torch.manual_seed(0)
g = oracle.tt_randn([128] * 10, 32, dtype=torch.float64)
inp = oracle.tt_add(g, g)
t_in = tn.Tensor([c.to(dev) for c in inp])
for kw in ({"eps": 1e-4}, {"eps": 1e-4, "rmax": 64}, {"rmax": 32}):
    f = lambda: tn.round_tt(t_in, **kw)
    sec, allt, out = bc._timeit(f, reps=7, warmup=2)
    kinds = bc._kinds(f)
    print(f"{kw}: {sec * 1e3:.3f} ms per call, ranks {out.ranks_tt.tolist()}; kernel ms/launches: "
          + ", ".join(f"{k} {v['ms']:.3f}/{v['launches']}" for k, v in kinds.items())
          + f"; sum {sum(v['ms'] for v in kinds.values()):.3f} ms")
