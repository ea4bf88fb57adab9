"""One resident fp64 batch of BASELINE config C2 (rank-64 TT, 10 cores x mode 128, 256 tensors) rounded a few times: the
workload of the fp64 kernel trace / PMC passes in profiles/ (rocprofv3 ... -- python tools/c2_step.py [B] [reps])."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tntorch_amd as tn
from tntorch_amd import _hipops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
_hipops.STREAM_CHUNKS_ENABLED = False   # one stream: traced kernel durations must not overlap
dev = torch.device("cuda")
N, I, r = 10, 128, 32
gen = torch.Generator(device=dev).manual_seed(5)
rr = [1] + [r] * (N - 1) + [1]
cores = []
for k in range(N):
    g = torch.randn((B, rr[k], I, rr[k + 1]), generator=gen, device=dev, dtype=torch.float64)
    if k == 0:
        c = torch.cat([g, g], dim=-1)
    elif k == N - 1:
        c = torch.cat([g, g], dim=-3)
    else:
        z = torch.zeros_like(g)
        c = torch.cat([torch.cat([g, z], dim=-1), torch.cat([z, g], dim=-1)], dim=-3)
    cores.append(c.contiguous())
t = tn.Tensor(cores, batch=True)
for _ in range(reps):
    out = tn.round_tt(t, rmax=r)
torch.cuda.synchronize()
print("ranks", out.ranks_tt.tolist())
