"""What a live process group costs the compute of a step (found with bench.py's forced N > 1 path on one GPU: 27.5 instead of
25.2 ms per step with NOTHING but `init_process_group("nccl")` different).  One variant per process:
    python tools/probes/dist_overhead_probe.py <variant> [B]
variants: none | gloo | nccl_lazy (no device_id, no collective) | nccl_eager (device_id=) | nccl_used (one all_reduce) |
          nccl_destroyed (used, then destroy_process_group) | nccl_hipri (used; the sweep on a high-priority stream)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

import bench
import tntorch_amd as tn

variant = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if variant == "gloo":
    dist.init_process_group("gloo", rank=0, world_size=1)
elif variant == "nccl_lazy":
    dist.init_process_group("nccl", rank=0, world_size=1)
elif variant.startswith("nccl"):
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if variant != "nccl_eager":
        x = torch.ones(1024, device=dev)
        dist.all_reduce(x)
        torch.cuda.synchronize()
    if variant == "nccl_destroyed":
        dist.destroy_process_group()
inp = bench.make_input(B, dev, seed=1234)
stream = torch.cuda.Stream(device=dev, priority=-1) if variant == "nccl_hipri" else torch.cuda.current_stream(dev)


def step():
    with torch.cuda.stream(stream):
        t = tn.Tensor(inp, batch=True)
        t.round_tt(rmax=32)
    return t


for _ in range(4):
    keep = step()
torch.cuda.synchronize()
evs = []
t0 = time.perf_counter()
for _ in range(16):
    if len(evs) >= 2:
        evs.pop(0).synchronize()
    keep = step()
    e = torch.cuda.Event(); e.record(stream); evs.append(e)
torch.cuda.synchronize()
print(f"{variant:16s} B={B}: {(time.perf_counter() - t0) / 16 * 1e3:.3f} ms/step", flush=True)
