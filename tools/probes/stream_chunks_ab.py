"""Sub-batch streams of a batch sweep at the benchmark's B: TTR_STREAM_CHUNKS = 1 .. 4 (two are the default from 128 items).
    python tools/probes/stream_chunks_ab.py [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
import tntorch_amd as tn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
inp = bench.make_input(B, dev, seed=1234)


def step():
    t = tn.Tensor(inp, batch=True); t.round_tt(rmax=32); return t


for rep in range(2):
    for ch in ("2", "1", "3", "4", "2"):
        os.environ["TTR_STREAM_CHUNKS"] = ch
        evs = []
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            if len(evs) >= 2:
                evs.pop(0).synchronize()
            step()
            e = torch.cuda.Event(); e.record(); evs.append(e)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f"B={B} chunks={ch}: {ms:.2f} ms/step = {B * 8 / ms * 1e3:.0f} cores/s")
