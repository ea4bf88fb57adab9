"""Sub-batch streams: product path (round_tt chunks internally) vs manual chunking vs single stream.  GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tntorch_amd as tn
from tntorch_amd import _hipops
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda", 0)
inp = bench.make_input(B, dev, 1234)

def timeit(step, reps=4):
    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def product():
    t = tn.Tensor(inp, batch=True); t.round_tt(rmax=32); return t

def manual(nchunks):
    chunks = [[c[i*B//nchunks:(i+1)*B//nchunks] for c in inp] for i in range(nchunks)]
    streams = [torch.cuda.Stream() for _ in range(nchunks)]
    def step():
        cur = torch.cuda.current_stream()
        outs = []
        for ch, s in zip(chunks, streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                t = tn.Tensor(ch, batch=True); t.round_tt(rmax=32); outs.append(t)
        for s in streams: cur.wait_stream(s)
        return outs
    return step

_hipops.STREAM_CHUNKS_ENABLED = False
print("product, single stream  :", f"{timeit(product):.2f} ms", flush=True)
for n in (2, 3, 2, 3):
    print(f"manual {n} chunks         :", f"{timeit(manual(n)):.2f} ms", flush=True)
_hipops.STREAM_CHUNKS_ENABLED = True
print("product, internal chunks:", f"{timeit(product):.2f} ms", flush=True)
