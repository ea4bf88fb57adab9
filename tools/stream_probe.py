import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tntorch_amd as tn
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
inp = bench.make_input(B, dev, 1234)
def run(nchunks, reps=4):
    chunks = [[c[i*B//nchunks:(i+1)*B//nchunks] for c in inp] for i in range(nchunks)]
    streams = [torch.cuda.Stream() for _ in range(nchunks)]
    def step():
        outs = []
        cur = torch.cuda.current_stream()
        for ch, s in zip(chunks, streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                t = tn.Tensor(ch, batch=True); t.round_tt(rmax=32); outs.append(t)
        for s in streams: cur.wait_stream(s)
        return outs
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for n in (1, 2, 4, 8):
    print(n, "chunks/streams:", f"{run(n):.2f} ms/step")
