"""Kernel timeline of a small-batch round_tt call (B = 1 by default): run under rocprofv3 --kernel-trace, then summarise.
    rocprofv3 --kernel-trace --output-format csv -d out -o kt -- python tools/probes/b1_trace.py run [B]
    python tools/probes/b1_trace.py sum out/.../kt_kernel_trace.csv
The run leaves a 60 ms pause before its last 10 calls; the summary analyses what follows the last long gap: per kernel name the
launches per call and the mean duration, the kernel time per call, the wall time per call and the idle time between kernels."""
import csv
import os
import sys
import time
from collections import defaultdict

REPS = 10
if sys.argv[1] == "run":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import torch
    import bench
    import tntorch_amd as tn

    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    inp = bench.make_input(B, torch.device("cuda", 0), 7)

    def f():
        t = tn.Tensor(inp, batch=True)
        t.round_tt(rmax=32)
        return t

    for _ in range(5):
        f()
    torch.cuda.synchronize()
    time.sleep(0.06)
    t0 = time.perf_counter()
    for _ in range(REPS):
        f()
    torch.cuda.synchronize()
    print(f"B={B}: {(time.perf_counter() - t0) / REPS * 1e3:.3f} ms per call (host clock, under the tracer)")
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    cut = 0
    for i in range(1, len(ev)):
        if ev[i][0] - ev[i - 1][1] > 30_000_000:
            cut = i
    ev = ev[cut:]
    span = ev[-1][1] - ev[0][0]
    busy = sum(e - s for s, e, _ in ev)
    per = defaultdict(lambda: [0, 0])
    for s, e, n in ev:
        n = n.split("(")[0][:90]
        per[n][0] += 1
        per[n][1] += e - s
    print(f"{len(ev)} kernels in {REPS} calls ({len(ev) / REPS:.1f} per call); per call: wall {span / REPS / 1e3:.1f} us, "
          f"kernel time {busy / REPS / 1e3:.1f} us, idle {(span - busy) / REPS / 1e3:.1f} us")
    for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"  {t / REPS / 1e3:8.1f} us/call  {c / REPS:5.1f} launches  {t / c / 1e3:7.1f} us each  {n}")
    if len(sys.argv) > 3:  # the last call's kernels in launch order: start offset, duration, gap to the previous kernel
        k = len(ev) // REPS
        last = ev[-k:]
        for i, (s, e, n) in enumerate(last):
            gap = s - last[i - 1][1] if i else 0
            print(f"  {(s - last[0][0]) / 1e3:8.1f} us  {(e - s) / 1e3:6.1f} us  gap {gap / 1e3:5.1f}  {n[:110]}")
