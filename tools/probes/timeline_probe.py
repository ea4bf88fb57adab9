"""Timeline of the kernels of a few metric steps from a rocprofv3 --kernel-trace CSV: busy time per queue, their union,
overlap, and the largest idle gaps -- to see what the sub-batch streams really overlap.
    rocprofv3 --kernel-trace --output-format csv -d out -o kt -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras
    python tools/probes/timeline_probe.py out/.../kt_kernel_trace.csv"""
import csv, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in rows]
ev.sort()
# the timed steps of `bench.py --steps 3 --warmup 2`: every chunked step launches the fused push+factor kernel 12 times
# (2 sub-batches x 6 cores); steps 3..5 = launches 24..59 of it (the single-stream profiling pass comes later)
push = [e for e in ev if "qr_factor_kernel" in e[3] and "true, 8" in e[3].replace("(bool)1", "true")]
if len(push) >= 60:
    t_lo, t_hi = push[24][0], push[59][1] + 25_000_000  # + the R2L sweep of the last step
    nxt = [e[0] for e in push[60:61]]
    if nxt:
        t_hi = min(t_hi, nxt[0])
    ev = [e for e in ev if t_lo <= e[0] < t_hi]
span = ev[-1][1] - ev[0][0]
per_q = defaultdict(int)
for s, e, q, _ in ev:
    per_q[q] += e - s
# union
union, cur_s, cur_e = 0, None, None
gaps = []
for s, e, q, name in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
            gaps.append((s - cur_e, name))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
tot = sum(per_q.values())
print(f"window {span/1e6:.2f} ms, {len(ev)} kernels; sum of kernel times {tot/1e6:.2f} ms, union (GPU busy) {union/1e6:.2f} ms, "
      f"overlapped {(tot-union)/1e6:.2f} ms, idle {(span-union)/1e6:.2f} ms")
for q, t in sorted(per_q.items()):
    print(f"  queue {q}: busy {t/1e6:.2f} ms")
gaps.sort(reverse=True)
print("largest idle gaps (us, next kernel):", [(round(g/1e3, 1), n[:40]) for g, n in gaps[:8]])
print(f"number of gaps > 5 us: {sum(1 for g, _ in gaps if g > 5000)}, total {sum(g for g, _ in gaps if g > 5000)/1e6:.2f} ms")
by_kind = defaultdict(lambda: [0, 0])
for s, e, q, name in ev:
    k = name.split("<")[0].replace("ttr::", "")
    by_kind[k][0] += e - s; by_kind[k][1] += 1
for k, (t, n) in sorted(by_kind.items(), key=lambda x: -x[1][0])[:10]:
    print(f"  {k}: {t/1e6:.2f} ms in {n} launches")
