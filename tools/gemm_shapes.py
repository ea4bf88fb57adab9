#!/usr/bin/env python3
"""Per-grid breakdown of one kernel family from a rocprofv3 kernel-trace CSV."""
import csv, sys, re
from collections import defaultdict
path, pat = sys.argv[1], sys.argv[2]
agg = defaultdict(list)
for r in csv.DictReader(open(path)):
    if pat in r["Kernel_Name"]:
        key = (re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "")[:40], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k}  calls={len(v):4d} total_ms={sum(v)/1e6:8.3f} avg_us={sum(v)/len(v)/1e3:8.1f}  {100*sum(v)/tot:5.1f}%")
