#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db or *_kernel_trace.csv) into a
per-kernel table: calls, total / average / min / max duration, share of GPU time.
Usage: python tools/rocprof_summary.py <results.db | kernel_trace.csv> [> profiles/rNN_kernel_stats.txt]"""
import csv
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:110]


def rows_from_db(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    namecol = "display_name" if "display_name" in cols else ("kernel_name" if "kernel_name" in cols else cols[-1])
    q = f"select s.{namecol}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id"
    for name, st, en in db.execute(q):
        yield name, (en - st)


def rows_from_csv(path):
    with open(path) as f:
        for r in csv.DictReader(f):
            yield r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])


def main():
    path = sys.argv[1]
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    agg = defaultdict(list)
    for name, dur in rows:
        agg[short(name)].append(dur)
    total = sum(sum(v) for v in agg.values())
    print(f"# rocprofv3 --kernel-trace summary of {path}")
    print(f"# total kernel time {total/1e6:.3f} ms over {sum(len(v) for v in agg.values())} dispatches")
    print(f"{'kernel':112s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{name:112s} {len(v):6d} {sum(v)/1e6:10.3f} {sum(v)/len(v)/1e3:10.1f} {min(v)/1e3:10.1f} {max(v)/1e3:10.1f} {100*sum(v)/total:6.2f}")


if __name__ == "__main__":
    main()
