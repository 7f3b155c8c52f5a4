#!/usr/bin/env python3
"""Per-kernel time of the LAST k steps of a rocprofv3 --kernel-trace CSV (a run that evolves a system for thousands of
steps: what a step costs at the end, kernel by kernel, with the gaps).  A step starts at every dispatch of `marker`.
usage: python tools/trace_tail.py <run_kernel_trace.csv> [marker substring = drift_pack_bbox] [k = 50]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "drift_pack_bbox"
k = int(sys.argv[3]) if len(sys.argv) > 3 else 50
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(marks) < k + 1:
    raise SystemExit(f"only {len(marks)} steps in the trace")
first, last = marks[-k - 1], marks[-1]
sel = rows[first:last]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(rows[last]["Start_Timestamp"])
agg = defaultdict(lambda: [0, 0])
busy = 0
for r in sel:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[name][0] += 1
    agg[name][1] += d
    busy += d
print(f"last {k} steps: {(t1 - t0) / k / 1e3:.1f} us per step on the stream, kernels {busy / k / 1e3:.1f} us, gaps {(t1 - t0 - busy) / k / 1e3:.1f} us")
for name, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name[:100]:100s} {c / k:5.1f} per step  {d / c / 1e3:9.1f} us each  {d / k / 1e3:9.1f} us per step")
