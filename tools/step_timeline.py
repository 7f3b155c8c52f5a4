#!/usr/bin/env python3
"""One step out of a rocprofv3 kernel trace: the launches between two occurrences of a marker kernel, with start
offsets, gaps and durations.  usage: python tools/step_timeline.py <..._kernel_trace.csv> [marker substring] [which]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "update_velocities"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -4
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[which], idx[which + 1]
t0 = int(rows[a]["End_Timestamp"])
prev = t0
for r in rows[a + 1:b + 1]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("nbh::", "")
    if "rocprim" in name:
        name = "rocprim-pass"
    print(f"{(st - t0) / 1e3:9.1f} gap {(st - prev) / 1e3:6.1f} dur {(en - st) / 1e3:8.1f} grid {r['Grid_Size_X']:>10} s{r['Stream_Id']} {name[:70]}")
    prev = max(prev, en)
