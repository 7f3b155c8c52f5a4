"""Timeline of one step from a rocprofv3 results db: python tools/trace_step.py <db> [marker] [step#]"""
import sqlite3
import sys

db, marker = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "tree_root")
which = int(sys.argv[3]) if len(sys.argv) > 3 else 20
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if marker in r[0]]
a, b = idx[which], idx[which + 1]
t0, prev = rows[a][1], None
for r in rows[a:b]:
    gap = (r[1] - prev) / 1e3 if prev else 0.0
    name = r[0].replace("nbh::", "").replace("void ", "")
    if "rocprim" in name:
        name = "rocprim:" + name.split("detail::")[-1][:40]
    print(f"{(r[1] - t0) / 1e3:8.1f}us dur {(r[2] - r[1]) / 1e3:7.1f} gap {gap:5.1f} grid {r[3]:8d} {name[:60]}")
    prev = r[2]
print("step total us", (rows[b][1] - t0) / 1e3, "kernels", b - a)
