#!/usr/bin/env python3
"""Reduces the rocprofv3 passes of tools/profile_direct.sh to one JSON entry (per-launch means of the
dominant kernel), with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE
and WRITE_SIZE are in KB; FETCH_SIZE x 2 on gfx950).
usage: python tools/pmc_summarise.py <prof dir> <kernel substring> <N> [out.json key]"""
import csv
import json
import os
import sys

prof, needle, n = sys.argv[1], sys.argv[2], int(sys.argv[3])


def counters(name):
    vals, dur = {}, []
    with open(os.path.join(prof, name, "run_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            if needle not in row["Kernel_Name"]:
                continue
            vals.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            vals[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
            dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    return {k: sum(v.values()) / len(v) for k, v in vals.items()}, (sum(dur) / len(dur) if dur else 0.0)


out = {"kernel_filter": needle, "bodies": n}
with open(os.path.join(prof, "stats", "run_kernel_stats.csv")) as f:
    for row in csv.DictReader(f):
        if needle in row["Name"]:
            out["kernel"] = row["Name"].split("(")[0].replace("void ", "")
            out["kernel_trace_calls"] = int(row["Calls"])
            out["kernel_trace_avg_ms"] = float(row["AverageNs"]) / 1e6
            break
fetch, _ = counters("fetch")
write, _ = counters("write")
tcc, _ = counters("tcc")
sq, dur_ns = counters("sq")
pairs = float(n) * n
t = dur_ns * 1e-9
out["launch_ms_profiled"] = dur_ns / 1e6
out["FETCH_SIZE_KB"] = fetch.get("FETCH_SIZE")
out["WRITE_SIZE_KB"] = write.get("WRITE_SIZE")
out["hbm_read_bytes_per_launch"] = 2.0 * 1024.0 * fetch.get("FETCH_SIZE", 0.0)   # gfx950: x 2
out["hbm_write_bytes_per_launch"] = 1024.0 * write.get("WRITE_SIZE", 0.0)
out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
out["algorithmic_bytes_per_launch"] = 32.0 * n
out.update({k: v for k, v in tcc.items()})
gui = sq.get("GRBM_GUI_ACTIVE", 0.0)
out["GRBM_GUI_ACTIVE_sum8xcd"] = gui
out["effective_clock_GHz"] = gui / 8.0 / t / 1e9 if t else None
out["SQ_INSTS_VALU"] = sq.get("SQ_INSTS_VALU")
out["valu_wave_instr_per_64_ordered_pairs"] = sq.get("SQ_INSTS_VALU", 0.0) / (pairs / 64.0)
out["SQ_ACTIVE_INST_VALU_quadcycles"] = sq.get("SQ_ACTIVE_INST_VALU")
# busy = quad-cycles x 4 / (1024 SIMDs x cycles), cycles from the measured clock
cycles = gui / 8.0
out["valu_busy_fraction"] = sq.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / 1024.0 / cycles if cycles else None
out["cycles_per_64_ordered_pairs"] = cycles * 1024.0 / (pairs / 64.0) if cycles else None
for k in ("SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"):
    out[k] = sq.get(k)
print(json.dumps(out, indent=1))
if len(sys.argv) > 5:
    path, key = sys.argv[4], sys.argv[5]
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[key] = out
    json.dump(doc, open(path, "w"), indent=1)
