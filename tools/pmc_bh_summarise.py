#!/usr/bin/env python3
"""Per-launch means of the Barnes-Hut walk from the passes of tools/profile_bh.sh -> JSON."""
import csv
import json
import sys
from collections import defaultdict

prof = sys.argv[1]
vals, dur = {}, []
for name in ("sq1", "sq2", "sq3"):
    acc = defaultdict(lambda: defaultdict(float))
    with open(f"{prof}/{name}/run_counter_collection.csv") as f:
        for row in csv.DictReader(f):
            if "bh_traverse_kernel" not in row["Kernel_Name"]:
                continue
            acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
            if name == "sq1" and row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in acc.items():
        vals[k] = sum(v.values()) / len(v)
t = sum(dur) / len(dur) * 1e-9
cycles = vals["GRBM_GUI_ACTIVE"] / 8.0
waves = vals["SQ_WAVES"]
out = {"kernel": "nbh::bh_traverse_kernel<false,false>", "workload": "two_galaxies N=1048576 theta=0.5 eps=0.05",
       "launch_ms_profiled": t * 1e3, "effective_clock_GHz": cycles / t / 1e9, "waves": waves,
       "valu_busy_fraction": vals["SQ_ACTIVE_INST_VALU"] * 4.0 / (cycles * 1024.0),
       "scalar_busy_fraction": vals["SQ_ACTIVE_INST_SCA"] * 4.0 / (cycles * 1024.0),
       "per_wave": {k: vals[f"SQ_INSTS_{k}"] / waves for k in ("VALU", "SALU", "SMEM", "LDS", "BRANCH")},
       "counters": vals}
print(json.dumps(out, indent=1))
