#!/usr/bin/env python3
"""Per-launch counter means of ONE kernel from the passes of tools/profile_hash.sh / profile_bh.sh
(directories sq1, sq2, sq3 under <prof dir>) -> JSON on stdout.
usage: python tools/pmc_kernel_summarise.py <prof dir> <kernel substring> [workload note]"""
import csv
import json
import os
import sys
from collections import defaultdict

prof, needle = sys.argv[1], sys.argv[2]
vals, dur, kname = {}, [], None
for name in sorted(os.listdir(prof)):
    path = os.path.join(prof, name, "run_counter_collection.csv")
    if not os.path.exists(path):
        continue
    acc = defaultdict(lambda: defaultdict(float))
    with open(path) as f:
        for row in csv.DictReader(f):
            if needle not in row["Kernel_Name"]:
                continue
            kname = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in acc.items():
        vals[k] = sum(v.values()) / len(v)
t = sum(dur) / len(dur) * 1e-9
cycles = vals["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
waves = vals.get("SQ_WAVES", 0.0)
simd_cycles = cycles * 1024.0             # 256 CUs x 4 SIMDs
out = {"kernel": kname, "workload": sys.argv[3] if len(sys.argv) > 3 else "",
       "launch_ms_profiled": t * 1e3, "effective_clock_GHz": cycles / t / 1e9, "waves": waves}
for key, ctr in (("valu_busy_fraction", "SQ_ACTIVE_INST_VALU"), ("scalar_busy_fraction", "SQ_ACTIVE_INST_SCA"),
                 ("lds_busy_fraction", "SQ_ACTIVE_INST_LDS")):
    if ctr in vals:
        out[key] = vals[ctr] * 4.0 / simd_cycles
if waves:
    out["per_wave"] = {k: vals[f"SQ_INSTS_{k}"] / waves for k in ("VALU", "SALU", "SMEM", "LDS", "BRANCH", "VMEM_RD",
                                                                   "VALU_TRANS") if f"SQ_INSTS_{k}" in vals}
    if "SQ_WAVE_CYCLES" in vals:
        out["wave_lifetime_cycles"] = vals["SQ_WAVE_CYCLES"] * 4.0 / waves
        out["mean_waves_per_simd"] = vals["SQ_WAVE_CYCLES"] * 4.0 / simd_cycles
out["counters"] = vals
print(json.dumps(out, indent=1))
