#!/usr/bin/env python3
"""Per-launch means of one kernel from the passes of tools/profile_step.sh -> JSON (incl. the `issue` view the bench
line quotes beside the byte roofline).   python tools/pmc_step_summarise.py <profdir> <kernel substring> "<workload text>" """
import csv
import json
import os
import sys
from collections import defaultdict

prof, kern, workload = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
vals, dur, kname = {}, [], None
for name in ("sq1", "sq2", "sq3", "sqc", "tcc"):
    path = f"{prof}/{name}/run_counter_collection.csv"
    if not os.path.exists(path):
        continue
    acc = defaultdict(lambda: defaultdict(float))
    with open(path) as f:
        for row in csv.DictReader(f):
            if kern not in row["Kernel_Name"]:
                continue
            kname = row["Kernel_Name"].split("(")[0]
            acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
            if name == "sq1" and row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in acc.items():
        vals[k] = sum(v.values()) / len(v)
t = sum(dur) / len(dur) * 1e-9
cycles = vals["GRBM_GUI_ACTIVE"] / 8.0           # summed over the 8 XCDs
waves = vals["SQ_WAVES"]
simds = 1024.0
out = {"kernel": kname, "workload": workload, "launches_averaged": len(dur), "launch_ms_profiled": t * 1e3,
       "effective_clock_GHz": cycles / t / 1e9, "waves": waves,
       "mean_waves_per_simd": vals["SQ_WAVE_CYCLES"] * 4.0 / (cycles * simds) if "SQ_WAVE_CYCLES" in vals else None,
       "per_wave": {k: vals[f"SQ_INSTS_{k}"] / waves for k in ("VALU", "SALU", "SMEM", "LDS", "VMEM_RD", "BRANCH", "VALU_TRANS")
                    if f"SQ_INSTS_{k}" in vals}}
if "SQ_ACTIVE_INST_VALU" in vals:
    valu = vals["SQ_ACTIVE_INST_VALU"] * 4.0 / (cycles * simds)
    sca = vals["SQ_ACTIVE_INST_SCA"] * 4.0 / (cycles * simds)
    out["valu_busy_fraction"] = valu
    out["scalar_busy_fraction"] = sca
    # instruction-issue view: cycles in which a SIMD's vector ALU (resp. the CU's scalar unit) is executing an
    # instruction / cycles available; the vector pipe is the binding one
    out["issue"] = {"bound": "valu_issue", "achieved": valu, "peak": 1.0, "unit": "fraction of SIMD cycles with the VALU busy",
                    "frac": valu, "scalar_unit_busy": sca,
                    "wait_inst_any_fraction_of_wave_cycles": (vals["SQ_WAIT_INST_ANY"] / vals["SQ_WAVE_CYCLES"]
                                                               if "SQ_WAIT_INST_ANY" in vals and "SQ_WAVE_CYCLES" in vals else None)}
if "SQC_DCACHE_REQ" in vals:
    out["scalar_cache_hit_rate"] = vals["SQC_DCACHE_HITS"] / max(vals["SQC_DCACHE_REQ"], 1.0)
if "TCC_HIT_sum" in vals:
    out["l2_hit_rate"] = vals["TCC_HIT_sum"] / max(vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"], 1.0)
    # TCC_EA0_RDREQ counts 32 B and 64 B requests alike: bytes from HBM are bounded by 64 B each (MI355X_MICROARCH.md)
    out["ea_read_requests"] = vals.get("TCC_EA0_RDREQ_sum")
    out["ea_write_requests"] = vals.get("TCC_EA0_WRREQ_sum")
out["counters"] = vals
print(json.dumps(out, indent=1))
