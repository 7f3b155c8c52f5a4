"""Short view of a rocprofv3 kernel_stats.csv: kernel (template arguments and parameter lists trimmed), calls, mean us,
total ms, share.  usage: python tools/kernel_stats_short.py <..._kernel_stats.csv> [rows]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for r in rows[:top]:
    name = r["Name"]
    if "rocprim" in name:
        what = ("onesweep_iteration" if "onesweep_iteration" in name else "onesweep_histogram" if "lambda(auto:1)#1}, rocprim" in name and "global_offsets" in name
                else "onesweep_offsets" if "global_offsets" in name else "merge_sort" if "merge" in name else "scan" if "scan" in name else "other")
        vt = "u64" if "unsigned long long" in name.split("radix_sort")[-1][:400] else "u32"
        short = f"rocprim::{what} ({vt} keys)"
    else:
        short = re.sub(r"\(.*", "", name).replace("void ", "").replace("nbh::", "")
    print(f"{short[:70]:70s} calls {int(r['Calls']):5d}  mean {float(r['AverageNs']) / 1e3:9.1f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms  {float(r['Percentage']):5.1f} %")
