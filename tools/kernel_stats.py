#!/usr/bin/env python3
"""Prints a rocprofv3 --stats kernel summary (run_kernel_stats.csv) compactly.  usage: kernel_stats.py <csv> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{r['Name'][:84]:84s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.1f} us {float(r['Percentage']):6.2f} %")
