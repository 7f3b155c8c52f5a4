#!/usr/bin/env python3
"""One-line view of a bench.py record: Direct ms/step, roofline fraction, general-mass launch, and (ms/step, kernel ms) of
the extras.  usage: python tools/bench_line.py <bench.json> [...]"""
import json
import sys

for p in sys.argv[1:]:
    d = json.load(open(p))
    r = d.get("roofline", {})
    gm = r.get("general_mass", {})
    extras = {k: (round(v.get("ms_per_step", 0), 4), round(v.get("roofline", {}).get("avg_kernel_ms", 0), 4))
              for k, v in d.get("extra", {}).items() if isinstance(v, dict) and "error" not in v}
    print(f"{p}: {d['value']:.4g} {d['unit']}, {d['ms_per_step']:.2f} ms/step, frac {r.get('frac', 0):.3f}, "
          f"general masses {gm.get('launch_ms', 0):.1f} ms ({gm.get('frac', 0):.3f}); extras {extras}")
