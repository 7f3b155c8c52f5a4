#!/usr/bin/env python3
"""Direct kernels at mid sizes: one-sided vs symmetric with R = 2, 4, 8 (ms per launch)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd as nb  # noqa: E402

SPLITS = [int(x) for x in sys.argv[1:]] or [0]
torch.cuda.set_device(0)
ctx = nb.default_context(0)
for n in (4096, 8192, 10000, 12288, 16384, 24576, 32768, 49152, 65536, 98304, 131072):
    ic = nb.ic.plummer(n, seed=42)
    p = torch.from_numpy(np.ascontiguousarray(np.stack([ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]], 1))).cuda()
    line = f"N={n}:"
    ctx.tuning(1, 0, 0)
    ms = nb.time_direct_packed(ctx, p, p, 1.0, 1e-6, 5)
    line += f"  one-sided {ms:.3f} ms ({float(n) * n / ms / 1e9:.2f}e12)"
    for R in (2, 4, 6, 8):
        for splits in SPLITS:
            ctx.tuning(3, R, splits)
            ms = nb.time_direct_packed(ctx, p, p, 1.0, 1e-6, 5)
            line += f"  R={R}/s{splits} {ms:.3f} ({float(n) * n / ms / 1e9:.2f})"
    print(line, flush=True)
ctx.tuning()
