#!/usr/bin/env python3
"""Run-to-run spread of the symmetric kernel timing: python tools/sweep_sym_repeat.py [N] [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd as nb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.cuda.set_device(0)
ctx = nb.default_context(0)
ic = nb.ic.plummer(n, seed=42)
p = torch.from_numpy(np.ascontiguousarray(np.stack([ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]], 1))).cuda()
for rep in range(reps):
    line = []
    for R in (8, 12, 16):
        ctx.tuning(3, R, 0)
        ms = nb.time_direct_packed(ctx, p, p, 1.0, 1e-6, 3)
        line.append(f"R={R}: {ms:.2f} ms")
    print(f"rep {rep}: " + "  ".join(line), flush=True)
ctx.tuning()
