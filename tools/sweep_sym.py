#!/usr/bin/env python3
"""Symmetric direct kernel: time vs bodies-per-lane R and splits.  python tools/sweep_sym.py [N] [equal|plummer]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd as nb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
torch.cuda.set_device(0)
ctx = nb.default_context(0)
ic = nb.ic.plummer(n, seed=42)
p = torch.from_numpy(np.ascontiguousarray(np.stack([ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]], 1))).cuda()
for label, q in (("equal-mass", p), ("mixed-mass", p * torch.tensor([1, 1, 1, 1.0], device="cuda") + torch.cat(
        [torch.zeros(n, 3, device="cuda"), 1e-9 * torch.arange(n, device="cuda", dtype=torch.float32)[:, None]], 1))):
    for R in (8, 16):
        for splits in (0, 8, 12, 24, 32, 48, 64):
            ctx.tuning(3, R, splits)
            ms = nb.time_direct_packed(ctx, q, q, 1.0, 1e-6, 3)
            rate = float(n) * n / (ms * 1e-3)
            print(f"{label} N={n} R={R} splits={splits}: {ms:.2f} ms  {rate:.3e} pairs/s  frac(20 flop)={rate * 20 / 157.3e12:.3f}",
                  flush=True)
ctx.tuning()
