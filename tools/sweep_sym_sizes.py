#!/usr/bin/env python3
"""Symmetric kernel, bodies per lane R vs problem size; also the two-set (pair) kernel on shard sizes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd as nb  # noqa: E402

torch.cuda.set_device(0)
ctx = nb.default_context(0)


def bodies(n, seed):
    ic = nb.ic.plummer(n, seed=seed)
    return torch.from_numpy(np.ascontiguousarray(np.stack([ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]], 1))).cuda()


for n in (32768, 65536, 131072, 262144, 524288, 1 << 20, 1 << 21):
    p = bodies(n, 42)
    line = f"all-pairs N={n}:"
    for R in (4, 8, 12, 16):
        ctx.tuning(3, R, 0)
        ms = nb.time_direct_packed(ctx, p, p, 1.0, 1e-6, 3 if n < (1 << 21) else 1)
        line += f"  R={R} {ms:.3f} ms ({float(n) * n / ms / 1e9:.2f}e12/s)"
    print(line, flush=True)
for n in (65536, 131072, 262144, 524288):
    a, b = bodies(n, 1), bodies(n, 2)
    acc_a, acc_b = torch.zeros((n, 4), device="cuda"), torch.zeros((n, 4), device="cuda")
    line = f"pair {n} x {n}:"
    for R in (4, 8, 16):
        ctx.tuning(-1, R, 0)
        nb.direct_forces_pair_packed(ctx, a, b, 1.0, 1e-6, acc_a, acc_b, False, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            nb.direct_forces_pair_packed(ctx, a, b, 1.0, 1e-6, acc_a, acc_b, False, False)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        line += f"  R={R} {ms:.3f} ms ({2.0 * n * n / ms / 1e9:.2f}e12 ordered/s)"
    print(line, flush=True)
ctx.tuning()
