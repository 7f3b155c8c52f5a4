#!/usr/bin/env python3
"""Would a Newton's-third-law (half-shell) spatial-hash kernel pay?  Lower bound by measurement: the wave-per-cell kernel
over the half shell WITHOUT reactions (nbody_hip_grid_tuning 5: half the candidate pairs, reaction arithmetic and
reaction traffic for free), against the full kernel, plus the measured cost of streaming the deterministic reaction
slots (13 x 12 B per body written by the force kernel and read back by a fixed-order sum).
Usage: python tools/hash_n3_bound.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb  # noqa: E402
from gpu_util import to_device  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


torch.cuda.set_device(0)
for n, half, cell, cutoff in ((4194304, 32.0, 1.0, 1.0), (4194304, 32.0, 1.0, 2.0), (2000000, 13.0, 1.0, 1.0)):
    ic = nb.ic.uniform_box(n, seed=42, lo=-half, hi=half)
    d, _ = to_device(nb, ic)
    grid = nb.SpatialHashGrid(n, cell)
    grid.build(d)
    t = {}
    for kern in (3, 5):
        grid.tuning(kern)
        t[kern] = timeit(lambda: grid.computeForces(d, cutoff, 1.0, 0.01))
    grid.tuning(0)
    # the reaction slots of a deterministic half-shell kernel: 13 planes of 3 floats per body, written once and read once
    slots = torch.empty((13, 3, n), dtype=torch.float32, device="cuda")
    acc = torch.empty((3, n), dtype=torch.float32, device="cuda")
    src = torch.rand((3, n), dtype=torch.float32, device="cuda")

    def stream():
        for k in range(13):
            slots[k].copy_(src)        # stands for the force kernel's coalesced slot stores (13 x 12 B per body)
        torch.sum(slots, dim=0, out=acc)   # the fixed-order sum reads them back
    ts = timeit(stream)
    tw = timeit(lambda: [slots[k].copy_(src) for k in range(13)])
    print(f"N={n} rho={n / grid.getTotalCells():.1f} cutoff={cutoff}: full kernel {t[3]:.3f} ms | half shell, no reactions (lower bound) "
          f"{t[5]:.3f} ms | reaction-slot traffic alone: write {tw:.3f} ms, write + fixed-order read {ts:.3f} ms | "
          f"bound + slots = {t[5] + ts:.3f} ms", flush=True)
    del slots, acc, src, grid, d
