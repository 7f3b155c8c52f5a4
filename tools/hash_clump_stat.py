#!/usr/bin/env python3
"""BASELINE config 5 as it evolves: cell occupancy, candidate pairs and step time after 0 / 250 / 1000 / 2000 / 4000 steps
(the uniform box clumps under its own short-range gravity).  Says whether a later step is slower because
there are more candidate pairs in all, or because single cells have become so crowded that their waves are the tail.
Usage: python tools/hash_clump_stat.py [N]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd as nb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
dt = 1e-3
h = 0.5 * (n / 16.0) ** (1 / 3)
ps = nb.ParticleSystem()
ps.initialize(nb.SimulationConfig(particle_count=n, dt=dt, force_method=nb.ForceMethod.SPATIAL_HASH, softening=0.01,
                                  spatial_hash_cell_size=1.0, spatial_hash_cutoff=1.0),
              initial_conditions=nb.ic.uniform_box(n, seed=42, lo=-h, hi=h))
if os.environ.get("HASH_KERNEL"):  # force one of the grid's force kernels (nbody_hip_grid_tuning)
    ps.force_calculator_.getGrid().tuning(int(os.environ["HASH_KERNEL"]))
done = 0
for upto in (0, 250, 1000, 2000, 4000):
    for _ in range(upto - done):
        ps.update(dt)
    done = upto
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ps.update(dt)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    done += 10
    grid = ps.force_calculator_.getGrid()
    cs, ce, _, _ = grid.copyCellDataToHost()
    gx, gy, gz = grid.getGridDims()
    cnt = (ce - cs).astype(np.int64).reshape(gz, gy, gx)
    pad = np.pad(cnt, 1)
    nb27 = np.zeros_like(cnt)
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                nb27 += pad[dz:dz + gz, dy:dy + gy, dx:dx + gx]
    per_cell = cnt * nb27
    pairs = float(per_cell.sum())
    print(f"after {upto:5d} steps: {ms:7.3f} ms/step; grid {gx}x{gy}x{gz}; bodies per cell max {cnt.max()}, p99.9 {np.quantile(cnt, 0.999):.0f}, "
          f"mean {cnt.mean():.1f}; candidate pairs {pairs:.3e} ({pairs / n:.0f} per body); heaviest cell {per_cell.max():.3e} pairs = "
          f"{per_cell.max() / pairs * 100:.2f} % of all; cells above 128 bodies: {(cnt > 128).sum()}", flush=True)
    # who holds the bodies, the units (one wave pass each in the wave-per-cell kernel) and the candidate pairs: by occupancy
    edges = [1, 2, 3, 5, 9, 17, 33, 65, 129, 1 << 30]
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = (cnt >= lo) & (cnt < hi)
        if sel.any():
            win = nb27[sel]
            print(f"      cells of {lo:3d}-{min(hi - 1, int(cnt.max())):3d} bodies: {int(sel.sum()):8d} cells, {int(cnt[sel].sum()):8d} bodies "
                  f"({cnt[sel].sum() / n * 100:5.1f} %), {per_cell[sel].sum() / pairs * 100:5.1f} % of the candidate pairs, window mean "
                  f"{win.mean():6.1f} max {int(win.max())}", flush=True)
