#!/usr/bin/env python3
"""Spatial-hash force kernels side by side (nbody_hip_grid_tuning): time and agreement, over a range of
occupancies.  Usage: python tools/hash_kernels.py [N ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb  # noqa: E402
from gpu_util import acc_of, to_device  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    torch.cuda.set_device(0)
    cases = [(4194304, 32.0, 1.0, 1.0), (4194304, 32.0, 1.0, 2.0), (1048576, 32.0, 1.0, 1.0),
             (4194304, 32.0, 2.0, 2.0), (262144, 32.0, 1.0, 1.0), (1048576, 16.0, 0.5, 0.5),
             (4194304, 32.0, 1.0, 0.5), (4194304, 32.0, 1.0, 0.7), (1048576, 50.0, 1.0, 1.0), (4194304, 50.0, 1.0, 1.0), (4194304, 42.0, 1.0, 1.0), (4194304, 38.0, 1.0, 1.0), (4194304, 35.0, 1.0, 1.0)]
    for n, half, cell, cutoff in cases:
        ic = nb.ic.uniform_box(n, seed=42, lo=-half, hi=half)
        d, _ = to_device(nb, ic)
        grid = nb.SpatialHashGrid(n, cell)
        tb = timeit(lambda: grid.build(d))
        rho = n / grid.getTotalCells()
        ref = None
        line = f"N={n} cell={cell} cutoff={cutoff} rho={rho:.2f} build {tb:.3f} ms |"
        for kern in (1, 2, 3, 4, 6, 7, 8, 9, 10):
            grid.tuning(kern)
            t = timeit(lambda: grid.computeForces(d, cutoff, 1.0, 0.01))
            a = acc_of(d).astype(np.float64)
            if ref is None:
                ref = a
                err = 0.0
            else:
                err = float((np.linalg.norm(a - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-30)).max())
            line += f" k{kern}: {t:.3f} ms (max rel diff {err:.1e})"
        print(line, flush=True)
        del grid, d


if __name__ == "__main__":
    main()
