#!/usr/bin/env python3
"""Times Barnes-Hut (BASELINE config 4 shape) and spatial hash (config 5 shape) on one GPU."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb  # noqa: E402
from gpu_util import to_device  # noqa: E402


def timeit(fn, iters=5, warm=2):
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
    which = sys.argv[1:] or ["bh", "hash"]
    if "bh" in which:
        for n, maker, name in ((1 << 20, nb.ic.two_galaxies, "two_galaxies"), (1 << 20, nb.ic.plummer, "plummer"),
                               (262144, nb.ic.plummer, "plummer")):
            ic = maker(n, seed=42)
            d, _ = to_device(nb, ic)
            for leaf_max in (1, 8, 16):
                tree = nb.BarnesHutTree(n)
                tree.setParams(10, leaf_max)
                tb = timeit(lambda: tree.build(d))
                tf = timeit(lambda: tree.computeForces(d, 0.5, 1.0, 0.1))
                tree.countVisits(True)
                tree.computeForces(d, 0.5, 1.0, 0.1)
                st = tree.stats()
                waves = (n + 63) // 64
                print(f"BH {name} N={n} leaf_max={leaf_max}: build {tb:.2f} ms, traverse {tf:.2f} ms, "
                      f"nodes {st['node_count']}, visits/wave {st['nodes_visited'] / waves:.0f}", flush=True)
                del tree
            calc = nb.BarnesHutCalculator(0.5)
            calc.setSofteningParameter(0.1)
            integ = nb.Integrator()
            ts = timeit(lambda: integ.integrate(d, calc, 1e-3))
            print(f"BH {name} N={n}: full Velocity-Verlet step {ts:.2f} ms -> {1e3 / ts:.1f} steps/s", flush=True)
    if "hash" in which:
        for n, half in ((4194304, 32.0), (524288, 16.0)):
            ic = nb.ic.uniform_box(n, seed=42, lo=-half, hi=half)
            d, _ = to_device(nb, ic)
            grid = nb.SpatialHashGrid(n, 1.0)
            tb = timeit(lambda: grid.build(d))
            tf = timeit(lambda: grid.computeForces(d, 1.0, 1.0, 0.01))
            print(f"HASH uniform N={n} cell=1 cutoff=1: build {tb:.2f} ms, forces {tf:.2f} ms, "
                  f"cells {grid.getTotalCells()}", flush=True)
            calc = nb.SpatialHashCalculator(1.0, 1.0)
            calc.setSofteningParameter(0.01)
            integ = nb.Integrator()
            ts = timeit(lambda: integ.integrate(d, calc, 1e-3))
            print(f"HASH uniform N={n}: full Velocity-Verlet step {ts:.2f} ms -> {1e3 / ts:.1f} steps/s", flush=True)


if __name__ == "__main__":
    main()
