#!/usr/bin/env python3
"""Barnes-Hut walk without replicas: plain (1), pair walk in plain order (3), pair walk cost-ordered (2):
time and node visits per wave.
Usage: python tools/bh_width_sweep.py [N ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb  # noqa: E402
from gpu_util import to_device  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


sizes = [int(a) for a in sys.argv[1:]] or [1 << 20]
torch.cuda.set_device(0)
for n in sizes:
    cases = [("two_galaxies", nb.ic.two_galaxies(n, seed=42), 0.1), ("plummer a=1", nb.ic.plummer(n, seed=42), 0.01),
             ("plummer a=0.1", nb.ic.plummer(n, seed=42, a=0.1, rmax=100.0), 1e-3)]
    for name, ic, eps in cases:
        d, _ = to_device(nb, ic)
        tree = nb.BarnesHutTree(n)
        tree.build(d)
        tree.tuning(1, 0)
        line = []
        for width in (1, 3, 2):
            tree.walkForm(width)
            ms = timeit(lambda: tree.computeForces(d, 0.5, 1.0, eps))
            tree.countVisits(True)
            tree.computeForces(d, 0.5, 1.0, eps)
            visits = tree.stats()["nodes_visited"]
            tree.countVisits(False)
            waves = (n + 63) // 64
            line.append(f"form {width}: {ms:7.3f} ms, {visits / waves:7.0f} nodes/wave")
        print(f"{name:14s} N={n}: " + ", ".join(line), flush=True)
        del tree, d
