#!/usr/bin/env python3
"""Barnes-Hut: tree depth (30-bit keys up to 10 levels, 63-bit keys beyond) vs build / walk time and
largest leaf.  Usage: python tools/bh_depth_sweep.py [N]"""
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


n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
torch.cuda.set_device(0)
cases = [("two_galaxies", nb.ic.two_galaxies(n, seed=42), 0.1), ("plummer a=1", nb.ic.plummer(n, seed=42), 0.01),
         ("plummer a=0.1", nb.ic.plummer(n, seed=42, a=0.1, rmax=100.0), 1e-3)]
for name, ic, eps in cases:
    d, _ = to_device(nb, ic)
    for depth in (10, 11, 12, 14, 16, 20):
        tree = nb.BarnesHutTree(n)
        tree.setParams(depth, 1)
        tb = timeit(lambda: tree.build(d))
        tw = timeit(lambda: tree.computeForces(d, 0.5, 1.0, eps), iters=3, warm=1)
        nodes = tree.copyNodesToHost()
        big = int(nodes["particle_count"][nodes["is_leaf"]].max())
        print(f"{name:14s} N={n} depth {depth:2d}: build {tb:6.2f} ms, walk {tw:8.2f} ms, nodes {tree.getNodeCount():8d}, "
              f"largest leaf {big}", flush=True)
        del tree, nodes
