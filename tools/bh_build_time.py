#!/usr/bin/env python3
"""Barnes-Hut build and walk time at the default depth.  Usage: python tools/bh_build_time.py [N ...]"""
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
for n in [int(a) for a in sys.argv[1:]] or [1 << 20]:
    for name, ic, eps in (("two_galaxies", nb.ic.two_galaxies(n, seed=42), 0.1), ("plummer", nb.ic.plummer(n, seed=42), 0.01)):
        d, _ = to_device(nb, ic)
        tree = nb.BarnesHutTree(n)
        tb = timeit(lambda: tree.build(d))
        tw = timeit(lambda: tree.computeForces(d, 0.5, 1.0, eps), iters=10)
        print(f"{name:13s} N={n}: build {tb:6.3f} ms, walk {tw:6.3f} ms, nodes {tree.getNodeCount()}", flush=True)
        del tree, d
