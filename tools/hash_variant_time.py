#!/usr/bin/env python3
"""One build of the library (NBODY_HIP_LIB=...), the wave-per-cell kernels of the spatial hash at config 5 (and a denser
and a sparser grid): time per launch and a digest of the accelerations, for A/B runs of build variants.
Usage: NBODY_HIP_LIB=path python tools/hash_variant_time.py [kernel ...]"""
import hashlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb  # noqa: E402
from gpu_util import acc_of, to_device  # noqa: E402


def timeit(fn, iters=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


kernels = [int(a) for a in sys.argv[1:]] or [3, 6]
for n, half, cell, cutoff in ((4194304, 32.0, 1.0, 1.0), (4194304, 32.0, 2.0, 2.0), (1048576, 32.0, 1.0, 1.0)):
    d, _ = to_device(nb, nb.ic.uniform_box(n, seed=42, lo=-half, hi=half))
    g = nb.SpatialHashGrid(n, cell)
    g.build(d)
    line = f"{os.path.basename(os.environ.get('NBODY_HIP_LIB', 'default'))} N={n} rho={n / g.getTotalCells():.1f} cutoff={cutoff}:"
    for k in kernels:
        g.tuning(k)
        t = timeit(lambda: g.computeForces(d, cutoff, 1.0, 0.01))
        dig = hashlib.sha1(acc_of(d).tobytes()).hexdigest()[:10]
        line += f"  k{k} {t:.4f} ms [{dig}]"
    print(line, flush=True)
