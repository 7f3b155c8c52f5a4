"""Split walk at mid sizes: replicas K with units of n/96 bodies (tools, not product)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb  # noqa: E402
from gpu_util import to_device  # noqa: E402


def timed(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for icname in ("plummer", "two_galaxies", "uniform_box"):
    for n in (65536, 100000, 131072, 200000, 262144, 524288):
        ic = getattr(nb.ic, icname)(n, seed=1)
        d, _ = to_device(nb, ic)
        tree = nb.BarnesHutTree(n)
        tree.build(d)
        line = f"{icname} N={n}:"
        for K in (1, 2, 4, 8):
            if K * n > 2 * 262144 * 2:
                continue
            tree.tuning(K, max(1, 96 // K))
            t = timed(lambda: tree.computeForces(d, 0.5, 1.0, 0.05))
            line += f"  K{K} {t:.0f}us"
        print(line, flush=True)
