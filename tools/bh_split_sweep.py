"""Split-traversal sweep: traversal time vs replicas / units per replica at small N (tools, not product)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb  # noqa: E402
from gpu_util import acc_of, to_device  # noqa: E402


def timed(fn, iters=30):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


for icname in ("plummer", "two_galaxies"):
    for n in (1024, 4096, 10000, 32768, 65536, 131072):
        ic = getattr(nb.ic, icname)(n, seed=1)
        d, h = to_device(nb, ic)
        tree = nb.BarnesHutTree(n)
        tb = timed(lambda: tree.build(d))
        line = f"{icname} N={n}: build {tb:.0f}us |"
        ref = None
        for K in (1, 4, 8, 16):
            for sl in ((0,) if K == 1 else (2, 4, 8, 16, 32)):
                tree.tuning(K, sl)
                t = timed(lambda: tree.computeForces(d, 0.5, 1.0, 0.05))
                a = acc_of(d)
                if ref is None:
                    ref = a
                err = np.abs(a - ref).max() / np.abs(ref).max()
                line += f" K{K}/u{sl} {t:.0f}us" + (f"(!{err:.1e})" if err > 1e-6 else "")
        tree.tuning(0, 0)
        t = timed(lambda: tree.computeForces(d, 0.5, 1.0, 0.05))
        print(line + f" | auto {t:.0f}us", flush=True)
