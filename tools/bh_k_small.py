import os, sys, time
import torch
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb
from gpu_util import to_device
def timed(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/iters*1e6
for icname, eps in (("plummer", 0.05), ("two_galaxies", 0.1)):
    for n in (1024, 4096, 10000, 20000, 40000):
        ic = getattr(nb.ic, icname)(n, seed=1)
        d, _ = to_device(nb, ic)
        tree = nb.BarnesHutTree(n); tree.build(d)
        line = f"{icname} N={n}:"
        tree.tuning(0, 0); line += f" auto {timed(lambda: tree.computeForces(d, 0.5, 1.0, eps)):.0f}us"
        for K in (4, 8, 16):
            for units in (0, 4, 8, 16):
                tree.tuning(K, units)
                line += f"  K{K}/u{units} {timed(lambda: tree.computeForces(d, 0.5, 1.0, eps)):.0f}"
        print(line, flush=True)
