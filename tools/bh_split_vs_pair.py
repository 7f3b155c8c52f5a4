import os, sys, time
import torch
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb
from gpu_util import to_device
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
torch.cuda.set_device(0)
for n in (8192, 16384, 32768, 65536, 100000, 131072):
    for name, ic, eps in (("two_galaxies", nb.ic.two_galaxies(n, seed=42), 0.1), ("plummer", nb.ic.plummer(n, seed=42), 0.01)):
        d, _ = to_device(nb, ic)
        tree = nb.BarnesHutTree(n); tree.build(d)
        out = []
        tree.tuning(0, 0); tree.walkForm(0)
        out.append(f"auto {timeit(lambda: tree.computeForces(d, 0.5, 1.0, eps)):.3f}")
        for k in (2, 4):
            tree.tuning(k, 0)
            out.append(f"K={k} {timeit(lambda: tree.computeForces(d, 0.5, 1.0, eps)):.3f}")
        tree.tuning(1, 0); tree.walkForm(1)
        out.append(f"K=1 plain {timeit(lambda: tree.computeForces(d, 0.5, 1.0, eps)):.3f}")
        tree.walkForm(2)
        out.append(f"K=1 pair {timeit(lambda: tree.computeForces(d, 0.5, 1.0, eps)):.3f}")
        print(f"{name:13s} N={n}: " + ", ".join(out) + " ms", flush=True)
