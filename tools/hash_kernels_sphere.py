import sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import nbody_amd as nb
from gpu_util import to_device, acc_of
def timeit(fn, iters=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/iters*1e3
for n, R, cell, cutoff in ((10000, 10.0, 1.0, 2.0), (100000, 10.0, 1.0, 2.0), (100000, 20.0, 1.0, 1.0), (1000000, 40.0, 1.0, 2.0)):
    ic = nb.ic.sphere(n, seed=42, radius=R)
    d,_ = to_device(nb, ic)
    g = nb.SpatialHashGrid(n, cell)
    tb = timeit(lambda: g.build(d))
    line = f"sphere N={n} R={R} cell={cell} cutoff={cutoff} rho_grid={n/g.getTotalCells():.2f} build {tb:.3f} |"
    for k in (0, 1, 2, 3, 6, 8, 9):
        g.tuning(k)
        line += f" k{k} {timeit(lambda: g.computeForces(d, cutoff, 1.0, 0.1)):.3f}"
    print(line, flush=True)
