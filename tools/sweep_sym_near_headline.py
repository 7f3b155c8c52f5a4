import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import nbody_amd as nb
torch.cuda.set_device(0)
ctx = nb.default_context(0)
def bodies(n, seed):
    ic = nb.ic.plummer(n, seed=seed)
    return torch.from_numpy(np.ascontiguousarray(np.stack([ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]], 1))).cuda()
for n in (600000, 700000, 786432, 900000, 1000000, 1048576, 1100000, 1200000, 1300000, 1400000, 1572864, 1800000):
    p = bodies(n, 42)
    line = f"N={n}:"
    best = {}
    for rep in range(2):
        for R in (8, 12, 16):
            ctx.tuning(3, R, 0)
            ms = nb.time_direct_packed(ctx, p, p, 1.0, 1e-6, 4)
            best[R] = min(best.get(R, 1e9), ms)
    for R in (8, 12, 16):
        line += f"  R={R} {best[R]:.2f} ms ({float(n) * n / best[R] / 1e9:.3f}e12/s)"
    print(line, flush=True)
