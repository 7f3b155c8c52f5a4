import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb, oracle_bind
o = oracle_bind.load()
ctx = nb.default_context(0)
for n in (1000, 5000, 20000):
    ic = nb.ic.plummer(n, seed=3)
    p = torch.from_numpy(np.ascontiguousarray(np.stack([ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]], 1))).cuda()
    ref = np.stack(o.direct_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], 1.0, 1e-6, 1), 1).astype(np.float64)
    for tpl in (2, 4, 6, 8):
        for splits in (0, 3):
            ctx.tuning(3, tpl, splits)
            a = nb.direct_forces_packed(ctx, p, p, 1.0, 1e-6).cpu().numpy()[:, :3]
            e = np.linalg.norm(a - ref, axis=1) / np.linalg.norm(ref, axis=1)
            print(f"n={n} tpl={tpl} splits={splits} max rel err {e.max():.3e} rms {np.sqrt((e**2).mean()):.3e}", flush=True)
for n in (262144, 1 << 20):
    ic = nb.ic.plummer(n, seed=42)
    p = torch.from_numpy(np.ascontiguousarray(np.stack([ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]], 1))).cuda()
    ctx.tuning(1, 4, 0)
    base = nb.direct_forces_packed(ctx, p, p, 1.0, 1e-6).cpu().numpy()[:, :3].astype(np.float64)
    ms0 = nb.time_direct_packed(ctx, p, p, 1.0, 1e-6, 2)
    for tpl in (4, 6, 8):
        for splits in (0, 2, 4):
            ctx.tuning(3, tpl, splits)
            a = nb.direct_forces_packed(ctx, p, p, 1.0, 1e-6).cpu().numpy()[:, :3]
            e = np.linalg.norm(a - base, axis=1) / np.linalg.norm(base, axis=1)
            ms = nb.time_direct_packed(ctx, p, p, 1.0, 1e-6, 2)
            print(f"N={n} sym tpl={tpl} splits={splits}: {ms:.2f} ms ({n*n/ms/1e9:.3f}e12 pairs/s) vs one-sided {ms0:.2f} ms; max diff {e.max():.2e}", flush=True)
