import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb, oracle_bind
o = oracle_bind.load()
ctx = nb.default_context(0)
n = 20000
ic = nb.ic.plummer(n, seed=3)
p = torch.from_numpy(np.ascontiguousarray(np.stack([ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]], 1))).cuda()
ref = np.stack(o.direct_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], 1.0, 1e-6, 1), 1).astype(np.float64)
for tpl in (2, 4):
    for splits in (1, 2, 5, 7, 11, 21, 0):
        for rep in range(2):
            ctx.tuning(3, tpl, splits)
            a = nb.direct_forces_packed(ctx, p, p, 1.0, 1e-6).cpu().numpy()[:, :3]
            e = np.linalg.norm(a - ref, axis=1) / np.linalg.norm(ref, axis=1)
            bad = np.nonzero(e > 1e-5)[0]
            S = 256 * tpl
            print(f"tpl={tpl} splits={splits} rep={rep} max {e.max():.2e} nbad {bad.size} blocks {sorted(set((bad // S).tolist()))[:12]} idx {bad[:6].tolist()}", flush=True)
