import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd as nb
ctx = nb.default_context(0)
for n in (16384, 32768, 65536, 131072, 262144, 524288):
    ic = nb.ic.plummer(n, seed=42)
    p = torch.from_numpy(np.ascontiguousarray(np.stack([ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]], 1))).cuda()
    ctx.tuning(-1, 0, 0)
    ms0 = nb.time_direct_packed(ctx, p, p, 1.0, 1e-6, 5)
    out = [f"N={n}: one-sided auto {ms0:.3f} ms ({n*n/ms0/1e9:.2f}e12)"]
    for tpl in (2, 4, 6, 8):
        ctx.tuning(3, tpl, 0)
        ms = nb.time_direct_packed(ctx, p, p, 1.0, 1e-6, 5)
        out.append(f"sym R={tpl} {ms:.3f} ms ({n*n/ms/1e9:.2f}e12)")
    print(" | ".join(out), flush=True)
