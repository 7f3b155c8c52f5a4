"""Largest-size smoke: Barnes-Hut at 8M bodies, spatial hash at 16M (timings + sanity)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb  # noqa: E402
from gpu_util import acc_of, to_device  # noqa: E402


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


n = int(sys.argv[1]) if len(sys.argv) > 1 else 8 << 20
ic = nb.ic.two_galaxies(n, seed=1)
d, h = to_device(nb, ic)
fc = nb.BarnesHutCalculator(0.5)
fc.setSofteningParameter(0.1)
integ = nb.Integrator()
fc.computeForces(d)
ms = timed(lambda: integ.integrate(d, fc, 1e-3))
st = fc.getTree().stats()
a = acc_of(d)
print(f"BH two_galaxies N={n}: {ms:.2f} ms/step, nodes {st['node_count']}, root mass {st['root_mass']:.1f} "
      f"(sum m {ic['mass'].astype(np.float64).sum():.1f}), finite {np.isfinite(a).all()}", flush=True)
# parity at this size: 256 sampled bodies against the oracle's tree (same depth, same opening rule)
import oracle_bind  # noqa: E402
o = oracle_bind.load()
hs = {k: getattr(d, k).cpu().numpy() for k in ("pos_x", "pos_y", "pos_z", "mass")}
idx = np.linspace(0, n - 1, 256).astype(np.int64)
r = o.barnes_hut_forces(hs["pos_x"], hs["pos_y"], hs["pos_z"], hs["mass"], idx, 1.0, float(np.float32(0.1) ** 2), 0.5)
fc.computeForces(d)
a = acc_of(d)
ref = np.stack(r[:3], 1).astype(np.float64)
err = np.linalg.norm(a[idx] - ref, axis=1) / np.linalg.norm(ref, axis=1)
print(f"   vs oracle tree on {idx.size} bodies: nodes {r[4]} (GPU {fc.getTree().getNodeCount()}), max rel err {err.max():.2e}", flush=True)
del d, fc
torch.cuda.empty_cache()
n2 = 2 * n
half = 0.5 * (n2 / 16.0) ** (1.0 / 3.0)
ic = nb.ic.uniform_box(n2, seed=2, lo=-half, hi=half)
d, h = to_device(nb, ic)
sh = nb.SpatialHashCalculator(1.0, 1.0)
sh.setSofteningParameter(0.01)
sh.computeForces(d)
ms = timed(lambda: integ.integrate(d, sh, 1e-3))
a = acc_of(d)
print(f"HASH uniform N={n2}: {ms:.2f} ms/step, cells {sh.getGrid().getTotalCells()}, finite {np.isfinite(a).all()}, "
      f"rms |a| {np.sqrt((a.astype(np.float64) ** 2).sum(1).mean()):.3f}", flush=True)
hs = {k: getattr(d, k).cpu().numpy() for k in ("pos_x", "pos_y", "pos_z", "mass")}
idx = np.linspace(0, n2 - 1, 128).astype(np.int64)
sh.computeForces(d)
a = acc_of(d)
ref = np.stack(o.direct_cutoff_forces(hs["pos_x"], hs["pos_y"], hs["pos_z"], hs["mass"], idx, 1.0,
                                      float(np.float32(0.01) ** 2), 1.0), 1).astype(np.float64)
err = np.linalg.norm(a[idx] - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-30)
print(f"   vs all pairs within the cutoff (oracle) on {idx.size} bodies: max rel err {err.max():.2e}", flush=True)
