import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb
from gpu_util import to_device, acc_of
g = np.load(os.path.join(ROOT, "tests/golden/plummer4096_direct.npz"))
ref = g["acc_f64acc"].astype(np.float64)
ic = {k: g[k] for k in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "mass")}
for first in (100, None):
    if first:
        d0, _ = to_device(nb, nb.ic.sphere(first, seed=42, radius=5.0))
        c0 = nb.DirectForceCalculator(); c0.setSofteningParameter(0.1); c0.computeForces(d0)
    d, _ = to_device(nb, ic)
    calc = nb.DirectForceCalculator()
    calc.setSofteningParameter(float(g["eps"]))
    for rep in range(3):
        calc.computeForces(d)
        a = acc_of(d).astype(np.float64)
        e = np.linalg.norm(a - ref, axis=1) / np.linalg.norm(ref, axis=1)
        bad = np.nonzero(e > 1e-5)[0]
        print("first", first, "rep", rep, "max", e.max(), "nbad", bad.size, "bad idx", bad[:20], flush=True)
        for b in bad[:3]:
            print("   ", b, a[b], ref[b])
