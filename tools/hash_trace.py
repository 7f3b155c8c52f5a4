"""Spatial-hash steps for rocprofv3 (counters or kernel trace): hash_trace.py [N] [steps] [force kernel (grid tuning)]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb  # noqa: E402
from gpu_util import to_device  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
half = 0.5 * (n / 16.0) ** (1.0 / 3.0)
d, h = to_device(nb, nb.ic.uniform_box(n, seed=42, lo=-half, hi=half))
fc = nb.SpatialHashCalculator(1.0, 1.0)
fc.setSofteningParameter(0.01)
integ = nb.Integrator()
fc.computeForces(d)
if len(sys.argv) > 3:
    fc.getGrid().tuning(int(sys.argv[3]))
for _ in range(steps):
    integ.integrate(d, fc, 1e-3)
torch.cuda.synchronize()
