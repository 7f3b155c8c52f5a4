"""Kinetic / potential energy reductions at BASELINE sizes (ms per call)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb  # noqa: E402
from gpu_util import to_device  # noqa: E402

for n in (65536, 262144, 1 << 20):
    d, _ = to_device(nb, nb.ic.plummer(n, seed=1))
    integ = nb.Integrator()
    integ.computeEnergiesF64(d, 1.0, 0.01)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ke = integ.computeKineticEnergy(d)
    t1 = time.perf_counter()
    pe = integ.computePotentialEnergy(d, 1.0, 0.01)
    t2 = time.perf_counter()
    print(f"N={n}: KE {1e3 * (t1 - t0):.3f} ms, PE {1e3 * (t2 - t1):.3f} ms ({n * float(n) / (t2 - t1):.3e} ordered pairs/s), "
          f"KE={ke:.6f} PE={pe:.6f}", flush=True)
