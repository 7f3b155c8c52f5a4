"""Barnes-Hut steps at a launch-bound size, for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb  # noqa: E402
from gpu_util import to_device  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
d, h = to_device(nb, nb.ic.plummer(n, seed=1))
fc = nb.BarnesHutCalculator(0.5)
fc.setSofteningParameter(0.05)
integ = nb.Integrator()
fc.computeForces(d)
integ.integrate_steps(d, fc, 1e-3, steps, graph=False)
torch.cuda.synchronize()
