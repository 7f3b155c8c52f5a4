"""Barnes-Hut steps for rocprofv3 (kernel trace or counters): bh_small_trace.py [N] [steps] [ic name] [walk form]"""
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
icname = sys.argv[3] if len(sys.argv) > 3 else "plummer"
d, h = to_device(nb, getattr(nb.ic, icname)(n, seed=1))
fc = nb.BarnesHutCalculator(0.5)
fc.setSofteningParameter(0.05)
integ = nb.Integrator()
fc.computeForces(d)
if len(sys.argv) > 4:
    fc.getTree().walkForm(int(sys.argv[4]))
integ.integrate_steps(d, fc, 1e-3, steps, graph=False)
torch.cuda.synchronize()
