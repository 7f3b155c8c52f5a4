"""Eager launches vs a recorded step graph: ms per Velocity-Verlet step for Barnes-Hut and direct
at launch-bound sizes.  python tools/graph_probe.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb  # noqa: E402
from gpu_util import to_device  # noqa: E402


def run(method, n, steps):
    ic = nb.ic.plummer(n, seed=1)
    out = {}
    for graph in (False, True):
        d, h = to_device(nb, ic)
        fc = nb.BarnesHutCalculator(0.5) if method == "bh" else nb.DirectForceCalculator()
        fc.setSofteningParameter(0.05)
        integ = nb.Integrator()
        fc.computeForces(d)
        integ.integrate_steps(d, fc, 1e-3, 3, graph=graph)  # warm / record
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        integ.integrate_steps(d, fc, 1e-3, steps, graph=graph)
        torch.cuda.synchronize()
        out[graph] = (time.perf_counter() - t0) / steps * 1e3
        out[("x", graph)] = d.pos_x.cpu().numpy()
    same = np.array_equal(out[("x", False)], out[("x", True)])
    print(f"{method} N={n}: eager {out[False]:.4f} ms/step, graph {out[True]:.4f} ms/step, "
          f"x{out[False] / out[True]:.2f}, identical trajectories: {same}", flush=True)


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    for n in (1024, 10000, 100000, 1000000):
        run("bh", n, steps if n < 1000000 else 50)
    for n in (1024, 10000, 65536):
        run("direct", n, steps)
