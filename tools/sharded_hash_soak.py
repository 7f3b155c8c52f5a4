#!/usr/bin/env python3
"""Soak run of the C-ABI sharded spatial hash (W virtual ranks on one GPU) through the clumping of a uniform box: every
`every` steps the accelerations the sharded system holds are compared with the single-GPU spatial hash evaluated on the
sharded system's own positions (body by body), and the step is checked against the Velocity-Verlet update.  Exercises
migration, halo exchange, the unit form of the wave-per-cell kernel inside the two-grid calls and the one-grid fallback
as the box expands.   Usage: python tools/sharded_hash_soak.py [W] [N] [steps] [every]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb  # noqa: E402
import oracle_bind  # noqa: E402
from gpu_util import acc_of, hash_bound, hash_margin, rel_err, to_device  # noqa: E402
from nbody_amd.sharded import Comm, ShardedHash  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 600
every = int(sys.argv[4]) if len(sys.argv) > 4 else 50
G, eps, cell, cutoff, dt = 1.0, 0.01, 1.0, 1.0, 1e-3
h = 0.5 * (n / 16.0) ** (1 / 3)
ic = nb.ic.uniform_box(n, seed=3, lo=-h, hi=h)
comm = Comm.init_all(W, [0] * W)
sysm = ShardedHash(comm, n, G, eps, cell, cutoff)
sysm.set_state(ic)
sysm.forces()
worst = 0.0
orc = oracle_bind.load()
for s0 in range(0, steps, every):
    sysm.step(dt, every)
    st = sysm.get_state()
    state = {k: st[k] for k in ("pos_x", "pos_y", "pos_z")}
    state["mass"] = ic["mass"]
    for k in ("vel_x", "vel_y", "vel_z"):
        state[k] = np.zeros(n, np.float32)
    d, _ = to_device(nb, state)
    fc = nb.SpatialHashCalculator(cell, cutoff)
    fc.setGravitationalConstant(G)
    fc.setSofteningParameter(eps)
    fc.computeForces(d)
    a1 = acc_of(d)
    a = np.stack([st["acc_x"], st["acc_y"], st["acc_z"]], 1)
    nz = np.linalg.norm(a1, axis=1) > 0
    assert np.all(a[~nz] == 0)
    e = rel_err(a[nz], a1[nz])
    # two fp32 evaluations of the same terms grouped differently: the derived condition-aware criterion of
    # tests/gpu_util.py (kappa = the condition number of the body's sum, from the oracle) -- the one every test uses
    _, _, kappa = orc.spatial_hash_forces_cond(st["pos_x"], st["pos_y"], st["pos_z"], ic["mass"], G,
                                               float(np.float32(eps) ** 2), cell, cutoff)
    bound = hash_bound(kappa[nz], "gpu")
    info = sysm.info()
    cs, ce, _, _ = fc.getGrid().copyCellDataToHost()
    worst = max(worst, float(e.max()))
    print(f"step {s0 + every:5d}: max rel diff sharded vs single GPU {e.max():.2e} (worst against its bound {float((e / bound).max()):.2f}, margin err / (u kappa) {hash_margin(e, kappa[nz]):.2f}); grid {info['dims']}, two_grid {info['two_grid']}, "
          f"migrated last step {info['migrated']}, halo bodies {info['halo_bodies']}, most crowded cell {int((ce - cs).max())}, "
          f"per rank {info['local_counts']}", flush=True)
    assert np.all(e <= bound), (e.max(), float((e / bound).max()))
print(f"ok: worst {worst:.2e}")
