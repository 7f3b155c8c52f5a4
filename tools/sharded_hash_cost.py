#!/usr/bin/env python3
"""Cost of the sharded spatial-hash step behind the C ABI (csrc/sharded_hash.hip) measured on ONE GPU.
W VIRTUAL ranks on device 0 issue everything an 8-GPU run issues (box all-reduce, partition, migration, own grid,
halo exchange || own x own, halo grid, boundary kernel, kick); wall / W is the per-rank step, compared with the
single-GPU ParticleSystem step on the same bodies and on the per-rank share.
Usage: python tools/sharded_hash_cost.py [N] [steps] [W ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd as nb  # noqa: E402
from nbody_amd.sharded import TRANSPORT_RCCL, Comm, ShardedHash  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
worlds = [int(v) for v in sys.argv[3:]] or [1, 2, 4, 8]
dt, eps, cell, cutoff = 1e-3, 0.01, 1.0, 1.0
torch.cuda.set_device(0)
half = 0.5 * (n / 16.0) ** (1.0 / 3.0)
ic = nb.ic.uniform_box(n, seed=42, lo=-half, hi=half)


def single(count):
    h = 0.5 * (count / 16.0) ** (1.0 / 3.0)
    ps = nb.ParticleSystem()
    ps.initialize(nb.SimulationConfig(particle_count=count, force_method=nb.ForceMethod.SPATIAL_HASH, dt=dt, softening=eps,
                                      spatial_hash_cell_size=cell, spatial_hash_cutoff=cutoff),
                  initial_conditions=nb.ic.uniform_box(count, seed=42, lo=-h, hi=h))
    for _ in range(3):
        ps.update(dt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ps.update(dt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


print(f"single GPU ParticleSystem(SPATIAL_HASH) N={n}: {single(n):.3f} ms/step")
for W in worlds:
    for transport in ([TRANSPORT_RCCL] if W == 1 else []) + [None]:
        comm = Comm.init_all(W, [0] * W, transport) if transport is not None else Comm.init_all(W, [0] * W)
        s = ShardedHash(comm, n, 1.0, eps, cell, cutoff)
        s.set_state(ic)
        s.forces()
        s.step(dt, 3)
        s.synchronize()
        t0 = time.perf_counter()
        s.step(dt, steps)
        s.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
        info = s.info()
        print(f"W={W} {'RCCL' if transport is not None else 'virtual ranks / peer copies'}: {wall:.3f} ms per step of all ranks = "
              f"{wall / W:.3f} ms per rank (shard {n // W}); halo bodies {info['halo_bodies']}, migrated last step {info['migrated']}, "
              f"two_grid {info['two_grid']}, dims {info['dims']}")
        s.close()
        comm.close()
    if W > 1:
        print(f"   single GPU step at the per-rank share N={n // W}: {single(n // W):.3f} ms/step")
