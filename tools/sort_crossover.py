#!/usr/bin/env python3
"""Where our own driver of the Onesweep kernels (csrc/onesweep.h: no fill launches) overtakes rocprim::radix_sort_pairs
(merge sort below 300,000 keys): Barnes-Hut build and spatial-hash step at several sizes with NBH_OWN_SORT_FROM = 0
(always ours) and = 2^30 (never).  One child process per setting (the variable is read when a tree / grid is created).
Usage: python tools/sort_crossover.py [N ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
import nbody_amd as nb
n = int(sys.argv[1])
dt = 1e-3
out = []
for method, ic, kw in (("bh", nb.ic.two_galaxies(n, seed=42), dict(force_method=nb.ForceMethod.BARNES_HUT, softening=0.1, barnes_hut_theta=0.5)),
                       ("hash", nb.ic.uniform_box(n, seed=42, lo=-0.5 * (n / 16.0) ** (1 / 3), hi=0.5 * (n / 16.0) ** (1 / 3)),
                        dict(force_method=nb.ForceMethod.SPATIAL_HASH, softening=0.01, spatial_hash_cell_size=1.0, spatial_hash_cutoff=1.0))):
    if method == "bh":
        ic["mass"] = (ic["mass"] / n).astype("float32")
    ps = nb.ParticleSystem()
    ps.initialize(nb.SimulationConfig(particle_count=n, dt=dt, **kw), initial_conditions=ic)
    for _ in range(10):
        ps.update(dt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        ps.update(dt)
    torch.cuda.synchronize()
    out.append("%%s %%.4f" %% (method, (time.perf_counter() - t0) * 10))
print(" ".join(out))
''' % ROOT

sizes = [int(v) for v in sys.argv[1:]] or [32768, 65536, 131072, 262144, 524288, 1048576]
for n in sizes:
    row = []
    for frm in ("0", str(1 << 30)):
        env = dict(os.environ, NBH_OWN_SORT_FROM=frm)
        r = subprocess.run([sys.executable, "-c", CHILD, str(n)], env=env, capture_output=True, text=True, timeout=600)
        row.append(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "failed: " + r.stderr[-300:])
    print(f"N={n:8d}  ours: {row[0]} ms/step   rocPRIM: {row[1]} ms/step", flush=True)
