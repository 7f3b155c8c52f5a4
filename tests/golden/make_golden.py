"""Generates the committed golden fixtures from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The reference has no runnable CPU/GPU force path in
this container, so the vectors come from oracle/nbody_oracle.c, which
tests/test_oracle_pins.py pins to the reference's own known-answer values.

plummer4096_direct.npz   BASELINE.json config 1: N=4096 Plummer (seed 42), G=1, eps=1e-3
  inputs  pos_x pos_y pos_z vel_x vel_y vel_z mass            float32
  acc_f32seq   mode 0: fp32 sequential accumulation (the reference loop's own arithmetic)
  acc_f64acc   mode 1: fp32 pair terms, fp64 accumulation (the parity oracle)
  acc_gold     mode 2: fp64 arithmetic
  ke, pe       energies (fp64 arithmetic) of the initial state
  step10_*     state after 10 Velocity-Verlet steps of dt=1e-3 (mode 1 forces)

twogalaxies2048_barnes_hut.npz   BASELINE config 4 in small: two-galaxy N=2048 (seed 7), theta=0.5, eps=0.1
  inputs + acc (oracle tree walk), node_count, root_mass; acc_direct = the exact sum (mode 1)

uniform4096_spatial_hash.npz     BASELINE config 5 in small: uniform box [-3.2,3.2]^3 N=4096 (seed 11),
  cell = 1; acc_c1 (cutoff 1.0) and acc_c2 (cutoff 2.0 > cell: the reference's 27-cell semantics);
  bbox_min/max, dims, cell_of (cell id per body), acc_cutoff_direct (all pairs within cutoff 1.0, no grid)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import nbody_amd  # noqa: E402
import oracle_bind  # noqa: E402


def main():
    o = oracle_bind.load()
    ic = nbody_amd.ic.plummer(4096, seed=42)
    G, eps = 1.0, 1e-3
    eps2 = float(np.float32(eps) * np.float32(eps))
    out = dict(ic)
    for name, mode in (("acc_f32seq", 0), ("acc_f64acc", 1), ("acc_gold", 2)):
        a = o.direct_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], G, eps2, mode)
        out[name] = np.stack(a, 1)
    s = oracle_bind.host_state(ic)
    out["ke"] = np.float64(o.kinetic_energy(s, 256, 2))
    out["pe"] = np.float64(o.potential_energy(s, G, eps, 256, 2))
    s["acc_x"], s["acc_y"], s["acc_z"] = (out["acc_f64acc"][:, k].copy() for k in range(3))
    o.integrate_direct(s, G, eps, 1e-3, 10, 1)
    for k in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "acc_x", "acc_y", "acc_z"):
        out["step10_" + k] = s[k]
    out["G"], out["eps"], out["dt"] = np.float32(G), np.float32(eps), np.float32(1e-3)
    path = os.path.join(ROOT, "tests", "golden", "plummer4096_direct.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")

    # Barnes-Hut
    n = 2048
    ic = nbody_amd.ic.two_galaxies(n, seed=7)
    eps2 = float(np.float32(0.1) * np.float32(0.1))
    bx, by, bz, root_mass, nodes = o.barnes_hut_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"],
                                                       np.arange(n), 1.0, eps2, 0.5)
    out = dict(ic)
    out["acc"] = np.stack([bx, by, bz], 1)
    out["node_count"], out["root_mass"] = np.int64(nodes), np.float64(root_mass)
    out["acc_direct"] = np.stack(o.direct_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], 1.0, eps2, 1), 1)
    out["G"], out["eps"], out["theta"] = np.float32(1.0), np.float32(0.1), np.float32(0.5)
    path = os.path.join(ROOT, "tests", "golden", "twogalaxies2048_barnes_hut.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")

    # spatial hash
    n = 4096
    ic = nbody_amd.ic.uniform_box(n, seed=11, lo=-3.2, hi=3.2)
    eps2 = float(np.float32(0.01) * np.float32(0.01))
    out = dict(ic)
    for name, cutoff in (("acc_c1", 1.0), ("acc_c2", 2.0)):
        out[name] = np.stack(o.spatial_hash_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], 1.0, eps2,
                                                   1.0, cutoff), 1)
    out["acc_cutoff_direct"] = np.stack(o.direct_cutoff_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"],
                                                               np.arange(n), 1.0, eps2, 1.0), 1)
    lo, hi, dims = o.hash_grid(ic["pos_x"], ic["pos_y"], ic["pos_z"], 1.0)
    out["bbox_min"], out["bbox_max"], out["dims"] = np.asarray(lo, np.float32), np.asarray(hi, np.float32), np.asarray(dims, np.int32)
    out["cell_of"] = np.array([o.cell_index((ic["pos_x"][i], ic["pos_y"][i], ic["pos_z"][i]), lo, 1.0, dims)
                               for i in range(n)], np.int32)
    out["G"], out["eps"], out["cell"] = np.float32(1.0), np.float32(0.01), np.float32(1.0)
    path = os.path.join(ROOT, "tests", "golden", "uniform4096_spatial_hash.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
