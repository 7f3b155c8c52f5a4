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


if __name__ == "__main__":
    main()
