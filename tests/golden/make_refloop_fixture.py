"""Generates tests/golden/direct_refloop.npz: accelerations computed by the REFERENCE's own CPU all-pairs
loop -- computeReferenceForces, examples/example_force_methods.cpp:34-67, compiled from /root/reference by
oracle/Makefile.ref into oracle/_ref/ref_force_loop_driver -- on two committed body sets:
  plummer4096   the bodies of tests/golden/plummer4096_direct.npz (BASELINE config 1), G and eps from that file
  sphere3000    3,000 bodies in a sphere, masses U[0.5, 1.5], G = 1.7, eps = 0.05 (general masses)
The driver reads the state through the reference's ParticleSystem::loadState (device round trip on this
repo's facade), so it runs on the GPU box:
    python tests/golden/make_refloop_fixture.py gpurun_out/direct_refloop.npz
and the file is then copied to tests/golden/.  The fixture is DATA (inputs + the reference's outputs)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import nbody_amd as nb  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_force_loop_driver")
LIB = os.path.join(ROOT, "n-body_amd", "lib")


def cases():
    g = np.load(os.path.join(HERE, "plummer4096_direct.npz"))
    yield "plummer4096", {k: g[k] for k in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "mass")}, \
        float(g["G"]), float(g["eps"])
    yield "sphere3000", nb.ic.sphere(3000, seed=11, radius=5.0, min_mass=0.5, max_mass=1.5), 1.7, 0.05


def run_reference_loop(ic, G, eps, workdir):
    """accelerations [n, 3] float32 as the reference's computeReferenceForces returns them"""
    n = ic["pos_x"].size
    st = nb.SimulationState(particle_count=n, simulation_time=0.0, dt=1e-3, G=G, softening=eps,
                            force_method=nb.ForceMethod.DIRECT_N2,
                            **{k: np.ascontiguousarray(ic[k], np.float32) for k in
                               ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "mass")})
    src, dst = os.path.join(workdir, "in.nbody"), os.path.join(workdir, "out.f32")
    nb.Serializer.save(src, st)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = LIB + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([DRIVER, src, dst], env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError(r.stdout + r.stderr)
    return np.fromfile(dst, dtype="<f4").reshape(n, 3).astype(np.float32), r.stdout


if __name__ == "__main__":
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, ic, G, eps in cases():
            acc, log = run_reference_loop(ic, G, eps, tmp)
            print(name, log.strip(), "|a| max", float(np.abs(acc).max()))
            for k in ("pos_x", "pos_y", "pos_z", "mass"):
                out[f"{name}_{k}"] = np.ascontiguousarray(ic[k], np.float32)
            out[f"{name}_G"], out[f"{name}_eps"] = np.float32(G), np.float32(eps)
            out[f"{name}_acc_refloop"] = acc
    np.savez_compressed(sys.argv[1], **out)
    print("wrote", sys.argv[1])
