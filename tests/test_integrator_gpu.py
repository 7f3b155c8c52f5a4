"""Velocity-Verlet and energy kernels (through the C ABI) against the oracle and the
reference's known answers (tests/test_integrator.cpp, examples/example_energy_conservation.cpp)."""
import math
import os

import numpy as np
import pytest
import torch

from gpu_util import acc_of, rel_err, to_device
from oracle_bind import host_state

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "plummer4096_direct.npz")


def _state(d):
    return {k: getattr(d, k).cpu().numpy().copy() for k in (
        "pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "acc_x", "acc_y", "acc_z",
        "acc_old_x", "acc_old_y", "acc_old_z", "mass")}


# tests/test_integrator.cpp:15-49
def test_single_step_position_update(nb, ctx):
    d, h = to_device(nb, dict(pos_x=np.zeros(1), pos_y=np.zeros(1), pos_z=np.zeros(1),
                              vel_x=np.ones(1), vel_y=np.zeros(1), vel_z=np.zeros(1),
                              mass=np.ones(1)))
    nb.Integrator().updatePositions(d, 0.1)
    nb.ParticleDataManager.copyToHost(h, d)
    assert abs(h.pos_x[0] - 0.1) < 1e-5 and abs(h.pos_y[0]) < 1e-5 and abs(h.pos_z[0]) < 1e-5


# tests/test_integrator.cpp:51-84
def test_kinetic_energy_kat(nb, ctx):
    d, _ = to_device(nb, dict(pos_x=np.zeros(2), pos_y=np.zeros(2), pos_z=np.zeros(2),
                              vel_x=np.array([1.0, 0.0]), vel_y=np.array([0.0, 2.0]),
                              vel_z=np.zeros(2), mass=np.array([1.0, 2.0])))
    assert abs(nb.Integrator().computeKineticEnergy(d) - 4.5) < 1e-4


def _binary(nb, r, v):
    return to_device(nb, dict(pos_x=np.array([-r, r]), pos_y=np.zeros(2), pos_z=np.zeros(2),
                              vel_x=np.zeros(2), vel_y=np.array([-v, v]), vel_z=np.zeros(2),
                              mass=np.ones(2)))


# tests/test_integrator.cpp:90-162 (see test_oracle_pins.py for why |dE| is compared with |PE|)
def test_binary_energy_drift(nb, ctx):
    G, eps, r = 1.0, 0.01, 5.0
    d, _ = _binary(nb, r, math.sqrt(G * 1.0 / (2 * r)))
    fc = nb.DirectForceCalculator()
    fc.setGravitationalConstant(G)
    fc.setSofteningParameter(eps)
    integ = nb.Integrator()
    fc.computeForces(d)
    e0 = integ.computeTotalEnergy(d, G, eps)
    for _ in range(100):
        integ.integrate(d, fc, 0.001)
    e1 = integ.computeTotalEnergy(d, G, eps)
    assert abs(integ.computeKineticEnergy(d) - 0.1) < 1e-4
    assert abs(e1 - e0) < 0.01 * 0.1


# examples/example_energy_conservation.cpp:26-147 -- the acceptance program named by north_star
def test_energy_conservation_example(nb, oracle, ctx):
    G, eps, dt = 1.0, 0.01, 1e-4
    d, _ = _binary(nb, 1.0, 0.5)  # accelerations zero at step 0, as in the example (:61-63)
    fc = nb.DirectForceCalculator()
    fc.setGravitationalConstant(G)
    fc.setSofteningParameter(eps)
    integ = nb.Integrator()
    e0 = integ.computeKineticEnergy(d) + integ.computePotentialEnergy(d, G, eps)
    assert abs(e0 - (-0.25)) < 1e-3
    max_drift = 0.0
    for _ in range(20):
        integ.integrate_steps(d, fc, dt, 5000)
        e = integ.computeKineticEnergy(d) + integ.computePotentialEnergy(d, G, eps)
        max_drift = max(max_drift, abs((e - e0) / e0) * 100.0)
    assert max_drift < 0.1  # "excellent" band of the example
    # same trajectory as the oracle's fp32 restatement, to fp32 round-off growth
    s = host_state(dict(pos_x=np.array([-1.0, 1.0]), pos_y=np.zeros(2), pos_z=np.zeros(2),
                        vel_x=np.zeros(2), vel_y=np.array([-0.5, 0.5]), vel_z=np.zeros(2),
                        mass=np.ones(2)))
    oracle.integrate_direct(s, G, eps, dt, 100000, 1)
    g = _state(d)
    for k in ("pos_x", "pos_y", "vel_x", "vel_y"):
        assert np.allclose(g[k], s[k], atol=2e-3), k


# the three stand-alone kernels against the oracle, ragged N and unaligned views
@pytest.mark.parametrize("n", [1, 3, 4, 5, 1000, 4099])
def test_vv_kernels_match_oracle(nb, oracle, ctx, n):
    rng = np.random.default_rng(n)
    ic = {k: rng.normal(0, 1, n).astype(np.float32) for k in (
        "pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "acc_x", "acc_y", "acc_z")}
    ic["mass"] = np.ones(n, np.float32)
    d, _ = to_device(nb, ic)
    s = host_state(ic)
    integ = nb.Integrator()
    dt = 0.01
    integ.storeOldAccelerations(d)
    s["acc_old_x"], s["acc_old_y"], s["acc_old_z"] = s["acc_x"].copy(), s["acc_y"].copy(), s["acc_z"].copy()
    integ.updatePositions(d, dt)
    oracle.update_positions(s, dt)
    d.acc_x.mul_(0.5)
    s["acc_x"] *= np.float32(0.5)
    integ.updateVelocities(d, dt)
    oracle.update_velocities(s, dt)
    g = _state(d)
    for k in s:
        assert np.allclose(g[k], s[k], rtol=3e-7, atol=1e-7), k


def test_vv_kernels_unaligned_views(nb, oracle, ctx):
    n = 1001
    rng = np.random.default_rng(5)
    big = {k: torch.from_numpy(rng.normal(0, 1, n + 1).astype(np.float32)).cuda() for k in nb._lib.FIELDS}
    d = nb.ParticleData()
    for k in nb._lib.FIELDS:
        setattr(d, k, big[k][1:])  # 4-byte offset: the scalar path
    d.count = n
    s = {k: big[k][1:].cpu().numpy().copy() for k in nb._lib.FIELDS}
    nb.Integrator().updatePositions(d, 0.02)
    oracle.update_positions(s, 0.02)
    nb.Integrator().updateVelocities(d, 0.02)
    oracle.update_velocities(s, 0.02)
    for k in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z"):
        assert np.allclose(getattr(d, k).cpu().numpy(), s[k], rtol=3e-7, atol=1e-7), k
        assert big[k][0].item() == pytest.approx(float(big[k][0]))  # neighbour untouched


# Integrator::integrate: fused path == the reference's 4-call sequence == oracle
def test_integrate_fused_equals_sequence_and_golden(nb, oracle, ctx):
    g = np.load(GOLD)
    ic = {k: g[k] for k in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "mass")}
    G, eps, dt = float(g["G"]), float(g["eps"]), float(g["dt"])
    fc = nb.DirectForceCalculator()
    fc.setGravitationalConstant(G)
    fc.setSofteningParameter(eps)
    integ = nb.Integrator()
    d1, _ = to_device(nb, ic)
    d2, _ = to_device(nb, ic)
    fc.computeForces(d1)   # ParticleSystem::initialize does this once (particle_system.cpp:88-91)
    fc.computeForces(d2)
    for _ in range(10):
        integ.integrate(d1, fc, dt)                      # fused
        integ.storeOldAccelerations(d2)                  # integrator.cu:224-238 spelled out
        integ.updatePositions(d2, dt)
        fc.computeForces(d2)
        integ.updateVelocities(d2, dt)
    s1, s2 = _state(d1), _state(d2)
    for k in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z"):
        assert np.allclose(s1[k], s2[k], rtol=1e-6, atol=1e-7), k
        assert np.allclose(s1[k], g["step10_" + k], rtol=2e-5, atol=2e-6), k
    a = np.stack([s1["acc_x"], s1["acc_y"], s1["acc_z"]], 1)
    ref = np.stack([g["step10_acc_x"], g["step10_acc_y"], g["step10_acc_z"]], 1)
    assert np.median(rel_err(a, ref)) < 1e-5


def test_energies_match_oracle(nb, oracle, ctx):
    g = np.load(GOLD)
    ic = {k: g[k] for k in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "mass")}
    d, _ = to_device(nb, ic)
    integ = nb.Integrator()
    ke, pe = integ.computeEnergiesF64(d, 1.0, float(g["eps"]))
    assert abs(ke - float(g["ke"])) / float(g["ke"]) < 1e-6
    assert abs(pe - float(g["pe"])) / abs(float(g["pe"])) < 1e-6
    # virial sanity of the Plummer model: 2 KE / |PE| ~ 1
    assert 0.9 < 2 * ke / abs(pe) < 1.1
    # float API == rounded fp64 value, total = fp32 sum (integrator.cu:291-293)
    assert integ.computeKineticEnergy(d) == np.float32(ke)
    assert integ.computePotentialEnergy(d, 1.0, float(g["eps"])) == np.float32(pe)
    # a larger, ragged case against the fp64 oracle
    ic = nb.ic.plummer(20001, seed=8)
    d, _ = to_device(nb, ic)
    s = host_state(ic)
    ke, pe = integ.computeEnergiesF64(d, 1.5, 0.01)
    assert abs(ke - oracle.kinetic_energy(s, 256, 2)) / ke < 1e-6
    ref = oracle.potential_energy(s, 1.5, 0.01, 256, 2)
    assert abs(pe - ref) / abs(ref) < 1e-6


# ---- step graphs (include/nbody_hip.h "step graphs"): a recorded step replays to the same bits ----

@pytest.mark.parametrize("method,n", [("bh", 3000), ("bh", 40000), ("direct", 2000), ("direct", 40000)])
def test_step_graph_replays_eager_steps(nb, ctx, method, n):
    ic = nb.ic.plummer(n, seed=31)
    out = []
    for graph in (False, True):
        d, _ = to_device(nb, ic)
        fc = nb.BarnesHutCalculator(0.5) if method == "bh" else nb.DirectForceCalculator()
        fc.setSofteningParameter(0.05)
        integ = nb.Integrator()
        fc.computeForces(d)
        integ.integrate_steps(d, fc, 1e-3, 4, graph=graph)
        integ.integrate_steps(d, fc, 1e-3, 3, graph=graph)   # second call reuses the recording
        out.append(np.stack([d.pos_x.cpu().numpy(), d.vel_y.cpu().numpy(), d.acc_z.cpu().numpy()]))
        if graph:
            fc.setSofteningParameter(0.06)                    # parameters are baked in: re-recorded
            integ.integrate_steps(d, fc, 1e-3, 2, graph=True)
            assert np.isfinite(d.pos_x.cpu().numpy()).all()
    assert np.array_equal(out[0], out[1])


# eager steps, replays and eager steps again on the same tree: the Barnes-Hut build keeps host-side state about
# its device buffers between calls (the re-armed bounding box), which a replay must invalidate
@pytest.mark.parametrize("n", [3000, 120000])
def test_step_graph_mixed_with_eager_steps(nb, ctx, n):
    ic = nb.ic.two_galaxies(n, seed=12)
    out = []
    for mixed in (False, True):
        d, _ = to_device(nb, ic)
        fc = nb.BarnesHutCalculator(0.5)
        fc.setSofteningParameter(0.05)
        integ = nb.Integrator()
        fc.computeForces(d)
        if mixed:
            integ.integrate_steps(d, fc, 1e-3, 2, graph=False)
            integ.integrate_steps(d, fc, 1e-3, 3, graph=True)
            integ.integrate_steps(d, fc, 1e-3, 2, graph=False)
            integ.integrate_steps(d, fc, 1e-3, 1, graph=True)
            integ.integrate_steps(d, fc, 1e-3, 1, graph=False)
        else:
            integ.integrate_steps(d, fc, 1e-3, 9, graph=False)
        out.append(np.stack([d.pos_x.cpu().numpy(), d.vel_y.cpu().numpy(), d.acc_z.cpu().numpy()]))
    assert np.array_equal(out[0], out[1])


def test_step_graph_rejects_host_round_trips(nb, ctx):
    ic = nb.ic.uniform_box(2000, seed=3, lo=-4.0, hi=4.0)
    d, _ = to_device(nb, ic)
    integ = nb.Integrator()
    sh = nb.SpatialHashCalculator(1.0, 1.0)
    sh.computeForces(d)
    with pytest.raises(nb.NBodyError):
        with ctx.capture():
            sh.computeForces(d)            # the grid size comes back to the host every build
    with pytest.raises(nb.NBodyError):
        with ctx.capture():
            integ.computeKineticEnergy(d)  # returns a value to the host
    # the context is usable again, eagerly and for a new recording
    ke = integ.computeKineticEnergy(d)
    assert np.isfinite(ke)
    with ctx.capture() as rec:
        integ.updatePositions(d, 1e-3)
    x0 = d.pos_x.cpu().numpy().copy()
    rec.graph.launch(2)
    ctx.synchronize()
    assert not np.array_equal(d.pos_x.cpu().numpy(), x0) or np.all(d.vel_x.cpu().numpy() == 0)
    # integrate_steps(graph=True) with a spatial-hash calculator silently stays eager
    integ.integrate_steps(d, sh, 1e-3, 2, graph=True)


# 8e: per-shard energies (nbody_hip_energies_packed) add up to the whole-system reductions
@pytest.mark.parametrize("n,parts", [(5000, 3), (777, 4), (40000, 8)])
def test_shard_energies_sum_to_whole(nb, oracle, ctx, n, parts):
    from gpu_util import packed
    from nbody_amd.distributed import HipBackend
    ic = nb.ic.plummer(n, seed=17)
    d, _ = to_device(nb, ic)
    integ = nb.Integrator()
    ke_ref, pe_ref = integ.computeEnergiesF64(d, 1.3, 0.02)
    posm = packed(ic)
    vel = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    for k, f in enumerate(("vel_x", "vel_y", "vel_z")):
        vel[:, k] = torch.from_numpy(ic[f]).cuda()
    b = HipBackend(ctx)
    cuts = np.linspace(0, n, parts + 1).astype(int)
    ke = pe = 0.0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        k, p = b.energies(posm[lo:hi], vel[lo:hi], int(lo), posm, 1.3, 0.02)
        ke, pe = ke + k, pe + p
    assert abs(ke - ke_ref) <= 1e-12 * abs(ke_ref)
    assert abs(pe - pe_ref) <= 1e-7 * abs(pe_ref)      # ordered vs unordered pair sums: fp32 terms round differently
    # disjoint sets: no self pair anywhere; the two directions agree
    a, c = posm[: n // 2], posm[n // 2:]
    _, pac = b.energies(a, vel[: n // 2], -n, c, 1.0, 0.02)
    _, pca = b.energies(c, vel[n // 2:], 1 << 40, a, 1.0, 0.02)
    assert abs(pac - pca) <= 1e-6 * abs(pac)   # fp32 per-pair terms round differently by direction
    s = host_state(ic)
    assert abs(pe_ref - oracle.potential_energy(s, 1.3, 0.02, 256, 2)) < 1e-5 * abs(pe_ref)


# The plugin contract (ref: include/nbody/force_calculator.hpp:36-58, integrator.cu:234):
# Integrator::integrate calls the calculator's VIRTUAL computeForces.  The fused launch sequence is
# an optimisation for exactly the engine's own DirectForceCalculator and must not swallow a subclass.
def test_integrate_calls_a_subclass_override(nb, ctx):
    ic = nb.ic.plummer(512, seed=5)

    class Counting(nb.DirectForceCalculator):
        calls = 0

        def computeForces(self, d_particles):
            Counting.calls += 1
            super().computeForces(d_particles)

    class ZeroForce(nb.ForceCalculator):
        calls = 0

        def computeForces(self, d_particles):
            ZeroForce.calls += 1

        def getMethod(self):
            return nb.ForceMethod.DIRECT_N2

    def run(fc):
        d, _ = to_device(nb, ic)
        fc.setGravitationalConstant(1.0)
        fc.setSofteningParameter(0.05)
        fc.computeForces(d)
        integ = nb.Integrator()
        for _ in range(3):
            integ.integrate(d, fc, 1e-3)
        integ.integrate_steps(d, fc, 1e-3, 2)
        return _state(d)

    plain = run(nb.DirectForceCalculator())
    Counting.calls = 0
    sub = run(Counting())
    assert Counting.calls == 1 + 3 + 2
    for k in ("pos_x", "vel_y", "acc_z", "acc_old_x"):
        assert np.array_equal(plain[k], sub[k]), k  # same arithmetic, fused or not
    d, _ = to_device(nb, ic)
    z = ZeroForce()
    nb.Integrator().integrate(d, z, 1e-3)
    assert ZeroForce.calls == 1
    # an out-of-range block size is reported by the fused path's caller too (validateBlockSize range)
    bad = nb.DirectForceCalculator(block_size=4096)
    with pytest.raises(nb.ValidationException):
        nb.Integrator().integrate(d, bad, 1e-3)


# ... and the same for the tree / grid calculators, whose drift rides on the build's packing pass when the
# calculator is exactly the engine's own: a subclass is called through its override, with the same trajectory
@pytest.mark.parametrize("method", ["bh", "hash"])
def test_integrate_fused_drift_build_equals_generic_path(nb, ctx, method):
    ic = nb.ic.uniform_box(6000, seed=2, lo=-4.0, hi=4.0) if method == "hash" else nb.ic.plummer(6000, seed=2)
    base = nb.BarnesHutCalculator if method == "bh" else nb.SpatialHashCalculator

    class Counting(base):
        calls = 0

        def computeForces(self, d_particles):
            Counting.calls += 1
            super().computeForces(d_particles)

    def run(fc):
        d, _ = to_device(nb, ic)
        fc.setGravitationalConstant(1.0)
        fc.setSofteningParameter(0.05)
        fc.computeForces(d)
        integ = nb.Integrator()
        for _ in range(4):
            integ.integrate(d, fc, 1e-3)
        return _state(d)

    make = (lambda c: c(0.5)) if method == "bh" else (lambda c: c(1.0, 1.0))
    fused = run(make(base))
    Counting.calls = 0
    generic = run(make(Counting))
    assert Counting.calls == 1 + 4
    for k in ("pos_x", "pos_z", "vel_y", "acc_z", "acc_old_x"):
        assert np.array_equal(fused[k], generic[k]), k


# A recorded step graph holds raw pointers into the context's workspaces (shared by every system on
# the device) and into its tree.  When a LARGER system later grows a workspace, or the tree is re-sized,
# the old buffers are freed: replaying must be refused (NBODY_HIP_ERR_STATE), not touch freed memory;
# Integrator.integrate_steps records again by itself.
def test_stale_step_graph_is_refused_and_rerecorded(nb, ctx):
    own = nb.Context()  # a context of its own: the shared default context's workspaces are already large
    small = nb.ic.plummer(3000, seed=1)
    d, _ = to_device(nb, small)
    fc = nb.DirectForceCalculator(ctx=own)
    fc.setSofteningParameter(0.05)
    integ = nb.Integrator(ctx=own)
    fc.computeForces(d)
    integ.integrate_steps(d, fc, 1e-3, 3, graph=True)
    old = integ._graph
    assert old is not None
    # a system 100x larger on the same context re-allocates posm / partial
    big = nb.ic.plummer(300000, seed=2)
    d2, _ = to_device(nb, big)
    fc2 = nb.DirectForceCalculator(ctx=own)
    fc2.setSofteningParameter(0.05)
    fc2.computeForces(d2)
    own.synchronize()
    with pytest.raises(nb.StateException, match="stale"):
        old.launch(1)
    # the high-level call recovers: same trajectory as an eager run from the same state
    ref, _ = to_device(nb, {k: getattr(d, k).cpu().numpy() for k in (
        "pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "mass")})
    for k in ("acc_x", "acc_y", "acc_z"):
        getattr(ref, k).copy_(getattr(d, k))
    integ.integrate_steps(d, fc, 1e-3, 3, graph=True)
    assert integ._graph is not old
    nb.Integrator(ctx=own).integrate_steps(ref, fc, 1e-3, 3, graph=False)
    for k in ("pos_x", "vel_y", "acc_z"):
        assert torch.equal(getattr(d, k), getattr(ref, k)), k
    # a tree re-sized after the recording
    bh = nb.BarnesHutCalculator(0.5, ctx=own)
    bh.setSofteningParameter(0.05)
    bh.computeForces(d)
    integ2 = nb.Integrator(ctx=own)
    integ2.integrate_steps(d, bh, 1e-3, 2, graph=True)
    g = integ2._graph
    bh.getTree().setParams(8, 4)
    with pytest.raises(nb.StateException, match="stale"):
        g.launch(1)
