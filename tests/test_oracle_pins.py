"""Pins the CPU oracle (oracle/nbody_oracle.c) to every known-answer value the reference's own
tests and examples hold for the hot path.  Values only -- the reference's
test sources are not copied.  CPU-only."""
import math
import os

import numpy as np
import pytest

from oracle_bind import host_state


# -- tests/test_force_calculation.cpp:13-30  TwoBodyForce ------------------------------------
def test_two_body_force(oracle):
    f = oracle.pair_force((0, 0, 0), (1, 0, 0), 1.0, 1.0, 1.0, 0.0)
    assert f[0] > 0
    assert abs(f[1]) < 1e-6 and abs(f[2]) < 1e-6
    assert abs(float(np.linalg.norm(f)) - 1.0) < 1e-5


# -- :32-44  SofteningPreventsInfinity ------------------------------------------------------
def test_softening_prevents_infinity(oracle):
    f = oracle.pair_force((0, 0, 0), (0.001, 0, 0), 1.0, 1.0, 1.0, 0.1)
    assert np.all(np.isfinite(f))


# -- :46-60  ForceDirection ------------------------------------------------------------------
def test_force_direction(oracle):
    f = oracle.pair_force((0, 0, 0), (1, 1, 1), 1.0, 1.0, 1.0, 0.01)
    d = np.ones(3) / math.sqrt(3)
    fd = f / np.linalg.norm(f)
    assert np.allclose(fd, d, atol=1e-5)


# -- :101-180 the three RapidCheck properties, re-expressed on seeded random inputs ----------
def test_force_properties(oracle):
    rng = np.random.default_rng(7)
    done = 0
    while done < 300:
        p1 = rng.uniform(-100, 100, 3).astype(np.float32)
        p2 = rng.uniform(-100, 100, 3).astype(np.float32)
        m2 = float(rng.uniform(0.01, 100))
        r = float(np.linalg.norm(p2.astype(np.float64) - p1))
        if r <= 0.01:
            continue
        eps = 0.01
        f = oracle.pair_force(p1, p2, 1.0, m2, 1.0, eps).astype(np.float64)
        # NB the reference's "expected" is G m2/(r^2+eps^2), an approximation of
        # G m2 r/(r^2+eps^2)^1.5 good to <1 % for r > 0.01... only when r >> eps: as in the
        # reference, the precondition r > 0.01 with eps = 0.01 is not enough by itself, so the
        # reference property can only be asserted where r >= 10 eps.
        if r >= 10 * eps:
            exp_mag = m2 / (r * r + eps * eps)
            assert abs(np.linalg.norm(f) - exp_mag) / exp_mag < 0.01
        dirv = (p2.astype(np.float64) - p1) / r
        assert float(np.dot(dirv, f / np.linalg.norm(f))) > 0.999
        done += 1
    for _ in range(200):  # SofteningFiniteness
        p1 = rng.uniform(-100, 100, 3).astype(np.float32)
        p2 = p1 if rng.random() < 0.2 else rng.uniform(-100, 100, 3).astype(np.float32)
        eps = float(rng.uniform(0.001, 10.0))
        f = oracle.pair_force(p1, p2, 1.0, 1.0, 1.0, eps)
        assert np.all(np.isfinite(f)) and float(np.linalg.norm(f)) < 1e10


# -- tests/test_integrator.cpp:15-49  SingleStepPositionUpdate -------------------------------
def test_single_step_position_update(oracle):
    s = host_state(dict(pos_x=np.zeros(1), pos_y=np.zeros(1), pos_z=np.zeros(1),
                        vel_x=np.ones(1), vel_y=np.zeros(1), vel_z=np.zeros(1), mass=np.ones(1)))
    oracle.update_positions(s, 0.1)
    assert abs(s["pos_x"][0] - 0.1) < 1e-5 and abs(s["pos_y"][0]) < 1e-5 and abs(s["pos_z"][0]) < 1e-5


# -- :51-84  KineticEnergyCalculation ---------------------------------------------------------
def test_kinetic_energy_kat(oracle):
    s = host_state(dict(pos_x=np.zeros(2), pos_y=np.zeros(2), pos_z=np.zeros(2),
                        vel_x=np.array([1.0, 0.0]), vel_y=np.array([0.0, 2.0]), vel_z=np.zeros(2),
                        mass=np.array([1.0, 2.0])))
    for mode in (0, 2):
        assert abs(oracle.kinetic_energy(s, 256, mode) - 4.5) < 1e-4


def _binary(r, v, m=1.0):
    return host_state(dict(pos_x=np.array([-r, r]), pos_y=np.zeros(2), pos_z=np.zeros(2),
                           vel_x=np.zeros(2), vel_y=np.array([-v, v]), vel_z=np.zeros(2),
                           mass=np.array([m, m])))


# -- :90-162  EnergyConservationProperty: r=5 binary, 100 steps of 1e-3, drift < 1 %.
#    With the reference's own numbers (bodies at -+5, v = sqrt(G m / 2r)) KE = 0.1 and
#    PE = -1/sqrt(100 + 1e-4), so E0 ~ 5e-8: the reference's RELATIVE drift |dE|/|E0| is
#    ill-conditioned (another sign its GPU tests were never run).  The
#    property is asserted here on the scale the energies actually have: |dE| < 1 % of |PE|. ------
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_binary_energy_drift(oracle, mode):
    G, eps, r, m = 1.0, 0.01, 5.0, 1.0
    s = _binary(r, math.sqrt(G * m / (2 * r)))
    ax, ay, az = oracle.direct_forces(s["pos_x"], s["pos_y"], s["pos_z"], s["mass"], G, eps * eps, mode)
    s["acc_x"], s["acc_y"], s["acc_z"] = ax, ay, az
    e0 = oracle.kinetic_energy(s, 256, 0) + oracle.potential_energy(s, G, eps, 256, 0)
    oracle.integrate_direct(s, G, eps, 0.001, 100, mode)
    e1 = oracle.kinetic_energy(s, 256, 0) + oracle.potential_energy(s, G, eps, 256, 0)
    assert abs(e0) < 1e-6  # the reference's initial total energy is ~0
    assert abs(oracle.kinetic_energy(s, 256, 0) - 0.1) < 1e-4
    assert abs(e1 - e0) < 0.01 * 0.1


# -- examples/example_energy_conservation.cpp:26-147: bodies (-+1,0,0), v=(0,-+0.5,0), m=1,
#    eps=0.01, dt=1e-4, 1e5 steps; E0 ~ -0.25; "excellent" < 0.1 %.  As in the example the
#    positions are overwritten AFTER initialize(), so a(0) = 0 on the first step (:61-63). ------
def test_energy_conservation_example(oracle):
    G, eps, dt = 1.0, 0.01, 1e-4
    s = _binary(1.0, 0.5)
    e0 = oracle.kinetic_energy(s, 256, 0) + oracle.potential_energy(s, G, eps, 256, 0)
    assert abs(e0 - (-0.25)) < 1e-3
    max_drift = 0.0
    for _ in range(20):  # report interval 5000 steps
        oracle.integrate_direct(s, G, eps, dt, 5000, 0)
        e = oracle.kinetic_energy(s, 256, 0) + oracle.potential_energy(s, G, eps, 256, 0)
        max_drift = max(max_drift, abs((e - e0) / e0) * 100.0)
    assert max_drift < 0.1  # the example's "excellent" band


# -- tests/test_spatial_hash.cpp:38-51  CellIndexCalculation (cell_size 2) --------------------
def test_cell_index_kats(oracle):
    dims = [64, 64, 64]
    bmin = [0.0, 0.0, 0.0]
    c = oracle.cell_index((0.5, 0.5, 0.5), bmin, 2.0, dims)
    assert c == 0
    c = oracle.cell_index((2.5, 4.5, 6.5), bmin, 2.0, dims)
    assert c == 1 + 2 * 64 + 3 * 64 * 64


# -- tests/test_spatial_hash.cpp:89-130: every body in exactly one cell, inside its bounds ----
def test_cell_partition_property(oracle):
    import oracle_bind as ob
    rng = np.random.default_rng(3)
    n = 500
    x, y, z = (rng.uniform(-10, 10, n).astype(np.float32) for _ in range(3))
    lo, hi = oracle.bbox(x, y, z)
    cell = 2.0
    dims = oracle.grid_dims(lo, hi, cell)
    # force_spatial_hash.cu:244-246
    assert dims == [int(math.ceil(np.float32(hi[a] - lo[a]) / cell)) + 1 for a in range(3)]
    cells = np.empty(n, np.int32)
    oracle.L.oracle_assign_cells(n, x, y, z, ob._f3(*lo), cell, ob._i3(*dims), cells)
    assert cells.min() >= 0 and cells.max() < dims[0] * dims[1] * dims[2]
    cx = cells % dims[0]
    cy = (cells // dims[0]) % dims[1]
    cz = cells // (dims[0] * dims[1])
    for c, p, l in ((cx, x, lo[0]), (cy, y, lo[1]), (cz, z, lo[2])):
        assert np.all(p >= l + c * cell - 1e-4) and np.all(p <= l + (c + 1) * cell + 1e-4)


# -- arithmetic-mode consistency: fp32 sequential vs fp64-accumulated vs fp64 -----------------
def test_direct_modes_agree(oracle, nb):
    ic = nb.ic.plummer(2048, seed=1)
    a = [oracle.direct_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], 1.0, 1e-6, m)
         for m in (0, 1, 2)]
    gold = np.stack(a[2], 1).astype(np.float64)
    nrm = np.linalg.norm(gold, axis=1)
    for k, tol in ((0, 2e-5), (1, 2e-6)):
        err = np.linalg.norm(np.stack(a[k], 1) - gold, axis=1) / nrm
        assert err.max() < tol, (k, err.max())


# -- spatial hash == direct-with-cutoff when cutoff <= cell (27-cell search complete) ---------
def test_spatial_hash_equals_direct_cutoff(oracle, nb):
    ic = nb.ic.uniform_box(3000, seed=5, lo=-6, hi=6)
    x, y, z, m = ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]
    ax, ay, az = oracle.spatial_hash_forces(x, y, z, m, 1.0, 1e-4, 1.0, 1.0)
    idx = np.arange(0, 3000, 7)
    bx, by, bz = oracle.direct_cutoff_forces(x, y, z, m, idx, 1.0, 1e-4, 1.0)
    assert np.allclose(ax[idx], bx, rtol=1e-6, atol=1e-6)
    assert np.allclose(ay[idx], by, rtol=1e-6, atol=1e-6)
    assert np.allclose(az[idx], bz, rtol=1e-6, atol=1e-6)


# -- Barnes-Hut contract (tests/test_barnes_hut.cpp:131-201, tests/test_spatial_hash.cpp:186-249):
#    root mass = sum m within 0.1 %; err(theta=0.3) <= 1.1 err(theta=0.8); theta=0.1 => < 10 %
#    magnitude error vs Direct at N=30/50, eps=0.1. -------------------------------------------
def test_barnes_hut_contract(oracle, nb):
    for n, radius in ((30, 5.0), (50, 10.0)):
        ic = nb.ic.sphere(n, seed=42, radius=radius)
        x, y, z, m = ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]
        idx = np.arange(n)
        dx, dy, dz = oracle.direct_forces(x, y, z, m, 1.0, 0.01, 1)
        dmag = np.sqrt(dx.astype(np.float64) ** 2 + dy ** 2 + dz ** 2)
        errs = {}
        for theta in (0.1, 0.3, 0.8):
            bx, by, bz, root_mass, nodes = oracle.barnes_hut_forces(x, y, z, m, idx, 1.0, 0.01, theta)
            assert nodes > 0
            assert abs(root_mass - m.sum()) / m.sum() < 1e-3
            bmag = np.sqrt(bx.astype(np.float64) ** 2 + by ** 2 + bz ** 2)
            errs[theta] = float(np.max(np.abs(bmag - dmag) / np.maximum(dmag, 1e-12)))
        assert errs[0.1] < 0.10
        assert errs[0.3] <= 1.1 * errs[0.8] + 1e-12


# ---- the committed golden fixtures of the two approximate methods: the oracle must keep producing them
# (tests/golden/make_golden.py; the GPU twins are in test_barnes_hut_gpu.py / test_spatial_hash_gpu.py)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_reproduces_barnes_hut_golden(oracle):
    g = np.load(os.path.join(GOLDEN, "twogalaxies2048_barnes_hut.npz"))
    n = g["pos_x"].size
    eps2 = float(np.float32(g["eps"]) * np.float32(g["eps"]))
    bx, by, bz, root_mass, nodes = oracle.barnes_hut_forces(g["pos_x"], g["pos_y"], g["pos_z"], g["mass"],
                                                            np.arange(n), float(g["G"]), eps2, float(g["theta"]))
    assert nodes == int(g["node_count"]) and root_mass == float(g["root_mass"])
    assert np.array_equal(np.stack([bx, by, bz], 1), g["acc"])
    # theta = 0.5: a few per cent of the exact sum at worst (docs/architecture/algorithms.md:450-454 "~1 %")
    err = np.linalg.norm(g["acc"] - g["acc_direct"], axis=1) / np.linalg.norm(g["acc_direct"], axis=1)
    assert np.median(err) < 0.01 and np.quantile(err, 0.99) < 0.06 and err.max() < 0.25


def test_oracle_reproduces_spatial_hash_golden(oracle):
    g = np.load(os.path.join(GOLDEN, "uniform4096_spatial_hash.npz"))
    eps2 = float(np.float32(g["eps"]) * np.float32(g["eps"]))
    for name, cutoff in (("acc_c1", 1.0), ("acc_c2", 2.0)):
        a = np.stack(oracle.spatial_hash_forces(g["pos_x"], g["pos_y"], g["pos_z"], g["mass"], float(g["G"]), eps2,
                                                float(g["cell"]), cutoff), 1)
        assert np.array_equal(a, g[name]), name
    # cutoff <= cell: the 27-cell search finds every pair within the cutoff
    assert np.allclose(g["acc_c1"], g["acc_cutoff_direct"], rtol=1e-5, atol=1e-6)
    lo, hi, dims = oracle.hash_grid(g["pos_x"], g["pos_y"], g["pos_z"], float(g["cell"]))
    assert list(dims) == list(g["dims"])


# The reference's OWN CPU all-pairs loop -- computeReferenceForces, examples/example_force_methods.cpp:34-67, the
# only CPU force loop the reference holds -- compiled from /root/reference (oracle/Makefile.ref ->
# oracle/_ref/ref_force_loop_driver) and run on two committed body sets by tests/golden/make_refloop_fixture.py;
# its output is tests/golden/direct_refloop.npz.  The oracle's arithmetic mode 0 is that loop restated: it must
# reproduce the reference-generated accelerations BIT FOR BIT (equal and general masses, G = 1 and G != 1), and
# the parity oracle (mode 1: the same fp32 terms, fp64 accumulation) and fp64 gold must agree with it to the
# rounding of a 4,096-term fp32 running sum.  This is what pins BASELINE config 1's fixture to the reference.
@pytest.mark.parametrize("name", ["plummer4096", "sphere3000"])
def test_direct_oracle_equals_reference_cpu_loop(oracle, name):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "direct_refloop.npz"))
    x, y, z, m = (g[f"{name}_{k}"] for k in ("pos_x", "pos_y", "pos_z", "mass"))
    G, eps = float(g[f"{name}_G"]), np.float32(g[f"{name}_eps"])
    eps2 = float(eps * eps)   # the loop forms eps * eps in fp32 (:55)
    ref = g[f"{name}_acc_refloop"]
    assert ref.shape == (x.size, 3) and np.all(np.isfinite(ref))
    a0 = np.stack(oracle.direct_forces(x, y, z, m, G, eps2, 0), 1)
    assert np.array_equal(a0, ref), "oracle mode 0 is not the reference's CPU loop any more"
    for mode in (1, 2):
        a = np.stack(oracle.direct_forces(x, y, z, m, G, eps2, mode), 1).astype(np.float64)
        e = np.linalg.norm(a - ref, axis=1) / np.linalg.norm(ref, axis=1)
        assert e.max() < 2e-5 and np.median(e) < 2e-6, (mode, e.max())
    if name == "plummer4096":  # the committed config-1 golden vectors are these very numbers
        gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "plummer4096_direct.npz"))
        assert np.array_equal(gold["acc_f32seq"], ref)
        assert np.array_equal(gold["pos_x"], x) and np.array_equal(gold["mass"], m)


# The cutoff / opening distance of the oracle is the chain nvcc's default contraction is ASSUMED to make of the
# reference's `dx*dx + dy*dy + dz*dz` (one product, two fused multiply-adds) -- no reference
# fixture covers that choice.  Cross-check at tolerance level: the vectors of the round-1 oracle, which formed the
# distance UNcontracted, are kept (tests/golden/uniform4096_spatial_hash_uncontracted_cutoff.npz); the two must agree
# to fp32 rounding of a term (no pair of this fixture sits within an ulp of the cutoff sphere).
def test_spatial_hash_fixture_vs_uncontracted_cutoff_distance():
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    new, old = np.load(os.path.join(here, "uniform4096_spatial_hash.npz")), \
        np.load(os.path.join(here, "uniform4096_spatial_hash_uncontracted_cutoff.npz"))
    for k in ("acc_c1", "acc_c2"):
        a, b = new[k].astype(np.float64), old[k].astype(np.float64)
        e = np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-300)
        assert e.max() < 5e-6, (k, e.max())
