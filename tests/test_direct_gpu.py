"""Parity of the HIP Direct-N^2 path (through the C ABI) with the CPU oracle.

Tolerance (BASELINE.json north_star): accelerations within 1e-5 RELATIVE of the CPU reference,
measured per body as ||a_gpu - a_ref|| / ||a_ref|| against the fp64-accumulated oracle
(mode 1: the reference's fp32 pair arithmetic without its summation-order noise)."""
import os

import numpy as np
import pytest
import torch

from gpu_util import acc_of, packed, rel_err, to_device

pytestmark = pytest.mark.gpu
TOL = 1e-5
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "plummer4096_direct.npz")


# mirrors DirectForceCalculatorTest.ComputeForces (tests/test_force_calculation.cpp:62-96):
# N=100 sphere R=5, G=1, eps=0.1 -> finite accelerations; plus numeric parity, which the
# reference never checks.
def test_compute_forces_reference_case(nb, oracle, ctx):
    ic = nb.ic.sphere(100, seed=42, radius=5.0)
    d, h = to_device(nb, ic)
    calc = nb.DirectForceCalculator(256)
    calc.setGravitationalConstant(1.0)
    calc.setSofteningParameter(0.1)
    calc.computeForces(d)
    nb.ParticleDataManager.copyToHost(h, d)
    a = np.stack([h.acc_x, h.acc_y, h.acc_z], 1)
    assert np.all(np.isfinite(a))
    ref = np.stack(oracle.direct_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], 1.0,
                                        calc.softening_eps2_, 1), 1)
    assert rel_err(a, ref).max() < TOL


# BASELINE.json config 1 against the committed golden vectors
def test_golden_plummer4096(nb, ctx):
    g = np.load(GOLD)
    ic = {k: g[k] for k in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "mass")}
    d, _ = to_device(nb, ic)
    calc = nb.DirectForceCalculator()
    calc.setSofteningParameter(float(g["eps"]))
    calc.setGravitationalConstant(float(g["G"]))
    calc.computeForces(d)
    a = acc_of(d)
    assert rel_err(a, g["acc_f64acc"]).max() < TOL
    assert rel_err(a, g["acc_gold"]).max() < TOL
    # the reference loop's own fp32 sequential sum is itself only ~4e-6 from gold here
    assert rel_err(a, g["acc_f32seq"]).max() < 2e-5


# every kernel variant x targets-per-lane x split count, ragged N
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("tpl", [1, 2, 4])
@pytest.mark.parametrize("splits", [0, 1, 3])
def test_variants_agree_with_oracle(nb, oracle, ctx, variant, tpl, splits):
    if variant in (1, 3) and tpl == 1:
        pytest.skip("packed body needs two targets per lane")
    n = 3001
    ic = nb.ic.plummer(n, seed=3)
    ref = np.stack(oracle.direct_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], 1.0, 1e-6, 1), 1)
    p = packed(ic)
    try:
        ctx.tuning(variant, tpl, splits)
        a = nb.direct_forces_packed(ctx, p, p, 1.0, 1e-6)
        torch.cuda.synchronize()
    finally:
        ctx.tuning()
    a = a.cpu().numpy()
    assert np.all(a[:, 3] == 0)
    assert rel_err(a[:, :3], ref).max() < TOL


# BASELINE.json config 2: N=262,144 Plummer, eps=1e-3, sampled targets vs the oracle
def test_config2_262144_sampled(nb, oracle, ctx):
    n = 262144
    ic = nb.ic.plummer(n, seed=42)
    eps2 = float(np.float32(1e-3) * np.float32(1e-3))
    d, _ = to_device(nb, ic)
    calc = nb.DirectForceCalculator()
    calc.setSofteningParameter(1e-3)
    calc.computeForces(d)
    a = acc_of(d)
    assert np.all(np.isfinite(a))
    rng = np.random.default_rng(0)
    r = np.sqrt(ic["pos_x"] ** 2 + ic["pos_y"] ** 2 + ic["pos_z"] ** 2)
    idx = np.unique(np.concatenate([rng.choice(n, 1536, replace=False),
                                    np.argsort(r)[:256],      # the core: strongest cancellation
                                    np.argsort(r)[-256:]]))   # the halo
    ref = np.stack(oracle.direct_forces_indexed(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"],
                                                idx, 1.0, eps2, 1), 1)
    e = rel_err(a[idx], ref)
    assert e.max() < TOL, (e.max(), np.sqrt((e ** 2).mean()))


def test_edge_cases(nb, oracle, ctx):
    calc = nb.DirectForceCalculator()
    # N = 1: no partner, zero acceleration
    d, _ = to_device(nb, dict(pos_x=np.array([1.0]), pos_y=np.array([2.0]), pos_z=np.array([3.0]),
                              mass=np.array([5.0])))
    calc.computeForces(d)
    assert np.all(acc_of(d) == 0)
    # two bodies, eps = 0 (tests/test_force_calculation.cpp:13-30 on the kernel): a_0 = (1,0,0)
    d, _ = to_device(nb, dict(pos_x=np.array([0.0, 1.0]), pos_y=np.zeros(2), pos_z=np.zeros(2),
                              mass=np.ones(2)))
    calc.setSofteningParameter(0.0)
    calc.computeForces(d)
    a = acc_of(d)
    assert abs(a[0, 0] - 1.0) < 1e-5 and abs(a[1, 0] + 1.0) < 1e-5 and np.all(a[:, 1:] == 0)
    # coincident distinct bodies: contribute nothing (finite), eps = 0 and eps > 0
    for eps in (0.0, 1e-9, 0.05):
        ic = nb.ic.sphere(300, seed=1, radius=2.0)
        for k in ("pos_x", "pos_y", "pos_z"):
            ic[k][17] = ic[k][5]
        d, _ = to_device(nb, ic)
        calc.setSofteningParameter(eps)
        calc.computeForces(d)
        a = acc_of(d)
        assert np.all(np.isfinite(a))
        keep = np.ones(300, bool)
        keep[[5, 17]] = False
        ref = np.stack(oracle.direct_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], 1.0,
                                            calc.softening_eps2_, 1), 1)
        if eps > 1e-3:  # with softening the reference's own sum is finite for the pair too
            assert rel_err(a, ref).max() < TOL
        else:
            assert rel_err(a[keep], ref[keep]).max() < TOL
    # zero-mass bodies feel forces and exert none
    ic = nb.ic.sphere(513, seed=2, radius=3.0)
    ic["mass"][::2] = 0.0
    d, _ = to_device(nb, ic)
    calc.setSofteningParameter(0.01)
    calc.computeForces(d)
    ref = np.stack(oracle.direct_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], 1.0,
                                        calc.softening_eps2_, 1), 1)
    assert rel_err(acc_of(d), ref).max() < TOL
    # ragged sizes around the tile / block boundaries
    for n in (2, 63, 64, 65, 255, 256, 257, 511, 1025):
        ic = nb.ic.plummer(n, seed=n)
        d, _ = to_device(nb, ic)
        calc.computeForces(d)
        ref = np.stack(oracle.direct_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], 1.0,
                                            calc.softening_eps2_, 1), 1)
        assert rel_err(acc_of(d), ref).max() < TOL, n


def test_error_behaviour(nb, ctx):
    ic = nb.ic.sphere(10, seed=1)
    d, _ = to_device(nb, ic)
    for bad in (0, -1, 2048):
        with pytest.raises(nb.ValidationException):
            nb.DirectForceCalculator(bad).computeForces(d)
    with pytest.raises(nb.ValidationException):
        nb.ParticleDataManager.allocateDevice(nb.ParticleData(), 0)
    empty = nb.ParticleData()
    empty.count = 5
    with pytest.raises(nb.StateException):
        nb.DirectForceCalculator().computeForces(empty)


def test_g_scaling_and_overwrite(nb, ctx):
    ic = nb.ic.plummer(2000, seed=9)
    d, _ = to_device(nb, ic)
    d.acc_x.fill_(123.0)  # computeForces OVERWRITES (force_calculator.hpp:45-58)
    c1 = nb.DirectForceCalculator()
    c1.computeForces(d)
    a1 = acc_of(d)
    c1.setGravitationalConstant(2.5)
    c1.computeForces(d)
    a2 = acc_of(d)
    assert np.allclose(a2, 2.5 * a1, rtol=3e-7, atol=0)


# the sharded (multi-GPU) form: target sub-ranges against all sources, and the
# local + remote split with `accumulate`
def test_packed_shards_equal_whole(nb, oracle, ctx):
    n = 5000
    ic = nb.ic.plummer(n, seed=11)
    p = packed(ic)
    whole = nb.direct_forces_packed(ctx, p, p, 1.0, 1e-6).cpu().numpy()
    bounds = [0, 700, 1400, 2600, 2601, 4096, n]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        part = nb.direct_forces_packed(ctx, p[lo:hi].contiguous(), p, 1.0, 1e-6).cpu().numpy()
        assert rel_err(part[:, :3], whole[lo:hi, :3]).max() < TOL
        # local pass, then the remote sources accumulated on top
        t = p[lo:hi].contiguous()
        acc = nb.direct_forces_packed(ctx, t, t, 1.0, 1e-6)
        rest = torch.cat([p[:lo], p[hi:]]).contiguous()
        nb.direct_forces_packed(ctx, t, rest, 1.0, 1e-6, out=acc, accumulate=True)
        assert rel_err(acc.cpu().numpy()[:, :3], whole[lo:hi, :3]).max() < TOL
    # no sources at all -> zeros
    z = nb.direct_forces_packed(ctx, p[:10].contiguous(), p[:0].contiguous(), 1.0, 1e-6)
    assert torch.all(z == 0)
    # arbitrary target points that are not bodies
    rng = np.random.default_rng(4)
    tx, ty, tz = (rng.normal(0, 2, 333).astype(np.float32) for _ in range(3))
    t = torch.from_numpy(np.stack([tx, ty, tz, np.zeros(333, np.float32)], 1)).cuda()
    a = nb.direct_forces_packed(ctx, t, p, 1.0, 1e-4).cpu().numpy()
    ref = np.stack(oracle.direct_forces_points(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"],
                                               tx, ty, tz, 1.0, 1e-4), 1)
    assert rel_err(a[:, :3], ref).max() < TOL


# BASELINE.json's full size through size-independent properties: Newton's third law
# (sum m_i a_i = 0), shard consistency and determinism at N = 1,048,576
def test_full_size_properties(nb, oracle, ctx):
    n = 1 << 20
    ic = nb.ic.plummer(n, seed=42)
    p = packed(ic)
    eps2 = float(np.float32(1e-3) * np.float32(1e-3))
    a = nb.direct_forces_packed(ctx, p, p, 1.0, eps2)
    b = nb.direct_forces_packed(ctx, p, p, 1.0, eps2)
    # the all-pairs case runs the symmetric kernel (direct_sym.hip) with its deterministic slot sums
    # (default): bit for bit, launch after launch; with fp64 atomics instead the fp32 results agree to
    # <= 1 ulp, almost always bit for bit
    assert torch.equal(a, b)
    try:
        ctx.deterministic(False)
        c = nb.direct_forces_packed(ctx, p, p, 1.0, eps2)
    finally:
        ctx.deterministic(True)
    diff = (a - c).abs()
    assert int((diff > 0).sum()) < 1000
    assert float((diff / a.abs().clamp_min(1e-30)).max()) < 2.5e-7
    try:
        ctx.tuning(1, 4, 0)
        c1 = nb.direct_forces_packed(ctx, p, p, 1.0, eps2)
        c2 = nb.direct_forces_packed(ctx, p, p, 1.0, eps2)
    finally:
        ctx.tuning()
    assert torch.equal(c1, c2)  # fixed summation order
    assert rel_err(c1.cpu().numpy()[:, :3], a.cpu().numpy()[:, :3]).max() < 1e-5
    ah = a.cpu().numpy()[:, :3].astype(np.float64)
    assert np.all(np.isfinite(ah))
    m = ic["mass"].astype(np.float64)
    net = np.abs((m[:, None] * ah).sum(0))
    scale = (m * np.linalg.norm(ah, axis=1)).sum()
    assert net.max() / scale < 1e-6
    # rank r of 8 owns targets [r N/8, (r+1) N/8): its shard equals the whole's slice
    lo, hi = 3 * n // 8, 4 * n // 8
    part = nb.direct_forces_packed(ctx, p[lo:hi].contiguous(), p, 1.0, eps2).cpu().numpy()
    assert rel_err(part[:, :3], ah[lo:hi]).max() < TOL
    # and a handful of bodies against the oracle
    idx = np.array([0, 1, 12345, n // 2, n - 1], dtype=np.int64)
    ref = np.stack(oracle.direct_forces_indexed(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"],
                                                idx, 1.0, eps2, 1), 1)
    assert rel_err(ah[idx], ref).max() < TOL


# The headline configuration itself (N = 1,048,576, default tuning: direct_sym_kernel<16,...> with the
# fp64 I-side sums in LDS) against the oracle on 2,048+ bodies -- 1,536 at random, the 256 innermost
# and the 256 outermost of the Plummer sphere -- for the equal-mass body set of the bench AND a
# general-mass one (the <16,false,false> instantiation).  Tolerance: north_star's 1e-5 per body.
@pytest.mark.parametrize("masses", ["equal", "general"])
def test_headline_size_vs_oracle(nb, oracle, ctx, masses):
    n = 1 << 20
    ic = nb.ic.plummer(n, seed=42)
    if masses == "general":
        ic["mass"] = (ic["mass"] * np.random.default_rng(7).uniform(0.75, 1.25, n)).astype(np.float32)
    p = packed(ic)
    eps2 = float(np.float32(1e-3) * np.float32(1e-3))
    a = nb.direct_forces_packed(ctx, p, p, 1.0, eps2).cpu().numpy()[:, :3]
    rng = np.random.default_rng(2)
    r = np.sqrt(ic["pos_x"].astype(np.float64) ** 2 + ic["pos_y"].astype(np.float64) ** 2
                + ic["pos_z"].astype(np.float64) ** 2)
    order = np.argsort(r)
    idx = np.unique(np.concatenate([rng.choice(n, 1536, replace=False), order[:256], order[-256:]]))
    assert idx.size >= 2000
    ref = np.stack(oracle.direct_forces_indexed(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], idx,
                                                1.0, eps2, 1), 1)
    err = rel_err(a[idx], ref)
    print(f"N = {n}, {masses} masses, {idx.size} oracle bodies: max rel err {err.max():.3e}, "
          f"rms {np.sqrt((err ** 2).mean()):.3e}")
    assert err.max() < 1e-5, f"max per-body relative error {err.max():.3e} over {idx.size} bodies ({masses} masses)"


# the symmetric (action = -reaction) kernel at sizes where it is the default, all register
# blockings, ragged N (padding inside the last superblock), even and odd superblock counts
@pytest.mark.parametrize("equal_mass", [True, False])
@pytest.mark.parametrize("n,tpl", [(33000, 0), (40000, 2), (50001, 4), (70000, 6), (100003, 8), (65536, 0),
                                   (13000, 0), (90001, 16)])
def test_symmetric_kernel_vs_oracle(nb, oracle, ctx, n, tpl, equal_mass):
    ic = nb.ic.plummer(n, seed=n)
    if not equal_mass:  # the general-mass instantiation (the equal-mass one exits on the device flag)
        ic["mass"] = (ic["mass"] * np.random.default_rng(n).uniform(0.2, 3.0, n)).astype(np.float32)
        ic["mass"][::97] = 0.0
    p = packed(ic)
    eps2 = 1e-6
    try:
        ctx.tuning(3 if tpl else -1, tpl, 0)
        a = nb.direct_forces_packed(ctx, p, p, 1.0, eps2).cpu().numpy()
    finally:
        ctx.tuning()
    rng = np.random.default_rng(1)
    r = np.sqrt(ic["pos_x"] ** 2 + ic["pos_y"] ** 2 + ic["pos_z"] ** 2)
    idx = np.unique(np.concatenate([rng.choice(n, 1024, replace=False), np.argsort(r)[:128],
                                    np.arange(n - 300, n), np.arange(300)]))
    ref = np.stack(oracle.direct_forces_indexed(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], idx,
                                                1.0, eps2, 1), 1)
    assert rel_err(a[idx, :3], ref).max() < TOL
    # Newton's third law holds to rounding by construction
    m = ic["mass"].astype(np.float64)
    net = np.abs((m[:, None] * a[:, :3].astype(np.float64)).sum(0))
    assert net.max() / (m * np.linalg.norm(a[:, :3], axis=1)).sum() < 1e-6


# two disjoint sets, every pair once: action on a, reaction on b
@pytest.mark.parametrize("masses", ["equal", "a_differs", "random"])
def test_pair_kernel(nb, oracle, ctx, masses):
    ic = nb.ic.plummer(9000, seed=77)
    if masses == "a_differs":
        ic["mass"][:300] *= np.float32(2.0)
    elif masses == "random":
        ic["mass"] = (ic["mass"] * np.random.default_rng(5).uniform(0.5, 2.0, 9000)).astype(np.float32)
    p = packed(ic)
    for na in (4000, 5000, 257, 8743):
        a, b = p[:na].contiguous(), p[na:].contiguous()
        acc_a = torch.full((na, 4), 7.0, device="cuda")
        acc_b = torch.full((9000 - na, 4), 7.0, device="cuda")
        nb.direct_forces_pair_packed(ctx, a, b, 1.5, 1e-4, acc_a, acc_b)
        ra = nb.direct_forces_packed(ctx, a, b, 1.5, 1e-4)
        rb = nb.direct_forces_packed(ctx, b, a, 1.5, 1e-4)
        assert rel_err(acc_a.cpu().numpy()[:, :3], ra.cpu().numpy()[:, :3]).max() < TOL
        assert rel_err(acc_b.cpu().numpy()[:, :3], rb.cpu().numpy()[:, :3]).max() < TOL
        assert torch.all(acc_a[:, 3] == 0) and torch.all(acc_b[:, 3] == 0)
        # accumulate flags
        nb.direct_forces_pair_packed(ctx, a, b, 1.5, 1e-4, acc_a, acc_b, accumulate_a=True, accumulate_b=True)
        assert rel_err(acc_a.cpu().numpy()[:, :3], 2 * ra.cpu().numpy()[:, :3]).max() < TOL
        assert rel_err(acc_b.cpu().numpy()[:, :3], 2 * rb.cpu().numpy()[:, :3]).max() < TOL
    with pytest.raises(nb.ValidationException):
        nb.direct_forces_pair_packed(ctx, a, b, 1.0, 0.0, acc_a, acc_b)


# shard-sized sets: both >= 49,152 bodies -> 16 bodies per lane with the fp64 sums in LDS
@pytest.mark.parametrize("masses", ["equal", "random"])
def test_pair_kernel_shard_sizes(nb, ctx, masses):
    n = 110003
    ic = nb.ic.plummer(n, seed=78)
    if masses == "random":
        ic["mass"] = (ic["mass"] * np.random.default_rng(6).uniform(0.5, 2.0, n)).astype(np.float32)
    p = packed(ic)
    na = 55001
    a, b = p[:na].contiguous(), p[na:].contiguous()
    acc_a = torch.full((na, 4), 3.0, device="cuda")
    acc_b = torch.full((n - na, 4), 3.0, device="cuda")
    nb.direct_forces_pair_packed(ctx, a, b, 1.0, 1e-6, acc_a, acc_b)
    try:
        ctx.tuning(1, 4, 0)  # the one-sided kernel (verified against the oracle above) as the reference
        ra = nb.direct_forces_packed(ctx, a, b, 1.0, 1e-6)
        rb = nb.direct_forces_packed(ctx, b, a, 1.0, 1e-6)
    finally:
        ctx.tuning()
    assert rel_err(acc_a.cpu().numpy()[:, :3], ra.cpu().numpy()[:, :3]).max() < TOL
    assert rel_err(acc_b.cpu().numpy()[:, :3], rb.cpu().numpy()[:, :3]).max() < TOL


# the sharded "pair" mode's arithmetic with the HIP kernels, W ranks emulated one after the other
# on one GPU: sum over ranks of the contribution buffers == the all-pairs result
@pytest.mark.parametrize("W", [2, 3, 4, 8])
def test_pair_mode_contributions_sum_to_whole(nb, ctx, W):
    from nbody_amd.distributed import HipBackend, pair_schedule, shard_bounds
    n = 48000 + W  # ragged: the last shard is padded
    ic = nb.ic.plummer(n, seed=W)
    eps2, G = 1e-6, 1.0
    p = packed(ic)
    try:
        ctx.tuning(1, 4, 0)
        whole = nb.direct_forces_packed(ctx, p, p, G, eps2).cpu().numpy()[:, :3]
    finally:
        ctx.tuning()
    be = HipBackend(ctx)
    S = shard_bounds(n, W, 0)[0]
    allp = torch.zeros((W * S, 4), dtype=torch.float32, device="cuda")
    allp[:n] = p
    total = torch.zeros((W * S, 4), dtype=torch.float32, device="cuda")
    for r in range(W):
        c = torch.zeros_like(total)
        own = allp[r * S:(r + 1) * S].contiguous()
        mine = c[r * S:(r + 1) * S]
        be.forces(own, own, G, eps2, mine, False)
        for i0, i1, sh, j0, j1 in pair_schedule(W, r, S):
            be.forces_pair(own[i0:i1], allp[sh * S + j0:sh * S + j1], G, eps2, mine[i0:i1],
                           c[sh * S + j0:sh * S + j1], True, False)
        total += c
    got = total[:n].cpu().numpy()[:, :3]
    assert rel_err(got, whole).max() < TOL


# nbody_hip_direct_deterministic: the symmetric kernel with one slot per contribution and a fixed-order
# sum instead of fp64 atomics -- bitwise reproducible like the reference's one-thread-per-body loop
# (ref: force_direct.cu:10-85), same values as the atomic form to fp64 rounding of the sums
@pytest.mark.parametrize("n,tpl,equal_mass", [(13000, 0, True), (40000, 2, False), (70000, 6, True), (100003, 8, False),
                                              (90001, 16, True), (262144, 0, False)])
def test_deterministic_symmetric_kernel(nb, oracle, ctx, n, tpl, equal_mass):
    ic = nb.ic.plummer(n, seed=n + 1)
    if not equal_mass:
        ic["mass"] = (ic["mass"] * np.random.default_rng(n).uniform(0.2, 3.0, n)).astype(np.float32)
    p = packed(ic)
    eps2 = 1e-6
    try:
        ctx.tuning(3 if tpl else -1, tpl, 0)
        ctx.deterministic(False)
        plain = nb.direct_forces_packed(ctx, p, p, 1.0, eps2)
        assert ctx.directInfo()["last_kernel"] == 1          # symmetric kernel with fp64 atomics
        ctx.deterministic(2)                                  # slot planes REQUIRED: no silent switch
        a = nb.direct_forces_packed(ctx, p, p, 1.0, eps2)
        assert ctx.directInfo()["last_kernel"] == 2
        b = nb.direct_forces_packed(ctx, p, p, 1.0, eps2)
        c = nb.direct_forces_packed(ctx, p, p, 1.0, eps2)
    finally:
        ctx.deterministic(True)
        ctx.tuning()
    assert torch.equal(a, b) and torch.equal(a, c)  # bit for bit, launch after launch
    # (the two forms may run with different bodies per lane -- general masses: 12 with slots, 16 with atomics --
    # i.e. other fp32 partial sums; the oracle comparison below is the bar)
    assert rel_err(a.cpu().numpy()[:, :3], plain.cpu().numpy()[:, :3]).max() < 5e-6
    rng = np.random.default_rng(1)
    idx = np.unique(np.concatenate([rng.choice(n, 768, replace=False), np.arange(n - 200, n), np.arange(200)]))
    ref = np.stack(oracle.direct_forces_indexed(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], idx,
                                                1.0, eps2, 1), 1)
    assert rel_err(a.cpu().numpy()[idx, :3], ref).max() < TOL
    assert torch.all(a[:, 3] == 0)


# ... and through the drop-in path: a Direct ParticleSystem-style run continues bit for bit after a
# save / load of its state (the pause / resume property, tests/test_serialization.cpp:171-216, at a size
# where the symmetric kernel is the default)
def test_deterministic_direct_save_load_continue(nb, ctx):
    n = 40000
    ic = nb.ic.plummer(n, seed=8)
    fc = nb.DirectForceCalculator()
    fc.setSofteningParameter(0.01)
    integ = nb.Integrator()

    def run(split):
        d, _ = to_device(nb, ic)
        fc.computeForces(d)
        for _ in range(split):
            integ.integrate(d, fc, 1e-3)
        if split < 4:  # "save": positions, velocities, masses only; accelerations are recomputed on load
            state = {k: getattr(d, k).cpu().numpy().copy() for k in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "mass")}
            d, _ = to_device(nb, state)
            fc.computeForces(d)
            for _ in range(4 - split):
                integ.integrate(d, fc, 1e-3)
        return np.stack([getattr(d, k).cpu().numpy() for k in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z")])

    assert ctx.directInfo()["deterministic_mode"] == 1   # the documented default: nothing to switch on
    whole, resumed = run(4), run(2)
    assert ctx.directInfo()["last_kernel"] == 2
    assert np.array_equal(whole, resumed)


# the two-set (shard pair) kernel follows the same switch: slot planes -> bitwise reproducible, same
# values as the atomic form to rounding; mode 2 never switches silently
@pytest.mark.parametrize("na,nb_,masses", [(55001, 55002, "equal"), (55001, 55002, "random"), (20000, 9000, "random"),
                                            (4000, 5000, "equal")])
def test_deterministic_pair_kernel(nb, ctx, na, nb_, masses):
    n = na + nb_
    ic = nb.ic.plummer(n, seed=79)
    if masses == "random":
        ic["mass"] = (ic["mass"] * np.random.default_rng(6).uniform(0.5, 2.0, n)).astype(np.float32)
    p = packed(ic)
    a, b = p[:na].contiguous(), p[na:].contiguous()

    def run():
        acc_a = torch.full((na, 4), 3.0, device="cuda")
        acc_b = torch.full((nb_, 4), 3.0, device="cuda")
        nb.direct_forces_pair_packed(ctx, a, b, 1.0, 1e-6, acc_a, acc_b)
        return acc_a, acc_b
    try:
        ctx.deterministic(False)
        pa, pb = run()
        assert ctx.directInfo()["last_kernel"] == 1
        ctx.deterministic(2)
        runs = [run() for _ in range(3)]
        assert ctx.directInfo()["last_kernel"] == 2
    finally:
        ctx.deterministic(True)
    for ra, rb in runs[1:]:
        assert torch.equal(ra, runs[0][0]) and torch.equal(rb, runs[0][1])
    assert rel_err(runs[0][0].cpu().numpy()[:, :3], pa.cpu().numpy()[:, :3]).max() < 5e-6
    assert rel_err(runs[0][1].cpu().numpy()[:, :3], pb.cpu().numpy()[:, :3]).max() < 5e-6
    try:
        ctx.tuning(1, 4, 0)  # the one-sided kernel (verified against the oracle above) as the reference
        oa = nb.direct_forces_packed(ctx, a, b, 1.0, 1e-6)
        ob = nb.direct_forces_packed(ctx, b, a, 1.0, 1e-6)
    finally:
        ctx.tuning()
    assert rel_err(runs[0][0].cpu().numpy()[:, :3], oa.cpu().numpy()[:, :3]).max() < TOL
    assert rel_err(runs[0][1].cpu().numpy()[:, :3], ob.cpu().numpy()[:, :3]).max() < TOL


# nbody_hip_direct_info / the memory policy of the slot planes: the plan is visible before the call, the mode
# that ran after it; a workspace four times too large is given back; mode 0 releases the planes; mode 2 fails
# loudly where mode 1 falls back
def test_direct_info_and_workspace_policy(nb, ctx):
    big, small = 300000, 20000
    pb_, ps_ = packed(nb.ic.plummer(big, seed=1)), packed(nb.ic.plummer(small, seed=2))
    plan = ctx.directInfo(big, 1e-6)
    assert plan["deterministic_mode"] == 1 and plan["kernel"] == 2 and plan["kernel_name"] == "symmetric+slots"
    assert plan["bodies_per_lane_equal"] == 16 and plan["bodies_per_lane_general"] == 16
    nb.direct_forces_packed(ctx, pb_, pb_, 1.0, 1e-6)
    held = ctx.directInfo()["workspace_bytes_held"]
    assert held >= plan["workspace_bytes_needed"] > 100 << 20
    assert ctx.directInfo()["last_kernel"] == 2
    nb.direct_forces_packed(ctx, ps_, ps_, 1.0, 1e-6)      # a much smaller problem follows: the planes go back
    assert ctx.directInfo()["workspace_bytes_held"] < held // 4
    nb.direct_forces_packed(ctx, pb_, pb_, 1.0, 1e-6)
    try:
        ctx.deterministic(False)                             # switching the planes off frees them
        assert ctx.directInfo()["workspace_bytes_held"] == 0
        assert ctx.directInfo(big, 1e-6)["kernel"] == 1
    finally:
        ctx.deterministic(True)
    # one-sided kernel below 12,288 bodies and for tiny softening
    assert ctx.directInfo(5000, 1e-6)["kernel"] == 0 and ctx.directInfo(big, 0.0)["kernel"] == 0
    # beyond the 24 GiB budget: mode 1 reports (and takes) the atomic form, mode 2 refuses
    huge = ctx.directInfo(6_000_000, 1e-6)
    assert huge["kernel"] == 1 and huge["slot_bytes_wanted"] > 24 << 30
    with pytest.raises(nb.ValidationException):
        ctx.deterministic(3)
    # a tight budget (a shared GPU): mode 1 takes the atomic form and says so, mode 2 refuses the call
    try:
        ctx.slotBudget(8 << 20)
        assert ctx.directInfo(big, 1e-6)["kernel"] == 1
        ref = nb.direct_forces_packed(ctx, pb_, pb_, 1.0, 1e-6)
        assert ctx.directInfo()["last_kernel"] == 1
        ctx.deterministic(2)
        with pytest.raises(nb.ResourceException, match="slot planes"):
            nb.direct_forces_packed(ctx, pb_, pb_, 1.0, 1e-6)
        with pytest.raises(nb.ResourceException, match="slot planes"):
            nb.direct_forces_pair_packed(ctx, pb_[:150000].contiguous(), pb_[150000:].contiguous(), 1.0, 1e-6,
                                         torch.empty((150000, 4), device="cuda"), torch.empty((150000, 4), device="cuda"))
        ctx.slotBudget(0)
        det = nb.direct_forces_packed(ctx, pb_, pb_, 1.0, 1e-6)
        assert ctx.directInfo()["last_kernel"] == 2
        assert rel_err(det.cpu().numpy()[:, :3], ref.cpu().numpy()[:, :3]).max() < 5e-6
    finally:
        ctx.slotBudget(0)
        ctx.deterministic(True)


# at the end of this file the session-wide context is in the documented default mode (the fixture in conftest.py
# restores it after every test; the tests after this file rely on it)
def test_default_context_is_deterministic_at_the_end(nb, ctx):
    info = ctx.directInfo(1 << 20, 1e-6)
    assert info["deterministic_mode"] == 1 and info["kernel"] == 2
    assert nb.default_context(0) is ctx
