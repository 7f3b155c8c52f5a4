"""The multi-GPU layer behind the C ABI (include/nbody_hip_comm.h, csrc/sharded.hip) on the test box's ONE GPU:
  * transport P2P with 1-8 VIRTUAL ranks on device 0 -- the whole orchestration of BASELINE config 3's step
    (index-range shards, all-gather by peer copies || own x own, every shard pair once, point-to-point
    reaction exchange, fixed-order sum + kick, events between the ranks' streams) against the single-GPU
    engine and the oracle;
  * transport RCCL with the one rank a box has (ncclCommInitRank / ncclCommInitAll, in-place all-gather,
    grouped send / recv code path with an empty schedule);
  * the plugin form (whole-system ParticleData in, accelerations out) through Integrator.integrate.
RCCL refuses two ranks on one device, so the W > 1 RCCL exchange is first run by the driver's multi-GPU bench."""
import numpy as np
import pytest
import torch

from gpu_util import acc_of, hash_bound, hash_margin, packed, rel_err, to_device

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _single_gpu(nb, ic, G, eps, dt, steps):
    d, _ = to_device(nb, ic)
    fc = nb.DirectForceCalculator()
    fc.setGravitationalConstant(G)
    fc.setSofteningParameter(eps)
    fc.computeForces(d)
    a0 = acc_of(d)
    integ = nb.Integrator()
    for _ in range(steps):
        integ.integrate(d, fc, dt)
    state = {k: getattr(d, k).cpu().numpy() for k in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z",
                                                       "acc_x", "acc_y", "acc_z")}
    return a0, state, (integ.computeKineticEnergyF64(d) if hasattr(integ, "computeKineticEnergyF64") else None)


@pytest.mark.parametrize("W,n,masses", [(1, 20000, "equal"), (2, 30001, "equal"), (3, 30000, "random"), (4, 50003, "equal"),
                                        (8, 140000, "equal"), (8, 40000, "random"), (5, 9, "equal")])
def test_virtual_ranks_step_equals_single_gpu_and_oracle(nb, oracle, ctx, W, n, masses):
    from nbody_amd.sharded import TRANSPORT_P2P, Comm, ShardedDirect
    G, eps, dt, steps = 1.3, 0.01, 1e-3, 3
    ic = nb.ic.plummer(n, seed=100 + W)
    if masses == "random":
        ic["mass"] = (ic["mass"] * np.random.default_rng(W).uniform(0.5, 2.0, n)).astype(np.float32)
    comm = Comm.init_all(W, [0] * W, TRANSPORT_P2P)
    assert comm.info() == {"world": W, "nlocal": W, "transport": TRANSPORT_P2P, "local_ranks": list(range(W))}
    sysm = ShardedDirect(comm, n, G, eps)
    sysm.set_state(ic)
    sysm.forces()
    a_sh = np.stack([sysm.get_state()[k] for k in ("acc_x", "acc_y", "acc_z")], 1)
    a0, ref_state, _ = _single_gpu(nb, ic, G, eps, dt, steps)
    assert rel_err(a_sh, a0).max() < TOL
    eps2 = float(np.float32(eps) * np.float32(eps))
    idx = np.unique(np.concatenate([np.random.default_rng(1).choice(n, min(n, 512), replace=False), [0, n - 1]]))
    orc = np.stack(oracle.direct_forces_indexed(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], idx, G, eps2, 1), 1)
    assert rel_err(a_sh[idx], orc).max() < TOL
    ke0, pe0 = sysm.energies()
    sysm.step(dt, steps)
    st = sysm.get_state()
    for k in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z"):
        assert np.allclose(st[k], ref_state[k], rtol=1e-5, atol=1e-6), k
    a_ref = np.stack([ref_state[k] for k in ("acc_x", "acc_y", "acc_z")], 1)
    assert rel_err(np.stack([st[k] for k in ("acc_x", "acc_y", "acc_z")], 1), a_ref).max() < 2e-5
    # energies: the shards' sums equal the oracle's fp64 reductions of the same state
    s0 = {k: np.ascontiguousarray(ic[k], np.float32) for k in ic}
    assert abs(ke0 - oracle.kinetic_energy(s0, 256, 2)) <= 1e-6 * abs(ke0) + 1e-12
    assert abs(pe0 - oracle.potential_energy(s0, G, eps, 256, 2)) <= 1e-5 * abs(pe0)
    ke1, pe1 = sysm.energies()
    assert abs((ke1 + pe1) - (ke0 + pe0)) < 1e-4 * abs(pe0)
    # bitwise reproducible run after run (deterministic two-set kernel + fixed-order sum of the received blocks)
    again = ShardedDirect(comm, n, G, eps)
    again.set_state(ic)
    again.forces()
    again.step(dt, steps)
    st2 = again.get_state()
    for k in st:
        assert np.array_equal(st[k], st2[k]), k
    again.close()
    sysm.close()
    comm.close()


# BASELINE config 3 at its full size: N = 1,048,576 Plummer bodies on 8 (virtual) ranks x 131,072 -- a(0) of ALL bodies
# against the single-GPU deterministic kernel and 2,048 oracle bodies, two steps, the whole run twice bit for bit
def test_config3_sharded_direct_full_size_8x131072(nb, oracle, ctx):
    from nbody_amd.sharded import Comm, ShardedDirect
    W, n, G, eps, dt = 8, 1 << 20, 1.0, 1e-3, 1e-3
    ic = nb.ic.plummer(n, seed=42)
    comm = Comm.init_all(W, [0] * W)
    runs = []
    for _ in range(2):
        sysm = ShardedDirect(comm, n, G, eps)
        sysm.set_state(ic)
        sysm.forces()
        st0 = sysm.get_state(what=("acc",))
        sysm.step(dt, 2)
        runs.append((st0, sysm.get_state()))
        sysm.close()
    a_sh = np.stack([runs[0][0][k] for k in ("acc_x", "acc_y", "acc_z")], 1)
    p = packed(ic)
    eps2 = float(np.float32(eps) * np.float32(eps))
    ctx.deterministic(2)                                     # slot planes required: the deterministic symmetric kernel
    a_one = nb.direct_forces_packed(ctx, p, p, G, eps2).cpu().numpy()[:, :3]
    e = rel_err(a_sh, a_one)
    idx = np.unique(np.concatenate([np.random.default_rng(3).choice(n, 2046, replace=False), [0, n - 1]]))
    orc = np.stack(oracle.direct_forces_indexed(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], idx, G, eps2, 1), 1)
    eo = rel_err(a_sh[idx], orc)
    print(f"config 3, 8 x 131072: a(0) of {n} bodies vs the single-GPU kernel max {e.max():.2e}, {idx.size} oracle bodies max {eo.max():.2e}")
    assert e.max() < TOL and eo.max() < TOL
    for k in runs[0][1]:
        assert np.all(np.isfinite(runs[0][1][k])), k
        assert np.array_equal(runs[0][1][k], runs[1][1][k]), k
    for k in runs[0][0]:
        assert np.array_equal(runs[0][0][k], runs[1][0][k]), k
    comm.close()


def test_virtual_ranks_tiny_softening_takes_the_one_sided_path(nb, oracle, ctx):
    from nbody_amd.sharded import Comm, ShardedDirect
    n, W = 6001, 3
    ic = nb.ic.plummer(n, seed=5)
    comm = Comm.init_all(W, [0] * W)
    sysm = ShardedDirect(comm, n, 1.0, 0.0)      # eps = 0: no pair kernel, rank r sums the gathered shards one-sided
    sysm.set_state(ic)
    sysm.forces()
    st = sysm.get_state()
    a = np.stack([st[k] for k in ("acc_x", "acc_y", "acc_z")], 1)
    ref = np.stack(oracle.direct_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], 1.0, 0.0, 1), 1)
    assert rel_err(a, ref).max() < TOL
    sysm.step(1e-4, 2)
    assert np.all(np.isfinite(sysm.get_state()["pos_x"]))


@pytest.mark.parametrize("how", ["init_rank", "init_all"])
def test_one_rccl_rank(nb, ctx, how):
    """ncclCommInitRank / ncclCommInitAll with the box's one GPU: librccl.so.1 is opened at run time, the step's
    RCCL calls (in-place all-gather; the grouped exchange is empty at W = 1) run, results equal the P2P transport."""
    from nbody_amd.sharded import TRANSPORT_RCCL, Comm, ShardedDirect
    n = 20000
    ic = nb.ic.plummer(n, seed=3)
    if how == "init_rank":
        uid = Comm.unique_id()
        assert len(uid) == 128 and any(uid)
        comm = Comm.init_rank(0, 0, 1, uid)
    else:
        comm = Comm.init_all(1, [0], TRANSPORT_RCCL)
    assert comm.info()["transport"] == TRANSPORT_RCCL and comm.info()["world"] == 1
    a = ShardedDirect(comm, n, 1.0, 0.02)
    a.set_state(ic)
    a.forces()
    a.step(1e-3, 2)
    ms = a.time_steps(1e-3, 1, 3)
    assert 0 < ms < 1000
    st = a.get_state()
    ke, pe = a.energies()
    p2p = Comm.init_all(1, [0])
    b = ShardedDirect(p2p, n, 1.0, 0.02)
    b.set_state(ic)
    b.forces()
    b.step(1e-3, 6)
    st_b = b.get_state()
    for k in st:
        assert np.array_equal(st[k], st_b[k]), k
    assert (ke, pe) == b.energies()
    with pytest.raises(nb.ValidationException):
        Comm.init_all(2, [0, 0], TRANSPORT_RCCL)       # RCCL needs distinct devices
    a.close(); b.close(); comm.close(); p2p.close()


# The plugin form: a multi-rank ForceCalculator behind the reference's interface -- computeForces(ParticleData*)
# on a whole-system ParticleData -- driven by the unchanged Integrator (its generic path: a subclass goes through
# the virtual computeForces, force_calculator.hpp:36-58)
@pytest.mark.parametrize("W", [2, 4, 8])
def test_plugin_form_through_integrator(nb, ctx, W):
    from nbody_amd.sharded import Comm, ShardedDirect
    n, G, eps, dt = 60000, 1.0, 0.05, 1e-3
    ic = nb.ic.plummer(n, seed=W)

    class ShardedCalculator(nb.ForceCalculator):
        def __init__(self):
            super().__init__()
            self.comm = Comm.init_all(W, [0] * W)
            self.sys = ShardedDirect(self.comm, n, G, eps)

        def computeForces(self, d):
            self.sys.compute_forces(d)

        def getMethod(self):
            return nb.ForceMethod.DIRECT_N2
    d1, _ = to_device(nb, ic)
    d2, _ = to_device(nb, ic)
    one = nb.DirectForceCalculator()
    one.setGravitationalConstant(G)
    one.setSofteningParameter(eps)
    many = ShardedCalculator()
    one.computeForces(d1)
    many.computeForces(d2)
    assert rel_err(acc_of(d2), acc_of(d1)).max() < TOL
    integ = nb.Integrator()
    for _ in range(3):
        integ.integrate(d1, one, dt)
        integ.integrate(d2, many, dt)
    torch.cuda.synchronize()
    for k in ("pos_x", "vel_y", "acc_z"):
        assert np.allclose(getattr(d2, k).cpu().numpy(), getattr(d1, k).cpu().numpy(), rtol=2e-5, atol=1e-6), k
    with pytest.raises(nb.ValidationException):
        small, _ = to_device(nb, nb.ic.plummer(100, seed=1))
        many.sys.compute_forces(small)


# ---- BASELINE config 5 behind the C ABI: nbody_hip_sharded_hash_* (csrc/sharded_hash.hip) ------------------------------


def _single_gpu_hash_forces(nb, st, mass, G, eps, cell, cutoff):
    ic = {k: st[k] for k in ("pos_x", "pos_y", "pos_z")}
    ic["mass"] = mass
    for k in ("vel_x", "vel_y", "vel_z"):
        ic[k] = np.zeros_like(mass)
    d, _ = to_device(nb, ic)
    fc = nb.SpatialHashCalculator(cell, cutoff)
    fc.setGravitationalConstant(G)
    fc.setSofteningParameter(eps)
    fc.computeForces(d)
    return acc_of(d), fc.getGrid().getGridDims()


# z-slab shards on 1-8 VIRTUAL ranks of the test GPU, bodies migrating between slabs and halo layers exchanged every
# step.  A trajectory cannot be compared with another run's beyond one step (the cutoff is a discontinuity: a pair one
# ulp inside it in one run and outside in the other changes a velocity by G m dt / r^2), so every step is checked on
# its own: the accelerations the sharded system holds equal the single-GPU spatial hash evaluated on the sharded
# system's OWN positions (same pair set bit for bit -- the distance test is the same arithmetic -- summed in another
# order for bodies near a slab boundary), and positions / velocities follow the Velocity-Verlet update
# (integrator.cu:25-62) from the previous state.  cutoff <= cell and cutoff > cell (the reference's incomplete
# 27-cell search, reproduced); a grid too sparse for the two-grid kernel takes the one-grid path; ranks without
# layers (gz < W)
@pytest.mark.parametrize("W,n,half,cell,cutoff", [(1, 20000, 6.0, 1.0, 1.0), (2, 30000, 6.0, 1.0, 1.0), (3, 30000, 6.0, 1.0, 2.0),
                                                  (4, 60000, 8.0, 1.0, 1.0), (8, 200000, 12.0, 1.0, 1.0), (4, 3000, 20.0, 1.0, 1.0),
                                                  (8, 5000, 2.0, 1.0, 1.0),
                                                  # below 8 bodies per cell the two-grid calls take the one-lane-per-body kernel
                                                  # (layer ranges, accumulate); from 500,000 bodies per rank its split form
                                                  (4, 40000, 9.0, 1.0, 1.0), (2, 1200000, 30.0, 1.0, 1.0)])
def test_sharded_hash_virtual_ranks_equal_single_gpu(nb, oracle, ctx, W, n, half, cell, cutoff):
    from nbody_amd.sharded import Comm, ShardedHash
    G, eps, dt, steps = 1.2, 0.05, 2e-2, 4
    ic = nb.ic.uniform_box(n, seed=100 + W, lo=-half, hi=half, min_mass=0.5, max_mass=1.5)
    rng = np.random.default_rng(W)     # velocities that carry bodies across slab boundaries within a few steps
    for k in ("vel_x", "vel_y", "vel_z"):
        ic[k] = rng.normal(0.0, 3.0, n).astype(np.float32)
    comm = Comm.init_all(W, [0] * W)
    sysm = ShardedHash(comm, n, G, eps, cell, cutoff)
    sysm.set_state(ic)
    sysm.forces()
    prev = sysm.get_state()
    for k in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z"):
        assert np.array_equal(prev[k], ic[k]), k
    eps2 = float(np.float32(eps) * np.float32(eps))
    orc, _, kappa = oracle.spatial_hash_forces_cond(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], G, eps2, cell, cutoff)
    a_sh = np.stack([prev[k] for k in ("acc_x", "acc_y", "acc_z")], 1)
    nz = np.linalg.norm(orc, axis=1) > 0
    assert np.all(a_sh[~nz] == 0)
    assert np.all(rel_err(a_sh[nz], orc[nz]) <= hash_bound(kappa[nz], "oracle"))   # derived: tests/gpu_util.py
    migrated = halo = 0
    for step in range(steps + 1):
        a_one, dims = _single_gpu_hash_forces(nb, prev, ic["mass"], G, eps, cell, cutoff)
        info = sysm.info()
        assert tuple(info["dims"]) == tuple(dims) and sum(info["local_counts"]) == n, (step, info, dims)
        a_sh = np.stack([prev[k] for k in ("acc_x", "acc_y", "acc_z")], 1)
        nz = np.linalg.norm(a_one, axis=1) > 0
        assert np.all(a_sh[~nz] == 0), step
        e = rel_err(a_sh[nz], a_one[nz])
        # two fp32 evaluations of the same terms grouped differently (bodies near a slab boundary): the derived
        # condition-aware criterion of tests/gpu_util.py, kappa from the oracle on the system's own positions
        _, _, kap = oracle.spatial_hash_forces_cond(prev["pos_x"], prev["pos_y"], prev["pos_z"], ic["mass"], G, eps2, cell, cutoff)
        assert np.all(e <= hash_bound(kap[nz], "gpu")), (step, e.max(), hash_margin(e, kap[nz]))
        assert W > 1 or np.mean(e == 0) > 0.5, (step, np.mean(e == 0))   # one rank: the same kernels in the same order
        if step == steps:
            break
        sysm.step(dt, 1)
        cur = sysm.get_state()
        migrated += sysm.info()["migrated"]
        halo += sysm.info()["halo_bodies"]
        hdt = np.float32(0.5 * dt)
        for ax in "xyz":
            vh = prev["vel_" + ax] + prev["acc_" + ax] * hdt
            assert np.allclose(cur["pos_" + ax], prev["pos_" + ax] + vh * np.float32(dt), rtol=3e-7, atol=1e-6), (step, ax)
            # (the kernel forms v + (a_old + a_new) h as add + fma: a few ulps of |v| + |a| h from this two-step form)
            scale = np.abs(vh) + np.abs(cur["acc_" + ax]) * hdt + 1.0
            assert np.all(np.abs(cur["vel_" + ax] - (vh + cur["acc_" + ax] * hdt)) <= 4 * 2.0 ** -23 * scale), (step, ax)
        prev = cur
    if W > 1 and n >= 20000:
        assert migrated > 0 and halo > 0
    if half == 20.0:
        assert not sysm.info()["two_grid"]                  # 41^3 cells for 3,000 bodies: the one-grid path
    elif n >= 20000:
        assert sysm.info()["two_grid"]
    # reproducible run after run, and step(dt, k) == k x step(dt, 1)
    again = ShardedHash(comm, n, G, eps, cell, cutoff)
    again.set_state(ic)
    again.forces()
    again.step(dt, steps)
    st2 = again.get_state()
    for k in prev:
        assert np.array_equal(prev[k], st2[k]), k
    again.close(); sysm.close(); comm.close()


# BASELINE config 5 at its full size: N = 4,194,304 bodies in the uniform box on 8 (virtual) ranks x 524,288 -- the first
# evaluation against the oracle on EVERY body, then three steps, each checked like the smaller cases above (accelerations
# against the single-GPU spatial hash on the system's own positions under the derived criterion, Velocity-Verlet update)
def test_config5_sharded_hash_full_size_8x524288(nb, oracle, ctx):
    from nbody_amd.sharded import Comm, ShardedHash
    W, n, G, eps, cell, cutoff, dt, steps = 8, 4194304, 1.0, 0.01, 1.0, 1.0, 1e-3, 3
    half = 0.5 * (n / 16.0) ** (1.0 / 3.0)                    # 16 bodies per unit volume
    ic = nb.ic.uniform_box(n, seed=42, lo=-half, hi=half)
    rng = np.random.default_rng(5)
    for k in ("vel_x", "vel_y", "vel_z"):                     # bodies cross slab boundaries within the three steps
        ic[k] = rng.normal(0.0, 20.0, n).astype(np.float32)
    comm = Comm.init_all(W, [0] * W)
    sysm = ShardedHash(comm, n, G, eps, cell, cutoff)
    sysm.set_state(ic)
    sysm.forces()
    prev = sysm.get_state()
    eps2 = float(np.float32(eps) * np.float32(eps))
    migrated = 0
    for step in range(steps + 1):
        orc, _, kappa = oracle.spatial_hash_forces_cond(prev["pos_x"], prev["pos_y"], prev["pos_z"], ic["mass"], G, eps2, cell, cutoff)
        a_sh = np.stack([prev[k] for k in ("acc_x", "acc_y", "acc_z")], 1)
        nz = np.linalg.norm(orc, axis=1) > 0
        assert np.all(a_sh[~nz] == 0), step
        e = rel_err(a_sh[nz], orc[nz])
        assert np.all(e <= hash_bound(kappa[nz], "oracle")), (step, e.max(), hash_margin(e, kappa[nz]))
        a_one, dims = _single_gpu_hash_forces(nb, prev, ic["mass"], G, eps, cell, cutoff)
        info = sysm.info()
        assert tuple(info["dims"]) == tuple(dims) and sum(info["local_counts"]) == n and info["two_grid"], (step, info)
        e1 = rel_err(a_sh[nz], a_one[nz])
        assert np.all(e1 <= hash_bound(kappa[nz], "gpu")), (step, e1.max(), hash_margin(e1, kappa[nz]))
        print(f"config 5, 8 x 524288, step {step}: every body vs the oracle max {e.max():.2e} (margin err / (u kappa) "
              f"{hash_margin(e, kappa[nz]):.2f}), vs the single-GPU grid max {e1.max():.2e}; per rank {info['local_counts']}")
        if step == steps:
            break
        sysm.step(dt, 1)
        cur = sysm.get_state()
        migrated += sysm.info()["migrated"]
        hdt = np.float32(0.5 * dt)
        for ax in "xyz":
            vh = prev["vel_" + ax] + prev["acc_" + ax] * hdt
            assert np.allclose(cur["pos_" + ax], prev["pos_" + ax] + vh * np.float32(dt), rtol=3e-7, atol=1e-5), (step, ax)
            scale = np.abs(vh) + np.abs(cur["acc_" + ax]) * hdt + 1.0
            assert np.all(np.abs(cur["vel_" + ax] - (vh + cur["acc_" + ax] * hdt)) <= 4 * 2.0 ** -23 * scale), (step, ax)
        prev = cur
    assert migrated > 0
    sysm.close(); comm.close()


def test_sharded_hash_one_rccl_rank_and_errors(nb, ctx):
    from nbody_amd.sharded import TRANSPORT_RCCL, Comm, ShardedHash
    n = 40000
    ic = nb.ic.uniform_box(n, seed=7, lo=-7.0, hi=7.0)
    comm = Comm.init_all(1, [0], TRANSPORT_RCCL)     # the RCCL all-reduces of the box and of the counts with one rank
    a = ShardedHash(comm, n, 1.0, 0.05, 1.0, 1.0)
    a.set_state(ic)
    a.forces()
    a.step(1e-2, 3)
    ms = a.time_steps(1e-2, 1, 3)
    assert 0 < ms < 100
    p2p = Comm.init_all(1, [0])
    b = ShardedHash(p2p, n, 1.0, 0.05, 1.0, 1.0)
    b.set_state(ic)
    b.forces()
    b.step(1e-2, 7)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    with pytest.raises(nb.ValidationException):
        ShardedHash(p2p, n, 1.0, 0.05, 0.0, 1.0)
    far = dict(ic)
    far["pos_x"] = ic["pos_x"].copy()
    far["pos_x"][0] = 1.0e6                          # > 1e8 cells: "grid too large", as the reference's build
    c = ShardedHash(p2p, n, 1.0, 0.05, 1.0, 1.0)
    with pytest.raises(nb.ResourceException, match="too large"):
        c.set_state(far)
    a.close(); b.close(); c.close(); comm.close(); p2p.close()


# a soak through the clumping of a uniform box (strong gravity, so that it happens within a few hundred steps): crowded
# cells, an expanding grid, bodies migrating every step, the unit form of the wave-per-cell kernel inside the two-grid
# calls and the one-grid fallback once the slabs are too sparse -- every 60 steps the accelerations the sharded system
# holds against the single-GPU spatial hash on the system's own positions (tools/sharded_hash_soak.py is the long form)
def test_sharded_hash_soak_through_clumping(nb, oracle, ctx):
    from nbody_amd.sharded import Comm, ShardedHash
    W, n, G, eps, dt = 3, 60000, 8.0, 0.02, 1e-3
    h = 0.5 * (n / 16.0) ** (1 / 3)
    ic = nb.ic.uniform_box(n, seed=4, lo=-h, hi=h)
    comm = Comm.init_all(W, [0] * W)
    sysm = ShardedHash(comm, n, G, eps, 1.0, 1.0)
    sysm.set_state(ic)
    sysm.forces()
    crowded, migrated, fallback = 0, 0, False
    for _ in range(5):
        sysm.step(dt, 60)
        st = sysm.get_state()
        a1, dims = _single_gpu_hash_forces(nb, st, ic["mass"], G, eps, 1.0, 1.0)
        a = np.stack([st["acc_x"], st["acc_y"], st["acc_z"]], 1)
        nz = np.linalg.norm(a1, axis=1) > 0
        assert np.all(a[~nz] == 0)
        _, _, kap = oracle.spatial_hash_forces_cond(st["pos_x"], st["pos_y"], st["pos_z"], ic["mass"], G,
                                                    float(np.float32(eps) * np.float32(eps)), 1.0, 1.0)
        e = rel_err(a[nz], a1[nz])
        assert np.all(e <= hash_bound(kap[nz], "gpu")), (e.max(), hash_margin(e, kap[nz]))   # derived: tests/gpu_util.py
        info = sysm.info()
        assert tuple(info["dims"]) == tuple(dims) and sum(info["local_counts"]) == n
        migrated += info["migrated"]
        fallback = fallback or not info["two_grid"]
        key = (np.floor((st["pos_x"] - st["pos_x"].min()) / 1.0).astype(np.int64) * 1000003
               + np.floor((st["pos_y"] - st["pos_y"].min()) / 1.0).astype(np.int64)) * 1000003 \
            + np.floor((st["pos_z"] - st["pos_z"].min()) / 1.0).astype(np.int64)
        crowded = max(crowded, int(np.unique(key, return_counts=True)[1].max()))
    assert migrated > 0 and crowded > 64, (migrated, crowded)   # the regime the unit form is for was reached
    sysm.close(); comm.close()
