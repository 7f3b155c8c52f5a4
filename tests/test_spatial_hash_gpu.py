"""Spatial-hash path (through the C ABI) against the oracle and the reference's own tests
(tests/test_spatial_hash.cpp).  The reference pins this path only structurally (partition
property, non-zero forces); numeric parity vs the oracle's restatement of
force_spatial_hash.cu:83-152 is added here (tolerance 1e-5 relative per body, fp32)."""
import numpy as np
import pytest
import torch

from gpu_util import HASH_C, hash_bound, hash_margin, acc_of, rel_err, to_device

pytestmark = pytest.mark.gpu
TOL = 1e-5
KERNELS = (1, 2, 3, 4, 7, 8, 9, 10)   # force kernels of the grid (nbody_hip_grid_tuning): cell runs; one wave per cell with 1 / 2 / 4 bodies per lane;
                            # 7 = the two-phase form (distance masks, then the accepted candidates only); 8 = one lane per body; 9 = that for the light cells, a wave per crowded cell (the automatic form below 8 bodies per cell); 10 = two bodies of a cell per lane


# tests/test_spatial_hash.cpp:15-36 GridConstruction + :53-83 ComputeForces
def test_grid_construction_and_finite_forces(nb, ctx):
    ic = nb.ic.sphere(100, seed=42, radius=5.0)
    d, h = to_device(nb, ic)
    grid = nb.SpatialHashGrid(100, 1.0)
    grid.build(d)
    assert grid.getTotalCells() > 0
    calc = nb.SpatialHashCalculator(1.0, 2.0)
    calc.setGravitationalConstant(1.0)
    calc.setSofteningParameter(0.1)
    calc.computeForces(d)
    assert np.all(np.isfinite(acc_of(d)))
    assert calc.getMethod() == nb.ForceMethod.SPATIAL_HASH


# tests/test_spatial_hash.cpp:38-51 CellIndexCalculation
def test_cell_index_kats(nb):
    assert nb.SpatialHashGrid.getCellIndex(0.5, 0.5, 0.5, 2.0) == (0, 0, 0)
    assert nb.SpatialHashGrid.getCellIndex(2.5, 4.5, 6.5, 2.0) == (1, 2, 3)


# tests/test_spatial_hash.cpp:89-130: every particle in exactly one cell, inside its bounds
@pytest.mark.parametrize("n,cell,seed", [(50, 1.0, 1), (500, 2.0, 2), (3000, 0.7, 3)])
def test_cell_assignment_property(nb, oracle, ctx, n, cell, seed):
    rng = np.random.default_rng(seed)
    ic = nb.ic.uniform_box(n, seed=seed, lo=-10, hi=10)
    d, h = to_device(nb, ic)
    grid = nb.SpatialHashGrid(n, cell)
    grid.build(d)
    cs, ce, pc, si = grid.copyCellDataToHost()
    dims, total = grid.getGridDims(), grid.getTotalCells()
    # grid geometry equals the oracle's restatement of build() (:225-246)
    lo, hi, odims = oracle.hash_grid(ic["pos_x"], ic["pos_y"], ic["pos_z"], cell)
    assert list(dims) == odims and total == odims[0] * odims[1] * odims[2]
    glo, ghi = grid.getBoundingBox()
    assert np.allclose(glo, lo, atol=0) and np.allclose(ghi, hi, atol=0)
    # bit-exact cell ids
    import oracle_bind as ob
    ocells = np.empty(n, np.int32)
    oracle.L.oracle_assign_cells(n, ic["pos_x"], ic["pos_y"], ic["pos_z"], ob._f3(*lo), cell,
                                 ob._i3(*odims), ocells)
    assert np.array_equal(pc, ocells)
    # partition: sorted_indices is a permutation, each body appears in the range of its own cell
    assert np.array_equal(np.sort(si), np.arange(n))
    counts = ce - cs
    assert counts.sum() == n and np.all(counts >= 0)
    for c in np.nonzero(counts)[0][:200]:
        members = si[cs[c]:ce[c]]
        assert np.all(pc[members] == c)
        assert np.all(np.diff(members) > 0)  # stable sort: index order inside a cell
    assert grid.verifyCellAssignment(h)


# tests/test_spatial_hash.cpp:134-182: some force is non-zero when bodies are within the cutoff
def test_forces_nonzero_property(nb, ctx):
    ic = nb.ic.uniform_box(64, seed=9, lo=-1, hi=1)
    d, _ = to_device(nb, ic)
    calc = nb.SpatialHashCalculator(1.0, 2.0)
    calc.setSofteningParameter(0.1)
    calc.computeForces(d)
    a = acc_of(d)
    assert np.all(np.isfinite(a)) and np.abs(a).max() > 0


# numeric parity with the oracle: cutoff <= cell (complete search), cutoff > cell (the
# reference's 27-cell search misses pairs: reproduced), tiny eps, and dense cells
@pytest.mark.parametrize("n,box,cell,cutoff,eps,tol", [
    (4000, 6.0, 1.0, 1.0, 0.01, TOL),
    (4000, 6.0, 1.0, 2.0, 0.01, TOL),     # reference defaults: cutoff 2 > cell 1
    (3000, 4.0, 0.5, 0.75, 0.1, TOL),
    (2000, 2.0, 1.5, 1.0, 0.0, TOL),      # eps = 0: guard variant
    # ~190 bodies per cell and eps far below the inter-body distance: ~5000 fp32 terms of size
    # up to 400 per body cancelling to |a| ~ 100; fp32 partial sums (folded to fp64 every 64
    # sources) limit agreement with the fp64-accumulated oracle to a few 1e-6 typical, 2e-5 max
    (5000, 1.5, 1.0, 1.0, 0.05, 2e-5),
    (257, 30.0, 1.0, 1.0, 0.01, TOL),     # sparse: most cells empty
])
def test_forces_match_oracle(nb, oracle, ctx, n, box, cell, cutoff, eps, tol):
    ic = nb.ic.uniform_box(n, seed=n, lo=-box, hi=box, min_mass=0.5, max_mass=1.5)
    d, _ = to_device(nb, ic)
    calc = nb.SpatialHashCalculator(cell, cutoff)
    calc.setGravitationalConstant(1.3)
    calc.setSofteningParameter(eps)
    calc.computeForces(d)
    a = acc_of(d)
    eps2 = float(np.float32(eps) * np.float32(eps))
    ref = np.stack(oracle.spatial_hash_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"],
                                              1.3, eps2, cell, cutoff), 1)
    nz = np.linalg.norm(ref, axis=1) > 0
    assert np.all(a[~nz] == 0)
    if nz.any():
        e = rel_err(a[nz], ref[nz])
        assert e.max() < tol and np.median(e) < 1e-6
    # determinism: a second evaluation is bitwise identical (stable binning)
    calc.computeForces(d)
    assert np.array_equal(acc_of(d), a)
    # every force kernel of the grid against the oracle (1 = cell runs with binary-searched ranges,
    # 2 / 3 / 4 = one wave per cell with 1 / 2 / 4 bodies per lane; the dense case has windows longer than
    # a wave's LDS batch and cells with more bodies than one chunk of target slots)
    g = calc.getGrid()
    for kern in KERNELS:
        g.tuning(kern)
        calc.computeForces(d)
        ak = acc_of(d)
        assert np.all(ak[~nz] == 0), kern
        if nz.any():
            assert rel_err(ak[nz], ref[nz]).max() < tol, kern
    g.tuning(0)


# clustered input (Plummer): very uneven cell occupancy
def test_plummer_clustered(nb, oracle, ctx):
    n = 20000
    ic = nb.ic.plummer(n, seed=4)
    d, _ = to_device(nb, ic)
    calc = nb.SpatialHashCalculator(0.5, 0.5)
    calc.setSofteningParameter(0.01)
    calc.computeForces(d)
    a = acc_of(d)
    ref = np.stack(oracle.spatial_hash_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"],
                                              1.0, float(np.float32(0.01) ** 2), 0.5, 0.5), 1)
    nz = np.linalg.norm(ref, axis=1) > 0
    assert rel_err(a[nz], ref[nz]).max() < TOL
    # and equals the direct sum restricted to the cutoff (27-cell search is complete here)
    idx = np.arange(0, n, 37)
    dc = np.stack(oracle.direct_cutoff_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], idx,
                                              1.0, float(np.float32(0.01) ** 2), 0.5), 1)
    keep = np.linalg.norm(dc, axis=1) > 0
    assert rel_err(a[idx][keep], dc[keep]).max() < TOL


def test_error_behaviour(nb, ctx):
    with pytest.raises(nb.ValidationException):
        nb.SpatialHashGrid(100, 0.0)
    # > 1e8 cells (force_spatial_hash.cu:252-254)
    ic = nb.ic.uniform_box(100, seed=1, lo=-500, hi=500)
    d, _ = to_device(nb, ic)
    grid = nb.SpatialHashGrid(100, 1.0)
    with pytest.raises(nb.ResourceException):
        grid.build(d)
    # capacity is fixed at creation (ref: sized from the first count, :372-374)
    big, _ = to_device(nb, nb.ic.uniform_box(200, seed=2, lo=-5, hi=5))
    with pytest.raises(nb.ValidationException):
        grid.build(big)
    # factory
    cfg = nb.SimulationConfig(force_method=nb.ForceMethod.SPATIAL_HASH, spatial_hash_cell_size=2.0,
                              spatial_hash_cutoff=1.5, softening=0.05, G=2.0)
    calc = nb.createForceCalculator(cfg.force_method, cfg)
    assert isinstance(calc, nb.SpatialHashCalculator)
    assert (calc.getCellSize(), calc.getCutoffRadius()) == (2.0, 1.5)
    assert calc.getGravitationalConstant() == 2.0 and calc.getSofteningParameter() == pytest.approx(0.05)


# BASELINE config 5 shape on one GPU at reduced size + the full per-GPU share
@pytest.mark.parametrize("n,half", [(524288, 16.0)])
def test_uniform_box_16_per_cell(nb, oracle, ctx, n, half):
    ic = nb.ic.uniform_box(n, seed=42, lo=-half, hi=half)  # 16 bodies per unit volume
    d, _ = to_device(nb, ic)
    calc = nb.SpatialHashCalculator(1.0, 1.0)
    calc.setSofteningParameter(0.01)
    calc.computeForces(d)
    a = acc_of(d)
    assert np.all(np.isfinite(a))
    idx = np.arange(0, n, 997)
    dc = np.stack(oracle.direct_cutoff_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], idx,
                                              1.0, float(np.float32(0.01) ** 2), 1.0), 1)
    assert rel_err(a[idx], dc).max() < TOL


# the sharded (z-slab) path's building blocks on one GPU: global-box binning, cell_z, and
# [own; halo] evaluation == the single-grid result for every slab of a 4-way split
@pytest.mark.parametrize("cutoff", [1.0, 2.0])
def test_packed_slabs_equal_whole(nb, oracle, ctx, cutoff):
    from nbody_amd.distributed import HipBackend, ShardedHashSystem, layer_owner
    n, cell, eps, G = 20000, 1.0, 0.05, 1.2
    ic = nb.ic.uniform_box(n, seed=21, lo=-6.0, hi=6.0, min_mass=0.5, max_mass=1.5)
    d, _ = to_device(nb, ic)
    calc = nb.SpatialHashCalculator(cell, cutoff)
    calc.setGravitationalConstant(G)
    calc.setSofteningParameter(eps)
    calc.computeForces(d)
    whole = acc_of(d)
    be = HipBackend(ctx)
    p = torch.from_numpy(np.ascontiguousarray(
        np.stack([ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]], 1))).cuda()
    bb = be.bbox(p).cpu().numpy()
    assert np.array_equal(bb[:3], [ic[k].min() for k in ("pos_x", "pos_y", "pos_z")])
    assert np.array_equal(bb[3:], [ic[k].max() for k in ("pos_x", "pos_y", "pos_z")])
    sysm = ShardedHashSystem(ic, G, eps, cell, cutoff, backend=be)  # world 1: grid helpers only
    lo, hi, dims = sysm._grid_from_bounds(bb.tolist())
    glo, ghi = calc.getGrid().getBoundingBox()
    assert np.allclose(lo, glo, atol=0) and np.allclose(hi, ghi, atol=0)
    assert tuple(dims) == calc.getGrid().getGridDims()
    cz = be.cell_z(p, lo[2], cell, dims[2]).cpu().numpy()
    _, _, pc, _ = calc.getGrid().copyCellDataToHost()
    assert np.array_equal(cz, pc // (dims[0] * dims[1]))
    owner = layer_owner(dims[2], 4)
    for r in range(4):
        layers = np.nonzero(owner == r)[0]
        own = np.nonzero(owner[cz] == r)[0]
        halo = np.nonzero((cz == layers[0] - 1) | (cz == layers[-1] + 1))[0]
        allb = torch.cat([p[torch.from_numpy(own).cuda()], p[torch.from_numpy(halo).cuda()]]).contiguous()
        acc = be.hash_forces(allb, lo + hi, cell, cutoff, G, eps).cpu().numpy()[: own.size, :3]
        nz = np.linalg.norm(whole[own], axis=1) > 0
        assert rel_err(acc[nz], whole[own][nz]).max() < TOL  # summation order differs per slab
        assert np.all(acc[~nz] == 0)
        # the overlapped form: own x own on the slab's grid, then its two boundary layers against a grid
        # of the halo layers, accumulated (nbody_hip_grid_forces_pair_packed); cell order of the own grid
        # starts with the lowest and ends with the highest layer (what the halo exchange sends)
        z0, z1 = int(layers[0]), int(layers[-1]) + 1
        own_t = p[torch.from_numpy(own).cuda()].contiguous()
        halo_t = p[torch.from_numpy(halo).cuda()].contiguous()
        be.grid_build("own", own_t, lo + hi, cell, z0, z1 - z0)
        acc2 = torch.full((own.size, 4), 7.0, device="cuda")
        assert be.grid_forces("own", "own", z0, z1 - z0, cutoff, G, eps, acc2, False)
        if halo.size:
            be.grid_build("halo", halo_t, lo + hi, cell, max(z0 - 1, 0), min(z1 + 1, dims[2]) - max(z0 - 1, 0))
            for z in sorted({z0, z1 - 1}):
                assert be.grid_forces("own", "halo", z, 1, cutoff, G, eps, acc2, True)
        a2 = acc2.cpu().numpy()
        assert np.all(a2[:, 3] == 0)
        assert rel_err(a2[nz, :3], whole[own][nz]).max() < TOL
        n_head, n_tail = int((cz[own] == z0).sum()), int((cz[own] == z1 - 1).sum())
        head = torch.empty((n_head, 4), device="cuda")
        tail = torch.empty((n_tail, 4), device="cuda")
        be.grid_sorted("own", 0, n_head, head)
        be.grid_sorted("own", own.size - n_tail, n_tail, tail)
        cz_of = lambda t: np.clip(np.floor((t.cpu().numpy()[:, 2] - np.float32(lo[2])) / np.float32(cell)), 0, dims[2] - 1)  # noqa: E731
        assert np.all(cz_of(head) == z0) and np.all(cz_of(tail) == z1 - 1)
    # the partition pass (csrc/slab.hip) against its numpy restatement: rows grouped by owner in input
    # order, send-matrix row, layer histogram, grid size
    W, rk = 4, 1
    vel = torch.from_numpy(np.random.default_rng(1).normal(size=(n, 4)).astype(np.float32)).cuda()
    acc0 = torch.from_numpy(np.random.default_rng(2).normal(size=(n, 4)).astype(np.float32)).cuda()
    gid = torch.arange(n, dtype=torch.int32, device="cuda") * 3
    rows, holes, stats, info = be.slab_partition(p, vel, acc0, gid, torch.from_numpy(bb).cuda(), cell, W, rk, 4096)
    rows, holes, stats, info = rows.cpu().numpy(), holes.cpu().numpy(), stats.cpu().numpy(), info.cpu().numpy()
    assert list(info) == [dims[0], dims[1], dims[2], 0]
    dest = owner[cz]
    leave = np.nonzero(dest != rk)[0]
    order = leave[np.argsort(dest[leave], kind="stable")]
    L = leave.size
    assert 0 < L < n
    assert np.array_equal(stats[rk * W:(rk + 1) * W], np.bincount(dest, minlength=W))
    assert stats[:W * W].sum() == n
    assert np.array_equal(stats[W * W:W * W + dims[2]], np.bincount(cz, minlength=dims[2]))
    assert stats[W * W + dims[2]:].sum() == 0
    pn, vn, an = p.cpu().numpy(), vel.cpu().numpy(), acc0.cpu().numpy()
    assert np.array_equal(holes[:L], leave)
    assert np.array_equal(rows[:L, 0:4], pn[order])
    assert np.array_equal(rows[:L, 4:7], vn[order, :3]) and np.all(rows[:L, 7] == 0)
    assert np.array_equal(rows[:L, 8:11], an[order, :3])
    assert np.array_equal(rows[:L, 12].view(np.int32), 3 * order) and np.array_equal(rows[:L, 13].view(np.int32), cz[order])
    # the fill pass: fewer, equally many and more arrivals than vacated slots
    rng = np.random.default_rng(5)
    for A in (L // 3, L, L + 1000, 0):
        arr = rng.normal(size=(A, 16)).astype(np.float32)
        arr[:, 12] = (np.arange(A, dtype=np.int32) + 10 ** 6).view(np.float32)
        cap = n + 2000
        bufs = [torch.zeros((cap, 4), device="cuda") for _ in range(3)]
        for t, src in zip(bufs, (pn, vn, an)):
            t[:n] = torch.from_numpy(src).cuda()
        gbuf = torch.zeros(cap, dtype=torch.int32, device="cuda")
        gbuf[:n] = gid
        be.slab_fill(bufs[0], bufs[1], bufs[2], gbuf, n, torch.from_numpy(holes).cuda(), L, torch.from_numpy(arr).cuda())
        n_new = n - L + A
        got_ids = gbuf[:n_new].cpu().numpy()
        keep = np.setdiff1d(np.arange(n), leave)
        assert np.array_equal(np.sort(got_ids), np.sort(np.concatenate([3 * keep, np.arange(A) + 10 ** 6])))  # a permutation
        gp, gv = bufs[0][:n_new].cpu().numpy(), bufs[1][:n_new].cpu().numpy()
        stay = got_ids < 10 ** 6
        assert np.array_equal(gp[stay], pn[got_ids[stay] // 3]) and np.array_equal(gv[stay], vn[got_ids[stay] // 3])
        came = got_ids[~stay] - 10 ** 6
        assert np.array_equal(gp[~stay], arr[came, 0:4]) and np.array_equal(gv[~stay][:, :3], arr[came, 4:7])
        # bodies that stayed below the new end did not move
        low = keep[keep < n_new]
        assert np.array_equal(got_ids[low], 3 * low)
    # world-1 system: a full step equals Integrator.integrate with the SoA calculator
    sysm.initial_forces()
    sysm.step(1e-3)
    assert sysm.path == "two-grid"
    integ = nb.Integrator()
    integ.integrate(d, calc, 1e-3)
    gid, pos, vel, acc = sysm.gather_global()
    assert np.allclose(pos[:, 0], d.pos_x.cpu().numpy(), rtol=1e-6, atol=1e-6)
    assert np.allclose(vel[:, 1], d.vel_y.cpu().numpy(), rtol=1e-5, atol=1e-6)


# The per-body bound of the whole-population comparisons: max(1e-5, C u kappa_i) with C DERIVED from the arithmetic of the
# two sides (tests/gpu_util.py) -- the strict 1e-5 of SURVEY section 7 wherever the worst case of
# fp32 terms and fp32 partial sums stays below it, the worst case itself beyond.  Nothing here is fitted: the measured
# margin (largest err / (u kappa) among the bodies above 1e-5; 1.1-2.2 in round 3) is printed with every comparison.
def assert_every_body(tag, a, ref, kappa, gold=None):
    nz = np.linalg.norm(ref, axis=1) > 0
    assert np.all(a[~nz] == 0), tag
    e = rel_err(a[nz], ref[nz])
    k = kappa[nz]
    bound = hash_bound(k, "oracle")
    worst = int(np.argmax(e / bound))
    over = e > TOL
    msg = (f"{tag}: {nz.sum()} bodies, max {e.max():.3e}, p99.99 {np.quantile(e, 0.9999):.3e}, median {np.median(e):.3e}, "
           f"above 1e-5: {over.sum()} (their kappa >= {k[over].min() if over.any() else 0:.0f}), measured margin "
           f"max err / (u kappa) = {hash_margin(e, k):.2f} against the derived C = {HASH_C['oracle']}, "
           f"worst vs bound: err {e[worst]:.3e} kappa {k[worst]:.0f} bound {bound[worst]:.3e}")
    print(msg)
    assert np.all(e <= bound), msg
    if gold is not None:                            # and the fp64 evaluation of the same pair set
        eg = rel_err(a[nz], gold[nz])
        assert np.all(eg <= hash_bound(k, "gold")), (tag, eg.max(), hash_margin(eg, k))
    return msg


# BASELINE config 5 at full size on one GPU (N = 4,194,304, 16 bodies per unit volume): EVERY body against the
# oracle's 27-cell search (oracle.spatial_hash_forces_cond: ~2 s on the box), for the default kernel and for each
# force kernel of the grid
def test_full_size_uniform_box(nb, oracle, ctx):
    n, half = 4194304, 32.0
    ic = nb.ic.uniform_box(n, seed=42, lo=-half, hi=half)
    d, _ = to_device(nb, ic)
    calc = nb.SpatialHashCalculator(1.0, 1.0)
    calc.setSofteningParameter(0.01)
    calc.computeForces(d)
    a = acc_of(d)
    assert np.all(np.isfinite(a))
    g = calc.getGrid()
    assert g.getGridDims() == (66, 66, 66) or g.getGridDims() == (65, 65, 65)
    ref, gold, kappa = oracle.spatial_hash_forces_cond(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], 1.0,
                                                       float(np.float32(0.01) ** 2), 1.0, 1.0)
    assert_every_body(f"N = {n} default kernel", a, ref, kappa, gold)
    # ... and equals the direct sum restricted to the cutoff on a sample (the 27-cell search is complete here)
    idx = np.linspace(0, n - 1, 1024).astype(np.int64)
    dc = np.stack(oracle.direct_cutoff_forces(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], idx, 1.0,
                                              float(np.float32(0.01) ** 2), 1.0), 1)
    assert np.array_equal(dc, ref[idx])
    # every force kernel of the grid gives the reference's 27-cell result, on every body
    g.build(d)
    for kern in KERNELS:
        g.tuning(kern)
        g.computeForces(d, 1.0, 1.0, 0.01)
        assert_every_body(f"N = {n} kernel {kern}", acc_of(d), ref, kappa)
    g.tuning(0)
    # short-range forces of a uniform medium cancel on average: the mean is far below the rms
    assert np.abs(a.mean(0)).max() < 0.02 * a.std(0).min()


# the same box with cutoff 2 > cell (the reference's default ratio: its 27-cell search misses pairs, reproduced) and
# a dense one (107 bodies per cell: windows longer than a wave's LDS batch), every body
@pytest.mark.parametrize("n,half,cutoff", [(4194304, 32.0, 2.0), (2000000, 13.0, 1.0)])
def test_full_size_other_regimes(nb, oracle, ctx, n, half, cutoff):
    ic = nb.ic.uniform_box(n, seed=42, lo=-half, hi=half)
    d, _ = to_device(nb, ic)
    g = nb.SpatialHashGrid(n, 1.0)
    g.build(d)
    ref, gold, kappa = oracle.spatial_hash_forces_cond(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], 1.0,
                                                       float(np.float32(0.01) ** 2), 1.0, cutoff)
    for kern in (0, 1):
        g.tuning(kern)
        g.computeForces(d, cutoff, 1.0, 0.01)
        assert_every_body(f"N = {n} cutoff {cutoff} kernel {kern}", acc_of(d), ref, kappa, gold if kern == 0 else None)
    g.tuning(0)


# BASELINE config 5 in small, against the committed golden vectors (tests/golden/make_golden.py): per body
def test_golden_uniform_4096(nb, oracle, ctx):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "uniform4096_spatial_hash.npz"))
    ic = {k: g[k] for k in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "mass")}
    d, _ = to_device(nb, ic)
    eps = np.float32(g["eps"])
    for name, cutoff in (("acc_c1", 1.0), ("acc_c2", 2.0)):
        calc = nb.SpatialHashCalculator(float(g["cell"]), cutoff)
        calc.setGravitationalConstant(float(g["G"]))
        calc.setSofteningParameter(float(g["eps"]))
        calc.computeForces(d)
        ref, _, kappa = oracle.spatial_hash_forces_cond(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], float(g["G"]),
                                                        float(eps * eps), float(g["cell"]), cutoff)
        assert np.array_equal(ref, g[name])   # the committed vectors are what the oracle computes
        assert_every_body(f"golden {name}", acc_of(d), g[name], kappa)
        assert list(calc.getGrid().getGridDims()) == list(g["dims"])
    cs, ce, pc, si = calc.getGrid().copyCellDataToHost()
    assert np.array_equal(pc, g["cell_of"])   # every body in the cell the oracle puts it in


# A blown-up system: bodies at +-inf (or +-3e38) on all three axes (NaN positions do not widen the box:
# min / max ignore them, here as in the reference's fminf / fmaxf reduction).  Every axis saturates at 2^30 cells; the
# running product must saturate too (2^30 * 2^30 * 2^30 wraps a 64-bit product to 0, which would slip
# under the 1e8-cell limit): "Spatial hash grid too large" (ref: force_spatial_hash.cu:252-254).
@pytest.mark.parametrize("bad", [np.inf, -np.inf, 3.0e38])
def test_grid_too_large_with_non_finite_positions(nb, ctx, bad):
    ic = nb.ic.uniform_box(1000, seed=9, lo=-4.0, hi=4.0)
    for k in ("pos_x", "pos_y", "pos_z"):
        ic[k][7] = np.float32(bad)
        ic[k][11] = np.float32(-bad)
    d, _ = to_device(nb, ic)
    calc = nb.SpatialHashCalculator(1.0, 1.0)
    with pytest.raises(nb.NBodyError, match="too large"):
        calc.computeForces(d)
    # explicit-bounds form used by the sharded path
    import ctypes as C
    from gpu_util import HASH_C, hash_bound, hash_margin, packed
    from nbody_amd._lib import check
    grid = nb.SpatialHashGrid(1000, 1e-3)
    p = packed(ic)
    bounds = (C.c_float * 6)(-1e38, -1e38, -1e38, 1e38, 1e38, 1e38)  # finite extent, 2^30 cells per axis
    with pytest.raises(nb.NBodyError, match="too large"):
        check(ctx._lib.nbody_hip_grid_build_packed(grid._h, p.data_ptr(), 1000, bounds))


# ---- the unit form of the wave-per-cell kernel (occupied cells in chunks of 64 R bodies; csrc/spatial_hash.hip) --------
def _clumpy_box(nb, n, seed):
    """a uniform box with dense knots in it: cells of several hundred bodies beside many empty ones"""
    rng = np.random.default_rng(seed)
    half = 12.0
    ic = nb.ic.uniform_box(n, seed=seed, lo=-half, hi=half, min_mass=0.5, max_mass=1.5)
    knots = rng.uniform(-half + 1, half - 1, (12, 3))
    k = n // 3
    which = rng.integers(0, len(knots), k)
    off = rng.normal(0.0, 0.35, (k, 3))
    for a, key in enumerate(("pos_x", "pos_y", "pos_z")):
        ic[key][:k] = (knots[which, a] + off[:, a]).astype(np.float32)
    return ic


# Every body against the oracle with the unit form forced (NBH_HASH_UNITS=2), with the cell-range form (=0) and with
# the automatic choice, which takes the list from the second call on (crowded cells seen by the first): identical
# arithmetic order in all three, so the results are equal bit for bit, chunked cells included.
@pytest.mark.parametrize("cell,cutoff", [(1.0, 1.0), (1.0, 1.7)])
def test_unit_form_of_the_cell_kernel_on_clumped_bodies(nb, oracle, ctx, monkeypatch, cell, cutoff):
    n = 60000
    ic = _clumpy_box(nb, n, 5)
    eps = 0.05
    ref, gold, kappa = oracle.spatial_hash_forces_cond(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], 1.0,
                                                       float(np.float32(eps) ** 2), cell, cutoff)
    got = {}
    for mode in ("2", "0", "1"):
        monkeypatch.setenv("NBH_HASH_UNITS", mode)
        d, _ = to_device(nb, ic)
        calc = nb.SpatialHashCalculator(cell, cutoff)
        calc.setSofteningParameter(eps)
        calc.computeForces(d)                      # (the grid is made here, with the mode of the environment)
        for kernel in (3, 2, 6, 7):                # two targets per lane (chunks of 128), one (chunks of 64), filtered, two-phase
            calc.getGrid().tuning(kernel)
            calc.computeForces(d)                  # (automatic mode: this call has seen the first one's statistics)
            got[mode, kernel] = acc_of(d)
        counts = np.diff(np.stack(calc.getGrid().copyCellDataToHost()[:2]), axis=0)[0]
        assert counts.max() > 300 and (counts == 0).mean() > 0.15      # chunked cells and empty cells are both there
    for kernel in (3, 2, 6, 7):
        assert np.array_equal(got["2", kernel], got["0", kernel]) and np.array_equal(got["1", kernel], got["0", kernel])
    nz = np.linalg.norm(ref, axis=1) > 0
    a = got["2", 3]
    assert np.all(a[~nz] == 0)
    e = rel_err(a[nz], ref[nz])
    assert np.all(e <= hash_bound(kappa[nz], "oracle")), e.max()


# ... and through the steps of a system whose statistics change under it: the form is chosen anew at every call
def test_unit_form_follows_the_statistics(nb, ctx, monkeypatch):
    monkeypatch.setenv("NBH_HASH_UNITS", "1")
    n = 40000
    ic = _clumpy_box(nb, n, 9)
    a = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NBH_HASH_UNITS", mode)
        ps = nb.ParticleSystem()
        ps.initialize(nb.SimulationConfig(particle_count=n, dt=1e-3, force_method=nb.ForceMethod.SPATIAL_HASH, softening=0.05,
                                          spatial_hash_cell_size=1.0, spatial_hash_cutoff=1.0), initial_conditions=ic)
        for _ in range(12):
            ps.update(1e-3)
        torch.cuda.synchronize()
        a[mode] = {k: getattr(ps.d_particles_, k).cpu().numpy() for k in ("pos_x", "vel_y", "acc_z")}
    for k in a["1"]:
        assert np.array_equal(a["1"][k], a["0"][k]), k


# more cells than bodies, bodies in knots: the start arrays come from the two-level search and the unit form skips the
# empty cells -- every body against the oracle, and equal to the cell-range form on the guess-based start arrays' side
# of the threshold (a grid of fewer cells over the same bodies)
def test_sparse_clumped_grid_against_the_oracle(nb, oracle, ctx, monkeypatch):
    n = 30000
    ic = _clumpy_box(nb, n, 11)
    eps, cutoff = 0.05, 0.4
    for cell in (0.4, 0.5):                        # 61^3 = 227,000 cells (> 2 n: two levels), 49^3 = 118,000 (guess)
        ref, gold, kappa = oracle.spatial_hash_forces_cond(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], 1.0,
                                                           float(np.float32(eps) ** 2), cell, cutoff)
        for mode in ("2", "1"):
            monkeypatch.setenv("NBH_HASH_UNITS", mode)
            d, _ = to_device(nb, ic)
            calc = nb.SpatialHashCalculator(cell, cutoff)
            calc.setSofteningParameter(eps)
            calc.computeForces(d)
            calc.computeForces(d)
            a = acc_of(d)
            nz = np.linalg.norm(ref, axis=1) > 0
            assert np.all(a[~nz] == 0)
            e = rel_err(a[nz], ref[nz])
            assert np.all(e <= hash_bound(kappa[nz], "oracle")), (cell, mode, e.max())


# the filtered form of the wave-per-cell kernel (nbody_hip_grid_tuning 6: window entries out of reach of the box of
# the cell's bodies are left out of LDS) against the oracle on every body: crowded cells (where it is the automatic
# choice), the clumped box, cutoff > cell, eps = 0, and windows too long for one LDS batch
@pytest.mark.parametrize("case", ["dense", "clumped", "strict", "guard"])
def test_filtered_form_of_the_cell_kernel(nb, oracle, ctx, monkeypatch, case):
    monkeypatch.setenv("NBH_HASH_UNITS", "2" if case == "clumped" else "1")
    eps, G = 0.05, 1.0
    if case == "dense":
        n, cell, cutoff = 120000, 2.0, 2.0
        ic = nb.ic.uniform_box(n, seed=21, lo=-10.0, hi=10.0)          # 15 bodies per unit volume = 120 per cell
    elif case == "clumped":
        n, cell, cutoff = 60000, 1.0, 1.0
        ic = _clumpy_box(nb, n, 6)
    elif case == "strict":
        n, cell, cutoff = 80000, 1.0, 1.6
        ic = nb.ic.uniform_box(n, seed=22, lo=-5.0, hi=5.0, min_mass=0.5, max_mass=1.5)   # 80 per cell, long windows
    else:
        n, cell, cutoff, eps = 30000, 1.0, 1.0, 0.0
        ic = nb.ic.uniform_box(n, seed=23, lo=-3.0, hi=3.0)             # eps = 0: the guarded instantiation, 140 per cell
    ref, gold, kappa = oracle.spatial_hash_forces_cond(ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"], G,
                                                       float(np.float32(eps) ** 2), cell, cutoff)
    d, _ = to_device(nb, ic)
    calc = nb.SpatialHashCalculator(cell, cutoff)
    calc.setSofteningParameter(eps)
    calc.computeForces(d)
    auto = acc_of(d)
    calc.getGrid().tuning(6)
    calc.computeForces(d)
    calc.computeForces(d)
    a = acc_of(d)
    assert np.all(np.isfinite(a))
    nz = np.linalg.norm(ref, axis=1) > 0
    assert np.all(a[~nz] == 0)
    e = rel_err(a[nz], ref[nz])
    assert np.all(e <= hash_bound(kappa[nz], "oracle")), (case, e.max())
    if case == "dense":                     # ... where the automatic choice is this form already
        assert np.array_equal(a, auto)


# The key pass is launched before the host has the grid record, on the previous build's key-bit count (csrc/spatial_hash.hip
# grid_build_packed); a box that grows or shrinks across a power of two of cells makes that count wrong and the pass is
# repeated.  Same grids, same forces with the speculation on, off, and forced wrong at every build; the start array by sorted
# position (round 4) against the per-cell search in the same breath.
@pytest.mark.gpu
def test_key_pass_ahead_of_the_grid_record(nb, ctx, monkeypatch):
    n = 300000
    small = nb.ic.uniform_box(n, seed=31, lo=-9.0, hi=9.0)        # 19^3 cells: 13 key bits
    large = {k: (v * 3.5 if k.startswith("pos") else v) for k, v in small.items()}  # 64^3: 18 key bits
    huge = {k: (v * 6.0 if k.startswith("pos") else v) for k, v in small.items()}   # 110^3 > 2 n cells: the two-level search
                                                                                     # for the start array, whose coarse level
                                                                                     # shares its scratch area with the gap list
    got = {}
    for mode, lb in (("0", "cell"), ("1", "position"), ("2", "position")):
        monkeypatch.setenv("NBH_HASH_SPECULATE", mode)
        monkeypatch.setenv("NBH_HASH_LB", lb)
        grid = nb.SpatialHashGrid(n, 1.0)
        res = []
        for ic in (small, small, large, large, huge, large, huge, small):   # right, wrong (grown), right, ..., wrong (shrunk)
            d, _ = to_device(nb, ic)
            grid.build(d)
            grid.computeForces(d, 1.0, 1.0, 0.05)
            cs, ce, pc, si = grid.copyCellDataToHost()
            res.append((cs.copy(), ce.copy(), si.copy(), acc_of(d).copy(), grid.getTotalCells()))
        got[mode] = res
    assert got["0"][0][4] < 2 ** 13 < 2 ** 17 < got["0"][2][4] < 2 * n < got["0"][4][4]
    for mode in ("1", "2"):
        for a, b in zip(got[mode], got["0"]):
            assert a[4] == b[4]
            for x, y in zip(a[:4], b[:4]):
                assert np.array_equal(x, y)
