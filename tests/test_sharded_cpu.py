"""The N>1 path (index-range shards + per-step all-gather + local/remote split) on CPU tensors
with the gloo backend, world_size 2 and 3.  The force math is injected from the oracle here (the
product backend is HIP-only); what is under test is the partition / collective / buffer-swap
logic of n-body_amd/distributed.py, which is identical on RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    """CPU stand-in for HipBackend, built on the oracle's restated arithmetic (tests only)."""

    def __init__(self):
        import oracle_bind
        self.o = oracle_bind.load()
        self.device = torch.device("cpu")

    def drift(self, posm, vel, acc, dt):
        p, v, a = posm.numpy(), vel.numpy(), acc.numpy()
        dt = np.float32(dt)
        h = np.float32(0.5) * dt * dt
        p[:, :3] += v[:, :3] * dt + a[:, :3] * h

    def kick(self, vel, acc_old, acc_new, dt):
        v = vel.numpy()
        v[:, :3] += (acc_old.numpy()[:, :3] + acc_new.numpy()[:, :3]) * (np.float32(0.5) * np.float32(dt))

    def forces(self, targets, sources, G, eps2, out, accumulate):
        t, s = targets.numpy(), sources.numpy()
        c = np.ascontiguousarray
        if s.shape[0] == 0:
            a = np.zeros((t.shape[0], 3), np.float32)
        else:
            a = np.stack(self.o.direct_forces_points(c(s[:, 0]), c(s[:, 1]), c(s[:, 2]), c(s[:, 3]),
                                                     c(t[:, 0]), c(t[:, 1]), c(t[:, 2]), G, eps2), 1)
        o = out.numpy()
        if accumulate:
            o[:, :3] += a
        else:
            o[:, :3] = a
            o[:, 3] = 0


    def forces_pair(self, a, b, G, eps2, acc_a, acc_b, accumulate_a, accumulate_b):
        self.forces(a, b, G, eps2, acc_a, accumulate_a)
        self.forces(b, a, G, eps2, acc_b, accumulate_b)

    def energies(self, posm, vel, self_offset, sources, G, eps):
        # the shard's KE and its share of the PE from the oracle's one-sided sums (fp64 accumulation)
        p, v, s = posm.numpy().astype(np.float64), vel.numpy().astype(np.float64), sources.numpy().astype(np.float64)
        ke = float((0.5 * p[:, 3] * (v[:, :3] ** 2).sum(1)).sum())
        d = s[None, :, :3] - p[:, None, :3]
        r2 = (d ** 2).sum(2) + float(np.float32(eps) * np.float32(eps))
        w = s[None, :, 3] / np.sqrt(r2)
        idx = np.arange(p.shape[0]) + self_offset
        ok = (idx >= 0) & (idx < s.shape[0])
        w[np.arange(p.shape[0])[ok], idx[ok]] = 0.0
        return ke, float(-0.5 * G * (p[:, 3] * w.sum(1)).sum())

    def tree_forces(self, posm_all, first, count, theta, G, eps, out_all):
        # the oracle's tree for the bodies [first, first + count) taken in index order (which bodies
        # a rank walks does not change any body's result)
        p = posm_all.numpy()
        c = np.ascontiguousarray
        idx = np.arange(first, first + count)
        eps2 = float(np.float32(eps) * np.float32(eps))
        bx, by, bz, _, _ = self.o.barnes_hut_forces(c(p[:, 0]), c(p[:, 1]), c(p[:, 2]), c(p[:, 3]), idx, G, eps2, theta)
        o = out_all.numpy()
        o[idx, 0], o[idx, 1], o[idx, 2], o[idx, 3] = bx, by, bz, 0

    # -- spatial hash (z-slab path) -------------------------------------------------------------
    def bbox(self, posm):
        p = posm.numpy()
        return torch.from_numpy(np.concatenate([p[:, :3].min(0), p[:, :3].max(0)]).astype(np.float32))

    def cell_z(self, posm, lo_z, cell, gz):
        z = posm.numpy()[:, 2]
        c = np.floor((z - np.float32(lo_z)) / np.float32(cell)).astype(np.int64)
        return torch.from_numpy(np.clip(c, 0, gz - 1).astype(np.int32))

    def hash_forces(self, posm_all, bounds, cell, cutoff, G, eps):
        p = posm_all.numpy()
        c = np.ascontiguousarray
        f = np.float32
        dims = [int(np.ceil((f(bounds[3 + a]) - f(bounds[a])) / f(cell))) + 1 for a in range(3)]
        eps2 = float(f(eps) * f(eps))
        a = np.stack(self.o.spatial_hash_forces_grid(c(p[:, 0]), c(p[:, 1]), c(p[:, 2]), c(p[:, 3]),
                                                     p.shape[0], G, eps2, cell, cutoff, bounds[:3], dims), 1)
        out = np.zeros((p.shape[0], 4), np.float32)
        out[:, :3] = a
        return torch.from_numpy(out)


    # -- the slab operations of the overlapped two-grid step (numpy restatements of csrc/slab.hip and of
    #    the two-grid force kernel; force sums by the oracle) ------------------------------------------
    def _geometry(self, gbox, cell):
        f = np.float32
        lo = [f(gbox[a]) - f(0.001) for a in range(3)]
        hi = [f(gbox[3 + a]) + f(0.001) for a in range(3)]
        dims = [int(np.ceil((hi[a] - lo[a]) / f(cell))) + 1 for a in range(3)]
        return lo, hi, dims

    def slab_partition(self, posm, vel, acc, gid, gbox, cell, world, rank, hist_cap):
        p, v, a, g = posm.numpy(), vel.numpy(), acc.numpy(), gid.numpy()
        lo, _, dims = self._geometry(gbox.numpy(), cell)
        gz = dims[2]
        z = np.clip(np.floor((p[:, 2] - lo[2]) / np.float32(cell)).astype(np.int64), 0, gz - 1)
        dest = ((z + 1) * world - 1) // gz
        leave = np.nonzero(dest != rank)[0]                      # ascending = the vacated slots
        order = leave[np.argsort(dest[leave], kind="stable")]   # rows: grouped by new owner, input order inside
        rows = np.zeros((p.shape[0], 16), np.float32)
        k = order.size
        rows[:k, 0:4], rows[:k, 4:7], rows[:k, 8:11] = p[order], v[order, :3], a[order, :3]
        rows[:k, 12] = g[order].astype(np.int32).view(np.float32)
        rows[:k, 13] = z[order].astype(np.int32).view(np.float32)
        holes = np.zeros(p.shape[0], np.int32)
        holes[:k] = leave
        stats = np.zeros(world * world + hist_cap, np.int32)
        stats[rank * world:(rank + 1) * world] = np.bincount(dest, minlength=world)
        if gz <= hist_cap:
            stats[world * world:world * world + gz] = np.bincount(z, minlength=gz)
        info = np.array([dims[0], dims[1], gz, int(gz > hist_cap)], np.int32)
        return torch.from_numpy(rows), torch.from_numpy(holes), torch.from_numpy(stats), torch.from_numpy(info)

    def slab_fill(self, posm, vel, acc, gid, n_old, holes, n_holes, arrivals):
        # numpy restatement of slab_fill_kernel: arrivals into the holes in order, surplus appended, surplus
        # holes closed with the bodies of the old tail (in order)
        r = arrivals.numpy()
        A, L = r.shape[0], n_holes
        h = holes.numpy()[:L]
        n_new = n_old - L + A
        arrs = (posm.numpy(), vel.numpy(), acc.numpy())
        g = gid.numpy()
        slots = np.concatenate([h[:min(A, L)], np.arange(n_old, n_old + max(A - L, 0))]).astype(np.int64)
        for arr, c0 in zip(arrs, (0, 4, 8)):
            arr[slots] = 0
            arr[slots, :4 if c0 == 0 else 3] = r[:, c0:c0 + (4 if c0 == 0 else 3)]
        g[slots] = np.ascontiguousarray(r[:, 12]).view(np.int32)
        if A < L:
            tail = np.setdiff1d(np.arange(n_new, n_old), h)       # bodies of the old tail, ascending
            to = h[A:A + tail.size]
            assert tail.size == L - A - np.count_nonzero(h >= n_new) and np.all(to < n_new)
            for arr in arrs:
                arr[to] = arr[tail]
            g[to] = g[tail]

    def grid_build(self, slot, posm, bounds, cell, z_first, z_count):
        p = posm.numpy().copy()
        f = np.float32
        dims = [int(np.ceil((f(bounds[3 + a]) - f(bounds[a])) / f(cell))) + 1 for a in range(3)]
        c = [np.clip(np.floor((p[:, a] - f(bounds[a])) / f(cell)).astype(np.int64), 0, dims[a] - 1) for a in range(3)]
        key = c[0] + c[1] * dims[0] + c[2] * dims[0] * dims[1]
        assert np.all((c[2] >= z_first) & (c[2] < z_first + z_count))  # the slab promise of the caller
        self.__dict__.setdefault("_grids", {})[slot] = dict(p=p, z=c[2], order=np.argsort(key, kind="stable"),
                                                            bounds=list(bounds), dims=dims, cell=cell)

    def grid_sorted(self, slot, first, count, out):
        g = self._grids[slot]
        out.numpy()[:count] = g["p"][g["order"][first:first + count]]

    def grid_forces(self, slot_t, slot_s, z_first, z_count, cutoff, G, eps, acc_out, accumulate):
        gt, gs = self._grids[slot_t], self._grids[slot_s]
        assert gt["dims"] == gs["dims"] and gt["bounds"][:3] == gs["bounds"][:3]
        sel = np.nonzero((gt["z"] >= z_first) & (gt["z"] < z_first + z_count))[0]
        if sel.size == 0:
            return True
        t = gt["p"][sel]
        if slot_t == slot_s:
            others = np.ones(gt["p"].shape[0], bool)
            others[sel] = False
            src = np.concatenate([t, gt["p"][others]])          # targets first: they are sources too
        else:
            tz = t.copy()
            tz[:, 3] = 0.0                                       # other grid: the targets exert nothing
            src = np.concatenate([tz, gs["p"]])
        c = np.ascontiguousarray
        eps2 = float(np.float32(eps) * np.float32(eps))
        a = np.stack(self.o.spatial_hash_forces_grid(c(src[:, 0]), c(src[:, 1]), c(src[:, 2]), c(src[:, 3]), sel.size,
                                                     G, eps2, gt["cell"], cutoff, gt["bounds"][:3], gt["dims"]), 1)
        o = acc_out.numpy()
        if accumulate:
            o[sel, :3] += a
        else:
            o[sel, :3] = a
            o[sel, 3] = 0
        return True

    def to_host(self, *tensors):
        return [t.numpy() for t in tensors]


class GenericOracleBackend(OracleBackend):
    """Without the slab operations: ShardedHashSystem takes its one-grid path."""
    slab_partition = property()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, steps, out_dir, mode=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nbody_amd
        from nbody_amd.distributed import ShardedDirectSystem, shard_bounds
        ic = nbody_amd.ic.plummer(n, seed=5)
        sysm = ShardedDirectSystem(ic, 1.0, 0.01, backend=OracleBackend(), device="cpu", mode=mode)
        S, lo, hi = shard_bounds(n, world, rank)
        assert (sysm.S, sysm.lo, sysm.hi) == (S, lo, hi)
        sysm.initial_forces()
        for _ in range(steps):
            sysm.step(1e-3)
        pos = sysm.gather_global("posm")
        vel = sysm.gather_global("vel")
        acc = sysm.gather_global("acc")
        ke, pe = sysm.energies()
        if rank == 0:
            np.savez(os.path.join(out_dir, f"w{world}.npz"), pos=pos, vel=vel, acc=acc, ke=ke, pe=pe)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,mode", [(2, 600, "pair"), (3, 601, "pair"), (4, 602, "pair"),
                                          (8, 808, "pair"),  # the driver's largest run: 8 ranks
                                          (2, 600, "gather"), (3, 601, "gather")])
def test_sharded_steps_match_single_process(tmp_path, world, n, mode, oracle, nb):
    from oracle_bind import host_state
    steps = 3
    mp.spawn(_worker, args=(world, _free_port(), n, steps, str(tmp_path), mode), nprocs=world, join=True)
    got = np.load(tmp_path / f"w{world}.npz")
    ic = nb.ic.plummer(n, seed=5)
    s = host_state(ic)
    eps2 = float(np.float32(0.01) * np.float32(0.01))
    s["acc_x"], s["acc_y"], s["acc_z"] = oracle.direct_forces(s["pos_x"], s["pos_y"], s["pos_z"],
                                                              s["mass"], 1.0, eps2, 1)
    oracle.integrate_direct(s, 1.0, 0.01, 1e-3, steps, 1)
    for k, col in (("pos_x", 0), ("pos_y", 1), ("pos_z", 2)):
        assert np.allclose(got["pos"][:, col], s[k], rtol=1e-6, atol=1e-6), k
    assert np.array_equal(got["pos"][:, 3], s["mass"])
    for k, col in (("vel_x", 0), ("vel_y", 1), ("vel_z", 2)):
        assert np.allclose(got["vel"][:, col], s[k], rtol=1e-5, atol=1e-6), k
    for k, col in (("acc_x", 0), ("acc_y", 1), ("acc_z", 2)):
        assert np.allclose(got["acc"][:, col], s[k], rtol=2e-5, atol=1e-6), k
    assert abs(float(got["ke"]) - oracle.kinetic_energy(s, 256, 2)) < 1e-6
    pe = oracle.potential_energy(s, 1.0, 0.01, 256, 2)
    assert abs(float(got["pe"]) - pe) < 1e-6 * abs(pe)


def test_shard_bounds(nb):
    from nbody_amd.distributed import shard_bounds
    for n, w in ((1 << 20, 8), (10, 3), (7, 8), (1, 1)):
        sizes, prev_hi = [], 0
        for r in range(w):
            S, lo, hi = shard_bounds(n, w, r)
            assert lo == min(n, prev_hi) or lo == prev_hi
            assert hi - lo <= S
            sizes.append(hi - lo)
            prev_hi = hi
        assert sum(sizes) == n
    assert shard_bounds(1 << 20, 8, 3) == (131072, 393216, 524288)


def _hash_worker(rank, world, port, n, steps, cutoff, out_dir, generic=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nbody_amd
        from nbody_amd.distributed import ShardedHashSystem
        ic = nbody_amd.ic.uniform_box(n, seed=11, lo=-4.0, hi=4.0, min_mass=0.5, max_mass=1.5)
        rng = np.random.default_rng(3)
        for k in ("vel_x", "vel_y", "vel_z"):
            ic[k] = rng.normal(0, 3.0, n).astype(np.float32)  # fast bodies: layers change owner
        sysm = ShardedHashSystem(ic, 1.0, 0.05, 1.0, cutoff, device="cpu",
                                 backend=GenericOracleBackend() if generic else OracleBackend())
        sysm.initial_forces()
        moved = 0
        for _ in range(steps):
            sysm.step(0.02)
            moved += sysm.migrated
        assert sysm.path == ("one-grid" if generic else "two-grid")
        gid, pos, vel, acc = sysm.gather_global()
        tot = torch.tensor([moved, sysm.halo_bodies], dtype=torch.int64)
        dist.all_reduce(tot)
        if rank == 0:
            np.savez(os.path.join(out_dir, f"h{world}.npz"), gid=gid, pos=pos, vel=vel, acc=acc,
                     moved=int(tot[0]), halo=int(tot[1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,cutoff,generic", [(2, 1.0, False), (3, 1.0, False), (3, 2.0, False), (4, 1.0, False),
                                                  (3, 2.0, True)])
def test_sharded_hash_matches_single_process(tmp_path, world, cutoff, generic, oracle, nb):
    """z-slab shards + migration + halo exchange == the single-grid run (the reference semantics,
    including the pairs its 27-cell search misses when cutoff > cell); overlapped two-grid step and the
    one-grid fallback."""
    from oracle_bind import host_state
    n, steps, dt, eps = 1500, 3, 0.02, 0.05
    mp.spawn(_hash_worker, args=(world, _free_port(), n, steps, cutoff, str(tmp_path), generic), nprocs=world,
             join=True)
    got = np.load(tmp_path / f"h{world}.npz")
    assert np.array_equal(got["gid"], np.arange(n))      # every body exactly once
    assert got["moved"] > 0 and got["halo"] > 0           # migration and halos were exercised
    ic = nb.ic.uniform_box(n, seed=11, lo=-4.0, hi=4.0, min_mass=0.5, max_mass=1.5)
    rng = np.random.default_rng(3)
    for k in ("vel_x", "vel_y", "vel_z"):
        ic[k] = rng.normal(0, 3.0, n).astype(np.float32)
    s = host_state(ic)
    eps2 = float(np.float32(eps) * np.float32(eps))

    def forces():
        return oracle.spatial_hash_forces(s["pos_x"], s["pos_y"], s["pos_z"], s["mass"], 1.0, eps2, 1.0, cutoff)

    s["acc_x"], s["acc_y"], s["acc_z"] = forces()
    for _ in range(steps):
        for k in ("x", "y", "z"):
            s["acc_old_" + k] = s["acc_" + k].copy()
        oracle.update_positions(s, dt)
        s["acc_x"], s["acc_y"], s["acc_z"] = forces()
        oracle.update_velocities(s, dt)
    for col, k in enumerate(("pos_x", "pos_y", "pos_z")):
        assert np.allclose(got["pos"][:, col], s[k], rtol=1e-6, atol=1e-6), k
    assert np.array_equal(got["pos"][:, 3], s["mass"])
    for col, k in enumerate(("vel_x", "vel_y", "vel_z")):
        assert np.allclose(got["vel"][:, col], s[k], rtol=1e-5, atol=1e-5), k
    for col, k in enumerate(("acc_x", "acc_y", "acc_z")):
        assert np.allclose(got["acc"][:, col], s[k], rtol=1e-5, atol=1e-5), k


def test_layer_owner(nb):
    from nbody_amd.distributed import layer_owner
    o = layer_owner(65, 8)
    assert o[0] == 0 and o[-1] == 7 and np.all(np.diff(o) >= 0)
    assert np.bincount(o, minlength=8).min() >= 8
    o = layer_owner(3, 8)   # fewer layers than ranks: some ranks own nothing
    assert len(set(o.tolist())) == 3


def test_pair_schedule_covers_every_pair_once(nb):
    from nbody_amd.distributed import pair_schedule
    for W in (1, 2, 3, 4, 5, 8):
        for S in (1, 4, 7):
            seen = np.zeros((W * S, W * S), dtype=np.int32)
            for r in range(W):
                for i0, i1, sh, j0, j1 in pair_schedule(W, r, S):
                    assert sh != r and 0 <= i0 < i1 <= S and 0 <= j0 < j1 <= S
                    seen[r * S + i0:r * S + i1, sh * S + j0:sh * S + j1] += 1
            both = seen + seen.T
            for a in range(W):
                for b in range(W):
                    blk = both[a * S:(a + 1) * S, b * S:(b + 1) * S]
                    assert np.all(blk == (0 if a == b else 1)), (W, S, a, b)
        # balance at even W: every rank evaluates the same number of body pairs
        work = [sum((i1 - i0) * (j1 - j0) for i0, i1, _, j0, j1 in pair_schedule(W, r, 8)) for r in range(W)]
        assert max(work) - min(work) <= 0, (W, work)


def _tree_worker(rank, world, port, n, steps, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nbody_amd
        from nbody_amd.distributed import ShardedTreeSystem
        ic = nbody_amd.ic.plummer(n, seed=6)
        sysm = ShardedTreeSystem(ic, 1.0, 0.02, 0.5, backend=OracleBackend(), device="cpu")
        sysm.initial_forces()
        for _ in range(steps):
            sysm.step(1e-3)
        pos, vel, acc = (sysm.gather_global(k) for k in ("posm", "vel", "acc"))
        if rank == 0:
            np.savez(os.path.join(out_dir, f"t{world}.npz"), pos=pos, vel=vel, acc=acc)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 500), (3, 501)])
def test_sharded_tree_matches_single_process(tmp_path, world, n, oracle, nb):
    """replicated tree + partitioned walk + reduce == the single-process Barnes-Hut run"""
    from oracle_bind import host_state
    steps, dt, eps, theta = 2, 1e-3, 0.02, 0.5
    mp.spawn(_tree_worker, args=(world, _free_port(), n, steps, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / f"t{world}.npz")
    s = host_state(nb.ic.plummer(n, seed=6))
    eps2 = float(np.float32(eps) * np.float32(eps))

    def forces():
        return oracle.barnes_hut_forces(s["pos_x"], s["pos_y"], s["pos_z"], s["mass"], np.arange(n), 1.0, eps2, theta)[:3]

    s["acc_x"], s["acc_y"], s["acc_z"] = forces()
    for _ in range(steps):
        for k in ("x", "y", "z"):
            s["acc_old_" + k] = s["acc_" + k].copy()
        oracle.update_positions(s, dt)
        s["acc_x"], s["acc_y"], s["acc_z"] = forces()
        oracle.update_velocities(s, dt)
    for col, k in enumerate(("pos_x", "pos_y", "pos_z")):
        assert np.allclose(got["pos"][:, col], s[k], rtol=1e-6, atol=1e-6), k
    for col, k in enumerate(("vel_x", "vel_y", "vel_z")):
        assert np.allclose(got["vel"][:, col], s[k], rtol=1e-5, atol=1e-6), k
    for col, k in enumerate(("acc_x", "acc_y", "acc_z")):
        assert np.allclose(got["acc"][:, col], s[k], rtol=1e-5, atol=1e-6), k
