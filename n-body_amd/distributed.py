"""Direct N^2 sharded over the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI).

The reference is single-GPU; this is new.  Bodies shard by contiguous index
range (rank r owns [r*S, (r+1)*S), S = ceil(N / world), the tail padded with zero-mass bodies that
exert no force).  One Velocity-Verlet step (ref: Integrator::integrate, integrator.cu:224-238):

    drift own bodies                                   (compute stream)
    all-gather {x,y,z,m} float4 of every shard         (RCCL stream; 16 B/body, 2 MiB per rank
                                                        at N = 2^20, once per step)
    own shard against itself                           (compute stream, overlaps the gather)
    wait for the gather
    mode "pair" (default): every PAIR of shards is evaluated by ONE rank, action and reaction
        together (Newton's third law, nbody_hip_direct_forces_pair_packed): rank r takes the
        shards r+1 .. r+(W-1)/2 of the ring, plus half of the antipodal rectangle when W is even;
        its contributions to everybody's accelerations go into one [N,4] buffer and a single
        reduce-scatter (sum) hands every rank the total for its own bodies.  Per rank that is
        N^2/(2W) pair evaluations instead of N^2/W.
    mode "gather": forces from the shards left and right of the own range with the one-sided
        kernel, accumulated (needed when eps^2 < 1e-12; no reduce-scatter)
    kick own bodies; swap acceleration buffers

The force math is the C-ABI call nbody_hip_direct_forces_packed; `backend` only exists so that
the partition / collective logic can be exercised on CPU tensors with the gloo backend in
tests/ (tests inject a CPU backend; nothing in this module imports the oracle).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from ._lib import check
from ._lib import finalizing as _lib_finalizing, track as _lib_track
from .api import Context, direct_forces_pair_packed, direct_forces_packed


class HipBackend:
    """The product path: every operation is a HIP kernel behind the C ABI."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.device = ctx.torch_device
        self._grids = {}
        self._tree = None
        _lib_track(self, "backend")

    def close(self):
        """Frees the lazily created grid / tree handles (also run at garbage collection)."""
        lib = self.ctx._lib
        if self.ctx.handle.value:
            for h, _, _ in self._grids.values():
                lib.nbody_hip_grid_destroy(h)
            if self._tree is not None:
                lib.nbody_hip_tree_destroy(self._tree)
        self._grids = {}
        self._tree = None

    def __del__(self):
        try:
            if not _lib_finalizing():  # (else: closed by the exit hook _lib.close_all, or the runtime is going down)
                self.close()
        except Exception:
            pass

    def drift(self, posm, vel, acc, dt):
        check(self.ctx._lib.nbody_hip_drift_packed(self.ctx.handle, posm.data_ptr(), vel.data_ptr(),
                                                   acc.data_ptr(), posm.shape[0], dt))

    def kick(self, vel, acc_old, acc_new, dt):
        check(self.ctx._lib.nbody_hip_kick_packed(self.ctx.handle, vel.data_ptr(), acc_old.data_ptr(),
                                                  acc_new.data_ptr(), vel.shape[0], dt))

    def forces(self, targets, sources, G, eps2, out, accumulate):
        direct_forces_packed(self.ctx, targets, sources, G, eps2, out=out, accumulate=accumulate)

    def forces_pair(self, a, b, G, eps2, acc_a, acc_b, accumulate_a, accumulate_b):
        direct_forces_pair_packed(self.ctx, a, b, G, eps2, acc_a, acc_b, accumulate_a, accumulate_b)

    def energies(self, posm, vel, self_offset, sources, G, eps):
        """(KE of the shard, its share of the PE): nbody_hip_energies_packed."""
        out = (C.c_double * 2)()
        check(self.ctx._lib.nbody_hip_energies_packed(self.ctx.handle, posm.data_ptr(), vel.data_ptr(),
                                                      posm.shape[0], self_offset, sources.data_ptr(),
                                                      sources.shape[0], G, eps, out))
        return float(out[0]), float(out[1])

    # -- Barnes-Hut (replicated tree, partitioned walk) ----------------------------------------
    def tree_forces(self, posm_all, first, count, theta, G, eps, out_all):
        """Builds the tree of posm_all and walks its Morton-sorted bodies [first, first + count):
        their accelerations go to the rows of out_all that are their original indices."""
        n = posm_all.shape[0]
        if getattr(self, "_tree_cap", 0) < n:
            if getattr(self, "_tree", None) is not None:
                self.ctx._lib.nbody_hip_tree_destroy(self._tree)
            h = C.c_void_p()
            check(self.ctx._lib.nbody_hip_tree_create(self.ctx.handle, n, C.byref(h)))
            self._tree, self._tree_cap = h, n
        check(self.ctx._lib.nbody_hip_tree_build_packed(self._tree, posm_all.data_ptr(), n))
        check(self.ctx._lib.nbody_hip_tree_compute_forces_packed(self._tree, first, count, theta, G, eps,
                                                                 out_all.data_ptr()))

    # -- spatial hash (z-slab path) ------------------------------------------------------------
    def bbox(self, posm):
        """{lo x,y,z, hi x,y,z} of packed bodies as a device tensor of 6 floats."""
        out = torch.empty(6, dtype=torch.float32, device=posm.device)
        check(self.ctx._lib.nbody_hip_bbox_packed(self.ctx.handle, posm.data_ptr(), posm.shape[0],
                                                  out.data_ptr()))
        return out

    def cell_z(self, posm, lo_z, cell, gz):
        cz = torch.empty(posm.shape[0], dtype=torch.int32, device=posm.device)
        check(self.ctx._lib.nbody_hip_cell_z_packed(self.ctx.handle, posm.data_ptr(), posm.shape[0],
                                                    lo_z, cell, gz, cz.data_ptr()))
        return cz

    def hash_forces(self, posm_all, bounds, cell, cutoff, G, eps):
        """Accelerations [n,4] of every body of posm_all on the grid with the given padded bounds
        (one grid, one launch: the fallback of the sharded step for grids too sparse or too tall)."""
        n = posm_all.shape[0]
        h = self._grid_handle("one", n, cell)
        check(self.ctx._lib.nbody_hip_grid_set_slab(h, 0, 0))
        b = (C.c_float * 6)(*bounds)
        check(self.ctx._lib.nbody_hip_grid_build_packed(h, posm_all.data_ptr(), n, b))
        acc = torch.empty((n, 4), dtype=torch.float32, device=posm_all.device)
        check(self.ctx._lib.nbody_hip_grid_compute_forces_packed(h, cutoff, G, eps, acc.data_ptr()))
        return acc

    def _grid_handle(self, slot, n, cell):
        """Grid handles by role ("own" bodies of the slab, "halo" layers, "one" = fallback), grown as needed."""
        grids = self.__dict__.setdefault("_grids", {})
        cur = grids.get(slot)
        if cur is None or cur[1] < n or cur[2] != cell:
            if cur is not None:
                self.ctx._lib.nbody_hip_grid_destroy(cur[0])
            h = C.c_void_p()
            cap = int(n * 1.5) + 1024
            check(self.ctx._lib.nbody_hip_grid_create(self.ctx.handle, cap, cell, C.byref(h)))
            grids[slot] = cur = (h, cap, cell)
        return cur[0]

    def slab_partition(self, posm, vel, acc, gid, gbox, cell, world, rank, hist_cap):
        """One pass over the rank's bodies (csrc/slab.hip): the bodies that change owner as rows [.,16]
        grouped by new owner + the slots they vacate, this rank's row of the send matrix [W,W] followed by
        its layer histogram [hist_cap], and {gx, gy, gz, overflow}; all on the device, nothing comes back
        to the host here.  The bodies that stay are not touched."""
        n = posm.shape[0]
        dev = gbox.device
        rows = torch.empty((n if world > 1 else 0, 16), dtype=torch.float32, device=dev)
        holes = torch.empty(n if world > 1 else 0, dtype=torch.int32, device=dev)
        stats = torch.empty(world * world + hist_cap, dtype=torch.int32, device=dev)
        info = torch.empty(4, dtype=torch.int32, device=dev)
        check(self.ctx._lib.nbody_hip_slab_partition(
            self.ctx.handle, posm.data_ptr(), vel.data_ptr(), acc.data_ptr(), gid.data_ptr(), n, gbox.data_ptr(),
            cell, world, rank, hist_cap, rows.data_ptr(), holes.data_ptr(), stats.data_ptr(),
            stats.data_ptr() + 4 * world * world, info.data_ptr()))
        return rows, holes, stats, info

    def slab_fill(self, posm, vel, acc, gid, n_old, holes, n_holes, arrivals):
        """The capacity arrays after the exchange: arrivals into the vacated slots, surplus appended, surplus
        slots closed from the end (nbody_hip_slab_fill).  The arrays must hold n_old - n_holes + arrivals rows."""
        check(self.ctx._lib.nbody_hip_slab_fill(self.ctx.handle, arrivals.data_ptr(), arrivals.shape[0],
                                                holes.data_ptr(), n_holes, n_old, posm.data_ptr(), vel.data_ptr(),
                                                acc.data_ptr(), gid.data_ptr()))

    def grid_build(self, slot, posm, bounds, cell, z_first, z_count):
        """Grid `slot` over `posm` on the global box `bounds`; the bodies lie in layers [z_first, +z_count)."""
        h = self._grid_handle(slot, posm.shape[0], cell)
        check(self.ctx._lib.nbody_hip_grid_set_slab(h, z_first, z_count))
        check(self.ctx._lib.nbody_hip_grid_build_packed(h, posm.data_ptr(), posm.shape[0], (C.c_float * 6)(*bounds)))

    def grid_sorted(self, slot, first, count, out):
        """out[:count] = bodies [first, first+count) of grid `slot` in cell order (a z layer is one run)."""
        check(self.ctx._lib.nbody_hip_grid_sorted_bodies(self._grids[slot][0], first, count, out.data_ptr()))

    def grid_forces(self, slot_t, slot_s, z_first, z_count, cutoff, G, eps, acc_out, accumulate):
        """acc_out rows of grid slot_t's bodies in layers [z_first, +z_count) (+)= forces from grid slot_s.
        False when a grid is too sparse for the two-grid kernel (the caller falls back)."""
        rc = self.ctx._lib.nbody_hip_grid_forces_pair_packed(self._grids[slot_t][0], self._grids[slot_s][0], z_first,
                                                             z_count, cutoff, G, eps, acc_out.data_ptr(),
                                                             1 if accumulate else 0)
        if rc == -4:  # NBODY_HIP_ERR_STATE: no per-cell start array
            return False
        check(rc)
        return True

    def to_host(self, *tensors):
        """The step's one host synchronisation: small device tensors -> numpy."""
        return [t.cpu().numpy() for t in tensors]


def shard_bounds(n: int, world: int, rank: int):
    """(shard size S, lo, hi) of `rank`: contiguous, every shard padded to S = ceil(n/world)."""
    s = (n + world - 1) // world
    lo = min(n, rank * s)
    hi = min(n, lo + s)
    return s, lo, hi


def pair_schedule(world: int, rank: int, S: int):
    """Shard pairs evaluated by `rank` in pair mode: list of (i0, i1, shard, j0, j1) meaning
    own bodies [i0, i1) x bodies [j0, j1) of `shard`, each body pair of two DIFFERENT shards
    appearing in exactly one rank's list.  Rank r takes the ring neighbours r+1 .. r+(W-1)/2; for
    even W the antipodal rectangle is cut in half: the lower rank takes its first half x the
    whole partner, the upper rank its whole shard x the partner's second half."""
    out = []
    for d in range(1, (world - 1) // 2 + 1):
        out.append((0, S, (rank + d) % world, 0, S))
    if world % 2 == 0 and world > 1:
        sh = (rank + world // 2) % world
        h = S // 2
        if rank < world // 2:
            if h > 0:
                out.append((0, h, sh, 0, S))
        else:
            out.append((0, S, sh, h, S))
    return out


class ShardedDirectSystem:
    """State of one rank: its shard as packed float4 arrays + the gathered source buffer."""

    def __init__(self, ic: dict, G: float, eps: float, backend=None, group=None, device=None,
                 mode: str | None = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = backend
        self.device = torch.device(device) if device is not None else backend.device
        self.G = float(G)
        self.eps2 = float(np.float32(eps) * np.float32(eps))
        if mode is None:
            mode = "pair" if (self.eps2 >= 1e-12 and hasattr(backend, "forces_pair")) else "gather"
        assert mode in ("pair", "gather")
        self.mode = mode
        self.contrib = None
        self._use_reduce_scatter = None
        n = int(ic["pos_x"].size)
        self.n = n
        self.S, self.lo, self.hi = shard_bounds(n, self.world, self.rank)
        S, lo, hi = self.S, self.lo, self.hi
        posm = np.zeros((S, 4), np.float32)  # padded bodies: origin, zero mass
        vel = np.zeros((S, 4), np.float32)
        for k, f in enumerate(("pos_x", "pos_y", "pos_z", "mass")):
            posm[:hi - lo, k] = ic[f][lo:hi]
        for k, f in enumerate(("vel_x", "vel_y", "vel_z")):
            if f in ic:
                vel[:hi - lo, k] = ic[f][lo:hi]
        self.posm = torch.from_numpy(posm).to(self.device)
        self.vel = torch.from_numpy(vel).to(self.device)
        self.acc = torch.zeros((S, 4), dtype=torch.float32, device=self.device)
        self.acc_prev = torch.zeros_like(self.acc)
        self.posm_all = torch.zeros((S * self.world, 4), dtype=torch.float32, device=self.device)
        self.steps_done = 0

    # -- one exchange + force evaluation ------------------------------------------------------
    def compute_forces(self, out: torch.Tensor):
        b = self.backend
        S, r = self.S, self.rank
        work = None
        if dist.is_initialized():  # also with one rank: same code path as the N-GPU run
            work = dist.all_gather_into_tensor(self.posm_all, self.posm, group=self.group,
                                               async_op=True)
        W = self.world
        if self.mode == "gather" or W == 1:
            # own shard against itself while the gather is in flight
            b.forces(self.posm, self.posm, self.G, self.eps2, out, False)
            if work is not None:
                work.wait()  # orders the compute stream after the collective
                if r > 0:
                    b.forces(self.posm, self.posm_all[:r * S], self.G, self.eps2, out, True)
                if r < W - 1:
                    b.forces(self.posm, self.posm_all[(r + 1) * S:], self.G, self.eps2, out, True)
            return
        # -- pair mode: each pair of shards once, reactions returned by a reduce-scatter ---------
        if self.contrib is None:
            self.contrib = torch.empty((S * W, 4), dtype=torch.float32, device=self.device)
        c = self.contrib
        c.zero_()
        mine = c[r * S:(r + 1) * S]
        b.forces(self.posm, self.posm, self.G, self.eps2, mine, False)  # overlaps the gather
        work.wait()
        for i0, i1, sh, j0, j1 in pair_schedule(W, r, S):
            b.forces_pair(self.posm[i0:i1], self.posm_all[sh * S + j0:sh * S + j1], self.G, self.eps2,
                          mine[i0:i1], c[sh * S + j0:sh * S + j1], True, False)
        if self._use_reduce_scatter is None:
            self._use_reduce_scatter = dist.get_backend(self.group) != "gloo"  # gloo has none
        if self._use_reduce_scatter:
            try:
                dist.reduce_scatter_tensor(out, c, op=dist.ReduceOp.SUM, group=self.group)
                return
            except (RuntimeError, NotImplementedError, AttributeError):
                self._use_reduce_scatter = False  # every rank runs the same build: all fall back alike
        dist.all_reduce(c, group=self.group)       # all-reduce + slice: 2x the bytes, same result
        out.copy_(mine)

    def initial_forces(self):
        """ref: ParticleSystem::initialize evaluates a(0) once (particle_system.cpp:88-91)."""
        self.compute_forces(self.acc)

    def step(self, dt: float):
        b = self.backend
        b.drift(self.posm, self.vel, self.acc, dt)
        self.acc, self.acc_prev = self.acc_prev, self.acc  # a_old <- a by buffer swap
        self.compute_forces(self.acc)
        b.kick(self.vel, self.acc_prev, self.acc, dt)
        self.steps_done += 1

    # -- inspection ---------------------------------------------------------------------------
    def local_state(self):
        k = self.hi - self.lo
        return {"posm": self.posm[:k], "vel": self.vel[:k], "acc": self.acc[:k]}

    def gather_global(self, name: str) -> np.ndarray:
        """Rows [0, n) of a per-shard array on every rank (test / checkpoint helper)."""
        t = getattr(self, name)
        if self.world == 1:
            return t[: self.n].cpu().numpy()
        full = torch.empty((self.S * self.world, 4), dtype=torch.float32, device=self.device)
        dist.all_gather_into_tensor(full, t.contiguous(), group=self.group)
        return full[: self.n].cpu().numpy()

    def energies(self, eps: float | None = None):
        """(KE, PE) of the whole system on every rank: each rank reduces its shard
        against the gathered bodies on the device, then ONE all-reduce of two doubles."""
        eps = float(np.sqrt(self.eps2)) if eps is None else float(eps)
        src = self.posm
        if self.world > 1:
            dist.all_gather_into_tensor(self.posm_all, self.posm, group=self.group)
            src = self.posm_all
        ke, pe = self.backend.energies(self.posm, self.vel, self.rank * self.S, src, self.G, eps)
        e = torch.tensor([ke, pe], dtype=torch.float64)
        if self.world > 1:
            if dist.get_backend(self.group) != "gloo":
                e = e.to(self.device)
            dist.all_reduce(e, group=self.group)
        return float(e[0]), float(e[1])

    def kinetic_energy(self) -> float:
        return self.energies()[0]


class ShardedTreeSystem(ShardedDirectSystem):
    """Barnes-Hut over W ranks: replicated tree, partitioned walk (SURVEY 8e "replicate the tree +
    all-gather positions").  Per step: all-gather of the float4 bodies -> every rank builds the SAME
    tree (the build is deterministic) -> rank r walks the bodies at positions [r n/W, (r+1) n/W) of
    the tree's Morton order, writing their accelerations at the bodies' original indices of a zeroed
    [n,4] buffer -> one reduce-scatter (sum) hands each rank the rows of its own index range.  Every
    row is written by exactly one rank, so the sum is exact: the result equals the single-GPU walk
    bit for bit whenever both use the same walk variant.  The build is replicated work (0.43 ms at
    2^20 bodies against 2.0/W ms of walk): this shards the walk, it does not scale the build."""

    def __init__(self, ic: dict, G: float, eps: float, theta: float, backend=None, group=None, device=None):
        super().__init__(ic, G, eps, backend=backend, group=group, device=device, mode="gather")
        self.theta = float(theta)
        self.eps = float(eps)
        self.contrib = torch.zeros((self.S * self.world, 4), dtype=torch.float32, device=self.device)

    def compute_forces(self, out: torch.Tensor):
        n, W, r, S = self.n, self.world, self.rank, self.S
        src = self.posm
        if dist.is_initialized():
            dist.all_gather_into_tensor(self.posm_all, self.posm, group=self.group)
            src = self.posm_all  # padding (zero-mass bodies at the origin) is all at the tail: rows >= n
        c = self.contrib
        c.zero_()
        lo, hi = (n * r) // W, (n * (r + 1)) // W
        self.backend.tree_forces(src[:n], lo, hi - lo, self.theta, self.G, self.eps, c[:n])
        if W == 1:
            out.copy_(c[:S])
            return
        if self._use_reduce_scatter is None:
            self._use_reduce_scatter = dist.get_backend(self.group) != "gloo"
        if self._use_reduce_scatter:
            try:
                dist.reduce_scatter_tensor(out, c, op=dist.ReduceOp.SUM, group=self.group)
                return
            except (RuntimeError, NotImplementedError, AttributeError):
                self._use_reduce_scatter = False
        dist.all_reduce(c, group=self.group)
        out.copy_(c[r * S:(r + 1) * S])


def layer_owner(gz: int, world: int) -> np.ndarray:
    """Owner rank of every z layer of the grid: contiguous slabs, rank r owns layers
    [r gz // world, (r+1) gz // world) (ranks may own none when gz < world)."""
    edges = [(r * gz) // world for r in range(world + 1)]
    owner = np.empty(gz, dtype=np.int64)
    for r in range(world):
        owner[edges[r]:edges[r + 1]] = r
    return owner


class ShardedHashSystem:
    """Spatial hash sharded by z-slabs of cells: the linear cell id
    x + y gx + z gx gy (force_spatial_hash.cu:48) makes a range of z layers a contiguous block of
    the cell-ordered body list, so rank r owns the bodies of the layers [r gz / W, (r+1) gz / W).
    One force evaluation (after the drift), with ONE host synchronisation and four collectives:

        global box    local min/max (HIP) -> one all-reduce (max of {-lo, hi}) -> every rank bins on the
                      SAME grid (box padded by 0.001, dims = ceil(extent/cell)+1, like the reference's
                      build, force_spatial_hash.cu:225-246); the grid is derived on the device
        partition     ONE pass over the rank's bodies (csrc/slab.hip): new layer and owner of every
                      body, 64-byte rows grouped by owner, the rank's row of the W x W send matrix and its
                      bodies-per-layer histogram -> one all-reduce (sum): every rank now knows who sends
                      it how many rows, how many bodies it will own and how many its halo layers hold
        host sync     box, matrix, histogram (the grid size decides validity, as in the reference)
        migration     one all-to-all of rows {x,y,z,m, v, a_old, id} (exact sizes; 64 B/body, few bodies)
        own grid      cell-ordered build of the rank's bodies: its lowest and highest layer are the head
                      and the tail of that order -- the halo its neighbours need, no selection pass
        halo          one all-to-all of those two layers (16 B/body), ASYNCHRONOUS ...
        own x own     ... while the 27-cell kernel evaluates every own body against the own bodies
        halo grid     after the halo has arrived: a grid over the two received layers, and the kernel again
                      for the rank's two boundary layers against it, accumulated

    One Velocity-Verlet step = drift, the above, kick (ref: Integrator::integrate).  Grids that are too
    sparse for the per-cell start arrays of the two-grid kernel, or taller than HIST_CAP layers, take the
    older one-grid path (`_compute_forces_generic`: exchange, then one build and one launch)."""

    HIST_CAP = 4096

    def __init__(self, ic: dict, G: float, eps: float, cell_size: float, cutoff: float,
                 backend=None, group=None, device=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = backend
        self.device = torch.device(device) if device is not None else backend.device
        self.G, self.eps = float(G), float(eps)
        self.cell, self.cutoff = float(cell_size), float(cutoff)
        n = int(ic["pos_x"].size)
        self.n = n
        # initial ownership: every rank evaluates the same global grid on the host
        lo, hi, dims = self._grid_from_bounds(
            [float(ic[k].min()) for k in ("pos_x", "pos_y", "pos_z")] +
            [float(ic[k].max()) for k in ("pos_x", "pos_y", "pos_z")])
        cz = np.clip(np.floor((ic["pos_z"] - np.float32(lo[2])) / np.float32(self.cell)).astype(np.int64),
                     0, dims[2] - 1)
        mine = np.nonzero(layer_owner(dims[2], self.world)[cz] == self.rank)[0]
        posm = np.stack([ic["pos_x"][mine], ic["pos_y"][mine], ic["pos_z"][mine], ic["mass"][mine]], 1)
        vel = np.zeros((mine.size, 4), np.float32)
        for k, f in enumerate(("vel_x", "vel_y", "vel_z")):
            if f in ic:
                vel[:, k] = ic[f][mine]
        dev = self.device
        # the rank's bodies live at the front of capacity arrays (bodies arrive and leave every step);
        # self.posm / vel / acc / gid are views of the first n rows
        self._buf = {"posm": torch.from_numpy(np.ascontiguousarray(posm.astype(np.float32))).to(dev),
                     "vel": torch.from_numpy(vel).to(dev),
                     "acc": torch.zeros((mine.size, 4), dtype=torch.float32, device=dev),
                     "acc2": torch.zeros((mine.size, 4), dtype=torch.float32, device=dev),  # swapped with acc
                     "gid": torch.from_numpy(mine.astype(np.int32)).to(dev)}  # global ids (verification only)
        self._set_count(mine.size)
        self.halo_bodies = 0
        self.migrated = 0
        self.path = ""  # "two-grid" (overlapped) or "one-grid" (fallback), of the last evaluation

    # -- storage ------------------------------------------------------------------------------
    def _set_count(self, n):
        self.posm, self.vel, self.acc, self.gid = (self._buf[k][:n] for k in ("posm", "vel", "acc", "gid"))

    def _reserve(self, n):
        """Capacity for n bodies (+ 1/8 slack), keeping the current rows."""
        if self._buf["posm"].shape[0] >= n:
            return
        cap = n + n // 8 + 1024
        for k, t in self._buf.items():
            grown = torch.empty((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            grown[:t.shape[0]] = t
            self._buf[k] = grown
        self._set_count(self.posm.shape[0])

    def _replace(self, posm, vel, acc, gid):
        self._buf = {"posm": posm, "vel": vel, "acc": acc, "acc2": torch.empty_like(acc), "gid": gid}
        self._set_count(posm.shape[0])

    def _adopt_new_accelerations(self):
        """a <- a_new by swapping the two acceleration arrays (no copy)."""
        self._buf["acc"], self._buf["acc2"] = self._buf["acc2"], self._buf["acc"]
        self._set_count(self.posm.shape[0])

    # -- grid --------------------------------------------------------------------------------
    def _grid_from_bounds(self, raw6):
        f = np.float32
        lo = [f(raw6[a]) - f(0.001) for a in range(3)]   # force_spatial_hash.cu:225-231
        hi = [f(raw6[3 + a]) + f(0.001) for a in range(3)]
        dims = [int(np.ceil((hi[a] - lo[a]) / f(self.cell))) + 1 for a in range(3)]  # :244-246
        return [float(v) for v in lo], [float(v) for v in hi], dims

    def _global_box(self):
        """{min x,y,z, max x,y,z} of all ranks' bodies as a device tensor: ONE all-reduce (max of {-lo, hi})."""
        if self.posm.shape[0] > 0:
            b = self.backend.bbox(self.posm)
        else:
            b = torch.tensor([3e38] * 3 + [-3e38] * 3, dtype=torch.float32, device=self.device)
        if self.world > 1:
            m = torch.cat([-b[:3], b[3:]])
            dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
            b = torch.cat([-m[:3], m[3:]])
        return b

    def compute_forces(self):
        b, W, r = self.backend, self.world, self.rank
        if not hasattr(b, "slab_partition"):
            return self._compute_forces_generic()
        dev = self.device
        n_loc = self.posm.shape[0]
        box = self._global_box()
        rows, holes, stats, info = b.slab_partition(self.posm, self.vel, self.acc, self.gid, box, self.cell, W, r,
                                                    self.HIST_CAP)
        if W > 1:  # send matrix (every rank filled its row) + bodies per layer, summed over the ranks
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group)
        box_h, stats_h, info_h = b.to_host(box, stats, info)  # the step's one host synchronisation
        gx, gy, gz, too_tall = (int(v) for v in info_h)
        if gx * gy * gz > 100000000:  # force_spatial_hash.cu:252-254, on every rank alike
            from ._lib import ResourceException
            raise ResourceException("Spatial hash grid too large: reduce cell_size or bounding box")
        lo, hi, dims = self._grid_from_bounds([float(v) for v in box_h])
        if too_tall or dims != [gx, gy, gz]:
            return self._compute_forces_generic((lo, hi, dims))
        M = stats_h[:W * W].reshape(W, W)
        hist = stats_h[W * W:W * W + gz]
        z_lo, z_hi = (r * gz) // W, ((r + 1) * gz) // W
        send, recv = [int(v) for v in M[r]], [int(v) for v in M[:, r]]
        n_new = sum(recv)
        self.migrated = n_loc - send[r]
        # migration: one all-to-all of 64-byte rows with exact sizes -- only of the bodies that change owner;
        # the bodies that stay never move: arrivals take the vacated slots (csrc/slab.hip)
        if W > 1:
            send[r] = recv[r] = 0
            n_leave, n_arrive = sum(send), sum(recv)
            got = torch.empty((n_arrive, 16), dtype=torch.float32, device=dev)
            dist.all_to_all_single(got, rows[:n_leave], output_split_sizes=recv, input_split_sizes=send,
                                   group=self.group)
            self._reserve(max(n_loc, n_new))
            b.slab_fill(self._buf["posm"], self._buf["vel"], self._buf["acc"], self._buf["gid"], n_loc, holes, n_leave,
                        got)
            self._set_count(n_new)
        bounds = lo + hi
        if n_new:
            b.grid_build("own", self.posm, bounds, self.cell, z_lo, z_hi - z_lo)
        # halo: the rank's lowest layer goes to the owner of the layer below, its highest to the owner of the
        # layer above; they are the head and the tail of the own grid's cell order
        halo, work = None, None
        if W > 1:
            owner = layer_owner(gz, W)
            s_split, r_split = [0] * W, [0] * W
            n_head = n_tail = 0
            if z_hi > z_lo:
                if z_lo > 0:
                    n_head = int(hist[z_lo])
                    s_split[int(owner[z_lo - 1])] = n_head
                    r_split[int(owner[z_lo - 1])] = int(hist[z_lo - 1])
                if z_hi < gz:
                    n_tail = int(hist[z_hi - 1])
                    s_split[int(owner[z_hi])] = n_tail
                    r_split[int(owner[z_hi])] = int(hist[z_hi])
            out = torch.empty((n_head + n_tail, 4), dtype=torch.float32, device=dev)
            if n_head:
                b.grid_sorted("own", 0, n_head, out[:n_head])
            if n_tail:
                b.grid_sorted("own", n_new - n_tail, n_tail, out[n_head:])
            halo = torch.empty((sum(r_split), 4), dtype=torch.float32, device=dev)
            work = dist.all_to_all_single(halo, out, output_split_sizes=r_split, input_split_sizes=s_split,
                                          group=self.group, async_op=True)
        # own x own while the halo layers are in flight -- if both grids are dense enough for the per-cell start
        # arrays of the two-grid kernel (stricter than the library's test, cells covered <= 16 bodies + 4096: safe).  Decided HERE, from
        # the layer histogram, before anything is launched: a grid that turns out too sparse after own x own has
        # run would throw that work away.
        acc_new = self._buf["acc2"][:n_new]
        n_halo = 0 if halo is None else int(halo.shape[0])
        layer_cells = gx * gy
        z0h, z1h = max(z_lo - 1, 0), min(z_hi + 1, gz)
        two_grid = ((z_hi - z_lo) * layer_cells <= 4 * n_new + 4096 and
                    (n_halo == 0 or (z1h - z0h) * layer_cells <= 4 * n_halo + 4096))
        if n_new and two_grid:
            two_grid = b.grid_forces("own", "own", z_lo, z_hi - z_lo, self.cutoff, self.G, self.eps, acc_new, False)
        if work is not None:
            work.wait()
        self.halo_bodies = n_halo
        if n_new and two_grid and self.halo_bodies:
            z0, z1 = max(z_lo - 1, 0), min(z_hi + 1, gz)
            b.grid_build("halo", halo, bounds, self.cell, z0, z1 - z0)
            for z in sorted({z_lo, z_hi - 1}):  # the boundary layers against the halo grid
                two_grid = two_grid and b.grid_forces("own", "halo", z, 1, self.cutoff, self.G, self.eps, acc_new, True)
        if n_new and not two_grid:  # too sparse for the per-cell start arrays: one grid over own + halo
            allb = self.posm if not self.halo_bodies else torch.cat([self.posm, halo]).contiguous()
            acc_new.copy_(b.hash_forces(allb, bounds, self.cell, self.cutoff, self.G, self.eps)[:n_new])
        self.path = "two-grid" if two_grid else "one-grid"
        self.acc_new = acc_new

    def _global_grid(self):
        return self._grid_from_bounds(self._global_box().cpu().numpy().tolist())

    # -- variable-size exchange: rows of `tensors` selected by `dest` go to their rank -----------
    def _exchange(self, tensors, dest):
        """All tensors travel as ONE packed float32 matrix (integer columns bit-cast), so an exchange
        is two collectives: the counts and the rows."""
        W = self.world
        cols, kinds = [], []
        for t in tensors:
            if t.dtype == torch.float32:
                c = t.reshape(t.shape[0], -1)
            elif t.dtype in (torch.int64, torch.int32):
                c = t.to(torch.int32).view(torch.float32).reshape(t.shape[0], 1)  # ids < 2^31
            else:
                raise TypeError(t.dtype)
            cols.append(c)
            kinds.append((t.dtype, t.shape[1:], c.shape[1]))
        packed = torch.cat(cols, 1) if len(cols) > 1 else cols[0]
        order = torch.argsort(dest, stable=True)
        counts = torch.bincount(dest, minlength=W).to(torch.int64)
        recv_counts = torch.empty_like(counts)
        dist.all_to_all_single(recv_counts, counts, group=self.group)
        send_split, recv_split = counts.tolist(), recv_counts.tolist()
        src = packed.index_select(0, order).contiguous()
        dst = torch.empty((sum(recv_split), packed.shape[1]), dtype=torch.float32, device=packed.device)
        dist.all_to_all_single(dst, src, output_split_sizes=recv_split, input_split_sizes=send_split,
                               group=self.group)
        out, c0 = [], 0
        for dtype, tail, width in kinds:
            piece = dst[:, c0:c0 + width]
            c0 += width
            if dtype in (torch.int64, torch.int32):
                out.append(piece.contiguous().view(torch.int32).reshape(-1).to(dtype))
            else:
                out.append(piece.reshape((dst.shape[0],) + tuple(tail)).contiguous())
        return out

    def _compute_forces_generic(self, grid=None):
        """The one-grid path (round 1): exchange first, then ONE build and ONE launch over [own; halo].
        Generic torch selection ops and a host round trip per exchange; kept for very tall or very sparse
        grids and for backends without the slab operations."""
        b = self.backend
        lo, hi, dims = grid if grid is not None else self._global_grid()
        self.path = "one-grid"
        gz = dims[2]
        owner_np = layer_owner(gz, self.world)
        cz = b.cell_z(self.posm, lo[2], self.cell, gz).to(torch.int64) if self.posm.shape[0] else \
            torch.empty(0, dtype=torch.int64, device=self.device)
        halo = None
        if self.world > 1:
            owner = torch.from_numpy(owner_np).to(self.device)
            dest = owner[cz]
            self.migrated = int((dest != self.rank).sum().item())
            # every rank takes part in the exchange even when nothing moves
            posm, vel, acc, gid, cz = self._exchange([self.posm, self.vel, self.acc, self.gid, cz], dest)
            self._replace(posm, vel, acc, gid)
            mine = np.nonzero(owner_np == self.rank)[0]
            send_rows, send_dest = [], []
            if mine.size:
                z_lo, z_hi = int(mine[0]), int(mine[-1]) + 1
                if z_lo > 0:
                    rows = torch.nonzero(cz == z_lo).flatten()
                    send_rows.append(rows)
                    send_dest.append(torch.full_like(rows, int(owner_np[z_lo - 1])))
                if z_hi < gz:
                    rows = torch.nonzero(cz == z_hi - 1).flatten()
                    send_rows.append(rows)
                    send_dest.append(torch.full_like(rows, int(owner_np[z_hi])))
            if send_rows:
                rows, hd = torch.cat(send_rows), torch.cat(send_dest)
            else:
                rows = torch.empty(0, dtype=torch.int64, device=self.device)
                hd = torch.empty(0, dtype=torch.int64, device=self.device)
            (halo,) = self._exchange([self.posm.index_select(0, rows)], hd)
            self.halo_bodies = int(halo.shape[0])
        n_loc = self.posm.shape[0]
        self.acc_new = self._buf["acc2"][:n_loc]
        if n_loc == 0:
            return
        allb = self.posm if halo is None or halo.shape[0] == 0 else torch.cat([self.posm, halo]).contiguous()
        self.acc_new.copy_(b.hash_forces(allb, lo + hi, self.cell, self.cutoff, self.G, self.eps)[:n_loc])

    def initial_forces(self):
        self.compute_forces()
        self._adopt_new_accelerations()

    def step(self, dt: float):
        b = self.backend
        if self.posm.shape[0]:
            b.drift(self.posm, self.vel, self.acc, dt)
        # self.acc (= a_old) migrates with the bodies inside compute_forces
        self.compute_forces()
        if self.posm.shape[0]:
            b.kick(self.vel, self.acc, self.acc_new, dt)
        self._adopt_new_accelerations()

    def gather_global(self):
        """(gid, posm, vel, acc) of all bodies on every rank, ordered by global id (test helper)."""
        parts = [self.gid.double().unsqueeze(1), self.posm.double(), self.vel.double(), self.acc.double()]
        loc = torch.cat(parts, 1).contiguous()
        if self.world == 1:
            full = loc
        else:
            counts = torch.tensor([loc.shape[0]], dtype=torch.int64, device=self.device)
            allc = [torch.zeros_like(counts) for _ in range(self.world)]
            dist.all_gather(allc, counts, group=self.group)
            sizes = [int(c.item()) for c in allc]
            cap = max(max(sizes), 1)
            padded = torch.zeros((cap, loc.shape[1]), dtype=loc.dtype, device=self.device)
            padded[: loc.shape[0]] = loc
            bufs = [torch.empty_like(padded) for _ in sizes]
            dist.all_gather(bufs, padded, group=self.group)
            full = torch.cat([b[:s] for b, s in zip(bufs, sizes)])
        full = full[torch.argsort(full[:, 0])].cpu().numpy()
        return full[:, 0].astype(np.int64), full[:, 1:5].astype(np.float32), \
            full[:, 5:9].astype(np.float32), full[:, 9:13].astype(np.float32)
