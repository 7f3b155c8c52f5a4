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
from .api import Context, direct_forces_pair_packed, direct_forces_packed


class HipBackend:
    """The product path: every operation is a HIP kernel behind the C ABI."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.device = ctx.torch_device
        self._grid = self._tree = None

    def close(self):
        """Frees the lazily created grid / tree handles (also run at garbage collection)."""
        lib = self.ctx._lib
        if self._grid is not None and self.ctx.handle.value:
            lib.nbody_hip_grid_destroy(self._grid)
        if self._tree is not None and self.ctx.handle.value:
            lib.nbody_hip_tree_destroy(self._tree)
        self._grid = self._tree = None

    def __del__(self):
        try:
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
        """Accelerations [n,4] of every body of posm_all on the grid with the given padded bounds."""
        n = posm_all.shape[0]
        if getattr(self, "_grid_cap", 0) < n or getattr(self, "_grid_cell", None) != cell:
            if getattr(self, "_grid", None) is not None:
                self.ctx._lib.nbody_hip_grid_destroy(self._grid)
            h = C.c_void_p()
            cap = int(n * 1.5) + 1024
            check(self.ctx._lib.nbody_hip_grid_create(self.ctx.handle, cap, cell, C.byref(h)))
            self._grid, self._grid_cap, self._grid_cell = h, cap, cell
        b = (C.c_float * 6)(*bounds)
        check(self.ctx._lib.nbody_hip_grid_build_packed(self._grid, posm_all.data_ptr(), n, b))
        acc = torch.empty((n, 4), dtype=torch.float32, device=posm_all.device)
        check(self.ctx._lib.nbody_hip_grid_compute_forces_packed(self._grid, cutoff, G, eps,
                                                                 acc.data_ptr()))
        return acc


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
    the cell-ordered body list, so rank r owns the bodies of its layers.  Per force evaluation:

        global box    local min/max (HIP) -> all-reduce MIN/MAX of 6 floats -> every rank bins on
                      the SAME grid (box padded by 0.001, dims = ceil(extent/cell)+1, like the
                      reference's build, force_spatial_hash.cu:225-246)
        migration     bodies whose layer now belongs to another rank move there (all-to-all of
                      {x,y,z,m}, velocity, previous acceleration: 48 B/body, few bodies per step)
        halo          the bodies of a rank's lowest / highest layer are sent to the owner of the
                      layer below / above (16 B/body: one cell layer per direction)
        forces        27-cell kernel on [own bodies; halo bodies]; only the own part is kept

    One Velocity-Verlet step = drift, the above, kick (ref: Integrator::integrate)."""

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
        self.posm = torch.from_numpy(np.ascontiguousarray(posm.astype(np.float32))).to(dev)
        self.vel = torch.from_numpy(vel).to(dev)
        self.acc = torch.zeros((mine.size, 4), dtype=torch.float32, device=dev)
        self.gid = torch.from_numpy(mine.astype(np.int64)).to(dev)  # global ids (verification only)
        self.halo_bodies = 0
        self.migrated = 0

    # -- grid --------------------------------------------------------------------------------
    def _grid_from_bounds(self, raw6):
        f = np.float32
        lo = [f(raw6[a]) - f(0.001) for a in range(3)]   # force_spatial_hash.cu:225-231
        hi = [f(raw6[3 + a]) + f(0.001) for a in range(3)]
        dims = [int(np.ceil((hi[a] - lo[a]) / f(self.cell))) + 1 for a in range(3)]  # :244-246
        return [float(v) for v in lo], [float(v) for v in hi], dims

    def _global_grid(self):
        b = self.backend.bbox(self.posm) if self.posm.shape[0] > 0 else torch.tensor(
            [3e38] * 3 + [-3e38] * 3, dtype=torch.float32, device=self.device)
        if self.world > 1:
            lo, hi = b[:3].clone(), b[3:].clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            b = torch.cat([lo, hi])
        return self._grid_from_bounds(b.cpu().numpy().tolist())

    # -- variable-size exchange: rows of `tensors` selected by `dest` go to their rank -----------
    def _exchange(self, tensors, dest):
        """All tensors travel as ONE packed float32 matrix (integer columns bit-cast), so an exchange
        is two collectives: the counts and the rows."""
        W = self.world
        cols, kinds = [], []
        for t in tensors:
            if t.dtype == torch.float32:
                c = t.reshape(t.shape[0], -1)
            elif t.dtype == torch.int64:
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
            if dtype == torch.int64:
                out.append(piece.contiguous().view(torch.int32).reshape(-1).to(torch.int64))
            else:
                out.append(piece.reshape((dst.shape[0],) + tuple(tail)).contiguous())
        return out

    def compute_forces(self):
        b = self.backend
        lo, hi, dims = self._global_grid()
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
            self.posm, self.vel, self.acc, self.gid, cz = self._exchange(
                [self.posm, self.vel, self.acc, self.gid, cz], dest)
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
        if n_loc == 0:
            self.acc = torch.zeros((0, 4), dtype=torch.float32, device=self.device)
            return
        allb = self.posm if halo is None or halo.shape[0] == 0 else torch.cat([self.posm, halo]).contiguous()
        acc_all = b.hash_forces(allb, lo + hi, self.cell, self.cutoff, self.G, self.eps)
        self.acc_new = acc_all[:n_loc].contiguous()

    def initial_forces(self):
        self.compute_forces()
        if self.posm.shape[0]:
            self.acc = self.acc_new

    def step(self, dt: float):
        b = self.backend
        if self.posm.shape[0]:
            b.drift(self.posm, self.vel, self.acc, dt)
        # self.acc (= a_old) migrates with the bodies inside compute_forces
        self.compute_forces()
        if self.posm.shape[0]:
            b.kick(self.vel, self.acc, self.acc_new, dt)
            self.acc = self.acc_new

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
