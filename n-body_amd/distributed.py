"""Direct N^2 sharded over the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI).

The reference is single-GPU; this is new.  Bodies shard by contiguous index
range (rank r owns [r*S, (r+1)*S), S = ceil(N / world), the tail padded with zero-mass bodies that
exert no force).  One Velocity-Verlet step (ref: Integrator::integrate, integrator.cu:224-238):

    drift own bodies                                   (compute stream)
    all-gather {x,y,z,m} float4 of every shard         (RCCL stream; 16 B/body, 2 MiB per rank
                                                        at N = 2^20, once per step)
    forces on own targets from OWN sources             (compute stream, overlaps the gather)
    wait for the gather
    forces from the shards left and right of the own range, accumulated
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
from .api import Context, direct_forces_packed


class HipBackend:
    """The product path: every operation is a HIP kernel behind the C ABI."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.device = ctx.torch_device

    def drift(self, posm, vel, acc, dt):
        check(self.ctx._lib.nbody_hip_drift_packed(self.ctx.handle, posm.data_ptr(), vel.data_ptr(),
                                                   acc.data_ptr(), posm.shape[0], dt))

    def kick(self, vel, acc_old, acc_new, dt):
        check(self.ctx._lib.nbody_hip_kick_packed(self.ctx.handle, vel.data_ptr(), acc_old.data_ptr(),
                                                  acc_new.data_ptr(), vel.shape[0], dt))

    def forces(self, targets, sources, G, eps2, out, accumulate):
        direct_forces_packed(self.ctx, targets, sources, G, eps2, out=out, accumulate=accumulate)


def shard_bounds(n: int, world: int, rank: int):
    """(shard size S, lo, hi) of `rank`: contiguous, every shard padded to S = ceil(n/world)."""
    s = (n + world - 1) // world
    lo = min(n, rank * s)
    hi = min(n, lo + s)
    return s, lo, hi


class ShardedDirectSystem:
    """State of one rank: its shard as packed float4 arrays + the gathered source buffer."""

    def __init__(self, ic: dict, G: float, eps: float, backend=None, group=None, device=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = backend
        self.device = torch.device(device) if device is not None else backend.device
        self.G = float(G)
        self.eps2 = float(np.float32(eps) * np.float32(eps))
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
        if self.world > 1:
            work = dist.all_gather_into_tensor(self.posm_all, self.posm, group=self.group,
                                               async_op=True)
        # own shard against itself while the gather is in flight
        b.forces(self.posm, self.posm, self.G, self.eps2, out, False)
        if work is not None:
            work.wait()  # orders the compute stream after the collective
            if r > 0:
                b.forces(self.posm, self.posm_all[:r * S], self.G, self.eps2, out, True)
            if r < self.world - 1:
                b.forces(self.posm, self.posm_all[(r + 1) * S:], self.G, self.eps2, out, True)

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

    def kinetic_energy(self) -> float:
        """0.5 sum m v^2 over all ranks: local fp64 sum + all-reduce of one double."""
        m = self.posm[:, 3].double()
        v2 = (self.vel[:, :3].double() ** 2).sum(1)
        ke = (0.5 * m * v2).sum().reshape(1)
        if self.world > 1:
            dist.all_reduce(ke, group=self.group)
        return float(ke.item())
