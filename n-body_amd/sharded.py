"""Host-side mirror of include/nbody_hip_comm.h: communicators and the sharded Direct N^2 system whose whole
step (drift -> all-gather || own x own -> shard pairs -> reaction exchange -> fixed-order sum + kick) runs behind
ONE C-ABI call per batch of steps (csrc/sharded.hip).  Nothing here does arithmetic and nothing imports oracle/.

Two launch models (see the header):
    Comm.init_all(ndev, devices, transport)      one process drives all devices (P2P peer copies or RCCL); the
                                                 same device may be listed several times (virtual ranks: tests)
    Comm.init_rank(device, rank, world, id)      one process per GPU over RCCL; Comm.unique_id() on rank 0, the
                                                 128 bytes travel by the launcher's own channel
                                                 (Comm.from_torch_distributed broadcasts them over a
                                                 torch.distributed process group)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, finalizing, load, track

TRANSPORT_P2P, TRANSPORT_RCCL = 0, 1
MAX_RANKS = 32


def shard_bounds(n: int, world: int, rank: int):
    """(S, lo, hi): nbody_hip_shard_bounds"""
    s, lo, hi = C.c_size_t(), C.c_size_t(), C.c_size_t()
    check(load().nbody_hip_shard_bounds(n, world, rank, C.byref(s), C.byref(lo), C.byref(hi)))
    return s.value, lo.value, hi.value


def pair_schedule(world: int, rank: int, S: int):
    """[(i0, i1, shard, j0, j1)]: nbody_hip_pair_schedule (the C++ twin of distributed.pair_schedule)"""
    rows = ((C.c_size_t * 5) * (MAX_RANKS // 2 + 1))()
    k = load().nbody_hip_pair_schedule(world, rank, S, rows, MAX_RANKS // 2 + 1)
    if k < 0:
        check(k)
    return [tuple(int(v) for v in rows[t]) for t in range(k)]


class Comm:
    def __init__(self, handle):
        self._h = handle
        self._lib = load()
        track(self, "comm")

    @staticmethod
    def init_all(ndev: int, devices=None, transport: int = TRANSPORT_P2P) -> "Comm":
        h = C.c_void_p()
        dev = (C.c_int * ndev)(*devices) if devices is not None else None
        check(load().nbody_hip_comm_init_all(ndev, dev, transport, C.byref(h)))
        return Comm(h)

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(load().nbody_hip_comm_unique_id(buf))
        return buf.raw

    @staticmethod
    def init_rank(device: int, rank: int, world: int, uid: bytes) -> "Comm":
        assert len(uid) == 128
        h = C.c_void_p()
        check(load().nbody_hip_comm_init_rank(device, rank, world, C.create_string_buffer(uid, 128), C.byref(h)))
        return Comm(h)

    @staticmethod
    def from_torch_distributed(device: int, group=None) -> "Comm":
        """One rank per process: the id is made on rank 0 and broadcast over the torch.distributed group (any
        backend), then every process joins with ncclCommInitRank."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return Comm.init_rank(device, rank, world, box[0])

    def info(self):
        w, nl, tr = C.c_int(), C.c_int(), C.c_int()
        ranks = (C.c_int * MAX_RANKS)()
        check(self._lib.nbody_hip_comm_info(self._h, C.byref(w), C.byref(nl), C.byref(tr), ranks))
        return {"world": w.value, "nlocal": nl.value, "transport": tr.value, "local_ranks": list(ranks[: nl.value])}

    def close(self):
        if self._h is not None and self._h.value:
            self._lib.nbody_hip_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            if not finalizing():  # (else: closed by the exit hook _lib.close_all, or the runtime is going down)
                self.close()
        except Exception:
            pass


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


class ShardedDirect:
    """nbody_hip_sharded_direct_*: BASELINE config 3 behind the C ABI."""

    def __init__(self, comm: Comm, n: int, G: float, eps: float):
        self.comm, self.n = comm, int(n)
        self._lib = load()
        self._h = C.c_void_p()
        check(self._lib.nbody_hip_sharded_direct_create(comm._h, n, G, eps, C.byref(self._h)))
        track(self, "system")

    def set_state(self, ic: dict):
        a = [_f(ic[k]) for k in ("pos_x", "pos_y", "pos_z", "mass")]
        v = [_f(ic.get(k)) for k in ("vel_x", "vel_y", "vel_z")]
        assert all(x.size == self.n for x in a)
        if any(x is None for x in v):
            v = [None, None, None]
        ptr = [x.ctypes.data if x is not None else None for x in a + v]
        check(self._lib.nbody_hip_sharded_direct_set_state(self._h, *ptr))

    def forces(self):
        check(self._lib.nbody_hip_sharded_direct_forces(self._h))

    def step(self, dt: float, steps: int = 1):
        check(self._lib.nbody_hip_sharded_direct_step(self._h, dt, steps))

    def time_steps(self, dt: float, warmup: int, steps: int) -> float:
        ms = C.c_float()
        check(self._lib.nbody_hip_sharded_direct_time_steps(self._h, dt, warmup, steps, C.byref(ms)))
        return ms.value

    def synchronize(self):
        check(self._lib.nbody_hip_sharded_direct_synchronize(self._h))

    def get_state(self, gather: bool = True, what=("pos", "vel", "acc")) -> dict:
        """numpy arrays of the whole system ("pos_x" .. "acc_z"); rows of remote ranks are only filled with gather"""
        names = [f"{w}_{c}" for w in ("pos", "vel", "acc") for c in "xyz"]
        out = {k: np.zeros(self.n, np.float32) for k in names if k[:3] in what}
        ptr = [out[k].ctypes.data if k in out else None for k in names]
        check(self._lib.nbody_hip_sharded_direct_get_state(self._h, *ptr, 1 if gather else 0))
        return out

    def energies(self):
        ke, pe = C.c_double(), C.c_double()
        check(self._lib.nbody_hip_sharded_direct_energies(self._h, C.byref(ke), C.byref(pe)))
        return ke.value, pe.value

    def compute_forces(self, d_particles):
        """the plugin form on a whole-system ParticleData (api.ParticleData on the first local rank's device)"""
        s = d_particles.struct()
        check(self._lib.nbody_hip_sharded_direct_compute_forces(self._h, C.byref(s)))

    def close(self):
        if self._h is not None and self._h.value:
            if self.comm._h is not None and self.comm._h.value:  # (the system borrows the communicator's streams)
                self._lib.nbody_hip_sharded_direct_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            if not finalizing():  # (else: closed by the exit hook _lib.close_all, or the runtime is going down)
                self.close()
        except Exception:
            pass


class ShardedHash:
    """nbody_hip_sharded_hash_*: BASELINE config 5 (z-slab shards, migration, halo exchange) behind the C ABI."""

    def __init__(self, comm: Comm, n: int, G: float, eps: float, cell_size: float, cutoff: float):
        self.comm, self.n = comm, int(n)
        self._lib = load()
        self._h = C.c_void_p()
        check(self._lib.nbody_hip_sharded_hash_create(comm._h, n, G, eps, cell_size, cutoff, C.byref(self._h)))
        track(self, "system")

    def set_state(self, ic: dict):
        a = [_f(ic[k]) for k in ("pos_x", "pos_y", "pos_z", "mass")]
        v = [_f(ic.get(k)) for k in ("vel_x", "vel_y", "vel_z")]
        assert all(x.size == self.n for x in a)
        if any(x is None for x in v):
            v = [None, None, None]
        check(self._lib.nbody_hip_sharded_hash_set_state(self._h, *[x.ctypes.data if x is not None else None for x in a + v]))

    def forces(self):
        check(self._lib.nbody_hip_sharded_hash_forces(self._h))

    def step(self, dt: float, steps: int = 1):
        check(self._lib.nbody_hip_sharded_hash_step(self._h, dt, steps))

    def time_steps(self, dt: float, warmup: int, steps: int) -> float:
        ms = C.c_float()
        check(self._lib.nbody_hip_sharded_hash_time_steps(self._h, dt, warmup, steps, C.byref(ms)))
        return ms.value

    def synchronize(self):
        check(self._lib.nbody_hip_sharded_hash_synchronize(self._h))

    def get_state(self, what=("pos", "vel", "acc")) -> dict:
        """numpy arrays of the whole system by global body id (rows of bodies held by other processes stay 0)"""
        names = [f"{w}_{c}" for w in ("pos", "vel", "acc") for c in "xyz"]
        out = {k: np.zeros(self.n, np.float32) for k in names if k[:3] in what}
        check(self._lib.nbody_hip_sharded_hash_get_state(self._h, *[out[k].ctypes.data if k in out else None for k in names]))
        return out

    def info(self) -> dict:
        dims, tg = (C.c_int * 3)(), C.c_int()
        mig, halo = C.c_ulonglong(), C.c_ulonglong()
        counts = (C.c_ulonglong * MAX_RANKS)()
        check(self._lib.nbody_hip_sharded_hash_info(self._h, dims, C.byref(tg), C.byref(mig), C.byref(halo), counts))
        nl = self.comm.info()["nlocal"]
        return {"dims": list(dims), "two_grid": bool(tg.value), "migrated": mig.value, "halo_bodies": halo.value,
                "local_counts": list(counts[:nl])}

    def close(self):
        if self._h is not None and self._h.value:
            if self.comm._h is not None and self.comm._h.value:
                self._lib.nbody_hip_sharded_hash_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            if not finalizing():  # (else: closed by the exit hook _lib.close_all, or the runtime is going down)
                self.close()
        except Exception:
            pass
