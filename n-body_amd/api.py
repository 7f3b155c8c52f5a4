"""Host-side mirror of the reference's C++ interface for the hot path, over the C ABI.

Names, argument meaning and error behaviour follow the reference headers so that the parity
tests read like the reference's own tests:

  ParticleData / ParticleDataManager    include/nbody/types.hpp:234-276, particle_data.hpp:9-36
  ForceCalculator / DirectForceCalculator / createForceCalculator
                                        include/nbody/force_calculator.hpp:36-231
  Integrator                            include/nbody/integrator.hpp:51-150
  SimulationConfig / ForceMethod        include/nbody/types.hpp:66-101,301-313
  validate*                             src/utils/error_handling.cpp:46-123

PyTorch is used only as the owner of device memory and of the HIP stream; every computation is
a call into libnbody_hip.so.  The compiled-language facade (C++) lives in n-body_amd/facade/.
"""
from __future__ import annotations

import ctypes as C
import enum
import math
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import (FIELDS, ParticleDataStruct, StateException, ValidationException, check)


class ForceMethod(enum.IntEnum):  # types.hpp:66-70
    DIRECT_N2 = 0
    BARNES_HUT = 1
    SPATIAL_HASH = 2


class InitDistribution(enum.IntEnum):  # types.hpp:82-86
    UNIFORM = 0
    SPHERICAL = 1
    DISK = 2


@dataclass
class SimulationConfig:  # types.hpp:301-313 (same defaults)
    particle_count: int = 10000
    init_distribution: InitDistribution = InitDistribution.SPHERICAL
    force_method: ForceMethod = ForceMethod.DIRECT_N2
    dt: float = 0.001
    G: float = 1.0
    softening: float = 0.1
    barnes_hut_theta: float = 0.5
    spatial_hash_cell_size: float = 1.0
    spatial_hash_cutoff: float = 2.0
    cuda_block_size: int = 256


# ---- validators (error_handling.cpp:46-123; same messages) ------------------------------------

def _finite(v):
    return not (math.isnan(v) or math.isinf(v))


def validateParticleCountRange(count):
    if count == 0:
        raise ValidationException("Particle count must be greater than 0")
    if count > 100000000:
        raise ValidationException("Particle count exceeds maximum supported (100M)")


def validateTimeStep(dt):  # same check order as error_handling.cpp:91-103
    if dt <= 0:
        raise ValidationException("Time step must be positive")
    if math.isnan(dt) or math.isinf(dt):
        raise ValidationException("Time step must be a finite number")
    if dt > 1.0:
        raise ValidationException("Time step is too large (max 1.0)")


def validateSoftening(eps):  # error_handling.cpp:105-113
    if eps < 0:
        raise ValidationException("Softening parameter must be non-negative")
    if math.isnan(eps) or math.isinf(eps):
        raise ValidationException("Softening parameter must be a finite number")


def validateTheta(theta):  # error_handling.cpp:115-123
    if theta < 0 or theta > 2.0:
        raise ValidationException("Barnes-Hut theta must be between 0 and 2")
    if math.isnan(theta) or math.isinf(theta):
        raise ValidationException("Barnes-Hut theta must be a finite number")


def validateSimulationConfig(cfg: SimulationConfig):
    validateParticleCountRange(cfg.particle_count)
    validateTimeStep(cfg.dt)
    validateSoftening(cfg.softening)
    if cfg.force_method == ForceMethod.BARNES_HUT:
        validateTheta(cfg.barnes_hut_theta)
    if cfg.G <= 0 or not _finite(cfg.G):
        raise ValidationException("Gravitational constant must be positive and finite")
    if cfg.force_method == ForceMethod.SPATIAL_HASH:
        if cfg.spatial_hash_cell_size <= 0 or not _finite(cfg.spatial_hash_cell_size):
            raise ValidationException("Spatial hash cell size must be positive and finite")
        if cfg.spatial_hash_cutoff <= 0 or not _finite(cfg.spatial_hash_cutoff):
            raise ValidationException("Spatial hash cutoff must be positive and finite")
    if cfg.cuda_block_size <= 0 or cfg.cuda_block_size > 1024:
        raise ValidationException("CUDA block size must be between 1 and 1024")


# ---- context -------------------------------------------------------------------------------

class Context:
    """One per (process, device).  Launches go to torch's current stream of that device so that
    torch.cuda events / synchronize() order against them."""

    def __init__(self, device: int | None = None):
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.DeviceException("no HIP device available (this library has no CPU fallback)")
        if device is None:
            device = torch.cuda.current_device()
        self.device = int(device)
        self.torch_device = torch.device("cuda", self.device)
        stream = torch.cuda.current_stream(self.torch_device).cuda_stream
        h = C.c_void_p()
        check(lib.nbody_hip_ctx_create(C.byref(h), self.device, C.c_void_p(stream)))
        self._h = h
        self._lib = lib
        _lib.track(self, "context")

    @property
    def handle(self):
        return self._h

    def use_stream(self, stream: "torch.cuda.Stream | None"):
        ptr = None if stream is None else C.c_void_p(stream.cuda_stream)
        check(self._lib.nbody_hip_ctx_set_stream(self._h, ptr))

    def synchronize(self):
        check(self._lib.nbody_hip_ctx_synchronize(self._h))

    def capture(self):
        """`with ctx.capture() as rec: ...calls...` records the calls into rec.graph (a StepGraph)
        instead of executing them (include/nbody_hip.h, "step graphs")."""
        return _Capture(self)

    def tuning(self, variant=-1, targets_per_lane=0, source_splits=0):
        check(self._lib.nbody_hip_direct_tuning(self._h, variant, targets_per_lane, source_splits))

    def deterministic(self, mode=True):
        """Direct forces bitwise reproducible (nbody_hip_direct_deterministic): 0 / False = fp64 atomics, 1 / True =
        slot planes when they fit the budget (default), 2 = slot planes required (else ResourceException)."""
        check(self._lib.nbody_hip_direct_deterministic(self.handle, int(mode)))

    def slotBudget(self, nbytes: int = 0):
        """most bytes of slot planes the deterministic Direct form may take (0 = default, 24 GiB)"""
        check(self._lib.nbody_hip_direct_slot_budget(self.handle, nbytes))

    def directInfo(self, count: int = 0, eps2: float = 1.0) -> dict:
        """What a Direct all-pairs call at `count` bodies runs, and what the last call ran (nbody_hip_direct_info)."""
        from ._lib import DirectInfoStruct
        s = DirectInfoStruct()
        check(self._lib.nbody_hip_direct_info(self.handle, count, eps2, C.byref(s)))
        out = {k: getattr(s, k) for k, _ in DirectInfoStruct._fields_ if not k.startswith("reserved")}
        out["kernel_name"] = ("one-sided", "symmetric+atomics", "symmetric+slots")[s.kernel]
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.nbody_hip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            if not _lib.finalizing():  # (else: closed by the exit hook _lib.close_all, or the runtime is going down)
                self.close()
        except Exception:
            pass


class StepGraph:
    """A recorded sequence of launches (hipGraphExec); launch(k) replays it k times in order."""

    def __init__(self, ctx: Context, handle):
        self._ctx, self._h = ctx, handle
        _lib.track(self, "graph")

    def launch(self, times: int = 1):
        check(self._ctx._lib.nbody_hip_graph_launch(self._h, int(times)))

    def close(self):
        if self._h is not None and self._h.value and self._ctx._h.value:
            self._ctx._lib.nbody_hip_graph_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            if not _lib.finalizing():  # (else: closed by the exit hook _lib.close_all, or the runtime is going down)
                self.close()
        except Exception:
            pass


class _Capture:
    def __init__(self, ctx: Context):
        self.ctx, self.graph = ctx, None

    def __enter__(self):
        check(self.ctx._lib.nbody_hip_capture_begin(self.ctx.handle))
        return self

    def __exit__(self, et, ev, tb):
        h = C.c_void_p()
        rc = self.ctx._lib.nbody_hip_capture_end(self.ctx.handle, C.byref(h))  # always restores the context
        if et is None:
            check(rc)
            self.graph = StepGraph(self.ctx, h)
        elif rc == 0:
            self.ctx._lib.nbody_hip_graph_destroy(h)
        return False


_default_ctx: dict[int, Context] = {}


def default_context(device: int | None = None) -> Context:
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


# ---- particle data ---------------------------------------------------------------------------

class ParticleData:
    """13 float32 arrays + count (types.hpp:234-276).  `where` is "device" (torch tensors on the
    GPU) or "host" (numpy arrays)."""

    def __init__(self):
        self.count = 0
        self.where = None
        for f in FIELDS:
            setattr(self, f, None)

    def struct(self) -> ParticleDataStruct:
        s = ParticleDataStruct()
        for f in FIELDS:
            a = getattr(self, f)
            if a is None:
                ptr = None
            elif isinstance(a, torch.Tensor):
                ptr = a.data_ptr()
            else:
                ptr = a.ctypes.data
            setattr(s, f, ptr)
        s.count = self.count
        return s


class ParticleDataManager:
    """particle_data.hpp:9-36 / particle_init.cu:143-283."""

    @staticmethod
    def allocateDevice(data: ParticleData, count: int, device=None):
        validateParticleCountRange(count)
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        # one slab, 64-float (256 B) aligned sub-arrays, zero-initialised (ref zeroes acc*)
        stride = (count + 63) // 64 * 64
        slab = torch.zeros(13 * stride, dtype=torch.float32, device=dev)
        for k, f in enumerate(FIELDS):
            setattr(data, f, slab[k * stride:k * stride + count])
        data._slab = slab
        data.count = count
        data.where = "device"

    @staticmethod
    def freeDevice(data: ParticleData):
        for f in FIELDS:
            setattr(data, f, None)
        data._slab = None
        data.count = 0

    @staticmethod
    def allocateHost(data: ParticleData, count: int):
        for f in FIELDS:
            setattr(data, f, np.zeros(count, dtype=np.float32))
        data.count = count
        data.where = "host"

    @staticmethod
    def freeHost(data: ParticleData):
        for f in FIELDS:
            setattr(data, f, None)
        data.count = 0

    _COPIED = ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "acc_x", "acc_y", "acc_z",
               "mass")  # acc_old is not copied by the reference either (particle_init.cu:257-283)

    @staticmethod
    def copyToDevice(d_data: ParticleData, h_data: ParticleData):
        if h_data.count > d_data.count:
            raise ValidationException("host count exceeds device capacity")
        n = h_data.count
        for f in ParticleDataManager._COPIED:
            getattr(d_data, f)[:n].copy_(torch.from_numpy(np.ascontiguousarray(getattr(h_data, f)[:n])))

    @staticmethod
    def copyToHost(h_data: ParticleData, d_data: ParticleData):
        if d_data.count > h_data.count:
            raise ValidationException("device count exceeds host capacity")
        n = d_data.count
        for f in ParticleDataManager._COPIED:
            getattr(h_data, f)[:n] = getattr(d_data, f)[:n].cpu().numpy()


# ---- host initialisers ------------------------------------------------------------------------

@dataclass
class UniformDistParams:  # types.hpp:330-335
    min_bounds: tuple = (0.0, 0.0, 0.0)
    max_bounds: tuple = (0.0, 0.0, 0.0)
    min_mass: float = 1.0
    max_mass: float = 1.0


@dataclass
class SphericalDistParams:  # types.hpp:346-351
    center: tuple = (0.0, 0.0, 0.0)
    radius: float = 10.0
    min_mass: float = 1.0
    max_mass: float = 1.0


@dataclass
class DiskDistParams:  # types.hpp:362-369
    center: tuple = (0.0, 0.0, 0.0)
    radius: float = 10.0
    thickness: float = 1.0
    min_mass: float = 1.0
    max_mass: float = 1.0
    rotation_speed: float = 1.0


class _Mt19937Floats:
    """std::mt19937(seed) feeding std::uniform_real_distribution<float> the way libstdc++ does
    (bits/random.tcc generate_canonical<float,24>): one 32-bit draw x per variate,
    u = float(x) / 2^32 (x rounded to nearest-even float; u >= 1 -> nextafter(1,0)), result
    a + (b - a) * u in float32.  numpy's legacy RandomState(seed) is the same init_genrand stream."""

    def __init__(self, seed: int):
        self._bits = np.random.RandomState(seed)._bit_generator

    def canonical(self, count: int) -> np.ndarray:
        x = self._bits.random_raw(count).astype(np.uint32)
        u = x.astype(np.float32) / np.float32(4294967296.0)
        return np.minimum(u, np.nextafter(np.float32(1), np.float32(0)))


def _dist(u, a, b):
    a, b = np.float32(a), np.float32(b)
    return u * (b - a) + a


class ParticleInitializer:
    """particle_data.hpp:38-57 / particle_init.cu:286-376: the three host recipes, drawing from the
    generator in the reference's order (per body).  Bit-exact for the uniform box; the sphere and
    disk go through cbrt / sin / cos / acos / sqrt, where numpy and glibc may differ in the last ulp
    (tests/test_system_cpu.py compares with the C++ facade built on libstdc++)."""

    @staticmethod
    def zeroVelocities(h: ParticleData):
        for f in ("vel_x", "vel_y", "vel_z"):
            getattr(h, f)[:] = 0

    @staticmethod
    def zeroAccelerations(h: ParticleData):
        for f in ("acc_x", "acc_y", "acc_z", "acc_old_x", "acc_old_y", "acc_old_z"):
            getattr(h, f)[:] = 0

    @staticmethod
    def initUniform(h: ParticleData, p: UniformDistParams, seed: int = 42):
        n = h.count
        u = _Mt19937Floats(seed).canonical(4 * n).reshape(n, 4)  # per body: x, y, z, mass
        for k, f in enumerate(("pos_x", "pos_y", "pos_z")):
            getattr(h, f)[:] = _dist(u[:, k], p.min_bounds[k], p.max_bounds[k])
        h.mass[:] = _dist(u[:, 3], p.min_mass, p.max_mass)
        ParticleInitializer.zeroVelocities(h)
        ParticleInitializer.zeroAccelerations(h)

    @staticmethod
    def initSpherical(h: ParticleData, p: SphericalDistParams, seed: int = 42):
        n = h.count
        f32 = np.float32
        u = _Mt19937Floats(seed).canonical(4 * n).reshape(n, 4)  # radius, azimuth, polar, mass
        r = np.cbrt(_dist(u[:, 0], 0, 1)) * f32(p.radius)
        theta = _dist(u[:, 1], 0, 1) * f32(2.0) * f32(3.14159265)
        phi = np.arccos(f32(2.0) * _dist(u[:, 2], 0, 1) - f32(1.0))
        h.pos_x[:] = f32(p.center[0]) + r * np.sin(phi) * np.cos(theta)
        h.pos_y[:] = f32(p.center[1]) + r * np.sin(phi) * np.sin(theta)
        h.pos_z[:] = f32(p.center[2]) + r * np.cos(phi)
        h.mass[:] = _dist(u[:, 3], p.min_mass, p.max_mass)
        ParticleInitializer.zeroVelocities(h)
        ParticleInitializer.zeroAccelerations(h)

    @staticmethod
    def initDisk(h: ParticleData, p: DiskDistParams, seed: int = 42):
        n = h.count
        f32 = np.float32
        u = _Mt19937Floats(seed).canonical(4 * n).reshape(n, 4)  # radius, azimuth, height, mass
        r = np.sqrt(_dist(u[:, 0], 0, 1)) * f32(p.radius)
        theta = _dist(u[:, 1], 0, 1) * f32(2.0) * f32(3.14159265)
        z = (_dist(u[:, 2], 0, 1) - f32(0.5)) * f32(p.thickness)
        h.pos_x[:] = f32(p.center[0]) + r * np.cos(theta)
        h.pos_y[:] = f32(p.center[1]) + r * np.sin(theta)
        h.pos_z[:] = f32(p.center[2]) + z
        v = f32(p.rotation_speed) * np.sqrt(r)
        h.vel_x[:] = -v * np.sin(theta)
        h.vel_y[:] = v * np.cos(theta)
        h.vel_z[:] = 0
        h.mass[:] = _dist(u[:, 3], p.min_mass, p.max_mass)
        ParticleInitializer.zeroAccelerations(h)


# ---- force calculators -----------------------------------------------------------------------

class ForceCalculator:
    """force_calculator.hpp:36-89: caches eps, eps^2, G; computeForces overwrites acc_*."""

    def __init__(self, ctx: Context | None = None):
        self.softening_eps_ = 0.01
        self.softening_eps2_ = 0.0001
        self.G_ = 1.0
        self._ctx = ctx

    @property
    def ctx(self) -> Context:
        if self._ctx is None:
            self._ctx = default_context()
        return self._ctx

    def setSofteningParameter(self, eps):
        self.softening_eps_ = float(eps)
        # eps*eps in fp32, as the C++ setter does with float members
        self.softening_eps2_ = float(np.float32(eps) * np.float32(eps))

    def setGravitationalConstant(self, G):
        self.G_ = float(G)

    def getSofteningParameter(self):
        return self.softening_eps_

    def getGravitationalConstant(self):
        return self.G_

    def computeForces(self, d_particles: ParticleData):
        raise NotImplementedError

    def _graph_key(self):
        """everything a recorded step bakes in besides the arrays"""
        return (self.G_, self.softening_eps_)

    def getMethod(self) -> ForceMethod:
        raise NotImplementedError


class DirectForceCalculator(ForceCalculator):
    """force_calculator.hpp:101-121 / force_direct.cu:101-106."""

    def __init__(self, block_size: int = 256, ctx: Context | None = None):
        super().__init__(ctx)
        self.block_size_ = block_size

    def computeForces(self, d_particles: ParticleData):
        s = d_particles.struct()
        check(self.ctx._lib.nbody_hip_direct_forces(self.ctx.handle, C.byref(s), self.G_,
                                                    self.softening_eps2_, self.block_size_))

    def getMethod(self):
        return ForceMethod.DIRECT_N2

    def setBlockSize(self, size):
        self.block_size_ = size

    def getBlockSize(self):
        return self.block_size_


class SpatialHashGrid:
    """spatial_hash_grid.hpp:9-59 / force_spatial_hash.cu:155-361."""

    def __init__(self, max_particles: int, cell_size: float = 1.0, ctx: Context | None = None):
        self.ctx = ctx or default_context()
        self.max_particles_ = int(max_particles)
        self.cell_size_ = float(cell_size)
        h = C.c_void_p()
        check(self.ctx._lib.nbody_hip_grid_create(self.ctx.handle, self.max_particles_,
                                                  self.cell_size_, C.byref(h)))
        self._h = h
        _lib.track(self, "grid")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            if self.ctx.handle.value:  # (a closed context has taken its stream with it: nothing left to wait for)
                self.ctx._lib.nbody_hip_grid_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            if not _lib.finalizing():  # (else: closed by the exit hook _lib.close_all, or the runtime is going down)
                self.close()
        except Exception:
            pass

    def setCellSize(self, size):
        check(self.ctx._lib.nbody_hip_grid_set_cell_size(self._h, size))
        self.cell_size_ = float(size)

    def tuning(self, kernel: int = 0):
        """force kernel: 0 automatic, 1 cell-run, 2 / 3 / 4 wave-per-cell with 1 / 2 / 4 bodies per lane, 6 = 3 with the
        window filtered by the box of the cell's bodies (crowded cells), 5 = timing probe (not forces), 7 = two-phase form,
        8 = one lane per body, 9 = split form (8 for the light cells, 3 for the crowded ones), 10 = two bodies of one cell
        per lane"""
        check(self.ctx._lib.nbody_hip_grid_tuning(self._h, kernel))

    def build(self, d_particles: ParticleData):
        s = d_particles.struct()
        self._last_count = d_particles.count
        check(self.ctx._lib.nbody_hip_grid_build(self._h, C.byref(s)))

    def driftBuild(self, d_particles: ParticleData, dt: float):
        """The drift of a Velocity-Verlet step and this build in one pass (nbody_hip_grid_drift_build)."""
        s = d_particles.struct()
        self._last_count = d_particles.count
        check(self.ctx._lib.nbody_hip_grid_drift_build(self._h, C.byref(s), dt))

    def computeForces(self, d_particles: ParticleData, cutoff: float, G: float, eps: float):
        s = d_particles.struct()
        check(self.ctx._lib.nbody_hip_grid_compute_forces(self._h, C.byref(s), cutoff, G, eps))

    def _info(self):
        dims, total = (C.c_int * 3)(), C.c_int()
        lo, hi = (C.c_float * 3)(), (C.c_float * 3)()
        check(self.ctx._lib.nbody_hip_grid_info(self._h, C.byref(dims), C.byref(total),
                                                C.byref(lo), C.byref(hi)))
        return list(dims), total.value, list(lo), list(hi)

    def getGridDims(self):
        return tuple(self._info()[0])

    def getCellSize(self):
        return self.cell_size_

    def getTotalCells(self):
        return self._info()[1]

    def getBoundingBox(self):
        _, _, lo, hi = self._info()
        return lo, hi

    def copyCellDataToHost(self):
        """-> (cell_start, cell_end, particle_cells, sorted_indices) as numpy int32 arrays."""
        dims, total, _, _ = self._info()
        n = self._count()
        cs, ce = np.empty(total, np.int32), np.empty(total, np.int32)
        pc, si = np.empty(n, np.int32), np.empty(n, np.int32)
        check(self.ctx._lib.nbody_hip_grid_copy_cell_data(self._h, cs.ctypes.data, ce.ctypes.data,
                                                          pc.ctypes.data, si.ctypes.data))
        return cs, ce, pc, si

    def _count(self):
        return self._last_count

    @staticmethod
    def getCellIndex(x, y, z, cell_size):
        """force_spatial_hash.cu:14-17: floor of each coordinate over the cell size (fp32)."""
        f = np.float32
        return tuple(int(np.floor(f(v) / f(cell_size))) for v in (x, y, z))

    def verifyCellAssignment(self, h_particles: ParticleData) -> bool:
        """force_spatial_hash.cu:333-362: every body lies inside the bounds of its (clamped) cell."""
        dims, _, lo, _ = self._info()
        _, _, pc, _ = self.copyCellDataToHost()
        cs = np.float32(self.cell_size_)
        cell = np.stack([pc % dims[0], (pc // dims[0]) % dims[1], pc // (dims[0] * dims[1])], 1)
        for a, f in enumerate(("pos_x", "pos_y", "pos_z")):
            p = getattr(h_particles, f)[: pc.size]
            exp = np.clip(np.floor((p - np.float32(lo[a])) / cs).astype(np.int64), 0, dims[a] - 1)
            if not np.array_equal(exp, cell[:, a]):
                return False
            cmin = np.float32(lo[a]) + cell[:, a].astype(np.float32) * cs
            if np.any(p < cmin) or np.any(p > cmin + cs):
                return False
        return True


class SpatialHashCalculator(ForceCalculator):
    """force_calculator.hpp:176-211 / force_spatial_hash.cu:364-377: the grid is created lazily at
    the first computeForces and sized from that particle count."""

    def __init__(self, cell_size: float = 1.0, cutoff_radius: float = 2.0, ctx: Context | None = None):
        super().__init__(ctx)
        self.grid_ = None
        self.cell_size_ = float(cell_size)
        self.cutoff_radius_ = float(cutoff_radius)

    def computeForces(self, d_particles: ParticleData):
        if self.grid_ is None:
            self.grid_ = SpatialHashGrid(d_particles.count, self.cell_size_, self.ctx)
        self.grid_._last_count = d_particles.count
        self.grid_.build(d_particles)
        self.grid_.computeForces(d_particles, self.cutoff_radius_, self.G_, self.softening_eps_)

    def getMethod(self):
        return ForceMethod.SPATIAL_HASH

    def setCellSize(self, size):
        # force_calculator.hpp:199: a plain store.  A grid that already exists was constructed with
        # the earlier size and keeps it (the reference never re-creates or re-sizes its grid,
        # force_spatial_hash.cu:372-374); the new value is used by the next grid this calculator creates.
        self.cell_size_ = float(size)

    def setCutoffRadius(self, radius):
        self.cutoff_radius_ = float(radius)

    def getCellSize(self):
        return self.cell_size_

    def getCutoffRadius(self):
        return self.cutoff_radius_

    def getGrid(self):
        return self.grid_


# include/nbody/barnes_hut_tree.hpp:9-30 -- 76 bytes
OCTREE_NODE_DTYPE = np.dtype([("center", np.float32, 3), ("half_size", np.float32),
                              ("center_of_mass", np.float32, 3), ("total_mass", np.float32),
                              ("children", np.int32, 8), ("particle_index", np.int32),
                              ("is_leaf", np.bool_), ("_pad", np.uint8, 3),
                              ("particle_count", np.int32)])
assert OCTREE_NODE_DTYPE.itemsize == 76


class BarnesHutTree:
    """barnes_hut_tree.hpp:33-81 / force_barnes_hut.cu:204-519."""

    def __init__(self, max_particles: int, ctx: Context | None = None):
        self.ctx = ctx or default_context()
        self.max_particles_ = int(max_particles)
        h = C.c_void_p()
        check(self.ctx._lib.nbody_hip_tree_create(self.ctx.handle, self.max_particles_, C.byref(h)))
        self._h = h
        self._count = 0
        self.h_nodes_ = None
        _lib.track(self, "tree")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            if self.ctx.handle.value:
                self.ctx._lib.nbody_hip_tree_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            if not _lib.finalizing():  # (else: closed by the exit hook _lib.close_all, or the runtime is going down)
                self.close()
        except Exception:
            pass

    def setParams(self, max_depth: int = 20, leaf_max: int = 1):
        check(self.ctx._lib.nbody_hip_tree_set_params(self._h, max_depth, leaf_max))
        self._params = (int(max_depth), int(leaf_max))

    def tuning(self, replicas: int = 0, split_level: int = 0):
        check(self.ctx._lib.nbody_hip_tree_tuning(self._h, replicas, split_level))

    def limitNodes(self, max_nodes: int = 0):
        """cap the node arrays (nbody_hip_tree_limit_nodes): a tree that needs more is cut, forces stay correct"""
        check(self.ctx._lib.nbody_hip_tree_limit_nodes(self._h, max_nodes))

    def walkForm(self, form: int = 0):
        """walk without replicas: 0 automatic, 1 plain, 2 pair walk (cost-ordered), 3 pair walk in plain order"""
        check(self.ctx._lib.nbody_hip_tree_walk_form(self._h, form))

    def countVisits(self, enable: bool = True):
        """stats()["nodes_visited"] is only maintained when enabled (it costs a launch per walk)."""
        check(self.ctx._lib.nbody_hip_tree_count_visits(self._h, 1 if enable else 0))

    def build(self, d_particles: ParticleData):
        s = d_particles.struct()
        check(self.ctx._lib.nbody_hip_tree_build(self._h, C.byref(s)))
        self._count = d_particles.count

    def driftBuild(self, d_particles: ParticleData, dt: float):
        """The drift of a Velocity-Verlet step and this build in one pass (nbody_hip_tree_drift_build)."""
        s = d_particles.struct()
        check(self.ctx._lib.nbody_hip_tree_drift_build(self._h, C.byref(s), dt))
        self._count = d_particles.count

    def computeForces(self, d_particles: ParticleData, theta: float, G: float, eps: float):
        s = d_particles.struct()
        check(self.ctx._lib.nbody_hip_tree_compute_forces(self._h, C.byref(s), theta, G, eps))

    def stats(self):
        nc, rm, nv = C.c_int(), C.c_float(), C.c_ulonglong()
        lb = (C.c_int * 24)()  # NBODY_HIP_TREE_LEVELS
        check(self.ctx._lib.nbody_hip_tree_stats(self._h, C.byref(nc), C.byref(rm), C.byref(nv),
                                                 C.byref(lb)))
        return {"node_count": nc.value, "root_mass": rm.value, "nodes_visited": nv.value,
                "level_base": list(lb)}

    def getNodeCount(self) -> int:
        return self.stats()["node_count"]

    def getMaxDepth(self) -> int:
        """deepest level that holds nodes (ref: BarnesHutTree::getMaxDepth, barnes_hut_tree.hpp:43: the depth the
        insertion reached)"""
        lb = self.stats()["level_base"]
        return max([l - 1 for l in range(1, len(lb)) if lb[l] > lb[l - 1]] + [0])

    def copyNodesToHost(self):
        n = self.getNodeCount()
        nodes = np.zeros(n, dtype=OCTREE_NODE_DTYPE)
        order = np.empty(self._count, np.int32)
        check(self.ctx._lib.nbody_hip_tree_copy_nodes(self._h, nodes.ctypes.data, n, order.ctypes.data))
        self.h_nodes_ = nodes
        self.sorted_indices_ = order
        return nodes

    def getNodes(self):
        return self.h_nodes_

    def verifyTreeStructure(self) -> bool:
        return self.getNodeCount() > 0  # force_barnes_hut.cu:505-509

    def verifyMassConservation(self, h_particles: ParticleData) -> bool:
        """force_barnes_hut.cu:511-519: |sum m - root mass| < 0.001 sum m (fp32 sum)."""
        total = np.float32(0)
        for v in h_particles.mass[: h_particles.count]:
            total = np.float32(total + v)
        root = np.float32(self.stats()["root_mass"])
        return bool(abs(total - root) < np.float32(0.001) * total)


class BarnesHutCalculator(ForceCalculator):
    """force_calculator.hpp:136-165 / force_barnes_hut.cu:521-532: the tree is created lazily at
    the first computeForces and sized from that particle count."""

    def __init__(self, theta: float = 0.5, ctx: Context | None = None):
        super().__init__(ctx)
        self.tree_ = None
        self.theta_ = float(theta)

    def computeForces(self, d_particles: ParticleData):
        if self.tree_ is None:
            self.tree_ = BarnesHutTree(d_particles.count, self.ctx)
        self.tree_.build(d_particles)
        self.tree_.computeForces(d_particles, self.theta_, self.G_, self.softening_eps_)

    def _graph_key(self):
        return (self.G_, self.softening_eps_, self.theta_, id(self.tree_), getattr(self.tree_, "_params", None))

    def getMethod(self):
        return ForceMethod.BARNES_HUT

    def setTheta(self, theta):
        self.theta_ = float(theta)

    def getTheta(self):
        return self.theta_

    def getTree(self):
        return self.tree_


def createForceCalculator(method: ForceMethod, config: SimulationConfig,
                          ctx: Context | None = None) -> ForceCalculator:
    """force_spatial_hash.cu:380-401: unknown enumerators fall back to Direct."""
    if method == ForceMethod.BARNES_HUT:
        calc = BarnesHutCalculator(config.barnes_hut_theta, ctx)
        calc.setGravitationalConstant(config.G)
        calc.setSofteningParameter(config.softening)
        return calc
    if method == ForceMethod.SPATIAL_HASH:
        calc = SpatialHashCalculator(config.spatial_hash_cell_size, config.spatial_hash_cutoff, ctx)
        calc.setGravitationalConstant(config.G)
        calc.setSofteningParameter(config.softening)
        return calc
    calc = DirectForceCalculator(config.cuda_block_size, ctx)
    calc.setGravitationalConstant(config.G)
    calc.setSofteningParameter(config.softening)
    return calc


# ---- integrator ------------------------------------------------------------------------------

class Integrator:
    """integrator.hpp:51-150 / integrator.cu:224-293."""

    def __init__(self, block_size: int = 256, ctx: Context | None = None):
        self.block_size_ = block_size
        self._ctx = ctx

    @property
    def ctx(self) -> Context:
        if self._ctx is None:
            self._ctx = default_context()
        return self._ctx

    @staticmethod
    def _fusable(force_calc) -> bool:
        return (type(force_calc) is DirectForceCalculator
                and isinstance(force_calc.block_size_, int) and 1 <= force_calc.block_size_ <= 1024)

    def integrate(self, d_particles: ParticleData, force_calc: ForceCalculator, dt: float):
        # integrator.cu:224-238.  The extension point is the virtual computeForces
        # (force_calculator.hpp:36-58): the fused launch sequence is taken only for exactly the
        # engine's own DirectForceCalculator, on that calculator's context, and only with a block
        # size its computeForces would accept (otherwise the call below reports it).
        if self._fusable(force_calc):
            s = d_particles.struct()
            fctx = force_calc.ctx
            check(fctx._lib.nbody_hip_integrate_direct(
                fctx.handle, C.byref(s), force_calc.G_, force_calc.softening_eps2_, dt, 1))
            return
        # exactly the engine's own tree / grid calculators (same rule): the drift rides on the packing pass of
        # their build (one pass over the bodies less); same arithmetic, same results
        if type(force_calc) is BarnesHutCalculator and force_calc.tree_ is not None:
            force_calc.tree_.driftBuild(d_particles, dt)
            force_calc.tree_.computeForces(d_particles, force_calc.theta_, force_calc.G_, force_calc.softening_eps_)
            self.updateVelocities(d_particles, dt)
            return
        if type(force_calc) is SpatialHashCalculator and force_calc.grid_ is not None:
            force_calc.grid_.driftBuild(d_particles, dt)
            force_calc.grid_.computeForces(d_particles, force_calc.cutoff_radius_, force_calc.G_,
                                           force_calc.softening_eps_)
            self.updateVelocities(d_particles, dt)
            return
        s = d_particles.struct()  # a_old <- a and the position update in one pass
        check(self.ctx._lib.nbody_hip_drift(self.ctx.handle, C.byref(s), dt))
        force_calc.computeForces(d_particles)
        self.updateVelocities(d_particles, dt)

    def integrate_steps(self, d_particles, force_calc: ForceCalculator, dt: float, steps: int,
                        graph: bool | None = None):
        """`steps` Velocity-Verlet steps queued back to back.  With graph=True the first step runs
        eagerly (it sizes the workspaces), the second is recorded into a hipGraph and the rest replay
        it; the recording is reused while (arrays, count, dt, calculator parameters) stay the same.
        Off by default: measured on MI355X (profiles/r01_step_graph_probe.txt) a replayed step takes
        the same time as the eager one at every size -- the gaps between the small dependent kernels
        are on the GPU side, not in the host's launch calls.  The spatial-hash step reads its grid
        size back every build and always runs eagerly."""
        if steps <= 0:
            return
        if graph is None:
            graph = False
        if graph and (isinstance(force_calc, SpatialHashCalculator) or force_calc.ctx is not self.ctx):
            graph = False  # not capturable / the step's launches would be split over two contexts
        if not graph:
            if self._fusable(force_calc):
                s = d_particles.struct()
                fctx = force_calc.ctx
                check(fctx._lib.nbody_hip_integrate_direct(
                    fctx.handle, C.byref(s), force_calc.G_, force_calc.softening_eps2_, dt, steps))
            else:
                for _ in range(steps):
                    self.integrate(d_particles, force_calc, dt)
            return
        ctx = force_calc.ctx  # the calculator's context is the one the step's launches go to
        key = (d_particles.pos_x.data_ptr(), d_particles.count, float(dt), id(force_calc), force_calc._graph_key(),
               id(ctx))

        def record(steps):
            self._graph, self._graph_for = None, None
            self.integrate(d_particles, force_calc, dt)  # eager: allocations happen here
            steps -= 1
            if steps > 0:
                with ctx.capture() as rec:
                    self.integrate(d_particles, force_calc, dt)
                self._graph, self._graph_for = rec.graph, key
            return steps

        if getattr(self, "_graph_for", None) != key:
            steps = record(steps)
            if steps == 0:
                return
        try:
            self._graph.launch(steps)
        except StateException:
            # stale recording: another system grew a workspace of the shared context, or the tree was
            # re-sized, since it was made (nbody_hip_graph_launch checks the generation) -> record again
            steps = record(steps)
            if steps:
                self._graph.launch(steps)

    def updatePositions(self, d_particles, dt):
        s = d_particles.struct()
        check(self.ctx._lib.nbody_hip_update_positions(self.ctx.handle, C.byref(s), dt))

    def updateVelocities(self, d_particles, dt):
        s = d_particles.struct()
        check(self.ctx._lib.nbody_hip_update_velocities(self.ctx.handle, C.byref(s), dt))

    def storeOldAccelerations(self, d_particles):
        s = d_particles.struct()
        check(self.ctx._lib.nbody_hip_store_accelerations(self.ctx.handle, C.byref(s)))

    def computeKineticEnergy(self, d_particles) -> float:
        s = d_particles.struct()
        out = C.c_float()
        check(self.ctx._lib.nbody_hip_kinetic_energy(self.ctx.handle, C.byref(s), C.byref(out)))
        return out.value

    def computePotentialEnergy(self, d_particles, G, eps) -> float:
        s = d_particles.struct()
        out = C.c_float()
        check(self.ctx._lib.nbody_hip_potential_energy(self.ctx.handle, C.byref(s), G, eps,
                                                       C.byref(out)))
        return out.value

    def computeTotalEnergy(self, d_particles, G, eps) -> float:
        # integrator.cu:291-293: fp32 sum of the two
        return float(np.float32(self.computeKineticEnergy(d_particles)) +
                     np.float32(self.computePotentialEnergy(d_particles, G, eps)))

    def computeEnergiesF64(self, d_particles, G, eps):
        s = d_particles.struct()
        ke, pe = C.c_double(), C.c_double()
        check(self.ctx._lib.nbody_hip_kinetic_energy_f64(self.ctx.handle, C.byref(s), C.byref(ke)))
        check(self.ctx._lib.nbody_hip_potential_energy_f64(self.ctx.handle, C.byref(s), G, eps,
                                                           C.byref(pe)))
        return ke.value, pe.value

    def setBlockSize(self, size):
        self.block_size_ = size

    def getBlockSize(self):
        return self.block_size_


# ---- packed (native) entry points ------------------------------------------------------------

def pack_posm(ctx: Context, x, y, z, m) -> torch.Tensor:
    n = x.numel()
    out = torch.empty((n, 4), dtype=torch.float32, device=x.device)
    check(ctx._lib.nbody_hip_pack_posm(ctx.handle, x.data_ptr(), y.data_ptr(), z.data_ptr(),
                                       m.data_ptr(), n, out.data_ptr()))
    return out


def direct_forces_packed(ctx: Context, targets: torch.Tensor, sources: torch.Tensor, G: float,
                         eps2: float, out: torch.Tensor | None = None, accumulate: bool = False):
    """targets [nt,4], sources [ns,4] float32 contiguous on the device -> acc [nt,4]."""
    assert targets.dtype == torch.float32 and sources.dtype == torch.float32
    assert targets.is_contiguous() and sources.is_contiguous()
    assert targets.dim() == 2 and targets.shape[1] == 4 and sources.dim() == 2 and sources.shape[1] == 4
    nt, ns = targets.shape[0], sources.shape[0]
    if out is None:
        assert not accumulate
        out = torch.empty((nt, 4), dtype=torch.float32, device=targets.device)
    assert out.is_contiguous() and out.shape == (nt, 4)
    check(ctx._lib.nbody_hip_direct_forces_packed(ctx.handle, targets.data_ptr(), nt,
                                                  sources.data_ptr(), ns, out.data_ptr(), G, eps2,
                                                  1 if accumulate else 0))
    return out


def direct_forces_pair_packed(ctx: Context, a: torch.Tensor, b: torch.Tensor, G: float, eps2: float,
                              acc_a: torch.Tensor, acc_b: torch.Tensor, accumulate_a: bool = False,
                              accumulate_b: bool = False):
    """Two disjoint body sets, each a x b pair evaluated once: acc_a (+)= forces on a from b,
    acc_b (+)= forces on b from a (the reactions)."""
    for t in (a, b, acc_a, acc_b):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 2 and t.shape[1] == 4
    assert acc_a.shape == a.shape and acc_b.shape == b.shape
    check(ctx._lib.nbody_hip_direct_forces_pair_packed(
        ctx.handle, a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], acc_a.data_ptr(),
        1 if accumulate_a else 0, acc_b.data_ptr(), 1 if accumulate_b else 0, G, eps2))


def time_direct_packed(ctx: Context, targets, sources, G, eps2, iters: int, out=None) -> float:
    nt, ns = targets.shape[0], sources.shape[0]
    if out is None:
        out = torch.empty((nt, 4), dtype=torch.float32, device=targets.device)
    ms = C.c_float()
    check(ctx._lib.nbody_hip_time_direct_packed(ctx.handle, targets.data_ptr(), nt,
                                                sources.data_ptr(), ns, out.data_ptr(), G, eps2,
                                                iters, C.byref(ms)))
    return ms.value
