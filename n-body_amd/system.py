"""ParticleSystem facade, SimulationState and the `.nbody` checkpoint (host side, Python).

  ParticleSystem    include/nbody/particle_system.hpp:139-393, src/core/particle_system.cpp:40-318
  SimulationState   include/nbody/simulation_state.hpp, src/utils/simulation_state.cpp:7-39
  Serializer        include/nbody/serialization.hpp:36-137, src/utils/serialization.cpp:9-135
                    (56-byte header {magic 0x4E424F44, version 1, u64 count, time, dt, G, eps,
                    method, reserved[4], pad} + pos_x,y,z, vel_x,y,z, mass as raw fp32)

Orchestration only: every force / integration / energy call goes through the C ABI via api.py.
The C++ counterpart is the reference's own particle_system.cpp linked against libnbody_facade.so
(oracle/Makefile.ref).
"""
from __future__ import annotations

import io
import struct
from dataclasses import dataclass, field

import numpy as np

from ._lib import ValidationException
from .api import (BarnesHutCalculator, DiskDistParams, ForceMethod, InitDistribution, Integrator, ParticleData,
                  ParticleDataManager, ParticleInitializer, SimulationConfig, SpatialHashCalculator,
                  SphericalDistParams, UniformDistParams,
                  createForceCalculator, validateSimulationConfig, validateSoftening,
                  validateTheta, validateTimeStep, _finite)

NBODY_MAGIC = 0x4E424F44
NBODY_VERSION = 1
MAX_PARTICLE_COUNT = 100_000_000
_HEADER = struct.Struct("<IIQffffI4I4x")  # 56 bytes, natural alignment of the C struct
assert _HEADER.size == 56
_ARRAYS = ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "mass")


def _f32(n=0):
    return np.zeros(n, dtype=np.float32)


@dataclass
class SimulationState:
    pos_x: np.ndarray = field(default_factory=_f32)
    pos_y: np.ndarray = field(default_factory=_f32)
    pos_z: np.ndarray = field(default_factory=_f32)
    vel_x: np.ndarray = field(default_factory=_f32)
    vel_y: np.ndarray = field(default_factory=_f32)
    vel_z: np.ndarray = field(default_factory=_f32)
    mass: np.ndarray = field(default_factory=_f32)
    particle_count: int = 0
    simulation_time: float = 0.0
    dt: float = 0.0
    G: float = 0.0
    softening: float = 0.0
    force_method: ForceMethod = ForceMethod.DIRECT_N2

    def __eq__(self, other):  # simulation_state.cpp:7-39: scalars exact, arrays within 1e-6
        if not isinstance(other, SimulationState):
            return NotImplemented
        if (self.particle_count != other.particle_count or self.force_method != other.force_method):
            return False
        for k in ("simulation_time", "dt", "G", "softening"):
            if abs(np.float32(getattr(self, k)) - np.float32(getattr(other, k))) > 1e-6:
                return False
        for k in _ARRAYS:
            a, b = getattr(self, k), getattr(other, k)
            if a.shape != b.shape or np.any(np.abs(a - b) > 1e-6):
                return False
        return True


class Serializer:
    @staticmethod
    def save(target, state: SimulationState):
        if isinstance(target, (str, bytes)):
            try:
                with open(target, "wb") as f:
                    Serializer.save(f, state)
            except OSError as e:
                raise RuntimeError(f"Failed to open file for writing: {target}") from e
            return
        target.write(_HEADER.pack(NBODY_MAGIC, NBODY_VERSION, int(state.particle_count),
                                  float(np.float32(state.simulation_time)), float(np.float32(state.dt)),
                                  float(np.float32(state.G)), float(np.float32(state.softening)),
                                  int(state.force_method), 0, 0, 0, 0))
        for k in _ARRAYS:
            target.write(np.ascontiguousarray(getattr(state, k), dtype="<f4").tobytes())

    @staticmethod
    def readHeader(stream):
        raw = stream.read(_HEADER.size)
        if len(raw) != _HEADER.size:
            raise RuntimeError("Failed to read file header: file may be truncated or corrupted")
        magic, version, count, t, dt, G, eps, method, *_ = _HEADER.unpack(raw)
        if magic != NBODY_MAGIC:
            raise RuntimeError("Invalid file format: wrong magic number")
        if version != NBODY_VERSION:
            raise RuntimeError("Unsupported file version")
        return dict(particle_count=count, simulation_time=t, dt=dt, G=G, softening=eps, force_method=method)

    @staticmethod
    def load(source) -> SimulationState:
        if isinstance(source, (str, bytes)):
            try:
                f = open(source, "rb")
            except OSError as e:
                raise RuntimeError(f"Failed to open file for reading: {source}") from e
            with f:
                return Serializer.load(f)
        h = Serializer.readHeader(source)
        n = h["particle_count"]
        if n > MAX_PARTICLE_COUNT:
            raise ValidationException(f"Particle count ({n}) exceeds maximum allowed ({MAX_PARTICLE_COUNT})")
        st = SimulationState(particle_count=n, simulation_time=h["simulation_time"], dt=h["dt"], G=h["G"],
                             softening=h["softening"], force_method=ForceMethod(h["force_method"]))
        for k in _ARRAYS:
            raw = source.read(4 * n)
            if len(raw) != 4 * n:
                raise RuntimeError("Failed to read particle data: file may be truncated or corrupted")
            setattr(st, k, np.frombuffer(raw, dtype="<f4").astype(np.float32))
        return st

    @staticmethod
    def validateFile(filename) -> bool:
        try:
            with open(filename, "rb") as f:
                Serializer.readHeader(f)
            return True
        except (OSError, RuntimeError):
            return False


class ParticleSystem:
    """The user-facing facade of the reference (particle_system.hpp:139-393)."""

    def __init__(self):
        self.d_particles_ = ParticleData()
        self.h_particles_ = ParticleData()
        self.particle_count_ = 0
        self.force_calculator_ = None
        self.integrator_ = None
        self.dt_, self.G_, self.softening_ = 0.001, 1.0, 0.1
        self.simulation_time_ = 0.0
        self.force_method_ = ForceMethod.DIRECT_N2
        self.is_paused_ = False
        self.is_initialized_ = False
        self.config_ = SimulationConfig()

    # -- memory ------------------------------------------------------------------------------
    def _allocate(self, count):
        self.particle_count_ = count
        ParticleDataManager.allocateDevice(self.d_particles_, count)
        ParticleDataManager.allocateHost(self.h_particles_, count)

    def _free(self):
        if self.is_initialized_:
            ParticleDataManager.freeDevice(self.d_particles_)
            ParticleDataManager.freeHost(self.h_particles_)
            self.is_initialized_ = False

    def _create_calculator(self):
        self.force_calculator_ = createForceCalculator(self.force_method_, self.config_)
        self.force_calculator_.setGravitationalConstant(self.G_)
        self.force_calculator_.setSofteningParameter(self.softening_)

    # -- lifecycle (particle_system.cpp:40-127) ------------------------------------------------
    def initialize(self, config: SimulationConfig, initial_conditions: dict | None = None):
        """`initial_conditions` (an ic.* dict) is an extension: the reference only knows its three
        built-in distributions (box +-10 / sphere R=10 / disk R=10,h=1,omega=0.5, :55-79)."""
        validateSimulationConfig(config)
        self.config_ = SimulationConfig(**vars(config))
        self.dt_, self.G_, self.softening_ = config.dt, config.G, config.softening
        self.force_method_ = config.force_method
        self.is_paused_ = False
        self._free()
        n = config.particle_count
        self._allocate(n)
        if initial_conditions is None:
            # particle_system.cpp:53-79, default seed 42 (particle_data.hpp:38-57): the same bodies as the
            # reference's C++ ParticleSystem under libstdc++ (api.ParticleInitializer)
            if config.init_distribution == InitDistribution.UNIFORM:
                ParticleInitializer.initUniform(self.h_particles_, UniformDistParams((-10, -10, -10), (10, 10, 10)))
            elif config.init_distribution == InitDistribution.SPHERICAL:
                ParticleInitializer.initSpherical(self.h_particles_, SphericalDistParams((0, 0, 0), 10.0))
            else:
                ParticleInitializer.initDisk(self.h_particles_, DiskDistParams((0, 0, 0), 10.0, 1.0, rotation_speed=0.5))
        else:
            for k, v in initial_conditions.items():
                getattr(self.h_particles_, k)[:] = v
        ParticleDataManager.copyToDevice(self.d_particles_, self.h_particles_)
        self._create_calculator()
        self.integrator_ = Integrator(config.cuda_block_size)
        self.force_calculator_.computeForces(self.d_particles_)  # a(0), :88-91
        self.simulation_time_ = 0.0
        self.is_initialized_ = True

    def initializeWithDistribution(self, particle_count, dist):
        self.initialize(SimulationConfig(particle_count=particle_count, init_distribution=dist))

    def update(self, dt: float):
        if not self.is_initialized_ or self.is_paused_:
            return
        self.integrator_.integrate(self.d_particles_, self.force_calculator_, dt)
        self.simulation_time_ = float(np.float32(self.simulation_time_) + np.float32(dt))

    def pause(self):
        self.is_paused_ = True

    def resume(self):
        self.is_paused_ = False

    def isPaused(self):
        return self.is_paused_

    def reset(self):
        if self.is_initialized_:
            self.initialize(self.config_)

    # -- parameters (:137-207) -----------------------------------------------------------------
    def setForceMethod(self, method: ForceMethod):
        if self.force_method_ != method:
            self.force_method_ = method
            self.config_.force_method = method
            self._create_calculator()

    def setGravitationalConstant(self, G):
        if G <= 0 or not _finite(G):
            raise ValidationException("Gravitational constant must be positive and finite")
        self.G_ = self.config_.G = G
        if self.force_calculator_:
            self.force_calculator_.setGravitationalConstant(G)

    def setSofteningParameter(self, eps):
        validateSoftening(eps)
        self.softening_ = self.config_.softening = eps
        if self.force_calculator_:
            self.force_calculator_.setSofteningParameter(eps)

    def setTimeStep(self, dt):
        validateTimeStep(dt)
        self.dt_ = self.config_.dt = dt

    def setBarnesHutTheta(self, theta):
        validateTheta(theta)
        self.config_.barnes_hut_theta = theta
        if isinstance(self.force_calculator_, BarnesHutCalculator):
            self.force_calculator_.setTheta(theta)

    def setSpatialHashCellSize(self, size):
        if size <= 0 or not _finite(size):
            raise ValidationException("Spatial hash cell size must be positive and finite")
        self.config_.spatial_hash_cell_size = size
        if isinstance(self.force_calculator_, SpatialHashCalculator):
            self.force_calculator_.setCellSize(size)

    def setSpatialHashCutoff(self, cutoff):
        if cutoff <= 0 or not _finite(cutoff):
            raise ValidationException("Spatial hash cutoff must be positive and finite")
        self.config_.spatial_hash_cutoff = cutoff
        if isinstance(self.force_calculator_, SpatialHashCalculator):
            self.force_calculator_.setCutoffRadius(cutoff)

    def getForceMethod(self):
        return self.force_method_

    def getGravitationalConstant(self):
        return self.G_

    def getSofteningParameter(self):
        return self.softening_

    def getTimeStep(self):
        return self.dt_

    def getSimulationTime(self):
        return self.simulation_time_

    def getParticleCount(self):
        return self.particle_count_

    def getDeviceData(self):
        return self.d_particles_

    def copyToHost(self, h_particles: ParticleData):
        ParticleDataManager.copyToHost(h_particles, self.d_particles_)

    # -- state (:213-302) ----------------------------------------------------------------------
    def getState(self) -> SimulationState:
        h = ParticleData()
        ParticleDataManager.allocateHost(h, self.particle_count_)
        ParticleDataManager.copyToHost(h, self.d_particles_)
        st = SimulationState(particle_count=self.particle_count_, simulation_time=self.simulation_time_,
                             dt=self.dt_, G=self.G_, softening=self.softening_,
                             force_method=self.force_method_)
        for k in _ARRAYS:
            setattr(st, k, getattr(h, k).copy())
        return st

    def setState(self, state: SimulationState):
        cfg = SimulationConfig(**vars(self.config_))
        cfg.particle_count, cfg.dt, cfg.G = state.particle_count, state.dt, state.G
        cfg.softening, cfg.force_method = state.softening, state.force_method
        validateSimulationConfig(cfg)
        self._free()
        self._allocate(state.particle_count)
        for k in _ARRAYS:
            getattr(self.h_particles_, k)[:] = getattr(state, k)
        # accelerations are not part of the state: zeroed, then recomputed (:274-283)
        ParticleDataManager.copyToDevice(self.d_particles_, self.h_particles_)
        self.simulation_time_ = state.simulation_time
        self.dt_, self.G_, self.softening_ = state.dt, state.G, state.softening
        self.force_method_ = state.force_method
        self.config_ = cfg
        self._create_calculator()
        self.integrator_ = Integrator(cfg.cuda_block_size)
        self.force_calculator_.computeForces(self.d_particles_)
        self.is_paused_ = False
        self.is_initialized_ = True

    def saveState(self, filename):
        Serializer.save(filename, self.getState())

    def loadState(self, filename):
        self.setState(Serializer.load(filename))

    # -- energies (:304-318) -------------------------------------------------------------------
    def computeKineticEnergy(self) -> float:
        return self.integrator_.computeKineticEnergy(self.d_particles_) if self.integrator_ else 0.0

    def computePotentialEnergy(self) -> float:
        if not self.integrator_:
            return 0.0
        return self.integrator_.computePotentialEnergy(self.d_particles_, self.G_, self.softening_)

    def computeTotalEnergy(self) -> float:
        return float(np.float32(self.computeKineticEnergy()) + np.float32(self.computePotentialEnergy()))
