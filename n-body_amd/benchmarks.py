"""The reference's named benchmark runner, on the HIP engine.

    python -m nbody_amd.benchmarks [--benchmark <name|all>] [--particle-count N]
                                   [--iterations N] [--output path] [--list]

Mirrors ref: benchmarks/benchmark_main.cpp -- the five benchmark names and descriptions (:221-236),
the options and their error texts (:241-285), the record fields of every benchmark (:59-218), the
main loop (:289-315) -- and the defaults of ref: scripts/benchmark.sh:12-13 (4,096 particles, 5
iterations; NBODY_BENCHMARK_PARTICLES / NBODY_BENCHMARK_ITERATIONS are honoured like the script
does).  Records go through observability.serializeBenchmarkRunRecords, which is byte-identical to the
reference's serializer (tests/test_cli_observability_cpu.py).

One deliberate difference: the reference stops its host clock around an ASYNCHRONOUS launch
(:124-129, no device sync), which times the launch call; here every iteration
is bracketed by a stream synchronisation, so `wall_time_ms` is the device time of the operation.
The force benchmarks add `pair_interactions_per_s` (direct) and `evaluations_per_s`; phase timings are
collected when NBODY_ENABLE_PROFILING=1 (the reference's compile-time switch, CMakeLists.txt:7).
The "(CUDA)" tag of the listing is the reference's text for "needs the device" and is kept verbatim.
"""
from __future__ import annotations

import io
import os
import re
import sys
import time
from dataclasses import dataclass

import numpy as np

from ._lib import ValidationException
from .api import ForceMethod, InitDistribution, SimulationConfig, createForceCalculator
from .observability import (BenchmarkRunRecord, ScopedPhaseProfile, consumeGlobalPhaseSnapshot,
                            globalPhaseProfiler, serializeBenchmarkRunRecords, writeBenchmarkRunRecords)
from .system import ParticleSystem, Serializer, SimulationState


@dataclass
class BenchmarkOptions:  # benchmark_main.cpp:18-23
    benchmark_name: str = "all"
    particle_count: int = 4096
    iterations: int = 5
    output_path: str = ""


def _profiling() -> bool:
    return os.environ.get("NBODY_ENABLE_PROFILING", "0") not in ("", "0", "OFF", "off", "false")


class _Scope:
    """NBODY_PROFILE_SCOPE: a no-op unless profiling is on."""

    def __init__(self, name, sync=None):
        self.inner = ScopedPhaseProfile(globalPhaseProfiler(), name, sync) if _profiling() else None

    def __enter__(self):
        if self.inner:
            self.inner.__enter__()
        return self

    def __exit__(self, *exc):
        if self.inner:
            self.inner.__exit__(*exc)
        return False


def makeState(particle_count: int) -> SimulationState:
    """benchmark_main.cpp:32-57, the same fp32 products."""
    i = np.arange(particle_count, dtype=np.int64)
    f = np.float32

    def col(mod, mul, scale):
        return (((i * mul) % mod).astype(f) * f(scale)).astype(f)

    return SimulationState(
        pos_x=col(97, 1, 0.01), pos_y=col(89, 3, 0.02), pos_z=col(83, 7, 0.03),
        vel_x=(f(0.001) * (i % 11).astype(f)).astype(f), vel_y=(f(0.002) * (i % 13).astype(f)).astype(f),
        vel_z=(f(0.003) * (i % 17).astype(f)).astype(f), mass=np.ones(particle_count, f),
        particle_count=particle_count, simulation_time=1.0, dt=0.001, G=1.0, softening=0.1,
        force_method=ForceMethod.DIRECT_N2)


def runSerializationBenchmark(o: BenchmarkOptions) -> BenchmarkRunRecord:  # :59-92
    rec = BenchmarkRunRecord("serialization.round_trip", ForceMethod.DIRECT_N2, o.particle_count, o.iterations)
    state = makeState(o.particle_count)
    consumeGlobalPhaseSnapshot()
    total_ms, total_bytes = 0.0, 0
    for _ in range(o.iterations):
        stream = io.BytesIO()
        t0 = time.perf_counter()
        with _Scope("serialization.save"):
            Serializer.save(stream, state)
        total_bytes += stream.tell()
        stream.seek(0)
        with _Scope("serialization.load"):
            loaded = Serializer.load(stream)
        total_ms += (time.perf_counter() - t0) * 1e3
        if loaded.particle_count != state.particle_count:
            raise RuntimeError("Serialization benchmark round-trip lost particle data")
    rec.metrics["wall_time_ms"] = total_ms / o.iterations
    rec.metrics["bytes_per_iteration"] = total_bytes / o.iterations
    rec.parameters["particle_count"] = float(o.particle_count)
    rec.phase_timings = consumeGlobalPhaseSnapshot()
    return rec


def _system(o: BenchmarkOptions, method: ForceMethod):
    cfg = SimulationConfig(particle_count=o.particle_count, force_method=method,
                           init_distribution=InitDistribution.SPHERICAL)
    ps = ParticleSystem()
    ps.initialize(cfg)
    return cfg, ps


def _sync(ps):
    ps.integrator_.ctx.synchronize()


_PHASE = {ForceMethod.DIRECT_N2: "force.direct_n2", ForceMethod.SPATIAL_HASH: "force.spatial_hash"}


def _timed_force_loop(o, ps, calc, method):
    """computeForces `iterations` times, each bracketed by a stream sync (device time)."""
    d = ps.getDeviceData()
    calc.computeForces(d)  # untimed: lazily creates the tree / grid (force_barnes_hut.cu:527-529)
    _sync(ps)
    consumeGlobalPhaseSnapshot()
    total_ms = 0.0
    for _ in range(o.iterations):
        t0 = time.perf_counter()
        if method == ForceMethod.BARNES_HUT and _profiling():
            tree = calc.getTree()  # the reference's phases of this method: barnes_hut.build (:283)
            with _Scope("barnes_hut.build", lambda: _sync(ps)):
                tree.build(d)
            with _Scope("barnes_hut.traverse", lambda: _sync(ps)):
                tree.computeForces(d, calc.theta_, calc.G_, calc.softening_eps_)
        elif method in _PHASE:
            with _Scope(_PHASE[method], lambda: _sync(ps)):
                calc.computeForces(d)
        else:
            calc.computeForces(d)
        _sync(ps)
        total_ms += (time.perf_counter() - t0) * 1e3
    return total_ms / o.iterations


def runForceBenchmark(o: BenchmarkOptions, method: ForceMethod, name: str) -> BenchmarkRunRecord:  # :95-133
    cfg, ps = _system(o, method)
    rec = BenchmarkRunRecord(name, method, o.particle_count, o.iterations)
    rec.parameters["particle_count"] = float(o.particle_count)
    rec.parameters["cuda_block_size"] = float(cfg.cuda_block_size)
    if method == ForceMethod.BARNES_HUT:
        rec.parameters["theta"] = cfg.barnes_hut_theta
    elif method == ForceMethod.SPATIAL_HASH:
        rec.parameters["cell_size"] = cfg.spatial_hash_cell_size
        rec.parameters["cutoff_radius"] = cfg.spatial_hash_cutoff
    calc = createForceCalculator(method, cfg)
    ms = _timed_force_loop(o, ps, calc, method)
    rec.metrics["wall_time_ms"] = ms
    rec.metrics["evaluations_per_s"] = 1e3 / ms
    if method == ForceMethod.DIRECT_N2:
        rec.metrics["pair_interactions_per_s"] = float(o.particle_count) ** 2 * 1e3 / ms
    rec.phase_timings = consumeGlobalPhaseSnapshot()
    return rec


def runDirectBenchmark(o):  # :165-167
    return runForceBenchmark(o, ForceMethod.DIRECT_N2, "force.direct_n2")


def runBarnesHutBenchmark(o: BenchmarkOptions) -> BenchmarkRunRecord:  # :169-212
    cfg, ps = _system(o, ForceMethod.BARNES_HUT)
    rec = BenchmarkRunRecord("force.barnes_hut", ForceMethod.BARNES_HUT, o.particle_count, o.iterations)
    rec.parameters["particle_count"] = float(o.particle_count)
    rec.parameters["theta"] = cfg.barnes_hut_theta
    calc = createForceCalculator(ForceMethod.BARNES_HUT, cfg)
    ms = _timed_force_loop(o, ps, calc, ForceMethod.BARNES_HUT)
    rec.metrics["wall_time_ms"] = ms
    rec.metrics["evaluations_per_s"] = 1e3 / ms
    rec.phase_timings = consumeGlobalPhaseSnapshot()
    for p in rec.phase_timings:  # phase-specific metrics (:203-209)
        if p.samples > 0:
            rec.metrics[p.name + "_ms"] = p.total_duration_ms / p.samples
    return rec


def runSpatialHashBenchmark(o):  # :214-216
    return runForceBenchmark(o, ForceMethod.SPATIAL_HASH, "force.spatial_hash")


def runIntegrationBenchmark(o: BenchmarkOptions) -> BenchmarkRunRecord:  # :135-163
    cfg, ps = _system(o, ForceMethod.DIRECT_N2)
    rec = BenchmarkRunRecord("integration.velocity_verlet", cfg.force_method, o.particle_count, o.iterations)
    rec.parameters["particle_count"] = float(o.particle_count)
    rec.parameters["dt"] = cfg.dt
    rec.parameters["cuda_block_size"] = float(cfg.cuda_block_size)
    ps.update(ps.getTimeStep())  # untimed: sizes the workspaces
    _sync(ps)
    consumeGlobalPhaseSnapshot()
    total_ms = 0.0
    for _ in range(o.iterations):
        t0 = time.perf_counter()
        with _Scope("simulation.update", lambda: _sync(ps)):
            ps.update(ps.getTimeStep())
        _sync(ps)
        total_ms += (time.perf_counter() - t0) * 1e3
    ms = total_ms / o.iterations
    rec.metrics["wall_time_ms"] = ms
    rec.metrics["steps_per_s"] = 1e3 / ms
    rec.metrics["pair_interactions_per_s"] = float(o.particle_count) ** 2 * 1e3 / ms
    rec.phase_timings = consumeGlobalPhaseSnapshot()
    return rec


# name, description, needs the device, runner (benchmark_main.cpp:218-238, same order)
BENCHMARKS = [
    ("serialization.round_trip", "Binary checkpoint serialization and load round-trip", False, runSerializationBenchmark),
    ("force.direct_n2", "Direct N^2 force calculation", True, runDirectBenchmark),
    ("force.barnes_hut", "Barnes-Hut force calculation", True, runBarnesHutBenchmark),
    ("force.spatial_hash", "Spatial hash force calculation", True, runSpatialHashBenchmark),
    ("integration.velocity_verlet", "Velocity Verlet integration step", True, runIntegrationBenchmark),
]


def usage() -> str:  # printUsage, :241-252
    lines = ["Usage: nbody_benchmarks [--benchmark <name|all>] [--particle-count N]",
             "                        [--iterations N] [--output path] [--list]", "",
             "Available benchmarks:"]
    for name, desc, dev, _ in BENCHMARKS:
        lines.append(f"  - {name}{' (CUDA)' if dev else ''}: {desc}")
    return "\n".join(lines) + "\n"


class _StdError(Exception):
    """what() of a std:: exception thrown by std::stoull."""


_STOULL = re.compile(r"[ \t\n\v\f\r]*([+-]?)([0-9]+)")


def _stoull(text: str) -> int:
    m = _STOULL.match(text)
    if not m or int(m.group(2)) > 0xFFFFFFFFFFFFFFFF:
        raise _StdError("stoull")  # std::invalid_argument / std::out_of_range: what() == "stoull"
    v = int(m.group(2))
    return (-v) % (1 << 64) if m.group(1) == "-" else v


class _Exit(Exception):
    pass


def parseOptions(argv, out) -> BenchmarkOptions:  # :254-285 (argv[0] is the program name)
    o = BenchmarkOptions(particle_count=_stoull(os.environ.get("NBODY_BENCHMARK_PARTICLES", "4096")),
                         iterations=_stoull(os.environ.get("NBODY_BENCHMARK_ITERATIONS", "5")))
    i = 1
    while i < len(argv):
        a = argv[i]
        more = i + 1 < len(argv)
        if a == "--benchmark" and more:
            i += 1
            o.benchmark_name = argv[i]
        elif a == "--particle-count" and more:
            i += 1
            o.particle_count = _stoull(argv[i])
        elif a == "--iterations" and more:
            i += 1
            o.iterations = _stoull(argv[i])
        elif a == "--output" and more:
            i += 1
            o.output_path = argv[i]
        elif a in ("--list", "--help", "-h"):
            out.write(usage())
            raise _Exit()
        else:
            raise ValidationException("Unknown benchmark argument: " + a)
        i += 1
    if o.iterations == 0:
        raise ValidationException("Benchmark iterations must be greater than zero")
    return o


def _what(e: Exception) -> str:
    if isinstance(e, ValidationException):
        s = str(e)
        return s if s.startswith("Validation Error: ") else "Validation Error: " + s
    return str(e)


def main(argv=None, out=None, err=None) -> int:  # :289-315
    argv = list(sys.argv if argv is None else argv)
    out = out or sys.stdout
    err = err or sys.stderr
    try:
        try:
            o = parseOptions(argv, out)
        except _Exit:
            return 0
        results = [run(o) for name, _, _, run in BENCHMARKS if o.benchmark_name in ("all", name)]
        if not results:
            raise ValidationException("Requested benchmark was not found")
        out.write(serializeBenchmarkRunRecords(results) + "\n")
        if o.output_path:
            writeBenchmarkRunRecords(o.output_path, results)
    except Exception as e:  # noqa: BLE001 -- the reference catches std::exception here
        err.write(f"Benchmark error: {_what(e)}\n")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
