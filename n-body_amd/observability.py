"""Phase profiler and structured benchmark records.

  PhaseProfiler / ScopedPhaseProfile / PhaseTiming   performance_observability.hpp:16-49, .cpp:54-87
  BenchmarkRunRecord + JSON                          performance_observability.hpp:51-63, .cpp:89-153
  (JSON shape pinned by tests/test_performance_observability.cpp:7-24)

The reference times phases with a host wall clock around ASYNCHRONOUS launches, i.e. it measures
launch cost in release builds.  Here a scope can synchronise the device
stream before it stops its clock (`sync=True`), which is what the headless benchmark mode uses; the
phase names are the reference's (`simulation.update`, `force.direct_n2`, ...).
"""
from __future__ import annotations

import json
import threading
import time
from dataclasses import dataclass, field

from .api import ForceMethod


@dataclass
class PhaseTiming:
    name: str
    total_duration_ms: float = 0.0
    samples: int = 0


class PhaseProfiler:
    def __init__(self):
        self._mutex = threading.Lock()
        self._phases: list[PhaseTiming] = []

    def record(self, name: str, duration_ms: float):
        with self._mutex:
            for p in self._phases:
                if p.name == name:
                    p.total_duration_ms += duration_ms
                    p.samples += 1
                    return
            self._phases.append(PhaseTiming(name, duration_ms, 1))

    def snapshot(self):
        with self._mutex:
            return [PhaseTiming(p.name, p.total_duration_ms, p.samples) for p in self._phases]

    def reset(self):
        with self._mutex:
            self._phases.clear()


class ScopedPhaseProfile:
    """`with ScopedPhaseProfile(profiler, "force.direct_n2", sync=ctx.synchronize): ...`"""

    def __init__(self, profiler: PhaseProfiler, name: str, sync=None):
        self.profiler, self.name, self.sync = profiler, name, sync

    def __enter__(self):
        self._t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if self.sync is not None:
            self.sync()
        self.profiler.record(self.name, (time.perf_counter() - self._t0) * 1e3)
        return False


_global = PhaseProfiler()


def globalPhaseProfiler() -> PhaseProfiler:
    return _global


def consumeGlobalPhaseSnapshot():
    snap = _global.snapshot()
    _global.reset()
    return snap


def forceMethodToString(method) -> str:
    return {ForceMethod.DIRECT_N2: "direct_n2", ForceMethod.BARNES_HUT: "barnes_hut",
            ForceMethod.SPATIAL_HASH: "spatial_hash"}.get(method, "unknown")


@dataclass
class BenchmarkRunRecord:
    benchmark_name: str = ""
    force_method: ForceMethod = ForceMethod.DIRECT_N2
    particle_count: int = 0
    iterations: int = 0
    metrics: dict = field(default_factory=dict)
    parameters: dict = field(default_factory=dict)
    phase_timings: list = field(default_factory=list)


def _escape(text: str) -> str:
    """escapeJson of the reference (performance_observability.cpp:10-35): only backslash, quote, newline,
    carriage return and tab are escaped; every other character goes out as it is."""
    return (text.replace("\\", "\\\\").replace("\"", "\\\"").replace("\n", "\\n").replace("\r", "\\r")
            .replace("\t", "\\t"))


def _q(text: str) -> str:
    return "\"" + _escape(text) + "\""


def _num(v) -> str:
    # the reference streams doubles with operator<< (6 significant digits, no trailing zeros):
    # 256.0 -> "256", 1.25 -> "1.25"
    return format(float(v), ".6g")


def _number_map(m: dict) -> str:
    # std::map iterates in key order
    return "{" + ",".join(f"{_q(k)}:{_num(m[k])}" for k in sorted(m)) + "}"


def serializeBenchmarkRunRecord(r: BenchmarkRunRecord) -> str:
    phases = ",".join(
        "{" + f"\"name\":{_q(p.name)},\"total_duration_ms\":{_num(p.total_duration_ms)},"
        f"\"samples\":{p.samples}" + "}" for p in r.phase_timings)
    return ("{" + f"\"benchmark_name\":{_q(r.benchmark_name)},"
            f"\"force_method\":\"{forceMethodToString(r.force_method)}\","
            f"\"particle_count\":{int(r.particle_count)},\"iterations\":{int(r.iterations)},"
            f"\"metrics\":{_number_map(r.metrics)},\"parameters\":{_number_map(r.parameters)},"
            f"\"phase_timings\":[{phases}]" + "}")


def serializeBenchmarkRunRecords(records) -> str:
    return "{\"benchmarks\":[" + ",".join(serializeBenchmarkRunRecord(r) for r in records) + "]}"


def writeBenchmarkRunRecords(path: str, records):
    try:
        with open(path, "w") as f:
            f.write(serializeBenchmarkRunRecords(records) + "\n")
    except OSError as e:
        raise RuntimeError(f"Failed to open benchmark output file: {path}") from e
