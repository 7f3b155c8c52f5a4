"""Headless command line of the reference: option parsing of
src/core/app_cli.cpp:49-177 (flags pinned by tests/test_app_cli.cpp:7-76) and the non-interactive
benchmark loop of src/main.cpp:335-416 (`nbody_sim --benchmark`).  The interactive viewer is out of
scope.  Run:  python -m nbody_amd.cli --particles 65536 --method barnes-hut --benchmark-steps 50
"""
from __future__ import annotations

import re
import sys
import time
from dataclasses import dataclass

import numpy as np

from ._lib import ValidationException
from .api import (ForceMethod, InitDistribution, SimulationConfig, validateParticleCountRange,
                  validateSoftening, validateTheta, validateTimeStep)

_METHODS = {"direct-n2": ForceMethod.DIRECT_N2, "direct_n2": ForceMethod.DIRECT_N2,
            "barnes-hut": ForceMethod.BARNES_HUT, "barnes_hut": ForceMethod.BARNES_HUT,
            "spatial-hash": ForceMethod.SPATIAL_HASH, "spatial_hash": ForceMethod.SPATIAL_HASH}


@dataclass
class AppCliOptions:  # include/nbody/app_cli.hpp:8-26
    particle_count: int = 10000
    force_method: ForceMethod = ForceMethod.DIRECT_N2
    dt: float = 0.001
    G: float = 1.0
    softening: float = 0.1
    barnes_hut_theta: float = 0.5
    spatial_hash_cell_size: float = 1.0
    spatial_hash_cutoff: float = 2.0
    benchmark_mode: bool = False
    benchmark_steps: int = 120
    benchmark_output_path: str = ""
    show_help: bool = False
    export_path: str = ""
    export_format: str = ""
    import_path: str = ""
    list_algorithms: bool = False
    show_diagnostics: bool = False


_STOULL = re.compile(r"[ \t\n\v\f\r]*([+-]?)([0-9]+)")
_STOF = re.compile(r"[ \t\n\v\f\r]*([+-]?(?:0[xX](?:[0-9a-fA-F]+\.?[0-9a-fA-F]*|\.[0-9a-fA-F]+)(?:[pP][+-]?[0-9]+)?"
                   r"|(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?|[iI][nN][fF](?:[iI][nN][iI][tT][yY])?"
                   r"|[nN][aA][nN](?:\([0-9a-zA-Z_]*\))?))")


def _size(value, flag):
    """std::stoull(value) as app_cli.cpp:30-36 uses it: leading blanks and a sign are accepted, the
    longest decimal prefix is converted (trailing text ignored), a minus sign wraps modulo 2^64."""
    m = _STOULL.match(value)
    if not m or int(m.group(2)) > 0xFFFFFFFFFFFFFFFF:
        raise ValidationException(f"Invalid numeric value for {flag}: {value}")
    v = int(m.group(2))
    return (-v) % (1 << 64) if m.group(1) == "-" else v


def _float(value, flag):
    """std::stof(value) (app_cli.cpp:38-44): strtof's longest prefix (decimal, hex, inf, nan), a result
    outside float's range (or a nonzero one that underflows it) is an error."""
    m = _STOF.match(value)
    if not m:
        raise ValidationException(f"Invalid numeric value for {flag}: {value}")
    text = m.group(1)
    low = text.lower().lstrip("+-")
    if low.startswith("0x"):
        v = float.fromhex(text)
    elif low.startswith("nan"):
        v = float("nan")
    else:
        v = float(text)
    if v == v and abs(v) != float("inf"):
        with np.errstate(over="ignore"):
            f32 = float(np.float32(v))
        if abs(f32) == float("inf") or (v != 0.0 and abs(f32) < 1.1754943508222875e-38):
            raise ValidationException(f"Invalid numeric value for {flag}: {value}")
        return f32
    return v


def parseAppCliOptions(argv) -> AppCliOptions:
    """argv[0] is the program name, as in the reference."""
    o = AppCliOptions()
    i = 1

    def need(flag):
        nonlocal i
        if i + 1 >= len(argv):
            raise ValidationException(f"Missing value for {flag}")
        i += 1
        return argv[i]

    while i < len(argv):
        a = argv[i]
        if a in ("--help", "-h"):
            o.show_help = True
        elif a == "--particles":
            o.particle_count = _size(need(a), a)
        elif a == "--method":
            v = need(a)
            if v not in _METHODS:
                raise ValidationException(f"Unsupported force method: {v}")
            o.force_method = _METHODS[v]
        elif a == "--dt":
            o.dt = _float(need(a), a)
        elif a == "--gravity":
            o.G = _float(need(a), a)
        elif a == "--softening":
            o.softening = _float(need(a), a)
        elif a == "--theta":
            o.barnes_hut_theta = _float(need(a), a)
        elif a == "--cell-size":
            o.spatial_hash_cell_size = _float(need(a), a)
        elif a == "--cutoff":
            o.spatial_hash_cutoff = _float(need(a), a)
        elif a == "--benchmark":
            o.benchmark_mode = True
        elif a == "--benchmark-steps":
            o.benchmark_steps = _size(need(a), a)
            o.benchmark_mode = True
        elif a == "--benchmark-output":
            o.benchmark_output_path = need(a)
            o.benchmark_mode = True
        elif a == "--export":
            o.export_path = need(a)
        elif a == "--export-format":
            o.export_format = need(a)
        elif a == "--import":
            o.import_path = need(a)
        elif a == "--list-algorithms":
            o.list_algorithms = True
        elif a == "--diagnostics":
            o.show_diagnostics = True
        elif a.startswith("-") and a:
            raise ValidationException(f"Unknown argument: {a}")
        else:
            o.particle_count = _size(a, "particle count")
        i += 1
    validateParticleCountRange(o.particle_count)
    validateTimeStep(o.dt)
    validateSoftening(o.softening)
    validateTheta(o.barnes_hut_theta)
    if o.G <= 0.0:
        raise ValidationException("Gravitational constant must be positive")
    if o.spatial_hash_cell_size <= 0.0:
        raise ValidationException("Spatial hash cell size must be positive")
    if o.spatial_hash_cutoff <= 0.0:
        raise ValidationException("Spatial hash cutoff must be positive")
    if o.benchmark_steps == 0:
        raise ValidationException("Benchmark steps must be greater than zero")
    return o


def appCliUsage() -> str:
    return ("Usage: python -m nbody_amd.cli [particle_count] [options]\n\n"
            "Simulation options:\n"
            "  --particles N          Set particle count\n"
            "  --method NAME          direct-n2 | barnes-hut | spatial-hash\n"
            "  --dt VALUE             Set integration time step\n"
            "  --gravity VALUE        Set gravitational constant\n"
            "  --softening VALUE      Set softening parameter\n"
            "  --theta VALUE          Set Barnes-Hut theta\n"
            "  --cell-size VALUE      Set spatial hash cell size\n"
            "  --cutoff VALUE         Set spatial hash cutoff radius\n"
            "  --benchmark            Run a non-interactive benchmark and exit\n"
            "  --benchmark-steps N    Set benchmark update steps\n"
            "  --benchmark-output P   Write benchmark JSON to path P\n"
            "\nData export/import:\n"
            "  --export PATH          Export particle state to file (.nbody checkpoint)\n"
            "  --export-format FMT    Export format: checkpoint (default)\n"
            "  --import PATH          Import particle state from a .nbody checkpoint\n"
            "\nDiagnostics:\n"
            "  --list-algorithms      List available force methods and exit\n"
            "  --diagnostics          Output diagnostic information\n"
            "  --help                 Show this message\n")


def runBenchmarkMode(o: AppCliOptions, out=sys.stdout) -> int:
    """src/main.cpp:335-416.  (The reference imports/exports through HDF5 when built with it; here
    both directions use the .nbody checkpoint.)"""
    import torch

    from .observability import (BenchmarkRunRecord, ScopedPhaseProfile, consumeGlobalPhaseSnapshot,
                                globalPhaseProfiler, serializeBenchmarkRunRecords,
                                writeBenchmarkRunRecords)
    from .system import ParticleSystem, Serializer

    cfg = SimulationConfig(particle_count=o.particle_count, init_distribution=InitDistribution.SPHERICAL,
                           force_method=o.force_method, dt=o.dt, G=o.G, softening=o.softening,
                           barnes_hut_theta=o.barnes_hut_theta,
                           spatial_hash_cell_size=o.spatial_hash_cell_size,
                           spatial_hash_cutoff=o.spatial_hash_cutoff)
    ps = ParticleSystem()
    ps.initialize(cfg)
    if o.import_path:
        print(f"Importing state from: {o.import_path}", file=out)
        ps.loadState(o.import_path)
        print(f"Imported {ps.getParticleCount()} particles", file=out)
    consumeGlobalPhaseSnapshot()
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(o.benchmark_steps):
        with ScopedPhaseProfile(globalPhaseProfiler(), "simulation.update"):
            ps.update(ps.getTimeStep())
    torch.cuda.synchronize()  # the reference stops its clock without a device sync
    wall_ms = (time.perf_counter() - start) * 1e3
    if o.export_path:
        if o.export_format not in ("", "checkpoint"):
            print(f"Warning: Unknown export format '{o.export_format}', using checkpoint", file=sys.stderr)
        print(f"Exporting checkpoint to: {o.export_path}", file=out)
        ps.saveState(o.export_path)
        print(f"Exported {ps.getParticleCount()} particles", file=out)
    n = ps.getParticleCount()
    rec = BenchmarkRunRecord(benchmark_name="application.benchmark_mode", force_method=ps.getForceMethod(),
                             particle_count=n, iterations=o.benchmark_steps)
    rec.metrics["wall_time_ms"] = wall_ms / o.benchmark_steps
    rec.metrics["steps_per_s"] = o.benchmark_steps / (wall_ms * 1e-3)
    if ps.getForceMethod() == ForceMethod.DIRECT_N2:
        rec.metrics["pair_interactions_per_s"] = float(n) * n * o.benchmark_steps / (wall_ms * 1e-3)
    rec.parameters.update({"dt": cfg.dt, "gravity": cfg.G, "softening": cfg.softening,
                           "particle_count": float(cfg.particle_count), "gpus": 1.0})
    if cfg.force_method == ForceMethod.BARNES_HUT:
        rec.parameters["theta"] = cfg.barnes_hut_theta
    elif cfg.force_method == ForceMethod.SPATIAL_HASH:
        rec.parameters["cell_size"] = cfg.spatial_hash_cell_size
        rec.parameters["cutoff_radius"] = cfg.spatial_hash_cutoff
    rec.phase_timings = consumeGlobalPhaseSnapshot()
    print(serializeBenchmarkRunRecords([rec]), file=out)
    if o.benchmark_output_path:
        writeBenchmarkRunRecords(o.benchmark_output_path, [rec])
    return 0


def main(argv=None) -> int:
    argv = list(sys.argv if argv is None else argv)
    try:
        o = parseAppCliOptions(argv)
    except ValidationException as e:
        print(f"Error: {e}\n\n{appCliUsage()}", file=sys.stderr)
        return 1
    if o.show_help:
        print(appCliUsage())
        return 0
    if o.list_algorithms:
        print("direct-n2     O(N^2)      exact all-pairs (symmetric kernel from N = 12288)")
        print("barnes-hut    O(N log N)  octree, opening angle --theta")
        print("spatial-hash  O(N)        cell grid, short range (--cell-size, --cutoff)")
        return 0
    if o.show_diagnostics:
        from . import _lib
        lib = _lib.load()
        print(f"libnbody_hip ABI {lib.nbody_hip_abi_version()}, HIP devices: {lib.nbody_hip_device_count()}")
    # there is no viewer: without --benchmark the run is still the headless loop
    return runBenchmarkMode(o)


if __name__ == "__main__":
    sys.exit(main())
