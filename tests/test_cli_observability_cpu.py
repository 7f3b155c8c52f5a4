"""f2 / f3 rows, CPU-only: option parsing (tests/test_app_cli.cpp:7-76), usage text, benchmark
record JSON (tests/test_performance_observability.cpp:7-24) and the phase profiler (:26-41)."""
import json
import time

import pytest


def test_parses_structured_simulation_options(nb):
    o = nb.cli.parseAppCliOptions(["nbody_sim", "--particles", "2048", "--method", "barnes-hut", "--dt", "0.002",
                                   "--theta", "0.7", "--softening", "0.05", "--benchmark", "--benchmark-steps", "12"])
    assert o.particle_count == 2048 and o.force_method == nb.ForceMethod.BARNES_HUT
    assert (o.dt, o.barnes_hut_theta, o.softening) == (0.002, 0.7, 0.05)
    assert o.benchmark_mode and o.benchmark_steps == 12


def test_cli_reference_cases(nb):
    P = nb.cli.parseAppCliOptions
    with pytest.raises(nb.ValidationException, match="Unsupported force method: mystery"):
        P(["nbody_sim", "--method", "mystery"])
    o = P(["nbody_sim", "--export", "output.nbody", "--export-format", "checkpoint", "--import", "input.nbody"])
    assert (o.export_path, o.export_format, o.import_path) == ("output.nbody", "checkpoint", "input.nbody")
    assert P(["nbody_sim", "--list-algorithms"]).list_algorithms
    assert P(["nbody_sim", "--diagnostics"]).show_diagnostics
    o = P(["nbody_sim", "--particles", "5000", "--method", "barnes-hut", "--export", "state.nbody", "--diagnostics"])
    assert (o.particle_count, o.force_method, o.export_path, o.show_diagnostics) == \
        (5000, nb.ForceMethod.BARNES_HUT, "state.nbody", True)
    # defaults (app_cli.hpp:8-26) and the positional particle count
    d = P(["nbody_sim"])
    assert (d.particle_count, d.dt, d.G, d.softening, d.benchmark_steps) == (10000, 0.001, 1.0, 0.1, 120)
    assert P(["nbody_sim", "4096"]).particle_count == 4096
    assert P(["nbody_sim", "--benchmark-output", "x.json"]).benchmark_mode
    for bad, msg in ((["--particles"], "Missing value"), (["--dt", "abc"], "Invalid numeric"),
                     (["--bogus"], "Unknown argument"), (["--particles", "0"], "greater than 0"),
                     (["--dt", "2"], "too large"), (["--gravity", "0"], "Gravitational"),
                     (["--cell-size", "-1"], "cell size"), (["--cutoff", "0"], "cutoff"),
                     (["--benchmark-steps", "0"], "Benchmark steps"), (["--theta", "3"], "theta")):
        with pytest.raises(nb.ValidationException, match=msg):
            P(["nbody_sim"] + bad)
    u = nb.cli.appCliUsage()
    for flag in ("--particles", "--method", "--dt", "--gravity", "--softening", "--theta", "--cell-size",
                 "--cutoff", "--benchmark-steps", "--benchmark-output", "--export", "--import"):
        assert flag in u


# tests/test_performance_observability.cpp:7-24
def test_benchmark_report_json_shape(nb):
    ob = nb.observability
    r = ob.BenchmarkRunRecord(benchmark_name="force.direct_n2", force_method=nb.ForceMethod.DIRECT_N2,
                              particle_count=4096, iterations=12)
    r.metrics["wall_time_ms"] = 1.25
    r.metrics["throughput_particles_per_second"] = 32768.0
    r.parameters["cuda_block_size"] = 256.0
    js = ob.serializeBenchmarkRunRecord(r)
    for needle in ('"benchmark_name":"force.direct_n2"', '"particle_count":4096', '"force_method":"direct_n2"',
                   '"wall_time_ms":1.25', '"cuda_block_size":256'):
        assert needle in js
    doc = json.loads(ob.serializeBenchmarkRunRecords([r, r]))
    assert len(doc["benchmarks"]) == 2 and doc["benchmarks"][0]["metrics"]["throughput_particles_per_second"] == 32768
    assert list(doc["benchmarks"][0]) == ["benchmark_name", "force_method", "particle_count", "iterations",
                                          "metrics", "parameters", "phase_timings"]
    assert ob.forceMethodToString(nb.ForceMethod.SPATIAL_HASH) == "spatial_hash"


# tests/test_performance_observability.cpp:26-41
def test_scoped_phase_profiler_accumulates(nb, tmp_path):
    ob = nb.observability
    prof = ob.PhaseProfiler()
    with ob.ScopedPhaseProfile(prof, "simulation.update"):
        time.sleep(0.002)
    snap = prof.snapshot()
    assert len(snap) == 1 and snap[0].name == "simulation.update"
    assert snap[0].total_duration_ms >= 0.5 and snap[0].samples == 1
    with ob.ScopedPhaseProfile(prof, "simulation.update"):
        pass
    with ob.ScopedPhaseProfile(prof, "force.direct_n2"):
        pass
    snap = prof.snapshot()
    assert [p.name for p in snap] == ["simulation.update", "force.direct_n2"] and snap[0].samples == 2
    prof.reset()
    assert prof.snapshot() == []
    g = ob.globalPhaseProfiler()
    g.record("x", 1.5)
    assert ob.consumeGlobalPhaseSnapshot()[0].total_duration_ms == 1.5 and g.snapshot() == []
    r = ob.BenchmarkRunRecord(benchmark_name="b", phase_timings=[ob.PhaseTiming("p", 2.5, 3)])
    path = tmp_path / "out.json"
    ob.writeBenchmarkRunRecords(str(path), [r])
    assert json.loads(path.read_text())["benchmarks"][0]["phase_timings"] == [
        {"name": "p", "total_duration_ms": 2.5, "samples": 3}]
    with pytest.raises(RuntimeError, match="Failed to open benchmark output file"):
        ob.writeBenchmarkRunRecords(str(tmp_path / "nodir" / "x.json"), [r])
