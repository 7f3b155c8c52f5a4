"""f2 / f3 rows, CPU-only: option parsing (tests/test_app_cli.cpp:7-76), usage text, benchmark
record JSON (tests/test_performance_observability.cpp:7-24) and the phase profiler (:26-41)."""
import json
import os
import re
import subprocess
import time

import numpy as np

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parses_structured_simulation_options(nb):
    o = nb.cli.parseAppCliOptions(["nbody_sim", "--particles", "2048", "--method", "barnes-hut", "--dt", "0.002",
                                   "--theta", "0.7", "--softening", "0.05", "--benchmark", "--benchmark-steps", "12"])
    assert o.particle_count == 2048 and o.force_method == nb.ForceMethod.BARNES_HUT
    f32 = lambda v: float(np.float32(v))  # noqa: E731  (options are C floats in the reference)
    assert (o.dt, o.barnes_hut_theta, o.softening) == (f32(0.002), f32(0.7), f32(0.05))
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


# f3 pinned to the reference's own parser: oracle/_ref/ref_cli_driver runs parseAppCliOptions compiled
# from /root/reference/src/core/app_cli.cpp (oracle/Makefile.ref; host only) on the same arguments
REF_CLI = os.path.join(ROOT, "oracle", "_ref", "ref_cli_driver")
ARGV_CASES = [
    [], ["4096"], ["--particles", "500", "--method", "barnes-hut", "--theta", "0.7"],
    ["--method", "spatial_hash", "--cell-size", "0.5", "--cutoff", "1.5", "--softening", "0"],
    ["--method", "direct_n2", "--dt", "1e-3", "--gravity", "6.674e-11"],
    ["--benchmark-steps", "77"], ["--benchmark-output", "out.json"], ["--benchmark"],
    ["--export", "a.nbody", "--export-format", "nbody", "--import", "b.nbody"],
    ["--list-algorithms"], ["--diagnostics"], ["-h"], ["--help", "123"],
    ["12abc"], ["+7"], [" 42"], ["1e3"], ["0x10"], ["-5"], ["abc"], [""],
    ["--dt", ".5"], ["--dt", "5."], ["--dt", "1.5x"], ["--dt", "inf"], ["--dt", "nan"], ["--dt", "-0.1"],
    ["--dt", "1e40"], ["--dt", "1e-50"], ["--dt", "0x1p-4"], ["--dt", "abc"], ["--dt"],
    ["--theta", "2.5"], ["--theta", "-1"], ["--softening", "-0.5"], ["--gravity", "0"], ["--gravity", "-2"],
    ["--cell-size", "0"], ["--cutoff", "-1"], ["--benchmark-steps", "0"], ["--particles", "0"],
    ["--particles", "100000001"], ["--method", "fmm"], ["--bogus"], ["--particles"],
    ["100", "200"], ["--particles", "10", "20"],
]


@pytest.mark.skipif(not os.path.exists(REF_CLI), reason="oracle/_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("args", ARGV_CASES, ids=lambda a: " ".join(a) or "(none)")
def test_cli_parser_equals_reference_parser(nb, args):
    from nbody_amd import cli
    r = subprocess.run([REF_CLI] + args, capture_output=True, text=True, timeout=30)
    try:
        o = cli.parseAppCliOptions(["prog"] + args)
        mine = None
    except nb.ValidationException as e:
        o, mine = None, str(e)
    if r.returncode != 0:
        ref_msg = r.stdout.rstrip("\n").replace("error: ", "", 1).replace("Validation Error: ", "", 1)
        assert mine == ref_msg, (mine, r.stdout)
        return
    assert mine is None, (mine, r.stdout)
    ref = dict(line.split(" ", 1) if " " in line else (line, "") for line in r.stdout.splitlines())
    f32 = lambda v: float(np.float32(v))  # noqa: E731
    assert int(ref["particle_count"]) == o.particle_count
    assert int(ref["force_method"]) == int(o.force_method)
    for key, val in (("dt", o.dt), ("G", o.G), ("softening", o.softening), ("theta", o.barnes_hut_theta),
                     ("cell_size", o.spatial_hash_cell_size), ("cutoff", o.spatial_hash_cutoff)):
        assert f32(float(ref[key])) == f32(val), key
    assert int(ref["benchmark_mode"]) == int(o.benchmark_mode) and int(ref["benchmark_steps"]) == o.benchmark_steps
    assert ref["benchmark_output"] == o.benchmark_output_path
    assert ref["export_path"] == o.export_path and ref["export_format"] == o.export_format
    assert ref["import_path"] == o.import_path
    assert int(ref["show_help"]) == int(o.show_help) and int(ref["list_algorithms"]) == int(o.list_algorithms)
    assert int(ref["show_diagnostics"]) == int(o.show_diagnostics)


# f2 pinned to the reference's own serializer: oracle/_ref/ref_observability_driver prints what
# serializeBenchmarkRunRecords / PhaseProfiler (compiled from the reference's performance_observability.cpp)
# produce for a fixed set of awkward records; the Python module must produce the same bytes
REF_OBS = os.path.join(ROOT, "oracle", "_ref", "ref_observability_driver")


@pytest.mark.skipif(not os.path.exists(REF_OBS), reason="oracle/_ref not built (needs /root/reference at build time)")
def test_benchmark_records_equal_reference_serializer(nb):
    from nbody_amd import observability as ob
    out = subprocess.run([REF_OBS], capture_output=True, timeout=30).stdout.decode("utf-8").split("\n")
    a = ob.BenchmarkRunRecord(benchmark_name="force.direct_n2", force_method=nb.ForceMethod.DIRECT_N2,
                              particle_count=1048576, iterations=20)
    a.metrics = {"wall_time_ms": 156.793342, "steps_per_s": 6.3778218, "pair_interactions_per_s": 7.0124892e12,
                 "zero": 0.0, "neg_zero": -0.0, "tiny": 1.25e-7, "exact": 256.0, "third": 1.0 / 3.0}
    a.parameters = {"dt": 0.001, "softening": 0.1, "gpus": 1.0, "big": 1e21, "negative": -42.5}
    prof = ob.PhaseProfiler()
    prof.record("simulation.update", 1.5)
    prof.record("force", 0.25)
    prof.record("simulation.update", 2.75)
    a.phase_timings = prof.snapshot()
    b = ob.BenchmarkRunRecord(benchmark_name="quote\" backslash\\ newline\n tab\t cr\r unicode \u03b1 end", force_method=7)
    b.metrics = {"inf": float("inf"), "nan": float("nan"), "key \"q\"": 1.0}
    c = ob.BenchmarkRunRecord(benchmark_name="empty", force_method=nb.ForceMethod.SPATIAL_HASH)
    assert ob.serializeBenchmarkRunRecords([a, b, c]) == out[0]
    assert ob.serializeBenchmarkRunRecord(a) == out[1]


# f2: the named benchmark runner pinned to the reference's OWN runner (oracle/_ref/nbody_benchmarks =
# benchmarks/benchmark_main.cpp compiled as the reference's headless CI does, NBODY_WITH_CUDA=0, so
# only serialization.round_trip is compiled in): option handling, error texts, the listing and the
# serialization record.  The four device benchmarks run on the GPU (tests/test_particle_system_gpu.py).
REF_BENCH = os.path.join(ROOT, "oracle", "_ref", "nbody_benchmarks")
BENCH_ARGV = [
    ["--list"], ["--help"], ["-h"], ["--bogus"], ["--particle-count", "abc"], ["--particle-count"],
    ["--iterations", "0"], ["--iterations"], ["--benchmark", "nope"], ["--benchmark"],
    ["--particle-count", "99999999999999999999999"], ["--iterations", "x3"], ["extra"],
    ["--benchmark", "serialization.round_trip", "--particle-count", "257", "--iterations", "3"],
    ["--benchmark", "serialization.round_trip", "--particle-count", " 12abc", "--iterations", "+2"],
    ["--benchmark", "serialization.round_trip", "--particle-count", "0", "--iterations", "1"],
]


def _run_mine(args, env=None):
    import io
    from nbody_amd import benchmarks
    out, err = io.StringIO(), io.StringIO()
    rc = benchmarks.main(["nbody_benchmarks"] + args, out=out, err=err)
    return rc, out.getvalue(), err.getvalue()


@pytest.mark.skipif(not os.path.exists(REF_BENCH), reason="oracle/_ref not built (needs /root/reference at build time)")
@pytest.mark.parametrize("args", BENCH_ARGV, ids=lambda a: " ".join(a))
def test_benchmark_runner_equals_reference_runner(nb, args, monkeypatch):
    import json
    monkeypatch.delenv("NBODY_BENCHMARK_PARTICLES", raising=False)
    monkeypatch.delenv("NBODY_BENCHMARK_ITERATIONS", raising=False)
    monkeypatch.delenv("NBODY_ENABLE_PROFILING", raising=False)
    r = subprocess.run([REF_BENCH] + args, capture_output=True, text=True, timeout=60)
    rc, out, err = _run_mine(args)
    assert rc == r.returncode
    assert err == r.stderr
    if args[0] in ("--list", "--help", "-h"):
        # the reference built without its GPU code lists one benchmark; its listing code prints the
        # other four as "  - name (CUDA): description" when they are compiled in (benchmark_main.cpp:226-236,246-250)
        mine, ref = out.splitlines(), r.stdout.splitlines()
        assert mine[:5] == ref[:5]
        assert mine[5:] == ["  - force.direct_n2 (CUDA): Direct N^2 force calculation",
                            "  - force.barnes_hut (CUDA): Barnes-Hut force calculation",
                            "  - force.spatial_hash (CUDA): Spatial hash force calculation",
                            "  - integration.velocity_verlet (CUDA): Velocity Verlet integration step"]
        return
    if rc != 0:
        assert out == r.stdout == ""
        return
    a, b = json.loads(out), json.loads(r.stdout)
    assert list(a) == list(b) == ["benchmarks"] and len(a["benchmarks"]) == len(b["benchmarks"]) == 1
    ra, rb = a["benchmarks"][0], b["benchmarks"][0]
    assert list(ra) == list(rb)  # same fields, same order
    for k in ("benchmark_name", "force_method", "particle_count", "iterations", "parameters", "phase_timings"):
        assert ra[k] == rb[k], k
    assert list(ra["metrics"]) == list(rb["metrics"]) == ["bytes_per_iteration", "wall_time_ms"]
    assert ra["metrics"]["bytes_per_iteration"] == rb["metrics"]["bytes_per_iteration"]
    # byte-identical once the (machine-dependent) time is masked
    mask = lambda s: re.sub(r'"wall_time_ms":[^,}]+', '"wall_time_ms":T', s)  # noqa: E731
    assert mask(out) == mask(r.stdout)


def test_benchmark_runner_state_and_output_file(nb, tmp_path, monkeypatch):
    import json
    from nbody_amd import benchmarks
    st = benchmarks.makeState(200)  # benchmark_main.cpp:46-53, fp32 products
    assert st.pos_x[98] == np.float32(1) * np.float32(0.01) and st.pos_y[30] == np.float32(1) * np.float32(0.02)
    assert st.pos_z[12] == np.float32(84 % 83) * np.float32(0.03) and st.vel_z[16] == np.float32(0.003) * np.float32(16)
    assert st.mass.dtype == np.float32 and np.all(st.mass == 1)
    monkeypatch.setenv("NBODY_BENCHMARK_PARTICLES", "64")   # scripts/benchmark.sh:12-13
    monkeypatch.setenv("NBODY_BENCHMARK_ITERATIONS", "2")
    monkeypatch.setenv("NBODY_ENABLE_PROFILING", "1")
    path = tmp_path / "sub.json"
    rc, out, err = _run_mine(["--benchmark", "serialization.round_trip", "--output", str(path)])
    assert rc == 0 and err == ""
    rec = json.loads(out)["benchmarks"][0]
    assert rec["particle_count"] == 64 and rec["iterations"] == 2
    assert rec["metrics"]["bytes_per_iteration"] == 56 + 7 * 4 * 64
    assert [p["name"] for p in rec["phase_timings"]] == ["serialization.save", "serialization.load"]
    assert all(p["samples"] == 2 for p in rec["phase_timings"])
    assert json.loads(path.read_text()) == json.loads(out)
