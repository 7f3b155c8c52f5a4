"""ParticleSystem facade (a13) on the GPU: the reference's behaviour
(src/core/particle_system.cpp:40-318) and its tests (tests/test_serialization.cpp:171-216)."""
import numpy as np
import pytest

from gpu_util import rel_err

pytestmark = pytest.mark.gpu


# tests/test_serialization.cpp:171-216 PauseResumePreservesState
def test_pause_resume_preserves_state(nb, ctx):
    cfg = nb.SimulationConfig(particle_count=50, init_distribution=nb.InitDistribution.SPHERICAL,
                              force_method=nb.ForceMethod.DIRECT_N2, dt=0.001)
    s = nb.ParticleSystem()
    s.initialize(cfg)
    for _ in range(10):
        s.update(cfg.dt)
    before = s.getState()
    s.pause()
    assert s.isPaused()
    for _ in range(10):
        s.update(cfg.dt)
    after = s.getState()
    assert before == after
    s.resume()
    assert not s.isPaused()
    # the reference asserts inequality after ONE step of dt = 1e-3 with SimulationState's 1e-6
    # tolerance; bodies at r ~ 10 move ~ a dt^2 ~ 1e-8 then, so several steps are taken here
    for _ in range(200):
        s.update(0.01)
    assert not (s.getState() == after)
    assert s.getSimulationTime() == pytest.approx(0.01 + 200 * 0.01, rel=1e-4)


@pytest.mark.parametrize("method", ["DIRECT_N2", "BARNES_HUT", "SPATIAL_HASH"])
def test_save_load_continues_identically(nb, ctx, tmp_path, method):
    m = nb.ForceMethod[method]
    cfg = nb.SimulationConfig(particle_count=2000, init_distribution=nb.InitDistribution.DISK,
                              force_method=m, dt=0.002, softening=0.05, G=1.0,
                              spatial_hash_cell_size=2.0, spatial_hash_cutoff=2.0)
    a = nb.ParticleSystem()
    a.initialize(cfg)
    for _ in range(5):
        a.update(cfg.dt)
    path = str(tmp_path / "ck.nbody")
    a.saveState(path)
    b = nb.ParticleSystem()
    # the checkpoint carries count, time, dt, G, eps and the method (serialization.hpp:36-65); theta,
    # cell size and cutoff stay those of the LOADING system (particle_system.cpp:261-266)
    b.initialize(nb.SimulationConfig(particle_count=7, spatial_hash_cell_size=2.0, spatial_hash_cutoff=2.0))
    b.loadState(path)
    assert b.getParticleCount() == 2000 and b.getForceMethod() == m
    assert b.getSimulationTime() == pytest.approx(a.getSimulationTime())
    assert b.getState() == a.getState()
    # accelerations are recomputed on load (particle_system.cpp:274-283): same next steps
    for _ in range(5):
        a.update(cfg.dt)
        b.update(cfg.dt)
    sa, sb = a.getState(), b.getState()
    for k in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z"):
        assert np.array_equal(getattr(sa, k), getattr(sb, k)), k


def test_set_force_method_parameters_and_energy(nb, ctx):
    cfg = nb.SimulationConfig(particle_count=3000, force_method=nb.ForceMethod.DIRECT_N2, softening=0.1)
    s = nb.ParticleSystem()
    s.initialize(cfg, initial_conditions=nb.ic.plummer(3000, seed=2))
    d = s.getDeviceData()
    direct = np.stack([d.acc_x.cpu().numpy(), d.acc_y.cpu().numpy(), d.acc_z.cpu().numpy()], 1)
    e_direct = s.computeTotalEnergy()
    assert s.computeKineticEnergy() > 0 and s.computePotentialEnergy() < 0
    s.setForceMethod(nb.ForceMethod.BARNES_HUT)
    s.setBarnesHutTheta(0.0)  # opens everything: equals Direct
    assert isinstance(s.force_calculator_, nb.BarnesHutCalculator) and s.force_calculator_.getTheta() == 0.0
    s.force_calculator_.computeForces(d)
    bh = np.stack([d.acc_x.cpu().numpy(), d.acc_y.cpu().numpy(), d.acc_z.cpu().numpy()], 1)
    assert rel_err(bh, direct).max() < 1e-5
    assert s.computeTotalEnergy() == pytest.approx(e_direct, rel=1e-6)  # energies do not depend on it
    s.setForceMethod(nb.ForceMethod.SPATIAL_HASH)
    s.setSpatialHashCellSize(3.0)
    s.setSpatialHashCutoff(2.5)
    assert (s.force_calculator_.getCellSize(), s.force_calculator_.getCutoffRadius()) == (3.0, 2.5)
    s.update(0.001)
    for bad in (lambda: s.setGravitationalConstant(0.0), lambda: s.setSofteningParameter(-1.0),
                lambda: s.setTimeStep(5.0), lambda: s.setBarnesHutTheta(3.0),
                lambda: s.setSpatialHashCellSize(0.0), lambda: s.setSpatialHashCutoff(float("nan"))):
        with pytest.raises(nb.ValidationException):
            bad()
    s.setGravitationalConstant(2.0)
    s.setSofteningParameter(0.2)
    assert s.force_calculator_.getGravitationalConstant() == 2.0
    assert s.force_calculator_.getSofteningParameter() == pytest.approx(0.2)
    t = s.getSimulationTime()
    s.reset()
    assert s.getSimulationTime() == 0.0 and t > 0 and s.getParticleCount() == 3000
    # an un-initialised system ignores update() and reports zero energies (:115-118, :304-311)
    u = nb.ParticleSystem()
    u.update(0.1)
    assert u.computeKineticEnergy() == 0.0 and u.getSimulationTime() == 0.0


# src/main.cpp:335-416: `nbody_sim --benchmark` headless loop, with export/import of a checkpoint
@pytest.mark.parametrize("method", ["direct-n2", "barnes-hut", "spatial-hash"])
def test_headless_benchmark_mode(nb, ctx, tmp_path, method):
    import io
    import json
    out_json, ck = tmp_path / "bench.json", tmp_path / "state.nbody"
    buf = io.StringIO()
    o = nb.cli.parseAppCliOptions(["nbody_sim", "--particles", "4096", "--method", method, "--benchmark-steps", "5",
                                   "--benchmark-output", str(out_json), "--export", str(ck), "--softening", "0.05"])
    assert nb.cli.runBenchmarkMode(o, out=buf) == 0
    doc = json.loads(out_json.read_text())["benchmarks"][0]
    assert doc["benchmark_name"] == "application.benchmark_mode" and doc["particle_count"] == 4096
    assert doc["force_method"] == method.replace("-", "_") and doc["iterations"] == 5
    assert doc["metrics"]["wall_time_ms"] > 0 and doc["metrics"]["steps_per_s"] > 0
    assert doc["phase_timings"][0]["name"] == "simulation.update" and doc["phase_timings"][0]["samples"] == 5
    assert doc["parameters"]["softening"] == pytest.approx(0.05)
    st = nb.Serializer.load(str(ck))
    assert st.particle_count == 4096 and st.simulation_time == pytest.approx(5 * 0.001, rel=1e-4)
    # and back in through --import
    o2 = nb.cli.parseAppCliOptions(["nbody_sim", "--particles", "16", "--method", method, "--benchmark-steps", "1",
                                    "--import", str(ck)])
    buf2 = io.StringIO()
    assert nb.cli.runBenchmarkMode(o2, out=buf2) == 0
    assert "Imported 4096 particles" in buf2.getvalue()
    assert json.loads(buf2.getvalue().strip().splitlines()[-1])["benchmarks"][0]["particle_count"] == 4096


# resource cycling: handles are released (device memory returns to where it was)
def test_create_destroy_cycles_release_memory(nb, ctx):
    import gc
    import torch
    from gpu_util import to_device
    ic = nb.ic.plummer(20000, seed=2)
    d, _ = to_device(nb, ic)

    def cycle():
        bh = nb.BarnesHutCalculator(0.5)
        bh.computeForces(d)
        sh = nb.SpatialHashCalculator(1.0, 1.0)
        sh.computeForces(d)
        ps = nb.ParticleSystem()
        ps.initialize(nb.SimulationConfig(particle_count=5000, force_method=nb.ForceMethod.BARNES_HUT))
        ps.update(1e-3)
        del bh, sh, ps
        gc.collect()
        torch.cuda.synchronize()

    for _ in range(3):
        cycle()
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(25):
        cycle()
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, (free0, free1)  # no growth beyond allocator noise


# ref: benchmarks/benchmark_main.cpp:95-236 -- the five named benchmarks at the defaults of
# scripts/benchmark.sh (4,096 particles, 5 iterations), records shaped as the reference's
def test_named_benchmark_runner_all_five(nb, ctx, tmp_path, monkeypatch):
    import io
    import json
    from nbody_amd import benchmarks
    monkeypatch.setenv("NBODY_ENABLE_PROFILING", "1")
    path = tmp_path / "benchmark-results.json"
    out, err = io.StringIO(), io.StringIO()
    assert benchmarks.main(["nbody_benchmarks", "--output", str(path)], out=out, err=err) == 0, err.getvalue()
    recs = json.loads(path.read_text())["benchmarks"]
    assert json.loads(out.getvalue())["benchmarks"] == recs
    assert [r["benchmark_name"] for r in recs] == ["serialization.round_trip", "force.direct_n2", "force.barnes_hut",
                                                   "force.spatial_hash", "integration.velocity_verlet"]
    assert [r["force_method"] for r in recs] == ["direct_n2", "direct_n2", "barnes_hut", "spatial_hash", "direct_n2"]
    by = {r["benchmark_name"]: r for r in recs}
    for r in recs:
        assert r["particle_count"] == 4096 and r["iterations"] == 5 and r["metrics"]["wall_time_ms"] > 0
        assert r["parameters"]["particle_count"] == 4096
    # parameters per benchmark as the reference records them (:111-118, :146-148, :184-185)
    assert set(by["force.direct_n2"]["parameters"]) == {"particle_count", "cuda_block_size"}
    assert set(by["force.barnes_hut"]["parameters"]) == {"particle_count", "theta"}
    assert by["force.barnes_hut"]["parameters"]["theta"] == pytest.approx(0.5)
    assert set(by["force.spatial_hash"]["parameters"]) == {"particle_count", "cuda_block_size", "cell_size", "cutoff_radius"}
    assert set(by["integration.velocity_verlet"]["parameters"]) == {"particle_count", "dt", "cuda_block_size"}
    # phases under the reference's names; Barnes-Hut copies them into metrics (:203-209)
    assert [p["name"] for p in by["force.direct_n2"]["phase_timings"]] == ["force.direct_n2"]
    assert by["force.barnes_hut"]["metrics"]["barnes_hut.build_ms"] > 0
    assert [p["samples"] for p in by["integration.velocity_verlet"]["phase_timings"]] == [5]
    # device-synchronised timing: 4096^2 pairs cannot take less than a microsecond
    assert 1e6 < by["force.direct_n2"]["metrics"]["pair_interactions_per_s"] < 1e14
    # one benchmark by name
    out2 = io.StringIO()
    assert benchmarks.main(["x", "--benchmark", "force.spatial_hash", "--particle-count", "1000", "--iterations", "2"],
                           out=out2, err=err) == 0
    only = json.loads(out2.getvalue())["benchmarks"]
    assert len(only) == 1 and only[0]["particle_count"] == 1000
