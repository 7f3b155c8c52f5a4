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
