"""The C-ABI library loads and exports every symbol include/nbody_hip.h declares; host-side
logic (validators, factory, struct layout) behaves like the reference.  CPU-only: no compute
call is made without a GPU -- and the one that is attempted must fail loudly."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = "".join(open(os.path.join(ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(ROOT, "include")))
                  if h.endswith(".h"))   # nbody_hip.h and nbody_hip_comm.h
    return sorted(set(re.findall(r"NBODY_HIP_API\s+[\w\s\*]+?\b(nbody_hip_\w+)\s*\(", src)))


def test_library_exports_every_header_symbol(nb):
    lib = nb._lib.load()
    names = _header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    # and the Python prototypes cover the header exactly
    assert sorted(nb._lib.PROTOTYPES) == names
    assert lib.nbody_hip_abi_version() == 1


def test_particle_data_layout_matches_reference(nb):
    # include/nbody/types.hpp:234-276: 13 float* + size_t = 112 bytes, count at offset 104
    S = nb._lib.ParticleDataStruct
    assert C.sizeof(S) == 112
    assert S.count.offset == 104
    assert [f for f, _ in S._fields_[:13]] == [
        "pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "acc_x", "acc_y", "acc_z",
        "acc_old_x", "acc_old_y", "acc_old_z", "mass"]


def test_facade_classes_have_the_reference_layout():
    """oracle/layout_probe.cpp compiled against the reference's headers and against the facade's
    header (-fno-access-control): size, alignment and every data-member offset of all classes that
    cross the C++ boundary are equal.  A caller built with the reference's headers puts
    BarnesHutTree / SpatialHashGrid on its own stack (ref: tests/test_barnes_hut.cpp:29,54,118,
    tests/test_spatial_hash.cpp:29,110) -- 96 bytes each."""
    import subprocess
    ref = os.path.join(ROOT, "oracle", "_ref")
    a, b = (os.path.join(ref, n) for n in ("layout_probe_ref", "layout_probe_facade"))
    if not (os.path.exists(a) and os.path.exists(b)):
        pytest.skip("oracle/_ref/layout_probe_* not built (needs /root/reference at build time)")
    ta = subprocess.run([a], capture_output=True, text=True, check=True).stdout
    tb = subprocess.run([b], capture_output=True, text=True, check=True).stdout
    assert ta == tb
    assert re.search(r"BarnesHutTree\s+size  96 align  8", ta)
    assert re.search(r"SpatialHashGrid\s+size  96 align  8", ta)
    assert len(ta.splitlines()) > 100


def test_no_cpu_fallback(nb):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = nb._lib.load()
    assert lib.nbody_hip_device_count() == 0
    h = C.c_void_p()
    rc = lib.nbody_hip_ctx_create(C.byref(h), 0, None)
    assert rc == nb._lib.ERR_DEVICE and not h.value
    assert b"no HIP device" in lib.nbody_hip_last_error()
    with pytest.raises(nb.DeviceException):
        nb.Context(0)
    s = nb._lib.ParticleDataStruct()
    assert lib.nbody_hip_particles_alloc(C.byref(s), 16) == nb._lib.ERR_DEVICE
    # argument validation happens before any device work
    assert lib.nbody_hip_particles_alloc(C.byref(s), 0) == nb._lib.ERR_VALIDATION
    assert b"greater than 0" in lib.nbody_hip_last_error()
    assert lib.nbody_hip_direct_forces(None, C.byref(s), 1.0, 0.0, 256) == nb._lib.ERR_STATE


def test_simulation_config_defaults(nb):
    c = nb.SimulationConfig()  # include/nbody/types.hpp:301-313
    assert (c.particle_count, c.dt, c.G, c.softening) == (10000, 0.001, 1.0, 0.1)
    assert (c.barnes_hut_theta, c.spatial_hash_cell_size, c.spatial_hash_cutoff) == (0.5, 1.0, 2.0)
    assert c.cuda_block_size == 256
    assert c.force_method == nb.ForceMethod.DIRECT_N2
    assert c.init_distribution == nb.InitDistribution.SPHERICAL
    assert [int(m) for m in nb.ForceMethod] == [0, 1, 2]


# tests/test_validation.cpp:13-197 (accept / reject ranges), src/utils/error_handling.cpp:46-123
def test_validators(nb):
    from nbody_amd import api
    V = nb.ValidationException
    api.validateParticleCountRange(1)
    api.validateParticleCountRange(100000000)
    for bad in (0, 100000001):
        with pytest.raises(V):
            api.validateParticleCountRange(bad)
    for ok in (1e-6, 0.001, 1.0):
        api.validateTimeStep(ok)
    for bad in (0.0, -0.1, 1.5, float("nan"), float("inf")):
        with pytest.raises(V):
            api.validateTimeStep(bad)
    for ok in (0.0, 0.01, 100.0):
        api.validateSoftening(ok)
    for bad in (-0.01, float("nan"), float("inf")):
        with pytest.raises(V):
            api.validateSoftening(bad)
    for ok in (0.0, 0.5, 2.0):
        api.validateTheta(ok)
    for bad in (-0.1, 2.1, float("nan")):
        with pytest.raises(V):
            api.validateTheta(bad)
    cfg = nb.SimulationConfig()
    api.validateSimulationConfig(cfg)
    for field, bad in (("G", 0.0), ("G", float("inf")), ("cuda_block_size", 0),
                       ("cuda_block_size", 2048), ("dt", 2.0), ("particle_count", 0)):
        c = nb.SimulationConfig(**{field: bad})
        with pytest.raises(V):
            api.validateSimulationConfig(c)
    c = nb.SimulationConfig(force_method=nb.ForceMethod.SPATIAL_HASH, spatial_hash_cell_size=0.0)
    with pytest.raises(V):
        api.validateSimulationConfig(c)
    c = nb.SimulationConfig(force_method=nb.ForceMethod.BARNES_HUT, barnes_hut_theta=3.0)
    with pytest.raises(V):
        api.validateSimulationConfig(c)
    # theta is only checked for Barnes-Hut (error_handling.cpp:51-53)
    api.validateSimulationConfig(nb.SimulationConfig(barnes_hut_theta=3.0))


def test_force_calculator_setters(nb):
    # force_calculator.hpp:60-88: defaults eps=0.01, eps2=1e-4, G=1; eps2 is the fp32 square
    calc = nb.DirectForceCalculator(256)
    assert calc.getSofteningParameter() == pytest.approx(0.01)
    assert calc.softening_eps2_ == pytest.approx(1e-4)
    assert calc.getGravitationalConstant() == 1.0
    calc.setSofteningParameter(0.1)
    assert calc.softening_eps2_ == float(np.float32(0.1) * np.float32(0.1))
    calc.setGravitationalConstant(2.0)
    assert calc.getGravitationalConstant() == 2.0
    assert calc.getMethod() == nb.ForceMethod.DIRECT_N2
    assert calc.getBlockSize() == 256


def test_ic_generators_are_deterministic(nb):
    a, b = nb.ic.plummer(1000, seed=42), nb.ic.plummer(1000, seed=42)
    for k in a:
        assert a[k].dtype == np.float32 and np.array_equal(a[k], b[k])
    assert abs(float(a["mass"].sum()) - 1.0) < 1e-5
    r = np.sqrt(a["pos_x"] ** 2 + a["pos_y"] ** 2 + a["pos_z"] ** 2)
    assert r.max() < 10.6 and 0.5 < np.median(r) < 2.0  # Plummer half-mass radius ~1.3 a
    d = nb.ic.two_galaxies(1000)
    assert d["pos_x"].size == 1000


# a14 pinned to the reference's own validators: oracle/_ref/ref_validate_driver runs the functions
# compiled from /root/reference/src/utils/error_handling.cpp; same verdict and same message, value by value
REF_VAL = os.path.join(ROOT, "oracle", "_ref", "ref_validate_driver")


@pytest.mark.skipif(not os.path.exists(REF_VAL), reason="oracle/_ref not built (needs /root/reference at build time)")
def test_validators_equal_reference_validators(nb):
    import struct
    import subprocess

    import numpy as np
    from nbody_amd import api

    def hexf(x):
        return "%08x" % struct.unpack("<I", struct.pack("<f", x))[0]

    floats = [0.0, -0.0, 1e-45, 1e-8, 1e-4, 1e-3, 0.5, 1.0, float(np.nextafter(np.float32(1), np.float32(2))), 1.5, 2.0,
              float(np.nextafter(np.float32(2), np.float32(3))), 2.5, 100.0, 3e38, -1e-45, -0.1, -5.0,
              float("inf"), float("-inf"), float("nan")]
    counts = [0, 1, 2, 4096, 100000000, 100000001, 2 ** 40]
    reqs, calls = [], []
    for c in counts:
        reqs.append(f"count {c}")
        calls.append(lambda c=c: api.validateParticleCountRange(c))
    for kind, fn in (("timestep", api.validateTimeStep), ("softening", api.validateSoftening), ("theta", api.validateTheta)):
        for x in floats:
            reqs.append(f"{kind} {hexf(x)}")
            calls.append(lambda fn=fn, x=x: fn(float(np.float32(x))))
    rng = np.random.default_rng(0)
    pick = lambda seq: seq[int(rng.integers(len(seq)))]  # noqa: E731
    for _ in range(300):  # random configs out of the same awkward values
        cfg = dict(particle_count=pick(counts), dt=pick(floats), softening=pick(floats), barnes_hut_theta=pick(floats),
                   G=pick(floats), cuda_block_size=pick([0, 1, 256, 1024, 1025, -3]), force_method=pick([0, 1, 2]),
                   spatial_hash_cell_size=pick(floats), spatial_hash_cutoff=pick(floats))
        reqs.append("config {particle_count} {} {} {} {} {cuda_block_size} {force_method} {} {}".format(
            hexf(cfg["dt"]), hexf(cfg["softening"]), hexf(cfg["barnes_hut_theta"]), hexf(cfg["G"]),
            hexf(cfg["spatial_hash_cell_size"]), hexf(cfg["spatial_hash_cutoff"]), **cfg))
        f32 = {k: (float(np.float32(v)) if isinstance(v, float) else v) for k, v in cfg.items()}
        f32["force_method"] = nb.ForceMethod(cfg["force_method"])
        calls.append(lambda f32=f32: api.validateSimulationConfig(nb.SimulationConfig(**f32)))
    out = subprocess.run([REF_VAL], input="\n".join(reqs) + "\n", capture_output=True, text=True, timeout=60)
    ref = out.stdout.splitlines()
    assert len(ref) == len(reqs)
    for req, expected, call in zip(reqs, ref, calls):
        try:
            call()
            got = "ok"
        except nb.ValidationException as e:
            got = "Validation Error: " + str(e)
        assert got == expected, (req, got, expected)
