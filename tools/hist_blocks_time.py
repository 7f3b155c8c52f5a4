import sys, time, torch
sys.path.insert(0, "/root/repo")
import nbody_amd as nb
dt = 1e-3
out = []
for method, n in (("bh", 1048576), ("hash", 4194304), ("hash", 1048576)):
    if method == "bh":
        ic = nb.ic.two_galaxies(n, seed=42); ic["mass"] = (ic["mass"] / n).astype("float32")
        kw = dict(force_method=nb.ForceMethod.BARNES_HUT, softening=0.1, barnes_hut_theta=0.5)
    else:
        h = 0.5 * (n / 16.0) ** (1 / 3)
        ic = nb.ic.uniform_box(n, seed=42, lo=-h, hi=h)
        kw = dict(force_method=nb.ForceMethod.SPATIAL_HASH, softening=0.01, spatial_hash_cell_size=1.0, spatial_hash_cutoff=1.0)
    ps = nb.ParticleSystem()
    ps.initialize(nb.SimulationConfig(particle_count=n, dt=dt, **kw), initial_conditions=ic)
    for _ in range(20):
        ps.update(dt)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(100):
            ps.update(dt)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 10)
    out.append("%s N=%d %.4f" % (method, n, best))
print(" | ".join(out))
