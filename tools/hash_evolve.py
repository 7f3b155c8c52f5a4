import os, sys, torch
sys.path.insert(0, "/root/repo")
import nbody_amd as nb
n = 4194304; dt = 1e-3
steps = int(sys.argv[1])
h = 0.5 * (n / 16.0) ** (1 / 3)
ps = nb.ParticleSystem()
ps.initialize(nb.SimulationConfig(particle_count=n, dt=dt, force_method=nb.ForceMethod.SPATIAL_HASH, softening=0.01,
                                  spatial_hash_cell_size=1.0, spatial_hash_cutoff=1.0),
              initial_conditions=nb.ic.uniform_box(n, seed=42, lo=-h, hi=h))
for _ in range(steps):
    ps.update(dt)
torch.cuda.synchronize()
