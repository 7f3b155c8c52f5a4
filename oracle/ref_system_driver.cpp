// ref_system_driver.cpp -- test infrastructure (oracle/): drives the REFERENCE's own, unmodified
// nbody::ParticleSystem (compiled from /root/reference/src/core/particle_system.cpp by Makefile.ref and
// linked against this repo's facade), so that tests can compare a run of the reference's orchestration
// code with the same run of n-body_amd/system.py.
//   ref_system_driver <method 0|1|2> <distribution 0|1|2> <particles> <steps> <dt> <out.nbody>
#include <cstdio>
#include <cstdlib>

#include "nbody/particle_system.hpp"

using namespace nbody;

int main(int argc, char** argv) {
  if (argc < 7) {
    std::fprintf(stderr, "usage: %s method distribution particles steps dt out.nbody\n", argv[0]);
    return 2;
  }
  try {
    SimulationConfig cfg;
    cfg.force_method = static_cast<ForceMethod>(std::atoi(argv[1]));
    cfg.init_distribution = static_cast<InitDistribution>(std::atoi(argv[2]));
    cfg.particle_count = static_cast<size_t>(std::strtoull(argv[3], nullptr, 10));
    const int steps = std::atoi(argv[4]);
    cfg.dt = static_cast<float>(std::atof(argv[5]));
    ParticleSystem system;
    system.initialize(cfg);
    for (int s = 0; s < steps; s++) system.update(cfg.dt);
    system.saveState(argv[6]);
    std::printf("time %.9g KE %.9g PE %.9g\n", system.getSimulationTime(), system.computeKineticEnergy(),
                system.computePotentialEnergy());
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
