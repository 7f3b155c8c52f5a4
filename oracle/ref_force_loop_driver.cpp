// ref_force_loop_driver.cpp -- test infrastructure (oracle/): runs the REFERENCE's own CPU all-pairs loop,
// computeReferenceForces (examples/example_force_methods.cpp:34-67) -- the only CPU force loop the reference
// holds -- on a checkpoint, and writes the forces it returns.  The example's translation unit is included
// as it lies under /root/reference (Makefile.ref passes -I$(REF)/examples); only its `main` is renamed, so the
// function under test is the reference's code compiled here, not a restatement.  The state reaches the
// function through the reference's own ParticleSystem::loadState / getState (linked on this repo's facade:
// the round trip through device memory runs on the GPU box; the loop itself is host code).
//   ref_force_loop_driver <in.nbody> <out.f32>      out = N x 3 float32, row i = {ax, ay, az} of body i
#define main ref_example_force_methods_main
#include "example_force_methods.cpp"
#undef main

#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
  if (argc != 3) {
    std::fprintf(stderr, "usage: %s in.nbody out.f32\n", argv[0]);
    return 2;
  }
  try {
    ParticleSystem system;
    system.loadState(argv[1]);
    std::vector<float> forces;
    computeReferenceForces(system, forces);
    std::FILE* f = std::fopen(argv[2], "wb");
    if (!f || std::fwrite(forces.data(), sizeof(float), forces.size(), f) != forces.size()) {
      std::fprintf(stderr, "cannot write %s\n", argv[2]);
      return 1;
    }
    std::fclose(f);
    std::printf("bodies %zu G %.9g eps %.9g\n", forces.size() / 3, system.getGravitationalConstant(),
                system.getSofteningParameter());
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
