// ic_dump.cpp -- host-only: writes the facade's ParticleInitializer output (libstdc++ mt19937 +
// uniform_real_distribution, the reference's recipe) as raw float32 so that the Python mirror can
// be compared with it.  usage: ic_dump uniform|spherical|disk <n> <seed> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "nbody_facade.hpp"

using namespace nbody;

int main(int argc, char** argv) {
  if (argc < 5) return 1;
  const size_t n = std::strtoull(argv[2], nullptr, 10);
  const unsigned seed = static_cast<unsigned>(std::strtoul(argv[3], nullptr, 10));
  ParticleData h;
  ParticleDataManager::allocateHost(h, n);
  if (!std::strcmp(argv[1], "uniform")) {
    UniformDistParams p; p.min_bounds = Vec3(-10, -5, -2.5f); p.max_bounds = Vec3(10, 5, 7.5f); p.min_mass = 0.5f; p.max_mass = 2.0f;
    ParticleInitializer::initUniform(h, p, seed);
  } else if (!std::strcmp(argv[1], "spherical")) {
    SphericalDistParams p; p.center = Vec3(1, -2, 0.5f); p.radius = 10.0f; p.min_mass = 1.0f; p.max_mass = 3.0f;
    ParticleInitializer::initSpherical(h, p, seed);
  } else {
    DiskDistParams p; p.center = Vec3(0.5f, 0, -1); p.radius = 10.0f; p.thickness = 1.0f; p.rotation_speed = 0.5f;
    ParticleInitializer::initDisk(h, p, seed);
  }
  FILE* f = std::fopen(argv[4], "wb");
  if (!f) return 2;
  float* arr[7] = {h.pos_x, h.pos_y, h.pos_z, h.vel_x, h.vel_y, h.vel_z, h.mass};
  for (float* a : arr) std::fwrite(a, sizeof(float), n, f);
  std::fclose(f);
  ParticleDataManager::freeHost(h);
  return 0;
}
