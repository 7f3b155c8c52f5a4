// facade_tests.cpp -- the reference's hot-path GoogleTest cases, re-expressed on a 40-line
// harness (GoogleTest/RapidCheck are not installed), run against the C++ facade on a real GPU.
// Each case names the reference test it mirrors.  Exit code = number of failed checks.
#include <cmath>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "nbody_facade.hpp"

using namespace nbody;

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                                        \
  do {                                                                                     \
    g_checks++;                                                                            \
    if (!(cond)) { g_fail++; std::printf("  FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); } \
  } while (0)
#define CHECK_NEAR(a, b, tol) CHECK(std::fabs((double)(a) - (double)(b)) <= (tol))

struct Case { const char* name; std::function<void()> fn; };

static void fill_sphere(ParticleData& d, ParticleData& h, size_t n, float radius) {
  ParticleDataManager::allocateDevice(d, n);
  ParticleDataManager::allocateHost(h, n);
  SphericalDistParams p;
  p.center = Vec3(0, 0, 0);
  p.radius = radius;
  ParticleInitializer::initSpherical(h, p);
  ParticleDataManager::copyToDevice(d, h);
}

int main() {
  std::vector<Case> cases = {
      {"ForceCalculationTest.TwoBodyForce (test_force_calculation.cpp:13-30)", [] {
         Vec3 f = computeGravitationalForceCPU(Vec3(0, 0, 0), Vec3(1, 0, 0), 1, 1, 1, 0);
         CHECK(f.x > 0); CHECK_NEAR(f.y, 0, 1e-6); CHECK_NEAR(f.z, 0, 1e-6); CHECK_NEAR(f.length(), 1.0, 1e-5);
       }},
      {"ForceCalculationTest.SofteningPreventsInfinity + ForceDirection (:32-60)", [] {
         Vec3 f = computeGravitationalForceCPU(Vec3(0, 0, 0), Vec3(0.001f, 0, 0), 1, 1, 1, 0.1f);
         CHECK(std::isfinite(f.x) && std::isfinite(f.y) && std::isfinite(f.z));
         Vec3 g = computeGravitationalForceCPU(Vec3(0, 0, 0), Vec3(1, 1, 1), 1, 1, 1, 0.01f).normalized();
         Vec3 d = Vec3(1, 1, 1).normalized();
         CHECK_NEAR(g.x, d.x, 1e-5); CHECK_NEAR(g.y, d.y, 1e-5); CHECK_NEAR(g.z, d.z, 1e-5);
       }},
      {"DirectForceCalculatorTest.ComputeForces (:62-96) + two-body value", [] {
         ParticleData d, h;
         fill_sphere(d, h, 100, 5.0f);
         DirectForceCalculator calc(256);
         calc.setGravitationalConstant(1.0f);
         calc.setSofteningParameter(0.1f);
         calc.computeForces(&d);
         ParticleDataManager::copyToHost(h, d);
         bool finite = true, nonzero = false;
         for (size_t i = 0; i < h.count; i++) {
           finite = finite && std::isfinite(h.acc_x[i]) && std::isfinite(h.acc_y[i]) && std::isfinite(h.acc_z[i]);
           nonzero = nonzero || h.acc_x[i] != 0.f;
         }
         CHECK(finite); CHECK(nonzero);
         // every body against the one-pair helper summed on the host
         double worst = 0;
         for (size_t i = 0; i < h.count; i++) {
           double a[3] = {0, 0, 0};
           for (size_t j = 0; j < h.count; j++) {
             if (j == i) continue;
             Vec3 f = computeGravitationalForceCPU(Vec3(h.pos_x[i], h.pos_y[i], h.pos_z[i]),
                                                   Vec3(h.pos_x[j], h.pos_y[j], h.pos_z[j]), h.mass[i], h.mass[j], 1.0f, 0.1f);
             a[0] += f.x; a[1] += f.y; a[2] += f.z;
           }
           const double n = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
           const double e = std::sqrt(std::pow(h.acc_x[i] - a[0], 2) + std::pow(h.acc_y[i] - a[1], 2) + std::pow(h.acc_z[i] - a[2], 2)) / n;
           worst = std::max(worst, e);
         }
         CHECK(worst < 1e-5);
         ParticleDataManager::freeDevice(d); ParticleDataManager::freeHost(h);
         CHECK(d.pos_x == nullptr && d.count == 0);  // particle_init.cu:170-198
       }},
      {"IntegratorTest.SingleStepPositionUpdate (test_integrator.cpp:15-49)", [] {
         ParticleData d, h;
         ParticleDataManager::allocateDevice(d, 1); ParticleDataManager::allocateHost(h, 1);
         h.vel_x[0] = 1.0f; h.mass[0] = 1.0f;
         ParticleDataManager::copyToDevice(d, h);
         Integrator integ;
         integ.updatePositions(&d, 0.1f);
         ParticleDataManager::copyToHost(h, d);
         CHECK_NEAR(h.pos_x[0], 0.1, 1e-5); CHECK_NEAR(h.pos_y[0], 0, 1e-5); CHECK_NEAR(h.pos_z[0], 0, 1e-5);
         ParticleDataManager::freeDevice(d); ParticleDataManager::freeHost(h);
       }},
      {"IntegratorTest.KineticEnergyCalculation (:51-84)", [] {
         ParticleData d, h;
         ParticleDataManager::allocateDevice(d, 2); ParticleDataManager::allocateHost(h, 2);
         h.vel_x[0] = 1.0f; h.vel_y[1] = 2.0f; h.mass[0] = 1.0f; h.mass[1] = 2.0f;
         ParticleDataManager::copyToDevice(d, h);
         Integrator integ;
         CHECK_NEAR(integ.computeKineticEnergy(&d), 4.5, 1e-4);
         ParticleDataManager::freeDevice(d); ParticleDataManager::freeHost(h);
       }},
      {"Integrator.EnergyConservationProperty (:90-162; |dE| vs |PE|, see DESIGN.md)", [] {
         ParticleData d, h;
         ParticleDataManager::allocateDevice(d, 2); ParticleDataManager::allocateHost(h, 2);
         const float r = 5.0f, v = std::sqrt(1.0f / (2 * r));
         h.pos_x[0] = -r; h.pos_x[1] = r; h.vel_y[0] = -v; h.vel_y[1] = v; h.mass[0] = h.mass[1] = 1.0f;
         ParticleDataManager::copyToDevice(d, h);
         DirectForceCalculator fc;
         fc.setGravitationalConstant(1.0f); fc.setSofteningParameter(0.01f);
         Integrator integ;
         fc.computeForces(&d);
         const float e0 = integ.computeTotalEnergy(&d, 1.0f, 0.01f);
         for (int s = 0; s < 100; s++) integ.integrate(&d, &fc, 0.001f);
         const float e1 = integ.computeTotalEnergy(&d, 1.0f, 0.01f);
         CHECK(std::fabs(e1 - e0) < 0.01 * 0.1);
         ParticleDataManager::freeDevice(d); ParticleDataManager::freeHost(h);
       }},
      {"BarnesHutTreeTest.TreeConstruction/MassConservation/ComputeForces (test_barnes_hut.cpp:15-94)", [] {
         ParticleData d, h;
         fill_sphere(d, h, 100, 5.0f);
         BarnesHutTree tree(100);
         tree.build(&d);
         CHECK(tree.getNodeCount() > 0); CHECK(tree.verifyTreeStructure());
         tree.copyNodesToHost();
         CHECK(tree.verifyMassConservation(&h));
         CHECK(tree.getNodes()[0].particle_count == 100);
         BarnesHutCalculator calc(0.5f);
         calc.setGravitationalConstant(1.0f); calc.setSofteningParameter(0.1f);
         calc.computeForces(&d);
         ParticleDataManager::copyToHost(h, d);
         bool finite = true;
         for (size_t i = 0; i < h.count; i++) finite = finite && std::isfinite(h.acc_x[i] + h.acc_y[i] + h.acc_z[i]);
         CHECK(finite);
         ParticleDataManager::freeDevice(d); ParticleDataManager::freeHost(h);
       }},
      {"BarnesHut.ThetaAccuracyProperty (test_barnes_hut.cpp:131-201, test_spatial_hash.cpp:186-249)", [] {
         ParticleData d, h;
         fill_sphere(d, h, 50, 10.0f);
         DirectForceCalculator direct;
         direct.setSofteningParameter(0.1f);
         direct.computeForces(&d);
         ParticleDataManager::copyToHost(h, d);
         std::vector<float> ref(h.count);
         for (size_t i = 0; i < h.count; i++) ref[i] = Vec3(h.acc_x[i], h.acc_y[i], h.acc_z[i]).length();
         auto max_err = [&](float theta) {
           BarnesHutCalculator bh(theta);
           bh.setSofteningParameter(0.1f);
           bh.computeForces(&d);
           ParticleDataManager::copyToHost(h, d);
           float worst = 0;
           for (size_t i = 0; i < h.count; i++) {
             const float m = Vec3(h.acc_x[i], h.acc_y[i], h.acc_z[i]).length();
             worst = std::max(worst, std::fabs(m - ref[i]) / std::max(ref[i], 1e-10f));
           }
           return worst;
         };
         const float e01 = max_err(0.1f), e03 = max_err(0.3f), e08 = max_err(0.8f);
         CHECK(e01 < 0.10f); CHECK(e03 <= 1.1f * e08 + 1e-7f);
         ParticleDataManager::freeDevice(d); ParticleDataManager::freeHost(h);
       }},
      {"SpatialHashGridTest.CellIndexCalculation (test_spatial_hash.cpp:38-51)", [] {
         int3 a = SpatialHashGrid::getCellIndex(0.5f, 0.5f, 0.5f, 2.0f);
         int3 b = SpatialHashGrid::getCellIndex(2.5f, 4.5f, 6.5f, 2.0f);
         CHECK(a.x == 0 && a.y == 0 && a.z == 0); CHECK(b.x == 1 && b.y == 2 && b.z == 3);
       }},
      {"SpatialHash GridConstruction/ComputeForces/CellAssignment (test_spatial_hash.cpp:15-130)", [] {
         ParticleData d, h;
         fill_sphere(d, h, 100, 5.0f);
         SpatialHashGrid grid(100, 1.0f);
         grid.build(&d);
         CHECK(grid.getTotalCells() > 0);
         CHECK(grid.verifyCellAssignment(&h));
         std::vector<int> cs, ce, pc, si;
         grid.copyCellDataToHost(cs, ce, pc, si);
         std::vector<int> seen(100, 0);
         int total = 0;
         for (size_t c = 0; c < cs.size(); c++)
           for (int k = cs[c]; k < ce[c]; k++) { seen[si[k]]++; total++; CHECK(pc[si[k]] == (int)c); }
         bool once = true;
         for (int s : seen) once = once && s == 1;
         CHECK(once); CHECK(total == 100);
         SpatialHashCalculator calc(1.0f, 2.0f);
         calc.setGravitationalConstant(1.0f); calc.setSofteningParameter(0.1f);
         calc.computeForces(&d);
         ParticleDataManager::copyToHost(h, d);
         bool finite = true, nonzero = false;
         for (size_t i = 0; i < h.count; i++) {
           finite = finite && std::isfinite(h.acc_x[i] + h.acc_y[i] + h.acc_z[i]);
           nonzero = nonzero || h.acc_x[i] != 0.f;
         }
         CHECK(finite); CHECK(nonzero);
         ParticleDataManager::freeDevice(d); ParticleDataManager::freeHost(h);
       }},
      {"ParticleInitializer bounds (test_particle_data.cpp:40-112) + copy round trip", [] {
         ParticleData h;
         ParticleDataManager::allocateHost(h, 1000);
         UniformDistParams u; u.min_bounds = Vec3(-1, -2, -3); u.max_bounds = Vec3(1, 2, 3); u.min_mass = 0.5f; u.max_mass = 2.0f;
         ParticleInitializer::initUniform(h, u);
         bool ok = true;
         for (size_t i = 0; i < h.count; i++)
           ok = ok && std::fabs(h.pos_x[i]) <= 1.001f && std::fabs(h.pos_y[i]) <= 2.001f && std::fabs(h.pos_z[i]) <= 3.001f &&
                h.mass[i] >= 0.5f && h.mass[i] <= 2.0f && h.vel_x[i] == 0.f;
         CHECK(ok);
         SphericalDistParams s; s.center = Vec3(1, 1, 1); s.radius = 3.0f;
         ParticleInitializer::initSpherical(h, s);
         ok = true;
         for (size_t i = 0; i < h.count; i++)
           ok = ok && (Vec3(h.pos_x[i], h.pos_y[i], h.pos_z[i]) - s.center).length() <= 3.0f + 1e-3f;
         CHECK(ok);
         DiskDistParams k; k.radius = 4.0f; k.thickness = 0.5f;
         ParticleInitializer::initDisk(h, k);
         ok = true;
         for (size_t i = 0; i < h.count; i++)
           ok = ok && std::hypot(h.pos_x[i], h.pos_y[i]) <= 4.0f + 1e-2f && std::fabs(h.pos_z[i]) <= 0.25f + 1e-3f;
         CHECK(ok);
         ParticleData d, back;
         ParticleDataManager::allocateDevice(d, 1000); ParticleDataManager::allocateHost(back, 1000);
         ParticleDataManager::copyToDevice(d, h); ParticleDataManager::copyToHost(back, d);
         bool same = true;
         for (size_t i = 0; i < 1000; i++) same = same && back.pos_x[i] == h.pos_x[i] && back.vel_y[i] == h.vel_y[i] && back.mass[i] == h.mass[i];
         CHECK(same);
         ParticleDataManager::freeDevice(d); ParticleDataManager::freeHost(h); ParticleDataManager::freeHost(back);
       }},
      {"createForceCalculator (force_spatial_hash.cu:380-401) + error mapping", [] {
         SimulationConfig cfg;
         cfg.G = 2.0f; cfg.softening = 0.05f; cfg.barnes_hut_theta = 0.7f; cfg.spatial_hash_cell_size = 2.0f; cfg.spatial_hash_cutoff = 1.5f;
         auto a = createForceCalculator(ForceMethod::DIRECT_N2, cfg);
         auto b = createForceCalculator(ForceMethod::BARNES_HUT, cfg);
         auto c = createForceCalculator(ForceMethod::SPATIAL_HASH, cfg);
         CHECK(dynamic_cast<DirectForceCalculator*>(a.get()) && a->getMethod() == ForceMethod::DIRECT_N2);
         CHECK(dynamic_cast<BarnesHutCalculator*>(b.get()) && dynamic_cast<BarnesHutCalculator*>(b.get())->getTheta() == 0.7f);
         CHECK(dynamic_cast<SpatialHashCalculator*>(c.get()) && dynamic_cast<SpatialHashCalculator*>(c.get())->getCutoffRadius() == 1.5f);
         CHECK(a->getGravitationalConstant() == 2.0f && a->getSofteningParameter() == 0.05f);
         bool threw = false;
         try { ParticleData d; ParticleDataManager::allocateDevice(d, 0); } catch (const ValidationException&) { threw = true; }
         CHECK(threw);
         threw = false;
         try { ParticleData d; d.count = 4; DirectForceCalculator(256).computeForces(&d); } catch (const CudaException&) { threw = true; }
         CHECK(threw);
       }},
      // MI355X-native addition: the SAME plugin interface, force work shared by several ranks (here: virtual ranks
      // on device 0; on a node: ShardedDirectCalculator() = all GPUs).  A reference-style caller -- Integrator::
      // integrate with a ForceCalculator* -- runs BASELINE config 3's step without knowing about the ranks.
      {"ShardedDirectCalculator (4 ranks, plugin interface) == DirectForceCalculator through Integrator::integrate", [] {
         const size_t n = 30000;
         ParticleData d1, d2, h1, h2;
         fill_sphere(d1, h1, n, 5.0f);
         fill_sphere(d2, h2, n, 5.0f);
         DirectForceCalculator one(256);
         ShardedDirectCalculator many(std::vector<int>{0, 0, 0, 0});
         CHECK(many.getDeviceCount() == 4 && many.getMethod() == ForceMethod::DIRECT_N2);
         for (ForceCalculator* c : {static_cast<ForceCalculator*>(&one), static_cast<ForceCalculator*>(&many)}) {
           c->setGravitationalConstant(1.5f);
           c->setSofteningParameter(0.05f);
         }
         one.computeForces(&d1);
         many.computeForces(&d2);
         Integrator integ(256);
         for (int s = 0; s < 3; s++) {
           integ.integrate(&d1, &one, 0.001f);
           integ.integrate(&d2, &many, 0.001f);
         }
         ParticleDataManager::copyToHost(h1, d1);
         ParticleDataManager::copyToHost(h2, d2);
         double worst_a = 0, worst_x = 0;
         for (size_t i = 0; i < n; i++) {
           const double ax = h1.acc_x[i], ay = h1.acc_y[i], az = h1.acc_z[i];
           const double dx = h2.acc_x[i] - ax, dy = h2.acc_y[i] - ay, dz = h2.acc_z[i] - az;
           worst_a = std::fmax(worst_a, std::sqrt(dx * dx + dy * dy + dz * dz) / std::sqrt(ax * ax + ay * ay + az * az));
           worst_x = std::fmax(worst_x, std::fabs((double)h2.pos_x[i] - h1.pos_x[i]) + std::fabs((double)h2.vel_y[i] - h1.vel_y[i]));
         }
         std::printf("  sharded vs single: max rel acc diff %.3e, max |dx|+|dvy| %.3e\n", worst_a, worst_x);
         CHECK(worst_a < 1e-5); CHECK(worst_x < 1e-5);
         // the setters are plain stores read at the next call
         many.setSofteningParameter(0.2f);
         one.setSofteningParameter(0.2f);
         one.computeForces(&d1); many.computeForces(&d2);
         ParticleDataManager::copyToHost(h1, d1); ParticleDataManager::copyToHost(h2, d2);
         CHECK_NEAR(h1.acc_x[17], h2.acc_x[17], 1e-5 * std::fabs(h1.acc_x[17]) + 1e-9);
         ParticleDataManager::freeDevice(d1); ParticleDataManager::freeDevice(d2);
         ParticleDataManager::freeHost(h1); ParticleDataManager::freeHost(h2);
       }},
  };
  for (auto& c : cases) {
    const int before = g_fail;
    try {
      c.fn();
    } catch (const std::exception& e) {
      g_fail++;
      std::printf("  EXCEPTION %s\n", e.what());
    }
    std::printf("[%s] %s\n", g_fail == before ? " OK " : "FAIL", c.name);
  }
  std::printf("%d checks, %d failed\n", g_checks, g_fail);
  return g_fail;
}
