// ref_tree_grid_driver.cpp -- test infrastructure, compiled against the REFERENCE's headers
// (-I/root/reference/include), linked against this repo's facade.  It does, with objects it
// constructs ITSELF from the reference's class definitions, what the reference's device tests do:
//   tests/test_barnes_hut.cpp:15-62,99-127    BuildTree, MassConservation, TreeContainsAllParticles
//   tests/test_barnes_hut.cpp:131-201         ApproximationConvergence (err(0.3) <= 1.1 err(0.8))
//   tests/test_spatial_hash.cpp:15-51,89-130  BuildGrid, CellIndexCalculation, CellAssignmentCorrectness
// plus the plugin contract of include/nbody/force_calculator.hpp:36-58 (a user subclass's virtual
// computeForces is what Integrator::integrate calls, integrator.cu:234).
// Every BarnesHutTree / SpatialHashGrid is built twice: as a plain local (what the tests do; the
// -fsanitize=address build of this file + the facade catches any write past the object) and inside
// a canary frame (bytes before and after the storage must survive every method).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <set>
#include <vector>

#include "nbody/barnes_hut_tree.hpp"
#include "nbody/force_calculator.hpp"
#include "nbody/integrator.hpp"
#include "nbody/particle_data.hpp"
#include "nbody/spatial_hash_grid.hpp"
#include "nbody/types.hpp"

using namespace nbody;

static_assert(sizeof(BarnesHutTree) == 96, "reference BarnesHutTree");
static_assert(sizeof(SpatialHashGrid) == 96, "reference SpatialHashGrid");

static int g_failed = 0;
#define EXPECT(cond)                                                          \
  do {                                                                        \
    if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); g_failed++; } \
  } while (0)

template <class T>
struct Framed {  // storage the size the REFERENCE header says, between two canaries
  unsigned char pre[64];
  alignas(16) unsigned char obj[sizeof(T)];
  unsigned char post[64];
  Framed() { std::memset(pre, 0xA5, sizeof pre); std::memset(post, 0x5A, sizeof post); }
  bool intact() const {
    for (unsigned char c : pre) if (c != 0xA5) return false;
    for (unsigned char c : post) if (c != 0x5A) return false;
    return true;
  }
};

struct Bodies {
  ParticleData d, h;
  explicit Bodies(size_t n) {
    ParticleDataManager::allocateDevice(d, n);
    ParticleDataManager::allocateHost(h, n);
  }
  ~Bodies() {
    ParticleDataManager::freeDevice(d);
    ParticleDataManager::freeHost(h);
  }
  void sphere(float radius, unsigned seed = 42, float mmin = 1.0f, float mmax = 1.0f) {
    SphericalDistParams p;
    p.center = Vec3(0, 0, 0); p.radius = radius; p.min_mass = mmin; p.max_mass = mmax;
    ParticleInitializer::initSpherical(h, p, seed);
    ParticleDataManager::copyToDevice(d, h);
  }
  void box(float half) {
    UniformDistParams p;
    p.min_bounds = Vec3(-half, -half, -half); p.max_bounds = Vec3(half, half, half);
    ParticleInitializer::initUniform(h, p);
    ParticleDataManager::copyToDevice(d, h);
  }
};

static void tree_build_and_mass() {
  {  // BuildTree (test_barnes_hut.cpp:15-36)
    Bodies b(100);
    b.sphere(10.0f);
    volatile unsigned long long before = 0x1122334455667788ull;
    BarnesHutTree tree(100);
    volatile unsigned long long after = 0x8877665544332211ull;
    tree.build(&b.d);
    EXPECT(tree.getNodeCount() > 0);
    EXPECT(before == 0x1122334455667788ull && after == 0x8877665544332211ull);
  }
  {  // MassConservation (test_barnes_hut.cpp:38-62)
    Bodies b(50);
    b.sphere(5.0f);
    BarnesHutTree tree(50);
    tree.build(&b.d);
    tree.copyNodesToHost();
    EXPECT(tree.verifyMassConservation(&b.h));
    EXPECT(tree.getNodes() != nullptr);
    // the root of the copied tree carries the total mass (force_barnes_hut.cu:511-519)
    EXPECT(std::fabs(tree.getNodes()[0].total_mass - 50.0f) < 0.05f);
    EXPECT(tree.getNodes()[0].particle_count == 50);
  }
  for (unsigned seed : {1u, 7u, 42u, 1234u, 99991u}) {  // TreeContainsAllParticles (:99-127)
    Bodies b(50);
    b.sphere(10.0f, seed);
    Framed<BarnesHutTree> f;
    BarnesHutTree* tree = new (f.obj) BarnesHutTree(50);
    EXPECT(f.intact());
    tree->build(&b.d);
    EXPECT(f.intact());
    tree->copyNodesToHost();
    EXPECT(tree->verifyMassConservation(&b.h));
    tree->computeForces(&b.d, 0.5f, 1.0f, 0.1f);
    ParticleDataManager::copyToHost(b.h, b.d);
    for (size_t i = 0; i < 50; i++) EXPECT(std::isfinite(b.h.acc_x[i]));
    EXPECT(f.intact());
    tree->~BarnesHutTree();
    EXPECT(f.intact());
  }
}

static float summed_error(Bodies& b, float theta, const std::vector<float>& ax, const std::vector<float>& ay,
                          const std::vector<float>& az) {
  BarnesHutCalculator calc(theta);
  calc.setGravitationalConstant(1.0f);
  calc.setSofteningParameter(0.1f);
  calc.computeForces(&b.d);
  ParticleDataManager::copyToHost(b.h, b.d);
  float e = 0.0f;
  for (size_t i = 0; i < b.h.count; i++) {
    const float dx = b.h.acc_x[i] - ax[i], dy = b.h.acc_y[i] - ay[i], dz = b.h.acc_z[i] - az[i];
    e += std::sqrt(dx * dx + dy * dy + dz * dz);
  }
  return e;
}

static void tree_convergence() {  // test_barnes_hut.cpp:131-201
  const size_t N = 50;
  Bodies b(N);
  b.sphere(10.0f, 42);
  DirectForceCalculator direct;
  direct.setGravitationalConstant(1.0f);
  direct.setSofteningParameter(0.1f);
  direct.computeForces(&b.d);
  ParticleDataManager::copyToHost(b.h, b.d);
  std::vector<float> ax(b.h.acc_x, b.h.acc_x + N), ay(b.h.acc_y, b.h.acc_y + N), az(b.h.acc_z, b.h.acc_z + N);
  const float e1 = summed_error(b, 0.8f, ax, ay, az), e2 = summed_error(b, 0.3f, ax, ay, az);
  EXPECT(e2 <= e1 * 1.1f);
  std::printf("  convergence: err(0.8) = %.4g, err(0.3) = %.4g\n", e1, e2);
}

static void grid_build_and_cells() {
  {  // BuildGrid (test_spatial_hash.cpp:15-36)
    Bodies b(100);
    b.box(10.0f);
    volatile unsigned long long before = 0x1122334455667788ull;
    SpatialHashGrid grid(100, 2.0f);
    volatile unsigned long long after = 0x8877665544332211ull;
    grid.build(&b.d);
    EXPECT(grid.getTotalCells() > 0);
    EXPECT(grid.getCellSize() == 2.0f);
    const int3 dims = grid.getGridDims();
    EXPECT(dims.x * dims.y * dims.z == grid.getTotalCells());
    EXPECT(before == 0x1122334455667788ull && after == 0x8877665544332211ull);
  }
  {  // CellIndexCalculation (:38-51)
    const int3 c1 = SpatialHashGrid::getCellIndex(0.5f, 0.5f, 0.5f, 2.0f);
    const int3 c2 = SpatialHashGrid::getCellIndex(2.5f, 4.5f, 6.5f, 2.0f);
    EXPECT(c1.x == 0 && c1.y == 0 && c1.z == 0);
    EXPECT(c2.x == 1 && c2.y == 2 && c2.z == 3);
  }
  for (float cell : {0.6f, 1.0f, 2.5f, 9.9f}) {  // CellAssignmentCorrectness (:89-130)
    const size_t N = 50;
    Bodies b(N);
    b.box(10.0f);
    Framed<SpatialHashGrid> f;
    SpatialHashGrid* grid = new (f.obj) SpatialHashGrid(N, cell);
    EXPECT(f.intact());
    grid->build(&b.d);
    EXPECT(f.intact());
    std::vector<int> cell_start, cell_end, particle_cells, sorted_indices;
    grid->copyCellDataToHost(cell_start, cell_end, particle_cells, sorted_indices);
    std::set<int> seen;
    bool once = true;
    for (int c = 0; c < grid->getTotalCells(); c++)
      for (int i = cell_start[c]; i < cell_end[c]; i++) once = once && seen.insert(sorted_indices[i]).second;
    EXPECT(once);
    EXPECT(seen.size() == N);
    EXPECT(grid->verifyCellAssignment(&b.h));
    grid->computeForces(&b.d, 2.0f * cell, 1.0f, 0.1f);
    ParticleDataManager::copyToHost(b.h, b.d);
    for (size_t i = 0; i < N; i++) EXPECT(std::isfinite(b.h.acc_x[i]));
    EXPECT(f.intact());
    grid->~SpatialHashGrid();
    EXPECT(f.intact());
  }
}

// A user plugin written against the reference's interface: derives from the engine's Direct
// calculator and overrides the virtual.  Integrator::integrate must call THIS computeForces.
struct CountingDirect : DirectForceCalculator {
  int calls = 0;
  void computeForces(ParticleData* d) override {
    calls++;
    DirectForceCalculator::computeForces(d);
  }
};
struct ZeroForce : ForceCalculator {  // a from-scratch strategy: no force at all
  int calls = 0;
  void computeForces(ParticleData*) override { calls++; }
  ForceMethod getMethod() const override { return ForceMethod::DIRECT_N2; }
};

static void plugin_contract() {
  const size_t N = 64;
  Bodies b(N);
  b.sphere(5.0f);
  Integrator integ;
  CountingDirect counting;
  counting.setGravitationalConstant(1.0f);
  counting.setSofteningParameter(0.1f);
  counting.computeForces(&b.d);
  counting.calls = 0;
  for (int s = 0; s < 3; s++) integ.integrate(&b.d, &counting, 1e-3f);
  EXPECT(counting.calls == 3);

  // same start, the engine's own calculator (fused launch sequence): same trajectory
  Bodies c(N);
  c.sphere(5.0f);
  DirectForceCalculator plain;
  plain.setGravitationalConstant(1.0f);
  plain.setSofteningParameter(0.1f);
  plain.computeForces(&c.d);
  for (int s = 0; s < 3; s++) integ.integrate(&c.d, &plain, 1e-3f);
  ParticleDataManager::copyToHost(b.h, b.d);
  ParticleDataManager::copyToHost(c.h, c.d);
  bool same = true;
  for (size_t i = 0; i < N; i++)
    same = same && b.h.pos_x[i] == c.h.pos_x[i] && b.h.vel_y[i] == c.h.vel_y[i] && b.h.acc_z[i] == c.h.acc_z[i];
  EXPECT(same);

  ZeroForce zero;  // getMethod() says DIRECT_N2, but it is not the engine's class: virtual call, no fusion
  integ.integrate(&c.d, &zero, 1e-3f);
  EXPECT(zero.calls == 1);
}

int main() {
  struct { const char* name; void (*fn)(); } cases[] = {
      {"tree_build_and_mass", tree_build_and_mass}, {"tree_convergence", tree_convergence},
      {"grid_build_and_cells", grid_build_and_cells}, {"plugin_contract", plugin_contract}};
  for (auto& c : cases) {
    const int before = g_failed;
    try {
      c.fn();
    } catch (const std::exception& e) {
      std::printf("FAIL %s threw: %s\n", c.name, e.what());
      g_failed++;
    }
    std::printf("%s %s\n", g_failed == before ? "PASS" : "FAIL", c.name);
  }
  std::printf("%s (%d failed checks)\n", g_failed ? "FAILED" : "ALL PASSED", g_failed);
  return g_failed ? 1 : 0;
}
