// layout_probe.cpp -- test infrastructure.  Prints size, alignment and EVERY data-member offset of
// the classes that cross the C++ plugin boundary.  oracle/Makefile.ref compiles it twice with
// -fno-access-control:
//   -DPROBE_REF     against the reference's own headers (/root/reference/include/nbody/*.hpp)
//   -DPROBE_FACADE  against n-body_amd/facade/include/nbody_facade.hpp
// and fails the build when the two tables differ; tests/test_abi.py diffs them again.  A caller
// compiled against the reference's headers reserves storage for these objects itself (its tests put
// BarnesHutTree / SpatialHashGrid on the stack, tests/test_barnes_hut.cpp:29,54,118,
// tests/test_spatial_hash.cpp:29,110), so the facade must not add or move a single byte.
#include <cstddef>
#include <cstdio>

#if defined(PROBE_REF)
#include "nbody/barnes_hut_tree.hpp"
#include "nbody/error_handling.hpp"
#include "nbody/force_calculator.hpp"
#include "nbody/integrator.hpp"
#include "nbody/particle_data.hpp"
#include "nbody/spatial_hash_grid.hpp"
#include "nbody/types.hpp"
#elif defined(PROBE_FACADE)
#include "nbody_facade.hpp"
#else
#error "define PROBE_REF or PROBE_FACADE"
#endif

using namespace nbody;

// the reference's values, measured with its own headers (g++ 11, x86-64)
static_assert(sizeof(ParticleData) == 112, "ParticleData");
static_assert(sizeof(SimulationConfig) == 48, "SimulationConfig");
static_assert(sizeof(OctreeNode) == 76, "OctreeNode");
static_assert(sizeof(BarnesHutTree) == 96, "BarnesHutTree");
static_assert(sizeof(SpatialHashGrid) == 96, "SpatialHashGrid");
static_assert(sizeof(Integrator) == 24, "Integrator");
static_assert(sizeof(DirectForceCalculator) == 24, "DirectForceCalculator");
static_assert(sizeof(BarnesHutCalculator) == 40, "BarnesHutCalculator");
static_assert(sizeof(SpatialHashCalculator) == 40, "SpatialHashCalculator");

#define TYPE(T) std::printf("%-24s size %3zu align %2zu\n", #T, sizeof(T), alignof(T))
#define MEMBER(T, m) \
  std::printf("  %-22s off %3zu size %3zu\n", #m, offsetof(T, m), sizeof(static_cast<T*>(nullptr)->m))

int main() {
  TYPE(Vec3); MEMBER(Vec3, x); MEMBER(Vec3, y); MEMBER(Vec3, z);
  TYPE(ParticleData);
  MEMBER(ParticleData, pos_x); MEMBER(ParticleData, pos_y); MEMBER(ParticleData, pos_z);
  MEMBER(ParticleData, vel_x); MEMBER(ParticleData, vel_y); MEMBER(ParticleData, vel_z);
  MEMBER(ParticleData, acc_x); MEMBER(ParticleData, acc_y); MEMBER(ParticleData, acc_z);
  MEMBER(ParticleData, acc_old_x); MEMBER(ParticleData, acc_old_y); MEMBER(ParticleData, acc_old_z);
  MEMBER(ParticleData, mass); MEMBER(ParticleData, count);
  TYPE(SimulationConfig);
  MEMBER(SimulationConfig, particle_count); MEMBER(SimulationConfig, init_distribution);
  MEMBER(SimulationConfig, force_method); MEMBER(SimulationConfig, dt); MEMBER(SimulationConfig, G);
  MEMBER(SimulationConfig, softening); MEMBER(SimulationConfig, barnes_hut_theta);
  MEMBER(SimulationConfig, spatial_hash_cell_size); MEMBER(SimulationConfig, spatial_hash_cutoff);
  MEMBER(SimulationConfig, cuda_block_size);
  TYPE(UniformDistParams);
  MEMBER(UniformDistParams, min_bounds); MEMBER(UniformDistParams, max_bounds);
  MEMBER(UniformDistParams, min_mass); MEMBER(UniformDistParams, max_mass);
  TYPE(SphericalDistParams);
  MEMBER(SphericalDistParams, center); MEMBER(SphericalDistParams, radius);
  MEMBER(SphericalDistParams, min_mass); MEMBER(SphericalDistParams, max_mass);
  TYPE(DiskDistParams);
  MEMBER(DiskDistParams, center); MEMBER(DiskDistParams, radius); MEMBER(DiskDistParams, thickness);
  MEMBER(DiskDistParams, min_mass); MEMBER(DiskDistParams, max_mass); MEMBER(DiskDistParams, rotation_speed);
  TYPE(OctreeNode);
  MEMBER(OctreeNode, center); MEMBER(OctreeNode, half_size); MEMBER(OctreeNode, center_of_mass);
  MEMBER(OctreeNode, total_mass); MEMBER(OctreeNode, children); MEMBER(OctreeNode, particle_index);
  MEMBER(OctreeNode, is_leaf); MEMBER(OctreeNode, particle_count);
  TYPE(BarnesHutTree);
  MEMBER(BarnesHutTree, d_nodes_); MEMBER(BarnesHutTree, d_sorted_indices_); MEMBER(BarnesHutTree, d_morton_codes_);
  MEMBER(BarnesHutTree, h_nodes_); MEMBER(BarnesHutTree, max_particles_); MEMBER(BarnesHutTree, max_nodes_);
  MEMBER(BarnesHutTree, node_count_); MEMBER(BarnesHutTree, max_depth_); MEMBER(BarnesHutTree, bbox_min_);
  MEMBER(BarnesHutTree, bbox_max_);
  TYPE(SpatialHashGrid);
  MEMBER(SpatialHashGrid, d_cell_start_); MEMBER(SpatialHashGrid, d_cell_end_);
  MEMBER(SpatialHashGrid, d_particle_cell_); MEMBER(SpatialHashGrid, d_sorted_indices_);
  MEMBER(SpatialHashGrid, d_cell_counts_); MEMBER(SpatialHashGrid, max_particles_);
  MEMBER(SpatialHashGrid, cell_size_); MEMBER(SpatialHashGrid, grid_dims_); MEMBER(SpatialHashGrid, total_cells_);
  MEMBER(SpatialHashGrid, bbox_min_); MEMBER(SpatialHashGrid, bbox_max_);
  TYPE(ForceCalculator);
  MEMBER(ForceCalculator, softening_eps_); MEMBER(ForceCalculator, softening_eps2_); MEMBER(ForceCalculator, G_);
  TYPE(DirectForceCalculator); MEMBER(DirectForceCalculator, block_size_);
  TYPE(BarnesHutCalculator); MEMBER(BarnesHutCalculator, tree_); MEMBER(BarnesHutCalculator, theta_);
  TYPE(SpatialHashCalculator);
  MEMBER(SpatialHashCalculator, grid_); MEMBER(SpatialHashCalculator, cell_size_);
  MEMBER(SpatialHashCalculator, cutoff_radius_);
  TYPE(Integrator);
  MEMBER(Integrator, block_size_); MEMBER(Integrator, d_scratch_); MEMBER(Integrator, scratch_blocks_);
  TYPE(ParticleDataManager);
  TYPE(ParticleInitializer);
  TYPE(CudaException);
  MEMBER(CudaException, error_msg_); MEMBER(CudaException, file_); MEMBER(CudaException, line_);
  TYPE(ResourceException); MEMBER(ResourceException, required_); MEMBER(ResourceException, available_);
  TYPE(ValidationException);
  return 0;
}
