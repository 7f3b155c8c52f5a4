// nbody_facade.hpp -- C++ host facade over the C ABI of libnbody_hip.so.
//
// One header that re-declares, in namespace nbody, the part of the reference's public C++
// interface that its CUDA translation units (src/cuda/*.cu) implement:
//   types           include/nbody/types.hpp              (Vec3, enums, ParticleData, configs)
//   exceptions      include/nbody/error_handling.hpp:29-102
//   ForceCalculator + Direct / BarnesHut / SpatialHash calculators, createForceCalculator,
//   computeGravitationalForceCPU                          include/nbody/force_calculator.hpp
//   Integrator      include/nbody/integrator.hpp
//   ParticleDataManager, ParticleInitializer             include/nbody/particle_data.hpp
//   BarnesHutTree / OctreeNode, SpatialHashGrid          barnes_hut_tree.hpp, spatial_hash_grid.hpp
// Names, signatures, data-member order and defaults match the reference so that objects compiled
// against the reference's own headers (e.g. its unmodified src/core/particle_system.cpp) link and
// run against libnbody_facade.so -- see INTEGRATION.md and oracle/Makefile.ref.  Every method
// forwards to one C-ABI call; there is no arithmetic in this layer.
#pragma once

#include <cmath>
#include <cstddef>
#include <memory>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#if !defined(__HIPCC__) && !defined(__CUDACC__)
#ifndef NBODY_FACADE_HAVE_INT3
#define NBODY_FACADE_HAVE_INT3
struct int3 { int x, y, z; };
inline constexpr int3 make_int3(int x, int y, int z) { return {x, y, z}; }
#endif
#endif

struct nbody_hip_ctx;
struct nbody_hip_tree;
struct nbody_hip_grid;
struct nbody_hip_comm;
struct nbody_hip_sharded_direct;

namespace nbody {

enum class ForceMethod { DIRECT_N2, BARNES_HUT, SPATIAL_HASH };
enum class InitDistribution { UNIFORM, SPHERICAL, DISK };

struct Vec3 {
  float x, y, z;
  Vec3() : x(0), y(0), z(0) {}
  Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
  Vec3 operator+(const Vec3& v) const { return {x + v.x, y + v.y, z + v.z}; }
  Vec3 operator-(const Vec3& v) const { return {x - v.x, y - v.y, z - v.z}; }
  Vec3 operator*(float s) const { return {x * s, y * s, z * s}; }
  Vec3 operator/(float s) const { return {x / s, y / s, z / s}; }
  Vec3& operator+=(const Vec3& v) { x += v.x; y += v.y; z += v.z; return *this; }
  Vec3& operator-=(const Vec3& v) { x -= v.x; y -= v.y; z -= v.z; return *this; }
  float dot(const Vec3& v) const { return x * v.x + y * v.y + z * v.z; }
  float length2() const { return dot(*this); }
  float length() const { return std::sqrt(length2()); }
  Vec3 normalized() const { const float l = length(); return l > 0 ? *this / l : Vec3(); }
};
inline Vec3 operator*(float s, const Vec3& v) { return v * s; }

struct ParticleData {  // 13 device (or host) arrays + count, 112 bytes
  float *pos_x = nullptr, *pos_y = nullptr, *pos_z = nullptr;
  float *vel_x = nullptr, *vel_y = nullptr, *vel_z = nullptr;
  float *acc_x = nullptr, *acc_y = nullptr, *acc_z = nullptr;
  float *acc_old_x = nullptr, *acc_old_y = nullptr, *acc_old_z = nullptr;
  float* mass = nullptr;
  size_t count = 0;
};

struct SimulationConfig {
  size_t particle_count = 10000;
  InitDistribution init_distribution = InitDistribution::SPHERICAL;
  ForceMethod force_method = ForceMethod::DIRECT_N2;
  float dt = 0.001f;
  float G = 1.0f;
  float softening = 0.1f;
  float barnes_hut_theta = 0.5f;
  float spatial_hash_cell_size = 1.0f;
  float spatial_hash_cutoff = 2.0f;
  int cuda_block_size = 256;
};

struct UniformDistParams { Vec3 min_bounds, max_bounds; float min_mass = 1.0f, max_mass = 1.0f; };
struct SphericalDistParams { Vec3 center; float radius = 10.0f, min_mass = 1.0f, max_mass = 1.0f; };
struct DiskDistParams {
  Vec3 center; float radius = 10.0f, thickness = 1.0f, min_mass = 1.0f, max_mass = 1.0f,
      rotation_speed = 1.0f;
};

// ---- exceptions (same names, what() formats and accessors as the reference) ----------------
class CudaException : public std::runtime_error {
public:
  CudaException(const char* msg, const char* file, int line)
      : std::runtime_error(format(msg, file, line)), error_msg_(msg), file_(file), line_(line) {}
  const char* getErrorMsg() const { return error_msg_; }
  const char* getFile() const { return file_; }
  int getLine() const { return line_; }
private:
  static std::string format(const char* msg, const char* file, int line) {
    std::ostringstream o; o << "CUDA Error: " << msg << " at " << file << ":" << line; return o.str();
  }
  const char* error_msg_; const char* file_; int line_;
};
class ResourceException : public std::runtime_error {
public:
  ResourceException(const char* msg, size_t required, size_t available)
      : std::runtime_error(format(msg, required, available)), required_(required), available_(available) {}
  size_t getRequired() const { return required_; }
  size_t getAvailable() const { return available_; }
private:
  static std::string format(const char* msg, size_t r, size_t a) {
    std::ostringstream o; o << msg << " - Required: " << r << " bytes, Available: " << a << " bytes"; return o.str();
  }
  size_t required_, available_;
};
class ValidationException : public std::runtime_error {
public:
  explicit ValidationException(const std::string& msg) : std::runtime_error("Validation Error: " + msg) {}
};

// ---- device structures ---------------------------------------------------------------------
struct OctreeNode {  // 76 bytes
  Vec3 center; float half_size; Vec3 center_of_mass; float total_mass;
  int children[8]; int particle_index; bool is_leaf; int particle_count;
  OctreeNode() : half_size(0), total_mass(0), particle_index(-1), is_leaf(true), particle_count(0) {
    for (int& c : children) c = -1;
  }
};

class BarnesHutTree {
public:
  BarnesHutTree(size_t max_particles);
  ~BarnesHutTree();
  void build(const ParticleData* d_particles);
  // (facade only, no data member: the drift of a Velocity-Verlet step + build in one pass over the bodies,
  // nbody_hip_tree_drift_build; Integrator::integrate uses it for exactly the engine's own calculator)
  void driftBuild(ParticleData* d_particles, float dt);
  void computeForces(ParticleData* d_particles, float theta, float G, float eps);
  int getNodeCount() const { return node_count_; }
  int getMaxDepth() const { return max_depth_; }
  const OctreeNode* getNodes() const { return h_nodes_.data(); }
  void copyNodesToHost();
  bool verifyTreeStructure() const;
  bool verifyMassConservation(const ParticleData* h_particles) const;
private:
  // EXACTLY the reference's data members (barnes_hut_tree.hpp:57-74), same order and types:
  // sizeof == 96 under both header sets, so a caller compiled against the reference's header
  // (its tests build these objects on the stack) reserves what this constructor writes.
  // The engine's handle lives in the d_nodes_ slot (the device node array belongs to the
  // handle); the other two device pointers stay null.  Checked by oracle/layout_probe.cpp.
  OctreeNode* d_nodes_ = nullptr;
  int* d_sorted_indices_ = nullptr;
  unsigned int* d_morton_codes_ = nullptr;
  std::vector<OctreeNode> h_nodes_;
  size_t max_particles_;
  size_t max_nodes_ = 0;
  int node_count_ = 0;
  int max_depth_ = 0;
  Vec3 bbox_min_, bbox_max_;
  nbody_hip_tree* handle() const { return reinterpret_cast<nbody_hip_tree*>(d_nodes_); }
};

class SpatialHashGrid {
public:
  SpatialHashGrid(size_t max_particles, float cell_size = 1.0f);
  ~SpatialHashGrid();
  void build(const ParticleData* d_particles);
  void driftBuild(ParticleData* d_particles, float dt);  // facade only, see BarnesHutTree::driftBuild
  void computeForces(ParticleData* d_particles, float cutoff, float G, float eps);
  int3 getGridDims() const { return grid_dims_; }
  float getCellSize() const { return cell_size_; }
  int getTotalCells() const { return total_cells_; }
  void copyCellDataToHost(std::vector<int>& cell_start, std::vector<int>& cell_end,
                          std::vector<int>& particle_cells, std::vector<int>& sorted_indices);
  bool verifyCellAssignment(const ParticleData* h_particles) const;
  static int3 getCellIndex(float x, float y, float z, float cell_size);
  static int hashCell(int3 cell, int3 grid_dims);
private:
  int *d_cell_start_ = nullptr, *d_cell_end_ = nullptr, *d_particle_cell_ = nullptr,
      *d_sorted_indices_ = nullptr, *d_cell_counts_ = nullptr;
  size_t max_particles_;
  float cell_size_;
  int3 grid_dims_{0, 0, 0};
  int total_cells_ = 0;
  Vec3 bbox_min_, bbox_max_;
  // the reference's data members only (spatial_hash_grid.hpp:36-52; sizeof == 96 under both
  // header sets).  The engine's handle lives in the d_cell_start_ slot; the body count of the
  // last build is asked from the handle.
  nbody_hip_grid* handle() const { return reinterpret_cast<nbody_hip_grid*>(d_cell_start_); }
};

// ---- the strategy interface (the plugin boundary) ---------------------------------------------
class ForceCalculator {
public:
  virtual ~ForceCalculator() = default;
  virtual void computeForces(ParticleData* d_particles) = 0;
  virtual ForceMethod getMethod() const = 0;
  void setSofteningParameter(float eps) { softening_eps_ = eps; softening_eps2_ = eps * eps; }
  void setGravitationalConstant(float G) { G_ = G; }
  float getSofteningParameter() const noexcept { return softening_eps_; }
  float getGravitationalConstant() const noexcept { return G_; }
protected:
  float softening_eps_ = 0.01f;
  float softening_eps2_ = 0.0001f;
  float G_ = 1.0f;
};

class DirectForceCalculator : public ForceCalculator {
public:
  explicit DirectForceCalculator(int block_size = 256);
  void computeForces(ParticleData* d_particles) override;
  ForceMethod getMethod() const noexcept override { return ForceMethod::DIRECT_N2; }
  void setBlockSize(int size) { block_size_ = size; }
  int getBlockSize() const noexcept { return block_size_; }
private:
  int block_size_;
};

// MI355X-native addition (no reference counterpart; the reference is single-GPU): Direct N^2 shared by `ndev`
// GPUs of one node behind the SAME plugin interface -- computeForces(ParticleData*) on a whole-system
// ParticleData that lives on device 0; positions go out and accelerations come back as peer copies over
// xGMI (include/nbody_hip_comm.h: nbody_hip_comm_init_all + nbody_hip_sharded_direct_compute_forces).
// Drops into Integrator::integrate / ParticleSystem like any user-defined ForceCalculator.
class ShardedDirectCalculator : public ForceCalculator {
public:
  // devices: the GPUs to share the work (empty = all visible devices); rccl: RCCL collectives instead of peer copies
  explicit ShardedDirectCalculator(std::vector<int> devices = {}, bool rccl = false);
  ~ShardedDirectCalculator() override;
  void computeForces(ParticleData* d_particles) override;
  ForceMethod getMethod() const noexcept override { return ForceMethod::DIRECT_N2; }
  int getDeviceCount() const noexcept { return static_cast<int>(devices_.size()); }
private:
  std::vector<int> devices_;
  bool rccl_;
  ::nbody_hip_comm* comm_ = nullptr;
  ::nbody_hip_sharded_direct* sys_ = nullptr;
  size_t count_ = 0;
  float built_G_ = 0.f, built_eps_ = -1.f;
};

class BarnesHutCalculator : public ForceCalculator {
public:
  explicit BarnesHutCalculator(float theta = 0.5f);
  ~BarnesHutCalculator();
  void computeForces(ParticleData* d_particles) override;
  ForceMethod getMethod() const noexcept override { return ForceMethod::BARNES_HUT; }
  void setTheta(float theta) { theta_ = theta; }
  float getTheta() const noexcept { return theta_; }
  BarnesHutTree* getTree() noexcept { return tree_.get(); }
private:
  std::unique_ptr<BarnesHutTree> tree_;
  float theta_;
};

class SpatialHashCalculator : public ForceCalculator {
public:
  SpatialHashCalculator(float cell_size = 1.0f, float cutoff_radius = 2.0f);
  ~SpatialHashCalculator();
  void computeForces(ParticleData* d_particles) override;
  ForceMethod getMethod() const noexcept override { return ForceMethod::SPATIAL_HASH; }
  void setCellSize(float size) { cell_size_ = size; }
  void setCutoffRadius(float radius) { cutoff_radius_ = radius; }
  float getCellSize() const noexcept { return cell_size_; }
  float getCutoffRadius() const noexcept { return cutoff_radius_; }
  SpatialHashGrid* getGrid() noexcept { return grid_.get(); }
private:
  std::unique_ptr<SpatialHashGrid> grid_;
  float cell_size_;
  float cutoff_radius_;
};

std::unique_ptr<ForceCalculator> createForceCalculator(ForceMethod method, const SimulationConfig& config);
Vec3 computeGravitationalForceCPU(const Vec3& p1, const Vec3& p2, float m1, float m2, float G, float eps);

class Integrator {
public:
  explicit Integrator(int block_size = 256);
  ~Integrator();
  void integrate(ParticleData* d_particles, ForceCalculator* force_calc, float dt);
  void updatePositions(ParticleData* d_particles, float dt);
  void updateVelocities(ParticleData* d_particles, float dt);
  void storeOldAccelerations(ParticleData* d_particles);
  float computeKineticEnergy(const ParticleData* d_particles);
  float computePotentialEnergy(const ParticleData* d_particles, float G, float eps);
  float computeTotalEnergy(const ParticleData* d_particles, float G, float eps);
  void ensureScratchBuffer(size_t particle_count);
  void setBlockSize(int size) { block_size_ = size; }
  int getBlockSize() const noexcept { return block_size_; }
private:
  int block_size_;
  float* d_scratch_ = nullptr;  // unused here: reductions use the context's workspace
  int scratch_blocks_ = 0;
};

class ParticleDataManager {
public:
  static void allocateDevice(ParticleData& data, size_t count);
  static void freeDevice(ParticleData& data);
  static void allocateHost(ParticleData& data, size_t count);
  static void freeHost(ParticleData& data);
  static void copyToDevice(ParticleData& d_data, const ParticleData& h_data);
  static void copyToHost(ParticleData& h_data, const ParticleData& d_data);
  static void copyPositionsToHost(float* h_pos_x, float* h_pos_y, float* h_pos_z, const ParticleData& d_data);
  static void copyPositionsToDevice(ParticleData& d_data, const float* h_pos_x, const float* h_pos_y,
                                    const float* h_pos_z);
};

class ParticleInitializer {
public:
  static void initUniform(ParticleData& h_data, const UniformDistParams& params, unsigned int seed = 42);
  static void initSpherical(ParticleData& h_data, const SphericalDistParams& params, unsigned int seed = 42);
  static void initDisk(ParticleData& h_data, const DiskDistParams& params, unsigned int seed = 42);
  static void zeroVelocities(ParticleData& h_data);
  static void zeroAccelerations(ParticleData& h_data);
private:
  static std::mt19937 createRNG(unsigned int seed);
};

// launch wrappers the reference declares as free functions (integrator.hpp:153-158)
void launchUpdatePositionsKernel(ParticleData* d_particles, float dt, int block_size);
void launchUpdateVelocitiesKernel(ParticleData* d_particles, float dt, int block_size);
void launchStoreAccelerationsKernel(ParticleData* d_particles, int block_size);
float launchComputeKineticEnergyKernel(const ParticleData* d_particles, int block_size);
float launchComputePotentialEnergyKernel(const ParticleData* d_particles, float G, float eps, int block_size);
void launchDirectForceKernel(ParticleData* d_particles, float G, float eps2, int block_size);

// the process-wide HIP context the facade launches on (device 0, null stream, like the reference)
nbody_hip_ctx* facadeContext();

}  // namespace nbody
