/*
 * nbody_hip.h -- C ABI of libnbody_hip.so, the MI355X (gfx950) force/integration engine.
 *
 * This is the drop-in boundary for the hot path of LessUp/n-body: every entry point
 * replaces one CUDA launch wrapper / class method of the reference (cited per
 * function as `ref: file:line`, paths relative to the reference checkout).  The
 * reference itself has no C ABI; its plugin API is the C++ strategy interface
 * `nbody::ForceCalculator` (include/nbody/force_calculator.hpp:36-89).  The C++
 * facade in n-body_amd/facade/ re-creates those classes on top of this header;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; every `float*` inside nbody_particle_data is a
 *    DEVICE pointer (hipMalloc'd, caller-owned) unless the function says "host".
 *  - every call returns 0 on success or a negative nbody_hip_status; the message is
 *    available from nbody_hip_last_error() (thread-local), formatted like the
 *    reference's CudaException ("<what> at file:line", error_handling.hpp:29-51).
 *  - calls are asynchronous on the context's HIP stream unless they return a value
 *    to the host (energies) or say "blocking".  Not thread-safe per context, like
 *    the reference (force_calculator.hpp:8-19).
 *  - no CPU fallback exists: without a HIP device every entry point fails with
 *    NBODY_HIP_ERR_DEVICE.
 */
#ifndef NBODY_HIP_H
#define NBODY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBODY_HIP_ABI_VERSION 1

/* exported with default visibility; everything else in the library is hidden */
#if defined(__GNUC__)
#define NBODY_HIP_API __attribute__((visibility("default")))
#else
#define NBODY_HIP_API
#endif

typedef enum nbody_hip_status {
  NBODY_HIP_OK = 0,
  NBODY_HIP_ERR_VALIDATION = -1, /* bad argument (ref: ValidationException) */
  NBODY_HIP_ERR_DEVICE = -2,     /* HIP runtime error (ref: CudaException) */
  NBODY_HIP_ERR_RESOURCE = -3,   /* out of memory / grid too large (ref: ResourceException) */
  NBODY_HIP_ERR_STATE = -4       /* call sequence error (NULL handle, not initialised) */
} nbody_hip_status;

/* ref: ForceMethod, include/nbody/types.hpp:66-70 (same enumerator order) */
typedef enum nbody_hip_force_method {
  NBODY_HIP_DIRECT_N2 = 0,
  NBODY_HIP_BARNES_HUT = 1,
  NBODY_HIP_SPATIAL_HASH = 2
} nbody_hip_force_method;

/*
 * ref: nbody::ParticleData, include/nbody/types.hpp:234-276.  Layout-identical
 * (13 float* then size_t count, 112 bytes), so a `nbody::ParticleData*` can be
 * passed straight through the boundary with a reinterpret_cast.
 */
typedef struct nbody_particle_data {
  float* pos_x; float* pos_y; float* pos_z;
  float* vel_x; float* vel_y; float* vel_z;
  float* acc_x; float* acc_y; float* acc_z;
  float* acc_old_x; float* acc_old_y; float* acc_old_z;
  float* mass;
  size_t count;
} nbody_particle_data;

/* 16-byte packed body used by the native (sharded / multi-GPU) entry points:
 * {x, y, z, mass}.  Accelerations use the same type with w unused. */
typedef struct nbody_float4 { float x, y, z, w; } nbody_float4;

typedef struct nbody_hip_ctx nbody_hip_ctx;

/* ---- library / context -------------------------------------------------- */

NBODY_HIP_API int nbody_hip_abi_version(void);
/* Which radix sort bins the bodies (Barnes-Hut build, spatial-hash build; the reference calls thrust::sort_by_key,
 * ref: src/cuda/force_barnes_hut.cu:276-280, force_spatial_hash.cu:286-288).  Above the crossover sizes one of
 *   - the library's driver of rocPRIM's Onesweep kernels (csrc/onesweep.h; compiled only for the rocPRIM version it was
 *     written against: driver_compiled), the default while its run-time self-test holds (self_test: 0 = not run yet --
 *     no tree / grid made --, 1 = it reproduced the public sort word for word in this process, 2 = it did not: off);
 *   - the hand-written sort of csrc/radix_sort.h (no rocPRIM), which runs when the driver is absent or off, or with
 *     NBH_SORT=own in the environment (own_self_test: the same three states);
 * below them, with NBH_SORT=public, or when both are off: rocprim::radix_sort_pairs.  rocprim_version: ROCPRIM_VERSION of
 * the build.  Any pointer may be NULL. */
NBODY_HIP_API int nbody_hip_sort_info(int* driver_compiled, int* self_test, int* own_self_test, int* rocprim_version);
NBODY_HIP_API const char* nbody_hip_last_error(void);

/* Number of HIP devices visible (0 when there is no GPU; never fails). */
NBODY_HIP_API int nbody_hip_device_count(void);

/* Creates a context on `device`.  `stream` is a hipStream_t the caller owns; NULL means HIP's
 * default (null) stream, which is what the reference launches on (force_direct.cu:93).  The
 * context never creates streams of its own. */
NBODY_HIP_API int nbody_hip_ctx_create(nbody_hip_ctx** out, int device, void* stream);
NBODY_HIP_API int nbody_hip_ctx_destroy(nbody_hip_ctx* ctx);
/* Re-targets later launches to another caller-owned stream (NULL = the null stream). */
NBODY_HIP_API int nbody_hip_ctx_set_stream(nbody_hip_ctx* ctx, void* stream);
/* Blocks until all work queued on the context's stream is done.
 * ref: CUDA_CHECK_KERNEL's debug-mode cudaDeviceSynchronize, error_handling.hpp:124-136 */
NBODY_HIP_API int nbody_hip_ctx_synchronize(nbody_hip_ctx* ctx);

/* ---- step graphs (MI355X-native addition; no reference counterpart) ------------------------
 * Between capture_begin and capture_end every call made with this context is recorded into a
 * hipGraph instead of being executed; graph_launch replays the recording `times` times on the
 * context's stream (one host call for a whole run of steps).  Rules: run the same calls once
 * eagerly first (workspaces are sized on first use and allocation is not capturable); the pointers
 * and body count are baked into the recording; calls that return data to the host (energies,
 * *_info, *_stats, download, the spatial-hash grid build) are not capturable and fail with
 * NBODY_HIP_ERR_STATE, after which capture_end reports the failure and restores the context.
 * Measured (profiles/r01_step_graph_probe.txt): replay == eager time per step on MI355X, the
 * inter-kernel gaps being GPU-side; the gain is host CPU time, not step rate. */
typedef struct nbody_hip_graph nbody_hip_graph;
NBODY_HIP_API int nbody_hip_capture_begin(nbody_hip_ctx* ctx);
NBODY_HIP_API int nbody_hip_capture_end(nbody_hip_ctx* ctx, nbody_hip_graph** out);
NBODY_HIP_API int nbody_hip_graph_launch(nbody_hip_graph* graph, int times);
NBODY_HIP_API int nbody_hip_graph_destroy(nbody_hip_graph* graph);

/* ---- a11: particle memory (ref: ParticleDataManager, src/cuda/particle_init.cu:143-283) */

/* ref: allocateDevice :143-168 -- 13 arrays of `count` floats, accelerations zeroed.
 * One slab allocation, 256-byte aligned sub-arrays; freed by nbody_hip_particles_free. */
NBODY_HIP_API int nbody_hip_particles_alloc(nbody_particle_data* d, size_t count);
/* ref: freeDevice :170-198 -- frees and nulls every pointer, count = 0. */
NBODY_HIP_API int nbody_hip_particles_free(nbody_particle_data* d);
/* ref: copyToDevice :257-269 -- host -> device of pos, vel, acc, mass (10 arrays; acc_old is
 * NOT copied, as in the reference).  Blocking. */
NBODY_HIP_API int nbody_hip_particles_upload(nbody_particle_data* d, const nbody_particle_data* h);
/* ref: copyToHost :271-283 -- device -> host of the same 10 arrays.  Blocking. */
NBODY_HIP_API int nbody_hip_particles_download(nbody_particle_data* h, const nbody_particle_data* d);

/* ---- a4: Direct N^2 (ref: src/cuda/force_direct.cu:10-106) ---------------- */

/* ref: launchDirectForceKernel(ParticleData*, G, eps2, block_size) :88-98.
 * Reads pos_*, mass; OVERWRITES acc_* with a_i = G * sum_{j != i} m_j r_ij (|r_ij|^2+eps2)^-3/2.
 * `block_size` is accepted for signature parity and validated (1..1024) but does not pick the
 * launch shape: tiling is chosen for the 256-CU part from `count`. */
NBODY_HIP_API int nbody_hip_direct_forces(nbody_hip_ctx* ctx, const nbody_particle_data* d, float G, float eps2,
                            int block_size);

/* Native form used by the sharded (multi-GPU) path: accelerations of `n_targets` packed
 * bodies due to `n_sources` packed bodies (targets may be a sub-range of the sources; a
 * coincident pair contributes exactly zero, which is how the self pair is skipped).
 * acc_out[i] = {ax, ay, az, 0}.  If `accumulate` is non-zero the result is ADDED to acc_out
 * (used to overlap the local-shard pass with the all-gather of the remote shards). */
NBODY_HIP_API int nbody_hip_direct_forces_packed(nbody_hip_ctx* ctx, const nbody_float4* targets,
                                   size_t n_targets, const nbody_float4* sources,
                                   size_t n_sources, nbody_float4* acc_out, float G, float eps2,
                                   int accumulate);

/* Two DISJOINT body sets a and b, every a x b pair evaluated ONCE (Newton's third law):
 * acc_a (+)= forces on a from b, acc_b (+)= forces on b from a.  The sharded path uses it so that
 * a pair of shards is computed by one rank only; the reactions travel back in a reduce-scatter.
 * Needs eps2 >= 1e-12. */
NBODY_HIP_API int nbody_hip_direct_forces_pair_packed(nbody_hip_ctx* ctx, const nbody_float4* a, size_t n_a,
                                                      const nbody_float4* b, size_t n_b, nbody_float4* acc_a,
                                                      int accumulate_a, nbody_float4* acc_b, int accumulate_b,
                                                      float G, float eps2);

/* SoA <-> packed conversion on device (x,y,z,mass -> float4 and back). */
NBODY_HIP_API int nbody_hip_pack_posm(nbody_hip_ctx* ctx, const float* x, const float* y, const float* z,
                        const float* mass, size_t count, nbody_float4* out);
NBODY_HIP_API int nbody_hip_unpack3(nbody_hip_ctx* ctx, const nbody_float4* in, size_t count, float* x,
                      float* y, float* z);

/* ---- a6: Velocity Verlet (ref: src/cuda/integrator.cu:11-48,122-152,224-238) */

/* ref: launchUpdatePositionsKernel :122-131   x += v dt + a (dt^2/2) */
NBODY_HIP_API int nbody_hip_update_positions(nbody_hip_ctx* ctx, nbody_particle_data* d, float dt);
/* ref: launchUpdateVelocitiesKernel :133-142  v += (a_old + a) (dt/2) */
NBODY_HIP_API int nbody_hip_update_velocities(nbody_hip_ctx* ctx, nbody_particle_data* d, float dt);
/* ref: launchStoreAccelerationsKernel :144-152  a_old = a */
NBODY_HIP_API int nbody_hip_store_accelerations(nbody_hip_ctx* ctx, nbody_particle_data* d);
/* storeOldAccelerations + updatePositions in one pass (what Integrator::integrate does first,
 * integrator.cu:224-231): a_old <- a ; x += v dt + a dt^2/2.  Same arithmetic, one launch instead of four. */
NBODY_HIP_API int nbody_hip_drift(nbody_hip_ctx* ctx, nbody_particle_data* d, float dt);
/* ref: Integrator::integrate with a DirectForceCalculator :224-238, `steps` times:
 * one fused drift pass (a_old<-a, x update, float4 pack), the force kernel, and the
 * velocity update fused into the force reduction epilogue. */
NBODY_HIP_API int nbody_hip_integrate_direct(nbody_hip_ctx* ctx, nbody_particle_data* d, float G, float eps2,
                               float dt, int steps);

/* Packed (float4) halves of the same step for the sharded multi-GPU path, where each rank
 * keeps only its target range: drift = x += v dt + a dt^2/2 on {x,y,z,m}; kick =
 * v += (a_old + a_new) dt/2.  The caller swaps its two acceleration buffers between steps
 * (no a_old copy).  ref: integrator.cu:16-19 and :31-34. */
NBODY_HIP_API int nbody_hip_drift_packed(nbody_hip_ctx* ctx, nbody_float4* posm, const nbody_float4* vel,
                                         const nbody_float4* acc, size_t count, float dt);
NBODY_HIP_API int nbody_hip_kick_packed(nbody_hip_ctx* ctx, nbody_float4* vel, const nbody_float4* acc_old,
                                        const nbody_float4* acc_new, size_t count, float dt);

/* ---- a7: energies (ref: src/cuda/integrator.cu:51-119,252-293) ------------ */

/* ref: Integrator::computeKineticEnergy :252-269 -- 0.5 sum m v^2.  fp64 reduction on device,
 * result rounded to float like the reference's return type.  Blocking. */
NBODY_HIP_API int nbody_hip_kinetic_energy(nbody_hip_ctx* ctx, const nbody_particle_data* d, float* out);
/* ref: Integrator::computePotentialEnergy :271-289 -- -G sum_{i<j} m_i m_j / sqrt(r^2+eps^2).
 * Takes eps (not eps^2) like the reference.  Blocking. */
NBODY_HIP_API int nbody_hip_potential_energy(nbody_hip_ctx* ctx, const nbody_particle_data* d, float G,
                               float eps, float* out);
/* fp64 variants of the two above (no final rounding to float). */
NBODY_HIP_API int nbody_hip_kinetic_energy_f64(nbody_hip_ctx* ctx, const nbody_particle_data* d, double* out);
NBODY_HIP_API int nbody_hip_potential_energy_f64(nbody_hip_ctx* ctx, const nbody_particle_data* d, float G,
                                   float eps, double* out);

/* e (SURVEY 8e "KE/PE via all-reduce of one fp64"): energies of ONE SHARD of a sharded run.
 * out[0] = sum over the targets of 0.5 m v^2; out[1] = -G/2 sum_{i in targets} m_i sum_{j in sources,
 * j != self_offset + i} m_j / sqrt(r_ij^2 + eps^2): target i is source number self_offset + i (for
 * disjoint sets pass self_offset >= n_sources or <= -n_targets, so that no target maps to a source).  Summed over a partition of the
 * bodies (each rank: its shard against the gathered bodies) the two numbers are the system's KE
 * and PE as nbody_hip_*_energy_f64 computes them.  Same per-term fp32 arithmetic, fp64 sums. */
NBODY_HIP_API int nbody_hip_energies_packed(nbody_hip_ctx* ctx, const nbody_float4* targets_posm,
                                            const nbody_float4* targets_vel, size_t n_targets,
                                            long long self_offset, const nbody_float4* sources_posm,
                                            size_t n_sources, float G, float eps, double out[2]);

/* ---- a9: spatial hash (ref: SpatialHashGrid / SpatialHashCalculator,
 *          src/cuda/force_spatial_hash.cu:14-377, include/nbody/spatial_hash_grid.hpp:9-59) ---- */

typedef struct nbody_hip_grid nbody_hip_grid;

/* ref: SpatialHashGrid(max_particles, cell_size) :155-168.  Sized once from `max_particles`
 * (the reference sizes its grid from the first count it sees, :372-374). */
NBODY_HIP_API int nbody_hip_grid_create(nbody_hip_ctx* ctx, size_t max_particles, float cell_size,
                                        nbody_hip_grid** out);
NBODY_HIP_API int nbody_hip_grid_destroy(nbody_hip_grid* grid);
/* ref: SpatialHashCalculator::setCellSize (force_calculator.hpp:199): takes effect at the next build */
NBODY_HIP_API int nbody_hip_grid_set_cell_size(nbody_hip_grid* grid, float cell_size);
/* Tuning hook for measurements: force kernel 0 = automatic, 1 = cell-run kernel (one workgroup per
 * run of cells along x, binary-searched ranges; the only one for very sparse grids), 2 / 3 / 4 = wave-per-
 * cell kernel with 1 / 2 / 4 bodies per lane (needs a grid of at most ~4 cells per body).  All give the
 * reference's 27-cell result; they differ by fp rounding of the summation order only.
 * 6 = two bodies per lane with the window filtered by the box of the cell's bodies (automatic from 8 bodies per cell
 *     when cutoff <= cell_size, from 40 when cutoff > cell_size);
 * 7 = two-phase form of the wave-per-cell kernel (distance masks first; measured, not the default);
 * 8 = one lane per body (automatic below 8 bodies per cell), 9 = its split form (8 for the cells of few bodies, 3 for
 *     the crowded ones: automatic from 500,000 bodies), 10 = two bodies of one cell per lane (measured: 7-17 % faster
 *     than 8 between 3 and 9 bodies per cell, bound by the L1 path; not chosen automatically);
 * 5 = TIMING PROBE, not a force kernel: the wave-per-cell kernel over the half shell (own cell + 13 forward
 * neighbours) without reactions -- a lower bound for the time of a Newton's-third-law variant;
 * its output is meaningless. */
NBODY_HIP_API int nbody_hip_grid_tuning(nbody_hip_grid* grid, int kernel);
/* ref: SpatialHashGrid::build :235-303 -- bounding box (padded 0.001), grid dims
 * ceil(extent/cell)+1, cell id per body, bodies ordered by cell.  More than 1e8 cells ->
 * NBODY_HIP_ERR_RESOURCE ("Spatial hash grid too large", :252-254).  One host round trip
 * (the grid size), like the reference. */
NBODY_HIP_API int nbody_hip_grid_build(nbody_hip_grid* grid, const nbody_particle_data* d);
/* nbody_hip_drift + nbody_hip_grid_build in one pass over the bodies (see nbody_hip_tree_drift_build); used by
 * Integrator::integrate for exactly the engine's own SpatialHashCalculator. */
NBODY_HIP_API int nbody_hip_grid_drift_build(nbody_hip_grid* grid, nbody_particle_data* d, float dt);
/* ref: SpatialHashGrid::computeForces(d_particles, cutoff, G, eps) :305-316 -- takes eps and
 * cutoff unsquared like the reference; OVERWRITES acc_*. */
NBODY_HIP_API int nbody_hip_grid_compute_forces(nbody_hip_grid* grid, nbody_particle_data* d,
                                                float cutoff, float G, float eps);
/* ref: getGridDims / getTotalCells (spatial_hash_grid.hpp:20-24) + the padded bounding box. */
NBODY_HIP_API int nbody_hip_grid_info(const nbody_hip_grid* grid, int dims[3], int* total_cells,
                                      float bbox_min[3], float bbox_max[3]);
/* Number of bodies of the last build (0 before the first): the length of the per-body arrays
 * copy_cell_data writes (the reference keeps it as the ParticleData count it was built from). */
NBODY_HIP_API int nbody_hip_grid_count(const nbody_hip_grid* grid, size_t* built_count);
/* ref: copyCellDataToHost :318-331 -- HOST output arrays (any may be NULL): cell_start/cell_end
 * [total_cells] (0/0 for an empty cell), particle_cells [count], sorted_indices [count].
 * Blocking. */
NBODY_HIP_API int nbody_hip_grid_copy_cell_data(nbody_hip_grid* grid, int* cell_start, int* cell_end,
                                                int* particle_cells, int* sorted_indices);

/* Packed forms used by the sharded (multi-GPU, z-slab) spatial-hash path.  `bounds` = {lo x,y,z,
 * hi x,y,z} of the ALREADY PADDED global box (NULL: box of these bodies padded by 0.001), so that
 * every rank bins on the same grid; acc_out[i] = {ax, ay, az, 0} for every body of the last build,
 * in its input order (a rank passes [own bodies; halo bodies] and keeps the first part). */
NBODY_HIP_API int nbody_hip_grid_build_packed(nbody_hip_grid* grid, const nbody_float4* posm, size_t count,
                                              const float* bounds);
NBODY_HIP_API int nbody_hip_grid_compute_forces_packed(nbody_hip_grid* grid, float cutoff, float G, float eps,
                                                       nbody_float4* acc_out);
/* Bodies [first, first + count) of the last build in CELL ORDER (x fastest, then y, then z: a z layer
 * is one contiguous run), copied to a DEVICE array.  The sharded step takes a rank's lowest and highest
 * layer from here: they are the halo its neighbours need.  Asynchronous. */
NBODY_HIP_API int nbody_hip_grid_sorted_bodies(nbody_hip_grid* grid, size_t first, size_t count, nbody_float4* out);
/* e3 (multi-GPU spatial hash; no reference counterpart): the z layers [z_first, z_first + z_count) of the
 * global grid are the only ones later nbody_hip_grid_build_packed calls WITH explicit bounds put bodies
 * in (a rank's slab, or its slab plus the two halo layers): the per-cell start array of the force kernel
 * then covers just those layers.  z_count <= 0 (default): the whole grid. */
NBODY_HIP_API int nbody_hip_grid_set_slab(nbody_hip_grid* grid, int z_first, int z_count);
/* Forces on the bodies of `targets` that lie in the layers [z_first, z_first + z_count) (z_count <= 0:
 * all) from the bodies of `sources` -- another grid built on the SAME global box and cell size, or the
 * same grid.  acc_out has one row per body of `targets` in its input order; only the rows of the bodies
 * in those layers are written (accumulate != 0: added to).  The sharded step evaluates own x own while
 * the halo layers are in flight, then boundary layers x halo with accumulate.  Both grids must be dense
 * enough to carry a per-cell start array (NBODY_HIP_ERR_STATE otherwise: use the one-grid path). */
NBODY_HIP_API int nbody_hip_grid_forces_pair_packed(nbody_hip_grid* targets, nbody_hip_grid* sources, int z_first,
                                                    int z_count, float cutoff, float G, float eps,
                                                    nbody_float4* acc_out, int accumulate);

/* The boundary pass of a z-slab decomposition without a second grid (round 4).  A rank's own grid (packed build with
 * nbody_hip_grid_set_slab) can hand one of its z layers to a neighbour READY FOR USE: nbody_hip_grid_export_layer writes
 * the layer's bodies in cell order (as many as the layer holds) and its start array rebased to 0 (dims[0] * dims[1] + 1
 * ints).  nbody_hip_grid_forces_layer_packed evaluates the targets of layer z of `grid` against such a layer (which is
 * layer src_z of the same global grid: same origin, dimensions, cell size), adding to acc_out when accumulate != 0.
 * Same pair set and arithmetic as the 27-cell search of ref: src/cuda/force_spatial_hash.cu:104-146 restricted to the
 * source layer's cells. */
NBODY_HIP_API int nbody_hip_grid_export_layer(nbody_hip_grid* grid, int z, nbody_float4* bodies_out, int* lb_out);
NBODY_HIP_API int nbody_hip_grid_forces_layer_packed(nbody_hip_grid* grid, int z, const nbody_float4* src_bodies,
                                                     const int* src_lb, int src_z, float cutoff, float G, float eps,
                                                     nbody_float4* acc_out, int accumulate);
/* One partition pass of a rank's bodies after the drift (csrc/slab.hip).  gbox_dev: DEVICE array
 * {min x,y,z, max x,y,z} of ALL ranks' bodies (the all-reduced nbody_hip_bbox_packed result, unpadded);
 * the global grid is derived from it on the device exactly as SpatialHashGrid::build does (ref:
 * force_spatial_hash.cu:225-246).  Layer z belongs to rank ((z + 1) world - 1) / gz.  Outputs, all DEVICE:
 *   rows_out        the bodies that CHANGE OWNER, 16 floats each {x,y,z,m | vx,vy,vz,0 | ax,ay,az,0 |
 *                   id,z,0,0 (int bits)}, grouped by new owner (ascending, own rank absent), each group in
 *                   input order; capacity n rows
 *   holes_out       the input positions of those bodies, ascending; capacity n ints
 *   send_matrix_dev world x world ints, zeroed, row `rank` = bodies per new owner, entry [rank][rank] = the
 *                   bodies that stay (sum over ranks = who sends how many rows to whom)
 *   hist_dev        hist_cap ints: this rank's bodies per layer (sum over ranks = population of every
 *                   layer, hence every rank's new body count and the size of its halo layers)
 *   info_dev        4 ints: gx, gy, gz, and 1 if gz > hist_cap (the caller must fall back)
 * The bodies that stay are not touched.  Asynchronous; gid may be NULL (ids = input positions); with
 * world == 1 rows_out / holes_out may be NULL. */
NBODY_HIP_API int nbody_hip_slab_partition(nbody_hip_ctx* ctx, const nbody_float4* posm, const nbody_float4* vel,
                                           const nbody_float4* acc, const int* gid, size_t n, const float* gbox_dev,
                                           float cell_size, int world, int rank, int hist_cap, float* rows_out,
                                           int* holes_out, int* send_matrix_dev, int* hist_dev, int* info_dev);
/* The same with the ranks' slabs bounded by PHYSICAL z coordinates: z_cuts = world - 1 ascending HOST floats (NULL:
 * equal layer counts, as above); the owner of a layer is the number of cuts at or below the layer's centre,
 * nbody_hip_slab_layer_owner -- the same arithmetic on the device and on the host, so that every rank's host can
 * derive the ranks' layer ranges.  A host that balances the slabs by body count picks the cuts from the (all-reduced)
 * layer histogram of the previous evaluation. */
NBODY_HIP_API int nbody_hip_slab_partition_cuts(nbody_hip_ctx* ctx, const nbody_float4* posm, const nbody_float4* vel,
                                                const nbody_float4* acc, const int* gid, size_t count,
                                                const float* gbox_device, float cell_size, int world, int rank,
                                                int hist_cap, float* rows_out, int* holes_out, int* send_matrix_device,
                                                int* hist_device, int* info_device, const float* z_cuts);
NBODY_HIP_API int nbody_hip_slab_layer_owner(int layer, float lo_z, float cell_size, int world, const float* z_cuts);
/* After the exchange: n_old bodies (arrays with room for n_old - n_holes + n_arrivals) of which the slots
 * holes[0..n_holes) (ascending, as written by the partition) are vacant, and n_arrivals rows received from
 * the other ranks -> the n_old - n_holes + n_arrivals bodies of the rank, contiguous from slot 0: arrivals go
 * into the vacant slots in order, surplus arrivals behind the old end, surplus slots are closed with the bodies
 * taken from the end.  Deterministic; moves 64 bytes per migrating body and nothing else.  gid may be NULL. */
NBODY_HIP_API int nbody_hip_slab_fill(nbody_hip_ctx* ctx, const float* rows, size_t n_arrivals, const int* holes,
                                      size_t n_holes, size_t n_old, nbody_float4* posm, nbody_float4* vel,
                                      nbody_float4* acc, int* gid);
/* min/max of packed bodies into a DEVICE array of 6 floats {lo x,y,z, hi x,y,z} (async; the ranks
 * all-reduce it).  ref: computeBoundingBoxKernel, force_barnes_hut.cu:66-110 */
NBODY_HIP_API int nbody_hip_bbox_packed(nbody_hip_ctx* ctx, const nbody_float4* posm, size_t count,
                                        float* bounds_device);
/* The float4 drift of nbody_hip_drift_packed (x += v dt + a dt^2/2, integrator.cu:16-19) and the box of the NEW
 * positions in ONE pass.  enc_device: 6 words private to the caller that hold the empty box {0xffffffff x3, 0 x3}
 * before the FIRST call; every call leaves them re-armed.  Async. */
NBODY_HIP_API int nbody_hip_drift_bbox_packed(nbody_hip_ctx* ctx, nbody_float4* posm, const nbody_float4* vel,
                                              const nbody_float4* acc, size_t count, float dt,
                                              unsigned int* enc_device, float* bounds_device);
/* z cell coordinate (clamped to [0, gz-1]) of packed bodies on a grid with origin lo_z; DEVICE
 * int output.  ref: assignCellsKernel, force_spatial_hash.cu:28-49 */
NBODY_HIP_API int nbody_hip_cell_z_packed(nbody_hip_ctx* ctx, const nbody_float4* posm, size_t count,
                                          float lo_z, float cell_size, int gz, int* cz_device);

/* ---- a8: Barnes-Hut (ref: BarnesHutTree / BarnesHutCalculator,
 *          src/cuda/force_barnes_hut.cu:204-532, include/nbody/barnes_hut_tree.hpp:9-81) ---- */

typedef struct nbody_hip_tree nbody_hip_tree;

/* Tuning hook for measurements: number of replicas (power of two, <= 16) that share each wave's
 * walk when there are few bodies, and the number of ownership units per replica (a unit is a subtree
 * of at most n / (units K) bodies; default: n / 96 whatever K); 0 = automatic.
 * Results are deterministic for a given setting; different settings differ by fp rounding only. */
NBODY_HIP_API int nbody_hip_tree_tuning(nbody_hip_tree* tree, int replicas, int units_per_replica);

/* Form of the walk without replicas: 1 = plain (one sibling node per step), 2 = pair walk (two sibling nodes per
 * packed instruction, node records as pair blocks; same interaction lists, a sibling group's fp32 sum formed as
 * (even siblings) + (odd siblings)), its waves scheduled longest-first from the node visits the previous walk
 * recorded; 3 = pair walk in plain order; 0 = automatic (2).  Results do not depend on the schedule.
 * The pair walk needs the even-aligned node ids a build gives trees of >= 98,304 bodies (smaller trees are walked with
 * replicas of the plain walk and keep plain ids); to force it on a smaller tree set the form BEFORE the build -- asking
 * for it afterwards fails with NBODY_HIP_ERR_STATE. */
NBODY_HIP_API int nbody_hip_tree_walk_form(nbody_hip_tree* tree, int form);
/* Test / stress hook: cap the node arrays at `max_nodes` (0 = the bound on the node count; the arrays also stop
 * at 2^28 - 1 nodes, the width of a child link).  A tree that would need more nodes is cut where the numbering
 * passes the capacity: the nodes beyond do not exist and their parents are leaves of several bodies, which interact
 * body by body -- forces stay correct (closer to the direct sum than the full tree's), the walk gets slower. */
NBODY_HIP_API int nbody_hip_tree_limit_nodes(nbody_hip_tree* tree, int max_nodes);
/* Node-visit counting for nbody_hip_tree_stats (off by default: it costs a memset launch and an
 * atomic per wave in every walk). */
NBODY_HIP_API int nbody_hip_tree_count_visits(nbody_hip_tree* tree, int enable);

/* Diagnostics of the last walk made with visit counting on: out[p] (p = 0..64) = internal-node tests made
 * with p lanes of the wave taking part, out[65 + q] = with q lanes accepting the node's monopole.
 * Blocking. */
NBODY_HIP_API int nbody_hip_tree_visit_histogram(nbody_hip_tree* tree, unsigned long long out[130]);

/* ref: BarnesHutTree(max_particles) :204-210 */
NBODY_HIP_API int nbody_hip_tree_create(nbody_hip_ctx* ctx, size_t max_particles, nbody_hip_tree** out);
NBODY_HIP_API int nbody_hip_tree_destroy(nbody_hip_tree* tree);
/* Tree shape: levels below the root (1..21; default 20, the depth the reference's insertion loop stops
 * at, :363) and the largest body count a leaf may hold (default 1, as in the reference, where a leaf
 * holds one particle; at the deepest level a leaf keeps whatever falls into its cell).  Up to 10 levels
 * the bodies are ordered by the reference's 30-bit Morton code (:23-38) in 32-bit keys; deeper trees use
 * 63-bit keys (21 bits per axis, a refinement of the same grid).  Re-sizes the key, flag and node arrays;
 * the tree must be rebuilt afterwards. */
NBODY_HIP_API int nbody_hip_tree_set_params(nbody_hip_tree* tree, int max_depth, int leaf_max);
/* ref: BarnesHutTree::build :282-289 -- bounding box, Morton keys, sort, octree, monopoles; all on
 * the device, no host round trip (the reference crosses PCIe >= 17 times here). */
NBODY_HIP_API int nbody_hip_tree_build(nbody_hip_tree* tree, const nbody_particle_data* d);
/* The drift of a Velocity-Verlet step (ref: Integrator::integrate, integrator.cu:224-238 -- storeOldAccelerations +
 * updatePositions: a_old <- a ; x += v dt + a dt^2/2) fused with the build that follows it: the positions are
 * advanced, packed and bounded in ONE pass over the bodies, then the tree is built on them.  Same arithmetic and
 * results as nbody_hip_drift followed by nbody_hip_tree_build.  Used by Integrator::integrate for exactly the
 * engine's own BarnesHutCalculator (a subclass goes through its virtual computeForces). */
NBODY_HIP_API int nbody_hip_tree_drift_build(nbody_hip_tree* tree, nbody_particle_data* d, float dt);
/* ref: BarnesHutTree::computeForces(d_particles, theta, G, eps) :488-498 -- OVERWRITES acc_*. */
NBODY_HIP_API int nbody_hip_tree_compute_forces(nbody_hip_tree* tree, nbody_particle_data* d,
                                                float theta, float G, float eps);

/* e (multi-GPU Barnes-Hut: replicated tree, partitioned walk; no reference counterpart).
 * build_packed: the same build from float4 {x, y, z, m} bodies (the gathered bodies of all ranks).
 * compute_forces_packed: walks only the bodies at positions [first_sorted, first_sorted + count) of
 * the tree's Morton-sorted order and writes their accelerations as {ax, ay, az, 0} at the bodies'
 * ORIGINAL indices of acc_out (n float4, rows of other bodies untouched).  Ranks that walk disjoint
 * ranges of the same tree produce, together, exactly the single-GPU result. */
NBODY_HIP_API int nbody_hip_tree_build_packed(nbody_hip_tree* tree, const nbody_float4* posm, size_t n);
NBODY_HIP_API int nbody_hip_tree_compute_forces_packed(nbody_hip_tree* tree, size_t first_sorted, size_t count,
                                                       float theta, float G, float eps, nbody_float4* acc_out);
/* ref: getNodeCount (barnes_hut_tree.hpp:41) and the root mass used by verifyMassConservation
 * (:511-519); plus node visits (per wave) of the last traversal and the first node id of every
 * level: level_base[l] for l = 0 .. max_depth + 1 (the last one = node count), repeated up to
 * NBODY_HIP_TREE_LEVELS entries (trees go down to depth 21; the deepest level that holds nodes is what
 * the reference's getMaxDepth reports).  Any output may be NULL.  Blocking. */
#define NBODY_HIP_TREE_LEVELS 24
NBODY_HIP_API int nbody_hip_tree_stats(nbody_hip_tree* tree, int* node_count, float* root_mass,
                                       unsigned long long* nodes_visited, int level_base[NBODY_HIP_TREE_LEVELS]);
/* ref: copyNodesToHost / getNodes :500-503 -- writes the tree into HOST memory in the reference's
 * OctreeNode layout (76 bytes, barnes_hut_tree.hpp:9-30), children indexed by octant; optionally
 * the Morton order (sorted position -> body index, `count` ints).  Blocking. */
NBODY_HIP_API int nbody_hip_tree_copy_nodes(nbody_hip_tree* tree, void* host_nodes, int capacity_nodes,
                                            int* sorted_indices);

/* ---- measurement helpers -------------------------------------------------- */

/* Runs the direct-force kernel `iters` times back to back on the context's stream between
 * two HIP events and returns the mean milliseconds per launch (blocking).  bench.py's
 * `roofline.achieved` comes from this; rocprofv3's kernel trace must agree. */
NBODY_HIP_API int nbody_hip_time_direct_packed(nbody_hip_ctx* ctx, const nbody_float4* targets,
                                 size_t n_targets, const nbody_float4* sources, size_t n_sources,
                                 nbody_float4* acc_out, float G, float eps2, int iters,
                                 float* ms_per_launch);

/* Direct forces are bitwise reproducible, like the reference's one-thread-per-body loop
 * (ref: force_direct.cu:10-85): in the symmetric (action = -reaction) kernels every contribution to a body
 * goes to a slot of its own and the slots are added in a fixed order.
 *   mode 1 (default)  slot planes when they fit the budget -- 24 GiB AND a quarter of the device memory that is
 *                     free when the workspace has to grow -- otherwise fp64 atomics (their order varies from
 *                     launch to launch: fp32 results differ in the last bit now and then); an allocation
 *                     failure of the slot planes also falls back to the atomic form.  nbody_hip_direct_info
 *                     tells which form a call at a given size takes, and which one the last call took;
 *   mode 2            slot planes REQUIRED: a call that cannot have them fails with NBODY_HIP_ERR_RESOURCE
 *                     instead of switching silently;
 *   mode 0            fp64 atomics; the slot planes the context holds are given back at once.
 * Slot planes: a reaction is one fp32 number per component (12 bytes per body and partner superblock), an
 * I-side sum is fp64 (24 bytes per body and split): 12 D + 24 splits bytes per body with D = N / (512 R) --
 * 1.9 GB at N = 2^20 (round 2: 3.4 GB), ~14 GB at N = 2^22; a workspace more than four times larger than the
 * current call needs is released first.  Time against the atomic form at N = 2^20: see DESIGN.md section 4.1.
 * The one-sided kernel (N < 12,288, rectangular sets) is reproducible anyway; the two-set kernel of the
 * sharded path follows the same switch (splits I-side slots per body of the first set, N_first / (256 R)
 * reaction slots per body of the second). */
NBODY_HIP_API int nbody_hip_direct_deterministic(nbody_hip_ctx* ctx, int mode);

/* Most bytes of slot planes the deterministic form may take on this context (default and bytes = 0: 24 GiB;
 * a quarter of the free device memory is the second limit whatever this says).  For shared GPUs. */
NBODY_HIP_API int nbody_hip_direct_slot_budget(nbody_hip_ctx* ctx, unsigned long long bytes);

/* What a Direct all-pairs call at `count` bodies runs with the context's current settings (plain host
 * arithmetic + one hipMemGetInfo when the workspace would have to grow), and what the last call ran.
 * ctx may be NULL: the answer for a context with default settings from the fixed budget alone (no device needed). */
typedef struct nbody_hip_direct_info_t {
  int deterministic_mode;     /* the context's setting: 0, 1 or 2 (see nbody_hip_direct_deterministic) */
  int kernel;                 /* at `count` bodies: 0 one-sided, 1 symmetric + fp64 atomics, 2 symmetric + slot planes */
  int bodies_per_lane_equal;  /* R of the equal-mass instantiation (symmetric kernels; else 0) */
  int bodies_per_lane_general;/* R of the general-mass instantiation */
  int reaction_slots;         /* D  (equal-mass shape) */
  int iside_slots;            /* splits (equal-mass shape) */
  int reserved0, reserved1;
  unsigned long long workspace_bytes_needed;  /* accumulator workspace of such a call */
  unsigned long long slot_bytes_wanted;       /* what the slot planes need (also when they were refused) */
  unsigned long long workspace_bytes_held;    /* what the context holds right now */
  int last_kernel;            /* what the last Direct call of this context ran (-1: none yet) */
  int reserved2;
} nbody_hip_direct_info_t;
NBODY_HIP_API int nbody_hip_direct_info(nbody_hip_ctx* ctx, size_t count, float eps2, nbody_hip_direct_info_t* out);

/* Tuning knobs for experiments.  variant: -1 automatic, 0 scalar body + LDS sources, 1 packed
 * (v_pk_*_f32) body + LDS sources, 2 scalar body + scalar-cache sources, 3 symmetric (action =
 * -reaction) kernel whenever targets == sources; the others 0 = automatic:
 * targets_per_lane: 1, 2 or 4 (6, 8, 12, 16: symmetric kernel only; 12: all pairs only); source_splits: number of source sub-ranges per target block. */
NBODY_HIP_API int nbody_hip_direct_tuning(nbody_hip_ctx* ctx, int variant, int targets_per_lane,
                            int source_splits);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_HIP_H */
