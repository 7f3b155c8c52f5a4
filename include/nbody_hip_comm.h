/*
 * nbody_hip_comm.h -- multi-GPU part of the C ABI of libnbody_hip.so ("nbody_hip_comm_{init,
 * destroy}(ndev) for the sharded variants"; section 8e: Direct N^2 shards by contiguous index range with one exchange
 * of positions per step).  The reference is single-GPU (no counterpart; what it replaces is the single launch of
 * ref: src/cuda/force_direct.cu:88-98 behind DirectForceCalculator::computeForces :100-106 and the step of
 * ref: src/cuda/integrator.cu:224-238, spread over the GPUs of one node).
 *
 * Ranks and transports
 *   A communicator is a group of `world` ranks, one GPU each.  Two ways to make one:
 *     nbody_hip_comm_init_all   ONE process drives all `ndev` devices (like ncclCommInitAll): every rank is local.
 *                               transport P2P : positions and reactions travel as direct peer copies
 *                                               (hipMemcpyAsync device-to-device over xGMI, ordered by HIP events);
 *                                               needs nothing but the HIP runtime.  The same device may be listed
 *                                               more than once ("virtual ranks": how the tests run 2-8 ranks on the
 *                                               one GPU of a test box).
 *                               transport RCCL: ncclCommInitAll + grouped collectives.
 *     nbody_hip_comm_init_rank  one process per GPU (the launch model of torchrun / mpirun): RCCL over xGMI;
 *                               the 128-byte id from nbody_hip_comm_unique_id travels by whatever channel the
 *                               launcher has (MPI_Bcast, a torch.distributed store, a file).
 *   librccl.so.1 is opened at run time (dlopen: the copy already in the process -- e.g. PyTorch's -- or ROCm's);
 *   the library has no link-time dependency on it, and the P2P transport never touches it.
 *
 * One Velocity-Verlet step of the sharded Direct system (per rank r of W, shard = S = ceil(N / W) bodies):
 *     drift own bodies (in place inside the gathered array)             compute stream
 *     all-gather {x,y,z,m}: 16 B per body, 2 MiB per rank at N = 2^20   comm stream   } overlapped
 *     own shard x own shard, symmetric kernel                           compute stream }
 *     every PAIR of shards is evaluated by ONE rank, action and reaction together (rank r takes the ring
 *       neighbours r+1 .. r+(W-1)/2 and, for even W, half of the antipodal rectangle): N^2 / (2W) pair evaluations
 *       per rank; the reactions on the partner shard go into a block of their own
 *     exchange: each block travels to the ONE rank that owns its bodies -- (W-1)/2 point-to-point messages of
 *       16 B x S per rank (no reduce-scatter over the whole body array: on xGMI's point-to-point links every
 *       message takes its own link)
 *     a_new = own part + the received blocks in a FIXED order ; v += (a_old + a_new) dt/2     (one kernel)
 *   With the deterministic two-set kernel (nbody_hip_direct_deterministic, default) the whole step is bitwise
 *   reproducible run after run; softening below 1e-6 (eps^2 < 1e-12) takes the one-sided kernel against the
 *   gathered bodies instead (no exchange).
 *
 * Devices and streams: the entry points of this header walk over the local ranks' devices (hipSetDevice) and restore
 * the CALLER'S CURRENT DEVICE on every return path (round 4; csrc/common.h DeviceGuard) -- the facade and the
 * reference's threading model are "current device, null stream" (ref: include/nbody/force_calculator.hpp:8-19), so the
 * caller's next allocation or null-stream launch lands where it did before the call.  The work itself runs on the
 * system's own streams (one compute and one comm stream per local rank); the resident calls (forces / step) are
 * asynchronous and ordered among themselves, nbody_hip_sharded_*_synchronize waits for them.  Only the plugin form
 * (nbody_hip_sharded_direct_compute_forces) touches caller memory asynchronously: it orders itself against the NULL
 * stream of the FIRST local rank's device (where `d` must live) at entry and exit, not against any other stream or
 * device of the caller.
 *
 * Limits of the sharded spatial hash (nbody_hip_sharded_hash_*): a grid of at most 1e8 cells like the reference's
 * (ref: src/cuda/force_spatial_hash.cu:252-254) AND at most 4,096 layers along z (the partition pass histograms the
 * layers in one workgroup's LDS): taller grids are refused with NBODY_HIP_ERR_RESOURCE, where the single-GPU grid
 * accepts any shape up to 1e8 cells.  At cell = cutoff that is a box 4,096 cutoffs tall.
 *
 * Failure of ONE rank in a multi-process run (one process per GPU, nbody_hip_comm_init_rank): the per-step buffers are
 * sized from the all-reduced send matrix, i.e. after the step's collective decision, and grown with 12-25 % headroom.  A
 * rank whose hipMalloc fails there returns NBODY_HIP_ERR_RESOURCE while its peers have already entered the point-to-point
 * group and wait for it: the host must treat a non-OK status of any rank as fatal for the job (destroy the communicator
 * / end the processes), as bench.py does with its watchdog.  With all ranks in one process (nbody_hip_comm_init_all)
 * the failing call returns before any rank has posted the group.
 *
 * Status codes and error text as in nbody_hip.h; RCCL failures are NBODY_HIP_ERR_COMM.
 */
#ifndef NBODY_HIP_COMM_H
#define NBODY_HIP_COMM_H

#include "nbody_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define NBODY_HIP_ERR_COMM (-5) /* RCCL error / RCCL not loadable */

#define NBODY_HIP_TRANSPORT_P2P 0
#define NBODY_HIP_TRANSPORT_RCCL 1
#define NBODY_HIP_MAX_RANKS 32

typedef struct nbody_hip_comm nbody_hip_comm;
typedef struct nbody_hip_comm_id { char bytes[128]; } nbody_hip_comm_id; /* = ncclUniqueId */

/* One process x ndev devices.  devices == NULL: 0 .. ndev-1.  P2P: peer access is enabled between distinct devices;
 * repeated device numbers make virtual ranks that share a GPU.  RCCL: devices must be distinct. */
NBODY_HIP_API int nbody_hip_comm_init_all(int ndev, const int* devices, int transport, nbody_hip_comm** out);
/* One process per GPU over RCCL: rank 0 calls nbody_hip_comm_unique_id and hands the id to the others. */
NBODY_HIP_API int nbody_hip_comm_unique_id(nbody_hip_comm_id* id);
NBODY_HIP_API int nbody_hip_comm_init_rank(int device, int rank, int world, const nbody_hip_comm_id* id,
                                           nbody_hip_comm** out);
/* Any output may be NULL.  local_ranks: the first `nlocal` entries are the ranks living in this process. */
NBODY_HIP_API int nbody_hip_comm_info(const nbody_hip_comm* comm, int* world, int* nlocal, int* transport,
                                      int local_ranks[NBODY_HIP_MAX_RANKS]);
NBODY_HIP_API int nbody_hip_comm_destroy(nbody_hip_comm* comm);

/* The host-side partition logic, exported so that it can be checked without a GPU: shard size S and [lo, hi) of
 * `rank`; and the shard pairs `rank` evaluates -- rows {i0, i1, partner shard, j0, j1} meaning own bodies
 * [i0, i1) x bodies [j0, j1) of the partner -- every pair of bodies of two different shards in exactly one
 * rank's list.  Returns the number of rows (<= NBODY_HIP_MAX_RANKS / 2 + 1) or a negative status. */
NBODY_HIP_API int nbody_hip_shard_bounds(size_t n, int world, int rank, size_t* shard, size_t* lo, size_t* hi);
NBODY_HIP_API int nbody_hip_pair_schedule(int world, int rank, size_t shard, size_t rows[][5], int max_rows);

/* ---- the sharded Direct N^2 system (BASELINE config 3) ---------------------------------------------------------- */
typedef struct nbody_hip_sharded_direct nbody_hip_sharded_direct;

/* n bodies over the communicator's ranks; eps is the softening length (eps^2 is formed in fp32 like
 * ForceCalculator::setSofteningParameter, force_calculator.hpp:52-57). */
NBODY_HIP_API int nbody_hip_sharded_direct_create(nbody_hip_comm* comm, size_t n, float G, float eps,
                                                  nbody_hip_sharded_direct** out);
NBODY_HIP_API int nbody_hip_sharded_direct_destroy(nbody_hip_sharded_direct* s);
/* Body state from HOST arrays of the WHOLE system (n floats each; velocities may be NULL = zero): every local rank
 * uploads its own range.  In the one-process-per-GPU model every process passes the same arrays.  Blocking. */
NBODY_HIP_API int nbody_hip_sharded_direct_set_state(nbody_hip_sharded_direct* s, const float* x, const float* y,
                                                     const float* z, const float* mass, const float* vx,
                                                     const float* vy, const float* vz);
/* a(0): one exchange + force evaluation (ref: ParticleSystem::initialize, particle_system.cpp:88-91). */
NBODY_HIP_API int nbody_hip_sharded_direct_forces(nbody_hip_sharded_direct* s);
/* `steps` Velocity-Verlet steps (see the file header).  Asynchronous; every rank of the communicator must call it. */
NBODY_HIP_API int nbody_hip_sharded_direct_step(nbody_hip_sharded_direct* s, float dt, int steps);
/* The same, timed by HIP events on the compute stream of the first local rank (after `warmup` untimed steps):
 * milliseconds per step.  Blocking. */
NBODY_HIP_API int nbody_hip_sharded_direct_time_steps(nbody_hip_sharded_direct* s, float dt, int warmup, int steps,
                                                      float* ms_per_step);
NBODY_HIP_API int nbody_hip_sharded_direct_synchronize(nbody_hip_sharded_direct* s);
/* HOST arrays of n floats (any may be NULL): each process fills the rows of ITS local ranks; with gather != 0 the
 * ranks first all-gather, so that every process gets every row.  Blocking. */
NBODY_HIP_API int nbody_hip_sharded_direct_get_state(nbody_hip_sharded_direct* s, float* x, float* y, float* z,
                                                     float* vx, float* vy, float* vz, float* ax, float* ay,
                                                     float* az, int gather);
/* KE and PE of the whole system on every rank: each rank reduces its shard against the gathered bodies
 * (nbody_hip_energies_packed) and the 2 x W doubles are summed in rank order.  Blocking. */
NBODY_HIP_API int nbody_hip_sharded_direct_energies(nbody_hip_sharded_direct* s, double* kinetic, double* potential);

/* The plugin form -- what a multi-GPU ForceCalculator::computeForces(ParticleData*) is made of (the facade's
 * ShardedDirectCalculator): `d` holds the WHOLE system in the reference's SoA layout on the device of the
 * first local rank; its positions are packed, handed to every rank, the force work is shared as above, and the
 * accelerations of all n bodies are written back into d->acc_*.  16 B/body out and 16 B/body back per call over
 * xGMI (0.3 ms at N = 2^20 against ~20 ms of force work on 8 GPUs).  In the one-process-per-GPU model every
 * process passes its own replica of `d` and gets every acceleration.  Asynchronous on the first local rank's
 * compute stream, which waits for the null stream of `d`'s device at entry and is waited for by it at exit. */
NBODY_HIP_API int nbody_hip_sharded_direct_compute_forces(nbody_hip_sharded_direct* s, nbody_particle_data* d);

/* ---- the sharded spatial hash (BASELINE config 5) --------------------------------------------------------------
 * z-slabs of cells over the communicator's ranks (the linear cell id x + y gx + z gx gy, ref: force_spatial_hash.cu:48,
 * makes a range of z layers a contiguous block of the cell-ordered body list).  One force evaluation: all-reduce of
 * the bounding box -> ONE partition pass (layer and owner of every body, the bodies that change owner as 64-byte
 * rows, send matrix, layer histogram) -> all-reduce (sum) -> the step's one host synchronisation -> the migrating
 * rows travel point to point, arrivals fill the vacated slots -> own grid -> the two boundary layers travel to the
 * neighbours (16 B per body) WHILE the wave-per-cell kernel evaluates own x own -> a grid over the received layers,
 * the boundary layers against it.  Same results as the single grid (ref: SpatialHashCalculator::computeForces,
 * force_spatial_hash.cu:334-377) up to the summation order of the boundary bodies.  csrc/sharded_hash.hip. */
typedef struct nbody_hip_sharded_hash nbody_hip_sharded_hash;

NBODY_HIP_API int nbody_hip_sharded_hash_create(nbody_hip_comm* comm, size_t n, float G, float eps, float cell_size,
                                                float cutoff, nbody_hip_sharded_hash** out);
NBODY_HIP_API int nbody_hip_sharded_hash_destroy(nbody_hip_sharded_hash* s);
/* HOST arrays of the whole system (n floats each; velocities may be NULL): every local rank keeps the bodies of its
 * slab of the initial global grid (every process passes the same arrays).  Blocking. */
NBODY_HIP_API int nbody_hip_sharded_hash_set_state(nbody_hip_sharded_hash* s, const float* x, const float* y,
                                                   const float* z, const float* mass, const float* vx,
                                                   const float* vy, const float* vz);
NBODY_HIP_API int nbody_hip_sharded_hash_forces(nbody_hip_sharded_hash* s);              /* a(0) */
/* `steps` Velocity-Verlet steps; each contains one host synchronisation (the grid size decides validity, as in the
 * reference), so the call returns when the last step's exchange is done and its tail is queued. */
NBODY_HIP_API int nbody_hip_sharded_hash_step(nbody_hip_sharded_hash* s, float dt, int steps);
NBODY_HIP_API int nbody_hip_sharded_hash_time_steps(nbody_hip_sharded_hash* s, float dt, int warmup, int steps,
                                                    float* ms_per_step);
NBODY_HIP_API int nbody_hip_sharded_hash_synchronize(nbody_hip_sharded_hash* s);
/* HOST arrays of n floats indexed by the bodies' GLOBAL ids (any may be NULL); each process fills the rows of the
 * bodies its local ranks hold.  Blocking. */
NBODY_HIP_API int nbody_hip_sharded_hash_get_state(nbody_hip_sharded_hash* s, float* x, float* y, float* z, float* vx,
                                                   float* vy, float* vz, float* ax, float* ay, float* az);
/* Of the last evaluation (any output may be NULL): global grid dims; 1 if every rank took the overlapped two-grid
 * path; bodies that changed owner and halo bodies received, summed over the local ranks; bodies per local rank. */
NBODY_HIP_API int nbody_hip_sharded_hash_info(const nbody_hip_sharded_hash* s, int dims[3], int* two_grid,
                                              unsigned long long* migrated, unsigned long long* halo_bodies,
                                              unsigned long long local_counts[NBODY_HIP_MAX_RANKS]);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_HIP_COMM_H */
