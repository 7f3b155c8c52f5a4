// common.h -- internal declarations shared by the HIP translation units of libnbody_hip.so.
// gfx950 (MI355X) only; wave64; no CUDA compatibility layer.
#pragma once

#ifndef NBH_SORT_MERGE_LIMIT
#define NBH_SORT_MERGE_LIMIT (300 * 1000)
#endif

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>

#include "nbody_hip.h"

namespace nbh {

constexpr int kWave = 64;
constexpr int kBlock = 256;   // 4 waves: one per SIMD of a CU
constexpr int kNumCU = 256;   // MI355X
// rocPRIM's radix sort switches from its merge-sort path (2 launches per doubling: 22 at 2^20 keys)
// to Onesweep above `merge_sort_limit` (default 2^20, i.e. exactly NOT at the BASELINE size); measured:
// merge sort wins at 262,144 keys (0.24 vs 0.26 ms tree build), Onesweep from 524,288 (0.14 vs 0.18 ms)
constexpr size_t kSortMergeLimit = NBH_SORT_MERGE_LIMIT;
// ... and our own driver of the Onesweep kernels (onesweep.h: no fill launches) takes over from kOwnSortFromTree
// bodies in the Barnes-Hut build (63-bit keys, index payload) and kOwnSortFromGrid in the spatial hash (32-bit keys,
// float4 + index payload, where the merge sort is dearer) -- profiles/r03_sort_crossover.txt.  NBH_OWN_SORT_FROM in
// the environment overrides both when a tree / grid is created (the hook tools/sort_crossover.py measures with, not
// an interface).
// workgroups of the key kernels that also histogram the digits (each flushes its non-empty bins with global atomics)
#ifndef NBH_HIST_BLOCKS
#define NBH_HIST_BLOCKS 256
#endif
#ifndef NBH_HIST_THREADS
#define NBH_HIST_THREADS 1024
#endif
constexpr size_t kOwnSortFromTree = 250000, kOwnSortFromGrid = 120000;
size_t own_sort_from(size_t compiled_default);

#ifdef __HIPCC__
// x + v dt + a (dt^2 / 2) as p + fma(a, h, v * dt): the contraction nvcc's default -fmad makes of
// integrator.cu:16-19 -- one definition, so that the SoA, fused and float4 drift kernels round identically
// (a sharded run then reproduces the single-GPU one bit for bit).
__device__ __forceinline__ float drift1(float p, float v, float a, float dt, float h) {
  return p + __builtin_fmaf(a, h, v * dt);
}
#endif

void set_error(const char* fmt, ...);
int fail(nbody_hip_status code, const char* file, int line, const char* fmt, ...);

#define NBH_FAIL(code, ...) ::nbh::fail((code), __FILE__, __LINE__, __VA_ARGS__)

// calls that hand data back to the host cannot be recorded into a step graph
#define NBH_NOT_CAPTURABLE(ctx, what)                                                          \
  do {                                                                                        \
    if ((ctx)->capturing) {                                                                   \
      (ctx)->capture_failed = true;                                                           \
      return NBH_FAIL(NBODY_HIP_ERR_STATE, "%s cannot be recorded into a step graph", what);   \
    }                                                                                         \
  } while (0)

// HIP call that turns an error into NBODY_HIP_ERR_DEVICE (or _RESOURCE for OOM),
// message "<call>: <hip error string> at file:line" (ref: CUDA_CHECK, error_handling.hpp:104-114)
#define NBH_HIP(call)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      return ::nbh::fail(e_ == hipErrorOutOfMemory ? NBODY_HIP_ERR_RESOURCE                   \
                                                   : NBODY_HIP_ERR_DEVICE,                    \
                         __FILE__, __LINE__, "%s: %s", #call, hipGetErrorString(e_));         \
    }                                                                                         \
  } while (0)

// After a kernel launch (ref: CUDA_CHECK_KERNEL, error_handling.hpp:116-136; no sync in release)
#define NBH_LAUNCH_CHECK() NBH_HIP(hipGetLastError())

struct Workspace {
  void* ptr = nullptr;
  size_t bytes = 0;
  unsigned long long generation = 0;  // bumped whenever `ptr` is replaced (step graphs bake it in)
  int reserve(size_t want);  // grows (never shrinks); returns status
  void release();
};

}  // namespace nbh

struct nbody_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;      // caller-owned stream launches go to (nullptr = null stream)
  nbh::Workspace posm;               // packed {x,y,z,m} of the bodies        (direct, energy)
  nbh::Workspace partial;            // per-source-split partial accelerations (direct)
  nbh::Workspace reduce;             // block partials for energy reductions
  double* host_scalar = nullptr;     // pinned, 4 doubles
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t capture_stream = nullptr;  // library-owned, only used while recording a step graph
  hipStream_t user_stream = nullptr;     // ctx->stream saved by capture_begin
  bool capturing = false;
  mutable bool capture_failed = false;           // a non-capturable call was made while recording
  // tuning overrides (variant -1 / others 0 = automatic)
  int tune_variant = -1, tune_tpl = 0, tune_splits = 0;
  // symmetric kernels: 0 = fp64 atomics; 1 = slot planes + fixed-order sum (bitwise reproducible) when they fit the
  // budget, else atomics; 2 = slot planes REQUIRED (a call that cannot have them fails with NBODY_HIP_ERR_RESOURCE)
  int deterministic = 1;
  size_t det_budget = (size_t)24 << 30;  // most slot-plane bytes the deterministic form may take (nbody_hip_direct_slot_budget)
  int last_direct_kernel = -1;  // what the last Direct launch ran: 0 one-sided, 1 symmetric + atomics, 2 symmetric + slots
  // bumped when a tree / grid of this context re-sizes or frees device arrays a recorded step graph
  // may point into; together with the workspaces' generations it dates a recording
  unsigned long long alloc_generation = 0;
  // replays of recorded step graphs so far: a replay runs launches this library did not see, so state the host
  // side assumes about device buffers between calls (a tree's re-armed bounding box) is dated with it
  unsigned long long graph_replays = 0;
  unsigned long long generation() const {
    return posm.generation + partial.generation + reduce.generation + alloc_generation;
  }
};

namespace nbh {

// v + (a_old + a) dt/2 as one explicit chain (add, then fma): integrator.cu:33-35 under nvcc's default
// contraction; shared by every kick (SoA, float4, fused into the force finalize) so they round alike
__device__ __forceinline__ float kick1(float v, float a_old, float a_new, float dt_half) {
  return __builtin_fmaf(a_old + a_new, dt_half, v);
}

// direct_sym.hip: launch shape of the symmetric all-pairs kernel for a given number of bodies per lane, and what
// a Direct call at n bodies will run (plain host arithmetic; nbody_hip_direct_info exports it)
struct SymShape {
  int R, NB, D, per, splits;
  size_t plane;
  size_t reaction_bytes() const;  // D x 3 float planes, 256-byte aligned
  size_t det_bytes() const;       // + splits x 3 double planes
  size_t atomic_bytes() const;    // 3 double planes
};
struct DirectPlan {
  bool symmetric = false;         // false: one-sided kernel (direct.hip)
  bool det = false;               // slot planes
  SymShape eq{}, gen{};           // equal-mass / general-mass instantiation
  size_t bytes = 0;               // workspace the call needs
  size_t det_bytes_wanted = 0;    // what the slot planes would need (also when they were refused)
};
#ifndef NBH_DET_GEN_R
#define NBH_DET_GEN_R 16
#endif
constexpr int kDetGenR = NBH_DET_GEN_R;  // bodies per lane of the general-mass instantiation in deterministic mode at sizes
                                         // where the equal-mass one takes 16 (round 2: 12 -- the slot kernel spilled into
                                         // AGPRs at 16; with the single loop body of round 3 it does not)
SymShape sym_shape(const nbody_hip_ctx* ctx, size_t n, int R);
DirectPlan direct_plan(const nbody_hip_ctx* ctx, size_t n, bool query_device);

// direct.hip
int direct_packed(nbody_hip_ctx* ctx, const float4* targets, size_t n_targets,
                  const float4* sources, size_t n_sources, float G, float eps2,
                  // output selection (exactly one of acc4 / soa is used)
                  float4* acc4, int accumulate, float* ax, float* ay, float* az,
                  // optional fused velocity update (soa output only): v += (a_old + a) * half_dt
                  float* vx, float* vy, float* vz, const float* aox, const float* aoy,
                  const float* aoz, float half_dt);
// direct_sym.hip: all-pairs with action = -reaction (targets == sources)
int direct_symmetric(nbody_hip_ctx* ctx, const float4* posm, size_t n, float G, float eps2,
                     float4* acc4, int accumulate, float* ax, float* ay, float* az, float* vx,
                     float* vy, float* vz, const float* aox, const float* aoy, const float* aoz,
                     float half_dt);
int direct_symmetric_pair(nbody_hip_ctx* ctx, const float4* pi, size_t ni, const float4* pj, size_t nj,
                          float G, float eps2, float4* acc_i, int accumulate_i, float4* acc_j,
                          int accumulate_j);
bool symmetric_pays(const nbody_hip_ctx* ctx, size_t n);
int pack_posm(nbody_hip_ctx* ctx, const float* x, const float* y, const float* z, const float* m,
              size_t n, float4* out);

// spatial_hash.hip: order-preserving-integer bounding box of packed bodies into enc[6]
// (min x,y,z then max x,y,z); decode with ordered_to_float on the device.
int launch_bbox_init(nbody_hip_ctx* ctx, unsigned int* enc);
// drift of a Velocity-Verlet step fused with the packing and the bounding box of the following build:
// a_old <- a ; x += v dt + a dt^2/2 ; posm <- {x, y, z, m} ; enc <- box   (nbody_hip_{tree,grid}_drift_build)
int launch_drift_pack_bbox(nbody_hip_ctx* ctx, nbody_particle_data* d, float dt, float4* posm, unsigned int* enc,
                           bool init = true);
int launch_bbox(nbody_hip_ctx* ctx, const float4* posm, int n, unsigned int* enc, bool init = true);
int launch_pack_bbox(nbody_hip_ctx* ctx, const float* x, const float* y, const float* z, const float* m, int n,
                     float4* posm, unsigned int* enc, bool init = true);

__device__ __forceinline__ unsigned int float_to_ordered(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(unsigned int o) {
  const unsigned int u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

// The *_destroy entry points may be reached while the process is exiting (a host language's garbage collector running
// after the HIP runtime's own static destructors): a runtime that is gone answers hipErrorDeinitialized -- or throws
// from inside (std::bad_variant_access out of hipStreamSynchronize was seen once: rc 134).  Then nothing is freed --
// the address space is about to disappear anyway -- and nothing may propagate across the C ABI.
inline bool runtime_alive() noexcept {
  try {
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    return n > 0;
  } catch (...) {
    return false;
  }
}
#define NBH_DESTROY_BEGIN          \
  if (!nbh::runtime_alive()) return NBODY_HIP_OK; \
  try {
#define NBH_DESTROY_END            \
  } catch (...) {                  \
  }                                \
  return NBODY_HIP_OK;

// Entry points that work on SEVERAL devices (the sharded systems, the communicator) hipSetDevice their way through the
// local ranks; the caller's current device is put back on every exit path -- the facade and the reference model are
// "current device, null stream" (ref: include/nbody/force_calculator.hpp:8-19), so a caller's next allocation or
// null-stream launch must land where it was before the call.
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() noexcept {
    try {
      if (hipGetDevice(&prev) != hipSuccess) {
        prev = -1;
        (void)hipGetLastError();
      }
    } catch (...) {  // (a runtime that is already gone: see runtime_alive)
      prev = -1;
    }
  }
  ~DeviceGuard() {
    try {
      if (prev >= 0) (void)hipSetDevice(prev);
    } catch (...) {
    }
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace nbh
