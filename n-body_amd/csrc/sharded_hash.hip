// sharded_hash.hip -- BASELINE config 5 behind the C ABI (include/nbody_hip_comm.h): the spatial hash sharded by z-slabs
// of cells over the ranks of a communicator, with migration of the bodies that change slab and a halo exchange of
// the boundary layers overlapped with the own x own force kernel.  The reference is single-GPU; what this spreads
// over the ranks is SpatialHashCalculator::computeForces (src/cuda/force_spatial_hash.cu:334-377) inside the step of
// Integrator::integrate (src/cuda/integrator.cu:224-238).  Same algorithm as the torch.distributed host
// (n-body_amd/distributed.py, ShardedHashSystem) -- that one stays as the gloo-testable twin -- over the kernels of
// spatial_hash.hip / slab.hip.
//
// One force evaluation (after the drift), per rank r of W.  The linear cell id x + y gx + z gx gy
// (force_spatial_hash.cu:48) makes a range of z layers a contiguous block of the cell-ordered body list; rank r owns the
// layers [r gz / W, (r + 1) gz / W) of the GLOBAL grid:
//   1 local min / max -> all-reduce (min of the lows, max of the highs): every rank bins on the same grid
//   2 ONE partition pass (nbody_hip_slab_partition): layer and new owner of every body; the bodies that CHANGE OWNER
//     as 64-byte rows grouped by owner; the rank's row of the W x W send matrix and its bodies-per-layer histogram
//   3 all-reduce (sum) of matrix + histogram: who sends whom how many rows, how many bodies every rank will own, how
//     many bodies its halo layers hold
//   4 THE host synchronisation of the step: box, matrix, histogram ("grid too large" is synchronous in the reference too)
//   5 migration: the rows travel point to point (exact sizes); arrivals fill the vacated slots (nbody_hip_slab_fill):
//     the bodies that stay -- almost all -- never move
//   6 own grid (cell-ordered build on the global box): its lowest and highest layer are the head and the tail of the
//     cell order -- the halo the neighbours need, no selection pass
//   7 halo: those two layers travel to the owners of the adjacent layers (16 B per body) on the comm stream ...
//   8 ... while the wave-per-cell kernel evaluates own x own on the compute stream
//   9 a grid over the received layers; the kernel again for the rank's two boundary layers against it, accumulated
// Grids too sparse for the per-cell start arrays of the two-grid kernel take one grid over own + halo bodies instead
// (decided from the histogram BEFORE anything is launched).  Transports as in sharded.hip: RCCL, or -- one process
// driving all devices -- peer copies ordered by events, with the two small reductions done by kernels that read the
// peers' buffers.  Because every rank synchronises with the host once per evaluation, no buffer is reused while a
// peer of the same process may still read it.

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "comm.h"

namespace nbh {

constexpr int kHistCap = 4096;  // layers the partition pass can histogram (taller grids are refused)

struct PeerPtrs {
  const void* p[NBODY_HIP_MAX_RANKS];
  int n;
};
// {min x,y,z, max x,y,z} over the ranks' boxes
__global__ void box_reduce_kernel(PeerPtrs in, float* __restrict__ out) {
  const int t = threadIdx.x;
  if (t >= 6) return;
  float v = static_cast<const float*>(in.p[0])[t];
  for (int k = 1; k < in.n; k++) {
    const float w = static_cast<const float*>(in.p[k])[t];
    v = t < 3 ? fminf(v, w) : fmaxf(v, w);
  }
  out[t] = v;
}
__global__ __launch_bounds__(kBlock) void int_sum_kernel(PeerPtrs in, int count, int* __restrict__ out) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock) {
    int v = 0;
    for (int k = 0; k < in.n; k++) v += static_cast<const int*>(in.p[k])[i];
    out[i] = v;
  }
}
__global__ void box_flip_kernel(float* __restrict__ box) {  // {lo, hi} <-> {lo, -hi}
  if (threadIdx.x >= 3 && threadIdx.x < 6) box[threadIdx.x] = -box[threadIdx.x];
}
__global__ void box_empty_kernel(float* __restrict__ box) {
  if (threadIdx.x < 6) box[threadIdx.x] = threadIdx.x < 3 ? 3.0e38f : -3.0e38f;
}
__global__ __launch_bounds__(kBlock) void iota_kernel(int* __restrict__ a, int n, int base) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) a[i] = base + i;
}

}  // namespace nbh

using namespace nbh;

namespace {

struct HShard {
  int rank = 0, device = 0;
  nbody_hip_ctx* ctx = nullptr;
  hipStream_t compute = nullptr, comm = nullptr;
  ncclComm_t nccl = nullptr;
  // the rank's bodies at the front of capacity arrays
  size_t cap = 0, n = 0;
  float4 *posm = nullptr, *vel = nullptr, *acc = nullptr, *acc2 = nullptr;
  int* gid = nullptr;
  // partition pass
  float* rows = nullptr;
  int* holes = nullptr;
  size_t part_cap = 0;
  int* stats = nullptr;            // W*W + kHistCap ints: send matrix row block | layer histogram of this rank
  // what the host reads once per evaluation, ONE block and one copy: stats_sum [nstats] | info [4] | gbox [6 floats]
  int *sum_block = nullptr, *h_block = nullptr;  // device, pinned mirror
  int *stats_sum = nullptr, *info = nullptr, *h_stats = nullptr, *h_info = nullptr;
  float *gbox = nullptr, *h_box = nullptr;
  float* box = nullptr;            // 6 floats: this rank's box
  unsigned int* enc = nullptr;     // 6 ordered-int words of the fused drift + box pass (nbody_hip_drift_bbox_packed)
  // exchange buffers
  float* got = nullptr;
  size_t got_cap = 0;
  float4 *halo_out = nullptr, *halo_in = nullptr, *cat = nullptr, *cat_acc = nullptr;
  size_t halo_out_cap = 0, halo_in_cap = 0, cat_cap = 0;
  // the start arrays of the exchanged boundary layers (nbody_hip_grid_export_layer): [2][lb_cap] ints each; out: my
  // lowest / highest layer, in: the layer below my slab / above it
  int *halo_lb_out = nullptr, *halo_lb_in = nullptr;
  size_t lb_cap = 0;
  nbody_hip_grid *g_own = nullptr, *g_one = nullptr;
  size_t g_own_cap = 0, g_one_cap = 0;
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr, t0 = nullptr, t1 = nullptr;
  // per evaluation (host)
  size_t n_leave = 0, n_arrive = 0, n_head = 0, n_tail = 0, n_halo = 0;
};

template <class T>
int regrow(T** p, size_t* cap, size_t need, size_t keep_rows, size_t row_elems) {
  if (need <= *cap) return NBODY_HIP_OK;
  const size_t ncap = need + need / 8 + 1024;
  T* q = nullptr;
  NBH_HIP(hipMalloc(reinterpret_cast<void**>(&q), ncap * row_elems * sizeof(T)));
  if (*p && keep_rows) NBH_HIP(hipMemcpy(q, *p, keep_rows * row_elems * sizeof(T), hipMemcpyDeviceToDevice));
  if (*p) NBH_HIP(hipFree(*p));
  *p = q;
  *cap = ncap;
  return NBODY_HIP_OK;
}

}  // namespace

struct nbody_hip_sharded_hash {
  nbody_hip_comm* comm = nullptr;
  size_t n_total = 0;
  int W = 1;
  float G = 1.f, eps = 0.f, cell = 1.f, cutoff = 1.f;
  std::vector<HShard> sh;
  HShard* by_rank[NBODY_HIP_MAX_RANKS] = {};
  bool have_state = false;
  // slab boundaries: W - 1 physical z coordinates (nbody_hip_slab_partition_cuts), chosen from the layer histogram so
  // that the slabs hold about the same number of bodies; empty: equal layer counts (NBH_SLAB_BALANCE=0, or before the
  // first state)
  std::vector<float> zcuts;
  bool balance = true;
  // diagnostics of the last evaluation
  int two_grid = 1;
  unsigned long long migrated = 0, halo_bodies = 0;
  int dims[3] = {0, 0, 0};
};

static void hshard_release(HShard& x) {
  (void)hipSetDevice(x.device);
  if (x.compute) (void)hipStreamSynchronize(x.compute);
  if (x.comm) (void)hipStreamSynchronize(x.comm);
  for (nbody_hip_grid* g : {x.g_own, x.g_one})
    if (g) (void)nbody_hip_grid_destroy(g);
  void* dev[] = {x.posm, x.vel, x.acc, x.acc2, x.gid, x.rows, x.holes, x.stats, x.sum_block, x.box, x.enc,
                 x.got, x.halo_out, x.halo_in, x.cat, x.cat_acc, x.halo_lb_out, x.halo_lb_in};
  for (void* p : dev) (void)hipFree(p);
  if (x.h_block) (void)hipHostFree(x.h_block);
  for (hipEvent_t e : {x.ev_a, x.ev_b, x.ev_c, x.ev_d, x.t0, x.t1})
    if (e) (void)hipEventDestroy(e);
  if (x.ctx) (void)nbody_hip_ctx_destroy(x.ctx);
  if (x.compute) (void)hipStreamDestroy(x.compute);
  if (x.comm) (void)hipStreamDestroy(x.comm);
}

extern "C" int nbody_hip_sharded_hash_destroy(nbody_hip_sharded_hash* s) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!s) return NBODY_HIP_OK;
  NBH_DESTROY_BEGIN
  for (auto& x : s->sh) hshard_release(x);
  delete s;
  NBH_DESTROY_END
}

extern "C" int nbody_hip_sharded_hash_create(nbody_hip_comm* comm, size_t n, float G, float eps, float cell_size,
                                             float cutoff, nbody_hip_sharded_hash** out) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!comm || !out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  *out = nullptr;
  if (n == 0) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Particle count must be greater than 0");
  if (n > 100000000u) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Particle count exceeds maximum supported (100M)");
  if (!(G > 0.0f) || !(G < INFINITY)) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Gravitational constant must be positive and finite");
  if (!(eps >= 0.0f) || !(eps < INFINITY)) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Softening parameter must be non-negative and finite");
  if (!(cell_size > 0.0f) || !(cell_size < INFINITY)) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Spatial hash cell size must be positive and finite");
  if (!(cutoff > 0.0f) || !(cutoff < INFINITY)) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Spatial hash cutoff must be positive and finite");
  nbody_hip_sharded_hash* s = new nbody_hip_sharded_hash();
  s->comm = comm;
  s->n_total = n;
  s->W = comm->world;
  s->G = G; s->eps = eps; s->cell = cell_size; s->cutoff = cutoff;
  {
    const char* env = std::getenv("NBH_SLAB_BALANCE");
    s->balance = !(env && env[0] == '0');
  }
  const int W = s->W;
  const size_t nstats = (size_t)W * W + kHistCap;
  s->sh.resize(comm->local.size());
  auto fail = [&](int rc) {
    nbody_hip_sharded_hash_destroy(s);
    return rc;
  };
  for (size_t k = 0; k < comm->local.size(); k++) {
    HShard& x = s->sh[k];
    x.rank = comm->local[k].rank;
    x.device = comm->local[k].device;
    x.nccl = comm->local[k].nccl;
    s->by_rank[x.rank] = &x;
    hipError_t e = hipSetDevice(x.device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&x.compute, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&x.comm, hipStreamNonBlocking);
    for (hipEvent_t* ev : {&x.ev_a, &x.ev_b, &x.ev_c, &x.ev_d})
      if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&x.t0);
    if (e == hipSuccess) e = hipEventCreate(&x.t1);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&x.stats), nstats * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&x.sum_block), (nstats + 10) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&x.box), 6 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&x.enc), 8 * sizeof(unsigned int));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&x.h_block), (nstats + 10) * sizeof(int), hipHostMallocDefault);
    if (e == hipSuccess) {
      const unsigned int empty[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
      e = hipMemcpy(x.enc, empty, sizeof(empty), hipMemcpyHostToDevice);
      x.stats_sum = x.sum_block;           x.h_stats = x.h_block;
      x.info = x.sum_block + nstats;       x.h_info = x.h_block + nstats;
      x.gbox = reinterpret_cast<float*>(x.sum_block + nstats + 4);
      x.h_box = reinterpret_cast<float*>(x.h_block + nstats + 4);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail(NBH_FAIL(e == hipErrorOutOfMemory ? NBODY_HIP_ERR_RESOURCE : NBODY_HIP_ERR_DEVICE,
                           "sharded spatial hash, rank %d on device %d: %s", x.rank, x.device, hipGetErrorString(e)));
    }
    if (int rc = nbody_hip_ctx_create(&x.ctx, x.device, x.compute)) return fail(rc);
  }
  *out = s;
  return NBODY_HIP_OK;
}

// padded box and grid dims exactly as SpatialHashGrid::build derives them (force_spatial_hash.cu:225-246)
static void grid_from_box(const float raw[6], float cell, float bounds[6], int dims[3]) {
  for (int a = 0; a < 3; a++) {
    bounds[a] = raw[a] - 0.001f;
    bounds[3 + a] = raw[3 + a] + 0.001f;
    const float cells = ceilf((bounds[3 + a] - bounds[a]) / cell);
    dims[a] = (cells < 1.0e9f && cells >= 0.0f) ? (int)cells + 1 : 0x40000000;
  }
}
static int layer_owner_equal(int z, int gz, int W) {  // rank r owns [r gz / W, (r + 1) gz / W)
  int r = (int)(((long long)(z + 1) * W - 1) / gz);
  while (r > 0 && (long long)r * gz / W > z) r--;
  while (r + 1 < W && (long long)(r + 1) * gz / W <= z) r++;
  return r;
}
// owner of every layer of a grid and the ranks' layer ranges, exactly as the partition pass on the device assigns them
struct SlabMap {
  std::vector<int> owner;       // [gz]
  int lo[NBODY_HIP_MAX_RANKS], hi[NBODY_HIP_MAX_RANKS];
};
static SlabMap slab_map(const nbody_hip_sharded_hash* s, float lo_z, int gz) {
  SlabMap m;
  const int W = s->W;
  m.owner.resize((size_t)gz);
  for (int r = 0; r < W; r++) m.lo[r] = m.hi[r] = 0;
  for (int z = 0; z < gz; z++)
    m.owner[(size_t)z] = s->zcuts.empty() ? layer_owner_equal(z, gz, W)
                                           : nbody_hip_slab_layer_owner(z, lo_z, s->cell, W, s->zcuts.data());
  for (int r = 0, z = 0; r < W; r++) {  // owners ascend with z: a rank's layers are contiguous
    while (z < gz && m.owner[(size_t)z] < r) z++;
    m.lo[r] = z;
    while (z < gz && m.owner[(size_t)z] == r) z++;
    m.hi[r] = z;
  }
  return m;
}
// cuts that give every rank about n / W bodies, from the bodies per layer of a grid with lower bound lo_z: the k-th cut is
// the lower edge of the first layer at which the running count reaches k n / W
static void balanced_cuts(nbody_hip_sharded_hash* s, const int* hist, int gz, float lo_z, size_t n) {
  const int W = s->W;
  s->zcuts.assign((size_t)(W > 1 ? W - 1 : 0), 0.f);
  long long run = 0;
  int z = 0;
  for (int k = 1; k < W; k++) {
    const long long want = (long long)((double)n * k / W);
    while (z < gz && run + hist[z] <= want) run += hist[z++];
    // layer z holds the body that crosses k n / W: it goes to the side that leaves the smaller error
    int b = z;
    if (z < gz && (want - run) * 2 > hist[z]) b = z + 1;
    s->zcuts[(size_t)(k - 1)] = lo_z + (float)b * s->cell;
  }
}

static int body_arrays_reserve(HShard& x, size_t need) {
  if (need <= x.cap) return NBODY_HIP_OK;
  NBH_HIP(hipSetDevice(x.device));
  NBH_HIP(hipStreamSynchronize(x.compute));
  NBH_HIP(hipStreamSynchronize(x.comm));
  size_t c1 = x.cap, c2 = x.cap, c3 = x.cap, c4 = x.cap, c5 = x.cap;
  if (int rc = regrow(&x.posm, &c1, need, x.n, 1)) return rc;
  if (int rc = regrow(&x.vel, &c2, need, x.n, 1)) return rc;
  if (int rc = regrow(&x.acc, &c3, need, x.n, 1)) return rc;
  if (int rc = regrow(&x.acc2, &c4, need, 0, 1)) return rc;
  if (int rc = regrow(&x.gid, &c5, need, x.n, 1)) return rc;
  x.cap = c1;
  return NBODY_HIP_OK;
}

static int grid_for(HShard& x, nbody_hip_grid** g, size_t* cap, size_t need, float cell) {
  if (*g && need <= *cap) return NBODY_HIP_OK;
  if (*g) (void)nbody_hip_grid_destroy(*g);
  *g = nullptr;
  const size_t ncap = need + need / 2 + 1024;
  if (int rc = nbody_hip_grid_create(x.ctx, ncap, cell, g)) return rc;
  *cap = ncap;
  return NBODY_HIP_OK;
}

// initial ownership: every process evaluates the same global grid on the host
extern "C" int nbody_hip_sharded_hash_set_state(nbody_hip_sharded_hash* s, const float* x, const float* y, const float* z,
                                                const float* mass, const float* vx, const float* vy, const float* vz) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!s) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null system");
  if (!x || !y || !z || !mass) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if ((vx || vy || vz) && !(vx && vy && vz)) return NBH_FAIL(NBODY_HIP_ERR_STATE, "velocity arrays: all three or none");
  const size_t n = s->n_total;
  float raw[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
  for (size_t i = 0; i < n; i++) {
    raw[0] = fminf(raw[0], x[i]); raw[3] = fmaxf(raw[3], x[i]);
    raw[1] = fminf(raw[1], y[i]); raw[4] = fmaxf(raw[4], y[i]);
    raw[2] = fminf(raw[2], z[i]); raw[5] = fmaxf(raw[5], z[i]);
  }
  float bounds[6];
  int dims[3];
  grid_from_box(raw, s->cell, bounds, dims);
  if ((long long)dims[0] * dims[1] * dims[2] > 100000000LL || dims[2] > kHistCap)
    return NBH_FAIL(NBODY_HIP_ERR_RESOURCE, "Spatial hash grid too large: reduce cell_size or bounding box");
  s->zcuts.clear();
  if (s->balance && s->W > 1) {
    std::vector<int> hist((size_t)dims[2], 0);
    for (size_t i = 0; i < n; i++) {
      int cz = (int)floorf((z[i] - bounds[2]) / s->cell);
      hist[(size_t)(cz < 0 ? 0 : (cz > dims[2] - 1 ? dims[2] - 1 : cz))]++;
    }
    balanced_cuts(s, hist.data(), dims[2], bounds[2], n);
  }
  const SlabMap map0 = slab_map(s, bounds[2], dims[2]);
  for (auto& sh : s->sh) {
    std::vector<float4> p, v;
    std::vector<int> id;
    for (size_t i = 0; i < n; i++) {
      int cz = (int)floorf((z[i] - bounds[2]) / s->cell);
      cz = cz < 0 ? 0 : (cz > dims[2] - 1 ? dims[2] - 1 : cz);
      if (map0.owner[(size_t)cz] != sh.rank) continue;
      p.push_back(make_float4(x[i], y[i], z[i], mass[i]));
      v.push_back(vx ? make_float4(vx[i], vy[i], vz[i], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f));
      id.push_back((int)i);
    }
    sh.n = 0;
    if (int rc = body_arrays_reserve(sh, p.size() + 1)) return rc;
    NBH_HIP(hipSetDevice(sh.device));
    NBH_HIP(hipStreamSynchronize(sh.compute));
    NBH_HIP(hipStreamSynchronize(sh.comm));
    sh.n = p.size();
    if (sh.n) {
      NBH_HIP(hipMemcpy(sh.posm, p.data(), sh.n * sizeof(float4), hipMemcpyHostToDevice));
      NBH_HIP(hipMemcpy(sh.vel, v.data(), sh.n * sizeof(float4), hipMemcpyHostToDevice));
      NBH_HIP(hipMemcpy(sh.gid, id.data(), sh.n * sizeof(int), hipMemcpyHostToDevice));
    }
    NBH_HIP(hipMemset(sh.acc, 0, sh.cap * sizeof(float4)));
    NBH_HIP(hipMemset(sh.acc2, 0, sh.cap * sizeof(float4)));
  }
  s->have_state = true;
  return NBODY_HIP_OK;
}

// ---- one exchange + force evaluation: result in acc2 of every local rank -----------------------------------------
// drift_dt != nullptr: the drift of the step runs inside the box pass (x += v dt + a dt^2/2 with a = acc)
static int hash_force_phase(nbody_hip_sharded_hash* s, const float* drift_dt = nullptr) {
  const int W = s->W;
  const size_t nstats = (size_t)W * W + kHistCap;
  const bool rccl = s->comm->transport == NBODY_HIP_TRANSPORT_RCCL;
  Rccl* api = rccl ? rccl_load() : nullptr;
  if (rccl && !api) return NBH_FAIL((nbody_hip_status)NBODY_HIP_ERR_COMM, "librccl.so.1 is not loaded");
  // -- 1: local boxes, global box ----------------------------------------------------------------------------------
  for (auto& x : s->sh) {
    NBH_HIP(hipSetDevice(x.device));
    if (x.n && drift_dt) {
      if (int rc = nbody_hip_drift_bbox_packed(x.ctx, reinterpret_cast<nbody_float4*>(x.posm), reinterpret_cast<nbody_float4*>(x.vel),
                                               reinterpret_cast<nbody_float4*>(x.acc), x.n, *drift_dt, x.enc, x.box))
        return rc;
    } else if (x.n) {
      if (int rc = nbody_hip_bbox_packed(x.ctx, reinterpret_cast<nbody_float4*>(x.posm), x.n, x.box)) return rc;
    } else {
      hipLaunchKernelGGL(box_empty_kernel, dim3(1), dim3(64), 0, x.compute, x.box);
    }
    if (rccl) hipLaunchKernelGGL(box_flip_kernel, dim3(1), dim3(64), 0, x.compute, x.box);  // {lo, -hi}: ONE min all-reduce
    NBH_HIP(hipEventRecord(x.ev_a, x.compute));
  }
  if (rccl) {
    if (s->sh.size() > 1) NBH_NCCL(api, api->GroupStart());
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      NBH_NCCL(api, api->AllReduce(x.box, x.gbox, 6, ncclFloat, ncclMin, x.nccl, x.compute));
    }
    if (s->sh.size() > 1) NBH_NCCL(api, api->GroupEnd());
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      hipLaunchKernelGGL(box_flip_kernel, dim3(1), dim3(64), 0, x.compute, x.gbox);
    }
  } else {
    PeerPtrs pp;
    pp.n = W;
    for (int r = 0; r < W; r++) pp.p[r] = s->by_rank[r]->box;
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      for (auto& p : s->sh)
        if (&p != &x) NBH_HIP(hipStreamWaitEvent(x.compute, p.ev_a, 0));
      hipLaunchKernelGGL(box_reduce_kernel, dim3(1), dim3(64), 0, x.compute, pp, x.gbox);
    }
  }
  // -- 2, 3: partition pass, sum of the send matrices and layer histograms --------------------------------------------
  for (auto& x : s->sh) {
    NBH_HIP(hipSetDevice(x.device));
    if (x.n > x.part_cap) {
      NBH_HIP(hipStreamSynchronize(x.compute));
      (void)hipFree(x.rows); (void)hipFree(x.holes);
      x.rows = nullptr; x.holes = nullptr;
      const size_t cap = x.n + x.n / 8 + 1024;
      NBH_HIP(hipMalloc(reinterpret_cast<void**>(&x.rows), cap * 16 * sizeof(float)));
      NBH_HIP(hipMalloc(reinterpret_cast<void**>(&x.holes), cap * sizeof(int)));
      x.part_cap = cap;
    }
    if (int rc = nbody_hip_slab_partition_cuts(x.ctx, reinterpret_cast<nbody_float4*>(x.posm), reinterpret_cast<nbody_float4*>(x.vel),
                                               reinterpret_cast<nbody_float4*>(x.acc), x.gid, x.n, x.gbox, s->cell, W, x.rank,
                                               kHistCap, x.rows, x.holes, x.stats, x.stats + (size_t)W * W, x.info,
                                               s->zcuts.empty() ? nullptr : s->zcuts.data()))
      return rc;
    NBH_HIP(hipEventRecord(x.ev_b, x.compute));
  }
  if (rccl) {
    if (s->sh.size() > 1) NBH_NCCL(api, api->GroupStart());
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      NBH_NCCL(api, api->AllReduce(x.stats, x.stats_sum, nstats, ncclInt32, ncclSum, x.nccl, x.compute));
    }
    if (s->sh.size() > 1) NBH_NCCL(api, api->GroupEnd());
  } else {
    PeerPtrs pp;
    pp.n = W;
    for (int r = 0; r < W; r++) pp.p[r] = s->by_rank[r]->stats;
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      for (auto& p : s->sh)
        if (&p != &x) NBH_HIP(hipStreamWaitEvent(x.compute, p.ev_b, 0));
      hipLaunchKernelGGL(int_sum_kernel, dim3((unsigned)((nstats + kBlock - 1) / kBlock)), dim3(kBlock), 0, x.compute, pp,
                         (int)nstats, x.stats_sum);
    }
  }
  // -- 4: the host synchronisation ------------------------------------------------------------------------------------
  for (auto& x : s->sh) {
    NBH_HIP(hipSetDevice(x.device));
    NBH_HIP(hipMemcpyAsync(x.h_block, x.sum_block, (nstats + 10) * sizeof(int), hipMemcpyDeviceToHost, x.compute));
  }
  for (auto& x : s->sh) {
    NBH_HIP(hipSetDevice(x.device));
    NBH_HIP(hipStreamSynchronize(x.compute));
    NBH_HIP(hipStreamSynchronize(x.comm));
  }
  HShard& f = s->sh[0];
  const int gx = f.h_info[0], gy = f.h_info[1], gz = f.h_info[2], too_tall = f.h_info[3];
  if ((long long)gx * gy * gz > 100000000LL)  // force_spatial_hash.cu:252-254, on every rank alike
    return NBH_FAIL(NBODY_HIP_ERR_RESOURCE, "Spatial hash grid too large: reduce cell_size or bounding box");
  float bounds[6];
  int dims[3];
  grid_from_box(f.h_box, s->cell, bounds, dims);
  if (too_tall || dims[0] != gx || dims[1] != gy || dims[2] != gz)
    return NBH_FAIL(NBODY_HIP_ERR_RESOURCE, "sharded spatial hash: grid of %d x %d x %d cells is taller than %d layers "
                    "(or the host and device disagree on it)", gx, gy, gz, kHistCap);
  s->dims[0] = gx; s->dims[1] = gy; s->dims[2] = gz;
  const int* M = f.h_stats;                    // M[src * W + dst]
  const int* hist = f.h_stats + (size_t)W * W; // bodies per layer, all ranks
  const long long layer_cells = (long long)gx * gy;
  const SlabMap map = slab_map(s, bounds[2], gz);  // (with the cuts the partition pass just used)
  if (s->balance && W > 1) {
    // cuts for the NEXT evaluation, from this evaluation's histogram (the same on every rank): adopted when they would
    // take 5 % off the largest slab (a moved cut sends a layer of bodies to another rank: not for every wobble)
    long long cur_max = 0, cand_max = 0;
    for (int r = 0; r < W; r++) {
      long long c = 0;
      for (int z = map.lo[r]; z < map.hi[r]; z++) c += hist[z];
      cur_max = c > cur_max ? c : cur_max;
    }
    const std::vector<float> old = s->zcuts;
    balanced_cuts(s, hist, gz, bounds[2], s->n_total);
    const SlabMap cand = slab_map(s, bounds[2], gz);
    for (int r = 0; r < W; r++) {
      long long c = 0;
      for (int z = cand.lo[r]; z < cand.hi[r]; z++) c += hist[z];
      cand_max = c > cand_max ? c : cand_max;
    }
    if (!(cand_max * 100 < cur_max * 95)) s->zcuts = old;
  }
  s->migrated = 0;
  s->halo_bodies = 0;
  // -- 5: migration ---------------------------------------------------------------------------------------------------
  for (auto& x : s->sh) {
    const int r = x.rank;
    size_t n_new = 0, n_leave = 0, n_arrive = 0;
    for (int q = 0; q < W; q++) {
      n_new += (size_t)M[q * W + r];
      if (q != r) { n_leave += (size_t)M[r * W + q]; n_arrive += (size_t)M[q * W + r]; }
    }
    x.n_leave = n_leave;
    x.n_arrive = n_arrive;
    s->migrated += n_leave;
    if (int rc = body_arrays_reserve(x, (x.n > n_new ? x.n : n_new) + 1)) return rc;
    NBH_HIP(hipSetDevice(x.device));
    if (n_arrive > x.got_cap) {
      (void)hipFree(x.got);
      x.got = nullptr;
      x.got_cap = n_arrive + n_arrive / 4 + 1024;
      NBH_HIP(hipMalloc(reinterpret_cast<void**>(&x.got), x.got_cap * 16 * sizeof(float)));
    }
  }
  if (W > 1) {
    if (rccl) NBH_NCCL(api, api->GroupStart());
    for (auto& x : s->sh) {
      const int r = x.rank;
      NBH_HIP(hipSetDevice(x.device));
      size_t soff = 0;  // rows are grouped by new owner, ascending, own rank absent
      for (int q = 0; q < W; q++) {
        if (q == r) continue;
        const size_t cnt = (size_t)M[r * W + q];
        if (cnt) {
          if (rccl) {
            NBH_NCCL(api, api->Send(x.rows + soff * 16, cnt * 16, ncclFloat, q, x.nccl, x.comm));
          } else {
            HShard* d = s->by_rank[q];
            size_t doff = 0;  // arrivals are ordered by source rank
            for (int p = 0; p < r; p++)
              if (p != q) doff += (size_t)M[p * W + q];
            NBH_HIP(hipMemcpyAsync(d->got + doff * 16, x.rows + soff * 16, cnt * 16 * sizeof(float), hipMemcpyDeviceToDevice, x.comm));
          }
        }
        soff += cnt;
      }
      if (rccl) {
        size_t doff = 0;
        for (int p = 0; p < W; p++) {
          if (p == r) continue;
          const size_t cnt = (size_t)M[p * W + r];
          if (cnt) NBH_NCCL(api, api->Recv(x.got + doff * 16, cnt * 16, ncclFloat, p, x.nccl, x.comm));
          doff += cnt;
        }
      }
    }
    if (rccl) NBH_NCCL(api, api->GroupEnd());
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      NBH_HIP(hipEventRecord(x.ev_c, x.comm));
    }
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      if (rccl) {
        NBH_HIP(hipStreamWaitEvent(x.compute, x.ev_c, 0));
      } else {
        for (auto& p : s->sh) NBH_HIP(hipStreamWaitEvent(x.compute, p.ev_c, 0));
      }
      if (int rc = nbody_hip_slab_fill(x.ctx, x.got, x.n_arrive, x.holes, x.n_leave, x.n, reinterpret_cast<nbody_float4*>(x.posm),
                                       reinterpret_cast<nbody_float4*>(x.vel), reinterpret_cast<nbody_float4*>(x.acc), x.gid))
        return rc;
      x.n = x.n - x.n_leave + x.n_arrive;
    }
  }
  // -- 6, 7: own grids; the boundary layers leave for the neighbours ---------------------------------------------------
  // Dense slabs (every rank's grid carries per-cell start arrays) take the two-grid form: own x own while the boundary
  // layers travel, then each boundary layer against the neighbour's layer AS THE NEIGHBOUR'S GRID HOLDS IT -- bodies in
  // cell order plus the layer's start array (nbody_hip_grid_export_layer): the receiver bins nothing.  The decision is
  // taken from the send matrix and the slab map, which every process holds alike: all ranks take the same form.
  const size_t lb_len = (size_t)layer_cells + 1;
  bool all_dense = true;
  for (int r = 0; r < W; r++) {
    long long n_r = 0;
    for (int q = 0; q < W; q++) n_r += M[q * W + r];
    if (n_r > 0 && (long long)(map.hi[r] - map.lo[r]) * layer_cells > 4LL * n_r + 4096) all_dense = false;
  }
  struct Plan { int z_lo, z_hi, lower, upper; size_t in_lower; };
  std::vector<Plan> plan(s->sh.size());
  for (size_t k = 0; k < s->sh.size(); k++) {
    HShard& x = s->sh[k];
    Plan& pl = plan[k];
    const int r = x.rank;
    pl.z_lo = map.lo[r];
    pl.z_hi = map.hi[r];
    pl.lower = pl.upper = -1;
    x.n_head = x.n_tail = x.n_halo = 0;
    size_t in_lower = 0, in_upper = 0;
    if (W > 1 && pl.z_hi > pl.z_lo) {
      if (pl.z_lo > 0) {
        pl.lower = map.owner[(size_t)(pl.z_lo - 1)];
        x.n_head = (size_t)hist[pl.z_lo];
        in_lower = (size_t)hist[pl.z_lo - 1];
      }
      if (pl.z_hi < gz) {
        pl.upper = map.owner[(size_t)pl.z_hi];
        x.n_tail = (size_t)hist[pl.z_hi - 1];
        in_upper = (size_t)hist[pl.z_hi];
      }
    }
    pl.in_lower = in_lower;
    x.n_halo = in_lower + in_upper;
    s->halo_bodies += x.n_halo;
    NBH_HIP(hipSetDevice(x.device));
    if (x.n) {
      if (int rc = grid_for(x, &x.g_own, &x.g_own_cap, x.n, s->cell)) return rc;
      if (int rc = nbody_hip_grid_set_slab(x.g_own, pl.z_lo, pl.z_hi - pl.z_lo)) return rc;
      if (int rc = nbody_hip_grid_build_packed(x.g_own, reinterpret_cast<nbody_float4*>(x.posm), x.n, bounds)) return rc;
    }
    const size_t n_out = x.n_head + x.n_tail;
    if (n_out > x.halo_out_cap) {
      NBH_HIP(hipStreamSynchronize(x.comm));
      (void)hipFree(x.halo_out);
      x.halo_out = nullptr;
      x.halo_out_cap = n_out + n_out / 4 + 1024;
      NBH_HIP(hipMalloc(reinterpret_cast<void**>(&x.halo_out), x.halo_out_cap * sizeof(float4)));
    }
    if (x.n_halo > x.halo_in_cap) {
      NBH_HIP(hipStreamSynchronize(x.comm));
      (void)hipFree(x.halo_in);
      x.halo_in = nullptr;
      x.halo_in_cap = x.n_halo + x.n_halo / 4 + 1024;
      NBH_HIP(hipMalloc(reinterpret_cast<void**>(&x.halo_in), x.halo_in_cap * sizeof(float4)));
    }
    if (all_dense && W > 1 && lb_len > x.lb_cap) {
      NBH_HIP(hipStreamSynchronize(x.comm));
      NBH_HIP(hipStreamSynchronize(x.compute));
      (void)hipFree(x.halo_lb_out); (void)hipFree(x.halo_lb_in);
      x.halo_lb_out = x.halo_lb_in = nullptr;
      x.lb_cap = lb_len + lb_len / 4 + 64;
      NBH_HIP(hipMalloc(reinterpret_cast<void**>(&x.halo_lb_out), 2 * x.lb_cap * sizeof(int)));
      NBH_HIP(hipMalloc(reinterpret_cast<void**>(&x.halo_lb_in), 2 * x.lb_cap * sizeof(int)));
    }
    // its lowest layer is the head of the cell order, its highest the tail
    if (all_dense) {
      if (x.n_head)
        if (int rc = nbody_hip_grid_export_layer(x.g_own, pl.z_lo, reinterpret_cast<nbody_float4*>(x.halo_out), x.halo_lb_out)) return rc;
      if (x.n_tail)
        if (int rc = nbody_hip_grid_export_layer(x.g_own, pl.z_hi - 1, reinterpret_cast<nbody_float4*>(x.halo_out + x.n_head),
                                                 x.halo_lb_out + x.lb_cap))
          return rc;
    } else {
      if (x.n_head)
        if (int rc = nbody_hip_grid_sorted_bodies(x.g_own, 0, x.n_head, reinterpret_cast<nbody_float4*>(x.halo_out))) return rc;
      if (x.n_tail)
        if (int rc = nbody_hip_grid_sorted_bodies(x.g_own, x.n - x.n_tail, x.n_tail, reinterpret_cast<nbody_float4*>(x.halo_out + x.n_head)))
          return rc;
    }
    NBH_HIP(hipEventRecord(x.ev_d, x.compute));
    NBH_HIP(hipStreamWaitEvent(x.comm, x.ev_d, 0));
  }
  s->two_grid = all_dense ? 1 : 0;
  if (W > 1) {
    if (rccl) NBH_NCCL(api, api->GroupStart());
    for (size_t k = 0; k < s->sh.size(); k++) {
      HShard& x = s->sh[k];
      const Plan& pl = plan[k];
      NBH_HIP(hipSetDevice(x.device));
      // receive layout: the lower neighbour's layer first, then the upper neighbour's (source-rank order); start arrays:
      // slot 0 = the layer below my slab, slot 1 = the layer above it
      const size_t in_lower = pl.in_lower;
      if (rccl) {
        if (x.n_head) NBH_NCCL(api, api->Send(x.halo_out, x.n_head * 4, ncclFloat, pl.lower, x.nccl, x.comm));
        if (x.n_tail) NBH_NCCL(api, api->Send(x.halo_out + x.n_head, x.n_tail * 4, ncclFloat, pl.upper, x.nccl, x.comm));
        if (in_lower) NBH_NCCL(api, api->Recv(x.halo_in, in_lower * 4, ncclFloat, pl.lower, x.nccl, x.comm));
        if (x.n_halo - in_lower) NBH_NCCL(api, api->Recv(x.halo_in + in_lower, (x.n_halo - in_lower) * 4, ncclFloat, pl.upper, x.nccl, x.comm));
        if (all_dense) {
          if (x.n_head) NBH_NCCL(api, api->Send(x.halo_lb_out, lb_len, ncclInt32, pl.lower, x.nccl, x.comm));
          if (x.n_tail) NBH_NCCL(api, api->Send(x.halo_lb_out + x.lb_cap, lb_len, ncclInt32, pl.upper, x.nccl, x.comm));
          if (in_lower) NBH_NCCL(api, api->Recv(x.halo_lb_in, lb_len, ncclInt32, pl.lower, x.nccl, x.comm));
          if (x.n_halo - in_lower) NBH_NCCL(api, api->Recv(x.halo_lb_in + x.lb_cap, lb_len, ncclInt32, pl.upper, x.nccl, x.comm));
        }
      } else {
        if (x.n_head) {  // my lowest layer is the UPPER halo of the owner of the layer below: behind its lower halo
          HShard* d = s->by_rank[pl.lower];
          const int dz_lo = map.lo[d->rank];
          const size_t d_in_lower = dz_lo > 0 ? (size_t)hist[dz_lo - 1] : 0;
          NBH_HIP(hipMemcpyAsync(d->halo_in + d_in_lower, x.halo_out, x.n_head * sizeof(float4), hipMemcpyDeviceToDevice, x.comm));
          if (all_dense)
            NBH_HIP(hipMemcpyAsync(d->halo_lb_in + d->lb_cap, x.halo_lb_out, lb_len * sizeof(int), hipMemcpyDeviceToDevice, x.comm));
        }
        if (x.n_tail) {  // my highest layer is the LOWER halo of the owner of the layer above: at the front
          HShard* d = s->by_rank[pl.upper];
          NBH_HIP(hipMemcpyAsync(d->halo_in, x.halo_out + x.n_head, x.n_tail * sizeof(float4), hipMemcpyDeviceToDevice, x.comm));
          if (all_dense)
            NBH_HIP(hipMemcpyAsync(d->halo_lb_in, x.halo_lb_out + x.lb_cap, lb_len * sizeof(int), hipMemcpyDeviceToDevice, x.comm));
        }
      }
    }
    if (rccl) NBH_NCCL(api, api->GroupEnd());
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      NBH_HIP(hipEventRecord(x.ev_c, x.comm));
    }
  }
  // -- 8: own x own while the halo layers are in flight ------------------------------------------------------------------
  for (size_t k = 0; k < s->sh.size(); k++) {
    HShard& x = s->sh[k];
    const Plan& pl = plan[k];
    if (x.n && all_dense)
      if (int rc = nbody_hip_grid_forces_pair_packed(x.g_own, x.g_own, pl.z_lo, pl.z_hi - pl.z_lo, s->cutoff, s->G, s->eps,
                                                     reinterpret_cast<nbody_float4*>(x.acc2), 0))
        return rc;
  }
  // -- 9: the boundary layers against the received layers ----------------------------------------------------------------
  for (size_t k = 0; k < s->sh.size(); k++) {
    HShard& x = s->sh[k];
    const Plan& pl = plan[k];
    if (!x.n) continue;
    NBH_HIP(hipSetDevice(x.device));
    if (W > 1) {
      if (rccl) {
        NBH_HIP(hipStreamWaitEvent(x.compute, x.ev_c, 0));
      } else {
        if (pl.lower >= 0) NBH_HIP(hipStreamWaitEvent(x.compute, s->by_rank[pl.lower]->ev_c, 0));
        if (pl.upper >= 0) NBH_HIP(hipStreamWaitEvent(x.compute, s->by_rank[pl.upper]->ev_c, 0));
      }
    }
    if (all_dense) {
      // my lowest layer against the layer below the slab, my highest against the layer above it (the same layer twice when
      // the slab is one layer thick); an empty received layer has nothing to add
      if (pl.in_lower)
        if (int rc = nbody_hip_grid_forces_layer_packed(x.g_own, pl.z_lo, reinterpret_cast<nbody_float4*>(x.halo_in), x.halo_lb_in,
                                                        pl.z_lo - 1, s->cutoff, s->G, s->eps, reinterpret_cast<nbody_float4*>(x.acc2), 1))
          return rc;
      if (x.n_halo - pl.in_lower)
        if (int rc = nbody_hip_grid_forces_layer_packed(x.g_own, pl.z_hi - 1, reinterpret_cast<nbody_float4*>(x.halo_in + pl.in_lower),
                                                        x.halo_lb_in + x.lb_cap, pl.z_hi, s->cutoff, s->G, s->eps,
                                                        reinterpret_cast<nbody_float4*>(x.acc2), 1))
          return rc;
    } else {  // too sparse for the per-cell start arrays: one grid over own + halo bodies
      const size_t nall = x.n + x.n_halo;
      if (nall > x.cat_cap) {
        NBH_HIP(hipStreamSynchronize(x.compute));
        (void)hipFree(x.cat); (void)hipFree(x.cat_acc);
        x.cat = x.cat_acc = nullptr;
        x.cat_cap = nall + nall / 4 + 1024;
        NBH_HIP(hipMalloc(reinterpret_cast<void**>(&x.cat), x.cat_cap * sizeof(float4)));
        NBH_HIP(hipMalloc(reinterpret_cast<void**>(&x.cat_acc), x.cat_cap * sizeof(float4)));
      }
      NBH_HIP(hipMemcpyAsync(x.cat, x.posm, x.n * sizeof(float4), hipMemcpyDeviceToDevice, x.compute));
      if (x.n_halo) NBH_HIP(hipMemcpyAsync(x.cat + x.n, x.halo_in, x.n_halo * sizeof(float4), hipMemcpyDeviceToDevice, x.compute));
      if (int rc = grid_for(x, &x.g_one, &x.g_one_cap, nall, s->cell)) return rc;
      if (int rc = nbody_hip_grid_set_slab(x.g_one, 0, 0)) return rc;
      if (int rc = nbody_hip_grid_build_packed(x.g_one, reinterpret_cast<nbody_float4*>(x.cat), nall, bounds)) return rc;
      if (int rc = nbody_hip_grid_compute_forces_packed(x.g_one, s->cutoff, s->G, s->eps, reinterpret_cast<nbody_float4*>(x.cat_acc))) return rc;
      NBH_HIP(hipMemcpyAsync(x.acc2, x.cat_acc, x.n * sizeof(float4), hipMemcpyDeviceToDevice, x.compute));
    }
  }
  return NBODY_HIP_OK;
}

static void adopt_new_accelerations(nbody_hip_sharded_hash* s) {
  for (auto& x : s->sh) std::swap(x.acc, x.acc2);
}

extern "C" int nbody_hip_sharded_hash_forces(nbody_hip_sharded_hash* s) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!s) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null system");
  if (!s->have_state) return NBH_FAIL(NBODY_HIP_ERR_STATE, "no body state (call nbody_hip_sharded_hash_set_state first)");
  if (int rc = hash_force_phase(s)) return rc;
  adopt_new_accelerations(s);
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_sharded_hash_step(nbody_hip_sharded_hash* s, float dt, int steps) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!s) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null system");
  if (!s->have_state) return NBH_FAIL(NBODY_HIP_ERR_STATE, "no body state (call nbody_hip_sharded_hash_set_state first)");
  if (!(dt > 0.0f) || !(dt <= 1.0f)) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Time step must be in range (0, 1]");
  for (int k = 0; k < steps; k++) {
    // the drift runs inside the force phase's box pass; acc (= a_old) migrates with the bodies there
    if (int rc = hash_force_phase(s, &dt)) return rc;
    for (auto& x : s->sh)
      if (x.n)
        if (int rc = nbody_hip_kick_packed(x.ctx, reinterpret_cast<nbody_float4*>(x.vel), reinterpret_cast<nbody_float4*>(x.acc),
                                           reinterpret_cast<nbody_float4*>(x.acc2), x.n, dt))
          return rc;
    adopt_new_accelerations(s);
  }
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_sharded_hash_synchronize(nbody_hip_sharded_hash* s) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!s) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null system");
  for (auto& x : s->sh) {
    NBH_HIP(hipSetDevice(x.device));
    NBH_HIP(hipStreamSynchronize(x.comm));
    NBH_HIP(hipStreamSynchronize(x.compute));
  }
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_sharded_hash_info(const nbody_hip_sharded_hash* s, int dims[3], int* two_grid,
                                           unsigned long long* migrated, unsigned long long* halo_bodies,
                                           unsigned long long local_counts[NBODY_HIP_MAX_RANKS]) {
  if (!s) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null system");
  if (dims) for (int a = 0; a < 3; a++) dims[a] = s->dims[a];
  if (two_grid) *two_grid = s->two_grid;
  if (migrated) *migrated = s->migrated;
  if (halo_bodies) *halo_bodies = s->halo_bodies;
  if (local_counts)
    for (size_t k = 0; k < s->sh.size(); k++) local_counts[k] = s->sh[k].n;
  return NBODY_HIP_OK;
}

// host arrays of n_total floats indexed by the bodies' GLOBAL ids; every process fills the rows of its local ranks
extern "C" int nbody_hip_sharded_hash_get_state(nbody_hip_sharded_hash* s, float* x, float* y, float* z, float* vx, float* vy,
                                                float* vz, float* ax, float* ay, float* az) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!s) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null system");
  if (!s->have_state) return NBH_FAIL(NBODY_HIP_ERR_STATE, "no body state");
  if (int rc = nbody_hip_sharded_hash_synchronize(s)) return rc;
  for (auto& sh : s->sh) {
    if (!sh.n) continue;
    NBH_HIP(hipSetDevice(sh.device));
    std::vector<int> id(sh.n);
    std::vector<float4> h(sh.n);
    NBH_HIP(hipMemcpy(id.data(), sh.gid, sh.n * sizeof(int), hipMemcpyDeviceToHost));
    auto scatter = [&](float* a, float* b, float* c, const float4* dev) -> int {
      if (!a && !b && !c) return NBODY_HIP_OK;
      NBH_HIP(hipMemcpy(h.data(), dev, sh.n * sizeof(float4), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < sh.n; i++) {
        const size_t g = (size_t)id[i];
        if (g >= s->n_total) return NBH_FAIL(NBODY_HIP_ERR_STATE, "body id %zu out of range", g);
        if (a) a[g] = h[i].x;
        if (b) b[g] = h[i].y;
        if (c) c[g] = h[i].z;
      }
      return NBODY_HIP_OK;
    };
    if (int rc = scatter(x, y, z, sh.posm)) return rc;
    if (int rc = scatter(vx, vy, vz, sh.vel)) return rc;
    if (int rc = scatter(ax, ay, az, sh.acc)) return rc;
  }
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_sharded_hash_time_steps(nbody_hip_sharded_hash* s, float dt, int warmup, int steps, float* ms_per_step) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!s || !ms_per_step) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (steps <= 0 || warmup < 0) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "bad timing arguments");
  if (int rc = nbody_hip_sharded_hash_step(s, dt, warmup)) return rc;
  if (int rc = nbody_hip_sharded_hash_synchronize(s)) return rc;
  HShard& x = s->sh[0];
  NBH_HIP(hipSetDevice(x.device));
  NBH_HIP(hipEventRecord(x.t0, x.compute));
  if (int rc = nbody_hip_sharded_hash_step(s, dt, steps)) return rc;
  NBH_HIP(hipSetDevice(x.device));
  NBH_HIP(hipEventRecord(x.t1, x.compute));
  if (int rc = nbody_hip_sharded_hash_synchronize(s)) return rc;
  float ms = 0.f;
  NBH_HIP(hipEventElapsedTime(&ms, x.t0, x.t1));
  *ms_per_step = ms / (float)steps;
  return NBODY_HIP_OK;
}
