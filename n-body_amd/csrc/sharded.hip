// sharded.hip -- the multi-GPU part of the C ABI (include/nbody_hip_comm.h): communicators (RCCL over xGMI, or
// direct peer copies when one process drives all devices) and the sharded Direct N^2 system of BASELINE config 3.
// The reference is single-GPU; what this spreads over the ranks is launchDirectForceKernel
// (src/cuda/force_direct.cu:88-98) and the step of Integrator::integrate (src/cuda/integrator.cu:224-238).
//
// Everything here is host orchestration over the entry points of nbody_hip.h plus one small kernel (sum of the
// received reaction blocks fused with the kick).  One host thread issues the work of all local ranks PHASE BY
// PHASE; every cross-rank dependency is a HIP event recorded in an earlier phase, so there is no circular wait.
//
//   phase        compute stream of rank r                         comm stream of rank r
//   drift        x += v dt + a dt^2/2 (own rows of posm_all)
//                record ev_ready
//   gather                                                        wait ev_ready ; P2P: wait ev_read of every peer
//                                                                 (its kernels of the last step still read its
//                                                                 posm_all), push own rows into every peer's
//                                                                 posm_all ; RCCL: in-place ncclAllGather
//                                                                 record ev_gather
//   own x own    symmetric kernel -> mine        (overlaps the gather)
//   pairs        wait ev_gather (P2P: of every rank) ; per task: two-set kernel, action added to mine,
//                reaction -> the task's send block ; record ev_read
//   exchange                                                      wait ev_read ; P2P: copy each send block into
//                                                                 the owner's receive block ; RCCL: grouped
//                                                                 ncclSend / ncclRecv ; record ev_sent
//   finalize     wait ev_sent (P2P: of every sender) ; a_new = mine + received blocks in rank-distance order ;
//                v += (a_old + a_new) dt / 2
// WAR hazards: a receive block is rewritten by a sender only after the sender passed "pairs" of the new step, i.e.
// after it saw the owner's ev_gather of the new step, which the owner records after its own finalize of the old one.

#include <cstring>
#include <vector>

#include "comm.h"

namespace nbh {

// ---- RCCL, resolved at run time -----------------------------------------------------------------------------------
Rccl* rccl_load() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.so ? &r : nullptr;
  tried = true;
  // the copy already mapped into the process (PyTorch ships one with the same soname) wins; then the loader path
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* so = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);
  for (int k = 0; !so && k < 3; k++) so = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
  if (!so) return nullptr;
  bool ok = true;
#define NBH_SYM(field, name)                                              \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(so, name));         \
  ok = ok && r.field != nullptr
  NBH_SYM(GetUniqueId, "ncclGetUniqueId");
  NBH_SYM(CommInitRank, "ncclCommInitRank");
  NBH_SYM(CommInitAll, "ncclCommInitAll");
  NBH_SYM(CommDestroy, "ncclCommDestroy");
  NBH_SYM(AllGather, "ncclAllGather");
  NBH_SYM(AllReduce, "ncclAllReduce");
  NBH_SYM(Send, "ncclSend");
  NBH_SYM(Recv, "ncclRecv");
  NBH_SYM(GroupStart, "ncclGroupStart");
  NBH_SYM(GroupEnd, "ncclGroupEnd");
  NBH_SYM(GetErrorString, "ncclGetErrorString");
#undef NBH_SYM
  if (!ok) {
    dlclose(so);
    return nullptr;
  }
  r.so = so;
  return &r;
}

static_assert(sizeof(nbody_hip_comm_id) == sizeof(ncclUniqueId), "nbody_hip_comm_id must hold an ncclUniqueId");

// a_new = mine + the received reaction blocks (fixed order) ; optional kick
struct InBlocks {
  int n;
  const float4* buf[NBODY_HIP_MAX_RANKS / 2];
  unsigned int j0[NBODY_HIP_MAX_RANKS / 2], j1[NBODY_HIP_MAX_RANKS / 2];
};
__global__ __launch_bounds__(kBlock) void shard_finalize_kernel(const float4* __restrict__ mine, InBlocks in, int n,
                                                                float4* __restrict__ acc_new,
                                                                float4* __restrict__ vel,
                                                                const float4* __restrict__ acc_old, float half_dt,
                                                                int kick) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float4 a = mine[i];
  for (int k = 0; k < in.n; k++) {
    if ((unsigned)i >= in.j0[k] && (unsigned)i < in.j1[k]) {
      const float4 b = in.buf[k][(unsigned)i - in.j0[k]];
      a.x += b.x; a.y += b.y; a.z += b.z;
    }
  }
  a.w = 0.f;
  acc_new[i] = a;
  if (kick) {
    float4 v = vel[i];
    const float4 o = acc_old[i];
    v.x = kick1(v.x, o.x, a.x, half_dt);
    v.y = kick1(v.y, o.y, a.y, half_dt);
    v.z = kick1(v.z, o.z, a.z, half_dt);
    vel[i] = v;
  }
}

}  // namespace nbh

using namespace nbh;

// ---- communicators ------------------------------------------------------------------------------------------------
extern "C" int nbody_hip_shard_bounds(size_t n, int world, int rank, size_t* shard, size_t* lo, size_t* hi) {
  if (world < 1 || rank < 0 || rank >= world) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "rank %d outside world %d", rank, world);
  const size_t s = (n + (size_t)world - 1) / (size_t)world;
  const size_t l = (size_t)rank * s < n ? (size_t)rank * s : n;
  const size_t h = l + s < n ? l + s : n;
  if (shard) *shard = s;
  if (lo) *lo = l;
  if (hi) *hi = h;
  return NBODY_HIP_OK;
}

// Rank r takes the ring neighbours r+1 .. r+(W-1)/2 whole; for even W the antipodal rectangle is cut in half: the
// lower rank takes its first half x the whole partner, the upper rank its whole shard x the partner's second half.
extern "C" int nbody_hip_pair_schedule(int world, int rank, size_t S, size_t rows[][5], int max_rows) {
  if (world < 1 || rank < 0 || rank >= world) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "rank %d outside world %d", rank, world);
  int k = 0;
  auto put = [&](size_t i0, size_t i1, int sh, size_t j0, size_t j1) {
    if (rows && k < max_rows) {
      rows[k][0] = i0; rows[k][1] = i1; rows[k][2] = (size_t)sh; rows[k][3] = j0; rows[k][4] = j1;
    }
    k++;
  };
  for (int d = 1; d <= (world - 1) / 2; d++) put(0, S, (rank + d) % world, 0, S);
  if (world % 2 == 0 && world > 1) {
    const int sh = (rank + world / 2) % world;
    const size_t h = S / 2;
    if (rank < world / 2) {
      if (h > 0) put(0, h, sh, 0, S);
    } else {
      put(0, S, sh, h, S);
    }
  }
  if (rows && k > max_rows) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "schedule has %d rows, room for %d", k, max_rows);
  return k;
}

extern "C" int nbody_hip_comm_init_all(int ndev, const int* devices, int transport, nbody_hip_comm** out) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null output pointer");
  *out = nullptr;
  if (ndev < 1 || ndev > NBODY_HIP_MAX_RANKS)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "ndev must be in [1, %d]", NBODY_HIP_MAX_RANKS);
  if (transport != NBODY_HIP_TRANSPORT_P2P && transport != NBODY_HIP_TRANSPORT_RCCL)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "unknown transport %d", transport);
  const int have = nbody_hip_device_count();
  if (have <= 0) return NBH_FAIL(NBODY_HIP_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
  std::vector<int> dev(ndev);
  for (int r = 0; r < ndev; r++) {
    dev[r] = devices ? devices[r] : r;
    if (dev[r] < 0 || dev[r] >= have)
      return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "device %d of rank %d out of range [0,%d)", dev[r], r, have);
    if (transport == NBODY_HIP_TRANSPORT_RCCL)
      for (int q = 0; q < r; q++)
        if (dev[q] == dev[r]) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "RCCL needs distinct devices (device %d twice)", dev[r]);
  }
  // peer access between distinct devices (both transports copy between local ranks)
  for (int r = 0; r < ndev; r++)
    for (int q = 0; q < ndev; q++) {
      if (dev[r] == dev[q]) continue;
      NBH_HIP(hipSetDevice(dev[r]));
      int can = 0;
      NBH_HIP(hipDeviceCanAccessPeer(&can, dev[r], dev[q]));
      if (!can) return NBH_FAIL(NBODY_HIP_ERR_DEVICE, "device %d cannot access device %d", dev[r], dev[q]);
      const hipError_t e = hipDeviceEnablePeerAccess(dev[q], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
        return NBH_FAIL(NBODY_HIP_ERR_DEVICE, "hipDeviceEnablePeerAccess(%d -> %d): %s", dev[r], dev[q], hipGetErrorString(e));
      (void)hipGetLastError();
    }
  nbody_hip_comm* c = new nbody_hip_comm();
  c->world = ndev;
  c->transport = transport;
  c->local.resize(ndev);
  for (int r = 0; r < ndev; r++) {
    c->local[r].rank = r;
    c->local[r].device = dev[r];
  }
  if (transport == NBODY_HIP_TRANSPORT_RCCL) {
    Rccl* api = rccl_load();
    if (!api) {
      delete c;
      return NBH_FAIL((nbody_hip_status)NBODY_HIP_ERR_COMM, "librccl.so.1 cannot be loaded: %s", dlerror());
    }
    std::vector<ncclComm_t> comms(ndev);
    const ncclResult_t r_ = api->CommInitAll(comms.data(), ndev, dev.data());
    if (r_ != ncclSuccess) {
      delete c;
      return NBH_FAIL((nbody_hip_status)NBODY_HIP_ERR_COMM, "ncclCommInitAll: %s", api->GetErrorString(r_));
    }
    for (int r = 0; r < ndev; r++) c->local[r].nccl = comms[r];
  }
  *out = c;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_comm_unique_id(nbody_hip_comm_id* id) {
  if (!id) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  Rccl* api = rccl_load();
  if (!api) return NBH_FAIL((nbody_hip_status)NBODY_HIP_ERR_COMM, "librccl.so.1 cannot be loaded: %s", dlerror());
  ncclUniqueId u;
  NBH_NCCL(api, api->GetUniqueId(&u));
  memcpy(id->bytes, &u, sizeof(u));
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_comm_init_rank(int device, int rank, int world, const nbody_hip_comm_id* id,
                                        nbody_hip_comm** out) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!out || !id) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  *out = nullptr;
  if (world < 1 || world > NBODY_HIP_MAX_RANKS || rank < 0 || rank >= world)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "rank %d / world %d out of range (at most %d ranks)", rank, world, NBODY_HIP_MAX_RANKS);
  const int have = nbody_hip_device_count();
  if (have <= 0) return NBH_FAIL(NBODY_HIP_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= have) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "device %d out of range [0,%d)", device, have);
  Rccl* api = rccl_load();
  if (!api) return NBH_FAIL((nbody_hip_status)NBODY_HIP_ERR_COMM, "librccl.so.1 cannot be loaded: %s", dlerror());
  NBH_HIP(hipSetDevice(device));
  ncclUniqueId u;
  memcpy(&u, id->bytes, sizeof(u));
  ncclComm_t comm = nullptr;
  NBH_NCCL(api, api->CommInitRank(&comm, world, u, rank));
  nbody_hip_comm* c = new nbody_hip_comm();
  c->world = world;
  c->transport = NBODY_HIP_TRANSPORT_RCCL;
  c->local.resize(1);
  c->local[0].rank = rank;
  c->local[0].device = device;
  c->local[0].nccl = comm;
  *out = c;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_comm_info(const nbody_hip_comm* c, int* world, int* nlocal, int* transport,
                                   int local_ranks[NBODY_HIP_MAX_RANKS]) {
  if (!c) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null communicator");
  if (world) *world = c->world;
  if (nlocal) *nlocal = (int)c->local.size();
  if (transport) *transport = c->transport;
  if (local_ranks)
    for (size_t k = 0; k < c->local.size(); k++) local_ranks[k] = c->local[k].rank;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_comm_destroy(nbody_hip_comm* c) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!c) return NBODY_HIP_OK;
  NBH_DESTROY_BEGIN
  if (c->transport == NBODY_HIP_TRANSPORT_RCCL) {
    Rccl* api = rccl_load();
    for (auto& m : c->local)
      if (m.nccl && api) {
        (void)hipSetDevice(m.device);
        (void)api->CommDestroy(m.nccl);
      }
  }
  delete c;
  NBH_DESTROY_END
}

// ---- the sharded Direct system --------------------------------------------------------------------------------------
namespace {

struct Task {       // own bodies [i0, i1) x rows [j0, j1) of shard `sh`; the reactions go to `send` (row 0 = body j0)
  size_t i0, i1, j0, j1;
  int sh;
  float4* send = nullptr;
};
struct Incoming {   // reactions on my rows [j0, j1) computed by rank `from`
  int from;
  size_t j0, j1;
  float4* buf = nullptr;
};

struct Shard {
  int rank = 0, device = 0;
  nbody_hip_ctx* ctx = nullptr;
  hipStream_t compute = nullptr, comm = nullptr;
  ncclComm_t nccl = nullptr;
  float4 *posm_all = nullptr, *vel = nullptr, *acc[2] = {nullptr, nullptr}, *mine = nullptr, *stage = nullptr;
  double* esum = nullptr;  // 2 W doubles (energies of every rank, RCCL all-gather)
  std::vector<Task> tasks;
  std::vector<Incoming> incoming;
  hipEvent_t ev_ready = nullptr, ev_gather = nullptr, ev_read = nullptr, ev_sent = nullptr, ev_aux = nullptr;
  hipEvent_t t0 = nullptr, t1 = nullptr;
};

}  // namespace

struct nbody_hip_sharded_direct {
  nbody_hip_comm* comm = nullptr;
  size_t n = 0, S = 0;
  int W = 1;
  float G = 1.f, eps = 0.f, eps2 = 0.f;
  bool pair_mode = true;
  int cur = 0;  // acc[cur] holds the current accelerations on every shard
  bool have_state = false;
  std::vector<Shard> sh;
  Shard* by_rank[NBODY_HIP_MAX_RANKS] = {};
  float4* own(Shard& s) const { return s.posm_all + (size_t)s.rank * S; }
};

static void shard_release(Shard& s) {
  (void)hipSetDevice(s.device);
  if (s.compute) (void)hipStreamSynchronize(s.compute);
  if (s.comm) (void)hipStreamSynchronize(s.comm);
  (void)hipFree(s.posm_all); (void)hipFree(s.vel); (void)hipFree(s.acc[0]); (void)hipFree(s.acc[1]);
  (void)hipFree(s.mine); (void)hipFree(s.stage); (void)hipFree(s.esum);
  for (auto& t : s.tasks) (void)hipFree(t.send);
  for (auto& in : s.incoming) (void)hipFree(in.buf);
  for (hipEvent_t e : {s.ev_ready, s.ev_gather, s.ev_read, s.ev_sent, s.ev_aux, s.t0, s.t1})
    if (e) (void)hipEventDestroy(e);
  if (s.ctx) (void)nbody_hip_ctx_destroy(s.ctx);
  if (s.compute) (void)hipStreamDestroy(s.compute);
  if (s.comm) (void)hipStreamDestroy(s.comm);
}

extern "C" int nbody_hip_sharded_direct_destroy(nbody_hip_sharded_direct* s) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!s) return NBODY_HIP_OK;
  NBH_DESTROY_BEGIN
  for (auto& x : s->sh) shard_release(x);
  delete s;
  NBH_DESTROY_END
}

extern "C" int nbody_hip_sharded_direct_create(nbody_hip_comm* comm, size_t n, float G, float eps,
                                               nbody_hip_sharded_direct** out) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!comm || !out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  *out = nullptr;
  if (n == 0) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Particle count must be greater than 0");
  if (n > 100000000u) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Particle count exceeds maximum supported (100M)");
  if (!(G > 0.0f) || !(G < INFINITY)) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Gravitational constant must be positive and finite");
  if (!(eps >= 0.0f) || !(eps < INFINITY)) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Softening parameter must be non-negative and finite");
  nbody_hip_sharded_direct* s = new nbody_hip_sharded_direct();
  s->comm = comm;
  s->n = n;
  s->W = comm->world;
  s->S = (n + (size_t)s->W - 1) / (size_t)s->W;
  s->G = G;
  s->eps = eps;
  s->eps2 = eps * eps;  // force_calculator.hpp:52-57
  s->pair_mode = s->eps2 >= 1e-12f && s->W > 1;
  const size_t S = s->S, all = S * (size_t)s->W;
  s->sh.resize(comm->local.size());
  auto fail = [&](int rc) {
    nbody_hip_sharded_direct_destroy(s);
    return rc;
  };
  for (size_t k = 0; k < comm->local.size(); k++) {
    Shard& x = s->sh[k];
    x.rank = comm->local[k].rank;
    x.device = comm->local[k].device;
    x.nccl = comm->local[k].nccl;
    s->by_rank[x.rank] = &x;
    hipError_t e = hipSetDevice(x.device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&x.compute, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&x.comm, hipStreamNonBlocking);
    for (hipEvent_t* ev : {&x.ev_ready, &x.ev_gather, &x.ev_read, &x.ev_sent, &x.ev_aux})
      if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&x.t0);
    if (e == hipSuccess) e = hipEventCreate(&x.t1);
    auto dmalloc = [&](float4** p, size_t rows) {
      if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(p), rows * sizeof(float4));
      if (e == hipSuccess) e = hipMemset(*p, 0, rows * sizeof(float4));
    };
    dmalloc(&x.posm_all, all);
    dmalloc(&x.vel, S);
    dmalloc(&x.acc[0], S);
    dmalloc(&x.acc[1], S);
    dmalloc(&x.mine, S);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&x.esum), 2 * (size_t)s->W * sizeof(double));
    if (e == hipSuccess && s->pair_mode) {
      size_t rows[NBODY_HIP_MAX_RANKS / 2 + 1][5];
      const int nt = nbody_hip_pair_schedule(s->W, x.rank, S, rows, NBODY_HIP_MAX_RANKS / 2 + 1);
      for (int t = 0; t < nt; t++) {
        Task task;
        task.i0 = rows[t][0]; task.i1 = rows[t][1]; task.sh = (int)rows[t][2]; task.j0 = rows[t][3]; task.j1 = rows[t][4];
        dmalloc(&task.send, task.j1 - task.j0);
        x.tasks.push_back(task);
      }
      // who sends me what: rank q's tasks that name me, in order of ring distance (a fixed summation order)
      for (int d = 1; d < s->W; d++) {
        const int q = (x.rank - d + s->W) % s->W;
        size_t qrows[NBODY_HIP_MAX_RANKS / 2 + 1][5];
        const int nq = nbody_hip_pair_schedule(s->W, q, S, qrows, NBODY_HIP_MAX_RANKS / 2 + 1);
        for (int t = 0; t < nq; t++)
          if ((int)qrows[t][2] == x.rank) {
            Incoming in;
            in.from = q; in.j0 = qrows[t][3]; in.j1 = qrows[t][4];
            dmalloc(&in.buf, in.j1 - in.j0);
            x.incoming.push_back(in);
          }
      }
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail(NBH_FAIL(e == hipErrorOutOfMemory ? NBODY_HIP_ERR_RESOURCE : NBODY_HIP_ERR_DEVICE,
                           "sharded Direct system, rank %d on device %d: %s", x.rank, x.device, hipGetErrorString(e)));
    }
    if (int rc = nbody_hip_ctx_create(&x.ctx, x.device, x.compute)) return fail(rc);
    if ((int)x.incoming.size() > NBODY_HIP_MAX_RANKS / 2)
      return fail(NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "too many incoming blocks"));
  }
  *out = s;
  return NBODY_HIP_OK;
}

// every local rank: upload its range of the whole-system host arrays
extern "C" int nbody_hip_sharded_direct_set_state(nbody_hip_sharded_direct* s, const float* x, const float* y,
                                                  const float* z, const float* mass, const float* vx,
                                                  const float* vy, const float* vz) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!s) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null system");
  if (!x || !y || !z || !mass) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if ((vx || vy || vz) && !(vx && vy && vz)) return NBH_FAIL(NBODY_HIP_ERR_STATE, "velocity arrays: all three or none");
  const size_t S = s->S;
  std::vector<float4> p(S), v(S);
  for (auto& sh : s->sh) {
    size_t lo, hi;
    nbody_hip_shard_bounds(s->n, s->W, sh.rank, nullptr, &lo, &hi);
    for (size_t i = 0; i < S; i++) {
      const size_t g = lo + i;
      const bool real = g < hi;
      p[i] = real ? make_float4(x[g], y[g], z[g], mass[g]) : make_float4(0.f, 0.f, 0.f, 0.f);  // padding: zero mass
      v[i] = real && vx ? make_float4(vx[g], vy[g], vz[g], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    NBH_HIP(hipSetDevice(sh.device));
    NBH_HIP(hipStreamSynchronize(sh.compute));
    NBH_HIP(hipStreamSynchronize(sh.comm));
    NBH_HIP(hipMemcpy(s->own(sh), p.data(), S * sizeof(float4), hipMemcpyHostToDevice));
    NBH_HIP(hipMemcpy(sh.vel, v.data(), S * sizeof(float4), hipMemcpyHostToDevice));
    NBH_HIP(hipMemset(sh.acc[0], 0, S * sizeof(float4)));
    NBH_HIP(hipMemset(sh.acc[1], 0, S * sizeof(float4)));
  }
  s->cur = 0;
  s->have_state = true;
  return NBODY_HIP_OK;
}

// ---- one exchange + force evaluation ------------------------------------------------------------------------------
// gathered: posm_all already holds every shard on every rank (compute_forces); kick: fuse v += (a_old + a_new) h
// into the finalize, a_old = acc[cur], result -> acc[1 - cur]; otherwise result -> acc[out_slot].
static int force_phase(nbody_hip_sharded_direct* s, bool gathered, bool kick, float half_dt, int out_slot) {
  const size_t S = s->S, all = S * (size_t)s->W;
  const int W = s->W;
  const bool rccl = s->comm->transport == NBODY_HIP_TRANSPORT_RCCL;
  Rccl* api = rccl ? rccl_load() : nullptr;
  if (rccl && !api) return NBH_FAIL((nbody_hip_status)NBODY_HIP_ERR_COMM, "librccl.so.1 is not loaded");
  // -- gather ---------------------------------------------------------------------------------------------------
  if (!gathered) {
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      NBH_HIP(hipEventRecord(x.ev_ready, x.compute));
      NBH_HIP(hipStreamWaitEvent(x.comm, x.ev_ready, 0));
    }
    if (rccl) {
      if (s->sh.size() > 1) NBH_NCCL(api, api->GroupStart());
      for (auto& x : s->sh) {
        NBH_HIP(hipSetDevice(x.device));
        NBH_NCCL(api, api->AllGather(s->own(x), x.posm_all, S * 4, ncclFloat, x.nccl, x.comm));
      }
      if (s->sh.size() > 1) NBH_NCCL(api, api->GroupEnd());
    } else {
      for (auto& x : s->sh) {
        NBH_HIP(hipSetDevice(x.device));
        for (auto& p : s->sh) {
          if (&p == &x) continue;
          NBH_HIP(hipStreamWaitEvent(x.comm, p.ev_read, 0));  // p's kernels of the last phase still read p.posm_all
          NBH_HIP(hipMemcpyAsync(p.posm_all + (size_t)x.rank * S, s->own(x), S * sizeof(float4), hipMemcpyDeviceToDevice, x.comm));
        }
      }
    }
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      NBH_HIP(hipEventRecord(x.ev_gather, x.comm));
    }
  }
  // -- own x own (overlaps the gather) ----------------------------------------------------------------------------
  for (auto& x : s->sh)
    if (int rc = nbody_hip_direct_forces_packed(x.ctx, reinterpret_cast<nbody_float4*>(s->own(x)), S,
                                                reinterpret_cast<nbody_float4*>(s->own(x)), S,
                                                reinterpret_cast<nbody_float4*>(x.mine), s->G, s->eps2, 0))
      return rc;
  // -- wait for the gathered bodies ---------------------------------------------------------------------------------
  if (!gathered)
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      if (rccl) {
        NBH_HIP(hipStreamWaitEvent(x.compute, x.ev_gather, 0));
      } else {
        for (auto& p : s->sh)
          if (&p != &x) NBH_HIP(hipStreamWaitEvent(x.compute, p.ev_gather, 0));
      }
    }
  // -- the other shards ---------------------------------------------------------------------------------------------
  for (auto& x : s->sh) {
    if (s->pair_mode) {
      for (auto& t : x.tasks)
        if (int rc = nbody_hip_direct_forces_pair_packed(
                x.ctx, reinterpret_cast<nbody_float4*>(s->own(x) + t.i0), t.i1 - t.i0,
                reinterpret_cast<nbody_float4*>(x.posm_all + (size_t)t.sh * S + t.j0), t.j1 - t.j0,
                reinterpret_cast<nbody_float4*>(x.mine + t.i0), 1, reinterpret_cast<nbody_float4*>(t.send), 0, s->G, s->eps2))
          return rc;
    } else if (W > 1) {  // one-sided kernel against the shards left and right of the own range (tiny softening)
      const size_t r = (size_t)x.rank;
      if (r > 0)
        if (int rc = nbody_hip_direct_forces_packed(x.ctx, reinterpret_cast<nbody_float4*>(s->own(x)), S,
                                                    reinterpret_cast<nbody_float4*>(x.posm_all), r * S,
                                                    reinterpret_cast<nbody_float4*>(x.mine), s->G, s->eps2, 1))
          return rc;
      if (r + 1 < (size_t)W)
        if (int rc = nbody_hip_direct_forces_packed(x.ctx, reinterpret_cast<nbody_float4*>(s->own(x)), S,
                                                    reinterpret_cast<nbody_float4*>(x.posm_all + (r + 1) * S), all - (r + 1) * S,
                                                    reinterpret_cast<nbody_float4*>(x.mine), s->G, s->eps2, 1))
          return rc;
    }
    NBH_HIP(hipSetDevice(x.device));
    NBH_HIP(hipEventRecord(x.ev_read, x.compute));
  }
  // -- exchange of the reaction blocks --------------------------------------------------------------------------------
  if (s->pair_mode) {
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      NBH_HIP(hipStreamWaitEvent(x.comm, x.ev_read, 0));
    }
    if (rccl) {
      NBH_NCCL(api, api->GroupStart());
      for (auto& x : s->sh) {
        NBH_HIP(hipSetDevice(x.device));
        for (auto& t : x.tasks) NBH_NCCL(api, api->Send(t.send, (t.j1 - t.j0) * 4, ncclFloat, t.sh, x.nccl, x.comm));
        for (auto& in : x.incoming) NBH_NCCL(api, api->Recv(in.buf, (in.j1 - in.j0) * 4, ncclFloat, in.from, x.nccl, x.comm));
      }
      NBH_NCCL(api, api->GroupEnd());
    } else {
      for (auto& x : s->sh) {
        NBH_HIP(hipSetDevice(x.device));
        for (auto& t : x.tasks) {
          Shard* dst = s->by_rank[t.sh];
          Incoming* slot = nullptr;
          for (auto& in : dst->incoming)
            if (in.from == x.rank) slot = &in;
          if (!slot || slot->j1 - slot->j0 != t.j1 - t.j0) return NBH_FAIL(NBODY_HIP_ERR_STATE, "schedule mismatch between ranks %d and %d", x.rank, t.sh);
          NBH_HIP(hipMemcpyAsync(slot->buf, t.send, (t.j1 - t.j0) * sizeof(float4), hipMemcpyDeviceToDevice, x.comm));
        }
      }
    }
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      NBH_HIP(hipEventRecord(x.ev_sent, x.comm));
    }
  }
  // -- finalize ---------------------------------------------------------------------------------------------------
  for (auto& x : s->sh) {
    NBH_HIP(hipSetDevice(x.device));
    InBlocks in;
    in.n = 0;
    if (s->pair_mode) {
      if (rccl) {
        NBH_HIP(hipStreamWaitEvent(x.compute, x.ev_sent, 0));
      } else {
        for (auto& b : x.incoming) NBH_HIP(hipStreamWaitEvent(x.compute, s->by_rank[b.from]->ev_sent, 0));
      }
      for (auto& b : x.incoming) {
        in.buf[in.n] = b.buf;
        in.j0[in.n] = (unsigned)b.j0;
        in.j1[in.n] = (unsigned)b.j1;
        in.n++;
      }
    }
    float4* a_new = kick ? x.acc[1 - s->cur] : x.acc[out_slot];
    hipLaunchKernelGGL(shard_finalize_kernel, dim3((unsigned)((S + kBlock - 1) / kBlock)), dim3(kBlock), 0, x.compute, x.mine,
                       in, (int)S, a_new, x.vel, x.acc[s->cur], half_dt, kick ? 1 : 0);
    NBH_LAUNCH_CHECK();
  }
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_sharded_direct_forces(nbody_hip_sharded_direct* s) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!s) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null system");
  if (!s->have_state) return NBH_FAIL(NBODY_HIP_ERR_STATE, "no body state (call nbody_hip_sharded_direct_set_state first)");
  return force_phase(s, false, false, 0.f, s->cur);
}

static int one_step(nbody_hip_sharded_direct* s, float dt) {
  for (auto& x : s->sh)
    if (int rc = nbody_hip_drift_packed(x.ctx, reinterpret_cast<nbody_float4*>(s->own(x)),
                                        reinterpret_cast<nbody_float4*>(x.vel),
                                        reinterpret_cast<nbody_float4*>(x.acc[s->cur]), s->S, dt))
      return rc;
  if (int rc = force_phase(s, false, true, 0.5f * dt, 0)) return rc;
  s->cur = 1 - s->cur;  // a_old <- a by buffer swap
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_sharded_direct_step(nbody_hip_sharded_direct* s, float dt, int steps) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!s) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null system");
  if (!s->have_state) return NBH_FAIL(NBODY_HIP_ERR_STATE, "no body state (call nbody_hip_sharded_direct_set_state first)");
  if (!(dt > 0.0f) || !(dt <= 1.0f)) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Time step must be in range (0, 1]");
  for (int k = 0; k < steps; k++)
    if (int rc = one_step(s, dt)) return rc;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_sharded_direct_synchronize(nbody_hip_sharded_direct* s) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!s) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null system");
  for (auto& x : s->sh) {
    NBH_HIP(hipSetDevice(x.device));
    NBH_HIP(hipStreamSynchronize(x.comm));
    NBH_HIP(hipStreamSynchronize(x.compute));
  }
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_sharded_direct_time_steps(nbody_hip_sharded_direct* s, float dt, int warmup, int steps,
                                                   float* ms_per_step) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!s || !ms_per_step) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (steps <= 0 || warmup < 0) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "bad timing arguments");
  if (int rc = nbody_hip_sharded_direct_step(s, dt, warmup)) return rc;
  if (int rc = nbody_hip_sharded_direct_synchronize(s)) return rc;
  Shard& x = s->sh[0];
  NBH_HIP(hipSetDevice(x.device));
  NBH_HIP(hipEventRecord(x.t0, x.compute));
  if (int rc = nbody_hip_sharded_direct_step(s, dt, steps)) return rc;
  NBH_HIP(hipSetDevice(x.device));
  NBH_HIP(hipEventRecord(x.t1, x.compute));
  if (int rc = nbody_hip_sharded_direct_synchronize(s)) return rc;
  float ms = 0.f;
  NBH_HIP(hipEventElapsedTime(&ms, x.t0, x.t1));
  *ms_per_step = ms / (float)steps;
  return NBODY_HIP_OK;
}

// all-gather of a per-shard [S] float4 array into `stage` [W S] on every local rank (blocking)
static int gather_rows(nbody_hip_sharded_direct* s, int which /*0 vel, 1 acc*/) {
  const size_t S = s->S, all = S * (size_t)s->W;
  const bool rccl = s->comm->transport == NBODY_HIP_TRANSPORT_RCCL;
  Rccl* api = rccl ? rccl_load() : nullptr;
  for (auto& x : s->sh) {
    NBH_HIP(hipSetDevice(x.device));
    if (!x.stage) NBH_HIP(hipMalloc(reinterpret_cast<void**>(&x.stage), all * sizeof(float4)));
    NBH_HIP(hipStreamSynchronize(x.compute));
    NBH_HIP(hipStreamSynchronize(x.comm));
  }
  if (rccl) {
    if (!api) return NBH_FAIL((nbody_hip_status)NBODY_HIP_ERR_COMM, "librccl.so.1 is not loaded");
    if (s->sh.size() > 1) NBH_NCCL(api, api->GroupStart());
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      const float4* src = which == 0 ? x.vel : x.acc[s->cur];
      NBH_NCCL(api, api->AllGather(src, x.stage, S * 4, ncclFloat, x.nccl, x.comm));
    }
    if (s->sh.size() > 1) NBH_NCCL(api, api->GroupEnd());
  } else {
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      const float4* src = which == 0 ? x.vel : x.acc[s->cur];
      for (auto& p : s->sh)
        NBH_HIP(hipMemcpyAsync(p.stage + (size_t)x.rank * S, src, S * sizeof(float4), hipMemcpyDeviceToDevice, x.comm));
    }
  }
  for (auto& x : s->sh) {
    NBH_HIP(hipSetDevice(x.device));
    NBH_HIP(hipStreamSynchronize(x.comm));
  }
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_sharded_direct_get_state(nbody_hip_sharded_direct* s, float* x, float* y, float* z, float* vx,
                                                  float* vy, float* vz, float* ax, float* ay, float* az, int gather) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!s) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null system");
  if (!s->have_state) return NBH_FAIL(NBODY_HIP_ERR_STATE, "no body state");
  if (int rc = nbody_hip_sharded_direct_synchronize(s)) return rc;
  const size_t S = s->S, n = s->n;
  const bool whole = gather != 0 && !s->comm->all_local();
  std::vector<float4> h(whole ? S * (size_t)s->W : S);
  auto scatter = [&](float* a, float* b, float* c, size_t lo, size_t count, const float4* src) {
    for (size_t i = 0; i < count; i++) {
      if (a) a[lo + i] = src[i].x;
      if (b) b[lo + i] = src[i].y;
      if (c) c[lo + i] = src[i].z;
    }
  };
  if (whole) {
    // positions: posm_all of any local rank holds every shard as of the last exchange (= the current positions);
    // velocities and accelerations are all-gathered
    Shard& r0 = s->sh[0];
    NBH_HIP(hipSetDevice(r0.device));
    if (x || y || z) {
      NBH_HIP(hipMemcpy(h.data(), r0.posm_all, h.size() * sizeof(float4), hipMemcpyDeviceToHost));
      scatter(x, y, z, 0, n, h.data());
    }
    for (int which = 0; which < 2; which++) {
      float *a = which ? ax : vx, *b = which ? ay : vy, *c = which ? az : vz;
      if (!a && !b && !c) continue;
      if (int rc = gather_rows(s, which)) return rc;
      NBH_HIP(hipSetDevice(r0.device));
      NBH_HIP(hipMemcpy(h.data(), r0.stage, h.size() * sizeof(float4), hipMemcpyDeviceToHost));
      scatter(a, b, c, 0, n, h.data());
    }
    return NBODY_HIP_OK;
  }
  for (auto& sh : s->sh) {
    size_t lo, hi;
    nbody_hip_shard_bounds(n, s->W, sh.rank, nullptr, &lo, &hi);
    NBH_HIP(hipSetDevice(sh.device));
    if (x || y || z) {
      NBH_HIP(hipMemcpy(h.data(), s->own(sh), S * sizeof(float4), hipMemcpyDeviceToHost));
      scatter(x, y, z, lo, hi - lo, h.data());
    }
    if (vx || vy || vz) {
      NBH_HIP(hipMemcpy(h.data(), sh.vel, S * sizeof(float4), hipMemcpyDeviceToHost));
      scatter(vx, vy, vz, lo, hi - lo, h.data());
    }
    if (ax || ay || az) {
      NBH_HIP(hipMemcpy(h.data(), sh.acc[s->cur], S * sizeof(float4), hipMemcpyDeviceToHost));
      scatter(ax, ay, az, lo, hi - lo, h.data());
    }
  }
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_sharded_direct_energies(nbody_hip_sharded_direct* s, double* kinetic, double* potential) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!s) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null system");
  if (!s->have_state) return NBH_FAIL(NBODY_HIP_ERR_STATE, "no body state");
  if (int rc = nbody_hip_sharded_direct_synchronize(s)) return rc;
  const size_t S = s->S, all = S * (size_t)s->W;
  const int W = s->W;
  std::vector<double> e(2 * (size_t)W, 0.0);
  // posm_all holds every shard as of the last exchange (nothing moves bodies after the drift of a step)
  for (auto& x : s->sh) {
    double out[2];
    if (int rc = nbody_hip_energies_packed(x.ctx, reinterpret_cast<nbody_float4*>(s->own(x)), reinterpret_cast<nbody_float4*>(x.vel), S,
                                           (long long)((size_t)x.rank * S), reinterpret_cast<nbody_float4*>(x.posm_all), all, s->G,
                                           s->eps, out))
      return rc;
    e[2 * (size_t)x.rank] = out[0];
    e[2 * (size_t)x.rank + 1] = out[1];
  }
  if (!s->comm->all_local()) {  // one process per GPU: all-gather the 2 doubles of every rank, then sum in rank order
    Rccl* api = rccl_load();
    if (!api) return NBH_FAIL((nbody_hip_status)NBODY_HIP_ERR_COMM, "librccl.so.1 is not loaded");
    if (s->sh.size() > 1) NBH_NCCL(api, api->GroupStart());
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      NBH_HIP(hipMemcpyAsync(x.esum + 2 * (size_t)x.rank, &e[2 * (size_t)x.rank], 2 * sizeof(double), hipMemcpyHostToDevice, x.comm));
      NBH_NCCL(api, api->AllGather(x.esum + 2 * (size_t)x.rank, x.esum, 2, ncclDouble, x.nccl, x.comm));
    }
    if (s->sh.size() > 1) NBH_NCCL(api, api->GroupEnd());
    Shard& r0 = s->sh[0];
    NBH_HIP(hipSetDevice(r0.device));
    NBH_HIP(hipStreamSynchronize(r0.comm));
    NBH_HIP(hipMemcpy(e.data(), r0.esum, e.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      NBH_HIP(hipStreamSynchronize(x.comm));
    }
  }
  double ke = 0.0, pe = 0.0;
  for (int r = 0; r < W; r++) {
    ke += e[2 * (size_t)r];
    pe += e[2 * (size_t)r + 1];
  }
  if (kinetic) *kinetic = ke;
  if (potential) *potential = pe;
  return NBODY_HIP_OK;
}

// the plugin form: the whole system in d (SoA, on the first local rank's device) -> d->acc_*
extern "C" int nbody_hip_sharded_direct_compute_forces(nbody_hip_sharded_direct* s, nbody_particle_data* d) {
  nbh::DeviceGuard dev_guard;  // the caller's current device is restored on return
  if (!s) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null system");
  if (!d) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null particle data");
  if (d->count != s->n) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "particle count %zu does not match the system's %zu", d->count, s->n);
  if (!d->pos_x || !d->pos_y || !d->pos_z || !d->mass || !d->acc_x || !d->acc_y || !d->acc_z)
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "particle data has null arrays");
  const size_t S = s->S, all = S * (size_t)s->W, n = s->n;
  const bool rccl = s->comm->transport == NBODY_HIP_TRANSPORT_RCCL;
  Rccl* api = rccl ? rccl_load() : nullptr;
  Shard& r0 = s->sh[0];
  NBH_HIP(hipSetDevice(r0.device));
  // order against the caller's work on the null stream of that device (the reference's threading model:
  // everything on the null stream, force_calculator.hpp:8-19)
  NBH_HIP(hipEventRecord(r0.ev_aux, nullptr));
  NBH_HIP(hipStreamWaitEvent(r0.compute, r0.ev_aux, 0));
  // the readers of the last phase on the other ranks must be done before their posm_all is overwritten
  for (auto& p : s->sh)
    if (&p != &r0) NBH_HIP(hipStreamWaitEvent(r0.compute, p.ev_read, 0));
  if (int rc = nbody_hip_pack_posm(r0.ctx, d->pos_x, d->pos_y, d->pos_z, d->mass, n, reinterpret_cast<nbody_float4*>(r0.posm_all)))
    return rc;
  NBH_HIP(hipEventRecord(r0.ev_ready, r0.compute));
  for (auto& p : s->sh) {
    if (&p == &r0) continue;
    NBH_HIP(hipMemcpyAsync(p.posm_all, r0.posm_all, n * sizeof(float4), hipMemcpyDeviceToDevice, r0.compute));
  }
  NBH_HIP(hipEventRecord(r0.ev_gather, r0.compute));
  for (auto& p : s->sh) {
    if (&p == &r0) continue;
    NBH_HIP(hipSetDevice(p.device));
    NBH_HIP(hipStreamWaitEvent(p.compute, r0.ev_gather, 0));
  }
  s->have_state = true;  // (positions only: a resident step after this needs set_state for the velocities)
  if (int rc = force_phase(s, true, false, 0.f, s->cur)) return rc;
  // accelerations of every shard -> stage of the first local rank -> d->acc_*
  for (auto& x : s->sh) {
    NBH_HIP(hipSetDevice(x.device));
    if (!x.stage) {
      NBH_HIP(hipStreamSynchronize(x.compute));
      NBH_HIP(hipMalloc(reinterpret_cast<void**>(&x.stage), all * sizeof(float4)));
    }
    NBH_HIP(hipEventRecord(x.ev_ready, x.compute));
    NBH_HIP(hipStreamWaitEvent(x.comm, x.ev_ready, 0));
  }
  if (!s->comm->all_local()) {
    if (!api) return NBH_FAIL((nbody_hip_status)NBODY_HIP_ERR_COMM, "librccl.so.1 is not loaded");
    if (s->sh.size() > 1) NBH_NCCL(api, api->GroupStart());
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      NBH_NCCL(api, api->AllGather(x.acc[s->cur], x.stage, S * 4, ncclFloat, x.nccl, x.comm));
    }
    if (s->sh.size() > 1) NBH_NCCL(api, api->GroupEnd());
  } else {
    for (auto& x : s->sh) {
      NBH_HIP(hipSetDevice(x.device));
      NBH_HIP(hipMemcpyAsync(r0.stage + (size_t)x.rank * S, x.acc[s->cur], S * sizeof(float4), hipMemcpyDeviceToDevice, x.comm));
    }
  }
  NBH_HIP(hipSetDevice(r0.device));
  for (auto& x : s->sh) {
    NBH_HIP(hipSetDevice(x.device));
    NBH_HIP(hipEventRecord(x.ev_sent, x.comm));
    NBH_HIP(hipSetDevice(r0.device));
    NBH_HIP(hipStreamWaitEvent(r0.compute, x.ev_sent, 0));
  }
  if (int rc = nbody_hip_unpack3(r0.ctx, reinterpret_cast<nbody_float4*>(r0.stage), n, d->acc_x, d->acc_y, d->acc_z)) return rc;
  NBH_HIP(hipEventRecord(r0.ev_aux, r0.compute));
  NBH_HIP(hipStreamWaitEvent(nullptr, r0.ev_aux, 0));
  return NBODY_HIP_OK;
}
