// api.hip -- context, error reporting, particle memory and the Direct-N^2 entry points of the
// C ABI declared in include/nbody_hip.h.

#include <cstdarg>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace nbh {

size_t own_sort_from(size_t compiled_default) {
  if (const char* e = std::getenv("NBH_OWN_SORT_FROM")) {
    char* end = nullptr;
    const long long v = std::strtoll(e, &end, 10);
    if (end != e && v >= 0) return (size_t)v;
  }
  return compiled_default;
}

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(nbody_hip_status code, const char* file, int line, const char* fmt, ...) {
  char msg[384];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(msg, sizeof(msg), fmt, ap);
  va_end(ap);
  const char* base = strrchr(file, '/');
  snprintf(g_err, sizeof(g_err), "%s at %s:%d", msg, base ? base + 1 : file, line);
  return (int)code;
}

int Workspace::reserve(size_t want) {
  if (want <= bytes) return NBODY_HIP_OK;
  // grow geometrically so a slowly growing problem does not reallocate every call
  size_t cap = bytes + bytes / 2;
  if (cap < want) cap = want;
  cap = (cap + 255) & ~(size_t)255;
  want = (want + 255) & ~(size_t)255;
  if (ptr) {
    // the old buffer goes first (its contents are never carried over), so that old + new are not held
    // together at the peak; queued work may still read it
    NBH_HIP(hipDeviceSynchronize());
    NBH_HIP(hipFree(ptr));
    ptr = nullptr;
    bytes = 0;
    generation++;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, cap);
  if (e != hipSuccess && cap > want) {  // the geometric head-room is a convenience, not a need
    (void)hipGetLastError();
    cap = want;
    e = hipMalloc(&p, cap);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return ::nbh::fail(e == hipErrorOutOfMemory ? NBODY_HIP_ERR_RESOURCE : NBODY_HIP_ERR_DEVICE, __FILE__, __LINE__,
                       "hipMalloc(%zu bytes of workspace): %s", cap, hipGetErrorString(e));
  }
  ptr = p;
  bytes = cap;
  generation++;
  return NBODY_HIP_OK;
}

void Workspace::release() {  // the caller has made sure nothing queued still uses the buffer
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  bytes = 0;
  generation++;
}

}  // namespace nbh

using namespace nbh;

extern "C" int nbody_hip_abi_version(void) { return NBODY_HIP_ABI_VERSION; }

extern "C" const char* nbody_hip_last_error(void) { return g_err; }

extern "C" int nbody_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

extern "C" int nbody_hip_ctx_create(nbody_hip_ctx** out, int device, void* stream) {
  if (!out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null output pointer");
  *out = nullptr;
  const int ndev = nbody_hip_device_count();
  if (ndev <= 0)
    return NBH_FAIL(NBODY_HIP_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "device %d out of range [0,%d)", device, ndev);
  NBH_HIP(hipSetDevice(device));
  nbody_hip_ctx* c = new nbody_hip_ctx();
  c->device = device;
  // NULL = HIP's default (null) stream, which is what the reference launches on
  // (force_direct.cu:93) and what torch.cuda.current_stream() is unless a side stream is active.
  c->stream = static_cast<hipStream_t>(stream);
  hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&c->host_scalar), 4 * sizeof(double), hipHostMallocDefault);
  if (e == hipSuccess) e = hipEventCreate(&c->ev0);
  if (e == hipSuccess) e = hipEventCreate(&c->ev1);
  if (e != hipSuccess) {
    nbody_hip_ctx_destroy(c);
    return NBH_FAIL(NBODY_HIP_ERR_DEVICE, "context resources: %s", hipGetErrorString(e));
  }
  *out = c;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_ctx_destroy(nbody_hip_ctx* ctx) {
  if (!ctx) return NBODY_HIP_OK;
  NBH_DESTROY_BEGIN
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->posm.release();
  ctx->partial.release();
  ctx->reduce.release();
  if (ctx->host_scalar) (void)hipHostFree(ctx->host_scalar);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->capture_stream) (void)hipStreamDestroy(ctx->capture_stream);
  delete ctx;
  NBH_DESTROY_END
}

extern "C" int nbody_hip_ctx_set_stream(nbody_hip_ctx* ctx, void* stream) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  ctx->stream = static_cast<hipStream_t>(stream);
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_ctx_synchronize(nbody_hip_ctx* ctx) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  NBH_NOT_CAPTURABLE(ctx, "nbody_hip_ctx_synchronize");
  NBH_HIP(hipSetDevice(ctx->device));
  NBH_HIP(hipStreamSynchronize(ctx->stream));
  return NBODY_HIP_OK;
}

// ---- step graphs -------------------------------------------------------------------------

struct nbody_hip_graph {
  nbody_hip_ctx* ctx = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  unsigned long long generation = 0;  // ctx->generation() when it was recorded
};

extern "C" int nbody_hip_capture_begin(nbody_hip_ctx* ctx) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (ctx->capturing) return NBH_FAIL(NBODY_HIP_ERR_STATE, "capture already in progress");
  NBH_HIP(hipSetDevice(ctx->device));
  // the null stream cannot be captured: record on a stream of our own, replay on the caller's
  if (!ctx->capture_stream) NBH_HIP(hipStreamCreateWithFlags(&ctx->capture_stream, hipStreamNonBlocking));
  NBH_HIP(hipStreamSynchronize(ctx->stream));
  NBH_HIP(hipStreamBeginCapture(ctx->capture_stream, hipStreamCaptureModeThreadLocal));
  ctx->user_stream = ctx->stream;
  ctx->stream = ctx->capture_stream;
  ctx->capturing = true;
  ctx->capture_failed = false;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_capture_end(nbody_hip_ctx* ctx, nbody_hip_graph** out) {
  if (!ctx || !out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  *out = nullptr;
  if (!ctx->capturing) return NBH_FAIL(NBODY_HIP_ERR_STATE, "no capture in progress");
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(ctx->capture_stream, &graph);
  ctx->stream = ctx->user_stream;
  ctx->capturing = false;
  if (e != hipSuccess || !graph || ctx->capture_failed) {
    (void)hipGetLastError();
    if (graph) (void)hipGraphDestroy(graph);
    return NBH_FAIL(NBODY_HIP_ERR_DEVICE, "hipStreamEndCapture: %s (a call inside the capture was not capturable)",
                    hipGetErrorString(e));
  }
  nbody_hip_graph* g = new nbody_hip_graph();
  g->ctx = ctx;
  g->graph = graph;
  g->generation = ctx->generation();
  const hipError_t ei = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
  if (ei != hipSuccess) {
    (void)hipGraphDestroy(graph);
    delete g;
    return NBH_FAIL(NBODY_HIP_ERR_DEVICE, "hipGraphInstantiate: %s", hipGetErrorString(ei));
  }
  *out = g;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_graph_launch(nbody_hip_graph* g, int times) {
  if (!g || !g->exec) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null graph");
  if (g->ctx->capturing) return NBH_FAIL(NBODY_HIP_ERR_STATE, "graph launch inside a capture");
  // the recording holds raw pointers into the context's workspaces and into the tree it was made
  // with: if any of them was re-allocated or freed since, replaying would touch freed memory
  if (g->generation != g->ctx->generation())
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "step graph is stale: device buffers of its context were re-allocated "
                    "after it was recorded (record it again)");
  NBH_HIP(hipSetDevice(g->ctx->device));
  g->ctx->graph_replays += (unsigned long long)(times > 0 ? times : 0);
  for (int i = 0; i < times; i++) NBH_HIP(hipGraphLaunch(g->exec, g->ctx->stream));
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_graph_destroy(nbody_hip_graph* g) {
  if (!g) return NBODY_HIP_OK;
  NBH_DESTROY_BEGIN
  (void)hipSetDevice(g->ctx->device);
  (void)hipStreamSynchronize(g->ctx->stream);
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
  NBH_DESTROY_END
}

// ---- particle memory -------------------------------------------------------------------

static inline float** field(nbody_particle_data* d, int k) { return &d->pos_x + k; }
static inline float* const* cfield(const nbody_particle_data* d, int k) { return &d->pos_x + k; }

extern "C" int nbody_hip_particles_alloc(nbody_particle_data* d, size_t count) {
  if (!d) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null particle data");
  if (count == 0) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Particle count must be greater than 0");
  if (count > 100000000u)  // ref: validateParticleCountRange, error_handling.cpp:76-84
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Particle count exceeds maximum supported (100M)");
  if (nbody_hip_device_count() <= 0)
    return NBH_FAIL(NBODY_HIP_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
  // One slab, 13 sub-arrays each padded to 256 B: one allocation instead of the reference's 13
  // (particle_init.cu:147-159) and every array 16-byte aligned for dwordx4 access.
  const size_t stride = ((count * sizeof(float)) + 255) & ~(size_t)255;
  char* base = nullptr;
  NBH_HIP(hipMalloc(reinterpret_cast<void**>(&base), stride * 13));
  // ref zeroes the six acceleration arrays (:161-166); zero everything so that no field is
  // ever uninitialised device memory.
  hipError_t e = hipMemset(base, 0, stride * 13);
  if (e != hipSuccess) {
    (void)hipFree(base);
    return NBH_FAIL(NBODY_HIP_ERR_DEVICE, "hipMemset: %s", hipGetErrorString(e));
  }
  for (int k = 0; k < 13; k++) *field(d, k) = reinterpret_cast<float*>(base + stride * k);
  d->count = count;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_particles_free(nbody_particle_data* d) {
  if (!d) return NBODY_HIP_OK;
  if (d->pos_x) {
    NBH_HIP(hipDeviceSynchronize());
    NBH_HIP(hipFree(d->pos_x));  // slab base
  }
  for (int k = 0; k < 13; k++) *field(d, k) = nullptr;
  d->count = 0;
  return NBODY_HIP_OK;
}

// the 10 arrays the reference copies (particle_init.cu:257-283): pos, vel, acc, mass
static const int kCopied[10] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 12};

extern "C" int nbody_hip_particles_upload(nbody_particle_data* d, const nbody_particle_data* h) {
  if (!d || !h) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null particle data");
  if (h->count > d->count)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "host count %zu exceeds device capacity %zu", h->count, d->count);
  const size_t bytes = h->count * sizeof(float);
  for (int k : kCopied) {
    if (!*cfield(h, k) || !*cfield(d, k)) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null array in particle data");
    NBH_HIP(hipMemcpy(*field(d, k), *cfield(h, k), bytes, hipMemcpyHostToDevice));
  }
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_particles_download(nbody_particle_data* h, const nbody_particle_data* d) {
  if (!d || !h) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null particle data");
  if (d->count > h->count)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "device count %zu exceeds host capacity %zu", d->count, h->count);
  const size_t bytes = d->count * sizeof(float);
  for (int k : kCopied) {
    if (!*cfield(h, k) || !*cfield(d, k)) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null array in particle data");
    NBH_HIP(hipMemcpy(*field(h, k), *cfield(d, k), bytes, hipMemcpyDeviceToHost));
  }
  return NBODY_HIP_OK;
}

// ---- Direct N^2 ------------------------------------------------------------------------

extern "C" int nbody_hip_pack_posm(nbody_hip_ctx* ctx, const float* x, const float* y,
                                   const float* z, const float* mass, size_t count,
                                   nbody_float4* out) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (count == 0) return NBODY_HIP_OK;
  if (!x || !y || !z || !mass || !out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (count > 0x3fffffffu) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "body count exceeds 2^30");
  NBH_HIP(hipSetDevice(ctx->device));
  return pack_posm(ctx, x, y, z, mass, count, reinterpret_cast<float4*>(out));
}

namespace nbh {
__global__ __launch_bounds__(kBlock) void unpack3_kernel(const float4* __restrict__ in, int n,
                                                         float* __restrict__ x,
                                                         float* __restrict__ y,
                                                         float* __restrict__ z) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    const float4 v = in[i];
    x[i] = v.x; y[i] = v.y; z[i] = v.z;
  }
}
}  // namespace nbh

extern "C" int nbody_hip_unpack3(nbody_hip_ctx* ctx, const nbody_float4* in, size_t count, float* x,
                                 float* y, float* z) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (count == 0) return NBODY_HIP_OK;
  if (!x || !y || !z || !in) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (count > 0x3fffffffu) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "body count exceeds 2^30");
  NBH_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(unpack3_kernel, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock),
                     0, ctx->stream, reinterpret_cast<const float4*>(in), (int)count, x, y, z);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_direct_forces(nbody_hip_ctx* ctx, const nbody_particle_data* d, float G,
                                       float eps2, int block_size) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (!d) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null particle data");
  if (block_size <= 0 || block_size > 1024)  // ref: validateSimulationConfig, error_handling.cpp:70-72
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "CUDA block size must be between 1 and 1024");
  const size_t n = d->count;
  if (n == 0) return NBODY_HIP_OK;
  if (!d->pos_x || !d->pos_y || !d->pos_z || !d->mass || !d->acc_x || !d->acc_y || !d->acc_z)
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "particle data has null arrays");
  if (n > 0x3fffffffu) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "body count exceeds 2^30");
  NBH_HIP(hipSetDevice(ctx->device));
  if (int rc = ctx->posm.reserve(n * sizeof(float4))) return rc;
  float4* posm = static_cast<float4*>(ctx->posm.ptr);
  if (int rc = pack_posm(ctx, d->pos_x, d->pos_y, d->pos_z, d->mass, n, posm)) return rc;
  return direct_packed(ctx, posm, n, posm, n, G, eps2, nullptr, 0, d->acc_x, d->acc_y, d->acc_z,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f);
}

extern "C" int nbody_hip_direct_forces_packed(nbody_hip_ctx* ctx, const nbody_float4* targets,
                                              size_t n_targets, const nbody_float4* sources,
                                              size_t n_sources, nbody_float4* acc_out, float G,
                                              float eps2, int accumulate) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (n_targets == 0) return NBODY_HIP_OK;
  if (!targets || !acc_out || (n_sources > 0 && !sources))
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  NBH_HIP(hipSetDevice(ctx->device));
  return direct_packed(ctx, reinterpret_cast<const float4*>(targets), n_targets,
                       reinterpret_cast<const float4*>(sources), n_sources, G, eps2,
                       reinterpret_cast<float4*>(acc_out), accumulate, nullptr, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f);
}

extern "C" int nbody_hip_direct_forces_pair_packed(nbody_hip_ctx* ctx, const nbody_float4* a, size_t n_a,
                                                   const nbody_float4* b, size_t n_b, nbody_float4* acc_a,
                                                   int accumulate_a, nbody_float4* acc_b, int accumulate_b,
                                                   float G, float eps2) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (n_a == 0 || n_b == 0) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "both body sets must be non-empty");
  if (!a || !b || !acc_a || !acc_b) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (n_a > 0x3fffffffu || n_b > 0x3fffffffu) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "body count exceeds 2^30");
  if (!(eps2 >= 1e-12f))
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "pair evaluation needs eps2 >= 1e-12 (use the one-sided entry point)");
  NBH_HIP(hipSetDevice(ctx->device));
  return direct_symmetric_pair(ctx, reinterpret_cast<const float4*>(a), n_a, reinterpret_cast<const float4*>(b),
                               n_b, G, eps2, reinterpret_cast<float4*>(acc_a), accumulate_a,
                               reinterpret_cast<float4*>(acc_b), accumulate_b);
}

extern "C" int nbody_hip_time_direct_packed(nbody_hip_ctx* ctx, const nbody_float4* targets,
                                            size_t n_targets, const nbody_float4* sources,
                                            size_t n_sources, nbody_float4* acc_out, float G,
                                            float eps2, int iters, float* ms_per_launch) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (!ms_per_launch || iters <= 0) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "bad timing arguments");
  NBH_HIP(hipSetDevice(ctx->device));
  // one untimed call so that workspace growth is outside the timed region
  if (int rc = nbody_hip_direct_forces_packed(ctx, targets, n_targets, sources, n_sources, acc_out, G, eps2, 0))
    return rc;
  NBH_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  for (int i = 0; i < iters; i++)
    if (int rc = nbody_hip_direct_forces_packed(ctx, targets, n_targets, sources, n_sources, acc_out, G, eps2, 0))
      return rc;
  NBH_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  NBH_HIP(hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  NBH_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *ms_per_launch = ms / (float)iters;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_direct_deterministic(nbody_hip_ctx* ctx, int mode) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (mode < 0 || mode > 2) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "mode must be 0 (atomics), 1 (slots when they fit) or 2 (slots required)");
  ctx->deterministic = mode;
  // the slot planes are the one big allocation of this library: give them back when they are switched off
  if (mode == 0 && ctx->partial.bytes > ((size_t)64 << 20) && !ctx->capturing) {
    NBH_HIP(hipSetDevice(ctx->device));
    NBH_HIP(hipDeviceSynchronize());
    ctx->partial.release();
  }
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_direct_slot_budget(nbody_hip_ctx* ctx, unsigned long long bytes) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  ctx->det_budget = bytes ? (size_t)bytes : ((size_t)24 << 30);
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_direct_info(nbody_hip_ctx* ctx, size_t count, float eps2, nbody_hip_direct_info_t* out) {
  if (!out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  // ctx == NULL: the plan of a context with default settings, from the fixed budget alone (no device needed:
  // what the CPU-side tests of the shape / budget arithmetic call)
  const nbody_hip_ctx defaults{};
  const nbody_hip_ctx* c = ctx ? ctx : &defaults;
  memset(out, 0, sizeof(*out));
  out->deterministic_mode = c->deterministic;
  out->last_kernel = c->last_direct_kernel;
  out->workspace_bytes_held = c->partial.bytes;
  if (count == 0) return NBODY_HIP_OK;
  if (count > 0x3fffffffu) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "body count exceeds 2^30");
  // the hipMemGetInfo of the budget test needs the context's device current
  const bool dev = ctx && nbody_hip_device_count() > 0 && hipSetDevice(ctx->device) == hipSuccess;
  DirectPlan p = direct_plan(c, count, dev);
  if (!(eps2 >= 1e-12f)) p.symmetric = false;  // the guard variant of the one-sided kernel handles tiny softening
  if (!p.symmetric) {
    out->kernel = 0;
    return NBODY_HIP_OK;
  }
  out->kernel = p.det ? 2 : 1;
  out->bodies_per_lane_equal = p.eq.R;
  out->bodies_per_lane_general = p.gen.R;
  out->reaction_slots = p.det ? p.eq.D : 0;
  out->iside_slots = p.det ? p.eq.splits : 1;
  out->workspace_bytes_needed = p.bytes;
  out->slot_bytes_wanted = p.det_bytes_wanted;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_direct_tuning(nbody_hip_ctx* ctx, int variant, int targets_per_lane,
                                       int source_splits) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (variant < -1 || variant > 3) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "variant must be -1 (auto), 0, 1, 2 or 3");
  if (targets_per_lane != 0 && targets_per_lane != 1 && targets_per_lane != 2 && targets_per_lane != 4 &&
      targets_per_lane != 6 && targets_per_lane != 8 && targets_per_lane != 12 && targets_per_lane != 16)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "targets_per_lane must be 0 (auto), 1, 2, 4 (6, 8, 16: symmetric kernel only)");
  if (source_splits < 0 || source_splits > 4096)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "source_splits must be in [0, 4096]");
  ctx->tune_variant = variant;
  ctx->tune_tpl = targets_per_lane;
  ctx->tune_splits = source_splits;
  return NBODY_HIP_OK;
}
