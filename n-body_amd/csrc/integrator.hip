// integrator.hip -- Velocity-Verlet kernels on the reference's 13-array SoA layout.
//
// Replaces updatePositionsKernel / updateVelocitiesKernel / storeAccelerationsKernel and
// Integrator::integrate (src/cuda/integrator.cu:11-48, 224-238).  All three are pure HBM
// streaming (reference: 120 B/body/step in 3 launches).  The stand-alone entry points keep
// the reference's one-kernel-per-call semantics (its tests call them individually,
// tests/test_integrator.cpp:15-49); nbody_hip_integrate_direct fuses a_old<-a, the position
// update and the float4 packing into ONE pass (80 B/body) and the velocity update into the
// force-reduction epilogue (direct.hip: direct_finalize_kernel).
//
// Every kernel is grid-stride with 16-byte accesses when all arrays are 16-byte aligned
// (always true for nbody_hip_particles_alloc memory), scalar otherwise.

#include "common.h"

namespace nbh {

constexpr int kMaxStreamBlocks = kNumCU * 8;  // 2048: guide G11 grid cap for streaming kernels

static inline int stream_blocks(size_t items) {
  size_t b = (items + kBlock - 1) / kBlock;
  if (b > (size_t)kMaxStreamBlocks) b = kMaxStreamBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

struct Soa3 { float* x; float* y; float* z; };
struct CSoa3 { const float* x; const float* y; const float* z; };

// (drift1: common.h)

// x += v*dt + a*(0.5*dt*dt)     (integrator.cu:16-19; same operation order)
template <int VEC>
__global__ __launch_bounds__(kBlock) void update_positions_kernel(Soa3 p, CSoa3 v, CSoa3 a,
                                                                  size_t n, float dt) {
  const float dt2_half = 0.5f * dt * dt;
  const size_t stride = (size_t)gridDim.x * kBlock;
  if (VEC == 4) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
#define NBH_POS_AXIS(c)                                                        \
  {                                                                            \
    float4 pp = reinterpret_cast<float4*>(p.c)[i];                             \
    const float4 vv = reinterpret_cast<const float4*>(v.c)[i];                 \
    const float4 aa = reinterpret_cast<const float4*>(a.c)[i];                 \
    pp.x = drift1(pp.x, vv.x, aa.x, dt, dt2_half);                             \
    pp.y = drift1(pp.y, vv.y, aa.y, dt, dt2_half);                             \
    pp.z = drift1(pp.z, vv.z, aa.z, dt, dt2_half);                             \
    pp.w = drift1(pp.w, vv.w, aa.w, dt, dt2_half);                             \
    reinterpret_cast<float4*>(p.c)[i] = pp;                                    \
  }
      NBH_POS_AXIS(x) NBH_POS_AXIS(y) NBH_POS_AXIS(z)
#undef NBH_POS_AXIS
    }
    // tail
    const size_t i = n4 * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
      p.x[i] = drift1(p.x[i], v.x[i], a.x[i], dt, dt2_half);
      p.y[i] = drift1(p.y[i], v.y[i], a.y[i], dt, dt2_half);
      p.z[i] = drift1(p.z[i], v.z[i], a.z[i], dt, dt2_half);
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
      p.x[i] = drift1(p.x[i], v.x[i], a.x[i], dt, dt2_half);
      p.y[i] = drift1(p.y[i], v.y[i], a.y[i], dt, dt2_half);
      p.z[i] = drift1(p.z[i], v.z[i], a.z[i], dt, dt2_half);
    }
  }
}

// v += (a_old + a_new)*(0.5*dt)   (integrator.cu:31-34)
template <int VEC>
__global__ __launch_bounds__(kBlock) void update_velocities_kernel(Soa3 v, CSoa3 ao, CSoa3 an,
                                                                   size_t n, float dt) {
  const float dt_half = 0.5f * dt;
  const size_t stride = (size_t)gridDim.x * kBlock;
  if (VEC == 4) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
#define NBH_VEL_AXIS(c)                                                        \
  {                                                                            \
    float4 vv = reinterpret_cast<float4*>(v.c)[i];                             \
    const float4 o = reinterpret_cast<const float4*>(ao.c)[i];                 \
    const float4 w = reinterpret_cast<const float4*>(an.c)[i];                 \
    vv.x = kick1(vv.x, o.x, w.x, dt_half);                                     \
    vv.y = kick1(vv.y, o.y, w.y, dt_half);                                     \
    vv.z = kick1(vv.z, o.z, w.z, dt_half);                                     \
    vv.w = kick1(vv.w, o.w, w.w, dt_half);                                     \
    reinterpret_cast<float4*>(v.c)[i] = vv;                                    \
  }
      NBH_VEL_AXIS(x) NBH_VEL_AXIS(y) NBH_VEL_AXIS(z)
#undef NBH_VEL_AXIS
    }
    const size_t i = n4 * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
      v.x[i] = kick1(v.x[i], ao.x[i], an.x[i], dt_half);
      v.y[i] = kick1(v.y[i], ao.y[i], an.y[i], dt_half);
      v.z[i] = kick1(v.z[i], ao.z[i], an.z[i], dt_half);
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
      v.x[i] = kick1(v.x[i], ao.x[i], an.x[i], dt_half);
      v.y[i] = kick1(v.y[i], ao.y[i], an.y[i], dt_half);
      v.z[i] = kick1(v.z[i], ao.z[i], an.z[i], dt_half);
    }
  }
}

// Fused drift for the integrate path: a_old <- a ; x += v dt + a dt^2/2 ; posm <- {x,y,z,m}
// (integrator.cu:44-46 + :16-19 + the pack the force kernel needs).  80 B/body.
__global__ __launch_bounds__(kBlock) void drift_pack_kernel(Soa3 p, CSoa3 v, CSoa3 a, Soa3 ao,
                                                            const float* __restrict__ m,
                                                            float4* __restrict__ posm, size_t n,
                                                            float dt) {
  const float dt2_half = 0.5f * dt * dt;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const float ax = a.x[i], ay = a.y[i], az = a.z[i];
    ao.x[i] = ax; ao.y[i] = ay; ao.z[i] = az;
    const float x = drift1(p.x[i], v.x[i], ax, dt, dt2_half);
    const float y = drift1(p.y[i], v.y[i], ay, dt, dt2_half);
    const float z = drift1(p.z[i], v.z[i], az, dt, dt2_half);
    p.x[i] = x; p.y[i] = y; p.z[i] = z;
    if (posm) posm[i] = make_float4(x, y, z, m[i]);
  }
}

// Packed (float4) drift / kick for the sharded multi-GPU path: each rank keeps its target
// range as float4 arrays.  drift: x += v dt + a dt^2/2 (mass in .w untouched); kick:
// v += (a_old + a_new) dt/2.  a_old is the PREVIOUS acceleration buffer (the host swaps the two
// acceleration buffers instead of copying, so storeAccelerationsKernel has no counterpart).
__global__ __launch_bounds__(kBlock) void drift_packed_kernel(float4* __restrict__ posm,
                                                              const float4* __restrict__ vel,
                                                              const float4* __restrict__ acc,
                                                              size_t n, float dt) {
  const float dt2_half = 0.5f * dt * dt;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    float4 p = posm[i];
    const float4 v = vel[i], a = acc[i];
    p.x = drift1(p.x, v.x, a.x, dt, dt2_half);
    p.y = drift1(p.y, v.y, a.y, dt, dt2_half);
    p.z = drift1(p.z, v.z, a.z, dt, dt2_half);
    posm[i] = p;
  }
}

__global__ __launch_bounds__(kBlock) void kick_packed_kernel(float4* __restrict__ vel,
                                                             const float4* __restrict__ acc_old,
                                                             const float4* __restrict__ acc_new,
                                                             size_t n, float dt) {
  const float dt_half = 0.5f * dt;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    float4 v = vel[i];
    const float4 o = acc_old[i], w = acc_new[i];
    v.x = kick1(v.x, o.x, w.x, dt_half);
    v.y = kick1(v.y, o.y, w.y, dt_half);
    v.z = kick1(v.z, o.z, w.z, dt_half);
    vel[i] = v;
  }
}

static bool all_aligned(const nbody_particle_data* d) {
  return aligned16(d->pos_x) && aligned16(d->pos_y) && aligned16(d->pos_z) && aligned16(d->vel_x) &&
         aligned16(d->vel_y) && aligned16(d->vel_z) && aligned16(d->acc_x) && aligned16(d->acc_y) &&
         aligned16(d->acc_z) && aligned16(d->acc_old_x) && aligned16(d->acc_old_y) &&
         aligned16(d->acc_old_z);
}

}  // namespace nbh

using namespace nbh;

static int check_pd(const nbody_hip_ctx* ctx, const nbody_particle_data* d) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (!d) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null particle data");
  if (d->count > 0 && (!d->pos_x || !d->pos_y || !d->pos_z || !d->vel_x || !d->vel_y ||
                       !d->vel_z || !d->acc_x || !d->acc_y || !d->acc_z || !d->mass))
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "particle data has null arrays");
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_update_positions(nbody_hip_ctx* ctx, nbody_particle_data* d, float dt) {
  if (int rc = check_pd(ctx, d)) return rc;
  const size_t n = d->count;
  if (n == 0) return NBODY_HIP_OK;
  NBH_HIP(hipSetDevice(ctx->device));
  Soa3 p{d->pos_x, d->pos_y, d->pos_z};
  CSoa3 v{d->vel_x, d->vel_y, d->vel_z}, a{d->acc_x, d->acc_y, d->acc_z};
  if (all_aligned(d))
    hipLaunchKernelGGL(update_positions_kernel<4>, dim3(stream_blocks((n + 3) / 4)), dim3(kBlock),
                       0, ctx->stream, p, v, a, n, dt);
  else
    hipLaunchKernelGGL(update_positions_kernel<1>, dim3(stream_blocks(n)), dim3(kBlock), 0,
                       ctx->stream, p, v, a, n, dt);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_update_velocities(nbody_hip_ctx* ctx, nbody_particle_data* d, float dt) {
  if (int rc = check_pd(ctx, d)) return rc;
  const size_t n = d->count;
  if (n == 0) return NBODY_HIP_OK;
  if (!d->acc_old_x || !d->acc_old_y || !d->acc_old_z)
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "particle data has null acc_old arrays");
  NBH_HIP(hipSetDevice(ctx->device));
  Soa3 v{d->vel_x, d->vel_y, d->vel_z};
  CSoa3 ao{d->acc_old_x, d->acc_old_y, d->acc_old_z}, an{d->acc_x, d->acc_y, d->acc_z};
  if (all_aligned(d))
    hipLaunchKernelGGL(update_velocities_kernel<4>, dim3(stream_blocks((n + 3) / 4)), dim3(kBlock),
                       0, ctx->stream, v, ao, an, n, dt);
  else
    hipLaunchKernelGGL(update_velocities_kernel<1>, dim3(stream_blocks(n)), dim3(kBlock), 0,
                       ctx->stream, v, ao, an, n, dt);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_store_accelerations(nbody_hip_ctx* ctx, nbody_particle_data* d) {
  if (int rc = check_pd(ctx, d)) return rc;
  const size_t n = d->count;
  if (n == 0) return NBODY_HIP_OK;
  if (!d->acc_old_x || !d->acc_old_y || !d->acc_old_z)
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "particle data has null acc_old arrays");
  NBH_HIP(hipSetDevice(ctx->device));
  // a_old = a is a plain device copy: the DMA/blit path already runs at HBM speed
  // (integrator.cu:39-48 spends a kernel on it).
  const size_t bytes = n * sizeof(float);
  NBH_HIP(hipMemcpyAsync(d->acc_old_x, d->acc_x, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  NBH_HIP(hipMemcpyAsync(d->acc_old_y, d->acc_y, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  NBH_HIP(hipMemcpyAsync(d->acc_old_z, d->acc_z, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return NBODY_HIP_OK;
}

// a_old <- a and x += v dt + a dt^2/2 in ONE pass (the reference's storeAccelerations + updatePositions
// pair, integrator.cu:224-231): for the steps whose force evaluation is not the fused direct one
extern "C" int nbody_hip_drift(nbody_hip_ctx* ctx, nbody_particle_data* d, float dt) {
  if (int rc = check_pd(ctx, d)) return rc;
  const size_t n = d->count;
  if (n == 0) return NBODY_HIP_OK;
  if (!d->acc_old_x || !d->acc_old_y || !d->acc_old_z)
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "particle data has null acc_old arrays");
  NBH_HIP(hipSetDevice(ctx->device));
  Soa3 p{d->pos_x, d->pos_y, d->pos_z}, ao{d->acc_old_x, d->acc_old_y, d->acc_old_z};
  CSoa3 v{d->vel_x, d->vel_y, d->vel_z}, a{d->acc_x, d->acc_y, d->acc_z};
  hipLaunchKernelGGL(drift_pack_kernel, dim3(stream_blocks(n)), dim3(kBlock), 0, ctx->stream, p, v, a, ao,
                     d->mass, static_cast<float4*>(nullptr), n, dt);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_integrate_direct(nbody_hip_ctx* ctx, nbody_particle_data* d, float G,
                                          float eps2, float dt, int steps) {
  if (int rc = check_pd(ctx, d)) return rc;
  const size_t n = d->count;
  if (n == 0 || steps <= 0) return NBODY_HIP_OK;
  if (!d->acc_old_x || !d->acc_old_y || !d->acc_old_z)
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "particle data has null acc_old arrays");
  NBH_HIP(hipSetDevice(ctx->device));
  if (int rc = ctx->posm.reserve(n * sizeof(float4))) return rc;
  float4* posm = static_cast<float4*>(ctx->posm.ptr);
  Soa3 p{d->pos_x, d->pos_y, d->pos_z}, ao{d->acc_old_x, d->acc_old_y, d->acc_old_z};
  CSoa3 v{d->vel_x, d->vel_y, d->vel_z}, a{d->acc_x, d->acc_y, d->acc_z};
  for (int s = 0; s < steps; s++) {
    hipLaunchKernelGGL(drift_pack_kernel, dim3(stream_blocks(n)), dim3(kBlock), 0, ctx->stream, p,
                       v, a, ao, d->mass, posm, n, dt);
    NBH_LAUNCH_CHECK();
    if (int rc = direct_packed(ctx, posm, n, posm, n, G, eps2, nullptr, 0, d->acc_x, d->acc_y,
                               d->acc_z, d->vel_x, d->vel_y, d->vel_z, d->acc_old_x, d->acc_old_y,
                               d->acc_old_z, 0.5f * dt))
      return rc;
  }
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_drift_packed(nbody_hip_ctx* ctx, nbody_float4* posm,
                                      const nbody_float4* vel, const nbody_float4* acc,
                                      size_t count, float dt) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (count == 0) return NBODY_HIP_OK;
  if (!posm || !vel || !acc) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  NBH_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(drift_packed_kernel, dim3(stream_blocks(count)), dim3(kBlock), 0, ctx->stream,
                     reinterpret_cast<float4*>(posm), reinterpret_cast<const float4*>(vel),
                     reinterpret_cast<const float4*>(acc), count, dt);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_kick_packed(nbody_hip_ctx* ctx, nbody_float4* vel,
                                     const nbody_float4* acc_old, const nbody_float4* acc_new,
                                     size_t count, float dt) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (count == 0) return NBODY_HIP_OK;
  if (!vel || !acc_old || !acc_new) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  NBH_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(kick_packed_kernel, dim3(stream_blocks(count)), dim3(kBlock), 0, ctx->stream,
                     reinterpret_cast<float4*>(vel), reinterpret_cast<const float4*>(acc_old),
                     reinterpret_cast<const float4*>(acc_new), count, dt);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}
