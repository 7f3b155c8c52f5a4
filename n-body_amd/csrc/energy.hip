// energy.hip -- kinetic / potential energy reductions.
//
// Replaces computeKineticEnergyKernel / computePotentialEnergyKernel and the host partial sums
// of Integrator::computeKineticEnergy / computePotentialEnergy (src/cuda/integrator.cu:51-119,
// 252-289).  Differences by design:
//   * KE: wave64 shuffle reduction, fp64 partials, second-stage reduce on device (the reference
//     tree-reduces fp32 in shared memory and sums block partials in fp32 on the host, :261-268);
//   * PE: the reference walks j>i from global memory per thread (untiled, triangular imbalance,
//     :97-103).  Here every block sweeps ALL sources through LDS tiles like the force kernel and
//     skips j == i by index; the ordered-pair sum is halved.  Per-tile fp32 sums are folded into
//     fp64 so that 1e6-body totals keep their digits.

#include "common.h"

namespace nbh {

constexpr int PTS = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// block sum of one double per thread; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* lds4) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) lds4[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  return r;
}

__global__ __launch_bounds__(kBlock) void kinetic_kernel(const float* __restrict__ vx,
                                                         const float* __restrict__ vy,
                                                         const float* __restrict__ vz,
                                                         const float* __restrict__ m, size_t n,
                                                         double* __restrict__ partial) {
  __shared__ double red[4];
  double acc = 0.0;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const float v2 = vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i];  // integrator.cu:61
    acc += (double)(0.5f * m[i] * v2);                               // :62
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// Sum `count` doubles with ONE block, fixed order -> bitwise reproducible.
__global__ __launch_bounds__(kBlock) void final_sum_kernel(const double* __restrict__ partial,
                                                           int count, double scale,
                                                           double* __restrict__ out) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < count; i += kBlock) acc += partial[i];
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) out[0] = s * scale;
}

// PE: block = 256 targets; grid.y splits the source range.  partial[by * gridDim.x + bx] =
// sum_i m_i * sum_{j in split, j != i} m_j / sqrt(r_ij^2 + eps^2)
// Targets and sources may be different arrays (one shard against the gathered bodies): target i
// is source number self_offset + i (no such source when the sets are disjoint).
// TRI (targets == sources, one body set): every unordered pair once, from its lower index -- the
// reference's j > i loop (integrator.cu:97) at tile granularity: source tiles that lie entirely
// below the block's own bodies are not even loaded.  Half the pair evaluations of the ordered sum.
template <bool TRI>
__global__ __launch_bounds__(kBlock) void potential_kernel(const float4* __restrict__ tgt, int nt,
                                                           long long self_offset,
                                                           const float4* __restrict__ posm, int n,
                                                           int src_per_split, float eps2,
                                                           double* __restrict__ partial) {
  __shared__ float4 tile[2][PTS];
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int ti = blockIdx.x * kBlock + tid;
  float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ti < nt) pi = tgt[ti];
  const long long self = self_offset + ti;
  const int i = (self >= 0 && self < n) ? (int)self : -1;
  const int j0 = blockIdx.y * src_per_split;
  const int j1 = min(n, j0 + src_per_split);
  const int ntiles = (j1 - j0 + PTS - 1) / PTS;
  auto load_src = [&](int j) -> float4 {
    return j < j1 ? posm[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  double total = 0.0;
  // first tile that reaches the block's own bodies (block-uniform)
  const int tstart = TRI ? max(0, ((int)blockIdx.x * kBlock - j0) / PTS) : 0;
  float4 pre = load_src(j0 + tstart * PTS + tid);
  for (int t = tstart; t < ntiles; t++) {
    const int b = t & 1;
    tile[b][tid] = pre;
    __syncthreads();
    pre = load_src(j0 + (t + 1) * PTS + tid);
    const int jb = j0 + t * PTS;
    float acc = 0.f;
#pragma unroll 8
    for (int k = 0; k < PTS; k++) {
      const float4 s = tile[b][k];
      const float dx = s.x - pi.x, dy = s.y - pi.y, dz = s.z - pi.z;
      const float r2 = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, __builtin_fmaf(dz, dz, eps2)));
      // j == i skipped by index (integrator.cu:97 starts at j = i+1); a padded source has m = 0.
      // r2 == 0 only when eps == 0 and the bodies coincide: the reference divides by zero there;
      // here such a pair contributes nothing.
      const float inv = __builtin_amdgcn_rsqf(r2);
      const bool take = TRI ? (jb + k > i) : (jb + k != i);
      acc += (take & (r2 > 0.f)) ? s.w * inv : 0.f;
    }
    total += (double)acc;
  }
  total *= (double)pi.w;
  const double s = block_sum(total, red);
  if (tid == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

__global__ __launch_bounds__(kBlock) void kinetic_packed_kernel(const float4* __restrict__ posm,
                                                                const float4* __restrict__ vel, size_t n,
                                                                double* __restrict__ partial) {
  __shared__ double red[4];
  double acc = 0.0;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const float4 v = vel[i];
    const float v2 = v.x * v.x + v.y * v.y + v.z * v.z;  // integrator.cu:61
    acc += (double)(0.5f * posm[i].w * v2);              // :62
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

}  // namespace nbh

using namespace nbh;

static int energy_check(const nbody_hip_ctx* ctx, const nbody_particle_data* d, const void* out) {
  if (ctx) NBH_NOT_CAPTURABLE(ctx, "an energy reduction");
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (!d || !out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (d->count > 0x3fffffffu) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "body count exceeds 2^30");
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_kinetic_energy_f64(nbody_hip_ctx* ctx, const nbody_particle_data* d,
                                            double* out) {
  if (int rc = energy_check(ctx, d, out)) return rc;
  const size_t n = d->count;
  if (n == 0) { *out = 0.0; return NBODY_HIP_OK; }
  if (!d->vel_x || !d->vel_y || !d->vel_z || !d->mass)
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "particle data has null arrays");
  NBH_HIP(hipSetDevice(ctx->device));
  size_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > 1024) blocks = 1024;
  if (int rc = ctx->reduce.reserve((blocks + 1) * sizeof(double))) return rc;
  double* partial = static_cast<double*>(ctx->reduce.ptr);
  hipLaunchKernelGGL(kinetic_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, d->vel_x,
                     d->vel_y, d->vel_z, d->mass, n, partial);
  NBH_LAUNCH_CHECK();
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, partial, (int)blocks,
                     1.0, partial + blocks);
  NBH_LAUNCH_CHECK();
  NBH_HIP(hipMemcpyAsync(ctx->host_scalar, partial + blocks, sizeof(double), hipMemcpyDeviceToHost,
                         ctx->stream));
  NBH_HIP(hipStreamSynchronize(ctx->stream));
  *out = ctx->host_scalar[0];
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_potential_energy_f64(nbody_hip_ctx* ctx, const nbody_particle_data* d,
                                              float G, float eps, double* out) {
  if (int rc = energy_check(ctx, d, out)) return rc;
  const size_t n = d->count;
  if (n == 0) { *out = 0.0; return NBODY_HIP_OK; }
  if (!d->pos_x || !d->pos_y || !d->pos_z || !d->mass)
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "particle data has null arrays");
  NBH_HIP(hipSetDevice(ctx->device));
  if (int rc = ctx->posm.reserve(n * sizeof(float4))) return rc;
  float4* posm = static_cast<float4*>(ctx->posm.ptr);
  if (int rc = pack_posm(ctx, d->pos_x, d->pos_y, d->pos_z, d->mass, n, posm)) return rc;
  const int bx = (int)((n + kBlock - 1) / kBlock);
  const int tiles = (int)((n + PTS - 1) / PTS);
  int splits = (kNumCU * 8 + bx - 1) / bx;
  if (splits > 64) splits = 64;
  if (splits > tiles) splits = tiles;
  if (splits < 1) splits = 1;
  const int tiles_per_split = (tiles + splits - 1) / splits;
  splits = (tiles + tiles_per_split - 1) / tiles_per_split;
  const size_t np = (size_t)bx * splits;
  if (int rc = ctx->reduce.reserve((np + 1) * sizeof(double))) return rc;
  double* partial = static_cast<double*>(ctx->reduce.ptr);
  hipLaunchKernelGGL(potential_kernel<true>, dim3(bx, splits), dim3(kBlock), 0, ctx->stream, posm, (int)n, 0LL,
                     posm, (int)n, tiles_per_split * PTS, eps * eps, partial);
  NBH_LAUNCH_CHECK();
  // every unordered pair once; sign and G here (integrator.cu:102: pe -= G mi mj / r)
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, partial, (int)np,
                     -(double)G, partial + np);
  NBH_LAUNCH_CHECK();
  NBH_HIP(hipMemcpyAsync(ctx->host_scalar, partial + np, sizeof(double), hipMemcpyDeviceToHost,
                         ctx->stream));
  NBH_HIP(hipStreamSynchronize(ctx->stream));
  *out = ctx->host_scalar[0];
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_kinetic_energy(nbody_hip_ctx* ctx, const nbody_particle_data* d,
                                        float* out) {
  double v = 0.0;
  if (!out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (int rc = nbody_hip_kinetic_energy_f64(ctx, d, &v)) return rc;
  *out = (float)v;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_potential_energy(nbody_hip_ctx* ctx, const nbody_particle_data* d, float G,
                                          float eps, float* out) {
  double v = 0.0;
  if (!out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (int rc = nbody_hip_potential_energy_f64(ctx, d, G, eps, &v)) return rc;
  *out = (float)v;
  return NBODY_HIP_OK;
}

// e: energies of one shard of a sharded run (the sum over the shards is the system's KE and PE)
extern "C" int nbody_hip_energies_packed(nbody_hip_ctx* ctx, const nbody_float4* targets_posm,
                                         const nbody_float4* targets_vel, size_t n_targets,
                                         long long self_offset, const nbody_float4* sources_posm,
                                         size_t n_sources, float G, float eps, double out[2]) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  NBH_NOT_CAPTURABLE(ctx, "an energy reduction");
  if (!out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  out[0] = out[1] = 0.0;
  if (n_targets == 0) return NBODY_HIP_OK;
  if (!targets_posm || !targets_vel || (n_sources > 0 && !sources_posm))
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (n_targets > 0x3fffffffu || n_sources > 0x3fffffffu)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "body count exceeds 2^30");
  NBH_HIP(hipSetDevice(ctx->device));
  const float4* tp = reinterpret_cast<const float4*>(targets_posm);
  const float4* tv = reinterpret_cast<const float4*>(targets_vel);
  const float4* sp = reinterpret_cast<const float4*>(sources_posm);
  const int bx = (int)((n_targets + kBlock - 1) / kBlock);
  const int kb = bx > 1024 ? 1024 : bx;
  const int tiles = (int)((n_sources + PTS - 1) / PTS);
  int splits = (kNumCU * 8 + bx - 1) / bx;
  if (splits > 64) splits = 64;
  if (splits > tiles) splits = tiles;
  if (splits < 1) splits = 1;
  const int tiles_per_split = tiles > 0 ? (tiles + splits - 1) / splits : 1;
  splits = tiles > 0 ? (tiles + tiles_per_split - 1) / tiles_per_split : 0;
  const size_t np = (size_t)bx * splits;
  if (int rc = ctx->reduce.reserve((np + (size_t)kb + 2) * sizeof(double))) return rc;
  double* partial = static_cast<double*>(ctx->reduce.ptr);
  double* kpart = partial + np;
  double* res = kpart + kb;  // {ke, pe}
  hipLaunchKernelGGL(kinetic_packed_kernel, dim3(kb), dim3(kBlock), 0, ctx->stream, tp, tv, n_targets, kpart);
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, kpart, kb, 1.0, res);
  if (np > 0) {
    hipLaunchKernelGGL(potential_kernel<false>, dim3(bx, splits), dim3(kBlock), 0, ctx->stream, tp, (int)n_targets,
                       self_offset, sp, (int)n_sources, tiles_per_split * PTS, eps * eps, partial);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, partial, (int)np,
                       -0.5 * (double)G, res + 1);
  } else {
    NBH_HIP(hipMemsetAsync(res + 1, 0, sizeof(double), ctx->stream));
  }
  NBH_LAUNCH_CHECK();
  NBH_HIP(hipMemcpyAsync(ctx->host_scalar, res, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NBH_HIP(hipStreamSynchronize(ctx->stream));
  out[0] = ctx->host_scalar[0];
  out[1] = ctx->host_scalar[1];
  return NBODY_HIP_OK;
}
