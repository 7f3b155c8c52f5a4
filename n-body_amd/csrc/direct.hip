// direct.hip -- Direct N^2 all-pairs accelerations for gfx950 (MI355X).
//
// Replaces computeForcesDirectKernel / launchDirectForceKernel of the reference
// (src/cuda/force_direct.cu:10-98).  Same mathematics,
//     a_i = G * sum_j m_j (r_j - r_i) (|r_j - r_i|^2 + eps^2)^(-3/2),
// different machine mapping:
//   * bodies are packed float4 {x,y,z,m}; one 16-byte LDS broadcast read feeds a whole
//     wave (the reference reads 4 scalar shared arrays per pair, force_direct.cu:61-68);
//   * no per-pair branch: the self pair (and any coincident pair) has dx=dy=dz=0 and
//     contributes f*0 = 0; padded sources carry m = 0 (the reference tests
//     `global_j < N && global_j != i` on every pair, :60);
//   * each lane owns R targets in registers (ILP, 1/R LDS reads per pair);
//   * the source range is cut into `splits` sub-ranges (grid.y) so that small N still
//     fills 256 CUs x 4 SIMDs; per-split partial sums are combined in fp64 by the
//     finalize kernel, which also applies G and (optionally) the Velocity-Verlet kick;
//   * per-tile fp32 partial sums are folded into fp64 running sums every TS sources:
//     a single fp32 accumulator over 2.6e5..1e6 terms random-walks to ~3e-5 relative,
//     above the 1e-5 parity bar (SURVEY.md section 7 "hard parts").
//
// Roofline: FP32 VALU issue bound (12 VALU + 1 transcendental per pair), not HBM and
// not MFMA (no contraction to map).  Algorithmic HBM bytes per launch: 16 N read + 16 N write.

#include "common.h"

namespace nbh {

constexpr int TS = 256;  // sources per LDS tile = one float4 per thread of the block

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }

// ---------------------------------------------------------------------------------
// One source against R targets, scalar form.  12 VALU + 1 v_rsq_f32 per pair.
// GUARD: exact handling of coincident pairs when eps2 is 0 or so small that
// m * rsq(eps2)^3 overflows (then f*0 would be NaN): contribution forced to 0.
// ---------------------------------------------------------------------------------
template <int R, bool GUARD>
__device__ __forceinline__ void interact(const float4 s, const float (&xi)[R], const float (&yi)[R],
                                         const float (&zi)[R], float (&ax)[R], float (&ay)[R],
                                         float (&az)[R], const float eps2) {
#pragma unroll
  for (int r = 0; r < R; r++) {
    const float dx = s.x - xi[r];
    const float dy = s.y - yi[r];
    const float dz = s.z - zi[r];
    float f;
    if constexpr (GUARD) {
      const float d2 = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
      const float inv = rsq(d2 + eps2);
      f = d2 > 0.0f ? s.w * inv * (inv * inv) : 0.0f;
    } else {
      const float r2 = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, __builtin_fmaf(dz, dz, eps2)));
      const float inv = rsq(r2);
      f = (s.w * inv) * (inv * inv);
    }
    ax[r] = __builtin_fmaf(f, dx, ax[r]);
    ay[r] = __builtin_fmaf(f, dy, ay[r]);
    az[r] = __builtin_fmaf(f, dz, az[r]);
  }
}

// Packed form: two targets per v_pk_*_f32 instruction (R even), NS sources per call.  Written in
// phases over the NS * R/2 independent chains (differences and r^2, then every v_rsq, then the
// factors, then the sums): left to itself the compiler finishes one chain after the other and pads
// the dependent packed operations with hazard nops.
template <int R, int NS>
__device__ __forceinline__ void interact_pk(const float4 (&s)[NS], const f2 (&xi)[R / 2],
                                            const f2 (&yi)[R / 2], const f2 (&zi)[R / 2],
                                            f2 (&ax)[R / 2], f2 (&ay)[R / 2], f2 (&az)[R / 2],
                                            const float eps2) {
  constexpr int H = R / 2;
  const f2 e2 = {eps2, eps2};
  f2 dx[NS * H], dy[NS * H], dz[NS * H], g[NS * H];
#pragma unroll
  for (int q = 0; q < NS; q++) {
    const f2 sx = {s[q].x, s[q].x}, sy = {s[q].y, s[q].y}, sz = {s[q].z, s[q].z};
#pragma unroll
    for (int r = 0; r < H; r++) {
      const int c = q * H + r;
      dx[c] = sx - xi[r]; dy[c] = sy - yi[r]; dz[c] = sz - zi[r];
      g[c] = __builtin_elementwise_fma(dx[c], dx[c], __builtin_elementwise_fma(dy[c], dy[c], __builtin_elementwise_fma(dz[c], dz[c], e2)));
    }
  }
#pragma unroll
  for (int c = 0; c < NS * H; c++) {
    g[c].x = rsq(g[c].x);
    g[c].y = rsq(g[c].y);
  }
#pragma unroll
  for (int q = 0; q < NS; q++) {
    const f2 sm = {s[q].w, s[q].w};
#pragma unroll
    for (int r = 0; r < H; r++) {
      const int c = q * H + r;
      g[c] = (sm * g[c]) * (g[c] * g[c]);
    }
  }
#pragma unroll
  for (int q = 0; q < NS; q++) {
#pragma unroll
    for (int r = 0; r < H; r++) {
      const int c = q * H + r;
      ax[r] = __builtin_elementwise_fma(g[c], dx[c], ax[r]);
      ay[r] = __builtin_elementwise_fma(g[c], dy[c], ay[r]);
      az[r] = __builtin_elementwise_fma(g[c], dz[c], az[r]);
    }
  }
}

// ---------------------------------------------------------------------------------
// Main kernel.  grid = (ceil(n_tgt / (256 R)), splits); block = 256.
// partial[split][i] = {sum_x, sum_y, sum_z, 0} over the split's sources (G not applied).
// VARIANT 0: scalar VALU body, LDS-broadcast sources
// VARIANT 1: packed (v_pk_*_f32) body, LDS-broadcast sources
// VARIANT 2: scalar VALU body, sources fetched by wave-uniform (scalar-cache) loads, no LDS
// ---------------------------------------------------------------------------------
template <int R, int VARIANT, bool GUARD>
__global__ __launch_bounds__(kBlock) void direct_kernel(const float4* __restrict__ tgt, int n_tgt,
                                                        const float4* __restrict__ src, int n_src,
                                                        int src_per_split,
                                                        float4* __restrict__ partial,
                                                        int n_tgt_pad, float eps2) {
  const int tid = threadIdx.x;
  const int tbase = blockIdx.x * (kBlock * R);

  float xi[R], yi[R], zi[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int i = tbase + r * kBlock + tid;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n_tgt) p = tgt[i];
    xi[r] = p.x; yi[r] = p.y; zi[r] = p.z;
  }

  double sx[R], sy[R], sz[R];
#pragma unroll
  for (int r = 0; r < R; r++) sx[r] = sy[r] = sz[r] = 0.0;

  const int j0 = blockIdx.y * src_per_split;
  const int j1 = min(n_src, j0 + src_per_split);
  const int ntiles = (j1 - j0 + TS - 1) / TS;

  if constexpr (VARIANT == 2) {
    // Wave-uniform source index: the compiler turns src[j] into s_load_dwordx4/x8/x16
    // (scalar cache), the pair math reads the source straight from SGPRs.
    for (int t = 0; t < ntiles; t++) {
      float ax[R], ay[R], az[R];
#pragma unroll
      for (int r = 0; r < R; r++) ax[r] = ay[r] = az[r] = 0.f;
      const int jb = j0 + t * TS;
      const int cnt = min(TS, j1 - jb);
      if (cnt == TS) {
#pragma unroll 8
        for (int k = 0; k < TS; k++) interact<R, GUARD>(src[jb + k], xi, yi, zi, ax, ay, az, eps2);
      } else {
        for (int k = 0; k < cnt; k++) interact<R, GUARD>(src[jb + k], xi, yi, zi, ax, ay, az, eps2);
      }
#pragma unroll
      for (int r = 0; r < R; r++) { sx[r] += (double)ax[r]; sy[r] += (double)ay[r]; sz[r] += (double)az[r]; }
    }
  } else {
    __shared__ float4 tile[2][TS];
    auto load_src = [&](int j) -> float4 {
      return j < j1 ? src[j] : make_float4(0.f, 0.f, 0.f, 0.f);  // padded source: m = 0
    };
    float4 pre = load_src(j0 + tid);
    for (int t = 0; t < ntiles; t++) {
      const int b = t & 1;
      tile[b][tid] = pre;
      __syncthreads();  // one barrier per tile: the other buffer is only rewritten after the
                        // next barrier, when every wave has left this tile's predecessor
      pre = load_src(j0 + (t + 1) * TS + tid);  // next tile in flight under the math

      if constexpr (VARIANT == 1) {
        f2 px[R / 2], py[R / 2], pz[R / 2], ax[R / 2], ay[R / 2], az[R / 2];
#pragma unroll
        for (int r = 0; r < R / 2; r++) {
          px[r] = f2{xi[2 * r], xi[2 * r + 1]};
          py[r] = f2{yi[2 * r], yi[2 * r + 1]};
          pz[r] = f2{zi[2 * r], zi[2 * r + 1]};
          ax[r] = ay[r] = az[r] = f2{0.f, 0.f};
        }
#pragma unroll 4
        for (int k = 0; k < TS; k += 2) {
          const float4 two[2] = {tile[b][k], tile[b][k + 1]};
          interact_pk<R, 2>(two, px, py, pz, ax, ay, az, eps2);
        }
#pragma unroll
        for (int r = 0; r < R / 2; r++) {
          sx[2 * r] += (double)ax[r].x; sx[2 * r + 1] += (double)ax[r].y;
          sy[2 * r] += (double)ay[r].x; sy[2 * r + 1] += (double)ay[r].y;
          sz[2 * r] += (double)az[r].x; sz[2 * r + 1] += (double)az[r].y;
        }
      } else {
        float ax[R], ay[R], az[R];
#pragma unroll
        for (int r = 0; r < R; r++) ax[r] = ay[r] = az[r] = 0.f;
#pragma unroll 8
        for (int k = 0; k < TS; k++) interact<R, GUARD>(tile[b][k], xi, yi, zi, ax, ay, az, eps2);
#pragma unroll
        for (int r = 0; r < R; r++) { sx[r] += (double)ax[r]; sy[r] += (double)ay[r]; sz[r] += (double)az[r]; }
      }
    }
  }

  float4* out = partial + (size_t)blockIdx.y * n_tgt_pad;
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int i = tbase + r * kBlock + tid;  // i < n_tgt_pad by construction
    out[i] = make_float4((float)sx[r], (float)sy[r], (float)sz[r], 0.f);
  }
}

// ---------------------------------------------------------------------------------
// Finalize: a_i = G * sum_over_splits partial (fp64), then one of
//   acc4[i] (= or +=) {a,0}            packed output (sharded path)
//   ax/ay/az[i] = a                    SoA output (ParticleData ABI)
//   + v += (a_old + a) * half_dt       fused Velocity-Verlet kick (integrator.cu:31-34)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void direct_finalize_kernel(
    const float4* __restrict__ partial, int splits, int n_tgt_pad, int n, float G,
    float4* __restrict__ acc4, int accumulate, float* __restrict__ ax, float* __restrict__ ay,
    float* __restrict__ az, float* __restrict__ vx, float* __restrict__ vy, float* __restrict__ vz,
    const float* __restrict__ aox, const float* __restrict__ aoy, const float* __restrict__ aoz,
    float half_dt) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  double x = 0.0, y = 0.0, z = 0.0;
  for (int s = 0; s < splits; s++) {
    const float4 p = partial[(size_t)s * n_tgt_pad + i];
    x += (double)p.x; y += (double)p.y; z += (double)p.z;
  }
  const float fx = (float)((double)G * x), fy = (float)((double)G * y), fz = (float)((double)G * z);
  if (acc4) {
    float4 o = make_float4(fx, fy, fz, 0.f);
    if (accumulate) { const float4 c = acc4[i]; o.x += c.x; o.y += c.y; o.z += c.z; }
    acc4[i] = o;
  } else {
    ax[i] = fx; ay[i] = fy; az[i] = fz;
    if (vx) {
      vx[i] = kick1(vx[i], aox[i], fx, half_dt);
      vy[i] = kick1(vy[i], aoy[i], fy, half_dt);
      vz[i] = kick1(vz[i], aoz[i], fz, half_dt);
    }
  }
}

__global__ __launch_bounds__(kBlock) void pack_posm_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ y,
                                                           const float* __restrict__ z,
                                                           const float* __restrict__ m, int n,
                                                           float4* __restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = make_float4(x[i], y[i], z[i], m[i]);
}

int pack_posm(nbody_hip_ctx* ctx, const float* x, const float* y, const float* z, const float* m,
              size_t n, float4* out) {
  if (n == 0) return NBODY_HIP_OK;
  const int blocks = (int)((n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(pack_posm_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, x, y, z, m,
                     (int)n, out);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

// ---------------------------------------------------------------------------------
// Launch-shape selection.
// ---------------------------------------------------------------------------------
struct Shape { int R; int splits; int src_per_split; int blocks_x; int n_tgt_pad; };

static Shape choose_shape(const nbody_hip_ctx* ctx, size_t n_tgt, size_t n_src) {
  Shape s;
  // Targets per lane: 4 once every CU still gets >= 4 blocks; small problems prefer more
  // blocks over more ILP.
  // Measured on MI355X (tools/sweep_direct.py): 4 targets per lane wins from N = 2.6e5 up;
  // below ~3e4 targets more blocks matter more than ILP.
  int R = n_tgt >= 32768 ? 4 : 2;
  if (ctx->tune_tpl == 1 || ctx->tune_tpl == 2 || ctx->tune_tpl == 4) R = ctx->tune_tpl;
  s.R = R;
  s.blocks_x = (int)((n_tgt + (size_t)kBlock * R - 1) / ((size_t)kBlock * R));
  s.n_tgt_pad = s.blocks_x * kBlock * R;
  const int tiles = (int)((n_src + TS - 1) / TS);
  // Aim for >= 16 blocks per CU queued so the last wave of blocks is short (measured: 4096
  // blocks beat 1024/2048 at N = 2^20 by 3-7 %).
  int want = (kNumCU * 16 + s.blocks_x - 1) / s.blocks_x;
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  if (ctx->tune_splits > 0) want = ctx->tune_splits;
  if (want > tiles) want = tiles > 0 ? tiles : 1;
  const int tiles_per_split = (tiles + want - 1) / want;
  s.src_per_split = tiles_per_split * TS;
  s.splits = tiles_per_split > 0 ? (tiles + tiles_per_split - 1) / tiles_per_split : 1;
  if (s.splits < 1) s.splits = 1;
  return s;
}

template <int R, int V, bool GD>
static void launch_direct(const nbody_hip_ctx* ctx, const Shape& s, const float4* tgt, int n_tgt,
                          const float4* src, int n_src, float4* partial, float eps2) {
  hipLaunchKernelGGL((direct_kernel<R, V, GD>), dim3(s.blocks_x, s.splits), dim3(kBlock), 0,
                     ctx->stream, tgt, n_tgt, src, n_src, s.src_per_split, partial, s.n_tgt_pad,
                     eps2);
}

int direct_packed(nbody_hip_ctx* ctx, const float4* targets, size_t n_targets,
                  const float4* sources, size_t n_sources, float G, float eps2, float4* acc4,
                  int accumulate, float* ax, float* ay, float* az, float* vx, float* vy, float* vz,
                  const float* aox, const float* aoy, const float* aoz, float half_dt) {
  if (n_targets == 0) return NBODY_HIP_OK;
  if (n_targets > 0x7fffffffu / 2 || n_sources > 0x7fffffffu / 2)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "body count exceeds 2^30 (int indexing, ref: force_direct.cu:89)");
  if (!(eps2 >= 0.0f)) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "eps2 must be >= 0");

  // symmetric (Newton's third law) kernel: the all-pairs case on one device
  if (targets == sources && n_targets == n_sources && eps2 >= 1e-12f && symmetric_pays(ctx, n_targets))
    return direct_symmetric(ctx, targets, n_targets, G, eps2, acc4, accumulate, ax, ay, az, vx, vy,
                            vz, aox, aoy, aoz, half_dt);

  ctx->last_direct_kernel = 0;
  const Shape s = choose_shape(ctx, n_targets, n_sources);
  int rc = ctx->partial.reserve((size_t)s.splits * s.n_tgt_pad * sizeof(float4));
  if (rc) return rc;
  float4* partial = static_cast<float4*>(ctx->partial.ptr);

  // m * rsq(eps2)^3 must stay finite for the branch-free self-pair trick.
  const bool guard = eps2 < 1e-12f;
  // automatic choice: the packed body (measured 4.2e12 vs 3.5e12 pairs/s for the scalar body)
  int variant = ctx->tune_variant;
  if (variant < 0 || variant > 2) variant = s.R >= 2 ? 1 : 0;
  if (guard && variant == 1) variant = 0;
  const int nt = (int)n_targets, ns = (int)n_sources;

#define NBH_DISPATCH(RR)                                                                         \
  do {                                                                                           \
    if (guard) {                                                                                 \
      if (variant == 2) launch_direct<RR, 2, true>(ctx, s, targets, nt, sources, ns, partial, eps2); \
      else launch_direct<RR, 0, true>(ctx, s, targets, nt, sources, ns, partial, eps2);          \
    } else if (variant == 1) {                                                                   \
      launch_direct<(RR < 2 ? 2 : RR), 1, false>(ctx, s, targets, nt, sources, ns, partial, eps2); \
    } else if (variant == 2) {                                                                   \
      launch_direct<RR, 2, false>(ctx, s, targets, nt, sources, ns, partial, eps2);              \
    } else {                                                                                     \
      launch_direct<RR, 0, false>(ctx, s, targets, nt, sources, ns, partial, eps2);              \
    }                                                                                            \
  } while (0)

  if (ns > 0) {
    if (s.R == 4) NBH_DISPATCH(4);
    else if (s.R == 2) NBH_DISPATCH(2);
    else {
      if (variant == 1) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "packed variant needs targets_per_lane >= 2");
      NBH_DISPATCH(1);
    }
    NBH_LAUNCH_CHECK();
  }
#undef NBH_DISPATCH

  const int fblocks = (int)((n_targets + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(direct_finalize_kernel, dim3(fblocks), dim3(kBlock), 0, ctx->stream, partial,
                     ns > 0 ? s.splits : 0, s.n_tgt_pad, nt, G, acc4, accumulate, ax, ay, az, vx,
                     vy, vz, aox, aoy, aoz, half_dt);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

}  // namespace nbh
