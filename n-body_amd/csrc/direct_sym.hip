// direct_sym.hip -- Direct N^2 exploiting Newton's third law (action = -reaction), gfx950.
//
// Same sum as direct.hip (and as computeForcesDirectKernel, src/cuda/force_direct.cu:10-85):
//     a_i = G sum_{j != i} m_j (r_j - r_i) (|r_j - r_i|^2 + eps^2)^(-3/2),
// but every UNORDERED pair {i, j} is evaluated once and used twice: the shared factor
// g = (d^2 + eps^2)^(-3/2) costs 3 sub + 3 fma + 1 rsq + 2 mul, then a_i += (g m_j) d and
// a_j -= (g m_i) d are 2 mul + 6 fma: 20 issue slots per unordered pair = 10 per ordered pair,
// against 16 for the one-sided kernel (the rsq alone is 4 slots, measured).
//
// Mapping.  Bodies are cut into superblocks of S = 256 R bodies.  Workgroup (A, split) holds
// superblock A in REGISTERS (each lane R bodies, the "I side") and walks the partner superblocks
// B = A + d (mod NB), d in its share of 0 .. NB/2 (a half ring: every unordered superblock pair
// exactly once; d = 0 is the diagonal, handled without reaction).  A partner is visited in chunks
// of 64 bodies (the "J side"): every wave loads the chunk with lane l holding J body l, together
// with a reaction accumulator that TRAVELS WITH THE BODY.  In each of 64 steps a lane pairs its R
// bodies with the J body it currently holds, adds the reaction into the travelling accumulator,
// and the seven J registers {x, y, z, m, hx, hy, hz} are rotated one lane along the wave
// (v_mov_b32_dpp wave_rol:1).  After 64 steps everything is back in its home lane.  The four
// waves of the block then combine their reactions through LDS (plain stores + barrier) and one
// wave adds the 64 x 3 sums to the global fp64 accumulator with hardware f64 atomics (lossless
// across XCDs, measured ~180 G adds/s).  The I-side sums stay in registers for the whole kernel
// (fp32 per 4 chunks, folded into fp64) and are added once at the end.
//
// With the atomics the result depends on their order in the last bit; the DET form (default for all pairs, see
// below) gives every contribution a slot of its own and is bitwise reproducible like direct.hip; agreement with
// the oracle is the same 1e-5 either way.  Used for the single-GPU all-pairs case (targets == sources) and, as
// RECT, for shard pairs; the one-sided kernel remains the path for N < 12,288 and for eps^2 < 1e-12.

#include <type_traits>

#include "common.h"

namespace nbh {

typedef float f2 __attribute__((ext_vector_type(2)));

// one-lane rotation of a wave64 register: lane l receives the value of lane (l + 1) & 63
__device__ __forceinline__ float wave_rot1(float v) {
  // mov_dpp (no "old" operand): every lane is written, so nothing has to be initialised first
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x134 /* wave_rol:1 */, 0xF, 0xF, false));
}

// R even.  grid = (NB_I, splits), block = 256.
// RECT = false: one body set; partner offsets d in 0 .. NB/2 (half ring), d = 0 without reaction.
// RECT = true : two DISJOINT body sets I (n bodies, posm) and J (nj bodies, posj): every I x J pair
//               once; offsets d = 0 .. NBJ-1 address the J superblocks directly; the reactions go to
//               accj64 (the sharded path sends them back to the owner of the J bodies).
//
// EQM: equal-mass specialisation.  When every body of the call has the same mass m0 > 0 (decided
// on the DEVICE from a min/max reduction, no host round trip: both instantiations are launched
// and the one that does not apply returns at once) the two multiplications g*m_j and g*m_i are
// dropped from every pair (18 instead of 20 issue slots per unordered pair) and m0 is applied once
// to the finished sums.
#ifndef NBH_SYM_ATTR
#define NBH_SYM_ATTR
#endif
constexpr int kSymBlocksPerCU = 16;  // workgroups aimed at per CU (tools/sweep_mid.py)
constexpr int kSymMinChunks = 4;     // at least 4 chunks (256 J bodies) per workgroup
constexpr float kFar = 1.0e18f;  // padding bodies sit here: (3e36)^-3/2 underflows to 0, no mass test needed

// DET: deterministic sums.  Instead of fp64 atomics every contribution gets a slot of its own, written once
// with plain (coalesced) stores; the finalize kernel adds a body's slots in a fixed order.  Bitwise
// reproducible like the reference's one-thread-per-body loop (force_direct.cu:10-85).
//   all pairs : the reaction of the partner at ring offset d -> REACTION slot d - 1 of the receiving body,
//               the I-side sum of split s -> I-SIDE slot s of the body's own superblock;
//   two sets  : the reaction of I superblock A on a J body -> reaction slot A of the J set, the I-side sum
//               of split s -> I-side slot s of the I set.
// A reaction is ONE fp32 number (the sum of the four waves' fp32 sums of a 64-step rotation), so its slot is
// a float: 12 bytes per (body, partner) instead of the 24 of round 2; the I-side sums are fp64.  The equal-
// mass factor m0 is applied by the finalize kernel.  Slot planes: 12 D + 24 splits bytes per body
// (1.9 GB at N = 2^20, 16 bodies per lane: D = 128, 16 splits).
//   acc64  : I-side accumulator -- atomics: [3][plane_i] doubles; DET: [splits][3][plane_i] doubles
//   accj   : reaction accumulator -- atomics: [3][plane] doubles (all pairs: the same array as acc64; two sets:
//            the J set's); DET: [D or NBI][3][plane] FLOATS
template <int R, bool RECT, bool EQM, bool DET = false>
__global__ __launch_bounds__(kBlock) NBH_SYM_ATTR void direct_sym_kernel(const float4* __restrict__ posm, int n,
                                                            const float4* __restrict__ posj, int nj,
                                                            int NB, int NBJ, int chunks_per_split,
                                                            double* __restrict__ acc64,
                                                            void* __restrict__ accj,
                                                            size_t plane_i, size_t plane_j,
                                                            const unsigned int* __restrict__ mass_range,
                                                            float eps2) {
  // mass_range = order-preserving encodings {min_I, max_I, min_J, max_J}
  const bool uniform = mass_range[0] == mass_range[1] && mass_range[2] == mass_range[3] &&
                       mass_range[0] == mass_range[2] && ordered_to_float(mass_range[0]) > 0.f;
  if (uniform != EQM) return;  // the other instantiation handles this call
  const float m0 = EQM ? ordered_to_float(mass_range[0]) : 1.0f;
  constexpr int S = kBlock * R;
  __shared__ float slab[2][4][3][64];  // per-wave reaction sums of a chunk, double buffered
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int A = blockIdx.x;
  const int D = RECT ? NBJ - 1 : NB / 2;     // largest partner offset
  // the block's share of the flat list of (partner offset d, 64-body chunk c) pairs: splitting at
  // chunk granularity (not whole partners) gives enough workgroups to fill the last round at
  // mid sizes (65,536 bodies: 64 superblocks x 33 offsets is only two rounds of the chip)
  constexpr int CPB = kBlock * R / 64;
  const int q0 = blockIdx.y * chunks_per_split;
  const int q1 = min((D + 1) * CPB, q0 + chunks_per_split);
  if (q0 >= q1) {
    if constexpr (DET) {  // an empty share still owns an I-side slot: it must read as zero
#pragma unroll
      for (int r = 0; r < R; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
          acc64[((size_t)blockIdx.y * 3 + c) * plane_i + (size_t)(A * S + r * kBlock + tid)] = 0.0;
    }
    return;
  }

  // I side: R bodies per lane, packed in pairs
  f2 xi[R / 2], yi[R / 2], zi[R / 2], mi[R / 2];
#pragma unroll
  for (int r = 0; r < R / 2; r++) {
    float4 p0 = make_float4(kFar, kFar, kFar, 0.f), p1 = p0;
    const int i0 = A * S + (2 * r) * kBlock + tid, i1 = i0 + kBlock;
    if (i0 < n) p0 = posm[i0];
    if (i1 < n) p1 = posm[i1];
    xi[r] = f2{p0.x, p1.x}; yi[r] = f2{p0.y, p1.y}; zi[r] = f2{p0.z, p1.z}; mi[r] = f2{p0.w, p1.w};
  }
  // fp64 sums of the I side.  R = 16 would need 96 VGPRs for them and push the kernel past the 256
  // architectural VGPRs (the rest are AGPRs, reached through copies -- measured: the fastest and
  // the slowest variant, depending on the allocator's mood); they live in LDS there instead,
  // touched once per 4 chunks, [component][r][thread] so that a wave's accesses are contiguous.
  constexpr bool LACC = R >= 8;
  constexpr int RR = LACC ? 1 : R;
  __shared__ double lacc[LACC ? 3 * R * kBlock : 1];
  double sx[RR], sy[RR], sz[RR];
  if constexpr (LACC) {
#pragma unroll
    for (int k = 0; k < 3 * R; k++) lacc[k * kBlock + tid] = 0.0;  // own slots only: no barrier needed
  } else {
#pragma unroll
    for (int r = 0; r < R; r++) sx[r] = sy[r] = sz[r] = 0.0;
  }
  f2 ax[R / 2], ay[R / 2], az[R / 2];
#pragma unroll
  for (int r = 0; r < R / 2; r++) ax[r] = ay[r] = az[r] = f2{0.f, 0.f};
  const f2 e2 = {eps2, eps2};

  // flat list of chunks: (offset d, chunk c of the partner); S/64 = CPB chunks per partner
  // partner d is skipped when the half ring would visit the pair twice (NB even, d = NB/2, upper half)
  auto partner_valid = [&](int d) { return RECT || !((NB % 2 == 0) && d == D && d > 0 && A >= D); };
  const int nchunks = q1 - q0;

  auto load_chunk = [&](int q) -> float4 {  // q-th chunk of this block's list, body `lane`
    const int d = (q0 + q) / CPB, c = (q0 + q) % CPB;
    const int B = RECT ? d : (A + d) % NB;
    const int j = B * S + c * 64 + lane;
    float4 p = make_float4(kFar, kFar, kFar, 0.f);
    if (RECT) { if (j < nj) p = posj[j]; }
    else if (j < n && partner_valid(d)) p = posm[j];
    return p;  // invalid partner / padding: zero mass, far away -> contributes nothing
  };
  auto flush_chunk = [&](int q) {  // one wave: combine the four waves' sums of chunk q, add to global
    const int pd = (q0 + q) / CPB, pc = (q0 + q) % CPB;
    if constexpr (DET) {
      if (RECT || pd > 0) {  // all pairs: slot pd - 1 of the partner's bodies (a partner the half ring skips
        // contributes zero); two sets: slot A of the J bodies.  Wave-uniform slot base, 32-bit element offsets
        // (a slot's three planes stay far below 2^32 floats)
        const size_t plane = RECT ? plane_j : plane_i;
        float* slot = static_cast<float*>(accj) + (size_t)(RECT ? A : pd - 1) * 3 * plane;
        const unsigned int j = (unsigned int)((RECT ? pd : (A + pd) % NB) * S + pc * 64 + lane), pl = (unsigned int)plane;
        const int sb = q & 1;
        const bool keep = partner_valid(pd);
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float v = (slab[sb][0][c][lane] + slab[sb][1][c][lane]) + (slab[sb][2][c][lane] + slab[sb][3][c][lane]);
          slot[c * pl + j] = keep ? v : 0.f;
        }
      }
    } else if ((RECT || pd > 0) && partner_valid(pd)) {
      const int j = (RECT ? pd : (A + pd) % NB) * S + pc * 64 + lane;
      const int sb = q & 1;
      double* dst = RECT ? static_cast<double*>(accj) : acc64;
      const size_t plane = RECT ? plane_j : plane_i;
#pragma unroll
      for (int c = 0; c < 3; c++) {  // component planes: 64 lanes x 8 B contiguous per atomic instruction
        const float v = (slab[sb][0][c][lane] + slab[sb][1][c][lane]) + (slab[sb][2][c][lane] + slab[sb][3][c][lane]);
        unsafeAtomicAdd(&dst[c * plane + (size_t)j], (double)v * (double)m0);
      }
    }
  };

  float4 jnext = load_chunk(0);
  for (int q = 0; q < nchunks; q++) {
    const float4 jcur = jnext;
    if (q + 1 < nchunks) jnext = load_chunk(q + 1);  // next chunk in flight under the math
    const int d = (q0 + q) / CPB;
    const bool react = RECT || d > 0;  // the diagonal superblock pair sees every ordered pair already
    if (q > 0) {
      __syncthreads();  // every wave has stored its sums of chunk q-1
      if (w == (q & 3)) flush_chunk(q - 1);
    }
    float jx = jcur.x, jy = jcur.y, jz = jcur.z, jm = jcur.w;
    float hjx = 0.f, hjy = 0.f, hjz = 0.f;
    auto steps = [&](auto react_tag) {
      constexpr bool REACT = decltype(react_tag)::value;
      constexpr int kStepUnroll = R >= 16 ? 1 : 2;  // two steps in flight, registers permitting
#pragma unroll kStepUnroll
      for (int k = 0; k < 64; k++) {
        const f2 sxj = {jx, jx}, syj = {jy, jy}, szj = {jz, jz}, smj = {jm, jm};
        // phases over the R/2 packed pairs: the dependent chain of one pair (r2 -> rsq -> g -> sums)
        // is interleaved with the other pairs' instead of being padded with hazard nops
        f2 dx[R / 2], dy[R / 2], dz[R / 2], g[R / 2];
#pragma unroll
        for (int r = 0; r < R / 2; r++) {
          dx[r] = sxj - xi[r]; dy[r] = syj - yi[r]; dz[r] = szj - zi[r];
          g[r] = __builtin_elementwise_fma(dx[r], dx[r], __builtin_elementwise_fma(dy[r], dy[r], __builtin_elementwise_fma(dz[r], dz[r], e2)));
        }
#pragma unroll
        for (int r = 0; r < R / 2; r++) {
          g[r].x = __builtin_amdgcn_rsqf(g[r].x);
          g[r].y = __builtin_amdgcn_rsqf(g[r].y);
        }
#pragma unroll
        for (int r = 0; r < R / 2; r++) g[r] = (g[r] * g[r]) * g[r];
        // the travelling reaction sum rides in the low half of the packed accumulator
        f2 hx = {hjx, 0.f}, hy = {hjy, 0.f}, hz = {hjz, 0.f};
#pragma unroll
        for (int r = 0; r < R / 2; r++) {
          const f2 fj = EQM ? g[r] : g[r] * smj;    // on the I bodies, from J
          ax[r] = __builtin_elementwise_fma(fj, dx[r], ax[r]);
          ay[r] = __builtin_elementwise_fma(fj, dy[r], ay[r]);
          az[r] = __builtin_elementwise_fma(fj, dz[r], az[r]);
          if (REACT) {
            const f2 fi = EQM ? g[r] : g[r] * mi[r];  // on the J body, from the I bodies
            hx = __builtin_elementwise_fma(-fi, dx[r], hx);  // a_j -= g m_i d  (neg is an operand modifier)
            hy = __builtin_elementwise_fma(-fi, dy[r], hy);
            hz = __builtin_elementwise_fma(-fi, dz[r], hz);
          }
        }
        if (REACT) {
          hjx = wave_rot1(hx.x + hx.y); hjy = wave_rot1(hy.x + hy.y); hjz = wave_rot1(hz.x + hz.y);
        }
        jx = wave_rot1(jx); jy = wave_rot1(jy); jz = wave_rot1(jz); jm = wave_rot1(jm);
      }
    };
    if constexpr (R >= 12 && !EQM) {
      // GENERAL masses at 12 / 16 bodies per lane: ONE loop body -- the diagonal superblock pair (d = 0: one partner
      // in D + 1, i.e. 0.8 % of the chunks at N = 2^20) computes its reactions too and drops them at the flush.  Two
      // instantiations of the 64-step loop cost registers: the slot kernel at R = 16 needed 256 VGPRs + 29 AGPR
      // copies (190 ms at N = 2^20, which is why round 2 ran it at R = 12); with one body it fits 246 (171.5 ms).
      // The equal-mass kernel keeps both bodies: with one it needs fewer registers (226) but the schedule the
      // compiler then picks is slower (R = 16: 170 ms against 158, measured, tools/direct_det_probe.py).
      (void)react;
      steps(std::true_type{});
    } else {
      if (react) steps(std::true_type{}); else steps(std::false_type{});
    }
    // after 64 one-lane rotations every J body (and its reaction sum) is back in its home lane
    slab[q & 1][w][0][lane] = hjx; slab[q & 1][w][1][lane] = hjy; slab[q & 1][w][2][lane] = hjz;
    if ((q & 3) == 3 || q == nchunks - 1) {  // fold the fp32 sums of <= 256 sources into fp64
#pragma unroll
      for (int r = 0; r < R / 2; r++) {
        if constexpr (LACC) {
          lacc[(0 * R + 2 * r) * kBlock + tid] += (double)ax[r].x; lacc[(0 * R + 2 * r + 1) * kBlock + tid] += (double)ax[r].y;
          lacc[(1 * R + 2 * r) * kBlock + tid] += (double)ay[r].x; lacc[(1 * R + 2 * r + 1) * kBlock + tid] += (double)ay[r].y;
          lacc[(2 * R + 2 * r) * kBlock + tid] += (double)az[r].x; lacc[(2 * R + 2 * r + 1) * kBlock + tid] += (double)az[r].y;
        } else {
          sx[2 * r] += (double)ax[r].x; sx[2 * r + 1] += (double)ax[r].y;
          sy[2 * r] += (double)ay[r].x; sy[2 * r + 1] += (double)ay[r].y;
          sz[2 * r] += (double)az[r].x; sz[2 * r + 1] += (double)az[r].y;
        }
        ax[r] = ay[r] = az[r] = f2{0.f, 0.f};
      }
    }
  }
  __syncthreads();
  if (w == (nchunks & 3)) flush_chunk(nchunks - 1);
  // I side
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int i = A * S + r * kBlock + tid;  // < NB * S: the accumulator is padded to that
    const double fx = LACC ? lacc[(0 * R + r) * kBlock + tid] : sx[LACC ? 0 : r];
    const double fy = LACC ? lacc[(1 * R + r) * kBlock + tid] : sy[LACC ? 0 : r];
    const double fz = LACC ? lacc[(2 * R + r) * kBlock + tid] : sz[LACC ? 0 : r];
    if constexpr (DET) {  // (m0 is applied by the finalize kernel)
      double* slot = acc64 + (size_t)blockIdx.y * 3 * plane_i;
      const unsigned int iu = (unsigned int)i, pl = (unsigned int)plane_i;
      slot[iu] = fx;
      slot[pl + iu] = fy;
      slot[2u * pl + iu] = fz;
    } else {
      unsafeAtomicAdd(&acc64[(size_t)i], fx * (double)m0);
      unsafeAtomicAdd(&acc64[plane_i + (size_t)i], fy * (double)m0);
      unsafeAtomicAdd(&acc64[2 * plane_i + (size_t)i], fz * (double)m0);
    }
  }
}

__global__ void mass_range_init_kernel(unsigned int* enc) {
  if (threadIdx.x < 4) enc[threadIdx.x] = (threadIdx.x & 1) ? 0u : 0xffffffffu;  // {min, max, min, max}
}

// min / max of the masses of a packed body set -> enc[0], enc[1] (order-preserving integers)
__global__ __launch_bounds__(kBlock) void mass_range_kernel(const float4* __restrict__ posm, int n,
                                                            unsigned int* __restrict__ enc) {
  __shared__ float red[4][2];
  float lo = INFINITY, hi = -INFINITY;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float m = posm[i].w;
    lo = fminf(lo, m); hi = fmaxf(hi, m);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_down(lo, off, 64));
    hi = fmaxf(hi, __shfl_down(hi, off, 64));
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w][0] = lo; red[w][1] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; k++) { lo = fminf(lo, red[k][0]); hi = fmaxf(hi, red[k][1]); }
    atomicMin(&enc[0], float_to_ordered(fminf(lo, red[0][0])));
    atomicMax(&enc[1], float_to_ordered(fmaxf(hi, red[0][1])));
  }
}

// Where a body's contributions lie: n_i I-side slots (fp64, [slot][3][plane]) and n_r reaction slots (fp32, same
// shape).  Atomic mode: one fp64 slot that already holds everything (n_i = 1, n_r = 0).
struct SlotLayout {
  const double* isl;
  const float* rsl;
  size_t plane;
  int n_i, n_r;
};

// acc = G * (sum of the body's slots, in a fixed order) ; SoA or float4 output ; optional fused Velocity-Verlet kick
// apply_m0: deterministic mode -- the equal-mass kernel stores its sums without the common mass factor
__global__ __launch_bounds__(kBlock) void direct_sym_finalize_kernel(
    SlotLayout eq, SlotLayout gen, int apply_m0, const unsigned int* __restrict__ mass_range, int n, float G,
    float4* __restrict__ acc4, int accumulate, float* __restrict__ ax, float* __restrict__ ay, float* __restrict__ az,
    float* __restrict__ vx, float* __restrict__ vy, float* __restrict__ vz, const float* __restrict__ aox,
    const float* __restrict__ aoy, const float* __restrict__ aoz, float half_dt) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  SlotLayout L = eq;
  double m0 = 1.0;
  if (mass_range) {  // the general-mass instantiation ran with a layout of its own (other bodies per lane)
    const bool uniform = mass_range[0] == mass_range[1] && mass_range[2] == mass_range[3] &&
                         mass_range[0] == mass_range[2] && ordered_to_float(mass_range[0]) > 0.f;
    if (!uniform) L = gen;
    else if (apply_m0) m0 = (double)ordered_to_float(mass_range[0]);
  }
  double sx = 0.0, sy = 0.0, sz = 0.0;
  for (int k = 0; k < L.n_r; k++) {
    const float* p = L.rsl + (size_t)k * 3 * L.plane;
    sx += (double)p[i]; sy += (double)p[L.plane + i]; sz += (double)p[2 * L.plane + i];
  }
  for (int k = 0; k < L.n_i; k++) {
    const double* p = L.isl + (size_t)k * 3 * L.plane;
    sx += p[i]; sy += p[L.plane + i]; sz += p[2 * L.plane + i];
  }
  const double gm = (double)G * m0;
  const float fx = (float)(gm * sx), fy = (float)(gm * sy), fz = (float)(gm * sz);
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

static void launch_mass_range(nbody_hip_ctx* ctx, const float4* pi, int ni, const float4* pj, int nj,
                              unsigned int* enc) {
  hipLaunchKernelGGL(mass_range_init_kernel, dim3(1), dim3(64), 0, ctx->stream, enc);
  const int bi = (ni + kBlock - 1) / kBlock, bj = (nj + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(mass_range_kernel, dim3(bi < 256 ? bi : 256), dim3(kBlock), 0, ctx->stream, pi, ni, enc);
  hipLaunchKernelGGL(mass_range_kernel, dim3(bj < 256 ? bj : 256), dim3(kBlock), 0, ctx->stream, pj, nj, enc + 2);
}

template <int R, bool RECT, bool DET>
static void launch_sym(nbody_hip_ctx* ctx, dim3 grid, const float4* pi, int ni, const float4* pj, int nj,
                       int NB, int NBJ, int per, double* acci, void* accj, const unsigned int* enc,
                       float eps2) {
  const size_t S = (size_t)kBlock * R;
  // both instantiations are queued; the one whose mass assumption does not hold exits at once
  hipLaunchKernelGGL((direct_sym_kernel<R, RECT, true, DET>), grid, dim3(kBlock), 0, ctx->stream, pi, ni, pj, nj,
                     NB, NBJ, per, acci, accj, (size_t)NB * S, (size_t)NBJ * S, enc, eps2);
  hipLaunchKernelGGL((direct_sym_kernel<R, RECT, false, DET>), grid, dim3(kBlock), 0, ctx->stream, pi, ni, pj, nj,
                     NB, NBJ, per, acci, accj, (size_t)NB * S, (size_t)NBJ * S, enc, eps2);
}

// symmetric kernel worth it from here (measured, tools/sweep_sym_sizes.py): below, the one-sided kernel
bool symmetric_pays(const nbody_hip_ctx* ctx, size_t n) {
  if (ctx->tune_variant == 3) return true;
  return ctx->tune_variant < 0 && n >= 12288;
}

// bodies per lane R, measured (tools/sweep_sym_sizes.py, profiles/r01_sym_R_sweep.txt): the larger R,
// the fewer rotations per pair but the fewer workgroups; the two-set kernel has NBI x splits
// workgroups whatever R is, so it takes R = 16 much earlier than the half-ring all-pairs kernel
static int sym_R(const nbody_hip_ctx* ctx, size_t n, bool two_sets) {
  if (ctx->tune_tpl == 2 || ctx->tune_tpl == 4 || ctx->tune_tpl == 6 || ctx->tune_tpl == 8 ||
      ctx->tune_tpl == 16 || (ctx->tune_tpl == 12 && !two_sets))
    return ctx->tune_tpl;
  if (two_sets) return n >= 49152 ? 16 : (n >= 16384 ? 8 : 4);
  // (tools/sweep_sym_sizes.py, deterministic slots: 262,144 bodies 10.45 ms at 16 per lane against 10.89 at 8;
  // 131,072: 2.80 ms at 12 -- two workgroups per CU -- against 2.92 at 8)
  return n >= 200000 ? 16 : (n >= 100000 ? 12 : (n >= 28000 ? 8 : 6));  // (6: tools/sweep_mid.py, 12,288 bodies 0.064 vs 0.076 ms at 4)
}

// ---- launch shapes and the slot-plane budget: plain host arithmetic, also exported through
// ---- nbody_hip_direct_info so that callers (and the CPU tests) can see what a call will do
static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

SymShape sym_shape(const nbody_hip_ctx* ctx, size_t n, int R) {
  SymShape c;
  c.R = R;
  const int S = kBlock * R;
  c.NB = (int)((n + S - 1) / S);
  c.D = c.NB / 2;
  // workgroups = NB x splits; a split is a run of 64-body chunks of the flat (offset, chunk) list
  const int total = (c.D + 1) * (S / 64);
  int splits = ctx->tune_splits > 0 ? ctx->tune_splits : (kNumCU * kSymBlocksPerCU + c.NB - 1) / c.NB;
  if (splits < 1) splits = 1;
  int per = (total + splits - 1) / splits;
  if (per < kSymMinChunks) per = kSymMinChunks;
  if (per > S / 64) per = (per + S / 64 - 1) / (S / 64) * (S / 64);  // whole partners when there are enough
  c.per = per;
  c.splits = (total + per - 1) / per;
  c.plane = (size_t)c.NB * S;
  return c;
}
// slot planes of the deterministic all-pairs form: D reaction slots (3 float planes) then `splits` I-side slots
// (3 double planes), the second block 256-byte aligned
size_t SymShape::reaction_bytes() const { return align256((size_t)D * 3 * plane * sizeof(float)); }
size_t SymShape::det_bytes() const { return reaction_bytes() + (size_t)splits * 3 * plane * sizeof(double); }
size_t SymShape::atomic_bytes() const { return 3 * plane * sizeof(double); }

// May the deterministic form take `bytes` of slot planes?  They must fit the context's budget and, unless the
// context already holds them, a quarter of the memory that is free on the device right now (plus what the
// workspace would give back) -- on a shared or smaller GPU the atomic form is the better citizen.
static bool det_budget_allows(const nbody_hip_ctx* ctx, size_t bytes, size_t* free_now) {
  if (free_now) *free_now = 0;
  if (bytes > ctx->det_budget) return false;
  if (bytes <= ctx->partial.bytes) return true;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return true;  // cannot tell: try, the allocation failure path falls back
  }
  if (free_now) *free_now = free_b;
  return bytes <= (free_b + ctx->partial.bytes) / 4;
}

DirectPlan direct_plan(const nbody_hip_ctx* ctx, size_t n, bool query_device) {
  DirectPlan p;
  p.symmetric = symmetric_pays(ctx, n);
  p.eq = sym_shape(ctx, n, sym_R(ctx, n, false));
  p.gen = p.eq;
  p.det = false;
  p.bytes = 0;
  if (!p.symmetric) return p;
  if (ctx->deterministic) {
    // general masses at 16 bodies per lane: the slot stores push the kernel past the 256 architectural registers
    // (191 ms against 170 ms with atomics at N = 2^20); 12 bodies per lane (78 KiB of LDS = two workgroups per CU)
    // run it in 172 ms where 8 take 177 (tools/direct_det_probe.py, same box)
    SymShape gen = (kDetGenR != 16 && p.eq.R == 16 && ctx->tune_tpl == 0) ? sym_shape(ctx, n, kDetGenR) : p.eq;
    size_t need = p.eq.det_bytes() > gen.det_bytes() ? p.eq.det_bytes() : gen.det_bytes();
    bool ok = query_device ? det_budget_allows(ctx, need, nullptr) : need <= ctx->det_budget;
    if (!ok && gen.R != p.eq.R) {  // the general-mass shape has more slots: try with the common shape
      gen = p.eq;
      need = p.eq.det_bytes();
      ok = query_device ? det_budget_allows(ctx, need, nullptr) : need <= ctx->det_budget;
    }
    if (ok) {
      p.det = true;
      p.gen = gen;
      p.bytes = need;
    }
    p.det_bytes_wanted = need;
  }
  if (!p.det) p.bytes = p.eq.atomic_bytes();
  return p;
}

namespace {
template <int R, bool EQM, bool DET>
void launch_all_pairs(nbody_hip_ctx* ctx, const SymShape& c, const float4* posm, int n, double* acci, void* accj,
                      const unsigned int* enc, float eps2) {
  hipLaunchKernelGGL((direct_sym_kernel<R, false, EQM, DET>), dim3(c.NB, c.splits), dim3(kBlock), 0, ctx->stream, posm, n,
                     posm, n, c.NB, c.NB, c.per, acci, accj, c.plane, c.plane, enc, eps2);
}
template <bool EQM, bool DET>
void launch_all_pairs_R(nbody_hip_ctx* ctx, const SymShape& c, const float4* posm, int n, double* acci, void* accj,
                        const unsigned int* enc, float eps2) {
  switch (c.R) {
    case 2: launch_all_pairs<2, EQM, DET>(ctx, c, posm, n, acci, accj, enc, eps2); break;
    case 6: launch_all_pairs<6, EQM, DET>(ctx, c, posm, n, acci, accj, enc, eps2); break;
    case 8: launch_all_pairs<8, EQM, DET>(ctx, c, posm, n, acci, accj, enc, eps2); break;
    case 12: launch_all_pairs<12, EQM, DET>(ctx, c, posm, n, acci, accj, enc, eps2); break;
    case 16: launch_all_pairs<16, EQM, DET>(ctx, c, posm, n, acci, accj, enc, eps2); break;
    default: launch_all_pairs<4, EQM, DET>(ctx, c, posm, n, acci, accj, enc, eps2); break;
  }
}
}  // namespace

int direct_symmetric(nbody_hip_ctx* ctx, const float4* posm, size_t n, float G, float eps2,
                     float4* acc4, int accumulate, float* ax, float* ay, float* az, float* vx,
                     float* vy, float* vz, const float* aox, const float* aoy, const float* aoz,
                     float half_dt) {
  // Both instantiations (equal masses / general masses) are queued; the one whose assumption about the
  // masses fails exits at once (decided on the device).  Each has a launch shape of its own.
  // Deterministic mode: one slot per contribution; every slot is written exactly once by the kernel, so
  // nothing has to be cleared.  Taken when asked for (nbody_hip_direct_deterministic) and the slot planes fit
  // the budget (det_budget_allows); "required" (mode 2) turns a miss into an error instead of the atomic form.
  DirectPlan plan = direct_plan(ctx, n, true);
  if (ctx->deterministic == 2 && !plan.det)
    return NBH_FAIL(NBODY_HIP_ERR_RESOURCE, "deterministic Direct sums were required but their slot planes (%zu bytes "
                    "for %zu bodies) exceed the budget (%zu bytes and a quarter of the free device memory)",
                    plan.det_bytes_wanted, n, ctx->det_budget);
  // a workspace left over from a much larger problem is given back first (the slot planes are the one big
  // allocation of this library)
  if (ctx->partial.bytes > ((size_t)64 << 20) && ctx->partial.bytes / 4 > plan.bytes && !ctx->capturing) {
    NBH_HIP(hipDeviceSynchronize());
    ctx->partial.release();
  }
  int rc = ctx->partial.reserve(plan.bytes);
  if (rc != NBODY_HIP_OK && plan.det && ctx->deterministic != 2) {
    // the slot planes could not be had after all: the atomic form needs 24 bytes per body
    (void)hipGetLastError();
    plan.det = false;
    plan.gen = plan.eq;
    plan.bytes = plan.eq.atomic_bytes();
    rc = ctx->partial.reserve(plan.bytes);
  }
  if (rc != NBODY_HIP_OK) return rc;
  const SymShape &eq = plan.eq, &gen = plan.gen;
  const bool det = plan.det;
  char* base = static_cast<char*>(ctx->partial.ptr);
  if (!det) NBH_HIP(hipMemsetAsync(base, 0, plan.bytes, ctx->stream));
  if (int rc2 = ctx->reduce.reserve(64)) return rc2;
  unsigned int* enc = static_cast<unsigned int*>(ctx->reduce.ptr);
  const int ni = (int)n;
  launch_mass_range(ctx, posm, ni, posm, ni, enc);
  SlotLayout Leq, Lgen;
  if (det) {
    // [reaction slots: D x 3 float planes][I-side slots: splits x 3 double planes], one layout per instantiation
    // (only one of the two runs; they share the workspace)
    Leq = SlotLayout{reinterpret_cast<const double*>(base + eq.reaction_bytes()), reinterpret_cast<const float*>(base),
                     eq.plane, eq.splits, eq.D};
    Lgen = SlotLayout{reinterpret_cast<const double*>(base + gen.reaction_bytes()), reinterpret_cast<const float*>(base),
                      gen.plane, gen.splits, gen.D};
    launch_all_pairs_R<true, true>(ctx, eq, posm, ni, const_cast<double*>(Leq.isl), base, enc, eps2);
    launch_all_pairs_R<false, true>(ctx, gen, posm, ni, const_cast<double*>(Lgen.isl), base, enc, eps2);
  } else {
    Leq = Lgen = SlotLayout{reinterpret_cast<const double*>(base), nullptr, eq.plane, 1, 0};
    launch_all_pairs_R<true, false>(ctx, eq, posm, ni, reinterpret_cast<double*>(base), base, enc, eps2);
    launch_all_pairs_R<false, false>(ctx, gen, posm, ni, reinterpret_cast<double*>(base), base, enc, eps2);
  }
  NBH_LAUNCH_CHECK();
  ctx->last_direct_kernel = det ? 2 : 1;
  const int fblocks = (int)((n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(direct_sym_finalize_kernel, dim3(fblocks), dim3(kBlock), 0, ctx->stream, Leq, Lgen, det ? 1 : 0, enc,
                     ni, G, acc4, accumulate, ax, ay, az, vx, vy, vz, aox, aoy, aoz, half_dt);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

// Two disjoint sets: acc_i (+)= forces on I from J ; acc_j = forces on J from I (the reactions).
// Deterministic form (same switch as the all-pairs kernel): `splits` fp64 I-side slots per I body, NBI fp32
// reaction slots per J body -- a sharded run is then reproducible launch after launch too.
int direct_symmetric_pair(nbody_hip_ctx* ctx, const float4* pi, size_t ni, const float4* pj, size_t nj,
                          float G, float eps2, float4* acc_i, int accumulate_i, float4* acc_j,
                          int accumulate_j) {
  // launch shape; the deterministic form has its own: 8 bodies per lane and half as many workgroups per CU -- every
  // split owns an fp64 I-side slot per body (24 B x splits x n_i written and read back), so fewer, longer splits pay:
  // 131,072 x 131,072 bodies: 5.04 ms against 5.47 at 16 bodies per lane / 128 splits (atomics: 4.95;
  // tools/pair_det_probe.py)
  int R = 0, NBI = 0, NBJ = 0, per = 0, splits = 0;
  size_t plane_i = 0, plane_j = 0, det_i = 0, det_j = 0;
  auto shape = [&](bool det_form) {
    const size_t nmin = ni < nj ? ni : nj;
    R = sym_R(ctx, nmin, true);
    if (det_form && ctx->tune_tpl == 0 && R > 8) R = 8;
    const int S = kBlock * R;
    NBI = (int)((ni + S - 1) / S);
    NBJ = (int)((nj + S - 1) / S);
    const int total = NBJ * (S / 64);
    const int per_cu = det_form ? kSymBlocksPerCU / 2 : kSymBlocksPerCU;
    splits = ctx->tune_splits > 0 ? ctx->tune_splits : (kNumCU * per_cu + NBI - 1) / NBI;
    if (splits < 1) splits = 1;
    per = (total + splits - 1) / splits;
    if (per < kSymMinChunks) per = kSymMinChunks;
    if (per > S / 64) per = (per + S / 64 - 1) / (S / 64) * (S / 64);
    splits = (total + per - 1) / per;
    plane_i = (size_t)NBI * S;
    plane_j = (size_t)NBJ * S;
    det_i = align256((size_t)splits * 3 * plane_i * sizeof(double));
    det_j = (size_t)NBI * 3 * plane_j * sizeof(float);
  };
  shape(ctx->deterministic != 0);
  bool det = ctx->deterministic != 0 && det_budget_allows(ctx, det_i + det_j, nullptr);
  if (ctx->deterministic == 2 && !det)
    return NBH_FAIL(NBODY_HIP_ERR_RESOURCE, "deterministic Direct sums were required but their slot planes (%zu bytes) "
                    "exceed the budget", det_i + det_j);
  if (!det) shape(false);
  size_t bi = align256(plane_i * 3 * sizeof(double)), bj = plane_j * 3 * sizeof(double);
  int rc = ctx->partial.reserve(det ? det_i + det_j : bi + bj);
  if (rc != NBODY_HIP_OK && det && ctx->deterministic != 2) {
    (void)hipGetLastError();
    det = false;
    shape(false);
    bi = align256(plane_i * 3 * sizeof(double));
    bj = plane_j * 3 * sizeof(double);
    rc = ctx->partial.reserve(bi + bj);
  }
  if (rc != NBODY_HIP_OK) return rc;
  char* base = static_cast<char*>(ctx->partial.ptr);
  double* acci = reinterpret_cast<double*>(base);
  void* accj = base + (det ? det_i : bi);
  if (!det) NBH_HIP(hipMemsetAsync(base, 0, bi + bj, ctx->stream));
  if (int rc2 = ctx->reduce.reserve(64)) return rc2;
  unsigned int* enc = static_cast<unsigned int*>(ctx->reduce.ptr);
  launch_mass_range(ctx, pi, (int)ni, pj, (int)nj, enc);
  const dim3 grid(NBI, splits);
#define NBH_PAIR_LAUNCH(RR)                                                                                          \
  if (det) launch_sym<RR, true, true>(ctx, grid, pi, (int)ni, pj, (int)nj, NBI, NBJ, per, acci, accj, enc, eps2);    \
  else launch_sym<RR, true, false>(ctx, grid, pi, (int)ni, pj, (int)nj, NBI, NBJ, per, acci, accj, enc, eps2)
  switch (R) {
    case 2: NBH_PAIR_LAUNCH(2); break;
    case 6: NBH_PAIR_LAUNCH(6); break;
    case 8: NBH_PAIR_LAUNCH(8); break;
    case 16: NBH_PAIR_LAUNCH(16); break;
    default: NBH_PAIR_LAUNCH(4); break;
  }
#undef NBH_PAIR_LAUNCH
  NBH_LAUNCH_CHECK();
  ctx->last_direct_kernel = det ? 2 : 1;
  SlotLayout Li, Lj;
  if (det) {
    Li = SlotLayout{acci, nullptr, plane_i, splits, 0};
    Lj = SlotLayout{nullptr, static_cast<const float*>(accj), plane_j, 0, NBI};
  } else {
    Li = SlotLayout{acci, nullptr, plane_i, 1, 0};
    Lj = SlotLayout{static_cast<const double*>(accj), nullptr, plane_j, 1, 0};
  }
  // (mass_range is read for the equal-mass factor only in the deterministic form; the atomic form applied it)
  hipLaunchKernelGGL(direct_sym_finalize_kernel, dim3((unsigned)((ni + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     ctx->stream, Li, Li, det ? 1 : 0, det ? enc : nullptr, (int)ni, G, acc_i, accumulate_i, nullptr, nullptr,
                     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f);
  hipLaunchKernelGGL(direct_sym_finalize_kernel, dim3((unsigned)((nj + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     ctx->stream, Lj, Lj, det ? 1 : 0, det ? enc : nullptr, (int)nj, G, acc_j, accumulate_j, nullptr, nullptr,
                     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

}  // namespace nbh
