// spatial_hash.hip -- short-range forces on a uniform cell grid, gfx950.
//
// Replaces SpatialHashGrid / SpatialHashCalculator of the reference
// (src/cuda/force_spatial_hash.cu:14-377).  Same semantics:
//   * grid = bounding box padded by 0.001, dims = ceil(extent/cell)+1 (:225-231, :244-246),
//     > 1e8 cells is an error (:252-254); cell id = x + y gx + z gx gy of floor((p-min)/cell),
//     clamped (:36-48);
//   * a body interacts with the bodies of its own and the 26 adjacent cells that exist
//     (non-periodic, :104-113) whose UNSOFTENED distance^2 is below cutoff^2 (:131-135);
//     softening is added after the test; the body itself is skipped (:124).
// Different machine mapping:
//   * binning is a stable LSD radix sort of (cell id, body index) (rocPRIM, AMD's native device
//     primitives) instead of atomic count / Thrust scan / atomic scatter (:52-80), so the
//     within-cell order is the body-index order and results are bitwise reproducible (the
//     reference's scatter order is non-deterministic);
//   * bodies are physically reordered into float4 {x,y,z,m} in cell order, so every cell -- and
//     every run of cells along x -- is one contiguous, coalesced range (the reference gathers
//     pos[sorted_indices[k]] per pair, :123-129);
//   * force kernel: one workgroup per RUN of W cells along x.  Its targets are one contiguous
//     range; its sources are the 9 rows (y+-1, z+-1) x cells [x0-1, x0+W], i.e. NINE contiguous
//     ranges, streamed through LDS tiles and broadcast to the waves exactly like the Direct
//     kernel.  Range ends come from binary searches in the sorted key array (no per-cell
//     start/end arrays on the force path).  When cutoff > cell_size the reference's 27-cell
//     search misses pairs; the STRICT variant reproduces that by testing |cx_j - cx_i| <= 1.
//
// Roofline: the build is HBM/sort bound (~48 B/body), the force kernel is VALU bound at
// ~9 (W+2)/W rho candidate pairs per body (rho = bodies per cell), fed from L2/LDS.

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/zip_iterator.hpp>

#include "common.h"
#include "onesweep.h"
#include "radix_sort.h"

// Binning = ONE stable sort by cell id that carries the bodies along: the values are (float4 body, original
// index) pairs -- a zip of the packed bodies with a counting iterator on the way in, of the cell-ordered body array
// and the index array on the way out -- so the separate gather pass of round 2 (87 us at 4.2 M bodies: a random
// 16-byte read per body) is gone, and with 10-bit digits the 19 cell-id bits of BASELINE config 5 are two Onesweep
// passes instead of three.
#ifndef NBH_HASH_RADIX_BITS
#define NBH_HASH_RADIX_BITS 10
#endif
#ifndef NBH_HASH_SPLIT_FILTER
#define NBH_HASH_SPLIT_FILTER false  // split form: the crowded cells take the unfiltered two-targets form (the filtered one measured slower there: 1.90 / 1.70 against 1.82 / 1.61 ms at 2,000 / 4,000 steps)
#endif
#ifndef NBH_HASH_SPLIT_CNT
#define NBH_HASH_SPLIT_CNT 6
#endif
#ifndef NBH_HASH_OWN_SORT
#define NBH_HASH_OWN_SORT 1   // 0: rocprim::radix_sort_pairs everywhere (A/B builds)
#endif
using SortConfig = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<1024, 8>,
                                        NBH_HASH_RADIX_BITS, rocprim::block_radix_rank_algorithm::match>,
    nbh::kSortMergeLimit>;

namespace nbh {

constexpr int HTS = 256;  // sources per LDS tile
typedef float f2 __attribute__((ext_vector_type(2)));

// |d|^2 as nvcc's default contraction forms the reference's `dx*dx + dy*dy + dz*dz`
// (force_spatial_hash.cu:131): one product, two fused multiply-adds.  The cutoff decision is taken on
// THIS value, in the kernels and in the oracle alike, so both sides take bit-identical decisions.
__device__ __forceinline__ float hash_dist2(float dx, float dy, float dz) {
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}
__device__ __forceinline__ f2 hash_dist2(f2 dx, f2 dy, f2 dz) {
  return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
}


__global__ void bbox_init_kernel(unsigned int* enc) {
  if (threadIdx.x < 3) enc[threadIdx.x] = 0xffffffffu;      // mins
  else if (threadIdx.x < 6) enc[threadIdx.x] = 0u;           // maxs
}

// per-thread min/max -> wave64 shuffle reduce -> LDS across the block's 4 waves -> 6 atomics per block
template <int WAVES = 4>
__device__ __forceinline__ void block_bbox_commit(float (&lo)[3], float (&hi)[3], float (&red)[WAVES][6],
                                                  unsigned int* __restrict__ enc) {
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64));
      hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64));
    }
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) { red[w][a] = lo[a]; red[w][3 + a] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    float v = red[0][a];
    for (int k = 1; k < WAVES; k++) v = a < 3 ? fminf(v, red[k][a]) : fmaxf(v, red[k][a]);
    if (a < 3) atomicMin(&enc[a], float_to_ordered(v));
    else atomicMax(&enc[a], float_to_ordered(v));
  }
}

// min/max of x,y,z: wave64 shuffle reduce -> LDS across the block's 4 waves -> 6 atomics per
// BLOCK on order-preserving integers, from at most 256 blocks (1,536 atomics in all: every wave
// hitting the same six words serialises at ~11 ns each, measured 282 us with 24,576 of them).
// The reference CAS-loops float atomics from every block (force_barnes_hut.cu:41-110).
__global__ __launch_bounds__(kBlock) void bbox_kernel(const float4* __restrict__ posm, int n,
                                                      unsigned int* __restrict__ enc) {
  __shared__ float red[4][6];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 p = posm[i];
    lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
    lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
    lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
  }
  block_bbox_commit(lo, hi, red, enc);
}

// the float4 drift (drift_packed_kernel, integrator.hip: same arithmetic) with the box of the NEW positions in the
// same pass: the sharded spatial hash's first kernel of a step
__global__ __launch_bounds__(kBlock) void drift_bbox_packed_kernel(float4* __restrict__ posm, const float4* __restrict__ vel,
                                                                   const float4* __restrict__ acc, int n, float dt,
                                                                   unsigned int* __restrict__ enc) {
  __shared__ float red[4][6];
  const float dt2_half = 0.5f * dt * dt;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    float4 p = posm[i];
    const float4 v = vel[i], a = acc[i];
    p.x = drift1(p.x, v.x, a.x, dt, dt2_half);
    p.y = drift1(p.y, v.y, a.y, dt, dt2_half);
    p.z = drift1(p.z, v.z, a.z, dt, dt2_half);
    posm[i] = p;
    lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
    lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
    lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
  }
  block_bbox_commit(lo, hi, red, enc);
}

// SoA -> float4 packing and the bounding box in ONE pass (the build's first two kernels fused):
// <= 256 workgroups stride over the bodies, write posm and keep the running min/max
__global__ __launch_bounds__(kBlock) void pack_bbox_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ z, const float* __restrict__ m,
                                                           int n, float4* __restrict__ posm,
                                                           unsigned int* __restrict__ enc) {
  __shared__ float red[4][6];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 p = make_float4(x[i], y[i], z[i], m[i]);
    posm[i] = p;
    lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
    lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
    lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
  }
  block_bbox_commit(lo, hi, red, enc);
}

// the same pass with the drift of the step in front: a_old <- a ; x += v dt + a dt^2/2 (integrator.cu:44-46,
// :16-19, arithmetic of drift_pack_kernel) ; posm <- {x, y, z, m} ; running min / max of the NEW positions
// (BLOCK: the grid stays at <= 256 workgroups for the sake of the six atomics per workgroup -- see bbox_kernel -- so the
// bytes in flight come from the workgroup size: 1,024 threads = four waves per SIMD instead of one; round 4)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void drift_pack_bbox_kernel(float* __restrict__ x, float* __restrict__ y,
                                                                 float* __restrict__ z, const float* __restrict__ vx,
                                                                 const float* __restrict__ vy, const float* __restrict__ vz,
                                                                 const float* __restrict__ ax, const float* __restrict__ ay,
                                                                 const float* __restrict__ az, float* __restrict__ aox,
                                                                 float* __restrict__ aoy, float* __restrict__ aoz,
                                                                 const float* __restrict__ m, int n, float dt,
                                                                 float4* __restrict__ posm, unsigned int* __restrict__ enc) {
  __shared__ float red[BLOCK / 64][6];
  const float dt2_half = 0.5f * dt * dt;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
    const float a0 = ax[i], a1 = ay[i], a2 = az[i];
    aox[i] = a0; aoy[i] = a1; aoz[i] = a2;
    const float4 p = make_float4(drift1(x[i], vx[i], a0, dt, dt2_half), drift1(y[i], vy[i], a1, dt, dt2_half),
                                 drift1(z[i], vz[i], a2, dt, dt2_half), m[i]);
    x[i] = p.x; y[i] = p.y; z[i] = p.z;
    posm[i] = p;
    lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
    lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
    lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
  }
  block_bbox_commit<BLOCK / 64>(lo, hi, red, enc);
}

int launch_drift_pack_bbox(nbody_hip_ctx* ctx, nbody_particle_data* d, float dt, float4* posm, unsigned int* enc,
                           bool init) {
  const int n = (int)d->count;
  const int blocks = (n + kBlock - 1) / kBlock;
  if (init) hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, ctx->stream, enc);
  if (n >= 256 * 1024) {
    hipLaunchKernelGGL(drift_pack_bbox_kernel<1024>, dim3(256), dim3(1024), 0, ctx->stream, d->pos_x,
                       d->pos_y, d->pos_z, d->vel_x, d->vel_y, d->vel_z, d->acc_x, d->acc_y, d->acc_z, d->acc_old_x,
                       d->acc_old_y, d->acc_old_z, d->mass, n, dt, posm, enc);
  } else {
    hipLaunchKernelGGL(drift_pack_bbox_kernel<kBlock>, dim3(blocks < 256 ? blocks : 256), dim3(kBlock), 0, ctx->stream, d->pos_x,
                       d->pos_y, d->pos_z, d->vel_x, d->vel_y, d->vel_z, d->acc_x, d->acc_y, d->acc_z, d->acc_old_x,
                       d->acc_old_y, d->acc_old_z, d->mass, n, dt, posm, enc);
  }
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

int launch_bbox_init(nbody_hip_ctx* ctx, unsigned int* enc) {
  hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, ctx->stream, enc);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

// init = false: enc already holds the empty box (the caller re-armed it, see tree_build_packed)
int launch_pack_bbox(nbody_hip_ctx* ctx, const float* x, const float* y, const float* z, const float* m, int n,
                     float4* posm, unsigned int* enc, bool init) {
  const int blocks = (n + kBlock - 1) / kBlock;
  if (init) hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, ctx->stream, enc);
  hipLaunchKernelGGL(pack_bbox_kernel, dim3(blocks < 256 ? blocks : 256), dim3(kBlock), 0, ctx->stream, x, y, z, m,
                     n, posm, enc);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

int launch_bbox(nbody_hip_ctx* ctx, const float4* posm, int n, unsigned int* enc, bool init) {
  const int blocks = (n + kBlock - 1) / kBlock;
  if (init) hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, ctx->stream, enc);
  hipLaunchKernelGGL(bbox_kernel, dim3(blocks < 256 ? blocks : 256), dim3(kBlock), 0, ctx->stream,
                     posm, n, enc);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

struct GridInfo {
  float bmin[3];
  float bmax[3];
  int dims[3];
  int pad;
  long long total;
};

// cells per axis (force_spatial_hash.cu:244-246) with the int conversion guarded: NaN / inf / huge
// extents become 2^30 cells, which the "grid too large" check then rejects
__host__ __device__ inline int grid_axis_cells(float lo, float hi, float cell) {
  const float cells = ceilf((hi - lo) / cell);
  return (cells < 1.0e9f && cells >= 0.0f) ? (int)cells + 1 : 0x40000000;
}
// running product of the axis sizes, SATURATING at 2^50 (three axes of 2^30 would wrap a 64-bit
// product to 0 and slip under the 1e8 limit)
__host__ __device__ inline long long grid_cells_times(long long total, int d) {
  const long long cap = 0x4000000000000LL;
  return (d <= 0 || total > cap / d) ? cap : total * d;
}

// decode, pad by `pad` (0.001 for the hash grid, force_spatial_hash.cu:225-231), size the grid
// host_info: the same record in pinned host memory (mapped into the device's address space): the build's one host
// round trip then needs only the stream synchronisation, not a device-to-host copy on top (a 40-byte
// hipMemcpyAsync ran as a 27 us copy kernel)
// (both grid_info kernels also zero the sort's digit counts, which assign_cells_kernel then accumulates: kHistWords)
constexpr int kHistPlaces = 3;  // grids hold < 1e8 cells = 27 key bits = three 10-bit digits
constexpr int kHistWords = kHistPlaces << NBH_HASH_RADIX_BITS;
constexpr int kHistCopies = onesweep::DigitHistogram<NBH_HASH_RADIX_BITS>::kCopies;
constexpr int kHistThreads = NBH_HIST_THREADS;  // workgroup of the key kernel (few, large workgroups: fewer flushes)
// ... and leave the empty box behind in enc: the next build's box pass needs no bbox_init_kernel
// seq: written into the record's `pad` word LAST (after a system-scope fence): the host polls that word in the mapped
// copy instead of waiting for the stream (see grid_build_packed)
__global__ void grid_info_kernel(unsigned int* __restrict__ enc, float cell, float pad,
                                 GridInfo* __restrict__ info, GridInfo* __restrict__ host_info,
                                 unsigned int* __restrict__ hist, int seq) {
  for (int t = threadIdx.x; t < kHistWords * kHistCopies; t += blockDim.x) hist[t] = 0u;
  if (threadIdx.x != 0) return;
  GridInfo gi;
  long long total = 1;
  for (int a = 0; a < 3; a++) {
    const float lo = ordered_to_float(enc[a]) - pad;
    const float hi = ordered_to_float(enc[3 + a]) + pad;
    enc[a] = 0xffffffffu;
    enc[3 + a] = 0u;
    gi.bmin[a] = lo;
    gi.bmax[a] = hi;
    const int d = grid_axis_cells(lo, hi, cell);
    gi.dims[a] = d;
    total = grid_cells_times(total, d);
  }
  gi.total = total;
  gi.pad = seq;
  *info = gi;
  if (host_info) {
    gi.pad = 0;
    *host_info = gi;
    __threadfence_system();
    __atomic_store_n(&host_info->pad, seq, __ATOMIC_RELEASE);
    __threadfence_system();
  }
}

// grid geometry decided on the host (explicit bounds: the sharded path): the record travels as a kernel argument,
// so there is no pinned staging buffer to protect and the build needs no stream synchronisation
__global__ void grid_info_set_kernel(GridInfo gi, GridInfo* __restrict__ info, unsigned int* __restrict__ hist) {
  for (int t = threadIdx.x; t < kHistWords * kHistCopies; t += blockDim.x) hist[t] = 0u;
  if (threadIdx.x == 0) *info = gi;
}

// (mis-speculated key pass, see grid_build_packed: the digit counts start again)
__global__ void hist_zero_kernel(unsigned int* __restrict__ hist) {
  for (int t = threadIdx.x; t < kHistWords * kHistCopies; t += blockDim.x) hist[t] = 0u;
}

__device__ __forceinline__ int cell_coord(float p, float lo, float cell, int dim) {
  int c = (int)floorf((p - lo) / cell);
  return min(max(c, 0), dim - 1);
}

// force_spatial_hash.cu:28-49.  The kernel that has every key in a register also does the sort's bookkeeping: it
// zeroes the look-back block of the Onesweep passes (zero_words) and, with hist_places > 0, accumulates the digit
// histograms (LDS, then one global atomic per non-empty bin and workgroup) -- a fill launch and rocPRIM's histogram
// kernel (43 us at 4.2 M bodies) less.  A fixed grid strides over the bodies.
__global__ __launch_bounds__(kHistThreads) void assign_cells_kernel(const float4* __restrict__ posm, int n,
                                                              const GridInfo* __restrict__ info,
                                                              float cell,
                                                              unsigned int* __restrict__ keys,
                                                              unsigned int* __restrict__ zero, unsigned int zero_words,
                                                              unsigned int* __restrict__ hist, int hist_places) {
  using Hist = onesweep::DigitHistogram<NBH_HASH_RADIX_BITS>;
  __shared__ unsigned int h[kHistWords];
  const int stride = gridDim.x * kHistThreads;
  for (unsigned int w = blockIdx.x * kHistThreads + threadIdx.x; w < zero_words; w += stride) zero[w] = 0u;
  if (hist_places) {
    Hist::zero(h, hist_places, threadIdx.x, kHistThreads);
    __syncthreads();
  }
  const int gx = info->dims[0], gy = info->dims[1], gz = info->dims[2];
  const float lx = info->bmin[0], ly = info->bmin[1], lz = info->bmin[2];
  for (int i = blockIdx.x * kHistThreads + threadIdx.x; i < n; i += stride) {
    const float4 p = posm[i];
    const int cx = cell_coord(p.x, lx, cell, gx);
    const int cy = cell_coord(p.y, ly, cell, gy);
    const int cz = cell_coord(p.z, lz, cell, gz);
    const unsigned int key = (unsigned int)(cx + cy * gx + cz * gx * gy);
    keys[i] = key;
    if (hist_places) Hist::add(h, key, hist_places);
  }
  if (hist_places) {
    __syncthreads();
    Hist::flush(h, hist_places, threadIdx.x, kHistThreads, hist, kHistWords);
  }
}

// Per-cell start array: cell_lb[c - base] = number of bodies whose cell id is below c (= sorted position of the
// first body of cell c) for the cells [base, base + count] -- the whole grid, or the z-slab of a rank.  Only built
// for grids that are not much larger than the body list; it replaces the binary searches of the force kernel by
// direct reads.  One thread per CELL: a search that starts at the position a uniform density would put the
// cell at and gallops from there (a handful of probes for near-uniform bodies, log n for any).  The first
// version had the thread at each sorted position fill the cells between its predecessor's cell and its own:
// one thread then wrote the whole empty top layer of the grid (dims = ceil(extent / cell) + 1), 4,400 dependent
// stores = 74 us at 4.2 M bodies.
__global__ __launch_bounds__(kBlock) void cell_lb_kernel(const unsigned int* __restrict__ keys, int n, int base,
                                                         int count, int* __restrict__ cell_lb) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t > count) return;
  const unsigned int c = (unsigned int)(base + t);  // lower bound of c in keys[0, n)
  int guess = (int)((long long)t * n / (count > 0 ? count : 1));
  guess = min(max(guess, 0), n);
  int lo, hi;
  if (guess < n && keys[guess] < c) {  // answer to the right of the guess
    lo = guess + 1;
    int step = 16;
    while (lo + step <= n && keys[lo + step - 1] < c) { lo += step; step <<= 1; }
    hi = min(n, lo + step);
  } else {                             // answer at or to the left of the guess
    hi = guess;
    int step = 16;
    while (hi - step >= 0 && keys[hi - step] >= c) { hi -= step; step <<= 1; }
    lo = max(0, hi - step);
  }
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < c) lo = mid + 1; else hi = mid;
  }
  cell_lb[t] = lo;
}

// The same array from the other side (round 4; grids with at most two cells per body): ONE THREAD PER SORTED POSITION.
// Position k owns the cells (keys[k - 1], keys[k]] -- their lower bound is k -- and position n the cells behind the last
// key: most threads find their predecessor in the same cell and leave, the first body of a cell stores one word, a short
// run of empty cells is filled by that thread, and a long one (64 cells or more: the empty top layers of a grid whose
// dimensions are ceil(extent / cell) + 1 are thousands) goes to a list that cell_gap_kernel fills with a wave per
// run.  Two coalesced key reads per body instead of ~19 dependent probes per cell (the uniform-density guess above is off
// by the square root of the position -- up to 2,000 bodies -- so it gallops eight times and then bisects): 8 + 3 us
// against 28 us at 4.2 M bodies, the same values.  The list's counters alternate between builds: the fill kernel of one
// build zeroes the counter of the next.
constexpr int kGapInline = 64;
__global__ __launch_bounds__(kBlock) void cell_mark_kernel(const unsigned int* __restrict__ keys, int n, int base, int count,
                                                           int* __restrict__ cell_lb, int* __restrict__ gap_count,
                                                           int* __restrict__ gaps) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k > n) return;
  const long long first = base, last = (long long)base + count;                    // the table's cells [first, last]
  const long long prev = k == 0 ? first - 1 : (long long)keys[k - 1];
  const long long cur = k == n ? last : (long long)keys[k];
  if (cur == prev) return;
  const long long lo = prev + 1 > first ? prev + 1 : first, hi = cur < last ? cur : last;
  if (lo > hi) return;
  if (hi - lo < kGapInline) {
    for (long long c = lo; c <= hi; c++) cell_lb[c - first] = k;
  } else {
    const int slot = atomicAdd(gap_count, 1);  // at most (count + 1) / kGapInline + 1 runs this long fit in the table
    gaps[3 * slot] = (int)(lo - first);
    gaps[3 * slot + 1] = (int)(hi - first);
    gaps[3 * slot + 2] = k;
  }
}
__global__ __launch_bounds__(kBlock) void cell_gap_kernel(const int* __restrict__ gap_count, const int* __restrict__ gaps,
                                                          int* __restrict__ cell_lb, int* __restrict__ next_count) {
  const int ng = *gap_count;
  // a wave per run (a clumped box has tens of thousands of runs of a few hundred empty cells)
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int g = blockIdx.x * (kBlock / 64) + w; g < ng; g += gridDim.x * (kBlock / 64)) {
    const int lo = gaps[3 * g], hi = gaps[3 * g + 1], k = gaps[3 * g + 2];
    for (int c = lo + lane; c <= hi; c += 64) cell_lb[c] = k;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *next_count = 0;
}

// The same array for grids with more cells than bodies, where the bodies are anything but evenly spread (a box that
// has expanded and clumped: 22 M cells for 4.2 M bodies, cells of 400 beside a majority of empty ones) and the
// uniform-density guess above is off by 10^5 positions: ~30 dependent probes per cell, 0.42 ms.  Two levels instead:
// every 64th cell by a full binary search, then every cell inside the bracket its two neighbours of the coarse level
// give (the bodies of 64 cells: a handful of probes).
constexpr int kLbCoarse = 64;
__global__ __launch_bounds__(kBlock) void cell_lb_coarse_kernel(const unsigned int* __restrict__ keys, int n, int base,
                                                                int count, int* __restrict__ coarse) {
  const int t = blockIdx.x * kBlock + threadIdx.x;  // cell base + t * kLbCoarse
  const int ncoarse = count / kLbCoarse + 2;  // entries floor(t / 64) and + 1 for every t in [0, count]
  if (t >= ncoarse) return;
  const long long c = (long long)base + (long long)t * kLbCoarse;
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((long long)keys[mid] < c) lo = mid + 1; else hi = mid;
  }
  coarse[t] = lo;
}
__global__ __launch_bounds__(kBlock) void cell_lb_fine_kernel(const unsigned int* __restrict__ keys, int base, int count,
                                                              const int* __restrict__ coarse, int* __restrict__ cell_lb) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t > count) return;
  const unsigned int c = (unsigned int)(base + t);
  int lo = coarse[t / kLbCoarse], hi = coarse[t / kLbCoarse + 1];  // keys[lo - 1] < first cell of the bracket <= c
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < c) lo = mid + 1; else hi = mid;
  }
  cell_lb[t] = lo;
}

// per-cell [start, end) for the inspection API (copyCellDataToHost); empty cells stay 0/0
__global__ __launch_bounds__(kBlock) void cell_ranges_kernel(const unsigned int* __restrict__ keys,
                                                             int n, int* __restrict__ cell_start,
                                                             int* __restrict__ cell_end) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const unsigned int c = keys[k];
  if (k == 0 || keys[k - 1] != c) cell_start[c] = k;
  if (k == n - 1 || keys[k + 1] != c) cell_end[c] = k + 1;
}

__device__ __forceinline__ int lower_bound_keys(const unsigned int* __restrict__ keys, int n,
                                                unsigned int v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ---------------------------------------------------------------------------------------
// Force kernel.  grid = (ceil(gx / (4 Wv)), gy, gz); block = 256 = 4 waves.
// Wave w owns the run of Wv cells [x0 + w Wv, +Wv) of row (y, z): its targets are one contiguous
// range (about 64 bodies: Wv = 64 / mean occupancy).  The block streams, row by row (9 rows:
// y+-1, z+-1), the sources of cells [x0-1, x0+4Wv] through LDS tiles; each wave only visits the
// part of a tile that lies in ITS cells [xw-1, xw+Wv], so a target tests 9 (Wv+2) rho candidates
// (2x the 27-cell minimum at Wv = 4) with all lanes busy.  All range ends come from one batch of
// binary searches in the sorted key array (9 rows x (4Wv+3) cell boundaries, one per lane).
// GUARD : eps2 so small that m*rsq(eps2)^3 may overflow -> d2 > 0 tested explicitly
// STRICT: cutoff > cell_size -> only x-adjacent cells interact (the reference's 27-cell search)
// ---------------------------------------------------------------------------------------
constexpr int kMaxWv = 16;
constexpr int kMaxBW = 4 * kMaxWv;

template <bool GUARD, bool STRICT>
__global__ __launch_bounds__(kBlock) void hash_force_kernel(
    const float4* __restrict__ sorted, const unsigned int* __restrict__ keys,
    const int* __restrict__ idx, int n, const GridInfo* __restrict__ info, int Wv, float cutoff2,
    float eps2, float G, float* __restrict__ acc_x, float* __restrict__ acc_y,
    float* __restrict__ acc_z, float4* __restrict__ acc4) {
  __shared__ int rowpos[9][kMaxBW + 3];  // sorted position of the first body of cell x0-1+j
  __shared__ int tpos[5];                // target range boundaries of the 4 waves
  __shared__ float4 tile[2][HTS];
  __shared__ int tile_cx[2][HTS];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int gx = info->dims[0], gy = info->dims[1], gz = info->dims[2];
  const int BW = 4 * Wv;
  const int x0 = blockIdx.x * BW, y = blockIdx.y, z = blockIdx.z;

  // cell boundaries: thread (r, j) -> lower_bound of cell (x0-1+j) in row r
  for (int q = tid; q < 9 * (BW + 3); q += kBlock) {
    const int r = q / (BW + 3), j = q - r * (BW + 3);
    const int yy = y + (r % 3) - 1, zz = z + (r / 3) - 1;
    int val = 0;
    if (yy >= 0 && yy < gy && zz >= 0 && zz < gz) {
      const unsigned int base = (unsigned int)((zz * gy + yy) * gx);
      const int cx = min(max(x0 - 1 + j, 0), gx);  // clamp to the row: [0, gx]
      val = lower_bound_keys(keys, n, base + (unsigned int)cx);
    }
    rowpos[r][j] = val;
  }
  __syncthreads();
  // own row is r = 4 (dy = dz = 0); wave w's targets start at cell x0 + w Wv = boundary 1 + w Wv
  if (tid < 5) tpos[tid] = rowpos[4][min(1 + tid * Wv, BW + 2)];
  __syncthreads();
  const int bt0 = tpos[0], bt1 = tpos[4];
  if (bt0 >= bt1) return;  // no targets in this block (uniform)
  const int t0 = tpos[w], t1 = tpos[w + 1];
  int maxcnt = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) maxcnt = max(maxcnt, tpos[k + 1] - tpos[k]);
  const int nchunks = (maxcnt + 63) / 64;  // uniform over the block

  int tc = 0;  // running tile counter: buffer = tc & 1, ONE barrier per tile (the other buffer is
               // only rewritten after the next barrier, when every wave has left it)
  for (int ch = 0; ch < nchunks; ch++) {
    const int t = t0 + ch * 64 + lane;
    const bool wave_active = t0 + ch * 64 < t1;  // wave-uniform
    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    int cxi = 0;
    if (t < t1) {
      pi = sorted[t];
      if (STRICT) cxi = (int)(keys[t] % (unsigned int)gx);
    }
    // fp32 sums of <= 64 sources folded into fp64: a dense cell neighbourhood is thousands of
    // terms with heavy cancellation (uniform interior); long fp32 running sums would cost digits
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int r = 0; r < 9; r++) {
      const int s0 = rowpos[r][0], s1 = rowpos[r][BW + 2];
      // this wave's part of the row: cells [xw-1, xw+Wv]
      const int ws0 = rowpos[r][w * Wv], ws1 = rowpos[r][min(w * Wv + Wv + 2, BW + 2)];
      for (int jb = s0; jb < s1; jb += HTS, tc++) {
        const int b = tc & 1;
        const int j = jb + tid;
        if (j < s1) {
          tile[b][tid] = sorted[j];
          if (STRICT) tile_cx[b][tid] = (int)(keys[j] % (unsigned int)gx);
        }
        __syncthreads();
        const int k0 = wave_active ? max(ws0, jb) - jb : 0;
        const int k1 = wave_active ? min(ws1, min(s1, jb + HTS)) - jb : 0;
        for (int kb = k0; kb < k1; kb += 64) {
          float ax = 0.f, ay = 0.f, az = 0.f;
          const int ke = min(kb + 64, k1);
          // four sources per round: the four LDS reads are in flight together; each source keeps its
          // own test, so that the expensive part (rsq, factor, sums) is skipped by the whole wave when
          // no lane is inside its cutoff (frequent: half of the scanned cells are not neighbours)
          int k = kb;
          for (; k + 4 <= ke; k += 4) {
            float4 s[4];
#pragma unroll
            for (int q = 0; q < 4; q++) s[q] = tile[b][k + q];
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const float dx = s[q].x - pi.x, dy = s[q].y - pi.y, dz = s[q].z - pi.z;
              const float d2 = hash_dist2(dx, dy, dz);
              bool ok = d2 < cutoff2;                           // :131, unsoftened distance
              if (GUARD) ok = ok && (d2 > 0.f);                 // coincident / self: contributes 0
              if (STRICT) ok = ok && (abs(tile_cx[b][k + q] - cxi) <= 1);
              if (ok) {
                const float inv = __builtin_amdgcn_rsqf(d2 + eps2);
                const float f = (s[q].w * inv) * (inv * inv);
                ax = __builtin_fmaf(f, dx, ax);
                ay = __builtin_fmaf(f, dy, ay);
                az = __builtin_fmaf(f, dz, az);
              }
            }
          }
          for (; k < ke; k++) {
            const float4 s = tile[b][k];
            const float dx = s.x - pi.x, dy = s.y - pi.y, dz = s.z - pi.z;
            const float d2 = hash_dist2(dx, dy, dz);
            const float inv = __builtin_amdgcn_rsqf(d2 + eps2);
            bool ok = d2 < cutoff2;
            if (GUARD) ok = ok && (d2 > 0.f);
            if (STRICT) ok = ok && (abs(tile_cx[b][k] - cxi) <= 1);
            const float f = ok ? (s.w * inv) * (inv * inv) : 0.f;
            ax = __builtin_fmaf(f, dx, ax);
            ay = __builtin_fmaf(f, dy, ay);
            az = __builtin_fmaf(f, dz, az);
          }
          sx += (double)ax; sy += (double)ay; sz += (double)az;
        }
      }
    }
    if (t < t1) {
      const int i = idx[t];
      const float fx = (float)((double)G * sx), fy = (float)((double)G * sy), fz = (float)((double)G * sz);
      if (acc4) {
        acc4[i] = make_float4(fx, fy, fz, 0.f);
      } else {
        acc_x[i] = fx; acc_y[i] = fy; acc_z[i] = fz;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Force kernel for grids with a cell_lb array: ONE WAVE PER CELL.
// The wave's targets are the cnt bodies of one cell, and every one of them has the SAME 27-cell
// window: nine contiguous runs of the cell-ordered list (cells cx-1..cx+1 of the rows y+-1, z+-1),
// which the wave gathers into its own LDS region as one flat list -- exactly the candidates of the
// reference's search (force_spatial_hash.cu:104-129), no row overhang, no |cx_j - cx_i| test, no
// block barrier.  The 64 lanes are T target slots x S slices (T = ceil(cnt / R), S = 64 / T): slice s
// takes the window entries s, s + S, s + 2S, ... so the lanes of one LDS read touch S consecutive
// float4 (conflict-free, each broadcast to T lanes), and every lane does useful work on a candidate
// of ITS cell's window: 27 rho candidate pairs per body instead of the 54 rho of the cell-run kernel.
// Each lane keeps R targets in registers (one LDS read, R pair evaluations).  Slice partials are
// fp32 per <= 32 entries, folded into fp64, and summed over the slices through LDS at the end.
// ---------------------------------------------------------------------------------------
constexpr int kWinCap = 512;  // window entries a wave holds in LDS at a time (8 KiB)
// one window entry against the lane's targets: R = 1 scalar, R even -> R/2 packed pairs
// The cutoff decision of the packed form WITHOUT compare + select (round 4): with c- the float below cutoff^2, h its ulp
// and K = c- / h + 1 (an integer <= 2^24: exact),
//     t = clamp(fma(d2, -1 / h, K))   (v_pk_fma_f32 ... clamp: one packed instruction for both pairs)
// is exactly 1 for d2 <= c-  <=>  d2 < cutoff^2 (the exact value K - d2 / h is >= 1, and rounding is monotone) and exactly
// 0 for d2 >= cutoff^2 (exact value <= 0); m t is m or 0 exactly, so the factor (inv (m t)) (inv inv) is bit for bit the
// selected one.  Two packed instructions instead of two compares and two selects per two pairs: 19 instead of 21
// instructions per window entry in the loop that is 70 % of the kernel.  (A NaN distance -- a non-finite position, which
// the validators of the interface reject -- gives t = 0 under DX10 clamp but inv = NaN: the pair poisons the target's sum,
// where the compare form would skip it.  Finite inputs only: an infinite d2 gives inv = 0, t = 0, f = 0.)  Needs 1 / h and K representable: cutoff^2 in [2^-100, 2^100];
// outside, and when eps^2 < 1e-12, the GUARD instantiation (compare + select, d2 > 0 test) runs.
#ifndef NBH_HASH_PAIR4
#define NBH_HASH_PAIR4 1
#endif
#ifndef NBH_HASH_PK_BOX
#define NBH_HASH_PK_BOX 1
#endif
struct CutConst {
  float nbig, k;  // -1 / h, K
};
__host__ __device__ inline bool cut_const_ok(float cutoff2) { return cutoff2 >= 7.9e-31f && cutoff2 <= 1.2e30f; }
__device__ __forceinline__ CutConst cut_const(float cutoff2) {
  const unsigned cm = __builtin_bit_cast(unsigned, cutoff2) - 1u;   // c-: the float below cutoff^2 (positive, normal)
  const unsigned e = (cm >> 23) & 255u;                              // c- = 1.m x 2^(e - 127), h = 2^(e - 150)
  const float big = __builtin_bit_cast(float, (277u - e) << 23);     // 1 / h = 2^(150 - e)
  CutConst c;
  c.nbig = -big;
  c.k = __builtin_bit_cast(float, cm) * big + 1.0f;                  // exact: an integer in (2^23, 2^24]
  return c;
}

template <bool GUARD, int R>
struct CellTargets {
  static constexpr int NP = R / 2;
  f2 px[NP], py[NP], pz[NP], ax[NP], ay[NP], az[NP];
  f2 nbig, kk;  // the cutoff decision's constants (non-GUARD)
  __device__ __forceinline__ void set_cut(float cutoff2) {
    if constexpr (!GUARD) {
      const CutConst c = cut_const(cutoff2);
      nbig = (f2)(c.nbig);
      kk = (f2)(c.k);
    }
  }
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int j = 0; j < NP; j++) ax[j] = ay[j] = az[j] = (f2)(0.f);
  }
  __device__ __forceinline__ void set(int q, float x, float y, float z) {
    px[q >> 1][q & 1] = x; py[q >> 1][q & 1] = y; pz[q >> 1][q & 1] = z;
  }
  __device__ __forceinline__ void pair(const float4 s, float cutoff2, float eps2) {
#ifdef NBH_PROBE_NO_PAIRS  // timing probe (tools/): everything but the pair arithmetic -- one add keeps the LDS read alive
    ax[0] += (f2)(s.x);
    return;
#endif
#pragma unroll
    for (int j = 0; j < NP; j++) {
      const f2 dx = (f2)(s.x) - px[j], dy = (f2)(s.y) - py[j], dz = (f2)(s.z) - pz[j];
      const f2 d2 = hash_dist2(dx, dy, dz);
      const f2 de = d2 + (f2)(eps2);
      f2 inv;
      inv.x = __builtin_amdgcn_rsqf(de.x);
      inv.y = __builtin_amdgcn_rsqf(de.y);
      f2 f;
      if constexpr (GUARD) {
        f = (inv * (f2)(s.w)) * (inv * inv);
        bool ok0 = d2.x < cutoff2, ok1 = d2.y < cutoff2;  // :131, unsoftened distance
        ok0 = ok0 && (d2.x > 0.f); ok1 = ok1 && (d2.y > 0.f);  // coincident / self: contributes 0
        f.x = ok0 ? f.x : 0.f;
        f.y = ok1 ? f.y : 0.f;
      } else {
        // t = 1 where d2 < cutoff^2 (:131, unsoftened distance), else 0; tm = t m, the mass taken from the upper half of
        // the entry's (z, w) register pair by operand selection (the compiler copied w into a register of its own first);
        // -1 / h comes from its scalar register pair (one scalar operand per instruction is allowed)
        f2 t, tm;
        const f2 zw = {s.z, s.w};
        asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(t) : "v"(d2), "s"(nbig), "v"(kk));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(tm) : "v"(t), "v"(zw));
        f = (inv * tm) * (inv * inv);
      }
      ax[j] = __builtin_elementwise_fma(f, dx, ax[j]);
      ay[j] = __builtin_elementwise_fma(f, dy, ay[j]);
      az[j] = __builtin_elementwise_fma(f, dz, az[j]);
    }
  }
#if NBH_HASH_PAIR4
  // four window entries, stage by stage (the four dependent chains side by side: the compiler otherwise runs one entry's
  // chain after the other and pads the back-to-back dependent packed instructions with s_nop)
  __device__ __forceinline__ void pair4(const float4 e0, const float4 e1, const float4 e2, const float4 e3, float eps2) {
    static_assert(NP == 1 && !GUARD, "two targets per lane, compare-free decision");
    const float4 e[4] = {e0, e1, e2, e3};
    f2 dx[4], dy[4], dz[4], d2[4], inv[4], t[4], tm[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { dx[k] = (f2)(e[k].x) - px[0]; dy[k] = (f2)(e[k].y) - py[0]; dz[k] = (f2)(e[k].z) - pz[0]; }
#pragma unroll
    for (int k = 0; k < 4; k++) d2[k] = hash_dist2(dx[k], dy[k], dz[k]);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const f2 de = d2[k] + (f2)(eps2);
      inv[k].x = __builtin_amdgcn_rsqf(de.x);
      inv[k].y = __builtin_amdgcn_rsqf(de.y);
      asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(t[k]) : "v"(d2[k]), "s"(nbig), "v"(kk));
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const f2 zw = {e[k].z, e[k].w};
      asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(tm[k]) : "v"(t[k]), "v"(zw));
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const f2 f = (inv[k] * tm[k]) * (inv[k] * inv[k]);
      ax[0] = __builtin_elementwise_fma(f, dx[k], ax[0]);
      ay[0] = __builtin_elementwise_fma(f, dy[k], ay[0]);
      az[0] = __builtin_elementwise_fma(f, dz[k], az[0]);
    }
  }
#endif
  __device__ __forceinline__ float get(int q, int c) const {
    return c == 0 ? ax[q >> 1][q & 1] : (c == 1 ? ay[q >> 1][q & 1] : az[q >> 1][q & 1]);
  }
};
template <bool GUARD>
struct CellTargets<GUARD, 1> {
  float px, py, pz, ax, ay, az;
  __device__ __forceinline__ void set_cut(float) {}
  __device__ __forceinline__ void clear() { ax = ay = az = 0.f; }
  __device__ __forceinline__ void set(int, float x, float y, float z) { px = x; py = y; pz = z; }
  __device__ __forceinline__ void pair(const float4 s, float cutoff2, float eps2) {
    const float dx = s.x - px, dy = s.y - py, dz = s.z - pz;
    const float d2 = hash_dist2(dx, dy, dz);
    const float inv = __builtin_amdgcn_rsqf(d2 + eps2);
    bool ok = d2 < cutoff2;
    if (GUARD) ok = ok && (d2 > 0.f);
    const float f = ok ? (s.w * inv) * (inv * inv) : 0.f;
    ax = __builtin_fmaf(f, dx, ax);
    ay = __builtin_fmaf(f, dy, ay);
    az = __builtin_fmaf(f, dz, az);
  }
  __device__ __forceinline__ float get(int, int c) const { return c == 0 ? ax : (c == 1 ? ay : az); }
};

// The box of a wave's targets: minimum / maximum over the 64 lanes of three coordinates each, without LDS.  Four row
// shifts (the running extremum of a row of 16 ends up in its last lane), then gfx9's two row broadcasts (lane 15 of a row
// into the next row, lane 31 into the upper half): six DPP instructions per value, the result in lane 63, returned
// wave-uniform (SGPRs).  Written out in assembly: through __builtin_amdgcn_update_dpp + fminf the compiler produced a
// v_mov_b32_dpp, a copy and up to three canonicalising v_max_f32 per step (150 instructions, as many as two LDS
// shuffles per step cost); here the six chains are interleaved, which also puts the five independent instructions between
// a write and the DPP read of the same register that the hardware wants (two wait states; s_nop in front for the first).
// A lane without a source in its row (bound_ctrl off) is not written and keeps its value.
#ifndef NBH_HASH_DPP_BOX
#define NBH_HASH_DPP_BOX 1
#endif
#ifndef NBH_HASH_PREFETCH_FIX
#define NBH_HASH_PREFETCH_FIX 1
#endif
#ifndef NBH_HASH_PAD_TAIL
#define NBH_HASH_PAD_TAIL 1
#endif
#ifndef NBH_HASH_PAR_REDUCE
#define NBH_HASH_PAR_REDUCE 1
#endif
__device__ __forceinline__ void wave_box(float (&lo)[3], float (&hi)[3]) {
#define NBH_BOX_STEP(ctrl)                       \
  "v_min_f32_dpp %0, %0, %0 " ctrl "\n\t"        \
  "v_min_f32_dpp %1, %1, %1 " ctrl "\n\t"        \
  "v_min_f32_dpp %2, %2, %2 " ctrl "\n\t"        \
  "v_max_f32_dpp %3, %3, %3 " ctrl "\n\t"        \
  "v_max_f32_dpp %4, %4, %4 " ctrl "\n\t"        \
  "v_max_f32_dpp %5, %5, %5 " ctrl "\n\t"
  asm("s_nop 1\n\t"
      NBH_BOX_STEP("row_shr:1 row_mask:0xf bank_mask:0xf")
      NBH_BOX_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
      NBH_BOX_STEP("row_shr:4 row_mask:0xf bank_mask:0xf")
      NBH_BOX_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
      NBH_BOX_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
      NBH_BOX_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
      : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]));
#undef NBH_BOX_STEP
#pragma unroll
  for (int a = 0; a < 3; a++) {
    lo[a] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lo[a]), 63));
    hi[a] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hi[a]), 63));
  }
}

// KC consecutive cells per wave, software-pipelined: while the wave evaluates cell c out of LDS, the
// loads of cell c+1's window are already in flight (into registers), so that only the first of a
// wave's cells pays the global-memory latency of its lookups.
#ifndef NBH_HASH_KC
#define NBH_HASH_KC 4
#endif
constexpr int kCellsPerWave = NBH_HASH_KC;  // (2 and 3 measured in round 4: 0.635 / 0.639-0.653 against 0.634-0.636 ms: no difference)
static_assert(kCellsPerWave >= 1 && kCellsPerWave <= 4, "the lookups of a wave's cells are one round of 16 lanes per cell");
constexpr double kFilterFrom = 40.0;  // bodies per cell from which the filtered form pays when cutoff > cell (see FILTER below)
constexpr double kBodyBelow = 8.0;    // bodies per cell below which one lane takes one body (hash_body_force_kernel): 0.72 against
                                      // 0.96 ms at 6.6 per cell, 0.85 against 0.81 at 8.8 (profiles/r04_hash_kernels.txt)
constexpr int kSplitFrom = 500000;   // bodies from which the automatic form below kBodyBelow is the split one
constexpr int kSplitCnt = NBH_HASH_SPLIT_CNT;  // ... except the bodies of cells this crowded: wave per cell (split form)
constexpr double kFilterFromInside = 8.0;  // ... and when cutoff <= cell: everywhere the wave-per-cell form runs at all (with the
                                          // straight-line gather of round 4 the box test pays from ~8 per cell: 0.81 against
                                          // 0.84 ms at 8.8 per cell, 0.78 against 0.86 at 11.2, 0.68 against 0.78 at 14.6,
                                          // 0.52 against 0.93 at cutoff = cell / 2: profiles/r04_hash_kernels.txt)

// A grid as the force kernel sees it.  lb covers the cells [base, base + count] of the (global) grid --
// the whole grid, or the z-slab a rank holds (sharded path); cells outside hold no bodies of this grid.
struct CellGridView {
  const float4* sorted;
  const int* lb;
  const int* idx;
  long long base, count;
  __device__ __forceinline__ int lower(long long c) const {
    const long long k = c - base;
    return lb[k < 0 ? 0 : (k > count ? count : k)];
  }
};

// Targets: the bodies of grid `tg` in the cells [cell_first, cell_end); sources: grid `sg` (the same grid,
// or -- sharded path -- the halo layers received from the neighbouring ranks).  ACCUM adds to acc4.
// HALF (timing probe only, nbody_hip_grid_tuning kernel 5): the window is cut to the half shell -- the cell itself
// and its 13 "forward" neighbours -- and NO reactions are applied, so the output is not the force.  Its time is a
// lower bound for any Newton's-third-law form of this kernel (half the candidate pairs, reaction arithmetic and
// reaction traffic free); DESIGN.md section 4.4 sets it against the cost of deterministic reaction slots.
//
// UNITS: the work list of cell_units_kernel instead of the cell range itself.  A unit is (occupied cell, chunk of at
// most 64 R of its bodies): empty cells cost nothing and a crowded cell is spread over as many waves as it has chunks
// (a uniform box that has clumped under its own gravity holds cells of hundreds of bodies beside a majority of empty
// ones: one wave per cell then leaves the step waiting for a few waves).  Wave w of the grid takes the unit groups
// w, w + waves, ... of the list, which holds the heavy units first; the host sizes the grid from the previous count, so
// a wave usually takes one group and the hardware's dispatch order (workgroups in index order) does the balancing:
// long units early, short ones in the tail.  Chunks are the same 64 R targets the cell loop forms, so both forms sum
// in the same order: bit-identical results.
//
// FILTER (crowded cells: from ~40 bodies per occupied cell): window entries farther than the cutoff from the BOX of the
// chunk's targets are left out of LDS -- they fail the cutoff test against every one of them.  Of the 27 cells' bodies
// only 1 + 6 + 12 pi/4 + 8 pi/6 = 20.6 cells' worth can reach a unit cell at all (fewer for the tighter box of the
// bodies): at 107 bodies per cell the kernel takes 4.6 ms instead of 5.8; at 15 per cell the per-entry box test and the
// compacting store cost more than the quarter of the pair loop they save (1.04 against 0.94 ms), so it is a form of its
// own.  Exact: what is kept is evaluated as before (in window order), what is dropped would have contributed 0; the
// margin of 1e-5 covers the rounding of the two distance computations.  Not bit-identical to the unfiltered form: the
// kept entries fall to other slices, so the partial sums group differently.
// Waves per workgroup.  The waves of a workgroup share nothing (every wave has its own LDS region and its own cells); what
// the workgroup size decides is when the LDS and the wave slots come back: a workgroup's resources are released when its
// LAST wave ends, and the waves of one workgroup take very different times (four cells each, 5 to 25 bodies).
#ifndef NBH_HASH_CELL_WPB
#define NBH_HASH_CELL_WPB 1
#endif
constexpr int kCellWPB = NBH_HASH_CELL_WPB;
static_assert(kCellWPB == 1 || kCellWPB == 2 || kCellWPB == 4, "the launch geometry is counted in groups of four waves");
template <bool GUARD, int R, bool HALF = false, bool UNITS = false, bool FILTER = false>
__global__ __launch_bounds__(64 * kCellWPB) __attribute__((amdgpu_waves_per_eu(4, 4))) void hash_cell_force_kernel(
    const CellGridView tgv, const CellGridView sgv, int gx, int gy, int gz, long long cell_first,
    long long cell_end, int blocks_per_xcd, float cutoff2, float eps2, float G, float* __restrict__ acc_x,
    float* __restrict__ acc_y, float* __restrict__ acc_z, float4* __restrict__ acc4, int accumulate,
    const int2* __restrict__ units = nullptr, const int* __restrict__ unit_count = nullptr,
    int* __restrict__ unit_count_host = nullptr, int unit_capacity = 0, int crowded_only = 0) {
  constexpr int KC = kCellsPerWave;
  __shared__ float4 win_all[kCellWPB][kWinCap + 64];  // (+ 64: the padding of the filtered form's last round)
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  float4* win = win_all[w];
  double* red = reinterpret_cast<double*>(win);  // reused after the pair loop: [R][3][64] doubles
  static_assert(R * 3 * 64 * sizeof(double) <= kWinCap * sizeof(float4), "reduction area");
  const float4* __restrict__ sorted = sgv.sorted;    // window entries
  const float4* __restrict__ tsorted = tgv.sorted;   // targets
  const int* __restrict__ idx = tgv.idx;
  // workgroup b runs on XCD b mod 8: give every XCD one contiguous eighth of the cells (z slabs), so
  // that neighbouring cells, whose windows overlap, share an L2
  // (blocks_per_xcd counts groups of four waves whatever the workgroup size)
  const long long wid = (long long)(blockIdx.x & 7) * (blocks_per_xcd * 4) + (long long)(blockIdx.x >> 3) * kCellWPB + w;
  const long long cell0 = cell_first + wid * KC;
  int grp = 0, grp_end = 0, n_units = 0, n_heavy = 0;  // UNITS: this wave's next group of KC units, the end of the list
  if constexpr (UNITS) {
    n_heavy = __builtin_amdgcn_readfirstlane(unit_count[0]);
    n_units = n_heavy + __builtin_amdgcn_readfirstlane(unit_count[2]);
    if (unit_count_host && blockIdx.x == 0 && threadIdx.x == 0) {  // chooses and sizes the next launch
      unit_count_host[0] = n_units;
      unit_count_host[1] = unit_count[1];
    }
    // split form (launch_cell_forces kern 9): without a crowded cell the one-lane-per-body kernel has taken every body
    if (crowded_only && __builtin_amdgcn_readfirstlane(unit_count[1]) == 0) return;
    // the list front to back in workgroup order (heavy units first, see cell_units_kernel)
    grp = (int)blockIdx.x * kCellWPB + w;
    grp_end = (n_units + KC - 1) / KC;
  } else {
    if (cell0 >= cell_end) return;
  }
  for (; !UNITS || grp < grp_end; grp += blocks_per_xcd * 32) {
  // One round of lookups for all KC cells: lane 16 c + r.  r < 9: run r of cell c's window (cells
  // cx-1..cx+1 of row y + r%3 - 1, z + r/3 - 1): vseg0 = first sorted position, vlen = length;
  // r = 9: vseg0 = first target of the cell, r = 10: vseg0 = end of its targets (r = 11, UNITS: the chunk).
  int vseg0 = 0, vlen = 0;
  {
    const int c = lane >> 4, r = lane & 15;
    long long cell = cell0 + c;
    bool have = c < KC && cell < cell_end;
    int chunk = 0;
    if constexpr (UNITS) {
      const int u = grp * KC + c;
      have = c < KC && u < n_units;
      if (have) {
        const int2 uu = units[u < n_heavy ? u : unit_capacity - 1 - (u - n_heavy)];
        cell = cell_first + uu.x;
        chunk = uu.y;
      }
    }
    if (have) {
      if (r < 9) {
        // (32-bit: a grid holds at most 1e8 cells; the 64-bit divisions this replaces were ~100 instructions each)
        const unsigned int c32 = (unsigned int)cell, layer = (unsigned int)gx * (unsigned int)gy;
        const unsigned int uz = c32 / layer, rem = c32 - uz * layer, uy = rem / (unsigned int)gx;
        const int cx = (int)(rem - uy * (unsigned int)gx), cy = (int)uy, cz = (int)uz;
        const int yy = cy + (r % 3) - 1, zz = cz + (r / 3) - 1;
        if (yy >= 0 && yy < gy && zz >= 0 && zz < gz && !(HALF && r < 4)) {
          const long long base = ((long long)zz * gy + yy) * gx;
          vseg0 = sgv.lower(base + max(cx - ((HALF && r == 4) ? 0 : 1), 0));
          vlen = sgv.lower(base + min(cx + 2, gx)) - vseg0;
        }
      } else if (r < 11) {
        vseg0 = tgv.lower(cell + (r - 9));
      } else if (r == 11) {
        vseg0 = chunk;
      }
    }
  }
  // exclusive prefix of the run lengths inside every group of 16 lanes: lane 16 c + r = flat position
  // of run r, lane 16 c + 9 = window length
  int vpre;
  {
    int incl = vlen;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const int up = __shfl_up(incl, off, 16);
      if ((lane & 15) >= off) incl += up;
    }
    vpre = incl - vlen;
  }
#define NBH_SEG0(c, r) __builtin_amdgcn_readlane(vseg0, 16 * (c) + (r))
#define NBH_PRE(c, r) __builtin_amdgcn_readlane(vpre, 16 * (c) + (r))

  float4 pf[9];  // first 64 entries of each run of the NEXT cell to be evaluated
  const unsigned lane16 = (unsigned)lane << 4;
  // (the explicit wait and the lane reads in front of the branches: the compiler's wait-count pass merges "maybe pending"
  // at every join, and with the lookup registers first read inside the conditional blocks it put s_waitcnt vmcnt(0) in
  // front of every one of the nine loads -- nine round trips to memory one after the other, per cell)
#if NBH_HASH_PREFETCH_FIX
#define NBH_PREFETCH(c)                                                                              \
  __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0): nothing useful is in flight here */              \
  _Pragma("unroll") for (int r = 0; r < 9; r++) {                                                    \
    const int p0 = NBH_PRE(c, r), b = FILTER ? NBH_PRE(c, r + 1) : min(NBH_PRE(c, r + 1), kWinCap);  \
    const int sg = NBH_SEG0(c, r);                                                                   \
    pf[r] = make_float4(0.f, 0.f, 0.f, 0.f);                                                         \
    if (p0 < b) /* uniform base + one clamped 32-bit byte offset: one VALU instruction per load */    \
      pf[r] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sorted + sg) +          \
                                               (size_t)min(lane16, (unsigned)(b - 1 - p0) << 4));    \
  }
#else
#define NBH_PREFETCH(c)                                                                              \
  _Pragma("unroll") for (int r = 0; r < 9; r++) {                                                    \
    const int p0 = NBH_PRE(c, r), b = FILTER ? NBH_PRE(c, r + 1) : min(NBH_PRE(c, r + 1), kWinCap);  \
    pf[r] = make_float4(0.f, 0.f, 0.f, 0.f);                                                         \
    if (p0 < b) pf[r] = sorted[NBH_SEG0(c, r) + (min(p0 + lane, b - 1) - p0)];                       \
  }
#endif
  NBH_PREFETCH(0)

  // (the filtered form is not unrolled over the wave's cells: 46 KB of code ran 10 % slower than 24 KB -- instruction
  // fetch; the plain form is small enough: 30 KB unrolled and 2 % faster than rolled)
#ifndef NBH_HASH_CELL_UNROLL
#define NBH_HASH_CELL_UNROLL FILTER ? 1 : KC
#endif
#pragma unroll NBH_HASH_CELL_UNROLL
  for (int c = 0; c < KC; c++) {
    int t0 = NBH_SEG0(c, 9), t1 = NBH_SEG0(c, 10);
    if constexpr (UNITS) {  // one chunk of the cell's bodies
      t0 += NBH_SEG0(c, 11) * (64 * R);
      t1 = min(t1, t0 + 64 * R);
    }
    const int Lw = NBH_PRE(c, 9);
    bool prefetched = true;  // the registers pf hold batch 0 of this cell
    for (int tb = t0; tb < t1; tb += 64 * R) {
      const int cnt = min(t1 - tb, 64 * R);
      const int T = (cnt + R - 1) / R;  // target slots
      const int S = 64 / T;             // slices
      // lane = sl T + slot without a per-lane division: T <= 64 is wave-uniform, so floor(lane / T) is
      // (lane * ceil(2^16 / T)) >> 16 (exact for lane < 1024)
      const int magic = (65536 + T - 1) / T;
      const int sl = (lane * magic) >> 16, slot = lane - sl * T;
      const bool live = sl < S;
      CellTargets<GUARD, R> tg;
      tg.set_cut(cutoff2);
      double sx[R], sy[R], sz[R];
      [[maybe_unused]] float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int q = 0; q < R; q++) {
        // (uniform base + a clamped 32-bit byte offset: cnt <= 64 R targets, so the offset is below 4 KiB)
        const float4 p = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tsorted + tb) +
                                                          (size_t)((unsigned)min(slot + q * T, t1 - 1 - tb) << 4));
        tg.set(q, p.x, p.y, p.z);
        sx[q] = sy[q] = sz[q] = 0.0;
        if constexpr (FILTER) {
          blo[0] = fminf(blo[0], p.x); bhi[0] = fmaxf(bhi[0], p.x);
          blo[1] = fminf(blo[1], p.y); bhi[1] = fmaxf(bhi[1], p.y);
          blo[2] = fminf(blo[2], p.z); bhi[2] = fmaxf(bhi[2], p.z);
        }
      }
      if constexpr (FILTER) {  // the box of the chunk's targets (every lane holds valid targets)
#if NBH_HASH_DPP_BOX
        wave_box(blo, bhi);
#else
#pragma unroll
        for (int a = 0; a < 3; a++) {
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) {
            blo[a] = fminf(blo[a], __shfl_xor(blo[a], off, 64));
            bhi[a] = fmaxf(bhi[a], __shfl_xor(bhi[a], off, 64));
          }
        }
#endif
      }
      if constexpr (FILTER) {
      // the pair loop over the Lb entries LDS holds
      auto evaluate = [&](int Lb) {
        const int iters = (Lb + S - 1) / S;
#if NBH_HASH_PAD_TAIL
        // entries of mass 0 up to iters S: every slice then holds exactly iters entries and the loop needs neither a
        // remainder nor a validity test (a padding entry contributes (inv (t 0)) (inv inv) = +0: eps^2 > 0 here, and the
        // GUARD form selects its 0); the region has room for kWinCap + 64
        if (lane < iters * S - Lb) win[Lb + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float4* wp = win + (live ? sl : 0);
        const int full = NBH_HASH_PAD_TAIL ? iters : iters - 1;
        // (fp32 partial sums of at most 64 entries -- what the a-priori bound of DESIGN.md section 4.4 is derived for, and
        // what the lane-per-body kernel does; the unfiltered form below folds every 32)
        for (int i0 = 0; i0 < full; i0 += 64) {
          const int i1 = min(i0 + 64, full);
          tg.clear();
          if constexpr (R == 1 || GUARD) {
#pragma unroll 4
            for (int it = i0; it < i1; it++, wp += S) tg.pair(*wp, cutoff2, eps2);
          } else {
            // (four entries a round, written out: the packed form's inline-assembly decision keeps the loop unroller away)
            int it = i0;
            for (; it + 4 <= i1; it += 4, wp += 4 * S) {
              const float4 e0 = wp[0], e1 = wp[S], e2 = wp[2 * S], e3 = wp[3 * S];
#if NBH_HASH_PAIR4
              if constexpr (R == 2) {
                tg.pair4(e0, e1, e2, e3, eps2);
              } else
#endif
              {
                tg.pair(e0, cutoff2, eps2);
                tg.pair(e1, cutoff2, eps2);
                tg.pair(e2, cutoff2, eps2);
                tg.pair(e3, cutoff2, eps2);
              }
            }
            for (; it < i1; it++, wp += S) tg.pair(*wp, cutoff2, eps2);
          }
#pragma unroll
          for (int q = 0; q < R; q++) { sx[q] += (double)tg.get(q, 0); sy[q] += (double)tg.get(q, 1); sz[q] += (double)tg.get(q, 2); }
        }
#if !NBH_HASH_PAD_TAIL
        {  // the lane's last entry may lie past the batch
          const bool valid = (int)(wp - win) < Lb;
          float4 s = win[valid ? (int)(wp - win) : 0];
          if (!valid) s.w = 0.f;
          tg.clear();
          tg.pair(s, cutoff2, eps2);
#pragma unroll
          for (int q = 0; q < R; q++) { sx[q] += (double)tg.get(q, 0); sy[q] += (double)tg.get(q, 1); sz[q] += (double)tg.get(q, 2); }
        }
#endif
        __builtin_amdgcn_wave_barrier();
      };
      // the window into LDS, run by run, 64 entries at a time, the entries out of reach left out (stable: the kept ones
      // stay in window order); whenever LDS is full the pair loop runs over what it holds.  The first 64 entries of
      // every run come from the round of loads issued while the previous cell was evaluated.
      const float keep2 = cutoff2 * 1.00001f;
      int wcount = 0;
      __builtin_amdgcn_wave_barrier();
      [[maybe_unused]] const f2 lo_xy = {blo[0], blo[1]}, hi_xy = {bhi[0], bhi[1]};
      // one round: the box test and the compacting store.  `left` = entries of the run from this round on: the lanes
      // behind them (which hold a clamped duplicate) are masked out of the ballot on the scalar side -- no per-lane
      // validity compare, and the store's predicate is the mask itself
      auto put = [&](const float4 e, const int left) {
#if NBH_HASH_PK_BOX
        // (x and y side by side in packed subtractions: 10 instead of 13 instructions per round)
        const f2 exy = {e.x, e.y};
        const f2 a = lo_xy - exy, b = exy - hi_xy;
        const float ex = fmaxf(fmaxf(a.x, b.x), 0.f), ey = fmaxf(fmaxf(a.y, b.y), 0.f);
#else
        const float ex = fmaxf(fmaxf(blo[0] - e.x, e.x - bhi[0]), 0.f);
        const float ey = fmaxf(fmaxf(blo[1] - e.y, e.y - bhi[1]), 0.f);
#endif
        const float ez = fmaxf(fmaxf(blo[2] - e.z, e.z - bhi[2]), 0.f);
        // (a NaN distance -- non-finite positions -- keeps the entry: its pairs take the ordinary path)
        const unsigned long long have_mask = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
        const unsigned long long mask = __ballot(!(__builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex)) > keep2)) & have_mask;
        if (__builtin_amdgcn_inverse_ballot_w64(mask))
          win[wcount + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u))] = e;
        wcount += __builtin_amdgcn_readfirstlane(__popcll(mask));
      };
      if (Lw <= kWinCap && prefetched) {
        // the whole window fits (every cell but the crowded ones) and its first 64 entries per run are in registers: nine
        // straight-line rounds and ONE pair loop behind them
#pragma unroll
        for (int r = 0; r < 9; r++) {
          const int len = NBH_PRE(c, r + 1) - NBH_PRE(c, r), seg = NBH_SEG0(c, r);
          if (len > 0) put(pf[r], len);
          for (int v = 64; v < len; v += 64) put(sorted[seg + min(v + lane, len - 1)], len - v);
        }
      } else {
#pragma unroll 1
        for (int r = 0; r < 9; r++) {
          const int len = NBH_PRE(c, r + 1) - NBH_PRE(c, r), seg = NBH_SEG0(c, r);
          for (int v = 0; v < len; v += 64) {
            // (a window of at most kWinCap entries -- a later chunk of a crowded cell in a sparse neighbourhood -- is never
            // split: the same single batch as the branch above, whichever of the two a form of the kernel takes)
            if (Lw > kWinCap && wcount + 64 > kWinCap) {
              evaluate(wcount);
              wcount = 0;
            }
            put(sorted[seg + min(v + lane, len - 1)], len - v);
          }
        }
      }
      if (prefetched) {
        prefetched = false;
        if (c + 1 < KC) { NBH_PREFETCH(c + 1) }  // in flight while this cell is evaluated
      }
      if (wcount) evaluate(wcount);
      } else {
      for (int vb = 0; vb < Lw; vb += kWinCap) {
        const int Lb = min(Lw - vb, kWinCap);
        __builtin_amdgcn_wave_barrier();
        // the batch [vb, vb + Lb) of the flat window list into LDS: the first 64 entries of every run
        // come from one round of loads (a run is three cells: usually all of it), the rest in a loop
        if (prefetched) {
#pragma unroll
          for (int r = 0; r < 9; r++) {
            const int p0 = NBH_PRE(c, r), b = min(NBH_PRE(c, r + 1), Lb);
            if (p0 + lane < b) win[p0 + lane] = pf[r];
            for (int v = p0 + 64 + lane; v < b; v += 64) win[v] = sorted[NBH_SEG0(c, r) + (v - p0)];
          }
          prefetched = false;
          if (c + 1 < KC) { NBH_PREFETCH(c + 1) }  // in flight while this cell is evaluated
        } else {  // later batches of a long window, later chunks of a crowded cell: plain loops
#pragma unroll
          for (int r = 0; r < 9; r++) {
            const int p0 = NBH_PRE(c, r), a = max(p0, vb), b = min(NBH_PRE(c, r + 1), vb + Lb);
            for (int v = a + lane; v < b; v += 64) win[v - vb] = sorted[NBH_SEG0(c, r) + (v - p0)];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // slice sl takes entries sl, sl + S, ...: all but a lane's last one are inside the batch
        // (fp32 sums of <= 32 entries, folded into fp64: a dense neighbourhood is hundreds of terms with
        // heavy cancellation; long fp32 running sums would cost digits)
        const int iters = (Lb + S - 1) / S;
        const float4* wp = win + (live ? sl : 0);
        for (int i0 = 0; i0 < iters - 1; i0 += 32) {
          const int i1 = min(i0 + 32, iters - 1);
          tg.clear();
          if constexpr (R == 1 || GUARD) {
#pragma unroll 4
            for (int it = i0; it < i1; it++, wp += S) tg.pair(*wp, cutoff2, eps2);
          } else {
            // (four entries a round, written out: the packed form's inline-assembly decision keeps the loop unroller away)
            int it = i0;
            for (; it + 4 <= i1; it += 4, wp += 4 * S) {
              const float4 e0 = wp[0], e1 = wp[S], e2 = wp[2 * S], e3 = wp[3 * S];
              tg.pair(e0, cutoff2, eps2);
              tg.pair(e1, cutoff2, eps2);
              tg.pair(e2, cutoff2, eps2);
              tg.pair(e3, cutoff2, eps2);
            }
            for (; it < i1; it++, wp += S) tg.pair(*wp, cutoff2, eps2);
          }
#pragma unroll
          for (int q = 0; q < R; q++) { sx[q] += (double)tg.get(q, 0); sy[q] += (double)tg.get(q, 1); sz[q] += (double)tg.get(q, 2); }
        }
        {  // the lane's last entry may lie past the batch
          const bool valid = (int)(wp - win) < Lb;
          float4 s = win[valid ? (int)(wp - win) : 0];
          if (!valid) s.w = 0.f;
          tg.clear();
          tg.pair(s, cutoff2, eps2);
#pragma unroll
          for (int q = 0; q < R; q++) { sx[q] += (double)tg.get(q, 0); sy[q] += (double)tg.get(q, 1); sz[q] += (double)tg.get(q, 2); }
        }
      }
      }
      // sum over the slices
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < R; q++) {
        red[(q * 3 + 0) * 64 + lane] = live ? sx[q] : 0.0;
        red[(q * 3 + 1) * 64 + lane] = live ? sy[q] : 0.0;
        red[(q * 3 + 2) * 64 + lane] = live ? sz[q] : 0.0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#if NBH_HASH_PAR_REDUCE
      // lane (target q, component, slot) of the first 3 R T: the S slice sums of one component of one target, in slice order
      // as before (the same fp64 additions: bit-identical), on 3 R times as many lanes as one lane per slot -- one pass
      // for cells of up to 20 bodies
      for (int j = lane; j < 3 * R * T; j += 64) {
        const int qc = (j * magic) >> 16, sl2 = j - qc * T;  // j / T, j % T (T <= 32, j < 384: exact)
        const int q = qc / 3, comp = qc - 3 * q;
        const int t = tb + sl2 + q * T;
        if (t < tb + cnt) {
          const double* rp = red + qc * 64 + sl2;
          double f = 0.0;
          for (int s2 = 0; s2 < S; s2++, rp += T) f += *rp;
          const int i = idx[t];
          const float o = (float)((double)G * f);
          if (acc4) {
            float* a = reinterpret_cast<float*>(acc4 + i);
            a[comp] = accumulate ? a[comp] + o : o;
            if (comp == 0) a[3] = 0.f;
          } else {
            (comp == 0 ? acc_x : (comp == 1 ? acc_y : acc_z))[i] = o;
          }
        }
      }
#else
      if (lane < T) {
#pragma unroll
        for (int q = 0; q < R; q++) {
          const int t = tb + lane + q * T;
          if (t < tb + cnt) {
            double fx = 0.0, fy = 0.0, fz = 0.0;
            for (int s2 = 0; s2 < S; s2++) {
              fx += red[(q * 3 + 0) * 64 + lane + s2 * T];
              fy += red[(q * 3 + 1) * 64 + lane + s2 * T];
              fz += red[(q * 3 + 2) * 64 + lane + s2 * T];
            }
            const int i = idx[t];
            const float ox = (float)((double)G * fx), oy = (float)((double)G * fy), oz = (float)((double)G * fz);
            if (acc4) {
              if (accumulate) {
                const float4 o = acc4[i];
                acc4[i] = make_float4(o.x + ox, o.y + oy, o.z + oz, 0.f);
              } else {
                acc4[i] = make_float4(ox, oy, oz, 0.f);
              }
            } else {
              acc_x[i] = ox; acc_y[i] = oy; acc_z[i] = oz;
            }
          }
        }
      }
#endif
    }
    if (prefetched && c + 1 < KC) {  // an empty cell (or an empty window): pass the pipeline on
      __builtin_amdgcn_wave_barrier();
      NBH_PREFETCH(c + 1)
    }
  }
    if constexpr (!UNITS) break;
    __builtin_amdgcn_wave_barrier();
  }
#undef NBH_SEG0
#undef NBH_PRE
#undef NBH_PREFETCH
}

// ---------------------------------------------------------------------------------------
// TWO-PHASE form of the wave-per-cell kernel (round 4; nbody_hip_grid_tuning 7, automatic for cutoff <= cell from 8
// bodies per cell).  Of the 27 rho candidates of a target only (4 pi / 3) / 27 = 15.5 % lie inside the cutoff sphere
// (cutoff = cell); the one-phase kernel above pays the whole pair evaluation (distance, rsq, factor, three FMAs: ~33 ns
// per wave step and SIMD) for every one of them, because with 64 lanes on 64 different pairs some lane is always inside.
// Here the decision and the evaluation are separated:
//   phase 1: every lane tests ITS candidates against its four targets with the distance chain only (3 sub, mul, 2 fma:
//            bit for bit the d2 of hash_dist2, so the decision is the reference's, :131) and shifts the outcome into a
//            32-bit mask per target (v_cmp + v_addc_co: the carry is the new bit) -- 8 VALU per candidate pair, ~8 ns;
//   phase 2: per target, the lanes walk the set bits of their masks (v_ffbh) and evaluate only those entries, in window
//            order, with the one-phase kernel's arithmetic (same operations in the same order on the same operands: the
//            accepted terms are bit-identical; only the grouping of the partial sums differs).
// Lanes = T target slots x S slices as before, four targets per lane (T = ceil(cnt / 4) <= 4 for a pass of <= 16 targets,
// S = 64 / T >= 16 slices, so a batch of <= 512 window entries is <= 32 candidates per lane: one mask word per target).
// A lane's partial sums are fp32 over its <= 32 candidates of a batch, folded into fp64 per batch; the slices are summed
// in fp64 through LDS in a fixed order: bitwise reproducible.  Cells of more than 16 bodies take further passes over the
// window, which stays in LDS.  What was measured is in DESIGN.md section 4.4.
// ---------------------------------------------------------------------------------------
constexpr int kWinCap2 = 512;                 // window entries per batch (<= 32 per lane at S >= 16 slices)
constexpr int kR2 = 4;                        // targets per lane
constexpr int kPass2 = 16;                    // targets per pass
constexpr int kUnitChunk2 = 64;               // UNITS: bodies per unit (four passes share one window load)
static_assert(kWinCap2 <= 32 * (64 / (kPass2 / kR2)), "one mask word per target and batch");

// mask <- (mask << 1) | (d < thr): the compare's carry-out is the add's carry-in
__device__ __forceinline__ void mask_push_lt(unsigned& m, float d, float thr) {
  asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(d), "v"(thr) : "vcc");
}

// the same for two window entries x four targets in one block (entry 0's bits first: the masks stay in window order)
// (w0, w1: the entries' fourth components, named as inputs so that the compiler keeps the reads whole: ds_read_b128
// takes 4 LDS cycles, the b96 it would otherwise choose takes 8)
__device__ __forceinline__ void mask_push_lt8(unsigned (&m)[4], const float (&d0)[4], const float (&d1)[4], const float (&thr)[4],
                                              float w0, float w1) {
  asm volatile(
      "v_cmp_lt_f32 vcc, %4, %12\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
      "v_cmp_lt_f32 vcc, %5, %13\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc\n\t"
      "v_cmp_lt_f32 vcc, %6, %14\n\tv_addc_co_u32 %2, vcc, %2, %2, vcc\n\t"
      "v_cmp_lt_f32 vcc, %7, %15\n\tv_addc_co_u32 %3, vcc, %3, %3, vcc\n\t"
      "v_cmp_lt_f32 vcc, %8, %12\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
      "v_cmp_lt_f32 vcc, %9, %13\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc\n\t"
      "v_cmp_lt_f32 vcc, %10, %14\n\tv_addc_co_u32 %2, vcc, %2, %2, vcc\n\t"
      "v_cmp_lt_f32 vcc, %11, %15\n\tv_addc_co_u32 %3, vcc, %3, %3, vcc"
      : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3])
      : "v"(d0[0]), "v"(d0[1]), "v"(d0[2]), "v"(d0[3]), "v"(d1[0]), "v"(d1[1]), "v"(d1[2]), "v"(d1[3]),
        "v"(thr[0]), "v"(thr[1]), "v"(thr[2]), "v"(thr[3]), "v"(w0), "v"(w1)
      : "vcc");
}

template <bool GUARD, bool UNITS>
__global__ __launch_bounds__(kBlock) void hash_cell_force2_kernel(
    const CellGridView tgv, const CellGridView sgv, int gx, int gy, int gz, long long cell_first,
    long long cell_end, int blocks_per_xcd, float cutoff2, float eps2, float G, float* __restrict__ acc_x,
    float* __restrict__ acc_y, float* __restrict__ acc_z, float4* __restrict__ acc4, int accumulate,
    const int2* __restrict__ units = nullptr, const int* __restrict__ unit_count = nullptr,
    int* __restrict__ unit_count_host = nullptr, int unit_capacity = 0) {
  constexpr int KC = kCellsPerWave;
  constexpr int R = kR2;
  __shared__ float4 win_all[4][kWinCap2 + 64];  // + 64: the far entries behind a batch, and the reach of a lane's last step
  __shared__ double red_all[4][R * 64];
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  float4* win = win_all[w];
  double* red = red_all[w];
  const float4* __restrict__ sorted = sgv.sorted;    // window entries
  const float4* __restrict__ tsorted = tgv.sorted;   // targets
  const int* __restrict__ idx = tgv.idx;
  const long long blk = (long long)(blockIdx.x & 7) * blocks_per_xcd + (blockIdx.x >> 3);  // XCD b mod 8: a contiguous eighth
  const long long cell0 = cell_first + (blk * 4 + w) * KC;
  int grp = 0, grp_end = 0, n_units = 0, n_heavy = 0;
  if constexpr (UNITS) {
    n_heavy = __builtin_amdgcn_readfirstlane(unit_count[0]);
    n_units = n_heavy + __builtin_amdgcn_readfirstlane(unit_count[2]);
    if (unit_count_host && blockIdx.x == 0 && threadIdx.x == 0) {  // chooses and sizes the next launch
      unit_count_host[0] = n_units;
      unit_count_host[1] = unit_count[1];
    }
    grp = (int)blockIdx.x * 4 + w;
    grp_end = (n_units + KC - 1) / KC;
  } else {
    if (cell0 >= cell_end) return;
  }
  for (; !UNITS || grp < grp_end; grp += blocks_per_xcd * 32) {
  // one round of lookups for all KC cells, lane 16 c + r (see hash_cell_force_kernel)
  int vseg0 = 0, vlen = 0;
  {
    const int c = lane >> 4, r = lane & 15;
    long long cell = cell0 + c;
    bool have = c < KC && cell < cell_end;
    int chunk = 0;
    if constexpr (UNITS) {
      const int u = grp * KC + c;
      have = c < KC && u < n_units;
      if (have) {
        const int2 uu = units[u < n_heavy ? u : unit_capacity - 1 - (u - n_heavy)];
        cell = cell_first + uu.x;
        chunk = uu.y;
      }
    }
    if (have) {
      if (r < 9) {
        const unsigned int c32 = (unsigned int)cell, layer = (unsigned int)gx * (unsigned int)gy;
        const unsigned int uz = c32 / layer, rem = c32 - uz * layer, uy = rem / (unsigned int)gx;
        const int cx = (int)(rem - uy * (unsigned int)gx), cy = (int)uy, cz = (int)uz;
        const int yy = cy + (r % 3) - 1, zz = cz + (r / 3) - 1;
        if (yy >= 0 && yy < gy && zz >= 0 && zz < gz) {
          const long long base = ((long long)zz * gy + yy) * gx;
          vseg0 = sgv.lower(base + max(cx - 1, 0));
          vlen = sgv.lower(base + min(cx + 2, gx)) - vseg0;
        }
      } else if (r < 11) {
        vseg0 = tgv.lower(cell + (r - 9));
      } else if (r == 11) {
        vseg0 = chunk;
      }
    }
  }
  int vpre;
  {
    int incl = vlen;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const int up = __shfl_up(incl, off, 16);
      if ((lane & 15) >= off) incl += up;
    }
    vpre = incl - vlen;
  }
#define NBH_SEG0(c, r) __builtin_amdgcn_readlane(vseg0, 16 * (c) + (r))
#define NBH_PRE(c, r) __builtin_amdgcn_readlane(vpre, 16 * (c) + (r))
  float4 pf[9];  // first 64 entries of each run of the NEXT cell to be evaluated
#define NBH_PREFETCH(c)                                                                              \
  _Pragma("unroll") for (int r = 0; r < 9; r++) {                                                    \
    const int p0 = NBH_PRE(c, r), b = min(NBH_PRE(c, r + 1), kWinCap2);                              \
    pf[r] = make_float4(0.f, 0.f, 0.f, 0.f);                                                         \
    if (p0 < b) pf[r] = sorted[NBH_SEG0(c, r) + (min(p0 + lane, b - 1) - p0)];                       \
  }
  NBH_PREFETCH(0)

#pragma unroll
  for (int c = 0; c < KC; c++) {
    int t0 = NBH_SEG0(c, 9), t1 = NBH_SEG0(c, 10);
    if constexpr (UNITS) {  // one chunk of the cell's bodies
      t0 += NBH_SEG0(c, 11) * kUnitChunk2;
      t1 = min(t1, t0 + kUnitChunk2);
    }
    const int Lw = NBH_PRE(c, 9);
    bool prefetched = true;   // the registers pf hold batch 0 of this cell
    bool resident = false;    // LDS holds the whole window (one batch) from an earlier pass
    for (int tb = t0; tb < t1; tb += kPass2) {
      const int cnt = min(t1 - tb, kPass2);
      const int T = (cnt + R - 1) / R;  // target slots: 1..4
      const int S = 64 / T;             // slices: 64, 32, 21, 16
      const int magic = (65536 + T - 1) / T;
      const int sl = (lane * magic) >> 16, slot = lane - sl * T;
      const bool live = sl < S;
      float tx[R], ty[R], tz[R], thr[R];
      double sx[R], sy[R], sz[R];
#pragma unroll
      for (int q = 0; q < R; q++) {
        const int k = slot + q * T;
        const float4 p = tsorted[tb + min(k, cnt - 1)];
        tx[q] = p.x; ty[q] = p.y; tz[q] = p.z;
        thr[q] = (live && k < cnt) ? cutoff2 : -1.0f;  // (no d2 is below -1: a slot without a target collects nothing)
        sx[q] = sy[q] = sz[q] = 0.0;
      }
      for (int vb = 0; vb < Lw; vb += kWinCap2) {
        const int Lb = min(Lw - vb, kWinCap2);
        if (!resident) {
          __builtin_amdgcn_wave_barrier();
          if (prefetched) {
#pragma unroll
            for (int r = 0; r < 9; r++) {
              const int p0 = NBH_PRE(c, r), b = min(NBH_PRE(c, r + 1), Lb);
              if (p0 + lane < b) win[p0 + lane] = pf[r];
              for (int v = p0 + 64 + lane; v < b; v += 64) win[v] = sorted[NBH_SEG0(c, r) + (v - p0)];
            }
            prefetched = false;
            if (c + 1 < KC) { NBH_PREFETCH(c + 1) }  // in flight while this cell is evaluated
          } else {
#pragma unroll
            for (int r = 0; r < 9; r++) {
              const int p0 = NBH_PRE(c, r), a = max(p0, vb), b = min(NBH_PRE(c, r + 1), vb + Lb);
              for (int v = a + lane; v < b; v += 64) win[v - vb] = sorted[NBH_SEG0(c, r) + (v - p0)];
            }
          }
          win[Lb + lane] = make_float4(1.0e30f, 0.f, 0.f, 0.f);  // behind the batch: entries no target accepts (d2 overflows to +inf;
                                                                  // finite, so that a lane without a set bit adds 0 x dx = 0)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          resident = Lw <= kWinCap2;
        }
        // ---- phase 1: the distance test of every candidate, one mask bit per candidate and target
        const int iters = (Lb + S - 1) / S;  // <= 32
        const float4* wp = win + (live ? sl : 0);
        unsigned m[R];
#pragma unroll
        for (int q = 0; q < R; q++) m[q] = 0u;
        // Four entries a round, two and two: the LDS reads of one pair are in flight while the other pair is tested.  Reads
        // past the last step are clamped to it (wave-uniform addresses); their bits are shifted out again below.
        const int last_it = iters - 1;
        const int rounds = (iters + 3) >> 2;
        if (iters > 0) {
          float4 a0 = wp[0], a1 = wp[min(1, last_it) * S];
          for (int r = 0; r < rounds; r++) {
            const int it = 4 * r;
            const float4 b0 = wp[min(it + 2, last_it) * S], b1 = wp[min(it + 3, last_it) * S];
            float d0[R], d1[R];
#pragma unroll
            for (int q = 0; q < R; q++) {
              d0[q] = hash_dist2(a0.x - tx[q], a0.y - ty[q], a0.z - tz[q]);   // :131, unsoftened distance
              d1[q] = hash_dist2(a1.x - tx[q], a1.y - ty[q], a1.z - tz[q]);
            }
            mask_push_lt8(m, d0, d1, thr, a0.w, a1.w);
            a0 = wp[min(it + 4, last_it) * S];
            a1 = wp[min(it + 5, last_it) * S];
#pragma unroll
            for (int q = 0; q < R; q++) {
              d0[q] = hash_dist2(b0.x - tx[q], b0.y - ty[q], b0.z - tz[q]);
              d1[q] = hash_dist2(b1.x - tx[q], b1.y - ty[q], b1.z - tz[q]);
            }
            mask_push_lt8(m, d0, d1, thr, b0.w, b1.w);
          }
        }
        const int extra = 4 * rounds - iters;  // bits of the clamped reads
        // ---- phase 2: the accepted candidates of each target, oldest bit (first window entry) first
        const int sh = 32 - iters;
        const unsigned wbase = live ? (unsigned)sl : 0u;
        const unsigned last = (unsigned)(iters - 1);
#pragma unroll
        for (int q = 0; q < R; q++) {
          unsigned mm = iters > 0 ? ((m[q] >> extra) << sh) : 0u;
          float ax = 0.f, ay = 0.f, az = 0.f;
          // (one step ahead: the entry of the next set bit is on its way from LDS while this one is evaluated)
          bool more = __builtin_amdgcn_ballot_w64(mm != 0u) != 0ull;
          unsigned lz = (unsigned)__builtin_clz(mm | 1u);
          bool valid = mm != 0u;
          float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
          if (more) e = win[wbase + __umul24(min(lz, last), (unsigned)S)];
          mm &= ~(0x80000000u >> lz);
          while (more) {
            more = __builtin_amdgcn_ballot_w64(mm != 0u) != 0ull;
            lz = (unsigned)__builtin_clz(mm | 1u);
            const bool valid2 = mm != 0u;
            float4 e2 = e;
            if (more) e2 = win[wbase + __umul24(min(lz, last), (unsigned)S)];
            mm &= ~(0x80000000u >> lz);
            const float dx = e.x - tx[q];
            float dy = e.y - ty[q];
            asm("" : "+v"(dy));  // (keeps the SLP vectoriser from pairing dy / dz into packed ops that need register moves)
            const float dz = e.z - tz[q];
            const float d2 = hash_dist2(dx, dy, dz);
            const float inv = __builtin_amdgcn_rsqf(d2 + eps2);
            bool ok = valid && (d2 < cutoff2);
            if (GUARD) ok = ok && (d2 > 0.f);  // coincident / self: contributes 0
            float mi = e.w * inv;
            asm("" : "+v"(mi));  // (likewise: no packed multiply of (m inv, inv inv))
            float f = mi * (inv * inv);
            f = ok ? f : 0.f;
            ax = __builtin_fmaf(f, dx, ax);
            ay = __builtin_fmaf(f, dy, ay);
            asm("" : "+v"(ay));  // (and no packed FMA of (ay, az))
            az = __builtin_fmaf(f, dz, az);
            e = e2;
            valid = valid2;
          }
          sx[q] += (double)ax; sy[q] += (double)ay; sz[q] += (double)az;
        }
      }
      // ---- the sum over the slices, one component at a time (red holds R x 64 doubles): lane 4 k + part sums the
      // slices part, part + 4, ... of target k = slot + q T, the four parts are added across the quad, and lane
      // 4 k + comp keeps component comp
      double res = 0.0;
      const int kk = lane >> 2, part = lane & 3;
      const int qk = (kk * magic) >> 16, slotk = kk - qk * T;  // target kk of the pass = (slot slotk, register qk)
#pragma unroll
      for (int comp = 0; comp < 3; comp++) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < R; q++) {
          const double v = comp == 0 ? sx[q] : (comp == 1 ? sy[q] : sz[q]);
          red[q * 64 + lane] = live ? v : 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double v = 0.0;
        if (kk < cnt)
          for (int s2 = part; s2 < S; s2 += 4) v += red[qk * 64 + slotk + s2 * T];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        if (part == comp) res = v;
      }
      if (kk < cnt) {
        const int i = idx[tb + kk];
        const float o = part < 3 ? (float)((double)G * res) : 0.f;
        if (acc4) {
          float* dst = reinterpret_cast<float*>(acc4 + i) + part;
          if (accumulate && part < 3) *dst = *dst + o; else *dst = o;
        } else if (part < 3) {
          (part == 0 ? acc_x : (part == 1 ? acc_y : acc_z))[i] = o;
        }
      }
    }
    if (prefetched && c + 1 < KC) {  // an empty cell (or an empty window): pass the pipeline on
      __builtin_amdgcn_wave_barrier();
      NBH_PREFETCH(c + 1)
    }
  }
    if constexpr (!UNITS) break;
    __builtin_amdgcn_wave_barrier();
  }
#undef NBH_SEG0
#undef NBH_PRE
#undef NBH_PREFETCH
}

// One z layer of a slab grid as a neighbour rank needs it for its boundary pass (sharded path): the layer's bodies in cell
// order and the layer's start array rebased to 0 -- a ready-made CellGridView of that layer, so the receiver bins nothing
// (round 3 built a second grid from the received bodies: six launches and a merge sort per rank and step).
__global__ __launch_bounds__(kBlock) void layer_export_kernel(const float4* __restrict__ sorted, const int* __restrict__ lb_layer,
                                                              int layer_cells, float4* __restrict__ bodies_out,
                                                              int* __restrict__ lb_out) {
  const int first = lb_layer[0], count = lb_layer[layer_cells] - first;
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) bodies_out[i] = sorted[first + i];
  for (int c = blockIdx.x * kBlock + threadIdx.x; c <= layer_cells; c += stride) lb_out[c] = lb_layer[c] - first;
}

// ---------------------------------------------------------------------------------------
// ONE LANE PER BODY (round 4; nbody_hip_grid_tuning 8, and the light cells of a clumped sparse grid).  The wave-per-cell
// kernels pay ~400 instructions of lookups, window staging and slice reduction per cell: the right price for a cell of
// 15 bodies and a window of 400 entries, 10x the pair work for a cell of one or two bodies in a thin neighbourhood --
// and a uniform box that has expanded and clumped (config 5 a few thousand steps in) is 1.1 M occupied cells of which
// 0.9 M hold one to four bodies.  Here a lane takes one body, looks its nine runs up itself and walks them entry by
// entry straight from the cell-ordered list (L1 / L2: neighbouring lanes are neighbouring bodies and read the same
// lines); no LDS, no cross-lane step.  Same pair set, same per-pair arithmetic; fp32 partial sums of <= 32 entries
// folded into fp64, in window order.
// ---------------------------------------------------------------------------------------
template <bool GUARD>
__global__ __launch_bounds__(kBlock) void hash_body_force_kernel(
    const CellGridView tgv, const CellGridView sgv, const unsigned int* __restrict__ tkeys, long long cell_first,
    long long cell_end, int gx, int gy, int gz, float cutoff2, float eps2, float G, float* __restrict__ acc_x,
    float* __restrict__ acc_y, float* __restrict__ acc_z, float4* __restrict__ acc4, int accumulate,
    const int* __restrict__ light, const int* __restrict__ light_count) {
  // the targets: the bodies of the cells [cell_first, cell_end) = one contiguous range of the cell-ordered list (the grid
  // is launched for every body the target grid holds; the threads beyond the range leave) -- or, in the split form, the
  // bodies whose sorted positions cell_units_kernel listed (those of the cells below kSplitCnt bodies)
  const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
  int t;
  // (light_count[-2] = bodies of the most crowded cell if above 64, else 0: cell_units_kernel.  No crowded cell: there is
  // nothing for the wave-per-cell kernel to take, every body is listed... so the list is skipped: the range itself)
  if (light && light_count[-2] != 0) {
    if (i >= *light_count) return;
    t = light[i];
  } else {
    t = tgv.lower(cell_first) + i;
    if (t >= tgv.lower(cell_end)) return;
  }
  const unsigned int c32 = tkeys[t], layer = (unsigned int)gx * (unsigned int)gy;
  const float4 p = tgv.sorted[t];
  const unsigned int uz = c32 / layer, rem = c32 - uz * layer, uy = rem / (unsigned int)gx;
  const int cx = (int)(rem - uy * (unsigned int)gx), cy = (int)uy, cz = (int)uz;
  const float4* __restrict__ sorted = sgv.sorted;
  double sx = 0.0, sy = 0.0, sz = 0.0;
  f2 ax = (f2)(0.f), ay = ax, az = ax;
  int run = 0;  // entries in the current fp32 partial sums
  [[maybe_unused]] f2 nbig, kk;
  if constexpr (!GUARD) {
    const CutConst c = cut_const(cutoff2);
    nbig = (f2)(c.nbig);
    kk = (f2)(c.k);
  }
  for (int r = 0; r < 9; r++) {
    const int yy = cy + (r % 3) - 1, zz = cz + (r / 3) - 1;
    if (yy < 0 || yy >= gy || zz < 0 || zz >= gz) continue;
    const long long base = ((long long)zz * gy + yy) * gx;
    const int k0 = sgv.lower(base + max(cx - 1, 0)), k1 = sgv.lower(base + min(cx + 2, gx));
    if constexpr (GUARD) {
      for (int k = k0; k < k1; k++) {
        const float4 e = sorted[k];
        const float dx = e.x - p.x, dy = e.y - p.y, dz = e.z - p.z;
        const float d2 = hash_dist2(dx, dy, dz);
        const float inv = __builtin_amdgcn_rsqf(d2 + eps2);
        const bool ok = (d2 < cutoff2) && (d2 > 0.f);  // :131, unsoftened distance; coincident / self: contributes 0
        const float f = ok ? (e.w * inv) * (inv * inv) : 0.f;
        ax.x = __builtin_fmaf(f, dx, ax.x);
        ay.x = __builtin_fmaf(f, dy, ay.x);
        az.x = __builtin_fmaf(f, dz, az.x);
        if (++run == 32) {
          sx += (double)ax.x; sy += (double)ay.x; sz += (double)az.x;
          ax = ay = az = (f2)(0.f);
          run = 0;
        }
      }
    } else {
      // four entries a round, two and two in the halves of packed instructions (the compare-free cutoff decision of
      // CellTargets); entries past the end of the run are its last entry again with mass 0.  The fp32 partial sums are
      // folded into fp64 before they would exceed 64 entries (32 per half), checked once per chunk of a run, not per
      // round.
      for (int kc = k0; kc < k1; kc += 32) {
        const int kend = min(kc + 32, k1);
        if (run + (kend - kc) > 64) {
          sx += (double)(ax.x + ax.y); sy += (double)(ay.x + ay.y); sz += (double)(az.x + az.y);
          ax = ay = az = (f2)(0.f);
          run = 0;
        }
        run += kend - kc;
        const int klast = kend - 1;
        for (int k = kc; k < kend; k += 4) {
          // (32-bit byte offsets from the uniform base: one shift per address instead of 64-bit index arithmetic)
          const char* sb = reinterpret_cast<const char*>(sorted);
          const float4 e0 = *reinterpret_cast<const float4*>(sb + ((unsigned)k << 4));
          float4 e1 = *reinterpret_cast<const float4*>(sb + ((unsigned)min(k + 1, klast) << 4));
          float4 e2 = *reinterpret_cast<const float4*>(sb + ((unsigned)min(k + 2, klast) << 4));
          float4 e3 = *reinterpret_cast<const float4*>(sb + ((unsigned)min(k + 3, klast) << 4));
          if (k + 1 > klast) e1.w = 0.f;
          if (k + 2 > klast) e2.w = 0.f;
          if (k + 3 > klast) e3.w = 0.f;
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const float4 ea = h ? e2 : e0, eb = h ? e3 : e1;
            const f2 dx = f2{ea.x, eb.x} - (f2)(p.x), dy = f2{ea.y, eb.y} - (f2)(p.y), dz = f2{ea.z, eb.z} - (f2)(p.z);
            const f2 d2 = hash_dist2(dx, dy, dz);
            const f2 de = d2 + (f2)(eps2);
            f2 inv;
            inv.x = __builtin_amdgcn_rsqf(de.x);
            inv.y = __builtin_amdgcn_rsqf(de.y);
            f2 tt;  // 1 where d2 < cutoff^2, else 0
            asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(tt) : "v"(d2), "v"(nbig), "v"(kk));
            const f2 f = (inv * (tt * f2{ea.w, eb.w})) * (inv * inv);
            ax = __builtin_elementwise_fma(f, dx, ax);
            ay = __builtin_elementwise_fma(f, dy, ay);
            az = __builtin_elementwise_fma(f, dz, az);
          }
        }
      }
    }
  }
  sx += (double)(ax.x + ax.y); sy += (double)(ay.x + ay.y); sz += (double)(az.x + az.y);
  const int o = tgv.idx[t];
  const float ox = (float)((double)G * sx), oy = (float)((double)G * sy), oz = (float)((double)G * sz);
  if (acc4) {
    if (accumulate) {
      const float4 q = acc4[o];
      acc4[o] = make_float4(q.x + ox, q.y + oy, q.z + oz, 0.f);
    } else {
      acc4[o] = make_float4(ox, oy, oz, 0.f);
    }
  } else {
    acc_x[o] = ox; acc_y[o] = oy; acc_z[o] = oz;
  }
}

// z cell coordinate of every body on a given grid (slab assignment of the sharded path)
// ---------------------------------------------------------------------------------------
// TWO BODIES OF ONE CELL PER LANE (round 4, tuning 10).  The lane-per-body kernel above spends 16.75 VALU instructions
// per candidate pair: it packs two window ENTRIES into the halves of its packed instructions, which costs a register
// transposition (13 moves per four entries) and a clamped address per entry.  Two TARGETS of the same cell have the same
// window, so here a lane takes two consecutive bodies of a cell (the list of cell_units_kernel with pair_items: an odd
// cell's last body alone, its twin a duplicate whose result is dropped) and every entry it loads serves both -- the
// arithmetic of CellTargets<false, 2>: 17 instructions per entry and TWO targets, half the loads, no transposition.
// No LDS, no cross-lane step; fp32 partial sums of at most 64 entries folded into fp64 in window order.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void hash_body2_force_kernel(
    const CellGridView tgv, const CellGridView sgv, const unsigned int* __restrict__ tkeys, int gx, int gy, int gz,
    float cutoff2, float eps2, float G, float* __restrict__ acc_x, float* __restrict__ acc_y, float* __restrict__ acc_z,
    float4* __restrict__ acc4, int accumulate, const int* __restrict__ items, const int* __restrict__ item_count) {
  const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
  if (i >= *item_count) return;
  const int item = items[i];
  const int t0 = item & 0x7fffffff;
  const bool single = item < 0;
  const int t1 = single ? t0 : t0 + 1;
  const unsigned int c32 = tkeys[t0], layer = (unsigned int)gx * (unsigned int)gy;
  const float4 p0 = tgv.sorted[t0], p1 = tgv.sorted[t1];
  const unsigned int uz = c32 / layer, rem = c32 - uz * layer, uy = rem / (unsigned int)gx;
  const int cx = (int)(rem - uy * (unsigned int)gx), cy = (int)uy, cz = (int)uz;
  // the nine runs' bounds first: eighteen independent loads in flight, not nine round trips one after the other
  int k0[9], k1[9];
#pragma unroll
  for (int r = 0; r < 9; r++) {
    const int yy = cy + (r % 3) - 1, zz = cz + (r / 3) - 1;
    k0[r] = k1[r] = 0;
    if (yy >= 0 && yy < gy && zz >= 0 && zz < gz) {
      const long long base = ((long long)zz * gy + yy) * gx;
      k0[r] = sgv.lower(base + max(cx - 1, 0));
      k1[r] = sgv.lower(base + min(cx + 2, gx));
    }
  }
  CellTargets<false, 2> tg;
  tg.set_cut(cutoff2);
  tg.set(0, p0.x, p0.y, p0.z);
  tg.set(1, p1.x, p1.y, p1.z);
  tg.clear();
  double sx[2] = {0.0, 0.0}, sy[2] = {0.0, 0.0}, sz[2] = {0.0, 0.0};
  int run = 0;  // entries in the current fp32 partial sums
  const char* sb = reinterpret_cast<const char*>(sgv.sorted);
#pragma unroll
  for (int r = 0; r < 9; r++) {
    for (int kc = k0[r]; kc < k1[r]; kc += 64) {
      const int kend = min(kc + 64, k1[r]);
      if (run + (kend - kc) > 64) {
#pragma unroll
        for (int q = 0; q < 2; q++) { sx[q] += (double)tg.get(q, 0); sy[q] += (double)tg.get(q, 1); sz[q] += (double)tg.get(q, 2); }
        tg.clear();
        run = 0;
      }
      run += kend - kc;
      const int klast = kend - 1;
      for (int k = kc; k < kend; k += 4) {
        const float4 e0 = *reinterpret_cast<const float4*>(sb + ((unsigned)k << 4));
        float4 e1 = *reinterpret_cast<const float4*>(sb + ((unsigned)min(k + 1, klast) << 4));
        float4 e2 = *reinterpret_cast<const float4*>(sb + ((unsigned)min(k + 2, klast) << 4));
        float4 e3 = *reinterpret_cast<const float4*>(sb + ((unsigned)min(k + 3, klast) << 4));
        if (k + 1 > klast) e1.w = 0.f;  // (past the end of the run: its last entry again, with mass 0)
        if (k + 2 > klast) e2.w = 0.f;
        if (k + 3 > klast) e3.w = 0.f;
        tg.pair4(e0, e1, e2, e3, eps2);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; q++) { sx[q] += (double)tg.get(q, 0); sy[q] += (double)tg.get(q, 1); sz[q] += (double)tg.get(q, 2); }
#pragma unroll
  for (int q = 0; q < 2; q++) {
    if (q == 1 && single) break;
    const int o = tgv.idx[q ? t1 : t0];
    const float ox = (float)((double)G * sx[q]), oy = (float)((double)G * sy[q]), oz = (float)((double)G * sz[q]);
    if (acc4) {
      if (accumulate) {
        const float4 a = acc4[o];
        acc4[o] = make_float4(a.x + ox, a.y + oy, a.z + oz, 0.f);
      } else {
        acc4[o] = make_float4(ox, oy, oz, 0.f);
      }
    } else {
      acc_x[o] = ox; acc_y[o] = oy; acc_z[o] = oz;
    }
  }
}

__global__ __launch_bounds__(kBlock) void cell_z_kernel(const float4* __restrict__ posm, int n,
                                                        float lo_z, float cell, int gz,
                                                        int* __restrict__ cz) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) cz[i] = cell_coord(posm[i].z, lo_z, cell, gz);
}

__global__ void bbox_decode_kernel(const unsigned int* __restrict__ enc, float* __restrict__ out) {
  if (threadIdx.x < 6) out[threadIdx.x] = ordered_to_float(enc[threadIdx.x]);
}
// ... leaving the empty box behind for the next pass (enc private to the caller: no bbox_init_kernel per step)
__global__ void bbox_decode_rearm_kernel(unsigned int* __restrict__ enc, float* __restrict__ out) {
  if (threadIdx.x < 6) {
    out[threadIdx.x] = ordered_to_float(enc[threadIdx.x]);
    enc[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
  }
}

}  // namespace nbh

using namespace nbh;

// stable radix sort of the cell ids carrying (body, original index) along; temp == nullptr: size query
// Which sort bins the bodies (above the crossover size; below it rocPRIM's public sort = a merge sort there):
//   kSortDriver  rocPRIM's Onesweep device functions under the driver of onesweep.h (fenced: version + self-test) -- the
//                default while the fence holds: 172 us for the two passes of config 5;
//   kSortOwn     the hand-written sort of radix_sort.h (no rocPRIM): 207 us -- what runs when the fence does not hold, or
//                with NBH_SORT=own;
//   kSortPublic  rocprim::radix_sort_pairs (public API): NBH_SORT=public, and every size below the crossover.
enum SortImpl { kSortPublic = 0, kSortDriver = 1, kSortOwn = 2 };
static SortImpl sort_impl_from_env() {
  const char* e = std::getenv("NBH_SORT");
  if (e && std::strcmp(e, "own") == 0) return kSortOwn;
  if (e && std::strcmp(e, "public") == 0) return kSortPublic;
  if (e && std::strcmp(e, "driver") == 0 && NBH_HASH_OWN_SORT && NBH_ONESWEEP_AVAILABLE) return kSortDriver;
  return (NBH_HASH_OWN_SORT && nbh::onesweep::usable()) ? kSortDriver : kSortOwn;
}

// stable radix sort of the cell ids carrying (body, original index) along; temp == nullptr: size query (room for every path)
static hipError_t sort_bodies_by_cell(void* temp, size_t& temp_bytes, unsigned int* keys_in, unsigned int* keys_out,
                                      const float4* bodies_in, float4* bodies_out, int* idx_out, size_t n, int bits,
                                      hipStream_t st, SortImpl impl = kSortPublic, bool cleared = false,
                                      unsigned int* hist = nullptr, unsigned int* error_host = nullptr) {
  auto vin = rocprim::make_zip_iterator(rocprim::make_tuple(bodies_in, rocprim::make_counting_iterator<int>(0)));
  auto vout = rocprim::make_zip_iterator(rocprim::make_tuple(bodies_out, idx_out));
  if (!temp) {
    size_t a = 0, b = 0, c = 0;
    hipError_t e = rocprim::radix_sort_pairs<SortConfig>(nullptr, a, keys_in, keys_out, vin, vout, n, 0, bits, st);
    if (e != hipSuccess) return e;
#if NBH_HASH_OWN_SORT && NBH_ONESWEEP_AVAILABLE
    e = nbh::onesweep::sort_pairs<NBH_HASH_RADIX_BITS>(nullptr, b, static_cast<const unsigned int*>(keys_in), keys_out, vin, vout,
                                                       n, 0u, (unsigned)bits, st);
    if (e != hipSuccess) return e;
#endif
    e = nbh::radix::sort_pairs<unsigned int, true>(nullptr, c, static_cast<const unsigned int*>(keys_in), keys_out, bodies_in,
                                                   bodies_out, nullptr, idx_out, n, 0u, (unsigned)bits, st, nullptr);
    temp_bytes = std::max(a, std::max(b, c));
    return e;
  }
#if NBH_HASH_OWN_SORT && NBH_ONESWEEP_AVAILABLE
  if (impl == kSortDriver)
    return nbh::onesweep::sort_pairs<NBH_HASH_RADIX_BITS>(temp, temp_bytes, static_cast<const unsigned int*>(keys_in), keys_out,
                                                          vin, vout, n, 0u, (unsigned)bits, st, cleared, hist, nbh::kHistCopies, nbh::kHistWords);
#endif
  if (impl == kSortOwn || impl == kSortDriver)  // (a driver request without the driver compiled in: the hand-written sort)
    return nbh::radix::sort_pairs<unsigned int, true>(temp, temp_bytes, static_cast<const unsigned int*>(keys_in), keys_out,
                                                      bodies_in, bodies_out, nullptr, idx_out, n, 0u, (unsigned)bits, st, error_host,
                                                      hist, hist ? nbh::kHistCopies : 1, hist ? (unsigned)nbh::kHistWords : 0u);
  return rocprim::radix_sort_pairs<SortConfig>(temp, temp_bytes, keys_in, keys_out, vin, vout, n, 0, bits, st);
}

// Run-time half of the dependency fence of onesweep.h: ONE buffer (200,000 bodies on a 20-bit grid, every key value
// shared by several bodies so that stability shows) through the Onesweep driver and through the public
// rocprim::radix_sort_pairs; keys, bodies and indices must agree word for word.  Once per process (the first grid).
static void grid_sort_self_test(hipStream_t st) {
  static std::atomic<bool> done{false};
  if (done.exchange(true)) return;
  const size_t n = 200000;
  const int bits = 20;
  std::vector<unsigned int> hk(n);
  std::vector<float4> hb(n);
  unsigned int x = 12345u;
  for (size_t i = 0; i < n; i++) {
    x = x * 1664525u + 1013904223u;
    hk[i] = (x >> 9) & ((1u << bits) - 1u) & ~0x3fu;  // 2^14 distinct cells: ~12 bodies each
    hb[i] = make_float4((float)i, (float)hk[i], 0.f, 1.f);
  }
  constexpr int kV = 3;  // 0: the driver, 1: the public sort, 2: the hand-written sort
  const SortImpl impl[kV] = {kSortDriver, kSortPublic, kSortOwn};
  const bool have[kV] = {NBH_HASH_OWN_SORT && NBH_ONESWEEP_AVAILABLE, true, true};
  unsigned int *k_in = nullptr, *k_out[kV] = {nullptr, nullptr, nullptr};
  float4 *b_in = nullptr, *b_out[kV] = {nullptr, nullptr, nullptr};
  int* i_out[kV] = {nullptr, nullptr, nullptr};
  unsigned int *err_h = nullptr, *err_d = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  bool ran = false, same[kV] = {false, true, false};
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&k_in), n * sizeof(unsigned int));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&b_in), n * sizeof(float4));
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&err_h), 64, hipHostMallocMapped);
  if (e == hipSuccess) {
    *err_h = 0u;
    e = hipHostGetDevicePointer(reinterpret_cast<void**>(&err_d), err_h, 0);
  }
  for (int v = 0; v < kV && e == hipSuccess; v++) {
    e = hipMalloc(reinterpret_cast<void**>(&k_out[v]), n * sizeof(unsigned int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&b_out[v]), n * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&i_out[v]), n * sizeof(int));
  }
  if (e == hipSuccess) e = sort_bodies_by_cell(nullptr, tmp_bytes, k_in, k_out[0], b_in, b_out[0], i_out[0], n, bits, st);
  if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 16);
  if (e == hipSuccess) e = hipMemcpyAsync(k_in, hk.data(), n * sizeof(unsigned int), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(b_in, hb.data(), n * sizeof(float4), hipMemcpyHostToDevice, st);
  for (int v = 0; v < kV && e == hipSuccess; v++) {
    if (!have[v]) continue;
    size_t tb = tmp_bytes;
    e = sort_bodies_by_cell(tmp, tb, k_in, k_out[v], b_in, b_out[v], i_out[v], n, bits, st, impl[v], false, nullptr, err_d);
  }
  if (e == hipSuccess) {
    std::vector<unsigned int> rk[kV];
    std::vector<float4> rb[kV];
    std::vector<int> ri[kV];
    for (int v = 0; v < kV && e == hipSuccess; v++) {
      if (!have[v]) continue;
      rk[v].resize(n); rb[v].resize(n); ri[v].resize(n);
      e = hipMemcpyAsync(rk[v].data(), k_out[v], n * sizeof(unsigned int), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipMemcpyAsync(rb[v].data(), b_out[v], n * sizeof(float4), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipMemcpyAsync(ri[v].data(), i_out[v], n * sizeof(int), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) {
      ran = true;
      bool sorted = true;
      for (size_t i = 1; sorted && i < n; i++) sorted = rk[1][i - 1] <= rk[1][i];
      for (int v = 0; v < kV; v += 2)
        same[v] = have[v] && sorted && std::memcmp(rk[v].data(), rk[1].data(), n * sizeof(unsigned int)) == 0 &&
                  std::memcmp(rb[v].data(), rb[1].data(), n * sizeof(float4)) == 0 &&
                  std::memcmp(ri[v].data(), ri[1].data(), n * sizeof(int)) == 0;
      if (*err_h) same[2] = false;  // (a look-back of the hand-written sort gave up)
    }
  }
  (void)hipGetLastError();
  (void)hipFree(k_in); (void)hipFree(b_in); (void)hipFree(tmp);
  if (err_h) (void)hipHostFree(err_h);
  for (int v = 0; v < kV; v++) { (void)hipFree(k_out[v]); (void)hipFree(b_out[v]); (void)hipFree(i_out[v]); }
  if (have[0]) nbh::onesweep::self_test_report(ran && same[0], "spatial-hash (32-bit keys, body + index payload)");
  nbh::radix::self_test_report(ran && same[2], "spatial-hash (32-bit keys, body + index payload)");
}

extern "C" int nbody_hip_sort_info(int* driver_compiled, int* self_test, int* own_self_test, int* rocprim_version) {
  if (driver_compiled) *driver_compiled = NBH_ONESWEEP_AVAILABLE;
  if (self_test) *self_test = nbh::onesweep::self_test_state().load(std::memory_order_acquire);
  if (own_self_test) *own_self_test = nbh::radix::self_test_state().load(std::memory_order_acquire);
  if (rocprim_version) *rocprim_version = (int)ROCPRIM_VERSION;
  return NBODY_HIP_OK;
}

struct nbody_hip_grid {
  nbody_hip_ctx* ctx = nullptr;
  size_t max_particles = 0;
  float cell_size = 1.0f;
  // device
  unsigned int* d_enc = nullptr;       // 6 ordered-int bbox words
  bool enc_armed = false;              // d_enc holds the empty box (left by the previous build's grid_info_kernel)
  GridInfo* d_info = nullptr;
  int info_seq = 0;                    // sequence number of the last grid_info_kernel (polled in h_info->pad)
  int last_sort_bits = 0;              // key bits of the previous build: the key pass is launched on them before the grid
                                       // record is back (0: no speculation); spec_mode: NBH_HASH_SPECULATE=0 off, 2 always wrong (test)
  int spec_mode = 1;
  GridInfo* h_info = nullptr;          // pinned
  GridInfo* h_info_dev = nullptr;      // the device's address of h_info (null: not mapped, copy instead)
  unsigned int *d_keys_a = nullptr, *d_keys_b = nullptr;
  int* d_idx_b = nullptr;              // cell order -> original body index
  float4* d_sorted = nullptr;
  void* d_sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  unsigned int* d_hist = nullptr;      // digit counts of the sort, accumulated by assign_cells_kernel (kHistWords)
  size_t own_sort_from = nbh::kOwnSortFromGrid;  // radix sort of our own (driver / hand-written) from this many bodies
  SortImpl sort_impl = kSortPublic;    // which one (sort_impl_from_env, at creation, after the self-tests)
  unsigned int* h_sort_err = nullptr;  // mapped host word the hand-written sort raises when a look-back gives up
  unsigned int* h_sort_err_dev = nullptr;
  int *d_cell_start = nullptr, *d_cell_end = nullptr;  // lazily sized (inspection API only)
  long long cell_capacity = 0;
  int* d_cell_lb = nullptr;            // first sorted position of every cell (+ 1 entry); dense grids only
  // the unit form of the wave-per-cell kernel (cell_units_kernel): work list, its two alternating counters, and
  // the previous lists' lengths in mapped host memory (they size the next launch; a stale value costs time, not results)
  int2* d_units = nullptr;
  size_t units_cap = 0;
  int* d_light = nullptr;              // split form: sorted positions of the bodies of the light cells (max_particles ints)
  int* d_unit_count = nullptr;         // [2][4]: {heavy units, bodies of the most crowded cell, other units, -}, alternating
  int* h_unit_hint = nullptr;          // pinned [2][4]: {units, most crowded cell, sequence word, -} for whole-range calls /
                                       // layer-range calls
  int* h_unit_hint_dev = nullptr;
  unsigned unit_flip = 0;
  unsigned stat_tick = 0;
  int stat_seq = 0;
  int split_cnt = kSplitCnt;           // split form: cells from this many bodies take the wave-per-cell kernel (NBH_HASH_SPLIT_CNT
                                       // in the environment at creation: A/B runs)
  double filter_from_inside = kFilterFromInside;  // bodies per cell from which the filtered form runs when cutoff <= cell
                                       // (NBH_HASH_FILTER_FROM in the environment at creation: A/B runs)
  int use_units = 1;                   // NBH_HASH_UNITS in the environment at creation: 0 = never (the cell-range form,
                                       // A/B), 2 = always (tests), default 1 = by the statistics of the previous call
  long long lb_capacity = 0;
  unsigned gap_tick = 0;          // which of the two gap-list counters this build uses (cell_mark_kernel)
  bool gap_counters_clean = false;  // the two counters behind the start array are zero or in the alternating protocol
  bool lb_by_position = true;     // NBH_HASH_LB=cell: the per-cell search (cell_lb_kernel) instead
  bool lb_valid = false;
  long long lb_base = 0, lb_count = 0;  // cells [lb_base, lb_base + lb_count] covered by d_cell_lb
  int slab_z0 = 0, slab_nz = 0;         // packed builds: z layers this grid holds (nz <= 0: all)
  int tune_kernel = 0;                 // 0 automatic, 1 cell-run kernel, 2 / 3 / 4 wave-per-cell kernel with R = 1 / 2 / 4,
                                       // 5 = half-shell TIMING PROBE (not forces), 6 = filtered form, 7 = two-phase form, 8 = one lane per body, 9 = split: one lane per body / wave per crowded cell
  bool ranges_valid = false;
  // host mirror of the last build
  GridInfo info{};
  size_t built_count = 0;
};

static void grid_release(nbody_hip_grid* g) {
  if (!g) return;
  (void)hipFree(g->d_enc); (void)hipFree(g->d_info); (void)hipFree(g->d_keys_a);
  (void)hipFree(g->d_keys_b); (void)hipFree(g->d_idx_b);
  (void)hipFree(g->d_sorted); (void)hipFree(g->d_sort_tmp); (void)hipFree(g->d_hist); (void)hipFree(g->d_cell_start);
  (void)hipFree(g->d_cell_end); (void)hipFree(g->d_cell_lb); (void)hipFree(g->d_units); (void)hipFree(g->d_unit_count);
  (void)hipFree(g->d_light);
  if (g->h_unit_hint) (void)hipHostFree(g->h_unit_hint);
  if (g->h_sort_err) (void)hipHostFree(g->h_sort_err);
  if (g->h_info) (void)hipHostFree(g->h_info);
  delete g;
}

extern "C" int nbody_hip_grid_create(nbody_hip_ctx* ctx, size_t max_particles, float cell_size,
                                     nbody_hip_grid** out) {
  if (!ctx || !out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  *out = nullptr;
  if (max_particles == 0 || max_particles > 0x3fffffffu)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "max_particles out of range");
  if (!(cell_size > 0.0f) || !(cell_size < INFINITY))
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Spatial hash cell size must be positive and finite");
  NBH_HIP(hipSetDevice(ctx->device));
  nbody_hip_grid* g = new nbody_hip_grid();
  g->ctx = ctx;
  g->max_particles = max_particles;
  g->own_sort_from = nbh::own_sort_from(nbh::kOwnSortFromGrid);
  grid_sort_self_test(ctx->stream);  // (once per process: the driver and the hand-written sort against the public sort)
  g->sort_impl = sort_impl_from_env();
  if (hipHostMalloc(reinterpret_cast<void**>(&g->h_sort_err), 64, hipHostMallocMapped) == hipSuccess) {
    *g->h_sort_err = 0u;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&g->h_sort_err_dev), g->h_sort_err, 0) != hipSuccess) g->h_sort_err_dev = nullptr;
  } else {
    g->h_sort_err = nullptr;
  }
  (void)hipGetLastError();
  g->cell_size = cell_size;
  const size_t n = max_particles;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&g->d_enc), 8 * sizeof(unsigned int));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g->d_info), sizeof(GridInfo));
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&g->h_info), sizeof(GridInfo), hipHostMallocMapped);
  if (e == hipSuccess) std::memset(g->h_info, 0, sizeof(GridInfo));
  if (e == hipSuccess && hipHostGetDevicePointer(reinterpret_cast<void**>(&g->h_info_dev), g->h_info, 0) != hipSuccess) {
    g->h_info_dev = nullptr;
    (void)hipGetLastError();
  }
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g->d_keys_a), n * sizeof(unsigned int));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g->d_keys_b), n * sizeof(unsigned int));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g->d_idx_b), n * sizeof(int));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g->d_sorted), n * sizeof(float4));
  if (e == hipSuccess) {
    size_t tmp = 0;
    e = sort_bodies_by_cell(nullptr, tmp, g->d_keys_a, g->d_keys_b, g->d_sorted, g->d_sorted, g->d_idx_b, n, 32, ctx->stream);
    if (e == hipSuccess) {
      g->sort_tmp_bytes = tmp;
      e = hipMalloc(&g->d_sort_tmp, tmp > 0 ? tmp : 16);
      if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g->d_hist), (size_t)kHistCopies * kHistWords * sizeof(unsigned int));
      if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g->d_unit_count), 8 * sizeof(int));
      if (e == hipSuccess) e = hipMemset(g->d_unit_count, 0, 8 * sizeof(int));
      if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&g->h_unit_hint), 8 * sizeof(int), hipHostMallocMapped);
      if (e == hipSuccess) {
        for (int k = 0; k < 8; k++) g->h_unit_hint[k] = 0;
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&g->h_unit_hint_dev), g->h_unit_hint, 0) != hipSuccess) {
          (void)hipGetLastError();
          g->h_unit_hint_dev = nullptr;
        }
        const char* env = std::getenv("NBH_HASH_UNITS");
        g->use_units = env && env[0] == '0' ? 0 : (env && env[0] == '2' ? 2 : 1);
        if (const char* sc = std::getenv("NBH_HASH_SPLIT_CNT")) {
          const int v = std::atoi(sc);
          if (v >= 2) g->split_cnt = v;
        }
        if (const char* ff = std::getenv("NBH_HASH_FILTER_FROM")) {
          const double v = std::atof(ff);
          if (v > 0.0) g->filter_from_inside = v;
        }
        if (const char* lbm = std::getenv("NBH_HASH_LB")) g->lb_by_position = !(lbm[0] == 'c');  // "cell": A/B switch
        if (const char* sp = std::getenv("NBH_HASH_SPECULATE")) g->spec_mode = std::atoi(sp);  // 0 off, 2 always wrong (tests)
      }
    }
  }
  if (e != hipSuccess) {
    grid_release(g);
    return NBH_FAIL(e == hipErrorOutOfMemory ? NBODY_HIP_ERR_RESOURCE : NBODY_HIP_ERR_DEVICE,
                    "spatial hash grid allocation: %s", hipGetErrorString(e));
  }
  *out = g;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_grid_destroy(nbody_hip_grid* g) {
  if (!g) return NBODY_HIP_OK;
  NBH_DESTROY_BEGIN
  (void)hipSetDevice(g->ctx->device);
  (void)hipStreamSynchronize(g->ctx->stream);
  grid_release(g);
  NBH_DESTROY_END
}

extern "C" int nbody_hip_grid_set_cell_size(nbody_hip_grid* g, float cell_size) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null grid");
  if (!(cell_size > 0.0f) || !(cell_size < INFINITY))
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Spatial hash cell size must be positive and finite");
  g->cell_size = cell_size;
  return NBODY_HIP_OK;
}
extern "C" int nbody_hip_grid_tuning(nbody_hip_grid* g, int kernel) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null grid");
  if (kernel < 0 || kernel > 10) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "kernel must be 0..10");
  g->tune_kernel = kernel;
  return NBODY_HIP_OK;
}

static int bits_for(long long total) {
  int b = 1;
  while ((1LL << b) < total && b < 32) b++;
  return b;
}

// bounds == nullptr: bounding box of the bodies padded by 0.001 (the reference's build);
// otherwise {lo x,y,z, hi x,y,z} is used as the (already padded) box -- the sharded path passes
// the GLOBAL box so that every rank bins on the same grid.
// soa != nullptr: posm is a scratch array to be filled from the SoA bodies (fused with the bounding box)
// drift_dt: soa is a step's state BEFORE its drift; the drift rides on the packing pass (nbody_hip_grid_drift_build)
static int grid_build_packed(nbody_hip_grid* g, float4* posm, size_t n, const float* bounds,
                             const nbody_particle_data* soa = nullptr, bool slab = false, const float* drift_dt = nullptr) {
  nbody_hip_ctx* ctx = g->ctx;
  // the grid dimensions come back to the host every build (they size the force launch)
  NBH_NOT_CAPTURABLE(ctx, "the spatial-hash grid build");
  hipStream_t st = ctx->stream;
  const int ni = (int)n;
  // which sort (SortImpl): above the crossover the driver of rocPRIM's Onesweep kernels while its fence holds, else the
  // hand-written sort; below it (and with NBH_SORT=public) the public rocPRIM sort.  Both of the first two take the digit
  // counts from assign_cells_kernel; only the driver also wants its look-back block cleared there.
  auto pick_impl = [&]() {
    SortImpl impl = n >= g->own_sort_from ? g->sort_impl : kSortPublic;
    if (impl == kSortDriver && !(NBH_HASH_OWN_SORT && nbh::onesweep::usable())) impl = kSortOwn;
    if (impl == kSortOwn && (!nbh::radix::usable() || !g->h_sort_err_dev)) impl = kSortPublic;
    return impl;
  };
  auto places_for = [&](SortImpl impl, int bits) {
    return impl != kSortPublic && bits <= kHistPlaces * NBH_HASH_RADIX_BITS ? (bits + NBH_HASH_RADIX_BITS - 1) / NBH_HASH_RADIX_BITS : 0;
  };
  // the key pass (assign_cells_kernel) and the sort for a grid of `bits` key bits:
  // (keys, bodies, indices) -> cell order: d_keys_b, d_sorted, d_idx_b
  auto key_pass = [&](int bits) -> hipError_t {
    const SortImpl impl = pick_impl();
    const bool driver = impl == kSortDriver;
    const size_t zero_words = driver ? onesweep::clear_words<NBH_HASH_RADIX_BITS>(n, 0u, (unsigned)bits) : 0;
    const int hist_places = places_for(impl, bits);
    hipLaunchKernelGGL(assign_cells_kernel, dim3(std::min((ni + kHistThreads - 1) / kHistThreads, NBH_HIST_BLOCKS)), dim3(kHistThreads), 0, st, posm, ni, g->d_info,
                       g->cell_size, g->d_keys_a, static_cast<unsigned int*>(g->d_sort_tmp), (unsigned int)zero_words,
                       g->d_hist, hist_places);
    size_t tmp = g->sort_tmp_bytes;
    return sort_bodies_by_cell(g->d_sort_tmp, tmp, g->d_keys_a, g->d_keys_b, posm, g->d_sorted, g->d_idx_b, n, bits, st,
                               impl, /*cleared=*/driver, hist_places ? g->d_hist : nullptr, g->h_sort_err_dev);
  };
  if (g->h_sort_err && *g->h_sort_err)
    return NBH_FAIL(NBODY_HIP_ERR_DEVICE, "the radix sort of an earlier build gave up in its look-back (csrc/radix_sort.h)");
  int spec_bits = 0;  // the bit count the key pass has already been launched on (0: not yet)
  if (bounds) {
    GridInfo gi{};
    long long total = 1;
    for (int a = 0; a < 3; a++) {
      gi.bmin[a] = bounds[a];
      gi.bmax[a] = bounds[3 + a];
      gi.dims[a] = grid_axis_cells(bounds[a], bounds[3 + a], g->cell_size);  // :244-246
      total = grid_cells_times(total, gi.dims[a]);
    }
    gi.total = total;
    g->info = gi;
    if (g->info.total > 100000000LL)
      return NBH_FAIL(NBODY_HIP_ERR_RESOURCE, "Spatial hash grid too large: reduce cell_size or bounding box");
    hipLaunchKernelGGL(grid_info_set_kernel, dim3(1), dim3(256), 0, st, gi, g->d_info, g->d_hist);
    NBH_LAUNCH_CHECK();
  } else {
    const bool armed = g->enc_armed;  // false on the first build and after one that failed before grid_info_kernel
    g->enc_armed = false;
    if (soa && drift_dt) {
      if (int rc = launch_drift_pack_bbox(ctx, const_cast<nbody_particle_data*>(soa), *drift_dt, posm, g->d_enc, !armed)) return rc;
    } else if (soa) {
      if (int rc = launch_pack_bbox(ctx, soa->pos_x, soa->pos_y, soa->pos_z, soa->mass, ni, posm, g->d_enc, !armed)) return rc;
    } else {
      if (int rc = launch_bbox(ctx, posm, ni, g->d_enc, !armed)) return rc;
    }
    hipLaunchKernelGGL(grid_info_kernel, dim3(1), dim3(256), 0, st, g->d_enc, g->cell_size, 0.001f, g->d_info,
                       g->h_info_dev, g->d_hist, ++g->info_seq);
    NBH_LAUNCH_CHECK();
    g->enc_armed = true;
    // The key pass and the sort do not wait for the host: what they need of the grid they read from the device's record, and the
    // launch parameters that depend on the grid (how many digit places to count, how many look-back words to clear)
    // depend on it only through the number of key bits -- which changes when the cell count crosses a power of two.  So
    // they are launched on the PREVIOUS build's bit count before the record is polled, and run while the host waits; if
    // the count turns out different (or the grid too large), the digit counts are zeroed and both repeated with the
    // right one.  Removes the 9-11 us the GPU idled between grid_info_kernel and the key pass.
    if (g->spec_mode && g->last_sort_bits > 0) {
      spec_bits = g->spec_mode == 2 ? (g->last_sort_bits > 10 ? g->last_sort_bits - 10 : g->last_sort_bits + 10) : g->last_sort_bits;
      NBH_HIP(key_pass(spec_bits));
    }
    // the one host round trip of the build: the grid size decides validity (and, for the
    // inspection API, allocation).  ref: 6 scalar cudaMemcpy D2H, force_spatial_hash.cu:213-218
    if (!g->h_info_dev) NBH_HIP(hipMemcpyAsync(g->h_info, g->d_info, sizeof(GridInfo), hipMemcpyDeviceToHost, st));
    if (g->h_info_dev) {
      // poll the record's sequence word in mapped host memory: the kernel's store is visible a few microseconds after it
      // retires, hipStreamSynchronize returned 20-30 us later (rocprofv3: the gap before assign_cells_kernel).  Bounded:
      // after 2 ms the stream is synchronised the ordinary way (which also surfaces a fault).
      const volatile int* seq = &g->h_info->pad;
      const auto t0 = std::chrono::steady_clock::now();
      unsigned spins = 0;
      while (*seq != g->info_seq) {
        if ((++spins & 1023u) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2e-3)
          break;
      }
      if (*seq != g->info_seq) NBH_HIP(hipStreamSynchronize(st));
      std::atomic_thread_fence(std::memory_order_acquire);
    } else {
      NBH_HIP(hipStreamSynchronize(st));
    }
    g->info = *g->h_info;
    if (g->info.total > 100000000LL)  // :252-254
      return NBH_FAIL(NBODY_HIP_ERR_RESOURCE, "Spatial hash grid too large: reduce cell_size or bounding box");
  }
  const int sort_bits = bits_for(g->info.total);
  if (spec_bits != sort_bits) {
    if (spec_bits) {
      hipLaunchKernelGGL(hist_zero_kernel, dim3(1), dim3(256), 0, st, g->d_hist);
      NBH_LAUNCH_CHECK();
    }
    NBH_HIP(key_pass(sort_bits));
  }
  g->last_sort_bits = sort_bits;
  g->lb_valid = false;
  {
    // cells the per-cell start array covers: the whole grid, or the z layers of this rank's slab
    long long base = 0, count = g->info.total;
    if (slab && g->slab_nz > 0) {
      const long long layer = (long long)g->info.dims[0] * g->info.dims[1];
      const long long z0 = g->slab_z0 < 0 ? 0 : g->slab_z0;
      long long z1 = (long long)g->slab_z0 + g->slab_nz;
      if (z1 > g->info.dims[2]) z1 = g->info.dims[2];
      base = z0 * layer;
      count = z1 > z0 ? (z1 - z0) * layer : 0;
    }
    // (16 cells per body: a box that has expanded and clumped is still walked cell by cell, with the empty cells skipped by
    // the unit list; beyond that the start array itself -- one search per cell -- costs more than the cell-run kernel's)
    const bool dense = count > 0 && count <= 16LL * (long long)n + 4096;
    if (dense && count + 1 > g->lb_capacity) {
      NBH_HIP(hipStreamSynchronize(st));
      (void)hipFree(g->d_cell_lb);
      g->d_cell_lb = nullptr;
      g->lb_capacity = 0;
      const long long cap = (count + 1) + (count + 1) / 2;  // grids grow and shrink with the box
      // (+ behind it the coarse level of the two-level search, or the two counters and the list of cell_mark_kernel's
      // long runs of empty cells: three words each, at most cap / kGapInline + 1 of them)
      static_assert(kGapInline == kLbCoarse, "one scratch area behind the start array serves both");
      const size_t extra = (size_t)(3 * (cap / kGapInline + 4) + 8);
      NBH_HIP(hipMalloc(reinterpret_cast<void**>(&g->d_cell_lb), ((size_t)cap + extra) * sizeof(int)));
      g->lb_capacity = cap;
      NBH_HIP(hipMemsetAsync(g->d_cell_lb + cap, 0, 2 * sizeof(int), st));  // both counters of the gap list
      g->gap_counters_clean = true;
      g->gap_tick = 0;
    }
    if (dense && count > 2LL * (long long)n) {  // more cells than bodies: two-level search (see cell_lb_coarse_kernel)
      int* coarse = g->d_cell_lb + g->lb_capacity;
      g->gap_counters_clean = false;
      const long long ncoarse = count / kLbCoarse + 2;
      hipLaunchKernelGGL(cell_lb_coarse_kernel, dim3((unsigned)((ncoarse + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                         g->d_keys_b, ni, (int)base, (int)count, coarse);
      hipLaunchKernelGGL(cell_lb_fine_kernel, dim3((unsigned)((count + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                         g->d_keys_b, (int)base, (int)count, coarse, g->d_cell_lb);
    } else if (dense && g->lb_by_position) {
      int* counters = g->d_cell_lb + g->lb_capacity;  // [2], then the list
      if (!g->gap_counters_clean) {  // (the two-level search above keeps its coarse level in the same scratch area)
        NBH_HIP(hipMemsetAsync(counters, 0, 2 * sizeof(int), st));
        g->gap_counters_clean = true;
      }
      int* cur = counters + (g->gap_tick & 1), *next = counters + ((g->gap_tick + 1) & 1);
      g->gap_tick++;
      hipLaunchKernelGGL(cell_mark_kernel, dim3((unsigned)((ni + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                         g->d_keys_b, ni, (int)base, (int)count, g->d_cell_lb, cur, counters + 2);
      hipLaunchKernelGGL(cell_gap_kernel, dim3(512), dim3(kBlock), 0, st, cur, counters + 2, g->d_cell_lb, next);
    } else if (dense) {
      hipLaunchKernelGGL(cell_lb_kernel, dim3((unsigned)((count + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                         g->d_keys_b, ni, (int)base, (int)count, g->d_cell_lb);
    }
    NBH_LAUNCH_CHECK();
    if (dense) {
      g->lb_valid = true;
      g->lb_base = base;
      g->lb_count = count;
    }
  }
  g->built_count = n;
  g->ranges_valid = false;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_grid_build(nbody_hip_grid* g, const nbody_particle_data* d) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null grid");
  if (!d) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null particle data");
  const size_t n = d->count;
  if (n == 0) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Particle count must be greater than 0");
  if (n > g->max_particles)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "particle count %zu exceeds the grid's capacity %zu "
                    "(sized from the first count seen, ref: force_spatial_hash.cu:372-374)", n, g->max_particles);
  if (!d->pos_x || !d->pos_y || !d->pos_z || !d->mass)
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "particle data has null arrays");
  nbody_hip_ctx* ctx = g->ctx;
  NBH_HIP(hipSetDevice(ctx->device));
  if (int rc = ctx->posm.reserve(n * sizeof(float4))) return rc;
  float4* posm = static_cast<float4*>(ctx->posm.ptr);
  return grid_build_packed(g, posm, n, nullptr, d);
}

extern "C" int nbody_hip_grid_drift_build(nbody_hip_grid* g, nbody_particle_data* d, float dt) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null grid");
  if (!d) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null particle data");
  const size_t n = d->count;
  if (n == 0) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Particle count must be greater than 0");
  if (n > g->max_particles)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "particle count %zu exceeds the grid's capacity %zu "
                    "(sized from the first count seen, ref: force_spatial_hash.cu:372-374)", n, g->max_particles);
  if (!d->pos_x || !d->pos_y || !d->pos_z || !d->mass || !d->vel_x || !d->vel_y || !d->vel_z || !d->acc_x || !d->acc_y ||
      !d->acc_z || !d->acc_old_x || !d->acc_old_y || !d->acc_old_z)
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "particle data has null arrays");
  nbody_hip_ctx* ctx = g->ctx;
  NBH_HIP(hipSetDevice(ctx->device));
  if (int rc = ctx->posm.reserve(n * sizeof(float4))) return rc;
  float4* posm = static_cast<float4*>(ctx->posm.ptr);
  return grid_build_packed(g, posm, n, nullptr, d, false, &dt);
}

extern "C" int nbody_hip_grid_build_packed(nbody_hip_grid* g, const nbody_float4* posm, size_t n,
                                           const float* bounds) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null grid");
  if (!posm) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (n == 0) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Particle count must be greater than 0");
  if (n > g->max_particles)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "particle count %zu exceeds the grid's capacity %zu", n, g->max_particles);
  if (bounds)
    for (int a = 0; a < 3; a++)
      if (!(bounds[3 + a] >= bounds[a]) || !(bounds[3 + a] - bounds[a] < INFINITY))
        return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "invalid grid bounds");
  NBH_HIP(hipSetDevice(g->ctx->device));
  return grid_build_packed(g, const_cast<float4*>(reinterpret_cast<const float4*>(posm)), n, bounds, nullptr,
                           bounds != nullptr);
}

namespace nbh {
// The work list of the unit form of hash_cell_force_kernel: one thread per cell of [cell_first, cell_end); a cell with
// cnt > 0 bodies appends ceil(cnt / chunk) units {cell - cell_first, chunk index}.  A workgroup scans its 256 counts
// and takes ONE slot range with an atomic (the list is in cell order inside a workgroup and in about cell order over
// all; the order does not reach the results: every target body belongs to exactly one unit).  count_next: the counter
// of the NEXT list, zeroed here (two counters alternate: no fill launch).
constexpr int kUnitCells = 8;  // consecutive cells per thread (one slot-range atomic per 2,048 cells: a workgroup per
                               // 256 cells made 86,000 atomics on one word at 22 M cells, 0.44 ms)
// Heavy first: the units of cells with more than kHeavyCell bodies (they sit in clumps: long windows too) go to the
// FRONT of the list, the others are written from the BACK down, and the force kernel takes the list front to back -- the
// hardware starts workgroups in index order, so the long units run first and the short ones fill the tail (longest
// processing time first; with the plain cell order a clump late in the list left the chip waiting for it).
constexpr int kHeavyCell = 48;
// BODIES (the split form, launch_cell_forces kern 9): the cells below min_cnt bodies make no units; their bodies' sorted
// positions go to the list `light` instead (count[3] = its length), for hash_body_force_kernel -- one lane per body, in
// full waves.  (The order of the list does not reach the results: a lane's sum involves no other lane.)
template <bool BODIES>
__global__ __launch_bounds__(kBlock) void cell_units_kernel(const CellGridView tgv, long long cell_first, long long cell_end,
                                                            int chunk, int2* __restrict__ units, int capacity,
                                                            int* __restrict__ count, int* __restrict__ count_next,
                                                            int min_cnt, int* __restrict__ light, int pair_items = 0) {
  // count[0]: heavy units, count[1]: bodies of the most crowded cell, count[2]: the other units (the host reads the
  // sum and the maximum, one call late, to choose the kernel form and to size its grid), count[3]: bodies in `light`
  __shared__ int wsum[3][kBlock / 64], wmax[kBlock / 64];
  __shared__ int base_s[3];
  const long long i0 = ((long long)blockIdx.x * kBlock + threadIdx.x) * kUnitCells;
  if (i0 == 0) { count_next[0] = 0; count_next[1] = 0; count_next[2] = 0; count_next[3] = 0; }
  int nus[kUnitCells];
  [[maybe_unused]] int lpos[kUnitCells], lcnt[kUnitCells];
  int nu[3] = {0, 0, 0};  // light units, heavy units, bodies for the list
  int mx = 0;
  unsigned heavy_mask = 0;
  {
    // the thread's nine start-array entries: two 16-byte loads and one word where the eight cells lie inside the array
    // and on a 16-byte boundary (the whole-grid call: always), nine clamped word loads otherwise
    int lbv[kUnitCells + 1];
    const long long k0 = cell_first + i0 - tgv.base;
    static_assert(kUnitCells == 8, "two int4 loads");
    if (cell_first + i0 + kUnitCells <= cell_end && k0 >= 0 && k0 + kUnitCells <= tgv.count && (k0 & 3) == 0) {
      const int4 a = *reinterpret_cast<const int4*>(tgv.lb + k0), b = *reinterpret_cast<const int4*>(tgv.lb + k0 + 4);
      lbv[0] = a.x; lbv[1] = a.y; lbv[2] = a.z; lbv[3] = a.w; lbv[4] = b.x; lbv[5] = b.y; lbv[6] = b.z; lbv[7] = b.w;
      lbv[8] = tgv.lb[k0 + 8];
    } else {
#pragma unroll
      for (int k = 0; k <= kUnitCells; k++) lbv[k] = cell_first + i0 + k <= cell_end ? tgv.lower(cell_first + i0 + k) : 0;
    }
    int lo = lbv[0];
#pragma unroll
    for (int k = 0; k < kUnitCells; k++) {
      nus[k] = 0;
      if constexpr (BODIES) lcnt[k] = 0;
      if (cell_first + i0 + k < cell_end) {
        const int hi = lbv[k + 1];
        const int cnt = hi - lo;
        nus[k] = cnt >= min_cnt ? (cnt + chunk - 1) / chunk : 0;
        if constexpr (BODIES) {
          lpos[k] = lo;
          lcnt[k] = cnt < min_cnt ? cnt : 0;
          nu[2] += pair_items ? (lcnt[k] + 1) >> 1 : lcnt[k];  // (pair_items: one list entry per two bodies of a cell)
        }
        lo = hi;
        const int h = cnt > kHeavyCell ? 1 : 0;
        heavy_mask |= (unsigned)h << k;
        nu[h] += nus[k];
        mx = max(mx, cnt);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
  int incl[3] = {nu[0], nu[1], nu[2]};
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int u0 = __shfl_up(incl[0], off, 64), u1 = __shfl_up(incl[1], off, 64);
    int u2 = 0;
    if constexpr (BODIES) u2 = __shfl_up(incl[2], off, 64);
    if ((int)(threadIdx.x & 63) >= off) { incl[0] += u0; incl[1] += u1; incl[2] += u2; }
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 63) { wsum[0][wv] = incl[0]; wsum[1][wv] = incl[1]; wsum[2][wv] = incl[2]; wmax[wv] = mx; }
  __syncthreads();
  int before[3] = {0, 0, 0}, total[3] = {0, 0, 0}, bmax = 0;
#pragma unroll
  for (int k = 0; k < kBlock / 64; k++) {
    if (k < wv) { before[0] += wsum[0][k]; before[1] += wsum[1][k]; before[2] += wsum[2][k]; }
    total[0] += wsum[0][k];
    total[1] += wsum[1][k];
    total[2] += wsum[2][k];
    bmax = max(bmax, wmax[k]);
  }
  if (threadIdx.x == 0) {
    base_s[1] = total[1] ? atomicAdd(count, total[1]) : 0;
    base_s[0] = total[0] ? atomicAdd(count + 2, total[0]) : 0;
    base_s[2] = (BODIES && total[2]) ? atomicAdd(count + 3, total[2]) : 0;
    if (bmax > 64) atomicMax(count + 1, bmax);  // (only crowded cells matter: most workgroups skip the atomic)
  }
  __syncthreads();
  int at[2] = {base_s[0] + before[0] + incl[0] - nu[0], base_s[1] + before[1] + incl[1] - nu[1]};
#pragma unroll
  for (int k = 0; k < kUnitCells; k++) {
    const int h = (heavy_mask >> k) & 1;
    for (int q = 0; q < nus[k]; q++) {
      const int a = at[h]++;
      const int slot = h ? a : capacity - 1 - a;  // heavy: from the front; light: from the back
      if (slot >= 0 && slot < capacity) units[slot] = make_int2((int)(i0 + k), q);
    }
  }
  if constexpr (BODIES) {
    int lat = base_s[2] + before[2] + incl[2] - nu[2];
#pragma unroll
    for (int k = 0; k < kUnitCells; k++) {
      if (pair_items) {  // (sorted position of the first of two bodies; top bit: the cell's last, odd one)
        for (int q = 0; q < lcnt[k]; q += 2) light[lat++] = (lpos[k] + q) | (q + 1 >= lcnt[k] ? (int)0x80000000u : 0);
      } else {
        for (int q = 0; q < lcnt[k]; q++) light[lat++] = lpos[k] + q;
      }
    }
  }
}
// the list's statistics for the host when the force kernel that follows is not the unit form (which exports them itself)
// (seq != 0: written LAST into host[2], after a system-scope fence: the host polls it)
__global__ void cell_units_export_kernel(const int* __restrict__ count, int* __restrict__ host, int seq = 0) {
  if (threadIdx.x == 0) {
    host[0] = count[0] + count[2];
    host[1] = count[1];
    if (seq) {
      __threadfence_system();
      __atomic_store_n(&host[2], seq, __ATOMIC_RELEASE);
      __threadfence_system();
    }
  }
}
}  // namespace nbh

// the work list of the cells [cell_first, cell_end) of grid gt (see cell_units_kernel); *cur: its two counters
static int make_unit_list(nbody_hip_ctx* ctx, nbody_hip_grid* gt, const CellGridView& tv, long long cell_first,
                          long long cell_end, int chunk, int** cur_out, int min_cnt = 1, bool bodies = false,
                          bool pair_items = false) {
  const size_t nb = gt->built_count;
  const size_t need = nb + nb / 64 + 1024;  // an occupied cell is at least one unit; chunks hold >= 64 bodies
  if (need > gt->units_cap) {
    NBH_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(gt->d_units);
    gt->d_units = nullptr;
    gt->units_cap = 0;
    const size_t cap = gt->max_particles + gt->max_particles / 64 + 1024;
    NBH_HIP(hipMalloc(reinterpret_cast<void**>(&gt->d_units), (cap > need ? cap : need) * sizeof(int2)));
    gt->units_cap = cap > need ? cap : need;
  }
  int* cur = gt->d_unit_count + 4 * (gt->unit_flip & 1);
  int* next = gt->d_unit_count + 4 * ((gt->unit_flip + 1) & 1);
  gt->unit_flip++;
  const long long cells = cell_end - cell_first;
  const dim3 ugrid((unsigned)((cells + kBlock * kUnitCells - 1) / (kBlock * kUnitCells)));
  if (bodies) {
    if (!gt->d_light) NBH_HIP(hipMalloc(reinterpret_cast<void**>(&gt->d_light), gt->max_particles * sizeof(int)));
    hipLaunchKernelGGL(cell_units_kernel<true>, ugrid, dim3(kBlock), 0, ctx->stream, tv, cell_first, cell_end, chunk, gt->d_units,
                       (int)gt->units_cap, cur, next, min_cnt, gt->d_light, pair_items ? 1 : 0);
  } else {
    hipLaunchKernelGGL(cell_units_kernel<false>, ugrid, dim3(kBlock), 0, ctx->stream, tv, cell_first, cell_end, chunk, gt->d_units,
                       (int)gt->units_cap, cur, next, min_cnt, nullptr);
  }
  NBH_LAUNCH_CHECK();
  *cur_out = cur;
  return NBODY_HIP_OK;
}

static int launch_cell_forces(nbody_hip_ctx* ctx, const CellGridView& tv, const CellGridView& sv, int gx, int gy,
                              int gz, long long cell_first, long long cell_end, int kern, bool guard, float cutoff2,
                              float eps2, float G, float* ax, float* ay, float* az, float4* acc4, int accumulate,
                              nbody_hip_grid* gt = nullptr, int hint_slot = 0, int* prebuilt = nullptr) {
  if (cell_end <= cell_first) return NBODY_HIP_OK;
  if (kern == 10 && (guard || cell_end - cell_first >= 0x7fffffffLL)) kern = 8;  // (eps ~ 0: the compare + select forms)
  if (kern == 10) {  // two bodies of one cell per lane, every cell (see hash_body2_force_kernel)
    if (!gt) return NBH_FAIL(NBODY_HIP_ERR_STATE, "the two-bodies-per-lane kernel needs the target grid");
    int* cur = nullptr;
    if (int rc = make_unit_list(ctx, gt, tv, cell_first, cell_end, 128, &cur, 0x7fffffff, true, true)) return rc;
    const unsigned blocks = (unsigned)((gt->built_count + kBlock - 1) / kBlock);  // (at most one item per body)
    hipLaunchKernelGGL(hash_body2_force_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, tv, sv, gt->d_keys_b, gx, gy, gz,
                       cutoff2, eps2, G, ax, ay, az, acc4, accumulate, gt->d_light, cur + 3);
    NBH_LAUNCH_CHECK();
    return NBODY_HIP_OK;
  }
  if (kern == 8 || kern == 9) {
    // 8: one lane per body for every body.  9 (the automatic form below kBodyBelow bodies per cell), SPLIT by cell: bodies
    // of cells with fewer than kSplitCnt bodies take one lane each; the cells of kSplitCnt and more -- the clumps of a
    // box that has clumped: cells of hundreds of bodies and windows of a thousand entries -- go on a work list and the
    // wave-per-cell kernel (two targets per lane) takes them.  The split is a function of the cell alone, so the result
    // is a function of the input alone; the list's length reaches the host one call late and only sizes the launch.
    if (!gt) return NBH_FAIL(NBODY_HIP_ERR_STATE, "the one-lane-per-body kernel needs the target grid");
    const bool split = kern == 9 && cell_end - cell_first < 0x7fffffffLL;  // (NBH_HASH_UNITS does not reach this: it
                                                                           // chooses between two bit-identical forms, this is a kernel shape)
    const unsigned blocks = (unsigned)((gt->built_count + kBlock - 1) / kBlock);
    int* cur = nullptr;
    if (split)  // the crowded cells' units and the light cells' bodies, one pass over the cells
      if (int rc = make_unit_list(ctx, gt, tv, cell_first, cell_end, 128, &cur, gt->split_cnt, true)) return rc;
    const int* light = split ? gt->d_light : nullptr;
    const int* light_count = split ? cur + 3 : nullptr;
    if (guard)
      hipLaunchKernelGGL((hash_body_force_kernel<true>), dim3(blocks), dim3(kBlock), 0, ctx->stream, tv, sv, gt->d_keys_b,
                         cell_first, cell_end, gx, gy, gz, cutoff2, eps2, G, ax, ay, az, acc4, accumulate, light, light_count);
    else
      hipLaunchKernelGGL((hash_body_force_kernel<false>), dim3(blocks), dim3(kBlock), 0, ctx->stream, tv, sv, gt->d_keys_b,
                         cell_first, cell_end, gx, gy, gz, cutoff2, eps2, G, ax, ay, az, acc4, accumulate, light, light_count);
    NBH_LAUNCH_CHECK();
    if (!split) return NBODY_HIP_OK;
    int* uhint = gt->h_unit_hint_dev ? gt->h_unit_hint_dev + 4 * hint_slot : nullptr;
    const int hint = uhint ? gt->h_unit_hint[4 * hint_slot] : 0;
    const long long bound = (long long)gt->built_count / gt->split_cnt + (long long)gt->built_count / 128 + 1;
    long long est = hint > 0 ? (long long)hint + hint / 64 + 64 : 4096;  // (first call: a guess; the kernel strides)
    if (est > bound) est = bound;
    long long ublocks = (est + 4 * kCellsPerWave - 1) / (4 * kCellsPerWave);
    if (ublocks < 8) ublocks = 8;
    const int uper = (int)((ublocks + 7) / 8);
    if (guard)
      hipLaunchKernelGGL((hash_cell_force_kernel<true, 2, false, true, NBH_HASH_SPLIT_FILTER>), dim3((unsigned)(uper * 8 * (4 / kCellWPB))), dim3(64 * kCellWPB), 0, ctx->stream,
                         tv, sv, gx, gy, gz, cell_first, cell_end, uper, cutoff2, eps2, G, ax, ay, az, acc4, accumulate,
                         gt->d_units, cur, uhint, (int)gt->units_cap, 1);
    else
      hipLaunchKernelGGL((hash_cell_force_kernel<false, 2, false, true, NBH_HASH_SPLIT_FILTER>), dim3((unsigned)(uper * 8 * (4 / kCellWPB))), dim3(64 * kCellWPB), 0, ctx->stream,
                         tv, sv, gx, gy, gz, cell_first, cell_end, uper, cutoff2, eps2, G, ax, ay, az, acc4, accumulate,
                         gt->d_units, cur, uhint, (int)gt->units_cap, 1);
    NBH_LAUNCH_CHECK();
    return NBODY_HIP_OK;
  }
  const long long nblk = (cell_end - cell_first + 4 * kCellsPerWave - 1) / (4 * kCellsPerWave);
  int per_xcd = (int)((nblk + 7) / 8);
  // The unit list (occupied cells in chunks of 64 R bodies) is what the kernel takes when the previous call of this
  // kind saw crowded cells (> 64 bodies) or many empty ones; while the bodies are spread evenly it takes the cell range
  // itself and the list is only made every eighth call, for its statistics (list + export are 19 us at 4.2 M bodies).
  const bool can_list = gt && gt->use_units && kern != 5 && cell_end - cell_first < 0x7fffffffLL;
  bool by_units = false;
  const int2* units = nullptr;
  const int* ucount = nullptr;
  int* uhint = nullptr;
  if (can_list) {
    const int R = (kern == 2 || kern == 7) ? 1 : (kern == 4 ? 4 : 2);  // (3, 6: two; 7: units of kUnitChunk2 = 64 bodies)
    static_assert(kUnitChunk2 == 64, "the two-phase kernel's units are the 64-body chunks of the one-body-per-lane list");
    const size_t nb = gt->built_count;
    const long long cells = cell_end - cell_first;
    uhint = gt->h_unit_hint_dev ? gt->h_unit_hint_dev + 4 * hint_slot : nullptr;
    const int hint = uhint ? gt->h_unit_hint[4 * hint_slot] : 0, crowd = uhint ? gt->h_unit_hint[4 * hint_slot + 1] : 0;
    by_units = prebuilt || gt->use_units == 2 || (hint > 0 && (crowd > 64 || (long long)hint * 10 < cells * 9));
    if (by_units || (uhint && (gt->stat_tick++ & 7) == 0)) {
      int* cur = prebuilt;  // (the caller made this very list: one body per lane, chunks of 64)
      if (!cur)
        if (int rc = make_unit_list(ctx, gt, tv, cell_first, cell_end, 64 * R, &cur)) return rc;
      units = gt->d_units;
      ucount = cur;
      if (!by_units) {
        hipLaunchKernelGGL(cell_units_export_kernel, dim3(1), dim3(64), 0, ctx->stream, cur, uhint);
        NBH_LAUNCH_CHECK();
      }
    }
    if (by_units) {
      // grid: one group of KC units per wave for the list length the previous call of this kind saw (+ 1/64); the
      // kernel strides if the list is longer.  Never more than the cells / bodies allow.
      long long bound = (long long)(cells < (long long)nb ? cells : (long long)nb) + (long long)(nb / (64 * R)) + 1;
      long long est = hint > 0 ? (long long)hint + hint / 64 + 64 : bound;
      if (est > bound) est = bound;
      long long blocks = (est + 4 * kCellsPerWave - 1) / (4 * kCellsPerWave);
      if (blocks < 8) blocks = 8;
      per_xcd = (int)((blocks + 7) / 8);
    }
  }
#define NBH_CELL_LAUNCH(GD, RR)                                                                                  \
  do {                                                                                                           \
    if (by_units)                                                                                                \
      hipLaunchKernelGGL((hash_cell_force_kernel<GD, RR, false, true>), dim3((unsigned)(per_xcd * 8 * (4 / kCellWPB))), dim3(64 * kCellWPB), 0, \
                         ctx->stream, tv, sv, gx, gy, gz, cell_first, cell_end, per_xcd, cutoff2, eps2, G, ax, ay, az,   \
                         acc4, accumulate, units, ucount, uhint, (int)gt->units_cap);                                                \
    else                                                                                                         \
      hipLaunchKernelGGL((hash_cell_force_kernel<GD, RR>), dim3((unsigned)(per_xcd * 8 * (4 / kCellWPB))), dim3(64 * kCellWPB), 0,        \
                         ctx->stream, tv, sv, gx, gy, gz, cell_first, cell_end, per_xcd, cutoff2, eps2, G, ax, ay, az,   \
                         acc4, accumulate);                                                                      \
  } while (0)
  if (kern == 7) {  // two-phase form: distance masks first, then the accepted candidates only
#define NBH_CELL2_LAUNCH(GD)                                                                                     \
  do {                                                                                                           \
    if (by_units)                                                                                                \
      hipLaunchKernelGGL((hash_cell_force2_kernel<GD, true>), dim3((unsigned)(per_xcd * 8)), dim3(kBlock), 0,     \
                         ctx->stream, tv, sv, gx, gy, gz, cell_first, cell_end, per_xcd, cutoff2, eps2, G, ax, ay, az,   \
                         acc4, accumulate, units, ucount, uhint, (int)gt->units_cap);                            \
    else                                                                                                         \
      hipLaunchKernelGGL((hash_cell_force2_kernel<GD, false>), dim3((unsigned)(per_xcd * 8)), dim3(kBlock), 0,    \
                         ctx->stream, tv, sv, gx, gy, gz, cell_first, cell_end, per_xcd, cutoff2, eps2, G, ax, ay, az,   \
                         acc4, accumulate);                                                                      \
  } while (0)
    if (guard) NBH_CELL2_LAUNCH(true); else NBH_CELL2_LAUNCH(false);
#undef NBH_CELL2_LAUNCH
  } else if (kern == 5) {  // timing probe (see the kernel): not forces
    hipLaunchKernelGGL((hash_cell_force_kernel<false, 2, true>), dim3((unsigned)(per_xcd * 8 * (4 / kCellWPB))), dim3(64 * kCellWPB), 0, ctx->stream, tv,
                       sv, gx, gy, gz, cell_first, cell_end, per_xcd, cutoff2, eps2, G, ax, ay, az, acc4, accumulate);
  } else if (kern == 2) { if (guard) NBH_CELL_LAUNCH(true, 1); else NBH_CELL_LAUNCH(false, 1); }
  else if (kern == 4) { if (guard) NBH_CELL_LAUNCH(true, 4); else NBH_CELL_LAUNCH(false, 4); }
  else if (kern == 6) {  // two targets per lane, window filtered by the box of the targets (crowded cells)
#define NBH_CELL_LAUNCH_F(GD)                                                                                    \
  do {                                                                                                           \
    if (by_units)                                                                                                \
      hipLaunchKernelGGL((hash_cell_force_kernel<GD, 2, false, true, true>), dim3((unsigned)(per_xcd * 8 * (4 / kCellWPB))), dim3(64 * kCellWPB), 0, \
                         ctx->stream, tv, sv, gx, gy, gz, cell_first, cell_end, per_xcd, cutoff2, eps2, G, ax, ay, az,   \
                         acc4, accumulate, units, ucount, uhint, (int)gt->units_cap);                            \
    else                                                                                                         \
      hipLaunchKernelGGL((hash_cell_force_kernel<GD, 2, false, false, true>), dim3((unsigned)(per_xcd * 8 * (4 / kCellWPB))), dim3(64 * kCellWPB), 0, \
                         ctx->stream, tv, sv, gx, gy, gz, cell_first, cell_end, per_xcd, cutoff2, eps2, G, ax, ay, az,   \
                         acc4, accumulate);                                                                      \
  } while (0)
    if (guard) NBH_CELL_LAUNCH_F(true); else NBH_CELL_LAUNCH_F(false);
#undef NBH_CELL_LAUNCH_F
  }
  else                { if (guard) NBH_CELL_LAUNCH(true, 2); else NBH_CELL_LAUNCH(false, 2); }
#undef NBH_CELL_LAUNCH
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

static int grid_forces_common(nbody_hip_grid* g, float cutoff, float G, float eps, float* ax,
                              float* ay, float* az, float4* acc4) {
  if (!(cutoff > 0.0f) || !(cutoff < INFINITY))
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Spatial hash cutoff must be positive and finite");
  nbody_hip_ctx* ctx = g->ctx;
  NBH_HIP(hipSetDevice(ctx->device));
  const int n = (int)g->built_count;
  const int gx = g->info.dims[0], gy = g->info.dims[1], gz = g->info.dims[2];
  // cells per wave: ~64 targets per wave at the mean occupancy
  const double rho = (double)n / (double)(g->lb_valid && g->lb_count > 0 ? g->lb_count : g->info.total);
  int W = rho > 0 ? (int)(64.0 / rho + 0.5) : kMaxWv;
  if (W < 1) W = 1;
  if (W > kMaxWv) W = kMaxWv;
  const dim3 grid((gx + 4 * W - 1) / (4 * W), gy, gz);
  if (grid.y > 65535u || grid.z > 65535u)
    return NBH_FAIL(NBODY_HIP_ERR_RESOURCE, "Spatial hash grid too large: reduce cell_size or bounding box");
  const float eps2 = eps * eps, cutoff2 = cutoff * cutoff;  // :312-313
  const bool guard = eps2 < 1e-12f || !cut_const_ok(cutoff2);  // (the general instantiation: compare + select, d2 > 0 test)
  const bool strict = cutoff > g->cell_size;
  // wave-per-cell kernel: needs the cell_lb array; pays from about two bodies per cell
  int kern = g->tune_kernel;
  // measured (tools/hash_kernels.py, profiles/r02_hash_kernels.txt): rho 0.9: cell-run 0.16 ms vs 0.28;
  // rho 3.7: 0.74 vs 0.31 (R = 1) / 0.49 (R = 2); rho 14.6: 1.87 vs 1.20 / 1.02; rho 107: 6.7 vs 8.1 / 6.1
  // clustered bodies (tools/hash_kernels_sphere.py; rho = bodies per cell of the WHOLE grid, the occupied part
  // is denser): sphere of 10,000 in a 21^3 grid, rho 1.1, cutoff 2 > cell: 0.050 vs 0.018 ms; 100,000 bodies,
  // rho 10.8, cutoff 2: 0.86 vs 0.10 ms (the cell-run kernel's |cx_j - cx_i| test); rho 1.5 / 1.9: 0.079 /
  // 0.78 vs 0.069 / 0.47 ms.  Uniform box at rho 0.9: 0.16 (cell runs) vs 0.28 ms.
  // Which kernel runs is a function of the INPUT alone (body count, grid, cutoff -- and, below one body per cell, the
  // occupancy of the grid just built): kernels of different shapes sum in different orders, and a choice that hung on
  // statistics arriving asynchronously from earlier calls would make the bits depend on timing.  (Between the cell
  // range and the unit list of ONE shape the previous call's statistics do decide: those two are bit-identical.)
  //   >= 1 body per cell (0.5 when cutoff > cell): wave per cell, 1 / 2 bodies per lane below / from 8 per cell, the
  //   filtered form from 40 per cell;
  //   below that, with start arrays (<= 16 cells per body): the unit list of THIS grid is made and its most crowded
  //   cell read back (a second poll of mapped host memory, ~10 us in a regime of multi-millisecond steps): crowded
  //   cells (> 64 bodies: a box that has expanded and clumped -- mean 0.2 bodies per cell, cells of 300 beside a
  //   majority of empty ones) take the wave-per-cell kernel over that list (3.5 ms against 4.9 ms), an evenly sparse
  //   grid takes the cell-run kernel (0.16 against 0.26 ms at 0.9 bodies per cell).
  //   round 4: below kBodyBelow bodies per cell ONE LANE PER BODY (hash_body_force_kernel) replaces both the wave-per-cell
  //   form with one body per lane and the cell-run kernel wherever the grid carries start arrays: 0.027 against 0.155 ms
  //   at 0.9 bodies per cell, 0.142 against 0.329 ms at 3.7, 0.77 against 0.87 ms at 6.6, level at 8.8
  //   (profiles/r04_hash_kernels.txt) -- and a box that has expanded and clumped (mean 0.2 bodies per cell, 0.9 M cells of
  //   one to four bodies beside clumps) needs neither a work list nor a read-back of its occupancy any more.
  int* prebuilt = nullptr;
  if (kern == 0) {
    if (!g->lb_valid) {
      kern = 1;
    } else if (rho >= kBodyBelow) {
      kern = rho < (strict ? kFilterFrom : g->filter_from_inside) ? 3 : 6;
    } else {
      // (small systems: the list pass and the third launch of the split form are ~10 us, a third of the whole evaluation
      // at 262,144 bodies; below kSplitFrom bodies every body takes a lane)
      kern = n >= kSplitFrom ? 9 : 8;
    }
  }
  if ((kern == 8 || kern == 9 || kern == 10) && !g->lb_valid) kern = 1;  // (no start arrays: the cell-run kernel)
  if (kern == 10 && guard) kern = 8;                                     // (eps ~ 0: the compare + select forms)
  if (kern != 1 && g->lb_valid) {
    const CellGridView view{g->d_sorted, g->d_cell_lb, g->d_idx_b, g->lb_base, g->lb_count};
    return launch_cell_forces(ctx, view, view, gx, gy, gz, g->lb_base, g->lb_base + g->lb_count, kern, guard, cutoff2,
                              eps2, G, ax, ay, az, acc4, 0, g, 0, prebuilt);
  }
#define NBH_HASH_LAUNCH(GD, ST)                                                                   \
  hipLaunchKernelGGL((hash_force_kernel<GD, ST>), grid, dim3(kBlock), 0, ctx->stream, g->d_sorted, \
                     g->d_keys_b, g->d_idx_b, n, g->d_info, W, cutoff2, eps2, G, ax, ay, az, acc4)
  if (guard) { if (strict) NBH_HASH_LAUNCH(true, true); else NBH_HASH_LAUNCH(true, false); }
  else       { if (strict) NBH_HASH_LAUNCH(false, true); else NBH_HASH_LAUNCH(false, false); }
#undef NBH_HASH_LAUNCH
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_grid_compute_forces(nbody_hip_grid* g, nbody_particle_data* d, float cutoff,
                                             float G, float eps) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null grid");
  if (!d) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null particle data");
  if (g->built_count == 0 || g->built_count != d->count)
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "grid was not built for this particle set");
  if (!d->acc_x || !d->acc_y || !d->acc_z) return NBH_FAIL(NBODY_HIP_ERR_STATE, "particle data has null arrays");
  return grid_forces_common(g, cutoff, G, eps, d->acc_x, d->acc_y, d->acc_z, nullptr);
}

extern "C" int nbody_hip_grid_compute_forces_packed(nbody_hip_grid* g, float cutoff, float G, float eps,
                                                    nbody_float4* acc_out) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null grid");
  if (!acc_out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (g->built_count == 0) return NBH_FAIL(NBODY_HIP_ERR_STATE, "grid has not been built");
  return grid_forces_common(g, cutoff, G, eps, nullptr, nullptr, nullptr, reinterpret_cast<float4*>(acc_out));
}

extern "C" int nbody_hip_grid_sorted_bodies(nbody_hip_grid* g, size_t first, size_t count, nbody_float4* out) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null grid");
  if (count == 0) return NBODY_HIP_OK;
  if (!out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (g->built_count == 0 || first + count > g->built_count)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "range [%zu, %zu) outside the %zu bodies of the last build", first,
                    first + count, g->built_count);
  NBH_HIP(hipSetDevice(g->ctx->device));
  NBH_HIP(hipMemcpyAsync(out, g->d_sorted + first, count * sizeof(float4), hipMemcpyDeviceToDevice, g->ctx->stream));
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_grid_set_slab(nbody_hip_grid* g, int z_first, int z_count) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null grid");
  g->slab_z0 = z_first;
  g->slab_nz = z_count;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_grid_forces_pair_packed(nbody_hip_grid* gt, nbody_hip_grid* gs, int z_first, int z_count,
                                                 float cutoff, float G, float eps, nbody_float4* acc_out,
                                                 int accumulate) {
  if (!gt || !gs) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null grid");
  if (!acc_out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (gt->built_count == 0 || gs->built_count == 0) return NBH_FAIL(NBODY_HIP_ERR_STATE, "grid has not been built");
  if (!(cutoff > 0.0f) || !(cutoff < INFINITY))
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Spatial hash cutoff must be positive and finite");
  if (!gt->lb_valid || !gs->lb_valid)
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "grid too sparse for the two-grid force kernel (no per-cell start array)");
  for (int a = 0; a < 3; a++)
    if (gt->info.dims[a] != gs->info.dims[a] || gt->info.bmin[a] != gs->info.bmin[a])
      return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "the two grids must share origin and dimensions");
  if (gt->cell_size != gs->cell_size) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "the two grids must share the cell size");
  nbody_hip_ctx* ctx = gt->ctx;
  NBH_HIP(hipSetDevice(ctx->device));
  const int gx = gt->info.dims[0], gy = gt->info.dims[1], gz = gt->info.dims[2];
  const long long layer = (long long)gx * gy;
  long long z0 = z_first < 0 ? 0 : z_first, z1 = z_count <= 0 ? gz : (long long)z_first + z_count;
  if (z1 > gz) z1 = gz;
  // only cells the target grid's start array covers can hold targets
  long long c0 = z0 * layer, c1 = z1 * layer;
  if (c0 < gt->lb_base) c0 = gt->lb_base;
  if (c1 > gt->lb_base + gt->lb_count) c1 = gt->lb_base + gt->lb_count;
  const float eps2 = eps * eps, cutoff2 = cutoff * cutoff;
  const double rho = (double)gt->built_count / (double)(gt->lb_count > 0 ? gt->lb_count : 1);
  int kern = gt->tune_kernel;
  if (kern < 2) kern = rho < kBodyBelow ? (gt->built_count >= (size_t)kSplitFrom ? 9 : 8) : (rho < (cutoff > gt->cell_size ? kFilterFrom : gt->filter_from_inside) ? 3 : 6);
  const CellGridView tv{gt->d_sorted, gt->d_cell_lb, gt->d_idx_b, gt->lb_base, gt->lb_count};
  const CellGridView sv{gs->d_sorted, gs->d_cell_lb, gs->d_idx_b, gs->lb_base, gs->lb_count};
  return launch_cell_forces(ctx, tv, sv, gx, gy, gz, c0, c1, kern, eps2 < 1e-12f || !cut_const_ok(cutoff2), cutoff2, eps2, G, nullptr, nullptr,
                            nullptr, reinterpret_cast<float4*>(acc_out), accumulate ? 1 : 0, gt, accumulate ? 1 : 0);
}

extern "C" int nbody_hip_grid_export_layer(nbody_hip_grid* g, int z, nbody_float4* bodies_out, int* lb_out) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null grid");
  if (!bodies_out || !lb_out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (g->built_count == 0) return NBH_FAIL(NBODY_HIP_ERR_STATE, "grid has not been built");
  if (!g->lb_valid) return NBH_FAIL(NBODY_HIP_ERR_STATE, "grid too sparse for per-cell start arrays: no layer to export");
  const long long layer = (long long)g->info.dims[0] * g->info.dims[1];
  const long long k0 = (long long)z * layer - g->lb_base;
  if (z < 0 || z >= g->info.dims[2] || k0 < 0 || k0 + layer > g->lb_count)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "layer %d is outside the z layers this grid holds", z);
  nbody_hip_ctx* ctx = g->ctx;
  NBH_HIP(hipSetDevice(ctx->device));
  const long long work = layer + 1 > (long long)g->built_count ? layer + 1 : (long long)g->built_count;
  long long blocks = (work + kBlock - 1) / kBlock;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(layer_export_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, g->d_sorted, g->d_cell_lb + k0,
                     (int)layer, reinterpret_cast<float4*>(bodies_out), lb_out);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_grid_forces_layer_packed(nbody_hip_grid* gt, int z, const nbody_float4* src_bodies, const int* src_lb,
                                                  int src_z, float cutoff, float G, float eps, nbody_float4* acc_out,
                                                  int accumulate) {
  if (!gt) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null grid");
  if (!src_bodies || !src_lb || !acc_out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (gt->built_count == 0) return NBH_FAIL(NBODY_HIP_ERR_STATE, "grid has not been built");
  if (!(cutoff > 0.0f) || !(cutoff < INFINITY))
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Spatial hash cutoff must be positive and finite");
  if (!gt->lb_valid) return NBH_FAIL(NBODY_HIP_ERR_STATE, "grid too sparse for the two-grid force kernel (no per-cell start array)");
  const int gx = gt->info.dims[0], gy = gt->info.dims[1], gz = gt->info.dims[2];
  if (z < 0 || z >= gz || src_z < 0 || src_z >= gz) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "layer outside the grid");
  nbody_hip_ctx* ctx = gt->ctx;
  NBH_HIP(hipSetDevice(ctx->device));
  const long long layer = (long long)gx * gy;
  long long c0 = (long long)z * layer, c1 = c0 + layer;
  if (c0 < gt->lb_base) c0 = gt->lb_base;
  if (c1 > gt->lb_base + gt->lb_count) c1 = gt->lb_base + gt->lb_count;
  const float eps2 = eps * eps, cutoff2 = cutoff * cutoff;
  const double rho = (double)gt->built_count / (double)(gt->lb_count > 0 ? gt->lb_count : 1);
  int kern = gt->tune_kernel;
  if (kern < 2) kern = rho < kBodyBelow ? (gt->built_count >= (size_t)kSplitFrom ? 9 : 8) : (rho < (cutoff > gt->cell_size ? kFilterFrom : gt->filter_from_inside) ? 3 : 6);
  const CellGridView tv{gt->d_sorted, gt->d_cell_lb, gt->d_idx_b, gt->lb_base, gt->lb_count};
  // the source layer as the sender exported it (nbody_hip_grid_export_layer): cells outside it hold nothing
  const CellGridView sv{reinterpret_cast<const float4*>(src_bodies), src_lb, nullptr, (long long)src_z * layer, layer};
  return launch_cell_forces(ctx, tv, sv, gx, gy, gz, c0, c1, kern, eps2 < 1e-12f || !cut_const_ok(cutoff2), cutoff2, eps2, G, nullptr,
                            nullptr, nullptr, reinterpret_cast<float4*>(acc_out), accumulate ? 1 : 0, gt, 1);
}

extern "C" int nbody_hip_bbox_packed(nbody_hip_ctx* ctx, const nbody_float4* posm, size_t n,
                                     float* bounds_device) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (!posm || !bounds_device) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (n == 0 || n > 0x3fffffffu) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "body count out of range");
  NBH_HIP(hipSetDevice(ctx->device));
  if (int rc = ctx->reduce.reserve(64)) return rc;
  unsigned int* enc = static_cast<unsigned int*>(ctx->reduce.ptr);
  if (int rc = launch_bbox(ctx, reinterpret_cast<const float4*>(posm), (int)n, enc)) return rc;
  hipLaunchKernelGGL(bbox_decode_kernel, dim3(1), dim3(64), 0, ctx->stream, enc, bounds_device);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_drift_bbox_packed(nbody_hip_ctx* ctx, nbody_float4* posm, const nbody_float4* vel,
                                           const nbody_float4* acc, size_t n, float dt, unsigned int* enc_device,
                                           float* bounds_device) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (!posm || !vel || !acc || !enc_device || !bounds_device) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (n == 0 || n > 0x3fffffffu) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "body count out of range");
  NBH_HIP(hipSetDevice(ctx->device));
  const int blocks = (int)((n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(drift_bbox_packed_kernel, dim3(blocks < 256 ? blocks : 256), dim3(kBlock), 0, ctx->stream,
                     reinterpret_cast<float4*>(posm), reinterpret_cast<const float4*>(vel),
                     reinterpret_cast<const float4*>(acc), (int)n, dt, enc_device);
  NBH_LAUNCH_CHECK();
  hipLaunchKernelGGL(bbox_decode_rearm_kernel, dim3(1), dim3(64), 0, ctx->stream, enc_device, bounds_device);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_cell_z_packed(nbody_hip_ctx* ctx, const nbody_float4* posm, size_t n, float lo_z,
                                       float cell_size, int gz, int* cz_device) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (n == 0) return NBODY_HIP_OK;
  if (!posm || !cz_device) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (!(cell_size > 0.0f) || gz < 1) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "invalid grid");
  NBH_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(cell_z_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                     reinterpret_cast<const float4*>(posm), (int)n, lo_z, cell_size, gz, cz_device);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_grid_info(const nbody_hip_grid* g, int dims[3], int* total_cells,
                                   float bbox_min[3], float bbox_max[3]) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null grid");
  for (int a = 0; a < 3; a++) {
    if (dims) dims[a] = g->built_count ? g->info.dims[a] : 0;
    if (bbox_min) bbox_min[a] = g->info.bmin[a];
    if (bbox_max) bbox_max[a] = g->info.bmax[a];
  }
  if (total_cells) *total_cells = g->built_count ? (int)g->info.total : 0;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_grid_count(const nbody_hip_grid* g, size_t* built_count) {
  if (!g || !built_count) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  *built_count = g->built_count;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_grid_copy_cell_data(nbody_hip_grid* g, int* cell_start, int* cell_end,
                                             int* particle_cells, int* sorted_indices) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null grid");
  if (g->built_count == 0) return NBH_FAIL(NBODY_HIP_ERR_STATE, "grid has not been built");
  nbody_hip_ctx* ctx = g->ctx;
  NBH_HIP(hipSetDevice(ctx->device));
  const int n = (int)g->built_count;
  const long long total = g->info.total;
  const int blocks = (n + kBlock - 1) / kBlock;
  if (cell_start || cell_end) {
    if (total > g->cell_capacity) {
      NBH_HIP(hipDeviceSynchronize());
      (void)hipFree(g->d_cell_start); (void)hipFree(g->d_cell_end);
      g->d_cell_start = g->d_cell_end = nullptr;
      g->cell_capacity = 0;
      NBH_HIP(hipMalloc(reinterpret_cast<void**>(&g->d_cell_start), total * sizeof(int)));
      NBH_HIP(hipMalloc(reinterpret_cast<void**>(&g->d_cell_end), total * sizeof(int)));
      g->cell_capacity = total;
      g->ranges_valid = false;
    }
    if (!g->ranges_valid) {
      NBH_HIP(hipMemsetAsync(g->d_cell_start, 0, total * sizeof(int), ctx->stream));
      NBH_HIP(hipMemsetAsync(g->d_cell_end, 0, total * sizeof(int), ctx->stream));
      hipLaunchKernelGGL(cell_ranges_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, g->d_keys_b, n,
                         g->d_cell_start, g->d_cell_end);
      NBH_LAUNCH_CHECK();
      g->ranges_valid = true;
    }
    if (cell_start) NBH_HIP(hipMemcpyAsync(cell_start, g->d_cell_start, total * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (cell_end) NBH_HIP(hipMemcpyAsync(cell_end, g->d_cell_end, total * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  }
  if (particle_cells) {
    // keys_a still holds the unsorted cell ids (radix_sort_pairs does not modify its input)
    NBH_HIP(hipMemcpyAsync(particle_cells, g->d_keys_a, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  }
  if (sorted_indices)
    NBH_HIP(hipMemcpyAsync(sorted_indices, g->d_idx_b, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  NBH_HIP(hipStreamSynchronize(ctx->stream));
  return NBODY_HIP_OK;
}
