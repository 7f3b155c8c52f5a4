// slab.hip -- z-slab sharding of the spatial hash over the GPUs of a node (SURVEY.md section 8e row 3;
// no reference counterpart: the reference is single-GPU).  The linear cell id x + y gx + z gx gy
// (ref: src/cuda/force_spatial_hash.cu:48) makes a range of z layers a contiguous block of cells, so
// rank r owns the bodies of the layers [r gz / W, (r+1) gz / W).  After the drift every rank runs ONE
// partition pass over its bodies:
//
//   slab_count_kernel    layer z of every body on the GLOBAL grid (box = all-reduced min/max, padded and
//                        sized exactly like SpatialHashGrid::build, :225-246), its new owner, per-block
//                        counts per owner, and the rank's histogram of bodies per layer
//   slab_scan_kernel     offsets of every (block, owner) group; row `rank` of the W x W send matrix
//   slab_leavers_kernel  the bodies that CHANGE OWNER as 64-byte rows {x,y,z,m | vx,vy,vz,0 | ax,ay,az,0 |
//                        id,z,0,0} grouped by new owner, each group in the bodies' original order, and
//                        the list of the slots they leave behind (ascending) -- deterministic, no sort,
//                        no host round trip; the bodies that stay (almost all) are not touched
//
// The host then all-reduces (sum) the send matrix and the layer histogram -- after which every rank
// knows how many rows it receives from whom, how many bodies it will own, and how many bodies its two
// halo layers hold -- and reads them back with the grid size in its ONE synchronisation of the step.
// After the exchange slab_fill_kernel puts the arrivals into the vacated slots (surplus arrivals are
// appended; surplus slots are closed with bodies taken from the end), so a step moves 64 bytes per
// MIGRATING body and nothing else.
#include "common.h"

namespace nbh {

constexpr int kMaxRanks = 64;
constexpr int kHistLds = 4096;

struct SlabGrid {
  float lo_z;
  int gx, gy, gz;
};

__host__ __device__ inline int slab_axis_cells(float lo, float hi, float cell) {
  const float cells = ceilf((hi - lo) / cell);  // force_spatial_hash.cu:244-246
  return (cells < 1.0e9f && cells >= 0.0f) ? (int)cells + 1 : 0x40000000;
}

__device__ __forceinline__ SlabGrid slab_grid(const float* __restrict__ gbox, float cell) {
  SlabGrid g;
  const float pad = 0.001f;  // :225-231
  g.lo_z = gbox[2] - pad;
  g.gx = slab_axis_cells(gbox[0] - pad, gbox[3] + pad, cell);
  g.gy = slab_axis_cells(gbox[1] - pad, gbox[4] + pad, cell);
  g.gz = slab_axis_cells(g.lo_z, gbox[5] + pad, cell);
  return g;
}

__device__ __forceinline__ int slab_layer(float z, const SlabGrid& g, float cell) {
  const int c = (int)floorf((z - g.lo_z) / cell);  // :36-48
  return min(max(c, 0), g.gz - 1);
}

// owner of layer z when rank r owns [r gz / W, (r+1) gz / W): r = ceil((z+1) W / gz) - 1
__host__ __device__ inline int slab_owner(int z, int gz, int world) {
  return (int)((((long long)z + 1) * world - 1) / gz);
}

// ... or, with cuts (n >= 0): the ranks' slabs are bounded by PHYSICAL z coordinates (chosen by the host so that the
// slabs hold about the same number of bodies): the owner of layer z is the number of cuts at or below the layer's
// centre, lo_z + (z + 1/2) cell -- formed with ONE fma on the device and on the host (nbody_hip_slab_layer_owner), so
// that both sides and all ranks assign every layer alike.  Monotone in z: a rank's layers are contiguous.
struct SlabCuts {
  float z[kMaxRanks];
  int n;  // world - 1 cuts, ascending; < 0: equal layer counts (slab_owner)
};
__host__ __device__ inline int slab_owner_cuts(int z, float lo_z, float cell, const SlabCuts& c) {
  const float zc = fmaf((float)z + 0.5f, cell, lo_z);
  int r = 0;
  for (int k = 0; k < c.n; k++) r += c.z[k] <= zc ? 1 : 0;
  return r;
}

// block b works on the contiguous chunk [b chunk, (b+1) chunk) in rounds of 256 bodies
__global__ __launch_bounds__(kBlock) void slab_count_kernel(const float4* __restrict__ posm, int n, int chunk,
                                                            const float* __restrict__ gbox, float cell, int world,
                                                            int hist_cap, unsigned char* __restrict__ dest,
                                                            int* __restrict__ layer_of, int* __restrict__ block_counts,
                                                            int* __restrict__ hist, int* __restrict__ info,
                                                            const SlabCuts cuts) {
  __shared__ int cnt[kMaxRanks];
  __shared__ int lhist[kHistLds];  // the block's own histogram of the first kHistLds layers: a few hundred
                                   // global atomics per block instead of one per body
  const SlabGrid g = slab_grid(gbox, cell);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    info[0] = g.gx; info[1] = g.gy; info[2] = g.gz;
    info[3] = g.gz > hist_cap ? 1 : 0;  // the layer histogram does not cover the grid: the host falls back
  }
  const int nh = min(min(hist_cap, g.gz), kHistLds);
  if (threadIdx.x < kMaxRanks) cnt[threadIdx.x] = 0;
  for (int k = threadIdx.x; k < nh; k += kBlock) lhist[k] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
  for (int i0 = lo; i0 < hi; i0 += kBlock) {
    const int i = i0 + threadIdx.x;
    int z = -1, d = -1;
    if (i < hi) {
      z = slab_layer(posm[i].z, g, cell);
      d = cuts.n >= 0 ? slab_owner_cuts(z, g.lo_z, cell, cuts) : slab_owner(z, g.gz, world);
      dest[i] = (unsigned char)d;
      layer_of[i] = z;
      if (z < nh) atomicAdd(&lhist[z], 1);
      else if (z < hist_cap) atomicAdd(&hist[z], 1);  // grids taller than the LDS histogram: rare
    }
    for (int k = 0; k < world; k++) {  // bodies per owner: one LDS atomic per wave and owner
      const unsigned long long m = __ballot(d == k);
      if (lane == 0 && m) atomicAdd(&cnt[k], __popcll(m));
    }
  }
  __syncthreads();
  if (threadIdx.x < world) block_counts[blockIdx.x * world + threadIdx.x] = cnt[threadIdx.x];
  for (int k = threadIdx.x; k < nh; k += kBlock)
    if (lhist[k]) atomicAdd(&hist[k], lhist[k]);
}

// one wave per owner d: exclusive offsets of the blocks inside group d (wave scan over the <= 256 block
// counts), the group's total, and entry d of row `rank` of the W x W send matrix
__global__ __launch_bounds__(64) void slab_scan_kernel(int* __restrict__ block_counts, int nblocks, int world, int rank,
                                                       int* __restrict__ totals, int* __restrict__ send_row) {
  const int d = blockIdx.x, lane = threadIdx.x;
  int run = 0;
  for (int b0 = 0; b0 < nblocks; b0 += 64) {
    const int b = b0 + lane;
    const int c = b < nblocks ? block_counts[b * world + d] : 0;
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    if (b < nblocks) block_counts[b * world + d] = run + incl - c;  // exclusive offset of block b in group d
    run += __shfl(incl, 63, 64);
  }
  if (lane == 0) {
    totals[d] = run;
    send_row[rank * world + d] = run;
  }
}

__global__ __launch_bounds__(kBlock) void slab_leavers_kernel(
    const float4* __restrict__ posm, const float4* __restrict__ vel, const float4* __restrict__ acc,
    const int* __restrict__ gid, int n, int chunk, int world, int rank, const unsigned char* __restrict__ dest,
    const int* __restrict__ layer_of, const int* __restrict__ block_offsets, const int* __restrict__ totals,
    float4* __restrict__ rows, int* __restrict__ holes) {
  __shared__ int running[kMaxRanks + 1];  // per new owner: next row of this block; [world]: next hole entry
  __shared__ int wave_cnt_all[2][4][kMaxRanks + 1];  // by round parity: a round's counts are still being added
                                                     // to `running` while the next round's are written
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if ((int)threadIdx.x <= world) {
    // rows: groups follow each other in owner order, the own group left out; holes: all leavers in index order
    const int d = threadIdx.x;
    int base = 0;
    if (d < world) {
      for (int k = 0; k < d; k++) base += k == rank ? 0 : totals[k];
      base += block_offsets[blockIdx.x * world + d];
    } else {
      for (int k = 0; k < world; k++) base += k == rank ? 0 : block_offsets[blockIdx.x * world + k];
    }
    running[d] = base;
  }
  const int lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
  int round = 0;
  for (int i0 = lo; i0 < hi; i0 += kBlock, round++) {
    int (*wave_cnt)[kMaxRanks + 1] = wave_cnt_all[round & 1];
    const int i = i0 + threadIdx.x;
    const int d = i < hi ? (int)dest[i] : -1;
    const bool leaves = d >= 0 && d != rank;
    const unsigned long long any = __ballot(leaves);
    // rank of this body among the leavers of its wave with the same new owner, and among all its leavers
    int my_rank = 0;
    const int hole_rank = __popcll(any & ((1ull << lane) - 1ull));
    if (any) {
      for (int k = 0; k < world; k++) {
        const unsigned long long m = __ballot(leaves && d == k);
        if (leaves && d == k) my_rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[w][k] = __popcll(m);
      }
    } else if (lane == 0) {
      for (int k = 0; k < world; k++) wave_cnt[w][k] = 0;
    }
    if (lane == 0) wave_cnt[w][world] = __popcll(any);
    __syncthreads();  // also orders the first round after the initialisation of `running`
    if (leaves) {
      int at = running[d] + my_rank, hat = running[world] + hole_rank;
      for (int ww = 0; ww < w; ww++) { at += wave_cnt[ww][d]; hat += wave_cnt[ww][world]; }
      float4* r = rows + (size_t)at * 4;
      r[0] = posm[i];
      const float4 v = vel[i], a = acc[i];
      r[1] = make_float4(v.x, v.y, v.z, 0.f);
      r[2] = make_float4(a.x, a.y, a.z, 0.f);
      r[3] = make_float4(__int_as_float(gid ? gid[i] : i), __int_as_float(layer_of[i]), 0.f, 0.f);
      holes[hat] = i;
    }
    __syncthreads();
    if ((int)threadIdx.x <= world) {
      int add = 0;
      for (int ww = 0; ww < 4; ww++) add += wave_cnt[ww][threadIdx.x];
      running[threadIdx.x] += add;
    }
    // the next round's ballots do not touch `running`; its barrier publishes the update
  }
}

__device__ __forceinline__ int lower_bound_int(const int* __restrict__ a, int n, int v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ void unpack_row(const float4* __restrict__ r, int at, float4* __restrict__ posm,
                                           float4* __restrict__ vel, float4* __restrict__ acc, int* __restrict__ gid) {
  posm[at] = r[0];
  vel[at] = r[1];
  acc[at] = r[2];
  if (gid) gid[at] = __float_as_int(r[3].x);
}

// n_old bodies with L vacated slots (holes, ascending) and A arrivals -> n_new = n_old - L + A bodies in
// [0, n_new).  Thread t < A: arrival t goes into hole t, or behind the old end when the holes are used up.
// Thread A + j (A < L only): the j-th body of the old tail [n_new, n_old) that is not itself a hole moves
// into hole A + j (all of those lie below n_new: the holes are ascending and exactly L - A slots must close).
__global__ __launch_bounds__(kBlock) void slab_fill_kernel(const float4* __restrict__ rows, int A,
                                                           const int* __restrict__ holes, int L, int n_old,
                                                           float4* __restrict__ posm, float4* __restrict__ vel,
                                                           float4* __restrict__ acc, int* __restrict__ gid) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t < A) {
    unpack_row(rows + (size_t)t * 4, t < L ? holes[t] : n_old + (t - L), posm, vel, acc, gid);
    return;
  }
  if (A >= L) return;
  const int n_new = n_old - L + A;
  const int p = n_new + (t - A);  // a slot of the old tail
  if (p >= n_old) return;
  const int kb = lower_bound_int(holes, L, n_new);  // holes[kb..] lie in the tail
  const int kp = lower_bound_int(holes, L, p);
  if (kp < L && holes[kp] == p) return;             // a hole itself: nothing to move
  const int j = (p - n_new) - (kp - kb);            // rank among the tail's bodies
  const int to = holes[A + j];
  posm[to] = posm[p];
  vel[to] = vel[p];
  acc[to] = acc[p];
  if (gid) gid[to] = gid[p];
}

}  // namespace nbh

using namespace nbh;

extern "C" int nbody_hip_slab_layer_owner(int layer, float lo_z, float cell_size, int world, const float* z_cuts) {
  if (!z_cuts || world < 1 || world > kMaxRanks) return -1;
  SlabCuts c;
  c.n = world - 1;
  for (int k = 0; k < c.n; k++) c.z[k] = z_cuts[k];
  return slab_owner_cuts(layer, lo_z, cell_size, c);
}

extern "C" int nbody_hip_slab_partition(nbody_hip_ctx* ctx, const nbody_float4* posm, const nbody_float4* vel,
                                        const nbody_float4* acc, const int* gid, size_t n, const float* gbox_dev,
                                        float cell_size, int world, int rank, int hist_cap, float* rows_out,
                                        int* holes_out, int* send_matrix_dev, int* hist_dev, int* info_dev) {
  return nbody_hip_slab_partition_cuts(ctx, posm, vel, acc, gid, n, gbox_dev, cell_size, world, rank, hist_cap, rows_out,
                                       holes_out, send_matrix_dev, hist_dev, info_dev, nullptr);
}

extern "C" int nbody_hip_slab_partition_cuts(nbody_hip_ctx* ctx, const nbody_float4* posm, const nbody_float4* vel,
                                             const nbody_float4* acc, const int* gid, size_t n, const float* gbox_dev,
                                             float cell_size, int world, int rank, int hist_cap, float* rows_out,
                                             int* holes_out, int* send_matrix_dev, int* hist_dev, int* info_dev,
                                             const float* z_cuts) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (!gbox_dev || !send_matrix_dev || !hist_dev || !info_dev) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "world must be in [1, %d] and rank inside it", kMaxRanks);
  if (!(cell_size > 0.0f) || !(cell_size < INFINITY))
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Spatial hash cell size must be positive and finite");
  if (hist_cap < 1) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "hist_cap must be positive");
  if (n > 0x3fffffffu) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "body count out of range");
  if (n > 0 && (!posm || !vel || !acc)) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (n > 0 && world > 1 && (!rows_out || !holes_out)) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  NBH_NOT_CAPTURABLE(ctx, "the slab partition");
  SlabCuts cuts;
  cuts.n = -1;
  if (z_cuts) {
    cuts.n = world - 1;
    for (int k = 0; k < cuts.n; k++) {
      if (k > 0 && !(z_cuts[k] >= z_cuts[k - 1])) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "slab cuts must ascend");
      cuts.z[k] = z_cuts[k];
    }
  }
  NBH_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int ni = (int)n;
  // <= 256 blocks, each a contiguous chunk (a multiple of the block size, so rounds are full)
  int nblocks = (ni + kBlock - 1) / kBlock;
  if (nblocks > 256) nblocks = 256;
  if (nblocks < 1) nblocks = 1;
  int chunk = (ni + nblocks - 1) / nblocks;
  chunk = (chunk + kBlock - 1) / kBlock * kBlock;
  if (chunk < kBlock) chunk = kBlock;
  // workspace: dest [n bytes] | layer [n ints] | block_counts [nblocks * world] | totals [world]
  const size_t off_layer = (n + 255) & ~(size_t)255;
  const size_t off_bc = off_layer + ((n * sizeof(int) + 255) & ~(size_t)255);
  const size_t off_gb = off_bc + (((size_t)nblocks * world * sizeof(int) + 255) & ~(size_t)255);
  if (int rc = ctx->partial.reserve(off_gb + (kMaxRanks + 1) * sizeof(int))) return rc;
  char* ws = static_cast<char*>(ctx->partial.ptr);
  unsigned char* dest = reinterpret_cast<unsigned char*>(ws);
  int* layer_of = reinterpret_cast<int*>(ws + off_layer);
  int* block_counts = reinterpret_cast<int*>(ws + off_bc);
  int* totals = reinterpret_cast<int*>(ws + off_gb);
  if (hist_dev == send_matrix_dev + (size_t)world * world) {  // one block (the sharded hosts'): one fill
    NBH_HIP(hipMemsetAsync(send_matrix_dev, 0, ((size_t)world * world + hist_cap) * sizeof(int), st));
  } else {
    NBH_HIP(hipMemsetAsync(send_matrix_dev, 0, (size_t)world * world * sizeof(int), st));
    NBH_HIP(hipMemsetAsync(hist_dev, 0, (size_t)hist_cap * sizeof(int), st));
  }
  hipLaunchKernelGGL(slab_count_kernel, dim3(nblocks), dim3(kBlock), 0, st, reinterpret_cast<const float4*>(posm), ni,
                     chunk, gbox_dev, cell_size, world, hist_cap, dest, layer_of, block_counts, hist_dev, info_dev, cuts);
  NBH_LAUNCH_CHECK();
  hipLaunchKernelGGL(slab_scan_kernel, dim3(world), dim3(64), 0, st, block_counts, nblocks, world, rank, totals,
                     send_matrix_dev);
  NBH_LAUNCH_CHECK();
  if (ni > 0 && world > 1) {
    hipLaunchKernelGGL(slab_leavers_kernel, dim3(nblocks), dim3(kBlock), 0, st, reinterpret_cast<const float4*>(posm),
                       reinterpret_cast<const float4*>(vel), reinterpret_cast<const float4*>(acc), gid, ni, chunk,
                       world, rank, dest, layer_of, block_counts, totals, reinterpret_cast<float4*>(rows_out),
                       holes_out);
    NBH_LAUNCH_CHECK();
  }
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_slab_fill(nbody_hip_ctx* ctx, const float* rows, size_t n_arrivals, const int* holes,
                                   size_t n_holes, size_t n_old, nbody_float4* posm, nbody_float4* vel,
                                   nbody_float4* acc, int* gid) {
  if (!ctx) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null context");
  if (n_arrivals == 0 && n_holes == 0) return NBODY_HIP_OK;
  if (!posm || !vel || !acc || (n_arrivals && !rows) || (n_holes && !holes))
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (n_holes > n_old || n_old + n_arrivals > 0x3fffffffu)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "body count out of range");
  NBH_HIP(hipSetDevice(ctx->device));
  const size_t threads = n_arrivals >= n_holes ? n_arrivals : n_holes;  // A arrivals + the L - A slots of the old tail
  hipLaunchKernelGGL(slab_fill_kernel, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                     reinterpret_cast<const float4*>(rows), (int)n_arrivals, holes, (int)n_holes, (int)n_old,
                     reinterpret_cast<float4*>(posm), reinterpret_cast<float4*>(vel), reinterpret_cast<float4*>(acc),
                     gid);
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}
