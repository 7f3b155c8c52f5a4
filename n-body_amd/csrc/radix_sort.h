// SPDX-License-Identifier: MIT
// radix_sort.h -- a hand-written stable LSD radix sort of (key, body, index) for gfx950: the sort that bins the bodies
// of the Barnes-Hut build (63-bit Morton keys, index payload) and of the spatial-hash build (cell ids, float4 body +
// index payload).  The reference calls thrust::sort_by_key there (ref: src/cuda/force_barnes_hut.cu:276-280,
// force_spatial_hash.cu:286-288); rounds 2-3 used rocPRIM (public API below the crossover sizes, its Onesweep device
// functions under our own driver above: onesweep.h, fenced in round 4).  This file owes rocPRIM nothing: no include, no
// internal layout; it is what runs when the rocPRIM version is not the one the driver was written against, and it is
// selectable (NBH_SORT=own) everywhere for A/B runs.
//
// One pass per 10-bit digit, "Onesweep" shape (Adinets & Merrill 2022): the digit histograms of ALL places are known up
// front (counted by the kernel that writes the keys, or by hist_kernel), so a pass is ONE kernel -- every workgroup
// ranks its tile, learns how many keys with each digit precede the tile from its predecessors' published counts
// (decoupled look-back) and scatters.  Per workgroup (1,024 threads, 8 keys each, tile = 8,192 keys):
//   1. ticket: the tile number is drawn from a counter when the workgroup starts, so every predecessor tile is already
//      running: the look-back cannot wait for a workgroup that has not been scheduled;
//   2. keys in wave-striped order (coalesced); per item a wave finds the lanes with the same digit by ten ballots
//      (mask arithmetic on the scalar unit), rank = lanes below in the mask; the first lane of every group advances
//      the wave's LDS counter of that digit -- ranks are in key order: STABLE;
//   3. per digit: prefix over the sixteen waves' counters, workgroup count; exclusive scan over the digits = the tile's
//      local order;
//   4. look-back, one thread per two digits: publish {PARTIAL, count} in ONE 32-bit word (flag in the top two bits, so
//      value and flag travel together: no fence), walk the predecessors adding PARTIAL counts until an INCLUSIVE one,
//      publish {INCLUSIVE, sum}.  Agent-scope relaxed atomics on both sides (MI355X_MICROARCH.md: a one-word granule
//      needs no ordering; polls bypass the L1).  Every spin is bounded (2^22 polls with s_sleep): on expiry the kernel
//      raises a word in mapped host memory and finishes -- the host refuses the result;
//   5. keys to their local sorted position in LDS with their tile-local source index; written out position by
//      position (runs of one digit are contiguous: coalesced), bodies and indices GATHERED from the tile's source range
//      (64 KB, cache resident) straight to the output -- the payload never passes through LDS.
// Cost: ~14 VALU per key per pass; the passes are bound by their 24-40 bytes per key of traffic.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>

namespace nbh {
namespace radix {

constexpr unsigned kBits = 10, kBins = 1u << kBits;
#ifndef NBH_RADIX_BLOCK
#define NBH_RADIX_BLOCK 1024  // (1,024 x 8 = 8,192-key tiles: 0.21 ms for the two passes of 4.2 M (key, body, index), 0.25 with 512 x 16, 0.34 with 512 x 8)
#endif
#ifndef NBH_RADIX_ITEMS
#define NBH_RADIX_ITEMS 8
#endif
constexpr unsigned kBlock = NBH_RADIX_BLOCK, kItems = NBH_RADIX_ITEMS, kTile = kBlock * kItems, kWaves = kBlock / 64;
static_assert(kBins % kBlock == 0 || kBlock % kBins == 0, "digits per thread");
constexpr unsigned kMaxPlaces = 7;  // 63 key bits
constexpr unsigned kFlagPartial = 1u << 30, kFlagInclusive = 2u << 30, kValueMask = (1u << 30) - 1u;
constexpr unsigned kSpinLimit = 1u << 22;

// digit counts of every place in one pass over the keys (LDS counters, one global atomic per non-empty bin and
// workgroup).  counts: [places][kBins], zeroed by the caller.
template <class Key>
__global__ __launch_bounds__(256) void hist_kernel(const Key* __restrict__ keys, unsigned n, unsigned begin_bit,
                                                   unsigned end_bit, unsigned places, unsigned* __restrict__ counts) {
  __shared__ unsigned h[kMaxPlaces * kBins];
  for (unsigned t = threadIdx.x; t < places * kBins; t += 256) h[t] = 0u;
  __syncthreads();
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    Key k = keys[i] >> begin_bit;
    if (end_bit - begin_bit < sizeof(Key) * 8) k &= (((Key)1) << (end_bit - begin_bit)) - (Key)1;  // (bits above the range do not sort)
    for (unsigned p = 0; p < places; p++) {
      atomicAdd(&h[p * kBins + (unsigned)(k & (Key)(kBins - 1))], 1u);
      k >>= kBits;
    }
  }
  __syncthreads();
  for (unsigned t = threadIdx.x; t < places * kBins; t += 256)
    if (h[t]) atomicAdd(&counts[t], h[t]);
}

// exclusive scan of each place's 1,024 counts -> offs[place][digit] = keys with a smaller digit at that place.
// counts may come in `copies` replicas copy_stride words apart (the key kernels spread their atomics): summed first.
static __global__ __launch_bounds__(256) void scan_kernel(const unsigned* __restrict__ counts, int copies, unsigned copy_stride,
                                                   unsigned* __restrict__ offs) {
  __shared__ unsigned wsum[4];
  const unsigned place = blockIdx.x, t = threadIdx.x;
  unsigned v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    unsigned c = 0;
    for (int r = 0; r < copies; r++) c += counts[(size_t)r * copy_stride + place * kBins + 4 * t + k];
    v[k] = s;  // exclusive inside the thread
    s += c;
  }
  unsigned incl = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned u = __shfl_up(incl, off, 64);
    if ((int)(t & 63) >= off) incl += u;
  }
  if ((t & 63) == 63) wsum[t >> 6] = incl;
  __syncthreads();
  unsigned before = 0;
  for (unsigned w = 0; w < (t >> 6); w++) before += wsum[w];
  const unsigned ex = before + incl - s;
#pragma unroll
  for (int k = 0; k < 4; k++) offs[place * kBins + 4 * t + k] = ex + v[k];
}

// One digit pass.  idx_in == nullptr: the payload index of key i is i (first pass of a sort by body number).
template <class Key, bool BODY>
__global__ __launch_bounds__(kBlock) void pass_kernel(const Key* __restrict__ keys_in, Key* __restrict__ keys_out,
                                                      const float4* __restrict__ body_in, float4* __restrict__ body_out,
                                                      const int* __restrict__ idx_in, int* __restrict__ idx_out, unsigned n,
                                                      unsigned bit, unsigned cur_bits, const unsigned* __restrict__ digit_offs,
                                                      unsigned* __restrict__ lookback, unsigned* __restrict__ ticket,
                                                      unsigned* __restrict__ error_host) {
  __shared__ unsigned short wave_cnt[kWaves][kBins];  // (<= 512 keys of a wave share a digit: 16 bits)
  __shared__ Key lds_keys[kTile];
  __shared__ unsigned short lds_src[kTile];
  __shared__ unsigned digit_start[kBins], gbase[kBins];
  __shared__ unsigned wtot[kWaves];
  __shared__ unsigned tile_s;
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) tile_s = atomicAdd(ticket, 1u);
  for (unsigned t = tid; t < kWaves * kBins / 2; t += kBlock) reinterpret_cast<unsigned*>(&wave_cnt[0][0])[t] = 0u;
  __syncthreads();
  const unsigned tile = tile_s, base = tile * kTile;
  const unsigned dmask = (1u << cur_bits) - 1u;

  // ---- keys, digits, ranks inside the wave (stable: items in order, lanes in order)
  Key key[kItems];
  unsigned dig[kItems], off[kItems];
  const unsigned wbase = base + wave * (64u * kItems);
#pragma unroll
  for (unsigned i = 0; i < kItems; i++) {
    const unsigned g = wbase + i * 64u + lane;
    key[i] = g < n ? keys_in[g] : (Key)0;
  }
#pragma unroll
  for (unsigned i = 0; i < kItems; i++) {
    const unsigned g = wbase + i * 64u + lane;
    const bool valid = g < n;
    const unsigned d = (unsigned)(key[i] >> bit) & dmask;
    dig[i] = d;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (unsigned b = 0; b < kBits; b++) {
      const unsigned long long set = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? set : ~set;
    }
    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0u));
    const unsigned cnt = (unsigned)__popcll(peers);
    unsigned old = 0;
    if (valid) old = wave_cnt[wave][d];
    off[i] = old + rank;
    __builtin_amdgcn_wave_barrier();  // (every lane has read the counter before the group's first lane advances it)
    if (valid && rank == 0) wave_cnt[wave][d] = (unsigned short)(old + cnt);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __syncthreads();

  // ---- per digit (kDpt consecutive digits per thread): prefix over the waves, workgroup count
  constexpr unsigned kDpt = kBins / kBlock;
  static_assert(kDpt >= 1 && kDpt * kBlock == kBins, "a whole number of digits per thread");
  unsigned bc[kDpt];
#pragma unroll
  for (unsigned k = 0; k < kDpt; k++) {
    const unsigned d = kDpt * tid + k;
    unsigned run = 0;
#pragma unroll
    for (unsigned w = 0; w < kWaves; w++) {
      const unsigned c = wave_cnt[w][d];
      wave_cnt[w][d] = (unsigned short)run;
      run += c;
    }
    bc[k] = run;
  }
  // exclusive scan of the counts over the digits: the tile's local order
  {
    unsigned s = 0;
#pragma unroll
    for (unsigned k = 0; k < kDpt; k++) s += bc[k];
    unsigned incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned u = __shfl_up(incl, o, 64);
      if ((int)lane >= o) incl += u;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned before = 0;
    for (unsigned w = 0; w < wave; w++) before += wtot[w];
    unsigned ex = before + incl - s;
#pragma unroll
    for (unsigned k = 0; k < kDpt; k++) {
      digit_start[kDpt * tid + k] = ex;
      ex += bc[k];
    }
  }
  // ---- decoupled look-back, first half: this tile's counts go out NOW (one word per digit: flag and value together), so
  // that the successors' walks find them while this workgroup is still busy with its own keys
#pragma unroll
  for (unsigned k = 0; k < kDpt; k++)
    __hip_atomic_store(lookback + (size_t)tile * kBins + kDpt * tid + k, (tile == 0 ? kFlagInclusive : kFlagPartial) | bc[k],
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();  // (digit_start)

  // ---- keys to their local sorted position, with where they came from
#pragma unroll
  for (unsigned i = 0; i < kItems; i++) {
    const unsigned local = wave * (64u * kItems) + i * 64u + lane;
    if (base + local < n) {
      const unsigned pos = digit_start[dig[i]] + wave_cnt[wave][dig[i]] + off[i];
      lds_keys[pos] = key[i];
      lds_src[pos] = (unsigned short)local;
    }
  }
  // ---- look-back, second half: how many keys with this digit lie in the tiles before this one (the predecessors have had
  // the time of the scatter above to publish)
#pragma unroll
  for (unsigned k = 0; k < kDpt; k++) {
    const unsigned d = kDpt * tid + k;
    unsigned excl = 0;
    if (tile != 0) {
      unsigned prev = tile - 1u, spins = 0;
      for (;;) {
        const unsigned v = __hip_atomic_load(lookback + (size_t)prev * kBins + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned flag = v & ~kValueMask;
        if (flag == kFlagInclusive) {
          excl += v & kValueMask;
          break;
        }
        if (flag == kFlagPartial) {
          excl += v & kValueMask;
          prev--;  // (tile 0 publishes INCLUSIVE: the walk ends there at the latest)
          continue;
        }
        if (++spins > kSpinLimit) {  // a predecessor never published: say so and finish (the host refuses the result)
          __hip_atomic_store(error_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      __hip_atomic_store(lookback + (size_t)tile * kBins + d, kFlagInclusive | ((excl + bc[k]) & kValueMask), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    gbase[d] = digit_offs[d] + excl;
  }
  __syncthreads();
  // ---- out: position by position (runs of one digit are contiguous in the output)
  const unsigned count = n - base < kTile ? n - base : kTile;
#pragma unroll
  for (unsigned j = 0; j < kItems; j++) {
    const unsigned pos = j * kBlock + tid;
    if (pos < count) {
      const Key k = lds_keys[pos];
      const unsigned d = (unsigned)(k >> bit) & dmask;
      const unsigned dst = gbase[d] + (pos - digit_start[d]);
      const unsigned src = base + lds_src[pos];
      keys_out[dst] = k;
      idx_out[dst] = idx_in ? idx_in[src] : (int)src;
      if constexpr (BODY) body_out[dst] = body_in[src];
    }
  }
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

// process-wide verdict of the run-time self-tests of THIS sort (the callers sort one buffer with it and with the public
// rocprim::radix_sort_pairs when their first tree / grid is made): 0 = not run, 1 = identical output, 2 = not: unused
inline std::atomic<int>& self_test_state() {
  static std::atomic<int> s{0};
  return s;
}
inline bool usable() { return self_test_state().load(std::memory_order_acquire) != 2; }
inline void self_test_report(bool same, const char* who) {
  int expect = 0;
  if (same) {
    self_test_state().compare_exchange_strong(expect, 1);
  } else {
    self_test_state().store(2, std::memory_order_release);
    std::fprintf(stderr, "libnbody_hip: the hand-written radix sort (csrc/radix_sort.h) does not reproduce rocprim::radix_sort_pairs "
                         "in the %s self-test; the public sort is used instead\n", who);
  }
}

struct Layout {
  unsigned places, tiles;
  size_t off_digit, off_ticket, off_lookback, clear_begin, clear_bytes, off_keys, off_body, off_idx, total;
};
template <class Key, bool BODY>
inline Layout layout(size_t n, unsigned begin_bit, unsigned end_bit) {
  Layout L{};
  L.places = (end_bit - begin_bit + kBits - 1) / kBits;
  L.tiles = (unsigned)((n + kTile - 1) / kTile);
  L.off_digit = 0;                                                       // [places][kBins] scanned digit offsets
  L.off_ticket = align_up((size_t)L.places * kBins * 4);                 // [places] tickets ...
  L.off_lookback = L.off_ticket + 256;                                   // ... [places][tiles][kBins] look-back words
  L.clear_begin = L.off_ticket;
  L.clear_bytes = 256 + (size_t)L.places * L.tiles * kBins * 4;
  L.off_keys = align_up(L.off_lookback + (size_t)L.places * L.tiles * kBins * 4);
  L.off_idx = L.off_keys + align_up(n * sizeof(Key));
  L.off_body = L.off_idx + align_up(n * sizeof(int));
  L.total = L.off_body + (BODY ? align_up(n * sizeof(float4)) : 0);
  return L;
}

// Stable sort of keys (bits [begin_bit, end_bit)) carrying (body, index).  tmp == nullptr: size query.  keys_in, body_in,
// idx_in are not modified; idx_in == nullptr: the index of input element i is i.  counts (optional): the digit counts
// [copy][places x kBins] the caller accumulated while it wrote the keys (digit p of key k = (k >> (begin_bit + 10 p)) &
// 1023), in `copies` replicas copy_stride words apart.  hist_scratch: places x kBins words the sort may use for its own
// counts when the caller brings none (inside tmp).  error_host: a word in mapped host memory (device address), raised
// when a look-back gave up; 0 on entry.
template <class Key, bool BODY>
hipError_t sort_pairs(void* tmp, size_t& tmp_bytes, const Key* keys_in, Key* keys_out, const float4* body_in,
                      float4* body_out, const int* idx_in, int* idx_out, size_t n, unsigned begin_bit, unsigned end_bit,
                      hipStream_t st, unsigned* error_host, const unsigned* counts = nullptr, int copies = 1,
                      unsigned copy_stride = 0) {
  if (n >= (1u << 30) || end_bit <= begin_bit || (end_bit - begin_bit + kBits - 1) / kBits > kMaxPlaces) return hipErrorInvalidValue;
  Layout L = layout<Key, BODY>(n, begin_bit, end_bit);
  const size_t own_counts = align_up((size_t)L.places * kBins * 4);
  if (!tmp) {
    tmp_bytes = L.total + own_counts;
    return hipSuccess;
  }
  if (tmp_bytes < L.total + own_counts) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  char* b = static_cast<char*>(tmp);
  unsigned* digit_offs = reinterpret_cast<unsigned*>(b + L.off_digit);
  unsigned* ticket = reinterpret_cast<unsigned*>(b + L.off_ticket);
  unsigned* lookback = reinterpret_cast<unsigned*>(b + L.off_lookback);
  Key* keys_tmp = reinterpret_cast<Key*>(b + L.off_keys);
  int* idx_tmp = reinterpret_cast<int*>(b + L.off_idx);
  float4* body_tmp = BODY ? reinterpret_cast<float4*>(b + L.off_body) : nullptr;
  unsigned* my_counts = reinterpret_cast<unsigned*>(b + L.total);
  hipError_t e = hipMemsetAsync(b + L.clear_begin, 0, L.clear_bytes, st);
  if (e != hipSuccess) return e;
  const unsigned un = (unsigned)n;
  if (!counts) {
    e = hipMemsetAsync(my_counts, 0, (size_t)L.places * kBins * 4, st);
    if (e != hipSuccess) return e;
    unsigned blocks = (un + 256u * 16u - 1u) / (256u * 16u);
    if (blocks > 1024u) blocks = 1024u;
    hipLaunchKernelGGL((hist_kernel<Key>), dim3(blocks), dim3(256), 0, st, keys_in, un, begin_bit, end_bit, L.places, my_counts);
    counts = my_counts;
    copies = 1;
    copy_stride = 0;
  }
  hipLaunchKernelGGL(scan_kernel, dim3(L.places), dim3(256), 0, st, counts, copies, copy_stride, digit_offs);
  // in -> tmp -> out -> tmp -> ... so that the last pass lands in the output
  bool to_output = (L.places - 1) % 2 == 0;
  const Key* kin = keys_in;
  const float4* bin = body_in;
  const int* iin = idx_in;
  unsigned bit = begin_bit;
  for (unsigned p = 0; p < L.places; p++, bit += kBits) {
    const unsigned cur = end_bit - bit < kBits ? end_bit - bit : kBits;
    Key* ko = to_output ? keys_out : keys_tmp;
    float4* bo = to_output ? body_out : body_tmp;
    int* io = to_output ? idx_out : idx_tmp;
    hipLaunchKernelGGL((pass_kernel<Key, BODY>), dim3(L.tiles), dim3(kBlock), 0, st, kin, ko, bin, bo, iin, io, un, bit, cur,
                       digit_offs + (size_t)p * kBins, lookback + (size_t)p * L.tiles * kBins, ticket + p, error_host);
    kin = ko;
    bin = bo;
    iin = io;
    to_output = !to_output;
  }
  return hipGetLastError();
}

}  // namespace radix
}  // namespace nbh
