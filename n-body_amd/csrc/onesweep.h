// SPDX-License-Identifier: MIT
// A host driver of our own around rocPRIM's Onesweep radix-sort DEVICE functions (ROCm 7.x:
// rocprim/device/detail/device_radix_sort.hpp: onesweep_histograms, onesweep_scan_histograms, onesweep_iteration).
//
// Why: rocprim::radix_sort_pairs issues, per digit pass, one hipMemsetAsync for the decoupled-look-back states and one
// for the ordered block id, plus one for the digit histograms: 13 fill launches around the 8 kernels of a 60-bit sort,
// each ~5-6 us on the stream (rocprofv3: 15 fills = 93 us of the 0.49 ms Barnes-Hut build at N = 2^20; 5 of the
// spatial hash's 0.32 ms build).  Here every pass has its OWN look-back states and block-id counter, laid out behind
// the histograms in one block that ONE fill clears: 13 launches -> 1.  The kernels, their tile shapes (histogram
// 256 x 12, pass 1024 x 8, `match` ranking) and therefore the sorted output are rocPRIM's.
//
// Passes alternate between a temporary pair of arrays and the output so that the last one lands in the output
// (in -> tmp -> out -> tmp -> ... -> out), as rocPRIM's own driver does.  Stable.  n < 2^30.
//
// Two more launches go when the caller helps: `cleared` (it zeroed the look-back block itself, e.g. in the kernel that
// writes the keys) and `hist` (it accumulated the digit histograms there too: DigitHistogram below).
//
// DEPENDENCY FENCE (round 4).  onesweep_histograms / onesweep_scan_histograms / onesweep_iteration, onesweep_lookback_state
// and block_id_wrapper live in rocprim::detail -- a private namespace with no compatibility promise -- and this driver
// re-states the temporary-storage layout their decoupled look-back expects.  So:
//   * compile time: the driver exists only for the rocPRIM version it was written and tested against
//     (NBH_ONESWEEP_TESTED_ROCPRIM = 4.2.0 = ROCm 7.2); with any other version NBH_ONESWEEP_AVAILABLE is 0 and the callers
//     (barnes_hut.hip, spatial_hash.hip) compile to the public rocprim::radix_sort_pairs alone.  -DNBH_ONESWEEP_FORCE=1
//     overrides the check for someone porting it to a newer rocPRIM (the run-time test below still applies);
//   * run time: the first tree / grid created in a process sorts one buffer with both paths and compares every output
//     word (sort_self_test in the two callers; ~1 ms once); a mismatch switches the process to the public sort and
//     says so on stderr (nbody_hip_sort_info reports which sort runs);
//   * tests: tests/test_sort_gpu.py compares the two paths' permutations and runs trees and grids on both.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <iterator>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/rocprim_version.hpp>

#define NBH_ONESWEEP_TESTED_ROCPRIM 400200
#ifndef NBH_ONESWEEP_FORCE
#define NBH_ONESWEEP_FORCE 0
#endif
#if NBH_ONESWEEP_FORCE || (defined(ROCPRIM_VERSION) && ROCPRIM_VERSION == NBH_ONESWEEP_TESTED_ROCPRIM)
#define NBH_ONESWEEP_AVAILABLE 1
#else
#define NBH_ONESWEEP_AVAILABLE 0
#endif

namespace nbh {
namespace onesweep {

// process-wide verdict of the run-time self-tests: 0 = not run yet, 1 = the driver's output equals the public sort's,
// 2 = it does not (or the test could not run): the driver is not used
inline std::atomic<int>& self_test_state() {
  static std::atomic<int> s{0};
  return s;
}
inline bool usable() { return NBH_ONESWEEP_AVAILABLE && self_test_state().load(std::memory_order_acquire) != 2; }
inline void self_test_report(bool same, const char* who) {
  int expect = 0;
  if (same) {
    self_test_state().compare_exchange_strong(expect, 1);
  } else {
    self_test_state().store(2, std::memory_order_release);
    std::fprintf(stderr, "libnbody_hip: the Onesweep driver (csrc/onesweep.h, rocPRIM internals of version %d) does not reproduce "
                         "rocprim::radix_sort_pairs in the %s self-test; the public sort is used instead\n",
                 (int)ROCPRIM_VERSION, who);
  }
}

constexpr unsigned kHistBlock = 256, kHistItems = 12;
constexpr unsigned kPassBlock = 1024, kPassItems = 8;

#if NBH_ONESWEEP_AVAILABLE
using BlockId = rocprim::detail::block_id_wrapper<unsigned int, true>;
using Lookback = rocprim::detail::onesweep_lookback_state;

template <unsigned RB, class Key>
__global__ __launch_bounds__(kHistBlock) void hist_kernel(const Key* keys, unsigned int* counts, unsigned int size,
                                                           unsigned int full_blocks, unsigned int begin_bit,
                                                           unsigned int end_bit) {
  rocprim::detail::onesweep_histograms<kHistBlock, kHistItems, RB, false>(keys, counts, size, full_blocks,
                                                                          rocprim::identity_decomposer{}, begin_bit, end_bit);
}

// copies > 1: the caller accumulated the counts in `copies` replicas `copy_stride` words apart (DigitHistogram::flush
// spreads the workgroups over them: global atomics on ONE word serialise, 512 workgroups on one bin are a 15 us chain);
// they are folded into replica 0 first
template <unsigned RB>
__global__ __launch_bounds__(kHistBlock) void scan_kernel(unsigned int* counts, int copies, unsigned int copy_stride) {
  if (copies > 1) {
    constexpr unsigned int R = 1u << RB;
    unsigned int* mine = counts + blockIdx.x * R;
    constexpr int kPer = (R + kHistBlock - 1) / kHistBlock, kMaxCopies = 8;
    unsigned int v[kPer][kMaxCopies];  // every load in flight before the first add
#pragma unroll
    for (int k = 0; k < kPer; k++)
#pragma unroll
      for (int c = 0; c < kMaxCopies; c++) {
        const unsigned int d = threadIdx.x + k * kHistBlock;
        v[k][c] = (c < copies && d < R) ? mine[(size_t)c * copy_stride + d] : 0u;
      }
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      unsigned int sum = 0;
#pragma unroll
      for (int c = 0; c < kMaxCopies; c++) sum += v[k][c];
      const unsigned int d = threadIdx.x + k * kHistBlock;
      if (d < R) mine[d] = sum;
    }
    __threadfence_block();
    __syncthreads();
  }
  rocprim::detail::onesweep_scan_histograms<kHistBlock, RB>(counts);
}

template <unsigned RB, class KI, class KO, class VI, class VO>
__global__ __launch_bounds__(kPassBlock) void pass_kernel(KI keys_in, KO keys_out, VI vals_in, VO vals_out,
                                                           unsigned int size, unsigned int* offs_in,
                                                           unsigned int* offs_out, Lookback* lookback, unsigned int bit,
                                                           unsigned int cur_bits, unsigned int full_blocks, BlockId bid) {
  rocprim::detail::onesweep_iteration<kPassBlock, kPassItems, RB, false, rocprim::block_radix_rank_algorithm::match>(
      keys_in, keys_out, vals_in, vals_out, size, offs_in, offs_out, lookback, rocprim::identity_decomposer{}, bit,
      cur_bits, full_blocks, bid);
}

#endif  // NBH_ONESWEEP_AVAILABLE

// Digit histograms of up to PLACES digit places in LDS, for a kernel that has the keys in registers anyway.
//   __shared__ unsigned int h[PLACES << RB];  zero(h); __syncthreads(); ... add(h, key >> begin_bit, places) per key ...
//   __syncthreads(); flush(h, places, ..., global_counts, copy_stride)   -- only the bins that are not empty
template <unsigned RB>
struct DigitHistogram {
  static constexpr unsigned int R = 1u << RB;
  template <class Key>
  static __device__ __forceinline__ void add(unsigned int* h, Key shifted_key, int places) {
    for (int p = 0; p < places; p++) {
      atomicAdd(&h[p * R + (unsigned int)(shifted_key & (Key)(R - 1))], 1u);
      shifted_key >>= RB;
    }
  }
  static __device__ __forceinline__ void zero(unsigned int* h, int places, int tid, int nthreads) {
    for (int t = tid; t < places * (int)R; t += nthreads) h[t] = 0u;
  }
  // global_counts: kCopies replicas copy_stride words apart; workgroup b adds to replica b % kCopies
  static constexpr int kCopies = 8;  // (scan_kernel folds at most 8)
  static __device__ __forceinline__ void flush(const unsigned int* h, int places, int tid, int nthreads,
                                               unsigned int* __restrict__ global_counts, unsigned int copy_stride) {
    unsigned int* dst = global_counts + (size_t)(blockIdx.x % kCopies) * copy_stride;
    for (int t = tid; t < places * (int)R; t += nthreads)
      if (h[t]) atomicAdd(&dst[t], h[t]);
  }
};

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

// 32-bit words at the front of the temporary storage that must be ZERO when the sort starts (digit histograms,
// carry, block ids, look-back states of every pass): sort_pairs clears them itself unless the caller says it already
// has (`cleared`: e.g. inside the kernel that writes the keys -- one launch less)
template <unsigned RB>
inline size_t clear_words(size_t n, unsigned int begin_bit, unsigned int end_bit) {
  constexpr unsigned int R = 1u << RB;
  const size_t places = (end_bit - begin_bit + RB - 1) / RB;
  const size_t blocks = (n + (size_t)kPassBlock * kPassItems - 1) / ((size_t)kPassBlock * kPassItems);
  return places * R + R + places + places * blocks * R;
}

#if NBH_ONESWEEP_AVAILABLE
// tmp == nullptr: size query.  keys_in may not alias the temporary arrays; vals_in / vals_out may be any iterators
// whose value type is VT (the temporary value array holds VT).
template <unsigned RB, class Key, class VI, class VO>
hipError_t sort_pairs(void* tmp, size_t& tmp_bytes, const Key* keys_in, Key* keys_out, VI vals_in, VO vals_out, size_t n,
                      unsigned int begin_bit, unsigned int end_bit, hipStream_t st, bool cleared = false,
                      unsigned int* hist = nullptr, int hist_copies = 1, unsigned int hist_copy_stride = 0) {
  using VT = typename std::iterator_traits<VI>::value_type;
  constexpr unsigned int R = 1u << RB;
  if (n >= (1u << 30) || end_bit <= begin_bit) return hipErrorInvalidValue;
  const unsigned int size = (unsigned int)n;
  const unsigned int places = (end_bit - begin_bit + RB - 1) / RB;
  const unsigned int pass_items = kPassBlock * kPassItems, hist_items = kHistBlock * kHistItems;
  const unsigned int blocks = (size + pass_items - 1) / pass_items;
  const unsigned int hblocks = (size + hist_items - 1) / hist_items;
  // [ counts: places x R | carry: R | block ids: places | look-back: places x blocks x R ]  <- one fill
  const size_t words = clear_words<RB>(n, begin_bit, end_bit);
  const size_t off_keys = align_up(words * sizeof(unsigned int));
  const size_t off_vals = off_keys + align_up(n * sizeof(Key));
  const size_t total = off_vals + align_up(n * sizeof(VT));
  if (!tmp) {
    tmp_bytes = total;
    return hipSuccess;
  }
  if (tmp_bytes < total) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  char* base = static_cast<char*>(tmp);
  unsigned int* counts = reinterpret_cast<unsigned int*>(base);  // (left unused when the caller brings the histograms)
  unsigned int* carry = counts + (size_t)places * R;
  unsigned int* ids = carry + R;
  Lookback* lookback = reinterpret_cast<Lookback*>(ids + places);
  Key* keys_tmp = reinterpret_cast<Key*>(base + off_keys);
  VT* vals_tmp = reinterpret_cast<VT*>(base + off_vals);
  if (!cleared) {
    hipError_t e = hipMemsetAsync(base, 0, words * sizeof(unsigned int), st);
    if (e != hipSuccess) return e;
  }
  // hist: places x R digit counts the caller accumulated while it wrote the keys (count of digit d of place p at
  // [p * R + d], digit = (key >> (begin_bit + p * RB)) & (R - 1)): rocPRIM's histogram kernel flushes every bin of
  // every 3,072-key tile with a global atomic (2.8 M atomics = 43 us at 4.2 M 32-bit keys) and is skipped then
  if (hist) {
    counts = hist;
  } else {
    hipLaunchKernelGGL((hist_kernel<RB, Key>), dim3(hblocks), dim3(kHistBlock), 0, st, keys_in, counts, size,
                       size % hist_items == 0 ? hblocks : hblocks - 1, begin_bit, end_bit);
  }
  hipLaunchKernelGGL((scan_kernel<RB>), dim3(places), dim3(kHistBlock), 0, st, counts, hist ? hist_copies : 1,
                     hist_copy_stride);
  const unsigned int full_blocks = size % pass_items == 0 ? blocks : blocks - 1;
  bool to_output = (places - 1) % 2 == 0;
  bool from_input = true;
  unsigned int bit = begin_bit;
  for (unsigned int place = 0; place < places; place++, bit += RB) {
    const unsigned int cur = end_bit - bit < RB ? end_bit - bit : RB;
    unsigned int* offs = counts + (size_t)place * R;
    Lookback* lb = lookback + (size_t)place * blocks * R;
    BlockId bid = BlockId::create(ids + place);
    if (from_input && to_output) {
      hipLaunchKernelGGL((pass_kernel<RB, const Key*, Key*, VI, VO>), dim3(blocks), dim3(kPassBlock), 0, st, keys_in,
                         keys_out, vals_in, vals_out, size, offs, carry, lb, bit, cur, full_blocks, bid);
    } else if (from_input) {
      hipLaunchKernelGGL((pass_kernel<RB, const Key*, Key*, VI, VT*>), dim3(blocks), dim3(kPassBlock), 0, st, keys_in,
                         keys_tmp, vals_in, vals_tmp, size, offs, carry, lb, bit, cur, full_blocks, bid);
    } else if (to_output) {
      hipLaunchKernelGGL((pass_kernel<RB, const Key*, Key*, const VT*, VO>), dim3(blocks), dim3(kPassBlock), 0, st,
                         static_cast<const Key*>(keys_tmp), keys_out, static_cast<const VT*>(vals_tmp), vals_out, size,
                         offs, carry, lb, bit, cur, full_blocks, bid);
    } else {  // out -> tmp: the output arrays are read back, so they must be readable through their iterators
      hipLaunchKernelGGL((pass_kernel<RB, const Key*, Key*, VO, VT*>), dim3(blocks), dim3(kPassBlock), 0, st,
                         static_cast<const Key*>(keys_out), keys_tmp, vals_out, vals_tmp, size, offs, carry, lb, bit, cur,
                         full_blocks, bid);
    }
    from_input = false;
    to_output = !to_output;
  }
  return hipGetLastError();
}
#else   // another rocPRIM: no driver; the callers never reach these (usable() is false), they only have to compile
template <unsigned RB, class Key, class VI, class VO>
hipError_t sort_pairs(void*, size_t& tmp_bytes, const Key*, Key*, VI, VO, size_t, unsigned int, unsigned int, hipStream_t,
                      bool = false, unsigned int* = nullptr, int = 1, unsigned int = 0) {
  tmp_bytes = 0;
  return hipErrorNotSupported;
}
#endif  // NBH_ONESWEEP_AVAILABLE

}  // namespace onesweep
}  // namespace nbh
