// barnes_hut.hip -- on-device octree build and wave-cooperative traversal, gfx950.
//
// Replaces BarnesHutTree / BarnesHutCalculator of the reference
// (src/cuda/force_barnes_hut.cu:204-532).  The reference builds its tree on the HOST with
// >= 17 PCIe crossings per step (:324-333, :434, :443-450, :484) and its insertion loop never
// subdivides an occupied leaf (:363-396), so only the CONTRACT is kept:
//   root cube   centre = bbox midpoint, half = max extent / 2 + 0.001           (:339-344)
//   Morton      10 bits per axis, x|y|z interleave                              (:23-38)
//   traversal   skip massless nodes; accept a node if it is a leaf or
//               (2 half)^2 / (d^2 + eps^2) < theta^2; skip the body itself      (:160-195)
// Everything runs on the device with no host round trip:
//   keys  -> stable radix sort (rocPRIM) -> bodies reordered into Morton order (float4)
//   build    level by level from the sorted keys: a body opens a node at level L when its
//            3L-bit key prefix differs from its predecessor's and its level-(L-1) cell holds
//            more than `leaf_max` bodies; node ids come from a prefix scan, so the nodes of a
//            level are in Morton order and the children of a node are CONSECUTIVE
//   monopoles bottom-up per level in fp64 (children summed in octant order)
//   traversal one wave walks the tree for 64 Morton-adjacent bodies with ONE shared stack in
//            LDS.  A stack entry is (first child, child count, 64-bit lane mask): lanes in the
//            mask test the node; lanes that accept it accumulate its monopole and drop out of
//            the mask; if any lane still needs it opened (ballot) its children are pushed with
//            the remaining mask.  Every body therefore gets exactly the interaction list of a
//            private depth-first walk, but node records are wave-uniform (scalar-cache) loads
//            and there is no divergence and no private stack (the reference keeps
//            `int stack[256]` per thread in scratch and drops children when it overflows,
//            :147-153,:186-194).
//
// Roofline: build = HBM streaming + sort (~100 B/body); traversal = latency/L2 bound:
// 32 B per node visited per WAVE (not per body).

#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"
#include "onesweep.h"
#include "radix_sort.h"

using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                             rocprim::default_config, nbh::kSortMergeLimit>;
#ifndef NBH_BH_RADIX_BITS
#define NBH_BH_RADIX_BITS 10
#endif
#ifndef NBH_BH_OWN_SORT
#define NBH_BH_OWN_SORT 1   // 0: rocprim::radix_sort_pairs everywhere (A/B builds)
#endif
#if NBH_BH_RADIX_BITS > 0
// 63-bit keys: digits of NBH_BH_RADIX_BITS bits per onesweep pass instead of the default 8 -- 60 key bits at the
// default depth are six passes instead of eight (build 0.59 -> 0.54 ms at N = 2^20; 11 bits do not fit the LDS of
// rocPRIM's histogram kernel)
using SortConfig64 = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<1024, 8>,
                                        NBH_BH_RADIX_BITS, rocprim::block_radix_rank_algorithm::match>,
    nbh::kSortMergeLimit>;
#else
using SortConfig64 = SortConfig;
#endif

namespace nbh {

#ifndef NBH_BH_XCD
#define NBH_BH_XCD 1
#endif
// pair walk: the fp32 sums of this many sibling groups (1 or 2) are folded into the fp64 totals at a time
#ifndef NBH_BH_FOLD
#define NBH_BH_FOLD 1
#endif

constexpr int kMaxDepth = 21;      // 63-bit Morton keys (the reference caps its insertion at depth 20, :363)
constexpr int kDepth32 = 10;       // up to here 30-bit keys in 32-bit words (the faster sort)
constexpr int kDefaultDepth = 20;  // the depth the reference's insertion loop stops at (:363).  Measured at
                                   // N = 2^20 (tools/bh_depth_sweep.py): build 0.41 -> 0.65 ms against depth 10, walk
                                   // unchanged for the BASELINE bodies -- and 90 -> 3.6 ms for a compact Plummer core
                                   // (a = 0.1), whose depth-10 cells hold up to 1,600 bodies
constexpr int kStack = 8 * (kMaxDepth + 2);
constexpr int kSplitBudget = 2097152;  // capacity of the partial-sum buffer: replicas * n
constexpr int kSplitAuto = 327680;     // automatic choice: replicas * n up to here (= 5120 waves; tools/bh_k_small.py)
constexpr int kMaxReplicas = 16;
constexpr int kPairFrom = 98304;       // bodies from which the walk runs without replicas (tools/bh_split_vs_pair.py)

struct TreeRoot {
  float lo[3];
  float half;
  float scale;  // 1024 / (2 half)
  int pad[3];
};

// root cube from the bounding box (enc = order-preserving encodings of {min x, y, z, max x, y, z})
__device__ __forceinline__ TreeRoot root_from_bbox(const unsigned int* __restrict__ enc) {
  TreeRoot r;
  float ext = 0.f, c[3];
  for (int a = 0; a < 3; a++) {
    const float lo = ordered_to_float(enc[a]), hi = ordered_to_float(enc[3 + a]);
    c[a] = (lo + hi) * 0.5f;               // :339-340
    ext = fmaxf(ext, hi - lo);
  }
  const float half = ext * 0.5f + 0.001f;  // :343
  for (int a = 0; a < 3; a++) r.lo[a] = c[a] - half;
  r.half = half;
  r.scale = 1024.0f / (2.0f * half);
  return r;
}

__device__ __forceinline__ unsigned int expand_bits10(unsigned int v) {  // :23-29
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

// 21 bits -> every third bit of a 64-bit word
__device__ __forceinline__ unsigned long long expand_bits21(unsigned long long v) {
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

// Key word and bits per axis: 10 bits per axis in 32-bit keys (max_depth <= 10), 21 bits per axis in
// 64-bit keys beyond.  The finer keys REFINE the coarser ones: the cube is cut into 2^B cells per axis
// with scale 2^B / (2 half) -- a power-of-two multiple of the 10-bit scale, so (p - lo) * scale is the
// same fp32 value times 2^11 and the top 10 bits of the 21-bit coordinate are the 10-bit coordinate.
template <class K> struct KeyTraits;
template <> struct KeyTraits<unsigned int> {
  static constexpr int kAxisBits = 10, kTop = 30;
  static constexpr float kRefine = 1.0f;
  static __device__ __forceinline__ unsigned int interleave(int x, int y, int z) {
    return (expand_bits10((unsigned)x) << 2) | (expand_bits10((unsigned)y) << 1) | expand_bits10((unsigned)z);
  }
};
template <> struct KeyTraits<unsigned long long> {
  static constexpr int kAxisBits = 21, kTop = 63;
  static constexpr float kRefine = 2048.0f;
  static __device__ __forceinline__ unsigned long long interleave(int x, int y, int z) {
    return (expand_bits21((unsigned long long)x) << 2) | (expand_bits21((unsigned long long)y) << 1) |
           expand_bits21((unsigned long long)z);
  }
};

// Keys of all bodies; every thread derives the root cube from the bounding box itself (six scalar loads), thread
// 0 publishes it for the later passes and re-arms the OTHER bounding-box buffer for the next build (two buffers
// alternate, so the build needs no separate init and no separate root launch).
// The kernel that has every key in a register also does the sort's bookkeeping (onesweep.h): it zeroes the look-back
// block of the passes and the odd-group marker planes (zero_a, zero_b: four fill launches less) and, with
// hist_places > 0, accumulates the digit histograms of the sort in LDS (rocPRIM's histogram kernel, 22 us at N = 2^20,
// is skipped then); the OTHER histogram buffer is zeroed for the next build.  A fixed grid strides over the bodies.
constexpr int kTreeHistPlaces = (63 + (NBH_BH_RADIX_BITS ? NBH_BH_RADIX_BITS : 8) - 1) / (NBH_BH_RADIX_BITS ? NBH_BH_RADIX_BITS : 8);
constexpr int kTreeHistWords = kTreeHistPlaces << (NBH_BH_RADIX_BITS ? NBH_BH_RADIX_BITS : 8);
constexpr int kTreeHistCopies = onesweep::DigitHistogram<(NBH_BH_RADIX_BITS ? NBH_BH_RADIX_BITS : 8)>::kCopies;
template <class K>
__global__ __launch_bounds__(NBH_HIST_THREADS) void morton_kernel(const float4* __restrict__ posm, int n,
                                                        const unsigned int* __restrict__ enc,
                                                        unsigned int* __restrict__ enc_next,
                                                        TreeRoot* __restrict__ root_out, int* __restrict__ level_base,
                                                        K* __restrict__ keys, int* __restrict__ idx,
                                                        unsigned int* __restrict__ zero_a, unsigned int words_a,
                                                        unsigned int* __restrict__ zero_b, unsigned int words_b,
                                                        unsigned int* __restrict__ hist, unsigned int* __restrict__ hist_next,
                                                        int hist_places, int first_bit) {
  using Hist = onesweep::DigitHistogram<(NBH_BH_RADIX_BITS ? NBH_BH_RADIX_BITS : 8)>;
  __shared__ unsigned int h[kTreeHistWords];
  const int t0 = blockIdx.x * NBH_HIST_THREADS + threadIdx.x, stride = gridDim.x * NBH_HIST_THREADS;
  for (unsigned int w = t0; w < words_a; w += stride) zero_a[w] = 0u;
  for (unsigned int w = t0; w < words_b; w += stride) zero_b[w] = 0u;
  for (int w = t0; w < kTreeHistWords * Hist::kCopies; w += stride) hist_next[w] = 0u;
  if (hist_places) {
    Hist::zero(h, hist_places, threadIdx.x, NBH_HIST_THREADS);
    __syncthreads();
  }
  const TreeRoot root = root_from_bbox(enc);
  if (t0 == 0) {
    *root_out = root;
    level_base[0] = 0;
    for (int a = 0; a < 3; a++) { enc_next[a] = 0xffffffffu; enc_next[3 + a] = 0u; }  // the empty box
  }
  const float s = root.scale * KeyTraits<K>::kRefine;
  const int top = (1 << KeyTraits<K>::kAxisBits) - 1;
  for (int i = t0; i < n; i += stride) {
    const float4 p = posm[i];
    int qx = (int)((p.x - root.lo[0]) * s);
    int qy = (int)((p.y - root.lo[1]) * s);
    int qz = (int)((p.z - root.lo[2]) * s);
    qx = min(max(qx, 0), top); qy = min(max(qy, 0), top); qz = min(max(qz, 0), top);
    const K key = KeyTraits<K>::interleave(qx, qy, qz);
    keys[i] = key;
    idx[i] = i;
    if (hist_places) Hist::add(h, key >> first_bit, hist_places);
  }
  if (hist_places) {
    __syncthreads();
    Hist::flush(h, hist_places, threadIdx.x, NBH_HIST_THREADS, hist, kTreeHistWords);
  }
}


// 32-byte traversal record: one s_load_dwordx8 per node, eight siblings = 256 contiguous bytes
struct __attribute__((aligned(32))) NodeRec {
  float cx, cy, cz, mass;  // centre of mass, mass
  float size2;             // (2 half)^2
  int first, count;        // range of the Morton-sorted body list
  unsigned int child;      // first child id | child count << 28 ; 0 = leaf
};

// Per-node build records (SoA)
struct TreeArrays {
  int* first;       // first sorted body
  int* last;        // one past the last sorted body
  int* child0;      // id of the first child, -1 = leaf
  int* child_last;  // id of the last child
  NodeRec* rec;     // 32-byte traversal record
  double4* m;       // fp64 monopole {com x, y, z, mass}
  // the same records as PAIR BLOCKS for the pair walk (bh_traverse_pair_kernel): nodes 2i and 2i + 1 share a
  // 48-byte block {cx cx, cy cy, cz cz, m m, s2 s2, link link}, so one s_load puts both nodes' values of a field
  // into an aligned SGPR pair = one packed operand; a sibling group (consecutive ids) is <= 5 contiguous blocks.
  //   m    : mass (0 for a leaf of several bodies: those interact body by body)
  //   s2   : (2 half)^2; -1 for a leaf (always accepted, never opened)
  //   link : internal node: first child | (children - 1) << 28 | kManyBit if one of the children is a leaf of
  //          several bodies; one-body leaf: 0; leaf of several bodies: kManyLeaf
  unsigned int* pb;
  // leaves of several bodies only exist where a node may not be split: below level scan_from (the depth limit,
  // or everywhere with leaf_max > 1) or when the node arrays overflowed (*node_total > capacity; node_total is
  // the UNCLAMPED count the numbering arrived at, level_base[kMaxDepth + 2])
  int scan_from, capacity;
  const int* node_total;
};
constexpr int kPairWords = 12;
constexpr unsigned int kManyBit = 0x80000000u;
constexpr unsigned int kManyLeaf = 0x70000000u;  // "eight children starting at node 0": not a possible link

// (runs after tree_fill_kernel: first / last / child0 of every node are final)
__device__ __forceinline__ void store_node(const TreeArrays& t, int nid, int level, const NodeRec& r) {
  t.rec[nid] = r;
  const bool leaf = r.child == 0u;
  unsigned int link = leaf ? (r.count == 1 ? 0u : kManyLeaf) : ((r.child & 0x0fffffffu) | (((r.child >> 28) - 1u) << 28));
  if (!leaf && (level + 1 >= t.scan_from || *t.node_total > t.capacity)) {  // a child may be a leaf of several bodies
    const int c0 = (int)(r.child & 0x0fffffffu), cn = (int)(r.child >> 28);
    for (int c = c0; c < c0 + cn; c++)
      if (t.child0[c] < 0 && t.last[c] - t.first[c] != 1) link |= kManyBit;
  }
  unsigned int* q = t.pb + (size_t)(nid >> 1) * kPairWords + (nid & 1);
  q[0] = __float_as_uint(r.cx); q[2] = __float_as_uint(r.cy); q[4] = __float_as_uint(r.cz);
  q[6] = __float_as_uint((leaf && r.count != 1) ? 0.f : r.mass);
  q[8] = __float_as_uint(leaf ? -1.0f : r.size2);
  q[10] = link;
}

// ---------------------------------------------------------------------------------------
// Topology from the sorted keys alone, all levels at once.
//
// A node at level L is a run of bodies sharing the 3L-bit key prefix ("group") whose PARENT group
// holds more than leaf_max bodies (force_barnes_hut.cu:197-209: a node is split while it holds
// more than one body; counts shrink monotonically down a branch, so the parent test implies
// every ancestor's).  Both facts are local in the sorted key list, so one kernel flags the first
// body of every node of every level, the flags are ranked per level (LevelRanks below: ids ascend
// by level, then by key = octant order; the children of a node are consecutive), and one kernel
// fills the ranges and child links from the ranks.
// ---------------------------------------------------------------------------------------

// how many of the `cap` bodies on one side of body i (dir = -1 / +1) share its prefix (key >> sp)
template <class K>
__device__ __forceinline__ int side_extent(const K* __restrict__ keys, int i, int dir, int cap, K prefix, int sp) {
  int lo = 0, hi = cap;  // the answer is in [lo, hi]; the predicate is monotone (sorted keys)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((keys[i + dir * mid] >> sp) == prefix) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// Node numbering without a scan over the (levels x n) flag array (round 2: one rocPRIM inclusive scan through a
// transform iterator, 84 us and an 88 MB output at N = 2^20, 21 levels).  The flags of level L are kept as BIT PLANES:
// plane[L][g] = ballot of the level-L flag over the 64 bodies of group g (= one wave of tree_flags_kernel), and
// off[L][g] = number of level-L flags before group g (one small workgroup per level scans the popcounts).  The
// inclusive rank of body j at level L -- how many level-L nodes start at or before j -- is then two loads and a
// popcount, wherever j lies.  4.3 MB of tables at N = 2^20.
//
// EVEN-ALIGNED SIBLING GROUPS.  The pair walk evaluates two nodes per packed instruction out of 48-byte blocks that
// hold nodes 2i and 2i + 1, so a sibling group should start on an even id: a group of two that starts odd costs two
// pair steps instead of one (unweighted over the two-galaxy tree of config 4: 2.00 pair steps per group against 1.66
// aligned).  Every group is therefore padded to an even number of ids: id = B[L] + (level-L nodes at or before the
// body) - 1 + (ODD-sized level-L groups that END at or before the body), B[L] = first id of level L (even).  The second
// count is a rank in a second set of planes: a group of odd size sets a marker at the position one past its last body
// (= its parent's `last`): every earlier group's marker lies at or before the start of a later group, a group's own
// marker lies behind all its bodies.  An odd group's last slot + 1 is a HOLE: an id without a node (first = -1).
struct LevelRanks {
  const unsigned long long* plane;  // [levels][G]
  const int* off;                   // [levels][G]: exclusive prefix of the popcounts of a level
  int G;
  __device__ __forceinline__ int rank(int L, int j) const {  // inclusive
    const size_t w = (size_t)L * G + (size_t)(j >> 6);
    return off[w] + __popcll(plane[w] & (~0ull >> (63 - (j & 63))));
  }
  // position of the r-th (1-based) flag of level L; r must be <= the level's total
  __device__ __forceinline__ int select(int L, int r) const {
    const int* o = off + (size_t)L * G;
    int lo = 0, hi = G - 1;  // largest g with off[g] < r
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (o[mid] < r) lo = mid; else hi = mid - 1;
    }
    unsigned long long w = plane[(size_t)L * G + lo];
    for (int k = r - o[lo]; k > 1; k--) w &= w - 1;  // drop the lowest k - 1 set bits
    return (lo << 6) + __ffsll((long long)w) - 1;
  }
};

template <class K>
__global__ __launch_bounds__(kBlock) void tree_flags_kernel(const K* __restrict__ keys,
                                                            int n, int max_depth, int leaf_max,
                                                            unsigned int* __restrict__ lvlmask,
                                                            const float4* __restrict__ posm,
                                                            const int* __restrict__ sorted_idx,
                                                            float4* __restrict__ sorted,
                                                            unsigned long long* __restrict__ plane,
                                                            unsigned long long* __restrict__ cum_plane, int G) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  unsigned int mask = 0;
  if (i < n) {
    sorted[i] = posm[sorted_idx[i]];  // the bodies in Morton order (the gather rides along: one launch less)
    const K k = keys[i];
    // first body of its group at level L  <=>  (d >> (top - 3L)) != 0; body 0 heads every level
    const K d = i == 0 ? ~(K)0 : (k ^ keys[i - 1]);
    bool open = true;  // the parent group is an internal node (level 0 has no parent)
    for (int L = 0; L <= max_depth; L++) {
      const int shift = KeyTraits<K>::kTop - 3 * L;
      if (L > 0 && open) {
        const int sp = shift + 3;
        const K prefix = k >> sp;
        const int a = side_extent(keys, i, -1, min(i, leaf_max), prefix, sp);
        int b = 0;
        if (a < leaf_max) b = side_extent(keys, i, +1, min(n - 1 - i, leaf_max - a), prefix, sp);
        open = a + b >= leaf_max;  // parent holds at least a + b + 1 > leaf_max bodies
      }
      const bool node = open && (d >> shift) != 0;
      if (node) mask |= 1u << L;
    }
    // levels at which body i heads a node (the fill pass walks the set bits)
    lvlmask[i] = mask;
  }
  // the wave's ballot of every level -> lane L keeps level L's and writes it with its popcount
  const int lane = threadIdx.x & 63, g = i >> 6;
  // ... and the OR of the levels 0 .. L (cum_plane[L]): the body behind the last one of a level-L node is the next
  // position that heads a node of ANY level <= L (it heads the next node of the coarsest level that ends there), so
  // the span pass finds a node's end as the next set bit of cum_plane[L] -- one load for all but the largest nodes
  unsigned long long mine = 0, mine_cum = 0, run = 0;
  for (int L = 0; L <= max_depth; L++) {
    const unsigned long long b = __ballot((mask >> L) & 1u);
    run |= b;
    if (lane == L) { mine = b; mine_cum = run; }
  }
  if (lane <= max_depth && g < G) {
    plane[(size_t)lane * G + g] = mine;
    cum_plane[(size_t)lane * G + g] = mine_cum;
  }
}

// one workgroup per (table, level): off[L][g] <- number of flags of plane[L] before group g, totals[L] <- flags of the
// level.  blockIdx.y selects the table (0: node flags, 1: odd-group markers)
constexpr int kScanBlock = 1024;
__global__ __launch_bounds__(kScanBlock) void level_scan_kernel(const unsigned long long* __restrict__ plane0,
                                                                int* __restrict__ off0, int* __restrict__ totals0,
                                                                const unsigned long long* __restrict__ plane1,
                                                                int* __restrict__ off1, int* __restrict__ totals1, int G) {
  constexpr int NW = kScanBlock / 64;
  __shared__ int wsum[NW];
  const int L = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned long long* pl = (blockIdx.y ? plane1 : plane0) + (size_t)L * G;
  int* o = (blockIdx.y ? off1 : off0) + (size_t)L * G;
  int* totals = blockIdx.y ? totals1 : totals0;
  // wave wv owns the contiguous chunk [c0, c1) (a multiple of 64 long): coalesced loads, shuffle scans
  const int per = ((G + NW - 1) / NW + 63) & ~63;
  const int c0 = min(G, wv * per), c1 = min(G, c0 + per);
  int sum = 0;
  for (int g = c0 + lane; g < c1; g += 64) sum += __popcll(pl[g]);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
  if (lane == 0) wsum[wv] = sum;
  __syncthreads();
  int run = 0, total = 0;
  for (int k = 0; k < NW; k++) {
    if (k < wv) run += wsum[k];
    total += wsum[k];
  }
  for (int g = c0; g < c1; g += 64) {
    const int c = g + lane < c1 ? __popcll(pl[g + lane]) : 0;
    int inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(inc, d, 64);
      if (lane >= d) inc += up;
    }
    if (g + lane < c1) o[g + lane] = run + inc - c;
    run += __shfl(inc, 63, 64);
  }
  if (tid == 0) totals[L] = total;
}

// Body ranges and group parities, one thread per BODY over the levels it heads (like the fill pass): the end of
// every node's body range (a gallop + binary search in the sorted keys) is stored under the node's PROVISIONAL id
// (plain rank, no padding) for the fill pass, and every internal node whose child count is odd sets its marker in
// the odd-group planes of the level below.
template <class K>
__global__ __launch_bounds__(kBlock) void tree_span_kernel(const K* __restrict__ keys, int n, int max_depth, int leaf_max,
                                                           const unsigned int* __restrict__ lvlmask, LevelRanks lr,
                                                           const int* __restrict__ totals, int capacity,
                                                           int* __restrict__ last_tmp,
                                                           unsigned long long* __restrict__ odd_plane,
                                                           const unsigned long long* __restrict__ cum_plane) {
  __shared__ int ubase[kMaxDepth + 3];
  if (threadIdx.x == 0) {
    int run = 0;
    for (int L = 0; L <= max_depth; L++) { ubase[L] = run; run += totals[L]; }
  }
  __syncthreads();
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  unsigned int m = lvlmask[i];
  const K key = keys[i];
  while (m) {
    const int L = __ffs(m) - 1;
    m &= m - 1;
    const int pid = ubase[L] + lr.rank(L, i) - 1;
    if (pid >= capacity) continue;
    const int shift = KeyTraits<K>::kTop - 3 * L;
    // one past the last body of the group = the next position that heads a node of a level <= L (tree_flags_kernel):
    // the next set bit of cum_plane[L] behind i, looked for in this word and the next three; larger nodes (rare) fall
    // back to the search in the keys: first j > i with another prefix
    int last = n;
    if (L > 0) {
      const unsigned long long* cp = cum_plane + (size_t)L * lr.G;
      const int g0 = i >> 6;
      unsigned long long wbits = (cp[g0] >> (i & 63)) >> 1;  // flags behind i in its own word
      int found = -1;
      if (wbits) {
        found = i + 1 + (__ffsll((long long)wbits) - 1);
      } else {
        for (int k = 1; k <= 3 && g0 + k < lr.G; k++) {
          const unsigned long long w2 = cp[g0 + k];
          if (w2) { found = ((g0 + k) << 6) + (__ffsll((long long)w2) - 1); break; }
        }
      }
      if (found >= 0) {
        last = min(found, n);
      } else {
        const K prefix = key >> shift;
        int lo = min(((g0 + 4) << 6), n), hi = n;  // (no flag up to here: the group reaches at least this far)
        if (lo < n && (keys[lo] >> shift) == prefix) {
          lo += 1;
          int step = 64;
          while (lo + step < hi && (keys[lo + step - 1] >> shift) == prefix) { lo += step; step <<= 1; }
          hi = min(hi, lo + step);
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((keys[mid] >> shift) == prefix) lo = mid + 1; else hi = mid;
          }
        }
        last = lo;
      }
    }
    last_tmp[pid] = last;
    if (L < max_depth && last - i > leaf_max) {
      const int cn = lr.rank(L + 1, last - 1) - lr.rank(L + 1, i) + 1;
      if (cn & 1) atomicOr(&odd_plane[(size_t)(L + 1) * lr.G + (size_t)(last >> 6)], 1ull << (last & 63));
    }
  }
}

// One thread per BODY: it fills the nodes this body heads (the set bits of its level mask; 1.3 nodes per
// body on average), instead of one thread per (level, body) entry of the flag array, 21 of 22 of which
// would only find a zero flag (145 -> 60 us at N = 2^20, 21 levels).  Ids are the even-aligned ones (see LevelRanks).
struct dd4;
struct PrefixSums {
  const dd4* P;     // entry k = sums over the workgroup's bodies before k
  const dd4* boff;  // workgroup offsets
};
__device__ __forceinline__ double4 prefix_monopole(const PrefixSums& ps, const float4* __restrict__ sorted, int first, int last);

// a hole's half of its pair block must be inert whatever an earlier build left there (the walk masks its lanes out, but
// 0 x NaN is NaN): at the origin, massless, marked as a leaf (never opened)
__device__ __forceinline__ void mark_hole(const TreeArrays& t, int hole) {
  t.first[hole] = -1;
  unsigned int* q = t.pb + (size_t)(hole >> 1) * kPairWords + (hole & 1);
  q[0] = 0u; q[2] = 0u; q[4] = 0u; q[6] = 0u;
  q[8] = __float_as_uint(-1.0f);
  q[10] = 0u;
}

// ALIGNED = false (trees that will not be walked by the pair kernel: fewer than kPairFrom bodies): plain ranks, no
// padding, no span pass -- the body ranges are searched here (the small-tree build is launch-bound: three launches less)
template <class K, bool ALIGNED>
__global__ __launch_bounds__(kBlock) void tree_fill_kernel(const K* __restrict__ keys,
                                                           int n, int max_depth, int leaf_max,
                                                           const unsigned int* __restrict__ lvlmask,
                                                           LevelRanks lr, LevelRanks odd, const int* __restrict__ totals,
                                                           const int* __restrict__ odd_totals,
                                                           const int* __restrict__ last_tmp,
                                                           TreeArrays t, int capacity,
                                                           int* __restrict__ level_base, int* __restrict__ level_real) {
  // ubase: provisional (unpadded) first id of a level = real nodes above it; base: first id of a level (even)
  __shared__ int ubase[kMaxDepth + 3], base[kMaxDepth + 3];
  if (threadIdx.x == 0) {
    int run = 0, idrun = 0;
    for (int L = 0; L <= max_depth; L++) {
      ubase[L] = run;
      base[L] = ALIGNED ? idrun : run;
      run += totals[L];
      idrun += L == 0 ? 2 : totals[L] + odd_totals[L];  // the root's slot 1 is a hole; every other level is even
    }
    ubase[max_depth + 1] = run;
    base[max_depth + 1] = ALIGNED ? idrun : run;
  }
  __syncthreads();
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i <= max_depth + 1) {  // thread L of the first block (256 > levels)
    level_base[i] = min(base[i], capacity);
    level_real[i] = ubase[i];
  }
  if (i == 0) {
    level_base[kMaxDepth + 2] = base[max_depth + 1];  // the UNCLAMPED id total (TreeArrays::node_total)
    if (ALIGNED && 1 < capacity) mark_hole(t, 1);      // the hole beside the root
  }
  if (i >= n) return;
  unsigned int m = lvlmask[i];
  [[maybe_unused]] const K key = keys[i];
  while (m) {
    const int L = __ffs(m) - 1;
    m &= m - 1;
    const int rk = lr.rank(L, i);
    const int pid = ubase[L] + rk - 1;
    const int nid = ALIGNED ? base[L] + rk - 1 + (L > 0 ? odd.rank(L, i) : 0) : pid;
    if (nid >= capacity || pid >= capacity) continue;  // beyond the node arrays: this node does not exist (its parent is a leaf)
    int last;
    if constexpr (ALIGNED) {
      last = last_tmp[pid];
    } else {
      // one past the last body of the group: first j > i with another prefix
      last = n;
      if (L > 0) {
        const int shift = KeyTraits<K>::kTop - 3 * L;
        const K prefix = key >> shift;
        int lo = i + 1, hi = n;
        int step = 1;  // gallop first: groups are short (a node of a deep level holds a handful of bodies)
        while (lo + step < hi && (keys[lo + step - 1] >> shift) == prefix) { lo += step; step <<= 1; }
        hi = min(hi, lo + step);
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if ((keys[mid] >> shift) == prefix) lo = mid + 1; else hi = mid;
        }
        last = lo;
      }
    }
    t.first[nid] = i;
    t.last[nid] = last;
    int c0 = -1, c1 = -1;
    if (L < max_depth && last - i > leaf_max) {
      // children: the flagged bodies of level L + 1 inside [i, last); body i is always one of them
      const int r0 = lr.rank(L + 1, i);
      const int cn = lr.rank(L + 1, last - 1) - r0 + 1;
      c0 = base[L + 1] + r0 - 1 + (ALIGNED ? odd.rank(L + 1, i) : 0);
      c1 = c0 + cn - 1;
      if (c1 >= capacity || ubase[L + 1] + r0 - 1 + cn - 1 >= capacity) { c0 = -1; c1 = -1; }
      else if (ALIGNED && (cn & 1) && c1 + 1 < capacity) mark_hole(t, c1 + 1);  // the hole that pads an odd group
    }
    t.child0[nid] = c0;
    t.child_last[nid] = c1;
  }
}

// monopoles of one level (deepest first) + the packed traversal records
template <int BLOCK>
__device__ __forceinline__ void monopole_level(int level, const int* __restrict__ level_base,
                                               const float4* __restrict__ sorted,
                                               const TreeRoot* __restrict__ root, TreeArrays t) {
  const int lo = level_base[level], hi = level_base[level + 1];
  for (int nid = lo + blockIdx.x * BLOCK + threadIdx.x; nid < hi; nid += gridDim.x * BLOCK) {
    if (t.first[nid] < 0) continue;  // a hole (padding of an odd sibling group)
    const int first = t.first[nid], cnt = t.last[nid] - first;
    const int c0 = t.child0[nid];
    double mx = 0.0, my = 0.0, mz = 0.0, ms = 0.0;
    double4 mono;
    int nchild = 0;
    if (c0 < 0) {  // leaf: its bodies, in sorted order
      for (int k = first; k < first + cnt; k++) {
        const float4 p = sorted[k];
        const double mb = (double)p.w;
        mx += mb * (double)p.x; my += mb * (double)p.y; mz += mb * (double)p.z; ms += mb;
      }
      if (cnt == 1) {
        const float4 p = sorted[first];
        mono = make_double4((double)p.x, (double)p.y, (double)p.z, ms);
      } else if (ms > 0.0) {
        mono = make_double4(mx / ms, my / ms, mz / ms, ms);
      } else {
        mono = make_double4(0.0, 0.0, 0.0, 0.0);
      }
    } else {
      const int c1 = t.child_last[nid];
      nchild = c1 - c0 + 1;
      for (int c = c0; c <= c1; c++) {  // consecutive ids = octant order
        const double4 cm = t.m[c];
        mx += cm.x * cm.w; my += cm.y * cm.w; mz += cm.z * cm.w; ms += cm.w;
      }
      mono = ms > 0.0 ? make_double4(mx / ms, my / ms, mz / ms, ms)
                      : make_double4(0.0, 0.0, 0.0, 0.0);
    }
    t.m[nid] = mono;
    const float h = ldexpf(root->half, -level);
    const float size = 2.0f * h;  // :168
    NodeRec r;
    r.cx = (float)mono.x; r.cy = (float)mono.y; r.cz = (float)mono.z; r.mass = (float)mono.w;
    r.size2 = size * size;
    r.first = first; r.count = cnt;
    r.child = c0 < 0 ? 0u : ((unsigned)c0 | ((unsigned)nchild << 28));
    store_node(t, nid, level, r);
  }
}

__global__ __launch_bounds__(kBlock) void level_monopole_kernel(int level,
                                                                const int* __restrict__ level_base,
                                                                const float4* __restrict__ sorted,
                                                                const TreeRoot* __restrict__ root,
                                                                TreeArrays t) {
  monopole_level<kBlock>(level, level_base, sorted, root, t);
}

// levels top_level .. 0 in ONE workgroup (a level reads the level below through global memory,
// ordered by the block barrier): the upper levels hold at most 8^L nodes, and with few bodies
// every level is small -- one launch instead of one per level.
constexpr int kTopBlock = 1024;
__global__ __launch_bounds__(kTopBlock) void top_monopole_kernel(int top_level,
                                                                 const int* __restrict__ level_base,
                                                                 const float4* __restrict__ sorted,
                                                                 const TreeRoot* __restrict__ root,
                                                                 TreeArrays t) {
  for (int level = top_level; level >= 0; level--) {
    monopole_level<kTopBlock>(level, level_base, sorted, root, t);
    __threadfence_block();  // one workgroup = one CU: its writes are visible to its own waves
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// Monopoles without the level-by-level dependency, for trees of up to kPrefixMax (1.5 M) bodies: a node covers a
// contiguous range of the Morton-sorted bodies, so its {sum m, sum m r} is a difference of two entries of the
// prefix sums over the sorted bodies.  Three launches instead of one per level (17 at the default depth, each a
// ~5 us floor).  The prefix sums are carried in DOUBLE-DOUBLE (exact products m x by fma, two-sum additions):
// the subtraction of two long prefixes would otherwise lose |prefix| / |node| of fp64's 1e-16 -- 1e-10 of a
// two-body node among a quarter of a million, enough to move the fp32 record of one node in a thousand by an
// ulp -- whereas a double-double difference is exact to ~1e-30 and the monopole is the correctly rounded fp64
// quotient, as close to the bottom-up fp64 sums (and to the oracle's) as those are to each other.  One-body leaves
// are copied, not subtracted.  Deterministic (fixed summation tree).  Larger trees keep the bottom-up passes:
// there the per-level kernels are bandwidth, not launch floor (N = 2^20: build 0.54 -> 0.51 ms with the
// prefixes, N = 2^22: 1.47 -> 1.51 ms).
// ---------------------------------------------------------------------------------------
#ifndef NBH_BH_PREFIX_MAX
#define NBH_BH_PREFIX_MAX 1572864
#endif
constexpr int kPrefixMax = NBH_BH_PREFIX_MAX;
constexpr int kPrefixBlock = 1024;

struct dd4 {  // {sum m x, sum m y, sum m z, sum m} as unevaluated sums hi + lo
  double hi[4], lo[4];
};
__device__ __forceinline__ void dd_add(double& ahi, double& alo, double bhi, double blo) {  // a += b
  const double s = ahi + bhi;
  const double bb = s - ahi;
  double e = (ahi - (s - bb)) + (bhi - bb);  // two-sum: s + e = ahi + bhi exactly
  e += alo + blo;
  const double hi = s + e;
  alo = e - (hi - s);  // fast two-sum (|s| >= |e|)
  ahi = hi;
}
__device__ __forceinline__ double shfl_up_f64(double v, int off) { return __shfl_up(v, off, 64); }

// inclusive scan of one dd4 per thread over a workgroup of kPrefixBlock threads (wave scans by shuffles, the wave
// totals by wave 0 through LDS); returns the EXCLUSIVE prefix of the thread, *total = the workgroup's sum
__device__ __forceinline__ dd4 block_exclusive_dd(const dd4& v, dd4* total) {
  constexpr int NW = kPrefixBlock / 64;
  __shared__ double tot[8][2 * NW];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  dd4 inc = v;
#pragma unroll
  for (int c = 0; c < 4; c++) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double uh = shfl_up_f64(inc.hi[c], off), ul = shfl_up_f64(inc.lo[c], off);
      if (lane >= off) dd_add(inc.hi[c], inc.lo[c], uh, ul);
    }
    if (lane == 63) { tot[c][wv] = inc.hi[c]; tot[4 + c][wv] = inc.lo[c]; }
  }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      double h = lane < NW ? tot[c][lane] : 0.0, l = lane < NW ? tot[4 + c][lane] : 0.0;
#pragma unroll
      for (int off = 1; off < NW; off <<= 1) {
        const double uh = shfl_up_f64(h, off), ul = shfl_up_f64(l, off);
        if (lane >= off) dd_add(h, l, uh, ul);
      }
      if (lane < NW) { tot[c][NW + lane] = h; tot[4 + c][NW + lane] = l; }  // inclusive scan of the wave totals
    }
  }
  __syncthreads();
  dd4 ex;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    // exclusive inside the wave = the lane before's inclusive value (exact, no subtraction)
    const double ph = shfl_up_f64(inc.hi[c], 1), pl = shfl_up_f64(inc.lo[c], 1);
    ex.hi[c] = lane > 0 ? ph : 0.0;
    ex.lo[c] = lane > 0 ? pl : 0.0;
    if (wv > 0) dd_add(ex.hi[c], ex.lo[c], tot[c][NW + wv - 1], tot[4 + c][NW + wv - 1]);
    total->hi[c] = tot[c][2 * NW - 1];
    total->lo[c] = tot[4 + c][2 * NW - 1];
  }
  return ex;
}

// pass 1: one body per thread (coalesced); P[k] = sums over the workgroup's bodies before k, btot[b] = the
// workgroup's total.  Entry n (the grand total) is thread n's "sum before".
__global__ __launch_bounds__(kPrefixBlock) void prefix_bodies_kernel(const float4* __restrict__ sorted, int n,
                                                                     dd4* __restrict__ P, dd4* __restrict__ btot) {
  const int k = blockIdx.x * kPrefixBlock + threadIdx.x;
  dd4 v;
#pragma unroll
  for (int c = 0; c < 4; c++) v.hi[c] = v.lo[c] = 0.0;
  if (k < n) {
    const float4 p = sorted[k];
    const double mb = (double)p.w;
    const double r[3] = {(double)p.x, (double)p.y, (double)p.z};
#pragma unroll
    for (int c = 0; c < 3; c++) {  // m x exactly: fp32 x fp32 fits fp64, the fma keeps it honest for any input
      v.hi[c] = mb * r[c];
      v.lo[c] = __builtin_fma(mb, r[c], -v.hi[c]);
    }
    v.hi[3] = mb;
  }
  dd4 total;
  const dd4 ex = block_exclusive_dd(v, &total);
  if (k <= n) P[k] = ex;
  if (threadIdx.x == 0) btot[blockIdx.x] = total;
}

// pass 2 (one workgroup): boff[b] = sum of the totals of the workgroups before b; a thread takes
// ceil(nblocks / kPrefixBlock) consecutive workgroups
__global__ __launch_bounds__(kPrefixBlock) void prefix_blocks_kernel(int nblocks, const dd4* __restrict__ btot,
                                                                     dd4* __restrict__ boff) {
  const int per = (nblocks + kPrefixBlock - 1) / kPrefixBlock;
  const int b0 = min(nblocks, (int)threadIdx.x * per), b1 = min(nblocks, b0 + per);
  dd4 v;
#pragma unroll
  for (int c = 0; c < 4; c++) v.hi[c] = v.lo[c] = 0.0;
  for (int b = b0; b < b1; b++) {
    const dd4 t = btot[b];
#pragma unroll
    for (int c = 0; c < 4; c++) dd_add(v.hi[c], v.lo[c], t.hi[c], t.lo[c]);
  }
  dd4 total;
  dd4 run = block_exclusive_dd(v, &total);
  for (int b = b0; b < b1; b++) {
    boff[b] = run;
    const dd4 t = btot[b];
#pragma unroll
    for (int c = 0; c < 4; c++) dd_add(run.hi[c], run.lo[c], t.hi[c], t.lo[c]);
  }
}

// monopole {com, mass} of the sorted bodies [first, last) from the prefix sums (a one-body node is its body)
__device__ __forceinline__ double4 prefix_monopole(const PrefixSums& ps, const float4* __restrict__ sorted, int first, int last) {
  if (last - first == 1) {
    const float4 p = sorted[first];
    return make_double4((double)p.x, (double)p.y, (double)p.z, (double)p.w);
  }
  // (P[last] + boff[its workgroup]) - (P[first] + boff[its workgroup]) in double-double
  dd4 hi = ps.P[last], lo = ps.P[first];
  const dd4 oh = ps.boff[last / kPrefixBlock], ol = ps.boff[first / kPrefixBlock];
  double s[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    dd_add(hi.hi[c], hi.lo[c], oh.hi[c], oh.lo[c]);
    dd_add(lo.hi[c], lo.lo[c], ol.hi[c], ol.lo[c]);
    dd_add(hi.hi[c], hi.lo[c], -lo.hi[c], -lo.lo[c]);
    s[c] = hi.hi[c] + hi.lo[c];
  }
  return s[3] > 0.0 ? make_double4(s[0] / s[3], s[1] / s[3], s[2] / s[3], s[3]) : make_double4(0.0, 0.0, 0.0, 0.0);
}

// one thread per NODE (the fill pass is one thread per body with 1..21 nodes each: fusing this arithmetic into it
// was measured slower, 144 us against 60 + 49 at N = 2^20 -- the per-body trip counts diverge)
__global__ __launch_bounds__(kBlock) void prefix_monopole_kernel(const int* __restrict__ level_base, int max_depth,
                                                                 const float4* __restrict__ sorted, PrefixSums ps,
                                                                 const TreeRoot* __restrict__ root, TreeArrays t) {
  // the id count is only known on the device: a fixed grid strides over it
  const int total = level_base[max_depth + 1];
  for (int nid = blockIdx.x * kBlock + threadIdx.x; nid < total; nid += gridDim.x * kBlock) {
    int level = 0;
    while (level < max_depth && nid >= level_base[level + 1]) level++;
    const int first = t.first[nid], last = t.last[nid], cnt = last - first;
    if (first < 0) continue;  // a hole (padding of an odd sibling group)
    const int c0 = t.child0[nid];
    const double4 mono = prefix_monopole(ps, sorted, first, last);
    const float h = ldexpf(root->half, -level);
    const float size = 2.0f * h;  // :168
    NodeRec r;
    r.cx = (float)mono.x; r.cy = (float)mono.y; r.cz = (float)mono.z; r.mass = (float)mono.w;
    r.size2 = size * size;
    r.first = first; r.count = cnt;
    r.child = c0 < 0 ? 0u : ((unsigned)c0 | ((unsigned)(t.child_last[nid] - c0 + 1) << 28));
    store_node(t, nid, level, r);
  }
}

constexpr int kVisitWords = 131;  // [0] node visits; [1..65] by lanes testing; [66..130] by lanes accepting

typedef float f2 __attribute__((ext_vector_type(2)));

// XCD-aware order of the walk's workgroups.  Workgroup b runs on XCD b % 8 (round-robin dispatch); giving XCD x
// the x-th contiguous eighth of the Morton-ordered targets keeps neighbouring walks -- which fetch the same nodes
// -- behind the same L2 (interleaving chunks of 1 .. 256 workgroups instead was measured: same or slower).
// A bijection for any nblk.
__device__ __forceinline__ int xcd_block(int b, int nblk) {
  if (NBH_BH_XCD == 0) return b;
  const int xcd = b % 8;
  return xcd * (nblk / 8) + min(xcd, nblk % 8) + b / 8;
}

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// |d|^2 of the walk: dx*dx, then two fused multiply-adds -- the chain nvcc's default contraction makes
// of force_barnes_hut.cu:165; oracle/nbody_oracle.c (bh_dist2) forms it identically, so both sides
// open exactly the same nodes.  (Everything else in the walk stays uncontracted except the
// explicit fma of the running sums, which takes no decision.)
__device__ __forceinline__ float bh_dist2(float dx, float dy, float dz) {
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// ---------------------------------------------------------------------------------------
// Traversal.  block = 256 = 4 independent waves; wave w of block b walks the tree for sorted
// bodies [b*256 + w*64, +64).
//
// SPLIT (few bodies): a walk is a chain of dependent node fetches, so its speed comes from having
// many waves in flight -- with n / 64 waves below a few thousand the chip idles.  gridDim.y = K
// replicas then share each wave's walk.  Nodes holding more than `unit_max` bodies form the "shared
// zone": every replica visits them, but such a node contributes only in the replica that owns it
// (id mod K).  A child of a shared node with at most unit_max bodies is a UNIT: only its owner
// enters it, and everything below belongs to that replica.  Units are bounded in size (n / 96
// bodies by default), so the replicas' loads are balanced whatever the mass distribution (a split
// at a fixed tree level left the replica owning a dense core with most of the work).  The
// replicas' fp64 partial sums are added in replica order by bh_combine_kernel (deterministic).
// ---------------------------------------------------------------------------------------
// HIST: diagnostics build of the same walk (lanes testing / accepting per node, nbody_hip_tree_count_visits);
// kept out of the production instantiation: the two extra scalar tests per node cost 13 % of the walk.
template <bool GUARD, bool SPLIT, bool HIST = false>
__global__ __launch_bounds__(kBlock) void bh_traverse_kernel(
    const NodeRec* __restrict__ nodes, const float4* __restrict__ sorted,
    const int* __restrict__ idx, int t_first, int n, float theta2, float eps2, float G,
    float* __restrict__ acc_x, float* __restrict__ acc_y, float* __restrict__ acc_z, float4* __restrict__ acc4,
    unsigned long long* __restrict__ visit_count, int unit_max, double* __restrict__ partial) {
  // walks the sorted bodies [t_first, t_first + n) (a sharded run gives each rank a range)
#pragma clang fp contract(off)  // distances and the opening test round exactly like the oracle
  __shared__ int4 stk[4][kStack];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int replica = SPLIT ? (int)blockIdx.y : 0;
  const int rmask = SPLIT ? (int)gridDim.y - 1 : 0;  // K is a power of two
  // XCD-aware order: workgroup b runs on XCD b % 8 (round-robin dispatch); giving XCD x the x-th
  // contiguous eighth of the Morton-ordered targets keeps neighbouring walks -- which fetch the same
  // nodes -- behind the same L2
  const int bid = xcd_block((int)blockIdx.x, (int)gridDim.x);
  const int tl = bid * kBlock + tid;  // position in the range
  const int t = t_first + tl;                // position in the sorted body list
  const bool valid = tl < n;
  float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid) pi = sorted[t];
  double sx = 0.0, sy = 0.0, sz = 0.0;  // fp32 sums of one sibling group are folded into fp64
  const unsigned long long m0 = __ballot(valid);
  if (m0 == 0ull) return;  // wave-uniform
  int sp = 0;
  // entry = (first child, child count | parent-is-shared << 8, 64-bit lane mask); the root's "parent" is shared
  if (lane == 0) stk[w][0] = make_int4(0, 1 | 256, (int)(unsigned)(m0 & 0xffffffffull), (int)(unsigned)(m0 >> 32));
  sp = 1;
  __builtin_amdgcn_wave_barrier();
  unsigned long long visited = 0;

  while (sp > 0) {
    sp--;
    const int4 e = stk[w][sp];
    const int c0 = rfl(e.x), cnf = rfl(e.y), cn = cnf & 0xff;
    const bool pshared = SPLIT && (cnf & 256);
    const unsigned long long M = ((unsigned long long)(unsigned)rfl(e.w) << 32) | (unsigned)rfl(e.z);
    const bool in = __builtin_amdgcn_inverse_ballot_w64(M);  // the scalar mask as the lane predicate: no VALU
    // the whole sibling group (<= 8 consecutive 32-byte records) is fetched up front with
    // wave-uniform (scalar-cache) loads: ONE memory round trip per group instead of one per node.
    // The node array is padded by 8 records, so reading past a short group is harmless.
    NodeRec rec[8];
#pragma unroll
    for (int k = 0; k < 8; k++) rec[k] = nodes[c0 + k];
    visited += cn;
    float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (k >= cn) break;  // wave-uniform
      const NodeRec nd = rec[k];
      // (a massless node, :161-162, needs no test of its own: it contributes m * g = 0 if accepted and
      // only massless descendants if opened)
      bool mine = true, shared = false;  // wave-uniform
      if (SPLIT && pshared) {
        shared = nd.count > unit_max;
        mine = ((c0 + k) & rmask) == replica;
        if (!shared && !mine) continue;  // another replica's unit
      }
      if (nd.child == 0u) {
        if (!mine) continue;
        // leaf: its bodies interact individually (exact), the body itself is skipped (:175)
        if (nd.count == 1) {
          // the record of a one-body leaf IS the body (the monopole pass copies x, y, z, m
          // unchanged): no fetch from the body list, i.e. one dependent memory round trip less
          // (-10 % in the split walk, -5..8 % at 1M bodies)
          const float dx = nd.cx - pi.x, dy = nd.cy - pi.y, dz = nd.cz - pi.z;
          const float d2 = bh_dist2(dx, dy, dz);
          const float inv = __builtin_amdgcn_rsqf(d2 + eps2);
          bool ok = in && (nd.first != t);
          if (GUARD) ok = ok && (d2 > 0.f);
          const float f = ok ? ((nd.mass * inv) * inv) * inv : 0.f;
          ax = __builtin_fmaf(f, dx, ax); ay = __builtin_fmaf(f, dy, ay); az = __builtin_fmaf(f, dz, az);
          continue;
        }
        for (int q = nd.first; q < nd.first + nd.count; q++) {
          const float4 s = sorted[q];
          const float dx = s.x - pi.x, dy = s.y - pi.y, dz = s.z - pi.z;
          const float d2 = bh_dist2(dx, dy, dz);
          const float inv = __builtin_amdgcn_rsqf(d2 + eps2);
          bool ok = in && (q != t);
          if (GUARD) ok = ok && (d2 > 0.f);
          const float f = ok ? ((s.w * inv) * inv) * inv : 0.f;
          ax = __builtin_fmaf(f, dx, ax); ay = __builtin_fmaf(f, dy, ay); az = __builtin_fmaf(f, dz, az);
        }
        continue;
      }
      const float dx = nd.cx - pi.x, dy = nd.cy - pi.y, dz = nd.cz - pi.z;
      const float dist2 = bh_dist2(dx, dy, dz) + eps2;  // :165
      // :171-172 `size2 / dist2 < theta2`, evaluated as size2 < theta2 * dist2 (dist2 > 0): the same
      // inequality without the IEEE division sequence; the oracle uses the same form
      const bool far = nd.size2 < theta2 * dist2;
      const unsigned long long F = __ballot(far);
      if constexpr (HIST) {  // lanes that test / accept this node
        if (lane == 0) {
          atomicAdd(&visit_count[1 + __popcll(M)], 1ull);
          atomicAdd(&visit_count[66 + __popcll(M & F)], 1ull);
        }
      }
      {  // one select on the factor; a masked region is if-converted by the compiler into three selects on
         // the sums anyway (measured: -4..8 % against `if (in && far && mine) {...}`; skipping the block
         // with a wave-uniform branch when no lane accepts: no gain, round 2)
        const float inv = __builtin_amdgcn_rsqf(dist2);
        const float f = (in && far && mine) ? ((nd.mass * inv) * inv) * inv : 0.f;
        ax = __builtin_fmaf(f, dx, ax); ay = __builtin_fmaf(f, dy, ay); az = __builtin_fmaf(f, dz, az);
      }
      // lanes of the group's mask that must open the node: scalar mask arithmetic on the compare
      // result (a ballot of `in && !far` goes through a VGPR and a second compare)
      const unsigned long long O = M & ~F;
      if (O != 0ull) {
        if (lane == 0)
          stk[w][sp] = make_int4((int)(nd.child & 0x0fffffffu), (int)(nd.child >> 28) | (shared ? 256 : 0),
                                 (int)(unsigned)(O & 0xffffffffull), (int)(unsigned)(O >> 32));
        sp++;
      }
    }
    sx += (double)ax; sy += (double)ay; sz += (double)az;
    __builtin_amdgcn_wave_barrier();
  }
  if (valid) {
    if (SPLIT) {
      double* p = partial + (size_t)replica * 3 * n;
      p[tl] = sx; p[(size_t)n + tl] = sy; p[2 * (size_t)n + tl] = sz;
    } else {
      const int i = idx[t];
      const float fx = (float)((double)G * sx), fy = (float)((double)G * sy), fz = (float)((double)G * sz);
      if (acc4) {
        acc4[i] = make_float4(fx, fy, fz, 0.f);
      } else {
        acc_x[i] = fx; acc_y[i] = fy; acc_z[i] = fz;
      }
    }
  }
  if (visit_count && lane == 0) atomicAdd(visit_count, visited);
}

// ---------------------------------------------------------------------------------------
// Pair walk (the walk without replicas, from kPairFrom bodies).  Same wave-shared walk as bh_traverse_kernel,
// but the siblings of a popped group are processed TWO AT A TIME: the pair blocks (TreeArrays::pb) put nodes
// (2i, 2i+1) into aligned SGPR pairs, so one packed instruction (v_pk_add / v_pk_mul / v_pk_fma_f32) forms the
// distance chain, the opening threshold, the m inv^3 chain and the three accumulations of both nodes for a
// lane's body; only the compare, the rsq and the select stay per node.  A group (consecutive ids c0 .. c0 + cn - 1)
// is the blocks c0 / 2 .. (c0 + cn - 1) / 2; the first block's even node / the last block's odd node may belong
// to a neighbouring group and are masked off.  A leaf is an always-accepted node (size2 = -1); its own body
// needs no exclusion (d = 0 adds nothing); leaves of several bodies (only at the depth limit or with
// leaf_max > 1) are flagged in their parent's link and take the body-by-body loop.  Interaction lists and the
// opening decisions are those of bh_traverse_kernel (same rounding of every test); the fp32 sum of a sibling
// group is formed as (even ids) + (odd ids) instead of in octant order, so results agree with the plain walk
// to fp32 rounding of a group sum.  One wave per workgroup, 80 SGPRs (= 8 waves per SIMD: a walk is a chain of
// dependent fetches, the waves in flight are its speed), schedule: walk_plan_kernel.
// ---------------------------------------------------------------------------------------
typedef unsigned int u16v __attribute__((ext_vector_type(16), aligned(16)));
typedef unsigned int u8v __attribute__((ext_vector_type(8), aligned(16)));
typedef unsigned int u4v __attribute__((ext_vector_type(4), aligned(16)));

// Arguments in one struct = the kernel-argument segment.  The walk loop has no SGPR to spare (80 = 8 waves per
// SIMD), so what only the epilogue needs (`cold` below) is NOT touched before the loop: it is read afterwards
// through a laundered kernarg pointer, which keeps the compiler from preloading it into SGPRs at entry.
struct PairArgs {
  const unsigned int* pb;     // pair blocks
  const NodeRec* rec;         // records (leaves of several bodies only)
  const float4* sorted;
  const int* order;           // cost-ordered schedule or null
  int t_first, n;
  float theta2, eps2;
  // cold
  const int* idx;
  float* acc_x; float* acc_y; float* acc_z;
  float4* acc4;
  unsigned long long* visit_count;
  int* cost_out;
  float G;
};

typedef int i3v __attribute__((ext_vector_type(3)));  // 16-byte aligned: ds_read_b96 / ds_write_b96 of a stack entry

template <bool GUARD, int WAVES>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_num_sgpr(80))) void bh_traverse_pair_kernel(PairArgs a) {
#pragma clang fp contract(off)  // distances and the opening test round exactly like the oracle
  const float4* __restrict__ sorted = a.sorted;
  const int* __restrict__ order = a.order;
  const int t_first = a.t_first, n = a.n;
  const float theta2 = a.theta2, eps2 = a.eps2;
  __shared__ int4 stk[WAVES][kStack];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // which 64 * WAVES bodies: the cost-ordered schedule of walk_order_kernel (-1 = padding), or the plain XCD order
  const int bid = order ? order[blockIdx.x] : xcd_block((int)blockIdx.x, (int)gridDim.x);
  if (bid < 0) return;
  const int tl = bid * (WAVES * 64) + tid;
  const int t = t_first + tl;
  const bool valid = tl < n;
  float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid) pi = sorted[t];
  const f2 px = f2{pi.x, pi.x}, py = f2{pi.y, pi.y}, pz = f2{pi.z, pi.z};
  double sx = 0.0, sy = 0.0, sz = 0.0;
  const unsigned long long m0 = __ballot(valid);
  if (m0 == 0ull) return;  // wave-uniform
  // entry = (link word of the group's parent as the records hold it: first child | (children - 1) << 28 | "a child
  // is a leaf of several bodies" << 31, 64-bit lane mask): three words, pushed as the parent's link VERBATIM
  // (round 2: four words, unpacked at the push: one register move and two scalar operations more per push / pop)
  if (lane == 0) stk[w][0] = make_int4(0, (int)(unsigned)(m0 & 0xffffffffull), (int)(unsigned)(m0 >> 32), 0);
  int sp = 1;
  __builtin_amdgcn_wave_barrier();
  // eps^2 and theta^2 as packed VGPR constants (SGPRs are the scarce resource of this kernel)
  float e2s, th2s;
  asm("v_mov_b32 %0, %1" : "=v"(e2s) : "s"(eps2));
  asm("v_mov_b32 %0, %1" : "=v"(th2s) : "s"(theta2));
  const f2 e2 = f2{e2s, e2s}, th2 = f2{th2s, th2s};
  unsigned long long visited = 0;
  f2 ax = f2{0.f, 0.f}, ay = ax, az = ax;
#if NBH_BH_FOLD != 1
  int groups = 0;
#endif

  while (sp > 0) {
    sp--;
    const i3v e = *reinterpret_cast<const i3v*>(&stk[w][sp]);
    const unsigned int link = (unsigned int)rfl(e.x);
    const unsigned int c0 = link & 0x0fffffffu;
    const int cn = (int)((link >> 28) & 7u) + 1;
    const unsigned long long M = ((unsigned long long)(unsigned)rfl(e.z) << 32) | (unsigned)rfl(e.y);
    visited += cn;
    // the group's pair blocks: sibling groups start on an EVEN id (see LevelRanks), so ids c0 .. c0 + cn - 1 are
    // blocks c0 / 2 .. c0 / 2 + ceil(cn / 2) - 1, at most four, one contiguous run: three always, the fourth only
    // for groups of seven or eight
    const unsigned int fb = c0 >> 1;
    const int nb = (cn + 1) >> 1;
    const unsigned int* gp = a.pb + (size_t)fb * kPairWords;
    // three blocks up front; a group that reaches blocks 3 and 4 (six or more children) fetches them into the
    // same registers once the first three are done: 36 instead of 60 SGPRs of node data keep the kernel at
    // <= 80 SGPRs = 8 waves per SIMD (a walk is a chain of dependent fetches: waves in flight are its speed)
    unsigned int blk[3 * kPairWords];
    {
      u16v v0 = *reinterpret_cast<const u16v*>(gp), v1 = *reinterpret_cast<const u16v*>(gp + 16);
      u4v v2 = *reinterpret_cast<const u4v*>(gp + 32);
      // (the empty asm pins every fetch up here: the compiler would otherwise sink the later blocks' loads to
      // their first use, a second dependent memory round trip per group)
      asm("" : "+s"(v0), "+s"(v1), "+s"(v2));
#pragma unroll
      for (int k = 0; k < 16; k++) { blk[k] = v0[k]; blk[16 + k] = v1[k]; }
#pragma unroll
      for (int k = 0; k < 4; k++) blk[32 + k] = v2[k];
    }
    // the last block's odd node is the padding of an odd group
    const unsigned long long Mlast = (cn & 1) ? 0ull : M;
#if NBH_BH_FOLD == 1
    ax = f2{0.f, 0.f}; ay = ax; az = ax;
#endif
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (k >= nb) break;  // wave-uniform
      if (k == 3) {
        u8v v3 = *reinterpret_cast<const u8v*>(gp + 36);
        u4v v4 = *reinterpret_cast<const u4v*>(gp + 44);
        asm("" : "+s"(v3), "+s"(v4));
#pragma unroll
        for (int i = 0; i < 8; i++) blk[i] = v3[i];
#pragma unroll
        for (int i = 0; i < 4; i++) blk[8 + i] = v4[i];
      }
      const unsigned int* q = blk + (k % 3) * kPairWords;
      const unsigned long long Ma = M;
      const unsigned long long Mb = k == nb - 1 ? Mlast : M;
      const f2 dx = f2{__uint_as_float(q[0]), __uint_as_float(q[1])} - px,
               dy = f2{__uint_as_float(q[2]), __uint_as_float(q[3])} - py,
               dz = f2{__uint_as_float(q[4]), __uint_as_float(q[5])} - pz;
      const f2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
      const f2 dist2 = d2 + e2;      // :165
      const f2 thr = th2 * dist2;    // :171-172 as size2 < theta2 * dist2
      // "not >=": a NaN distance counts as far (the plain walk opens such a node down to the NaN body, which
      // then spoils the same sums); leaves (size2 = -1) are far for every lane, so they are never opened
      const unsigned long long Fa = __ballot(!(__uint_as_float(q[8]) >= thr.x)),
                               Fb = __ballot(!(__uint_as_float(q[9]) >= thr.y));
      const f2 inv = f2{__builtin_amdgcn_rsqf(dist2.x), __builtin_amdgcn_rsqf(dist2.y)};
      const f2 fac = ((f2{__uint_as_float(q[6]), __uint_as_float(q[7])} * inv) * inv) * inv;
      // the body's own leaf needs no exclusion (:175): d = 0 there, so it adds fac * 0 = 0 (fac is finite: eps^2 >=
      // 1e-12 in this instantiation).  With eps ~ 0 (GUARD) every coincident body is dropped by the d2 > 0 test.
      unsigned long long Aa = Ma & Fa, Ab = Mb & Fb;
      if (GUARD) {
        if ((int)q[8] < 0) Aa &= __ballot(d2.x > 0.f);
        if ((int)q[9] < 0) Ab &= __ballot(d2.y > 0.f);
      }
      const f2 f = f2{__builtin_amdgcn_inverse_ballot_w64(Aa) ? fac.x : 0.f,
                      __builtin_amdgcn_inverse_ballot_w64(Ab) ? fac.y : 0.f};
      ax = __builtin_elementwise_fma(f, dx, ax);
      ay = __builtin_elementwise_fma(f, dy, ay);
      az = __builtin_elementwise_fma(f, dz, az);
      const unsigned long long Oa = Ma & ~Fa, Ob = Mb & ~Fb;
      if (Oa != 0ull) {
        // (every lane stores the same entry to the same address: no lane predicate to keep in SGPRs)
        *reinterpret_cast<i3v*>(&stk[w][sp]) = i3v{(int)q[10], (int)(unsigned)(Oa & 0xffffffffull), (int)(unsigned)(Oa >> 32)};
        sp++;
      }
      if (Ob != 0ull) {
        *reinterpret_cast<i3v*>(&stk[w][sp]) = i3v{(int)q[11], (int)(unsigned)(Ob & 0xffffffffull), (int)(unsigned)(Ob >> 32)};
        sp++;
      }
    }
    if (link & kManyBit) {  // leaves of several bodies (depth limit, leaf_max > 1): body by body, exact
      const bool in = __builtin_amdgcn_inverse_ballot_w64(M);
      for (int k = 0; k < cn; k++) {
        const NodeRec nd = a.rec[c0 + k];
        if (nd.child != 0u || nd.count == 1) continue;
        for (int q = nd.first; q < nd.first + nd.count; q++) {
          const float4 s = sorted[q];
          const float ex = s.x - pi.x, ey = s.y - pi.y, ez = s.z - pi.z;
          const float q2 = bh_dist2(ex, ey, ez);
          const float qi = __builtin_amdgcn_rsqf(q2 + eps2);
          bool ok = in && (q != t);
          if (GUARD) ok = ok && (q2 > 0.f);
          const float g = ok ? ((s.w * qi) * qi) * qi : 0.f;
          ax.x = __builtin_fmaf(g, ex, ax.x); ay.x = __builtin_fmaf(g, ey, ay.x); az.x = __builtin_fmaf(g, ez, az.x);
        }
      }
    }
#if NBH_BH_FOLD == 1
    sx += (double)(ax.x + ax.y); sy += (double)(ay.x + ay.y); sz += (double)(az.x + az.y);
#else
    if ((++groups & (NBH_BH_FOLD - 1)) == 0) {  // fp32 sums of NBH_BH_FOLD sibling groups folded into fp64
      sx += (double)(ax.x + ax.y); sy += (double)(ay.x + ay.y); sz += (double)(az.x + az.y);
      ax = f2{0.f, 0.f}; ay = ax; az = ax;
    }
#endif
    __builtin_amdgcn_wave_barrier();
  }
#if NBH_BH_FOLD != 1
  sx += (double)(ax.x + ax.y); sy += (double)(ay.x + ay.y); sz += (double)(az.x + az.y);
#endif
  // epilogue arguments, fetched now (see PairArgs)
  const PairArgs* ka = (const PairArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm("" : "+s"(ka));
  if (valid) {
    const int i = ka->idx[t];
    const float G = ka->G;
    const float fx = (float)((double)G * sx), fy = (float)((double)G * sy), fz = (float)((double)G * sz);
    float4* acc4 = ka->acc4;
    if (acc4) {
      acc4[i] = make_float4(fx, fy, fz, 0.f);
    } else {
      ka->acc_x[i] = fx; ka->acc_y[i] = fy; ka->acc_z[i] = fz;
    }
  }
  if (lane == 0) {
    if (int* cost_out = ka->cost_out) cost_out[bid * WAVES + w] = (int)visited;  // next walk's schedule
    if (unsigned long long* visit_count = ka->visit_count) atomicAdd(visit_count, visited);
  }
}

// ---------------------------------------------------------------------------------------
// Cost-ordered schedule of the pair walk.  A wave's walk is one long dependent chain (~0.5 ms at N = 2^20, the
// kernel lasts 2 1/2 of them) and walks differ in length by the local density, so with the plain order the
// launch ends in a long tail of half-empty SIMDs.  Bodies move little in a step: the node visits every wave
// recorded in the PREVIOUS walk (cost_out) predict this one.  walk_plan_kernel cuts the Morton order into eight
// contiguous ranges of equal total cost, one per XCD (workgroup b runs on XCD b % 8; contiguous ranges keep
// neighbouring walks behind one L2), each at most `cap` waves long; walk_order_kernel (one workgroup per XCD)
// orders a range by decreasing cost -- a counting sort over 256 cost classes: longest walks first, the order
// inside a class is immaterial -- straight into the interleaved list order[8 * cap] (-1 = padding).  Results do
// not depend on the schedule.
// ---------------------------------------------------------------------------------------
constexpr int kPlanBlock = 1024;
__global__ __launch_bounds__(kPlanBlock) void walk_plan_kernel(const int* __restrict__ cost, int waves, int cap,
                                                               int* __restrict__ bounds /* 9 */) {
  __shared__ unsigned long long part[kPlanBlock];
  const int tid = threadIdx.x;
  const int per = (waves + kPlanBlock - 1) / kPlanBlock;
  const int i0 = min(waves, tid * per), i1 = min(waves, i0 + per);
  unsigned long long sum = 0;
  for (int i = i0; i < i1; i++) sum += (unsigned int)cost[i] + 1u;
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < kPlanBlock; off <<= 1) {  // inclusive scan of the chunk sums
    const unsigned long long v = tid >= off ? part[tid - off] : 0ull;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  if (tid == 0) {
    // bounds[x] = first wave of XCD x: the chunk (of `per` waves) at which the running cost passes x / 8 of the
    // total, moved so that no range exceeds cap and the rest still fits into the remaining ranges
    const unsigned long long total = part[kPlanBlock - 1];
    int prev = 0;
    bounds[0] = 0;
    for (int x = 1; x < 8; x++) {
      const unsigned long long want = total / 8 * (unsigned long long)x;
      int lo = 0, hi = kPlanBlock;  // first chunk whose inclusive sum exceeds want
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (part[mid] > want) hi = mid; else lo = mid + 1;
      }
      int i = min(waves, lo * per);
      i = max(i, prev);
      i = min(i, prev + cap);
      i = max(i, waves - (8 - x) * cap);
      i = min(max(i, 0), waves);
      bounds[x] = i;
      prev = i;
    }
    bounds[8] = waves;
  }
}

constexpr int kCostClasses = 256;
__global__ __launch_bounds__(kPlanBlock) void walk_order_kernel(const int* __restrict__ cost,
                                                                const int* __restrict__ bounds, int cap,
                                                                int* __restrict__ order) {
  __shared__ int hist[kCostClasses];
  __shared__ int lo_hi[2];
  const int x = blockIdx.x, tid = threadIdx.x;
  const int start = bounds[x], end = bounds[x + 1];
  if (tid < kCostClasses) hist[tid] = 0;
  if (tid == 0) { lo_hi[0] = 0x7fffffff; lo_hi[1] = 0; }
  __syncthreads();
  int lo = 0x7fffffff, hi = 0;
  for (int i = start + tid; i < end; i += kPlanBlock) { lo = min(lo, cost[i]); hi = max(hi, cost[i]); }
  atomicMin(&lo_hi[0], lo);
  atomicMax(&lo_hi[1], hi);
  __syncthreads();
  lo = lo_hi[0];
  const int span = max(lo_hi[1] - lo, 0) + 1;
  auto cls = [&](int c) {  // class 0 = the most expensive
    return (int)(((long long)(lo_hi[1] - c) * kCostClasses) / span);
  };
  for (int i = start + tid; i < end; i += kPlanBlock) atomicAdd(&hist[cls(cost[i])], 1);
  __syncthreads();
  if (tid == 0) {  // exclusive scan of 256 counters
    int run = 0;
    for (int k = 0; k < kCostClasses; k++) { const int c = hist[k]; hist[k] = run; run += c; }
  }
  __syncthreads();
  for (int i = start + tid; i < end; i += kPlanBlock) {
    const int r = atomicAdd(&hist[cls(cost[i])], 1);
    order[r * 8 + x] = i;
  }
  for (int r = end - start + tid; r < cap; r += kPlanBlock) order[r * 8 + x] = -1;
}

__global__ __launch_bounds__(kBlock) void bh_combine_kernel(const double* __restrict__ partial, int replicas,
                                                            const int* __restrict__ idx, int t_first, int n,
                                                            float G, float* __restrict__ acc_x,
                                                            float* __restrict__ acc_y,
                                                            float* __restrict__ acc_z,
                                                            float4* __restrict__ acc4) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= n) return;
  double sx = 0.0, sy = 0.0, sz = 0.0;
  for (int r = 0; r < replicas; r++) {
    const double* p = partial + (size_t)r * 3 * n;
    sx += p[t]; sy += p[(size_t)n + t]; sz += p[2 * (size_t)n + t];
  }
  const int i = idx[t_first + t];
  const float fx = (float)((double)G * sx), fy = (float)((double)G * sy), fz = (float)((double)G * sz);
  if (acc4) {
    acc4[i] = make_float4(fx, fy, fz, 0.f);
  } else {
    acc_x[i] = fx; acc_y[i] = fy; acc_z[i] = fz;
  }
}

}  // namespace nbh

using namespace nbh;

// ref layout: OctreeNode, include/nbody/barnes_hut_tree.hpp:9-30 (76 bytes)
struct RefOctreeNode {
  float center[3]; float half_size; float com[3]; float total_mass;
  int children[8]; int particle_index; bool is_leaf; int particle_count;
};
static_assert(sizeof(RefOctreeNode) == 76, "OctreeNode layout");

struct nbody_hip_tree {
  nbody_hip_ctx* ctx = nullptr;
  size_t max_particles = 0;
  int max_depth = kDefaultDepth;
  int leaf_max = 1;
  int capacity = 0;
  unsigned int* d_enc = nullptr;  // two bounding-box buffers of 8 words (see morton_kernel)
  unsigned int* d_hist = nullptr; // two digit-count buffers of the sort (kTreeHistWords each), alternating like d_enc
  int sort_impl = 0;              // above the crossover: 1 = driver of rocPRIM's Onesweep kernels (onesweep.h, fenced), 2 = the
                                  // hand-written sort (radix_sort.h), 0 = rocprim::radix_sort_pairs (NBH_SORT in the environment)
  unsigned int* h_sort_err = nullptr;      // mapped host word the hand-written sort raises when a look-back gives up
  unsigned int* h_sort_err_dev = nullptr;
  unsigned int enc_flip = 0;
  bool enc_armed = false;
  unsigned long long enc_replays = 0;
  TreeRoot* d_root = nullptr;
  int* d_level_base = nullptr;  // kMaxDepth + 3 ints
  void *d_keys_a = nullptr, *d_keys_b = nullptr;  // 32-bit keys up to depth 10, 64-bit keys beyond
  bool wide() const { return max_depth > kDepth32; }
  int *d_idx_a = nullptr, *d_idx_b = nullptr;
  float4* d_sorted = nullptr;
  // node numbering (LevelRanks): bit planes and per-group offsets of every level, the level totals
  unsigned long long* d_plane = nullptr;  // 3 x (max_depth + 1) * rank_G: node flags, odd-group markers, cumulative flags
  int* d_rank_off = nullptr;              // 2 x (max_depth + 1) * rank_G
  int* d_totals = nullptr;                // 2 x (kMaxDepth + 3): nodes per level, odd groups per level
  int* d_level_real = nullptr;            // kMaxDepth + 3: real nodes above every level (ids have holes)
  int* d_last_tmp = nullptr;              // capacity: end of every node's body range, by provisional id
  int rank_G = 0;                         // groups of 64 bodies at max_particles
  int node_limit = 0;                     // nbody_hip_tree_limit_nodes: 0 = the bound on the node count
  TreeArrays t{};
  void* d_tmp = nullptr;
  size_t tmp_bytes = 0;
  unsigned long long* d_visits = nullptr;
  double* d_partial = nullptr;  // replicas x 3 x n fp64 partial sums (split traversal)
  dd4* d_prefix = nullptr;  // prefix_cap prefix sums of the sorted bodies + workgroup totals + offsets
  size_t prefix_cap = 0;
  int tune_replicas = 0, tune_split_level = 0;  // 0 = automatic
  size_t own_sort_from = kOwnSortFromTree;     // onesweep.h driver from this many bodies (63-bit keys)
  int tune_form = 0;                            // walk without replicas: 0 = automatic, 1 = plain, 2 = pair walk
  // cost-ordered schedule of the pair walk (walk_plan_kernel): node visits per wave of the previous walk
  int* d_cost = nullptr;         // waves
  int* d_order = nullptr;        // 8 * cap
  int* d_bounds = nullptr;       // 9
  int cost_first = -1, cost_n = -1;  // the range the recorded costs belong to
  bool tune_schedule = true;
  bool count_visits = false;
  size_t built_count = 0;
  // A side stream for the small passes that do not depend on the build's critical path: the double-double prefix sums
  // of the sorted bodies (needed only by the monopole pass) run beside the level ranks + fill pass, and the walk's
  // cost-ordered schedule (a function of the PREVIOUS walk's visit counts) beside the whole build.  Fork / join by
  // events on the context's stream; not used while a step graph is being recorded.
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_plan = nullptr;
  bool aligned = false;          // the last build padded the sibling groups to even ids (what the pair walk needs)
  bool plan_pending = false;     // the schedule for the full range was queued on the side stream by the last build
  int plan_n = -1, plan_cap = 0;
};

static void tree_release(nbody_hip_tree* g) {
  if (!g) return;
  void* ptrs[] = {g->d_enc, g->d_hist, g->d_root, g->d_level_base, g->d_keys_a, g->d_keys_b, g->d_idx_a,
                  g->d_idx_b, g->d_sorted, g->d_plane, g->d_rank_off, g->d_totals, g->d_level_real, g->d_last_tmp,
                  g->t.first, g->t.last, g->t.child0, g->t.child_last, g->t.rec, g->t.m, g->t.pb,
                  g->d_tmp, g->d_visits, g->d_partial, g->d_prefix, g->d_cost, g->d_order, g->d_bounds};
  for (void* p : ptrs) (void)hipFree(p);
  if (g->h_sort_err) (void)hipHostFree(g->h_sort_err);
  for (hipEvent_t e : {g->ev_fork, g->ev_join, g->ev_plan})
    if (e) (void)hipEventDestroy(e);
  if (g->side) (void)hipStreamDestroy(g->side);
  delete g;
}

template <class T>
static hipError_t dmalloc(T** p, size_t count) {
  return hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
}

// everything whose size depends on the tree shape: keys (32- or 64-bit), the level-major flag / scan
// arrays, the sort / scan scratch and the node arrays
static void tree_sort_self_test(hipStream_t st);  // (defined below nbody_hip_tree_create)

static int tree_alloc_nodes(nbody_hip_tree* g) {
  void* ptrs[] = {g->t.first, g->t.last, g->t.child0, g->t.child_last, g->t.rec, g->t.m, g->t.pb,
                  g->d_keys_a, g->d_keys_b, g->d_plane, g->d_rank_off, g->d_last_tmp, g->d_tmp};
  for (void* p : ptrs) (void)hipFree(p);
  g->t = TreeArrays{};
  g->d_keys_a = g->d_keys_b = nullptr;
  g->d_plane = nullptr;
  g->d_rank_off = nullptr;
  g->d_last_tmp = nullptr;
  g->d_tmp = nullptr;
  const size_t n = g->max_particles;
  // leaves <= n; internal nodes per level <= n / (leaf_max + 1)
  // leaves <= n; internal nodes per level <= n / (leaf_max + 1), each with at most one padding id for its children
  size_t cap = n + 2 * (size_t)(g->max_depth + 1) * (n / (size_t)(g->leaf_max + 1) + 1) + 16;
  // 28-bit child ids.  A tree that would need more nodes than the arrays hold is cut where the numbering passes the
  // capacity: the nodes beyond do not exist, their parents become leaves of several bodies (exact body-by-body
  // interactions; tree_fill_kernel / store_node flag them from the UNCLAMPED node total) -- slower, never wrong.
  if (cap > 0x0fffffffu) cap = 0x0fffffffu;
  if (g->node_limit > 0 && (size_t)g->node_limit < cap) cap = (size_t)g->node_limit;
  g->capacity = (int)cap;
  const size_t kbytes = n * (g->wide() ? sizeof(unsigned long long) : sizeof(unsigned int));
  g->rank_G = (int)((n + 64) / 64);  // positions 0 .. n (a marker may sit one past the last body)
  const size_t nrank = 2 * (size_t)(g->max_depth + 1) * (size_t)g->rank_G;
  hipError_t e = hipMalloc(&g->d_keys_a, kbytes);
  if (e == hipSuccess) e = hipMalloc(&g->d_keys_b, kbytes);
  if (e == hipSuccess) e = dmalloc(&g->d_plane, nrank + nrank / 2);  // + the cumulative planes (tree_flags_kernel)
  if (e == hipSuccess) e = dmalloc(&g->d_rank_off, nrank);
  if (e == hipSuccess) {
    size_t t1 = 0;
    if (g->wide())
      e = rocprim::radix_sort_pairs<SortConfig64>(nullptr, t1, static_cast<unsigned long long*>(g->d_keys_a),
                                                static_cast<unsigned long long*>(g->d_keys_b), g->d_idx_a, g->d_idx_b,
                                                n, 0, 63, g->ctx->stream);
    else
      e = rocprim::radix_sort_pairs<SortConfig>(nullptr, t1, static_cast<unsigned int*>(g->d_keys_a),
                                                static_cast<unsigned int*>(g->d_keys_b), g->d_idx_a, g->d_idx_b, n, 0,
                                                30, g->ctx->stream);
    g->own_sort_from = own_sort_from(kOwnSortFromTree);
    if (e == hipSuccess && g->wide() && NBH_BH_RADIX_BITS > 0 && n >= g->own_sort_from)
      tree_sort_self_test(g->ctx->stream);  // (once per process: the driver and the hand-written sort against the public sort)
    if (e == hipSuccess && g->wide() && NBH_BH_RADIX_BITS > 0 && n >= g->own_sort_from) {  // a radix sort of our own: room for both
      size_t t2 = 0, t3 = 0;
      if (onesweep::usable())
        e = onesweep::sort_pairs<NBH_BH_RADIX_BITS ? NBH_BH_RADIX_BITS : 8>(
            nullptr, t2, static_cast<const unsigned long long*>(g->d_keys_a), static_cast<unsigned long long*>(g->d_keys_b),
            g->d_idx_a, g->d_idx_b, n, 0, 63, g->ctx->stream);
      if (e == hipSuccess)
        e = radix::sort_pairs<unsigned long long, false>(nullptr, t3, static_cast<const unsigned long long*>(g->d_keys_a),
                                                         static_cast<unsigned long long*>(g->d_keys_b), nullptr, nullptr, g->d_idx_a,
                                                         g->d_idx_b, n, 0, 63, g->ctx->stream, nullptr);
      t1 = std::max(t1, std::max(t2, t3));
      const char* env = std::getenv("NBH_SORT");
      g->sort_impl = (NBH_BH_OWN_SORT && onesweep::usable()) ? 1 : 2;
      if (env && std::strcmp(env, "own") == 0) g->sort_impl = 2;
      if (env && std::strcmp(env, "public") == 0) g->sort_impl = 0;
      if (!g->h_sort_err) {
        if (hipHostMalloc(reinterpret_cast<void**>(&g->h_sort_err), 64, hipHostMallocMapped) == hipSuccess) {
          *g->h_sort_err = 0u;
          if (hipHostGetDevicePointer(reinterpret_cast<void**>(&g->h_sort_err_dev), g->h_sort_err, 0) != hipSuccess) g->h_sort_err_dev = nullptr;
        } else {
          g->h_sort_err = nullptr;
        }
        (void)hipGetLastError();
      }
    }
    g->tmp_bytes = t1;
    if (e == hipSuccess) e = hipMalloc(&g->d_tmp, g->tmp_bytes > 0 ? g->tmp_bytes : 16);
  }
  if (e == hipSuccess) e = dmalloc(&g->d_last_tmp, cap);
  if (e == hipSuccess) e = dmalloc(&g->t.first, cap);
  if (e == hipSuccess) e = dmalloc(&g->t.last, cap);
  if (e == hipSuccess) e = dmalloc(&g->t.child0, cap);
  if (e == hipSuccess) e = dmalloc(&g->t.child_last, cap);
  if (e == hipSuccess) e = dmalloc(&g->t.rec, cap + 8);  // + 8: sibling-group prefetch reads ahead
  if (e == hipSuccess) e = hipMemset(g->t.rec, 0, (cap + 8) * sizeof(NodeRec));
  if (e == hipSuccess) e = dmalloc(&g->t.m, cap);
  if (e == hipSuccess) e = dmalloc(&g->t.pb, (cap / 2 + 8) * kPairWords);  // + 8: group fetches read ahead
  if (e == hipSuccess) e = hipMemset(g->t.pb, 0, (cap / 2 + 8) * kPairWords * sizeof(unsigned int));
  if (e != hipSuccess)
    return NBH_FAIL(e == hipErrorOutOfMemory ? NBODY_HIP_ERR_RESOURCE : NBODY_HIP_ERR_DEVICE,
                    "Barnes-Hut tree allocation (%zu bodies, depth %d, %zu nodes): %s", n, g->max_depth, cap,
                    hipGetErrorString(e));
  g->t.scan_from = g->leaf_max > 1 ? 0 : g->max_depth;
  g->t.capacity = g->capacity;
  g->t.node_total = g->d_level_base + kMaxDepth + 2;  // the unclamped total (tree_fill_kernel)
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_tree_create(nbody_hip_ctx* ctx, size_t max_particles, nbody_hip_tree** out) {
  if (!ctx || !out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  *out = nullptr;
  if (max_particles == 0 || max_particles > 0x07ffffffu)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "max_particles out of range");
  NBH_HIP(hipSetDevice(ctx->device));
  nbody_hip_tree* g = new nbody_hip_tree();
  g->ctx = ctx;
  g->max_particles = max_particles;
  const size_t n = max_particles;
  hipError_t e = dmalloc(&g->d_enc, 16);
  if (e == hipSuccess) e = dmalloc(&g->d_hist, 2 * kTreeHistCopies * kTreeHistWords);
  if (e == hipSuccess) e = dmalloc(&g->d_root, 1);
  if (e == hipSuccess) e = dmalloc(&g->d_level_base, kMaxDepth + 3);
  if (e == hipSuccess) e = dmalloc(&g->d_totals, 2 * (kMaxDepth + 3));
  if (e == hipSuccess) e = dmalloc(&g->d_level_real, kMaxDepth + 3);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->side, hipStreamNonBlocking);
  for (hipEvent_t* ev : {&g->ev_fork, &g->ev_join, &g->ev_plan})
    if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
  if (e == hipSuccess) e = dmalloc(&g->d_idx_a, n);
  if (e == hipSuccess) e = dmalloc(&g->d_idx_b, n);
  if (e == hipSuccess) e = dmalloc(&g->d_sorted, n);
  if (e == hipSuccess) e = dmalloc(&g->d_visits, kVisitWords);
  if (e == hipSuccess) e = dmalloc(&g->d_partial, (size_t)3 * kSplitBudget);
  g->prefix_cap = std::min((size_t)kPrefixMax, n) + 8;
  if (e == hipSuccess) e = dmalloc(&g->d_prefix, g->prefix_cap + 2 * (kPrefixMax / kPrefixBlock + 2));
  {
    const size_t waves = n / 64 + 1, cap = waves / 8 + waves / 32 + 8;
    if (e == hipSuccess) e = dmalloc(&g->d_cost, waves);
    if (e == hipSuccess) e = dmalloc(&g->d_order, 8 * cap);
    if (e == hipSuccess) e = dmalloc(&g->d_bounds, 16);
  }
  if (e != hipSuccess) {
    tree_release(g);
    return NBH_FAIL(e == hipErrorOutOfMemory ? NBODY_HIP_ERR_RESOURCE : NBODY_HIP_ERR_DEVICE,
                    "Barnes-Hut tree allocation: %s", hipGetErrorString(e));
  }
  if (int rc = tree_alloc_nodes(g)) {
    tree_release(g);
    return rc;
  }
  *out = g;
  return NBODY_HIP_OK;
}

// Run-time half of the dependency fence of onesweep.h for the tree's instantiation (64-bit Morton keys, index payload):
// one buffer through the Onesweep driver and through the public rocprim::radix_sort_pairs, every output word compared.
// Once per process (the first tree that could take the driver).
static void tree_sort_self_test(hipStream_t st) {
#if NBH_BH_RADIX_BITS > 0
  static std::atomic<bool> done{false};
  if (done.exchange(true)) return;
  const size_t n = 200000;
  const unsigned first_bit = 3, end_bit = 63;  // (a depth-20 build sorts bits 3..62)
  std::vector<unsigned long long> hk(n);
  std::vector<int> hv(n);
  unsigned long long x = 88172645463325252ull;
  for (size_t i = 0; i < n; i++) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    hk[i] = (x >> 1) & ~0xfffffull;  // clustered low digits: many equal keys, so that stability shows
    hv[i] = (int)i;
  }
  constexpr int kV = 3;  // 0: the driver, 1: the public sort, 2: the hand-written sort
  const bool have[kV] = {NBH_BH_OWN_SORT && NBH_ONESWEEP_AVAILABLE, true, true};
  unsigned long long *k_in = nullptr, *k_out[kV] = {nullptr, nullptr, nullptr};
  int *v_in = nullptr, *v_out[kV] = {nullptr, nullptr, nullptr};
  unsigned int *err_h = nullptr, *err_d = nullptr;
  void* tmp = nullptr;
  size_t t1 = 0, t2 = 0, t3 = 0;
  bool ran = false, same[kV] = {false, true, false};
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&k_in), n * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&v_in), n * sizeof(int));
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&err_h), 64, hipHostMallocMapped);
  if (e == hipSuccess) {
    *err_h = 0u;
    e = hipHostGetDevicePointer(reinterpret_cast<void**>(&err_d), err_h, 0);
  }
  for (int v = 0; v < kV && e == hipSuccess; v++) {
    e = hipMalloc(reinterpret_cast<void**>(&k_out[v]), n * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&v_out[v]), n * sizeof(int));
  }
  if (e == hipSuccess) e = rocprim::radix_sort_pairs<SortConfig64>(nullptr, t1, k_in, k_out[1], v_in, v_out[1], n, first_bit, end_bit, st);
  if (e == hipSuccess && have[0]) e = onesweep::sort_pairs<NBH_BH_RADIX_BITS>(nullptr, t2, static_cast<const unsigned long long*>(k_in), k_out[0], v_in, v_out[0], n, first_bit, end_bit, st);
  if (e == hipSuccess) e = radix::sort_pairs<unsigned long long, false>(nullptr, t3, static_cast<const unsigned long long*>(k_in), k_out[2], nullptr, nullptr, v_in, v_out[2], n, first_bit, end_bit, st, nullptr);
  const size_t tb = std::max(t1, std::max(t2, t3));
  if (e == hipSuccess) e = hipMalloc(&tmp, tb > 0 ? tb : 16);
  if (e == hipSuccess) e = hipMemcpyAsync(k_in, hk.data(), n * sizeof(unsigned long long), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(v_in, hv.data(), n * sizeof(int), hipMemcpyHostToDevice, st);
  if (e == hipSuccess && have[0]) {
    size_t b = tb;
    e = onesweep::sort_pairs<NBH_BH_RADIX_BITS>(tmp, b, static_cast<const unsigned long long*>(k_in), k_out[0], v_in, v_out[0], n, first_bit, end_bit, st);
  }
  if (e == hipSuccess) {
    size_t b = tb;
    e = rocprim::radix_sort_pairs<SortConfig64>(tmp, b, k_in, k_out[1], v_in, v_out[1], n, first_bit, end_bit, st);
  }
  if (e == hipSuccess) {
    size_t b = tb;
    e = radix::sort_pairs<unsigned long long, false>(tmp, b, static_cast<const unsigned long long*>(k_in), k_out[2], nullptr, nullptr, v_in, v_out[2], n, first_bit, end_bit, st, err_d);
  }
  if (e == hipSuccess) {
    std::vector<unsigned long long> rk[kV];
    std::vector<int> rv[kV];
    for (int v = 0; v < kV && e == hipSuccess; v++) {
      if (!have[v]) continue;
      rk[v].resize(n); rv[v].resize(n);
      e = hipMemcpyAsync(rk[v].data(), k_out[v], n * sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipMemcpyAsync(rv[v].data(), v_out[v], n * sizeof(int), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) {
      ran = true;
      bool sorted = true;
      for (size_t i = 1; sorted && i < n; i++) sorted = (rk[1][i - 1] >> first_bit) <= (rk[1][i] >> first_bit);
      for (int v = 0; v < kV; v += 2)
        same[v] = have[v] && sorted && std::memcmp(rk[v].data(), rk[1].data(), n * sizeof(unsigned long long)) == 0 &&
                  std::memcmp(rv[v].data(), rv[1].data(), n * sizeof(int)) == 0;
      if (*err_h) same[2] = false;  // (a look-back of the hand-written sort gave up)
    }
  }
  (void)hipGetLastError();
  (void)hipFree(k_in); (void)hipFree(v_in); (void)hipFree(tmp);
  if (err_h) (void)hipHostFree(err_h);
  for (int v = 0; v < kV; v++) { (void)hipFree(k_out[v]); (void)hipFree(v_out[v]); }
  if (have[0]) nbh::onesweep::self_test_report(ran && same[0], "Barnes-Hut (64-bit keys, index payload)");
  nbh::radix::self_test_report(ran && same[2], "Barnes-Hut (64-bit keys, index payload)");
#else
  (void)st;
#endif
}

extern "C" int nbody_hip_tree_destroy(nbody_hip_tree* g) {
  if (!g) return NBODY_HIP_OK;
  NBH_DESTROY_BEGIN
  (void)hipSetDevice(g->ctx->device);
  if (g->side) (void)hipStreamSynchronize(g->side);
  (void)hipStreamSynchronize(g->ctx->stream);
  g->ctx->alloc_generation++;  // a step graph recorded with this tree is stale now
  tree_release(g);
  NBH_DESTROY_END
}

extern "C" int nbody_hip_tree_set_params(nbody_hip_tree* g, int max_depth, int leaf_max) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null tree");
  if (max_depth < 1 || max_depth > kMaxDepth)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "max_depth must be in [1, %d]", kMaxDepth);
  if (leaf_max < 1 || leaf_max > 1024)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "leaf_max must be in [1, 1024]");
  // node ids are 28-bit: a deep tree over very many bodies does not fit the bound on its node count
  if (max_depth > kDepth32 &&
      (size_t)g->max_particles + (size_t)(max_depth + 1) * (g->max_particles / (size_t)(leaf_max + 1) + 1) + 16 > 0x0fffffffu)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "max_depth %d with leaf_max %d needs more than 2^28 nodes for %zu bodies",
                    max_depth, leaf_max, g->max_particles);
  NBH_HIP(hipSetDevice(g->ctx->device));
  if (g->side) NBH_HIP(hipStreamSynchronize(g->side));
  NBH_HIP(hipStreamSynchronize(g->ctx->stream));
  g->max_depth = max_depth;
  g->leaf_max = leaf_max;
  g->built_count = 0;
  g->ctx->alloc_generation++;  // the node arrays are replaced: recorded step graphs are stale
  return tree_alloc_nodes(g);
}

// test / stress hook: cap the node arrays below the bound on the node count (0 = the bound).  A tree that needs more
// nodes is cut at the capacity (see tree_alloc_nodes): forces stay exact-or-better, only slower.
extern "C" int nbody_hip_tree_limit_nodes(nbody_hip_tree* g, int max_nodes) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null tree");
  if (max_nodes < 0 || (max_nodes > 0 && max_nodes < 16)) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "max_nodes must be 0 or >= 16");
  NBH_HIP(hipSetDevice(g->ctx->device));
  if (g->side) NBH_HIP(hipStreamSynchronize(g->side));
  NBH_HIP(hipStreamSynchronize(g->ctx->stream));
  g->node_limit = max_nodes;
  g->built_count = 0;
  g->ctx->alloc_generation++;
  return tree_alloc_nodes(g);
}

// the build proper, from packed bodies
// soa != nullptr: posm is a scratch array to be filled from the SoA bodies (fused with the bounding box)
// drift_dt: soa is a step's state BEFORE its drift; the drift rides on the packing pass (nbody_hip_tree_drift_build)
static int tree_build_packed(nbody_hip_tree* g, float4* posm, size_t n, const nbody_particle_data* soa = nullptr,
                             const float* drift_dt = nullptr) {
  nbody_hip_ctx* ctx = g->ctx;
  hipStream_t st = ctx->stream;
  const int ni = (int)n;
  const int blocks = (ni + kBlock - 1) / kBlock;
  if (g->h_sort_err && *g->h_sort_err)
    return NBH_FAIL(NBODY_HIP_ERR_DEVICE, "the radix sort of an earlier build gave up in its look-back (csrc/radix_sort.h)");

  // two bounding-box buffers alternate: this build's was re-armed by the previous build's morton_kernel
  unsigned int* enc = g->d_enc + 8 * (g->enc_flip & 1);
  unsigned int* enc_next = g->d_enc + 8 * ((g->enc_flip + 1) & 1);
  unsigned int* hist = g->d_hist + kTreeHistCopies * kTreeHistWords * (g->enc_flip & 1);          // (zeroed by the previous build's
  unsigned int* hist_next = g->d_hist + kTreeHistCopies * kTreeHistWords * ((g->enc_flip + 1) & 1);  //  morton_kernel, like the box)
  g->enc_flip++;
  // armed: false on the first build and after a build that failed half way.  A recorded step graph replays this
  // very launch sequence on the SAME buffer every time, so while capturing (and right after) the init stays in.
  const bool armed = g->enc_armed && !ctx->capturing && g->enc_replays == ctx->graph_replays;
  g->enc_armed = false;
  if (soa && drift_dt) {
    if (int rc = launch_drift_pack_bbox(ctx, const_cast<nbody_particle_data*>(soa), *drift_dt, posm, enc, !armed)) return rc;
  } else if (soa) {
    if (int rc = launch_pack_bbox(ctx, soa->pos_x, soa->pos_y, soa->pos_z, soa->mass, ni, posm, enc, !armed)) return rc;
  } else {
    if (int rc = launch_bbox(ctx, posm, ni, enc, !armed)) return rc;
  }
  const bool side_ok = g->side != nullptr && !ctx->capturing;
  // the walk's cost-ordered schedule depends only on the visit counts the previous walk of the full range recorded:
  // queue it on the side stream now, beside the build (28 us off the step's critical path)
  g->plan_pending = false;
  {
    const int waves = (ni + 63) / 64;
    if (side_ok && g->tune_schedule && g->cost_first == 0 && g->cost_n == ni && waves >= 2048) {
      const int cap = waves / 8 + waves / 32 + 8;
      NBH_HIP(hipEventRecord(g->ev_fork, st));
      NBH_HIP(hipStreamWaitEvent(g->side, g->ev_fork, 0));
      hipLaunchKernelGGL(walk_plan_kernel, dim3(1), dim3(kPlanBlock), 0, g->side, g->d_cost, waves, cap, g->d_bounds);
      hipLaunchKernelGGL(walk_order_kernel, dim3(8), dim3(kPlanBlock), 0, g->side, g->d_cost, g->d_bounds, cap, g->d_order);
      NBH_LAUNCH_CHECK();
      NBH_HIP(hipEventRecord(g->ev_plan, g->side));
      g->plan_pending = true;
      g->plan_n = ni;
      g->plan_cap = cap;
    }
  }
  // topology of every level: keys, sort, flags, ranks, fill (see tree_flags_kernel)
  const int levels = g->max_depth + 1;
  const bool fused = ni <= kPrefixMax;
  bool prefix_forked = false;
  auto topology = [&](auto* ka, auto* kb, int first_bit, int key_bits) -> int {
    using K = std::remove_pointer_t<decltype(ka)>;
    // the ranks are laid out for THIS build's body count: G groups of 64 over the positions 0 .. ni
    const int G = (ni + 64) / 64;
    const size_t tbl = (size_t)levels * (size_t)G;
    unsigned long long* odd_plane = g->d_plane + tbl;
    // even-aligned sibling groups are what the pair walk needs: trees it will walk (from kPairFrom bodies, or when that
    // walk form is forced) get them, smaller trees keep plain ids and save three launches
    g->aligned = ni >= kPairFrom || g->tune_form == 2;
    bool own_sort = false;  // a radix sort of our own (1 = driver, 2 = hand-written): the key kernel counts its digits
    int impl = 0;
    size_t sort_words = 0;
    if constexpr (sizeof(K) == 8 && NBH_BH_RADIX_BITS > 0) {
      impl = n >= g->own_sort_from ? g->sort_impl : 0;
      if (impl == 1 && !(NBH_BH_OWN_SORT && onesweep::usable())) impl = 2;
      if (impl == 2 && (!radix::usable() || !g->h_sort_err_dev)) impl = 0;
      own_sort = impl != 0;
      if (impl == 1) sort_words = onesweep::clear_words<NBH_BH_RADIX_BITS ? NBH_BH_RADIX_BITS : 8>(n, (unsigned)first_bit, (unsigned)key_bits);
    }
    int hist_places = 0;
    if constexpr (sizeof(K) == 8 && NBH_BH_RADIX_BITS > 0)
      if (own_sort) hist_places = (key_bits - first_bit + NBH_BH_RADIX_BITS - 1) / NBH_BH_RADIX_BITS;
    if (hist_places && !armed) NBH_HIP(hipMemsetAsync(hist, 0, kTreeHistCopies * kTreeHistWords * sizeof(unsigned int), st));
    hipLaunchKernelGGL(morton_kernel<K>, dim3(std::min((ni + NBH_HIST_THREADS - 1) / NBH_HIST_THREADS, NBH_HIST_BLOCKS)), dim3(NBH_HIST_THREADS), 0, st, posm, ni, enc, enc_next,
                       g->d_root, g->d_level_base, ka, g->d_idx_a, static_cast<unsigned int*>(g->d_tmp),
                       (unsigned int)sort_words, reinterpret_cast<unsigned int*>(odd_plane),
                       g->aligned ? (unsigned int)(2 * tbl) : 0u, hist, hist_next, hist_places, first_bit);
    NBH_LAUNCH_CHECK();
    g->enc_armed = !ctx->capturing;
    g->enc_replays = ctx->graph_replays;
    size_t tmp = g->tmp_bytes;
    using Cfg = std::conditional_t<sizeof(K) == 8, SortConfig64, SortConfig>;
    // above rocPRIM's merge-sort range: its Onesweep kernels under our own driver (no fill launches, onesweep.h)
    if constexpr (sizeof(K) == 8 && NBH_BH_RADIX_BITS > 0) {
      if (impl == 1) {
        NBH_HIP(onesweep::sort_pairs<NBH_BH_RADIX_BITS ? NBH_BH_RADIX_BITS : 8>(
            g->d_tmp, tmp, static_cast<const K*>(ka), kb, g->d_idx_a, g->d_idx_b, n, (unsigned)first_bit, (unsigned)key_bits, st,
            /*cleared=*/true, hist_places ? hist : nullptr, kTreeHistCopies, kTreeHistWords));
      } else if (impl == 2) {
        NBH_HIP((radix::sort_pairs<K, false>(g->d_tmp, tmp, static_cast<const K*>(ka), kb, nullptr, nullptr, g->d_idx_a, g->d_idx_b, n,
                                             (unsigned)first_bit, (unsigned)key_bits, st, g->h_sort_err_dev, hist_places ? hist : nullptr,
                                             hist_places ? kTreeHistCopies : 1, hist_places ? (unsigned)kTreeHistWords : 0u)));
      } else {
        NBH_HIP(rocprim::radix_sort_pairs<Cfg>(g->d_tmp, tmp, ka, kb, g->d_idx_a, g->d_idx_b, n, first_bit, key_bits, st));
      }
    } else {
      NBH_HIP(rocprim::radix_sort_pairs<Cfg>(g->d_tmp, tmp, ka, kb, g->d_idx_a, g->d_idx_b, n, first_bit, key_bits, st));
    }
    unsigned int* lvlmask = reinterpret_cast<unsigned int*>(g->d_idx_a);  // idx_a is free after the sort
    int* odd_off = g->d_rank_off + tbl;
    int* odd_totals = g->d_totals + (kMaxDepth + 3);
    hipLaunchKernelGGL(tree_flags_kernel<K>, dim3((unsigned)((ni + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, kb, ni,
                       g->max_depth, g->leaf_max, lvlmask, posm, g->d_idx_b, g->d_sorted, g->d_plane, g->d_plane + 2 * tbl, G);
    if (fused && side_ok) {
      // the prefix sums of the sorted bodies only feed the monopole pass: beside the ranks + fill pass
      const int pblocks = ni / kPrefixBlock + 1;
      dd4* btot = g->d_prefix + g->prefix_cap;
      dd4* boff = btot + kPrefixMax / kPrefixBlock + 2;
      NBH_HIP(hipEventRecord(g->ev_fork, st));
      NBH_HIP(hipStreamWaitEvent(g->side, g->ev_fork, 0));
      hipLaunchKernelGGL(prefix_bodies_kernel, dim3(pblocks), dim3(kPrefixBlock), 0, g->side, g->d_sorted, ni, g->d_prefix, btot);
      hipLaunchKernelGGL(prefix_blocks_kernel, dim3(1), dim3(kPrefixBlock), 0, g->side, pblocks, btot, boff);
      NBH_HIP(hipEventRecord(g->ev_join, g->side));
      prefix_forked = true;
    }
    const LevelRanks lr{g->d_plane, g->d_rank_off, G}, odd{odd_plane, odd_off, G};
    // node ranks -> (aligned: body ranges + odd-group markers -> marker ranks) -> fill
    hipLaunchKernelGGL(level_scan_kernel, dim3(levels, 1), dim3(kScanBlock), 0, st, g->d_plane, g->d_rank_off, g->d_totals,
                       odd_plane, odd_off, odd_totals, G);
    if (g->aligned) {
      hipLaunchKernelGGL(tree_span_kernel<K>, dim3(blocks), dim3(kBlock), 0, st, kb, ni, g->max_depth, g->leaf_max,
                         reinterpret_cast<const unsigned int*>(g->d_idx_a), lr, g->d_totals, g->capacity, g->d_last_tmp, odd_plane,
                         g->d_plane + 2 * tbl);
      hipLaunchKernelGGL(level_scan_kernel, dim3(levels, 1), dim3(kScanBlock), 0, st, odd_plane, odd_off, odd_totals,
                         odd_plane, odd_off, odd_totals, G);
      hipLaunchKernelGGL((tree_fill_kernel<K, true>), dim3(blocks), dim3(kBlock), 0, st, kb, ni, g->max_depth, g->leaf_max,
                         reinterpret_cast<const unsigned int*>(g->d_idx_a), lr, odd, g->d_totals, odd_totals, g->d_last_tmp,
                         g->t, g->capacity, g->d_level_base, g->d_level_real);
    } else {
      hipLaunchKernelGGL((tree_fill_kernel<K, false>), dim3(blocks), dim3(kBlock), 0, st, kb, ni, g->max_depth, g->leaf_max,
                         reinterpret_cast<const unsigned int*>(g->d_idx_a), lr, odd, g->d_totals, odd_totals, g->d_last_tmp,
                         g->t, g->capacity, g->d_level_base, g->d_level_real);
    }
    NBH_LAUNCH_CHECK();
    return NBODY_HIP_OK;
  };
  if (g->wide()) {
    // only the 3 * max_depth leading bits of the 63 shape the tree (bodies that share them share a leaf of the
    // deepest level, where the order is immaterial: such a leaf interacts body by body)
    if (int rc = topology(static_cast<unsigned long long*>(g->d_keys_a), static_cast<unsigned long long*>(g->d_keys_b),
                          63 - 3 * g->max_depth, 63))
      return rc;
  } else {
    if (int rc = topology(static_cast<unsigned int*>(g->d_keys_a), static_cast<unsigned int*>(g->d_keys_b), 0, 30)) return rc;
  }
  if (fused) {
    // trees of <= kPrefixMax bodies: every node's monopole from the double-double prefix sums of the sorted bodies
    // (ni + 1 prefix entries: entry ni, the total, is thread ni's "sum before")
    const int pblocks = ni / kPrefixBlock + 1;
    dd4* btot = g->d_prefix + g->prefix_cap;
    dd4* boff = btot + kPrefixMax / kPrefixBlock + 2;
    if (prefix_forked) {
      NBH_HIP(hipStreamWaitEvent(st, g->ev_join, 0));
    } else {
      hipLaunchKernelGGL(prefix_bodies_kernel, dim3(pblocks), dim3(kPrefixBlock), 0, st, g->d_sorted, ni, g->d_prefix, btot);
      hipLaunchKernelGGL(prefix_blocks_kernel, dim3(1), dim3(kPrefixBlock), 0, st, pblocks, btot, boff);
    }
    // (about two ids per body in practice; a fixed grid strides over whatever the numbering arrived at)
    const size_t want = (2 * n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(prefix_monopole_kernel, dim3((unsigned)std::min<size_t>(std::max<size_t>(want, 1), 16384)), dim3(kBlock), 0, st,
                       g->d_level_base, g->max_depth, g->d_sorted, PrefixSums{g->d_prefix, boff}, g->d_root, g->t);
  } else {
    // monopoles bottom-up: wide levels one launch each, the narrow top in a single workgroup
    const int top = g->max_depth < 4 ? g->max_depth : 4;
    for (int L = g->max_depth; L > top; L--)
      hipLaunchKernelGGL(level_monopole_kernel, dim3(1024), dim3(kBlock), 0, st, L, g->d_level_base,
                         g->d_sorted, g->d_root, g->t);
    hipLaunchKernelGGL(top_monopole_kernel, dim3(1), dim3(kTopBlock), 0, st, top, g->d_level_base, g->d_sorted,
                       g->d_root, g->t);
  }
  NBH_LAUNCH_CHECK();
  g->built_count = n;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_tree_build(nbody_hip_tree* g, const nbody_particle_data* d) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null tree");
  if (!d) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null particle data");
  const size_t n = d->count;
  if (n == 0) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Particle count must be greater than 0");
  if (n > g->max_particles)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "particle count %zu exceeds the tree's capacity %zu "
                    "(sized from the first count seen, ref: force_barnes_hut.cu:527-529)", n, g->max_particles);
  if (!d->pos_x || !d->pos_y || !d->pos_z || !d->mass)
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "particle data has null arrays");
  nbody_hip_ctx* ctx = g->ctx;
  NBH_HIP(hipSetDevice(ctx->device));
  if (int rc = ctx->posm.reserve(n * sizeof(float4))) return rc;
  float4* posm = static_cast<float4*>(ctx->posm.ptr);
  return tree_build_packed(g, posm, n, d);
}

extern "C" int nbody_hip_tree_drift_build(nbody_hip_tree* g, nbody_particle_data* d, float dt) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null tree");
  if (!d) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null particle data");
  const size_t n = d->count;
  if (n == 0) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Particle count must be greater than 0");
  if (n > g->max_particles)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "particle count %zu exceeds the tree's capacity %zu "
                    "(sized from the first count seen, ref: force_barnes_hut.cu:527-529)", n, g->max_particles);
  if (!d->pos_x || !d->pos_y || !d->pos_z || !d->mass || !d->vel_x || !d->vel_y || !d->vel_z || !d->acc_x || !d->acc_y ||
      !d->acc_z || !d->acc_old_x || !d->acc_old_y || !d->acc_old_z)
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "particle data has null arrays");
  nbody_hip_ctx* ctx = g->ctx;
  NBH_HIP(hipSetDevice(ctx->device));
  if (int rc = ctx->posm.reserve(n * sizeof(float4))) return rc;
  float4* posm = static_cast<float4*>(ctx->posm.ptr);
  return tree_build_packed(g, posm, n, d, &dt);
}

extern "C" int nbody_hip_tree_build_packed(nbody_hip_tree* g, const nbody_float4* posm, size_t n) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null tree");
  if (!posm) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (n == 0) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Particle count must be greater than 0");
  if (n > g->max_particles)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "particle count %zu exceeds the tree's capacity %zu", n, g->max_particles);
  NBH_HIP(hipSetDevice(g->ctx->device));
  return tree_build_packed(g, const_cast<float4*>(reinterpret_cast<const float4*>(posm)), n);
}

// walk of the sorted bodies [first, first + count); output SoA (ax, ay, az) or float4 (acc4), at
// the bodies' ORIGINAL indices
static int tree_walk(nbody_hip_tree* g, int first, int count, float theta, float G, float eps, float* ax,
                     float* ay, float* az, float4* acc4) {
  if (!(theta >= 0.0f) || theta > 2.0f)  // ref: validateTheta, error_handling.cpp:115-123
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "Barnes-Hut theta must be between 0 and 2");
  nbody_hip_ctx* ctx = g->ctx;
  NBH_HIP(hipSetDevice(ctx->device));
  const int n = count;
  const int blocks = (n + kBlock - 1) / kBlock;
  const float eps2 = eps * eps, theta2 = theta * theta;  // :494-496
  unsigned long long* visits = g->count_visits ? g->d_visits : nullptr;
  if (visits) NBH_HIP(hipMemsetAsync(visits, 0, kVisitWords * sizeof(unsigned long long), ctx->stream));
  // a schedule queued beside the build reads the cost array this walk is about to rewrite: order after it in any case
  if (g->plan_pending && !ctx->capturing) NBH_HIP(hipStreamWaitEvent(ctx->stream, g->ev_plan, 0));
  if (n == 0) return NBODY_HIP_OK;
  // replicas of the walk when there are too few waves to hide the fetch latency (see the kernel)
  int K = 1;
  while (K < kMaxReplicas && (size_t)(2 * K) * (size_t)n <= (size_t)kSplitAuto) K *= 2;
  if (n >= kPairFrom) K = 1;  // the pair walk (8 waves per SIMD, scheduled) overtakes two replicas from here
  if (g->tune_replicas > 0) {
    K = 1;
    while (K < g->tune_replicas && K < kMaxReplicas && (size_t)(2 * K) * (size_t)n <= (size_t)kSplitBudget) K *= 2;
  }
  // ownership units of at most n_total / 96 bodies (of the whole tree, not of the walked range): measured
  // best at K x (units per replica) ~ 64..128 for every K (tools/bh_split_sweep.py)
  const size_t units = g->tune_split_level > 0 ? (size_t)g->tune_split_level * (size_t)K : 96;
  int unit_max = (int)(g->built_count / units);
  if (unit_max < 1) unit_max = 1;
  const bool guard = eps2 < 1e-12f;
  // walk without replicas: the pair walk unless the plain one is asked for -- or the tree was built with plain ids
  // (fewer than kPairFrom bodies and the pair form not forced when it was built): its pair blocks are not group-aligned
  int form = g->tune_form > 0 ? g->tune_form : 2;
  if (!g->aligned) {
    if (g->tune_form == 2)
      return NBH_FAIL(NBODY_HIP_ERR_STATE, "the pair walk was asked for after a build with plain node ids (%zu bodies < %d): "
                      "set the walk form before the build", g->built_count, kPairFrom);
    form = 1;
  }
#define NBH_BH_LAUNCH(GD, SP, GRID)                                                                       \
  hipLaunchKernelGGL((bh_traverse_kernel<GD, SP>), GRID, dim3(kBlock), 0, ctx->stream, g->t.rec, g->d_sorted, \
                     g->d_idx_b, first, n, theta2, eps2, G, ax, ay, az, acc4, visits,                     \
                     unit_max, g->d_partial)
  if (K == 1 && visits && g->tune_form != 2) form = 1;  // counting on: the diagnostics instantiation of the
  if (K == 1 && visits && form == 1) {                  // plain walk (the pair walk visits the same nodes)
    if (guard) hipLaunchKernelGGL((bh_traverse_kernel<true, false, true>), dim3(blocks), dim3(kBlock), 0, ctx->stream,
                                  g->t.rec, g->d_sorted, g->d_idx_b, first, n, theta2, eps2, G, ax, ay, az, acc4, visits,
                                  unit_max, g->d_partial);
    else hipLaunchKernelGGL((bh_traverse_kernel<false, false, true>), dim3(blocks), dim3(kBlock), 0, ctx->stream,
                            g->t.rec, g->d_sorted, g->d_idx_b, first, n, theta2, eps2, G, ax, ay, az, acc4, visits,
                            unit_max, g->d_partial);
  } else if (K == 1 && form == 2) {
    // one wave per workgroup: the dispatcher refills single wave slots (4-wave groups: +1.5 %, 16-wave: +7 %)
    const int waves = (n + 63) / 64;
    const int* order = nullptr;
    int grid = waves;
    if (g->tune_schedule && g->cost_first == first && g->cost_n == n && waves >= 2048) {
      // the previous walk of this range recorded every wave's node visits: longest first, equal cost per XCD
      const int cap = waves / 8 + waves / 32 + 8;
      if (g->plan_pending && first == 0 && g->plan_n == n && g->plan_cap == cap && !ctx->capturing) {
        // queued beside the build (tree_build_packed); the stream already waits for it (above)
      } else {
        hipLaunchKernelGGL(walk_plan_kernel, dim3(1), dim3(kPlanBlock), 0, ctx->stream, g->d_cost, waves, cap,
                           g->d_bounds);
        hipLaunchKernelGGL(walk_order_kernel, dim3(8), dim3(kPlanBlock), 0, ctx->stream, g->d_cost, g->d_bounds, cap,
                           g->d_order);
      }
      order = g->d_order;
      grid = 8 * cap;
    }
    g->plan_pending = false;  // a second walk of the same build recorded new costs: it plans for itself
    PairArgs pa;
    pa.pb = g->t.pb; pa.rec = g->t.rec; pa.sorted = g->d_sorted; pa.order = order;
    pa.t_first = first; pa.n = n; pa.theta2 = theta2; pa.eps2 = eps2;
    pa.idx = g->d_idx_b; pa.acc_x = ax; pa.acc_y = ay; pa.acc_z = az; pa.acc4 = acc4;
    pa.visit_count = visits; pa.cost_out = g->d_cost; pa.G = G;
    if (guard) hipLaunchKernelGGL((bh_traverse_pair_kernel<true, 1>), dim3(grid), dim3(64), 0, ctx->stream, pa);
    else hipLaunchKernelGGL((bh_traverse_pair_kernel<false, 1>), dim3(grid), dim3(64), 0, ctx->stream, pa);
    g->cost_first = first;
    g->cost_n = n;
  } else if (K == 1) {
    if (guard) NBH_BH_LAUNCH(true, false, dim3(blocks)); else NBH_BH_LAUNCH(false, false, dim3(blocks));
  } else {
    if (guard) NBH_BH_LAUNCH(true, true, dim3(blocks, K)); else NBH_BH_LAUNCH(false, true, dim3(blocks, K));
    hipLaunchKernelGGL(bh_combine_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, g->d_partial, K,
                       g->d_idx_b, first, n, G, ax, ay, az, acc4);
  }
#undef NBH_BH_LAUNCH
  NBH_LAUNCH_CHECK();
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_tree_compute_forces(nbody_hip_tree* g, nbody_particle_data* d, float theta,
                                             float G, float eps) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null tree");
  if (!d) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null particle data");
  if (g->built_count == 0 || g->built_count != d->count)
    return NBH_FAIL(NBODY_HIP_ERR_STATE, "tree was not built for this particle set");
  if (!d->acc_x || !d->acc_y || !d->acc_z) return NBH_FAIL(NBODY_HIP_ERR_STATE, "particle data has null arrays");
  return tree_walk(g, 0, (int)g->built_count, theta, G, eps, d->acc_x, d->acc_y, d->acc_z, nullptr);
}

extern "C" int nbody_hip_tree_compute_forces_packed(nbody_hip_tree* g, size_t first_sorted, size_t count,
                                                    float theta, float G, float eps, nbody_float4* acc_out) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null tree");
  if (!acc_out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  if (g->built_count == 0) return NBH_FAIL(NBODY_HIP_ERR_STATE, "tree has not been built");
  if (first_sorted > g->built_count || count > g->built_count - first_sorted)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "range [%zu, %zu) exceeds the %zu bodies of the tree", first_sorted,
                    first_sorted + count, g->built_count);
  return tree_walk(g, (int)first_sorted, (int)count, theta, G, eps, nullptr, nullptr, nullptr,
                   reinterpret_cast<float4*>(acc_out));
}

extern "C" int nbody_hip_tree_count_visits(nbody_hip_tree* g, int enable) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null tree");
  g->count_visits = enable != 0;
  if (!g->count_visits) {
    NBH_HIP(hipSetDevice(g->ctx->device));
    NBH_HIP(hipMemsetAsync(g->d_visits, 0, kVisitWords * sizeof(unsigned long long), g->ctx->stream));
  }
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_tree_visit_histogram(nbody_hip_tree* g, unsigned long long out[130]) {
  if (!g || !out) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null argument");
  NBH_NOT_CAPTURABLE(g->ctx, "tree inspection");
  NBH_HIP(hipSetDevice(g->ctx->device));
  NBH_HIP(hipStreamSynchronize(g->ctx->stream));
  NBH_HIP(hipMemcpy(out, g->d_visits + 1, 130 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return NBODY_HIP_OK;
}

// experiments: replicas / units per replica of the split traversal (0 = automatic)
extern "C" int nbody_hip_tree_tuning(nbody_hip_tree* g, int replicas, int split_level) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null tree");
  if (replicas < 0 || replicas > kMaxReplicas || split_level < 0 || split_level > 1024)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "replicas must be in [0, %d], units per replica in [0, 1024]", kMaxReplicas);
  g->tune_replicas = replicas;
  g->tune_split_level = split_level;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_tree_walk_form(nbody_hip_tree* g, int form) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null tree");
  if (form < 0 || form > 3) return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "walk form must be 0 (automatic), 1, 2 or 3");
  g->tune_form = form == 3 ? 2 : form;
  g->tune_schedule = form != 3;
  return NBODY_HIP_OK;
}

extern "C" int nbody_hip_tree_stats(nbody_hip_tree* g, int* node_count, float* root_mass,
                                    unsigned long long* nodes_visited_per_wave_total,
                                    int level_base_out[NBODY_HIP_TREE_LEVELS]) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null tree");
  if (g->built_count == 0) return NBH_FAIL(NBODY_HIP_ERR_STATE, "tree has not been built");
  nbody_hip_ctx* ctx = g->ctx;
  NBH_NOT_CAPTURABLE(ctx, "tree inspection");
  NBH_HIP(hipSetDevice(ctx->device));
  NBH_HIP(hipStreamSynchronize(ctx->stream));
  // ids have holes (even-aligned sibling groups): what is reported is the number of NODES, per level and in all
  int lb[kMaxDepth + 3], ids[kMaxDepth + 3];
  NBH_HIP(hipMemcpy(lb, g->d_level_real, sizeof(lb), hipMemcpyDeviceToHost));
  NBH_HIP(hipMemcpy(ids, g->d_level_base, sizeof(ids), hipMemcpyDeviceToHost));
  int nodes = lb[g->max_depth + 1];
  if (ids[kMaxDepth + 2] > g->capacity) {  // the tree was cut at the capacity: count what exists
    std::vector<int> first((size_t)g->capacity);
    NBH_HIP(hipMemcpy(first.data(), g->t.first, first.size() * sizeof(int), hipMemcpyDeviceToHost));
    nodes = 0;
    for (int nid = 0; nid < ids[g->max_depth + 1]; nid++) nodes += first[nid] >= 0;
  }
  if (node_count) *node_count = nodes;
  if (level_base_out)
    for (int k = 0; k < NBODY_HIP_TREE_LEVELS; k++) level_base_out[k] = k <= g->max_depth + 1 ? lb[k] : lb[g->max_depth + 1];
  if (root_mass) {
    NodeRec r;
    NBH_HIP(hipMemcpy(&r, g->t.rec, sizeof(r), hipMemcpyDeviceToHost));
    *root_mass = r.mass;
  }
  if (nodes_visited_per_wave_total)
    NBH_HIP(hipMemcpy(nodes_visited_per_wave_total, g->d_visits, sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return NBODY_HIP_OK;
}

// ref: BarnesHutTree::copyNodesToHost / getNodes (force_barnes_hut.cu:500-503): the tree in the
// reference's OctreeNode layout, children indexed by octant (x = bit 2, y = bit 1, z = bit 0).
extern "C" int nbody_hip_tree_copy_nodes(nbody_hip_tree* g, void* host_nodes, int capacity_nodes,
                                         int* sorted_indices) {
  if (!g) return NBH_FAIL(NBODY_HIP_ERR_STATE, "null tree");
  if (g->built_count == 0) return NBH_FAIL(NBODY_HIP_ERR_STATE, "tree has not been built");
  nbody_hip_ctx* ctx = g->ctx;
  NBH_NOT_CAPTURABLE(ctx, "tree inspection");
  NBH_HIP(hipSetDevice(ctx->device));
  NBH_HIP(hipStreamSynchronize(ctx->stream));
  int lb[kMaxDepth + 3];
  NBH_HIP(hipMemcpy(lb, g->d_level_base, sizeof(lb), hipMemcpyDeviceToHost));
  const int nids = lb[g->max_depth + 1];  // ids in use, holes included
  const int n = (int)g->built_count;
  if (sorted_indices)
    NBH_HIP(hipMemcpy(sorted_indices, g->d_idx_b, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
  if (!host_nodes) return NBODY_HIP_OK;
  // the engine's ids have holes (padding of odd sibling groups, nodes cut off by the capacity): the reference's
  // array has none, so ids are compacted here and the child links rewritten
  std::vector<int> first_of(nids), compact(nids, -1);
  NBH_HIP(hipMemcpy(first_of.data(), g->t.first, (size_t)nids * sizeof(int), hipMemcpyDeviceToHost));
  int count = 0;
  for (int nid = 0; nid < nids; nid++)
    if (first_of[nid] >= 0) compact[nid] = count++;
  if (capacity_nodes < count)
    return NBH_FAIL(NBODY_HIP_ERR_VALIDATION, "node buffer too small: %d < %d", capacity_nodes, count);
  std::vector<NodeRec> rec(nids);
  std::vector<unsigned long long> keys(n);  // 30-bit keys widened: one code path below
  std::vector<int> idx(n);
  TreeRoot root;
  NBH_HIP(hipMemcpy(rec.data(), g->t.rec, (size_t)nids * sizeof(NodeRec), hipMemcpyDeviceToHost));
  const int axis_bits = g->wide() ? 21 : 10;
  if (g->wide()) {
    NBH_HIP(hipMemcpy(keys.data(), g->d_keys_b, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  } else {
    std::vector<unsigned int> k32(n);
    NBH_HIP(hipMemcpy(k32.data(), g->d_keys_b, (size_t)n * sizeof(unsigned int), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) keys[i] = k32[i];
  }
  NBH_HIP(hipMemcpy(idx.data(), g->d_idx_b, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
  NBH_HIP(hipMemcpy(&root, g->d_root, sizeof(root), hipMemcpyDeviceToHost));
  RefOctreeNode* out = static_cast<RefOctreeNode*>(host_nodes);
  int level = 0;
  for (int nid = 0; nid < nids; nid++) {
    while (level < g->max_depth && nid >= lb[level + 1]) level++;
    if (compact[nid] < 0) continue;
    RefOctreeNode& o = out[compact[nid]];
    const int first = rec[nid].first, cnt = rec[nid].count;
    const unsigned int ci = rec[nid].child;
    const unsigned long long k = keys[first];
    unsigned int q[3] = {0, 0, 0};
    for (int bit = 0; bit < axis_bits; bit++) {
      q[0] |= (unsigned)((k >> (3 * bit + 2)) & 1ull) << bit;
      q[1] |= (unsigned)((k >> (3 * bit + 1)) & 1ull) << bit;
      q[2] |= (unsigned)((k >> (3 * bit + 0)) & 1ull) << bit;
    }
    const float h = ldexpf(root.half, -level);
    for (int ax = 0; ax < 3; ax++)
      o.center[ax] = root.lo[ax] + ((float)(q[ax] >> (axis_bits - level)) + 0.5f) * (2.0f * h);
    o.half_size = h;
    o.com[0] = rec[nid].cx; o.com[1] = rec[nid].cy; o.com[2] = rec[nid].cz;
    o.total_mass = rec[nid].mass;
    for (int c = 0; c < 8; c++) o.children[c] = -1;
    o.is_leaf = ci == 0u;
    o.particle_index = (o.is_leaf && cnt >= 1) ? idx[first] : -1;
    o.particle_count = cnt;
    if (!o.is_leaf) {
      const int c0 = (int)(ci & 0x0fffffffu), nc = (int)(ci >> 28);
      const int shift = 3 * axis_bits - 3 - 3 * level;
      for (int c = c0; c < c0 + nc; c++) o.children[(keys[rec[c].first] >> shift) & 7ull] = compact[c];
    }
  }
  return NBODY_HIP_OK;
}
