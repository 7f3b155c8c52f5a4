// radix_test.hip -- the hand-written radix sort of csrc/radix_sort.h against std::stable_sort, both instantiations the
// library uses (32-bit keys + float4 body + index; 64-bit keys + index), sizes around the tile boundaries, bit ranges,
// equal keys (stability), and its time per sort.   Build: make -C n-body_amd/csrc microbench ; run: n-body_amd/lib/radix_test
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <vector>

#include "../radix_sort.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static unsigned long long rng_state = 88172645463325252ull;
static unsigned long long rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

template <class Key, bool BODY>
static int run_case(size_t n, unsigned b0, unsigned b1, int pattern, unsigned* err_dev, volatile unsigned* err_host, bool timing) {
  using namespace nbh::radix;
  std::vector<Key> hk(n);
  std::vector<float4> hb(BODY ? n : 0);
  for (size_t i = 0; i < n; i++) {
    Key k;
    if (pattern == 0) k = (Key)rnd();
    else if (pattern == 1) k = (Key)(rnd() & 0xff) << b0;           // many equal keys
    else if (pattern == 2) k = (Key)i << b0;                          // sorted
    else if (pattern == 3) k = (Key)(n - 1 - i) << b0;                // reversed
    else k = (Key)0x5555aaaa;                                         // all equal
    hk[i] = k;
    if (BODY) hb[i] = make_float4((float)i, (float)(k & 0xffff), 1.f, 2.f);
  }
  Key *k_in, *k_out;
  float4 *b_in = nullptr, *b_out = nullptr;
  int* i_out;
  void* tmp;
  size_t tb = 0;
  CHECK((sort_pairs<Key, BODY>(nullptr, tb, (const Key*)nullptr, (Key*)nullptr, nullptr, nullptr, nullptr, nullptr, n, b0, b1, 0, err_dev)));
  CHECK(hipMalloc(&k_in, (n + 1) * sizeof(Key)));
  CHECK(hipMalloc(&k_out, (n + 1) * sizeof(Key)));
  CHECK(hipMalloc(&i_out, (n + 1) * sizeof(int)));
  if (BODY) { CHECK(hipMalloc(&b_in, (n + 1) * sizeof(float4))); CHECK(hipMalloc(&b_out, (n + 1) * sizeof(float4))); }
  CHECK(hipMalloc(&tmp, tb));
  CHECK(hipMemcpy(k_in, hk.data(), n * sizeof(Key), hipMemcpyHostToDevice));
  if (BODY) CHECK(hipMemcpy(b_in, hb.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  *err_host = 0;
  size_t t2 = tb;
  CHECK((sort_pairs<Key, BODY>(tmp, t2, k_in, k_out, b_in, b_out, nullptr, i_out, n, b0, b1, 0, err_dev)));
  CHECK(hipDeviceSynchronize());
  if (*err_host) { printf("  look-back gave up (error word set)\n"); return 1; }
  std::vector<Key> rk(n);
  std::vector<int> ri(n);
  std::vector<float4> rb(BODY ? n : 0);
  CHECK(hipMemcpy(rk.data(), k_out, n * sizeof(Key), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(ri.data(), i_out, n * sizeof(int), hipMemcpyDeviceToHost));
  if (BODY) CHECK(hipMemcpy(rb.data(), b_out, n * sizeof(float4), hipMemcpyDeviceToHost));
  std::vector<int> perm(n);
  std::iota(perm.begin(), perm.end(), 0);
  const Key mask = b1 - b0 >= sizeof(Key) * 8 ? ~(Key)0 : ((((Key)1) << (b1 - b0)) - 1);
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int c) { return ((hk[a] >> b0) & mask) < ((hk[c] >> b0) & mask); });
  size_t bad = 0;
  for (size_t i = 0; i < n; i++) {
    if (ri[i] != perm[i] || rk[i] != hk[perm[i]]) bad++;
    if (BODY && (rb[i].x != hb[perm[i]].x || rb[i].y != hb[perm[i]].y)) bad++;
  }
  float ms = 0.f;
  if (timing && !bad) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++) { t2 = tb; (void)sort_pairs<Key, BODY>(tmp, t2, k_in, k_out, b_in, b_out, nullptr, i_out, n, b0, b1, 0, err_dev); }
    CHECK(hipEventRecord(e0, 0));
    for (int w = 0; w < 10; w++) { t2 = tb; (void)sort_pairs<Key, BODY>(tmp, t2, k_in, k_out, b_in, b_out, nullptr, i_out, n, b0, b1, 0, err_dev); }
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 10;
    // the same with the digit counts brought by the caller (as the key kernels of the builds do): the passes alone
    {
      const unsigned places = (b1 - b0 + 9) / 10;
      unsigned* counts;
      CHECK(hipMalloc(&counts, places * 1024 * 4));
      CHECK(hipMemset(counts, 0, places * 1024 * 4));
      hipLaunchKernelGGL((hist_kernel<Key>), dim3(1024), dim3(256), 0, 0, k_in, (unsigned)n, b0, b1, places, counts);
      CHECK(hipDeviceSynchronize());
      float ms2 = 0.f;
      CHECK(hipEventRecord(e0, 0));
      for (int w = 0; w < 10; w++) { t2 = tb; (void)sort_pairs<Key, BODY>(tmp, t2, k_in, k_out, b_in, b_out, nullptr, i_out, n, b0, b1, 0, err_dev, counts, 1, 0); }
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms2, e0, e1));
      printf("  [passes alone, counts brought: %.3f ms] ", ms2 / 10);
      (void)hipFree(counts);
    }
  }
  printf("  %s keys%s n=%zu bits [%u,%u) pattern %d: %s", sizeof(Key) == 8 ? "64-bit" : "32-bit", BODY ? " + body" : "", n, b0, b1, pattern,
         bad ? "MISMATCH" : "ok");
  if (bad) printf(" (%zu elements)", bad);
  if (timing && !bad) printf("  %.3f ms per sort (%u passes)", ms, (b1 - b0 + 9) / 10);
  printf("\n");
  (void)hipFree(k_in); (void)hipFree(k_out); (void)hipFree(i_out); (void)hipFree(tmp);
  if (BODY) { (void)hipFree(b_in); (void)hipFree(b_out); }
  return bad ? 1 : 0;
}

int main() {
  unsigned* err_host = nullptr;
  unsigned* err_dev = nullptr;
  CHECK(hipHostMalloc(reinterpret_cast<void**>(&err_host), 64, hipHostMallocMapped));
  CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&err_dev), err_host, 0));
  int fails = 0;
  const size_t sizes[] = {1, 63, 64, 65, 4095, 4096, 4097, 8192, 100000, 262144, 1048576 + 5};
  for (size_t n : sizes) {
    for (int pattern = 0; pattern < 5; pattern++) {
      fails += run_case<unsigned int, true>(n, 0, 20, pattern, err_dev, err_host, false);
      fails += run_case<unsigned long long, false>(n, 3, 63, pattern, err_dev, err_host, false);
    }
    fails += run_case<unsigned int, true>(n, 0, 27, 0, err_dev, err_host, false);
    fails += run_case<unsigned int, true>(n, 5, 11, 0, err_dev, err_host, false);
    fails += run_case<unsigned long long, false>(n, 0, 30, 0, err_dev, err_host, false);
  }
  printf("timing:\n");
  fails += run_case<unsigned int, true>(4194304, 0, 19, 0, err_dev, err_host, true);
  fails += run_case<unsigned int, true>(4194304, 0, 20, 1, err_dev, err_host, true);
  fails += run_case<unsigned int, true>(524288, 0, 19, 0, err_dev, err_host, true);
  fails += run_case<unsigned long long, false>(1048576, 3, 63, 0, err_dev, err_host, true);
  fails += run_case<unsigned long long, false>(4194304, 3, 63, 0, err_dev, err_host, true);
  printf("%s (%d failing cases)\n", fails ? "FAILED" : "all cases ok", fails);
  return fails ? 1 : 0;
}
