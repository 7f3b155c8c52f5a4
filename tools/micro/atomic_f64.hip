// Are global f64 / f32 / u64 atomic adds lossless across all XCDs on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_f64(double* a, int n) { for (int i = threadIdx.x; i < n; i += blockDim.x) unsafeAtomicAdd(&a[i], 1.0); }
__global__ void k_f64safe(double* a, int n) { for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&a[i], 1.0); }
__global__ void k_f32(float* a, int n) { for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&a[i], 1.0f); }
__global__ void k_u64(unsigned long long* a, int n) { for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&a[i], 1ull); }
template <class T, class K> int run(const char* name, K kern, int blocks, int n) {
  T* d; CHECK(hipMalloc(&d, n * sizeof(T))); CHECK(hipMemset(d, 0, n * sizeof(T)));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, n); hipEventRecord(e1);
  CHECK(hipDeviceSynchronize()); float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<T> h(n); CHECK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost));
  long long bad = 0; double mn = 1e300; for (int i = 0; i < n; i++) { if ((double)h[i] != (double)blocks) bad++; if ((double)h[i] < mn) mn = (double)h[i]; }
  printf("%-10s blocks=%d n=%d: wrong=%lld min=%.0f expected=%d  %.3f ms  %.1f G adds/s\n", name, blocks, n, bad, mn, blocks, ms, (double)blocks * n / ms / 1e6);
  hipFree(d); return 0; }
int main() {
  for (int n : {64, 4096, 1 << 20}) {
    int blocks = n >= (1 << 20) ? 512 : 8192;
    run<double>("f64unsafe", k_f64, blocks, n);
    run<double>("f64safe", k_f64safe, blocks, n);
    run<float>("f32", k_f32, blocks, n);
    run<unsigned long long>("u64", k_u64, blocks, n);
  }
  return 0;
}
