// valu_rates.hip -- measures the FP32 VALU issue rates that bound the Direct N^2 kernel on
// gfx950: v_fma_f32, v_pk_fma_f32, v_rsq_f32 and the 12-VALU + 1-rsq pair body, at 1..8 waves
// per SIMD.  Output: one line per (kernel, occupancy): wave-instructions per cycle per SIMD
// (from s_memtime deltas) and chip-wide Ginstr/s (from hipEvent time).
// Build: make -C n-body_amd/csrc microbench ; run: n-body_amd/lib/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int CH = 8;  // independent chains per lane

__global__ __launch_bounds__(256) void k_fma(float* out, unsigned long long* cyc, float b, float c) {
  float a[CH];
  for (int k = 0; k < CH; k++) a[k] = threadIdx.x * 1e-3f + k;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < CH; k++) a[k] = __builtin_fmaf(a[k], b, c);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int k = 0; k < CH; k++) s += a[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_pkfma(float* out, unsigned long long* cyc, float b, float c) {
  f2 a[CH];
  for (int k = 0; k < CH; k++) a[k] = f2{threadIdx.x * 1e-3f + k, threadIdx.x * 2e-3f + k};
  const f2 bb = {b, b * 1.0001f}, cc = {c, c * 1.0001f};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < CH; k++) a[k] = __builtin_elementwise_fma(a[k], bb, cc);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int k = 0; k < CH; k++) s += a[k].x + a[k].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_rsq(float* out, unsigned long long* cyc, float b, float c) {
  float a[CH];
  for (int k = 0; k < CH; k++) a[k] = threadIdx.x * 1e-3f + k + 1.0f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < CH; k++) a[k] = __builtin_amdgcn_rsqf(a[k]);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int k = 0; k < CH; k++) s += a[k];
  out[blockIdx.x * 256 + threadIdx.x] = s + b + c;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// 1 rsq per 12 fma: the pair body's mix, all independent chains
__global__ __launch_bounds__(256) void k_mix(float* out, unsigned long long* cyc, float b, float c) {
  float a[12], r[2];
  for (int k = 0; k < 12; k++) a[k] = threadIdx.x * 1e-3f + k;
  r[0] = 1.5f + threadIdx.x; r[1] = 2.5f + threadIdx.x;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < 12; k++) a[k] = __builtin_fmaf(a[k], b, c);
    r[0] = __builtin_amdgcn_rsqf(r[0]);
#pragma unroll
    for (int k = 0; k < 12; k++) a[k] = __builtin_fmaf(a[k], b, c);
    r[1] = __builtin_amdgcn_rsqf(r[1]);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = r[0] + r[1]; for (int k = 0; k < 12; k++) s += a[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// same mix with packed fma: 6 pk_fma + 1 rsq
__global__ __launch_bounds__(256) void k_mixpk(float* out, unsigned long long* cyc, float b, float c) {
  f2 a[6]; float r[2];
  for (int k = 0; k < 6; k++) a[k] = f2{threadIdx.x * 1e-3f + k, threadIdx.x * 2e-3f + k};
  const f2 bb = {b, b * 1.0001f}, cc = {c, c * 1.0001f};
  r[0] = 1.5f + threadIdx.x; r[1] = 2.5f + threadIdx.x;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < 6; k++) a[k] = __builtin_elementwise_fma(a[k], bb, cc);
    r[0] = __builtin_amdgcn_rsqf(r[0]);
#pragma unroll
    for (int k = 0; k < 6; k++) a[k] = __builtin_elementwise_fma(a[k], bb, cc);
    r[1] = __builtin_amdgcn_rsqf(r[1]);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = r[0] + r[1]; for (int k = 0; k < 6; k++) s += a[k].x + a[k].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(float*, unsigned long long*, float, float);

static int run(const char* name, kern_t k, int waves_per_simd, double instr_per_iter, float* out,
               unsigned long long* cyc, std::vector<unsigned long long>& h) {
  const int blocks = 256 * waves_per_simd;  // 256 CUs, one 4-wave block = 1 wave per SIMD
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, cyc, 0.999f, 0.001f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  const int reps = 5;
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, cyc, 0.999f, 0.001f);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  CHECK(hipMemcpy(h.data(), cyc, blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.begin() + blocks * 4);
  const double med = (double)h[blocks * 2];
  const double total_instr = instr_per_iter * ITERS;                   // per wave
  const double per_simd_per_cycle = total_instr * waves_per_simd / med; // wave-instr / cycle / SIMD
  const double ginstr = total_instr * blocks * 4 / (ms * 1e-3) / 1e9;
  printf("%-8s waves/SIMD=%d  median cycles/wave=%.0f  wave-instr/clk/SIMD=%.3f  chip Gwave-instr/s=%.1f  ms=%.3f  (clk est %.2f GHz)\n",
         name, waves_per_simd, med, per_simd_per_cycle, ginstr, ms, med / (ms * 1e-3) / 1e9);
  return 0;
}

int main() {
  float* out; unsigned long long* cyc;
  CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
  CHECK(hipMalloc(&cyc, 256 * 8 * 4 * sizeof(unsigned long long)));
  std::vector<unsigned long long> h(256 * 8 * 4);
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  CUs=%d  clock=%d kHz  arch=%s\n", p.name, p.multiProcessorCount, p.clockRate, p.gcnArchName);
  for (int w : {1, 2, 4, 8}) {
    run("fma", k_fma, w, CH, out, cyc, h);
    run("pk_fma", k_pkfma, w, CH, out, cyc, h);
    run("rsq", k_rsq, w, CH, out, cyc, h);
    run("mix12+1", k_mix, w, 26, out, cyc, h);
    run("mixpk6+1", k_mixpk, w, 14, out, cyc, h);
  }
  return 0;
}
