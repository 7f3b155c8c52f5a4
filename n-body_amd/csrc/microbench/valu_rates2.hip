// valu_rates2.hip -- issue rates of the instructions the two-phase spatial-hash kernel is built from (gfx950):
// v_dot2c_f32_f16, v_fma_mix_f32 (f16 operands), v_cmp + v_addc_co (mask shift-in), v_ffbh_u32, v_bfi_b32, and the
// phase-1 bodies (two dot2c / three fma_mix, + cmp + addc).  Output as valu_rates.hip: chip-wide Gwave-instr/s and
// cycles per wave-instruction per SIMD at 1..8 waves per SIMD.
// Build: make -C n-body_amd/csrc microbench ; run: n-body_amd/lib/valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int CH = 8;

#define PROLOGUE                                                                         \
  float a[CH];                                                                           \
  for (int k = 0; k < CH; k++) a[k] = threadIdx.x * 1e-3f + k;                           \
  const h2 hb = {(_Float16)b, (_Float16)(b * 0.5f)}, hc = {(_Float16)c, (_Float16)1.0f}; \
  unsigned long long t0 = __builtin_readcyclecounter();
#define EPILOGUE                                                                         \
  unsigned long long t1 = __builtin_readcyclecounter();                                  \
  float s = 0; for (int k = 0; k < CH; k++) s += a[k];                                   \
  out[blockIdx.x * 256 + threadIdx.x] = s;                                               \
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;

__global__ __launch_bounds__(256) void k_dot2c(float* out, unsigned long long* cyc, float b, float c) {
  PROLOGUE
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < CH; k++) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[k]) : "v"(hb), "v"(hc));
  }
  EPILOGUE
}
__global__ __launch_bounds__(256) void k_fmamix(float* out, unsigned long long* cyc, float b, float c) {
  PROLOGUE
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < CH; k++)
      asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(a[k]) : "v"(hb), "v"(hc));
  }
  EPILOGUE
}
__global__ __launch_bounds__(256) void k_cmpaddc(float* out, unsigned long long* cyc, float b, float c) {
  PROLOGUE
  unsigned m[CH];
  for (int k = 0; k < CH; k++) m[k] = threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < CH; k++)
      asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m[k]) : "v"(a[k]), "v"(c) : "vcc");
  }
  for (int k = 0; k < CH; k++) a[k] += (float)m[k];
  EPILOGUE
}
__global__ __launch_bounds__(256) void k_ffbh_bfi(float* out, unsigned long long* cyc, float b, float c) {
  PROLOGUE
  unsigned m[CH];
  for (int k = 0; k < CH; k++) m[k] = 0xffffffffu - threadIdx.x;
  const unsigned top = 0x80000000u + (unsigned)(b > 10.f);
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < CH; k++) {
      unsigned lz, t;
      asm volatile("v_ffbh_u32 %0, %2\n\tv_lshrrev_b32 %1, %0, %3\n\tv_bfi_b32 %2, %1, 0, %2" : "=&v"(lz), "=&v"(t), "+v"(m[k]) : "v"(top));
    }
  }
  for (int k = 0; k < CH; k++) a[k] += (float)m[k];
  EPILOGUE
}
// phase-1 body, dot2 form: per candidate and target  2 dot2c (+ the move that seeds the accumulator) + cmp + addc
__global__ __launch_bounds__(256) void k_p1dot(float* out, unsigned long long* cyc, float b, float c) {
  PROLOGUE
  unsigned m[CH];
  for (int k = 0; k < CH; k++) m[k] = threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < CH; k++) {
      float d;
      asm volatile("v_mov_b32 %0, %2\n\tv_dot2c_f32_f16 %0, %3, %4\n\tv_dot2c_f32_f16 %0, %4, %3\n\t"
                   "v_cmp_lt_f32 vcc, %0, %5\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc"
                   : "=&v"(d), "+v"(m[k]) : "v"(a[k]), "v"(hb), "v"(hc), "v"(c) : "vcc");
    }
  }
  for (int k = 0; k < CH; k++) a[k] += (float)m[k];
  EPILOGUE
}
// phase-1 body, fma_mix form: 3 fma_mix + cmp + addc
__global__ __launch_bounds__(256) void k_p1mix(float* out, unsigned long long* cyc, float b, float c) {
  PROLOGUE
  unsigned m[CH];
  for (int k = 0; k < CH; k++) m[k] = threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < CH; k++) {
      float d;
      asm volatile("v_fma_mix_f32 %0, %3, %2, %2 op_sel_hi:[1,0,0]\n\t"
                   "v_fma_mix_f32 %0, %3, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                   "v_fma_mix_f32 %0, %4, %2, %0 op_sel_hi:[1,0,0]\n\t"
                   "v_cmp_lt_f32 vcc, %0, %5\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc"
                   : "=&v"(d), "+v"(m[k]) : "v"(a[k]), "v"(hb), "v"(hc), "v"(c) : "vcc");
    }
  }
  for (int k = 0; k < CH; k++) a[k] += (float)m[k];
  EPILOGUE
}
// phase-1 body, exact fp32 form: 3 sub + mul + 2 fma + cmp + addc
__global__ __launch_bounds__(256) void k_p1exact(float* out, unsigned long long* cyc, float b, float c) {
  PROLOGUE
  unsigned m[CH];
  for (int k = 0; k < CH; k++) m[k] = threadIdx.x;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < CH; k++) {
      float dx, dy, dz;
      asm volatile("v_sub_f32 %0, %4, %5\n\tv_sub_f32 %1, %4, %6\n\tv_sub_f32 %2, %5, %6\n\t"
                   "v_mul_f32 %0, %0, %0\n\tv_fma_f32 %0, %1, %1, %0\n\tv_fma_f32 %0, %2, %2, %0\n\t"
                   "v_cmp_lt_f32 vcc, %0, %6\n\tv_addc_co_u32 %3, vcc, %3, %3, vcc"
                   : "=&v"(dx), "=&v"(dy), "=&v"(dz), "+v"(m[k]) : "v"(a[k]), "v"(b), "v"(c) : "vcc");
    }
  }
  for (int k = 0; k < CH; k++) a[k] += (float)m[k];
  EPILOGUE
}


__global__ __launch_bounds__(256) void k_pkfmaf16(float* out, unsigned long long* cyc, float b, float c) {
  PROLOGUE
  h2 v[CH];
  for (int k = 0; k < CH; k++) v[k] = h2{(_Float16)(threadIdx.x * 1e-3f), (_Float16)k};
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < CH; k++) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(v[k]) : "v"(hb), "v"(hc));
  }
  for (int k = 0; k < CH; k++) a[k] += (float)v[k].x + (float)v[k].y;
  EPILOGUE
}
__global__ __launch_bounds__(256) void k_alignbit(float* out, unsigned long long* cyc, float b, float c) {
  PROLOGUE
  unsigned m[CH];
  for (int k = 0; k < CH; k++) m[k] = threadIdx.x + k;
  const unsigned u = __builtin_bit_cast(unsigned, b);
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < CH; k++) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(m[k]) : "v"(u));
  }
  for (int k = 0; k < CH; k++) a[k] += (float)m[k];
  EPILOGUE
}
__global__ __launch_bounds__(256) void k_dot4c(float* out, unsigned long long* cyc, float b, float c) {
  PROLOGUE
  int m[CH];
  for (int k = 0; k < CH; k++) m[k] = threadIdx.x + k;
  const int u = __builtin_bit_cast(int, b), w = __builtin_bit_cast(int, c);
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < CH; k++) asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(m[k]) : "v"(u), "v"(w));
  }
  for (int k = 0; k < CH; k++) a[k] += (float)m[k];
  EPILOGUE
}
// phase-1 body, int8 form: move (seed = |s|^2) + dot4c + cmp + addc per candidate and target
__global__ __launch_bounds__(256) void k_p1dot4(float* out, unsigned long long* cyc, float b, float c) {
  PROLOGUE
  unsigned m[CH];
  for (int k = 0; k < CH; k++) m[k] = threadIdx.x;
  const int u = __builtin_bit_cast(int, b), w = __builtin_bit_cast(int, c);
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < CH; k++) {
      int d;
      asm volatile("v_mov_b32 %0, %2\n\tv_dot4c_i32_i8 %0, %3, %4\n\t"
                   "v_cmp_lt_i32 vcc, %0, %4\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc"
                   : "=&v"(d), "+v"(m[k]) : "v"(a[k]), "v"(u), "v"(w) : "vcc");
    }
  }
  for (int k = 0; k < CH; k++) a[k] += (float)m[k];
  EPILOGUE
}
// phase-1 body, packed-f16 form: per candidate and TWO targets  pk_add + 3 pk_fma + alignbit + lshl + alignbit
__global__ __launch_bounds__(256) void k_p1pkh(float* out, unsigned long long* cyc, float b, float c) {
  PROLOGUE
  unsigned m[CH];
  for (int k = 0; k < CH; k++) m[k] = threadIdx.x;
  for (int it = 0; it < ITERS; it += 2) {  // (CH chains x 2 targets per body: half the iterations for the same target count)
#pragma unroll
    for (int k = 0; k < CH; k += 2) {
      unsigned r, r2;
      asm volatile("v_pk_add_f16 %0, %4, %5\n\tv_pk_fma_f16 %0, %4, %5, %0\n\tv_pk_fma_f16 %0, %5, %4, %0\n\t"
                   "v_pk_fma_f16 %0, %4, %4, %0\n\tv_alignbit_b32 %2, %2, %0, 31\n\tv_lshlrev_b32 %1, 16, %0\n\t"
                   "v_alignbit_b32 %3, %3, %1, 31"
                   : "=&v"(r), "=&v"(r2), "+v"(m[k]), "+v"(m[k + 1]) : "v"(hb), "v"(hc));
      asm volatile("v_pk_add_f16 %0, %4, %5\n\tv_pk_fma_f16 %0, %4, %5, %0\n\tv_pk_fma_f16 %0, %5, %4, %0\n\t"
                   "v_pk_fma_f16 %0, %4, %4, %0\n\tv_alignbit_b32 %2, %2, %0, 31\n\tv_lshlrev_b32 %1, 16, %0\n\t"
                   "v_alignbit_b32 %3, %3, %1, 31"
                   : "=&v"(r), "=&v"(r2), "+v"(m[k]), "+v"(m[k + 1]) : "v"(hc), "v"(hb));
    }
  }
  for (int k = 0; k < CH; k++) a[k] += (float)m[k];
  EPILOGUE
}
// phase-2 body: ffbh + mad (address) + lshr + bfi + 3 sub + mul + 2 fma + add + rsq + 3 mul + cmp + cndmask + 3 fma
__global__ __launch_bounds__(256) void k_p2(float* out, unsigned long long* cyc, float b, float c) {
  PROLOGUE
  unsigned m[CH];
  for (int k = 0; k < CH; k++) m[k] = 0xffffffffu - threadIdx.x;
  const unsigned top = 0x80000000u + (unsigned)(b > 10.f);
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < CH; k++) {
      unsigned lz, t, ad;
      float dx, dy, dz, d2, inv, f;
      asm volatile("v_ffbh_u32 %0, %3\n\tv_mad_u32_u24 %2, %0, %4, %4\n\tv_lshrrev_b32 %1, %0, %4\n\tv_bfi_b32 %3, %1, 0, %3"
                   : "=&v"(lz), "=&v"(t), "=&v"(ad), "+v"(m[k]) : "v"(top));
      asm volatile("v_sub_f32 %0, %9, %10\n\tv_sub_f32 %1, %9, %11\n\tv_sub_f32 %2, %10, %11\n\t"
                   "v_mul_f32 %3, %0, %0\n\tv_fma_f32 %3, %1, %1, %3\n\tv_fma_f32 %3, %2, %2, %3\n\t"
                   "v_add_f32 %4, %3, %11\n\tv_rsq_f32 %4, %4\n\tv_mul_f32 %5, %4, %10\n\tv_mul_f32 %4, %4, %4\n\t"
                   "v_mul_f32 %5, %5, %4\n\tv_cmp_lt_f32 vcc, %3, %11\n\tv_cndmask_b32 %5, 0, %5, vcc\n\t"
                   "v_fma_f32 %6, %5, %0, %6\n\tv_fma_f32 %7, %5, %1, %7\n\tv_fma_f32 %8, %5, %2, %8"
                   : "=&v"(dx), "=&v"(dy), "=&v"(dz), "=&v"(d2), "=&v"(inv), "=&v"(f), "+v"(ax), "+v"(ay), "+v"(az)
                   : "v"(a[k]), "v"(b), "v"(c) : "vcc");
    }
  }
  a[0] += ax + ay + az;
  for (int k = 0; k < CH; k++) a[k] += (float)m[k];
  EPILOGUE
}

typedef void (*kern_t)(float*, unsigned long long*, float, float);

static int run(const char* name, kern_t k, int waves_per_simd, double instr_per_iter, float* out,
               unsigned long long* cyc, std::vector<unsigned long long>& h) {
  const int blocks = 256 * waves_per_simd;
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
  const double total_instr = instr_per_iter * ITERS;
  const double ginstr = total_instr * blocks * 4 / (ms * 1e-3) / 1e9;
  printf("%-9s waves/SIMD=%d  instr/iter=%2.0f  cycles/wave-instr/SIMD=%.2f  chip Gwave-instr/s=%.1f  ms=%.3f  ns per body per SIMD=%.2f\n",
         name, waves_per_simd, instr_per_iter / CH, med / (total_instr * waves_per_simd), ginstr, ms,
         ms * 1e6 / ((double)ITERS * CH * waves_per_simd));
  return 0;
}

int main() {
  float* out; unsigned long long* cyc;
  CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
  CHECK(hipMalloc(&cyc, 256 * 8 * 4 * sizeof(unsigned long long)));
  std::vector<unsigned long long> h(256 * 8 * 4);
  for (int w : {1, 2, 3, 4, 8}) {
    run("dot2c", k_dot2c, w, CH, out, cyc, h);
    run("fma_mix", k_fmamix, w, CH, out, cyc, h);
    run("cmp+addc", k_cmpaddc, w, 2 * CH, out, cyc, h);
    run("ffbh3", k_ffbh_bfi, w, 3 * CH, out, cyc, h);
    run("p1dot", k_p1dot, w, 5 * CH, out, cyc, h);
    run("p1mix", k_p1mix, w, 5 * CH, out, cyc, h);
    run("p1exact", k_p1exact, w, 8 * CH, out, cyc, h);
    run("pk_fma_h", k_pkfmaf16, w, CH, out, cyc, h);
    run("alignbit", k_alignbit, w, CH, out, cyc, h);
    run("dot4c", k_dot4c, w, CH, out, cyc, h);
    run("p1dot4", k_p1dot4, w, 4 * CH, out, cyc, h);
    run("p1pkh", k_p1pkh, w, 3.5 * CH, out, cyc, h);
    run("p2", k_p2, w, 20 * CH, out, cyc, h);
  }
  return 0;
}
