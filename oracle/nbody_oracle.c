/*
 * nbody_oracle.c -- CPU restatement of the LessUp/n-body hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (n-body_amd/, include/,
 * bench.py's GPU leg) may import, link or call this file.  Allowed callers:
 * tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 *
 * The reference has no CPU force path (every force/integrator symbol is CUDA,
 * CMakeLists.txt:149-158), so this is a restatement of the reference arithmetic
 * in plain C, each function citing the reference file:line it follows
 * (paths relative to the reference checkout).  Pinning: tests/test_oracle_pins.py
 * checks it against every known-answer value the reference's own tests hold for
 * this path (tests/test_force_calculation.cpp:13-60, tests/test_integrator.cpp:15-162,
 * examples/example_energy_conservation.cpp:26-147, tests/test_spatial_hash.cpp:38-51).
 *
 * Arithmetic modes (argument `mode` of the force routines):
 *   0  FAITHFUL  fp32 everywhere, one sequential fp32 accumulator per target, in
 *                source-index order, skip j==i  (force_direct.cu:58-75,
 *                examples/example_force_methods.cpp:44-66)
 *   1  F32TERMS  the same fp32 pair terms, accumulated in fp64 (removes the
 *                summation-order noise; this is the parity oracle for the GPU)
 *   2  GOLD      fp64 arithmetic on the fp32 inputs
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp; -ffp-contract=off so that fp32
 * products/sums round exactly as the C source says).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

ORACLE_API int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

ORACLE_API void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------------- */
/* a5: one-pair acceleration, force_direct.cu:109-117                         */
/*     r = p2 - p1; d2 = |r|^2 + eps*eps; inv = 1/sqrtf(d2); f = G*m2*inv^3   */
/* ------------------------------------------------------------------------- */
ORACLE_API void oracle_pair_force(const float p1[3], const float p2[3], float m1, float m2,
                                  float G, float eps, float out[3]) {
  (void)m1; /* unused in the reference too */
  float rx = p2[0] - p1[0], ry = p2[1] - p1[1], rz = p2[2] - p1[2];
  float dist2 = (rx * rx + ry * ry + rz * rz) + eps * eps;
  float inv_dist = 1.0f / sqrtf(dist2);
  float inv_dist3 = inv_dist * inv_dist * inv_dist;
  float f = G * m2 * inv_dist3;
  out[0] = rx * f;
  out[1] = ry * f;
  out[2] = rz * f;
}

/* ------------------------------------------------------------------------- */
/* a4: all-pairs accelerations, force_direct.cu:58-75 (operation order) and   */
/*     example_force_methods.cpp:44-66 (the CPU loop).                         */
/*     Targets are [t0, t1) of the n bodies, or the list `tidx` when non-NULL  */
/*     (n_t entries); sources are all n bodies; the self pair is skipped by    */
/*     index exactly as the reference does (force_direct.cu:60).               */
/*     eps2 is passed already squared, as the kernel receives it (:15).        */
/* ------------------------------------------------------------------------- */
static void direct_one_target(size_t i, size_t n, const float* x, const float* y, const float* z,
                              const float* m, float G, float eps2, int mode, float out[3]) {
  if (mode == 0) {
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    const float xi = x[i], yi = y[i], zi = z[i];
    for (size_t j = 0; j < n; j++) {
      if (j == i) continue;
      float dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi;
      float dist2 = dx * dx + dy * dy + dz * dz + eps2;
      float inv_dist = 1.0f / sqrtf(dist2); /* rsqrtf on device */
      float inv_dist3 = inv_dist * inv_dist * inv_dist;
      float f = G * m[j] * inv_dist3;
      ax += f * dx;
      ay += f * dy;
      az += f * dz;
    }
    out[0] = ax; out[1] = ay; out[2] = az;
  } else if (mode == 1) {
    double ax = 0.0, ay = 0.0, az = 0.0;
    const float xi = x[i], yi = y[i], zi = z[i];
    for (size_t j = 0; j < n; j++) {
      if (j == i) continue;
      float dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi;
      float dist2 = dx * dx + dy * dy + dz * dz + eps2;
      float inv_dist = 1.0f / sqrtf(dist2);
      float inv_dist3 = inv_dist * inv_dist * inv_dist;
      float f = G * m[j] * inv_dist3;
      ax += (double)(f * dx);
      ay += (double)(f * dy);
      az += (double)(f * dz);
    }
    out[0] = (float)ax; out[1] = (float)ay; out[2] = (float)az;
  } else {
    double ax = 0.0, ay = 0.0, az = 0.0;
    const double xi = x[i], yi = y[i], zi = z[i];
    for (size_t j = 0; j < n; j++) {
      if (j == i) continue;
      double dx = (double)x[j] - xi, dy = (double)y[j] - yi, dz = (double)z[j] - zi;
      double dist2 = dx * dx + dy * dy + dz * dz + (double)eps2;
      double inv_dist = 1.0 / sqrt(dist2);
      double f = (double)G * (double)m[j] * inv_dist * inv_dist * inv_dist;
      ax += f * dx;
      ay += f * dy;
      az += f * dz;
    }
    out[0] = (float)ax; out[1] = (float)ay; out[2] = (float)az;
  }
}

ORACLE_API void oracle_direct_forces(size_t n, const float* x, const float* y, const float* z,
                                     const float* m, size_t t0, size_t t1, float* ax, float* ay,
                                     float* az, float G, float eps2, int mode) {
#pragma omp parallel for schedule(static)
  for (long long ii = (long long)t0; ii < (long long)t1; ii++) {
    float o[3];
    direct_one_target((size_t)ii, n, x, y, z, m, G, eps2, mode, o);
    ax[ii - t0] = o[0];
    ay[ii - t0] = o[1];
    az[ii - t0] = o[2];
  }
}

ORACLE_API void oracle_direct_forces_indexed(size_t n, const float* x, const float* y,
                                             const float* z, const float* m, size_t n_t,
                                             const int64_t* tidx, float* ax, float* ay, float* az,
                                             float G, float eps2, int mode) {
#pragma omp parallel for schedule(static)
  for (long long k = 0; k < (long long)n_t; k++) {
    float o[3];
    direct_one_target((size_t)tidx[k], n, x, y, z, m, G, eps2, mode, o);
    ax[k] = o[0];
    ay[k] = o[1];
    az[k] = o[2];
  }
}

/* Accelerations on arbitrary target POINTS (not members of the source set; no
 * self-skip).  Used to check the sharded kernel's target/source split. */
ORACLE_API void oracle_direct_forces_points(size_t n_s, const float* sx, const float* sy,
                                            const float* sz, const float* sm, size_t n_t,
                                            const float* tx, const float* ty, const float* tz,
                                            float* ax, float* ay, float* az, float G, float eps2) {
#pragma omp parallel for schedule(static)
  for (long long k = 0; k < (long long)n_t; k++) {
    double a0 = 0, a1 = 0, a2 = 0;
    for (size_t j = 0; j < n_s; j++) {
      float dx = sx[j] - tx[k], dy = sy[j] - ty[k], dz = sz[j] - tz[k];
      float d2 = dx * dx + dy * dy + dz * dz;
      if (d2 == 0.0f) continue; /* coincident point == the body itself */
      float dist2 = d2 + eps2;
      float inv = 1.0f / sqrtf(dist2);
      float f = G * sm[j] * (inv * inv * inv);
      a0 += (double)(f * dx);
      a1 += (double)(f * dy);
      a2 += (double)(f * dz);
    }
    ax[k] = (float)a0; ay[k] = (float)a1; az[k] = (float)a2;
  }
}

/* ------------------------------------------------------------------------- */
/* a6: Velocity Verlet pieces, integrator.cu:11-48                            */
/* ------------------------------------------------------------------------- */
ORACLE_API void oracle_update_positions(size_t n, float* px, float* py, float* pz,
                                        const float* vx, const float* vy, const float* vz,
                                        const float* ax, const float* ay, const float* az,
                                        float dt) {
  /* integrator.cu:16-19 */
  for (size_t i = 0; i < n; i++) {
    float dt2_half = 0.5f * dt * dt;
    px[i] += vx[i] * dt + ax[i] * dt2_half;
    py[i] += vy[i] * dt + ay[i] * dt2_half;
    pz[i] += vz[i] * dt + az[i] * dt2_half;
  }
}

ORACLE_API void oracle_update_velocities(size_t n, float* vx, float* vy, float* vz,
                                         const float* aox, const float* aoy, const float* aoz,
                                         const float* anx, const float* any_, const float* anz,
                                         float dt) {
  /* integrator.cu:31-34 */
  for (size_t i = 0; i < n; i++) {
    float dt_half = 0.5f * dt;
    vx[i] += (aox[i] + anx[i]) * dt_half;
    vy[i] += (aoy[i] + any_[i]) * dt_half;
    vz[i] += (aoz[i] + anz[i]) * dt_half;
  }
}

/* One full step of Integrator::integrate with the Direct calculator,
 * integrator.cu:224-238: a_old <- a; x += v dt + a dt^2/2; a <- F(x); v += (a_old+a) dt/2.
 * `mode` selects the force arithmetic (see header). */
ORACLE_API void oracle_integrate_direct(size_t n, float* px, float* py, float* pz, float* vx,
                                        float* vy, float* vz, float* ax, float* ay, float* az,
                                        float* aox, float* aoy, float* aoz, const float* m,
                                        float G, float eps, float dt, int steps, int mode) {
  for (int s = 0; s < steps; s++) {
    memcpy(aox, ax, n * sizeof(float)); /* integrator.cu:44-46 */
    memcpy(aoy, ay, n * sizeof(float));
    memcpy(aoz, az, n * sizeof(float));
    oracle_update_positions(n, px, py, pz, vx, vy, vz, ax, ay, az, dt);
    oracle_direct_forces(n, px, py, pz, m, 0, n, ax, ay, az, G, eps * eps, mode);
    oracle_update_velocities(n, vx, vy, vz, aox, aoy, aoz, ax, ay, az, dt);
  }
}

/* ------------------------------------------------------------------------- */
/* a7: energies, integrator.cu:51-119 + host partial sums :252-289            */
/*     mode 0: fp32 with the reference's structure (per-thread fp32 value,     */
/*             per-block tree reduction of `block` values, fp32 host sum of    */
/*             the block partials); mode 2: fp64.                              */
/* ------------------------------------------------------------------------- */
static float block_tree_sum(float* v, int block) {
  /* integrator.cu:69-74: for (s = block/2; s > 0; s >>= 1) v[t] += v[t+s] */
  for (int s = block / 2; s > 0; s >>= 1)
    for (int t = 0; t < s; t++) v[t] += v[t + s];
  return v[0];
}

ORACLE_API double oracle_kinetic_energy(size_t n, const float* vx, const float* vy,
                                        const float* vz, const float* m, int block, int mode) {
  if (mode == 2) {
    double ke = 0.0;
    for (size_t i = 0; i < n; i++) {
      double v2 = (double)vx[i] * vx[i] + (double)vy[i] * vy[i] + (double)vz[i] * vz[i];
      ke += 0.5 * (double)m[i] * v2;
    }
    return ke;
  }
  float* buf = (float*)malloc((size_t)block * sizeof(float));
  float total = 0.0f;
  for (size_t b0 = 0; b0 < n; b0 += (size_t)block) {
    for (int t = 0; t < block; t++) {
      size_t i = b0 + (size_t)t;
      float ke = 0.0f;
      if (i < n) {
        float v2 = vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i]; /* :61 */
        ke = 0.5f * m[i] * v2;                                      /* :62 */
      }
      buf[t] = ke;
    }
    total += block_tree_sum(buf, block); /* :264-266 host sum */
  }
  free(buf);
  return (double)total;
}

ORACLE_API double oracle_potential_energy(size_t n, const float* x, const float* y,
                                          const float* z, const float* m, float G, float eps,
                                          int block, int mode) {
  if (mode == 2) {
    double pe = 0.0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : pe)
    for (long long i = 0; i < (long long)n; i++) {
      double s = 0.0;
      for (size_t j = (size_t)i + 1; j < n; j++) {
        double dx = (double)x[j] - x[i], dy = (double)y[j] - y[i], dz = (double)z[j] - z[i];
        double r = sqrt(dx * dx + dy * dy + dz * dz + (double)eps * (double)eps);
        s -= (double)G * (double)m[i] * (double)m[j] / r;
      }
      pe += s;
    }
    return pe;
  }
  float* per = (float*)malloc(n * sizeof(float));
#pragma omp parallel for schedule(dynamic, 64)
  for (long long i = 0; i < (long long)n; i++) {
    float pe = 0.0f;
    float xi = x[i], yi = y[i], zi = z[i], mi = m[i];
    for (size_t j = (size_t)i + 1; j < n; j++) { /* :97-103 */
      float dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi;
      float r = sqrtf(dx * dx + dy * dy + dz * dz + eps * eps);
      pe -= G * mi * m[j] / r;
    }
    per[i] = pe;
  }
  float* buf = (float*)malloc((size_t)block * sizeof(float));
  float total = 0.0f;
  for (size_t b0 = 0; b0 < n; b0 += (size_t)block) {
    for (int t = 0; t < block; t++) {
      size_t i = b0 + (size_t)t;
      buf[t] = i < n ? per[i] : 0.0f;
    }
    total += block_tree_sum(buf, block);
  }
  free(buf);
  free(per);
  return (double)total;
}

/* ------------------------------------------------------------------------- */
/* a9: spatial hash.  Grid sizing force_spatial_hash.cu:244-246, cell index    */
/*     :28-49 (getCellIndex, clamp), force :83-152 (27 neighbour cells, skip   */
/*     out-of-grid cells, cutoff test on UNSOFTENED r^2, self skipped by       */
/*     index).  The within-cell order is unspecified in the reference          */
/*     (atomic scatter :71-80); the oracle visits bodies in index order and    */
/*     accumulates in fp64 when mode != 0 so order does not matter.            */
/* ------------------------------------------------------------------------- */
ORACLE_API void oracle_bbox(size_t n, const float* x, const float* y, const float* z,
                            float bmin[3], float bmax[3]) {
  /* force_barnes_hut.cu:66-110: plain min/max over the bodies */
  bmin[0] = bmin[1] = bmin[2] = INFINITY;
  bmax[0] = bmax[1] = bmax[2] = -INFINITY;
  for (size_t i = 0; i < n; i++) {
    if (x[i] < bmin[0]) bmin[0] = x[i];
    if (y[i] < bmin[1]) bmin[1] = y[i];
    if (z[i] < bmin[2]) bmin[2] = z[i];
    if (x[i] > bmax[0]) bmax[0] = x[i];
    if (y[i] > bmax[1]) bmax[1] = y[i];
    if (z[i] > bmax[2]) bmax[2] = z[i];
  }
}

ORACLE_API void oracle_grid_dims(const float bmin[3], const float bmax[3], float cell_size,
                                 int dims[3]) {
  /* force_spatial_hash.cu:244-246: dims = (int)ceilf(extent / cell) + 1 */
  for (int a = 0; a < 3; a++) dims[a] = (int)ceilf((bmax[a] - bmin[a]) / cell_size) + 1;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

ORACLE_API int oracle_cell_index(float px, float py, float pz, const float bmin[3],
                                 float cell_size, const int dims[3]) {
  /* force_spatial_hash.cu:36-48: floor((p-min)/cell), clamp to [0,dim-1], x + y*gx + z*gx*gy */
  int cx = (int)floorf((px - bmin[0]) / cell_size);
  int cy = (int)floorf((py - bmin[1]) / cell_size);
  int cz = (int)floorf((pz - bmin[2]) / cell_size);
  cx = clampi(cx, 0, dims[0] - 1);
  cy = clampi(cy, 0, dims[1] - 1);
  cz = clampi(cz, 0, dims[2] - 1);
  return cx + cy * dims[0] + cz * dims[0] * dims[1];
}

ORACLE_API void oracle_assign_cells(size_t n, const float* x, const float* y, const float* z,
                                    const float bmin[3], float cell_size, const int dims[3],
                                    int* cell_of) {
  for (size_t i = 0; i < n; i++) cell_of[i] = oracle_cell_index(x[i], y[i], z[i], bmin, cell_size, dims);
}

/* Grid of SpatialHashGrid::build: bounding box padded by 0.001 on every side
 * (force_spatial_hash.cu:225-231), then dims = ceil(extent / cell) + 1 (:244-246). */
ORACLE_API void oracle_hash_grid(size_t n, const float* x, const float* y, const float* z,
                                 float cell_size, float bmin[3], float bmax[3], int dims[3]) {
  oracle_bbox(n, x, y, z, bmin, bmax);
  for (int a = 0; a < 3; a++) {
    bmin[a] -= 0.001f;
    bmax[a] += 0.001f;
  }
  oracle_grid_dims(bmin, bmax, cell_size, dims);
}

/* |d|^2 of the cutoff test as the reference's device code forms it: `dx*dx + dy*dy + dz*dz`
 * (force_spatial_hash.cu:131) under nvcc's default contraction (-fmad=true) is one product and two
 * fused multiply-adds.  The HIP kernels write the same chain explicitly, so that both sides take the
 * cutoff decision (r2 < cutoff^2, :131-135) on bit-identical values -- a pair sitting on the cutoff
 * sphere to the last ulp would otherwise be counted by one side only, and the truncated force is not
 * small there.  (This file is compiled with -ffp-contract=off: only explicit fmaf() fuses.) */
static inline float hash_dist2(float dx, float dy, float dz) { return fmaf(dz, dz, fmaf(dy, dy, dx * dx)); }

/* Forces on an EXPLICIT grid (origin bmin, dims): the sharded path bins every rank's bodies on
 * the global grid.  n_t <= n: only the first n_t bodies are targets (own bodies first, then halo). */
/* gold / abs_sum (optional, n_t x 3 doubles / n_t doubles): the same pair set (decided by the fp32 cutoff test)
 * summed in fp64 arithmetic, and sum_j |t_ij| of the fp32 term vectors -- kappa_i = abs_sum_i / |a_i| is the
 * condition number of body i's sum: every fp32 evaluation of the terms (this one, the reference's kernel with
 * rsqrtf and contraction, the HIP kernels) is only defined up to ~1 ulp per term, i.e. up to 2^-24 kappa_i
 * relative to |a_i| (tests/test_spatial_hash_gpu.py states its per-body bound with it). */
static int hash_forces_grid(size_t n, size_t n_t, const float* x, const float* y, const float* z, const float* m,
                            float* ax, float* ay, float* az, float G, float eps2, float cell_size, float cutoff,
                            const float bmin[3], const int dims[3], double* gold, double* abs_sum) {
  long long cells = (long long)dims[0] * dims[1] * dims[2];
  if (cells > 100000000LL) return -1; /* force_spatial_hash.cu:252-254 */
  int* cell_of = (int*)malloc(n * sizeof(int));
  int* start = (int*)calloc((size_t)cells + 1, sizeof(int));
  int* order = (int*)malloc(n * sizeof(int));
  oracle_assign_cells(n, x, y, z, bmin, cell_size, dims, cell_of);
  for (size_t i = 0; i < n; i++) start[cell_of[i] + 1]++;
  for (long long c = 0; c < cells; c++) start[c + 1] += start[c];
  int* fill = (int*)malloc((size_t)cells * sizeof(int));
  memcpy(fill, start, (size_t)cells * sizeof(int));
  for (size_t i = 0; i < n; i++) order[fill[cell_of[i]]++] = (int)i; /* stable, index order */
  free(fill);
  const float cutoff2 = cutoff * cutoff; /* :100 */
#pragma omp parallel for schedule(dynamic, 256)
  for (long long i = 0; i < (long long)n_t; i++) {
    float xi = x[i], yi = y[i], zi = z[i];
    int cx = clampi((int)floorf((xi - bmin[0]) / cell_size), 0, dims[0] - 1);
    int cy = clampi((int)floorf((yi - bmin[1]) / cell_size), 0, dims[1] - 1);
    int cz = clampi((int)floorf((zi - bmin[2]) / cell_size), 0, dims[2] - 1);
    double a0 = 0, a1 = 0, a2 = 0;
    double g0 = 0, g1 = 0, g2 = 0, sabs = 0;
    for (int dz = -1; dz <= 1; dz++)
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          int nx = cx + dx, ny = cy + dy, nz = cz + dz;
          if (nx < 0 || nx >= dims[0] || ny < 0 || ny >= dims[1] || nz < 0 || nz >= dims[2])
            continue; /* :110-113 */
          int c = nx + ny * dims[0] + nz * dims[0] * dims[1];
          for (int k = start[c]; k < start[c + 1]; k++) {
            int j = order[k];
            if (j == (int)i) continue; /* :124 */
            float ddx = x[j] - xi, ddy = y[j] - yi, ddz = z[j] - zi;
            float r2 = hash_dist2(ddx, ddy, ddz);
            if (r2 < cutoff2) { /* :131-135: cutoff on the unsoftened distance */
              float dist2 = r2 + eps2;
              float inv = 1.0f / sqrtf(dist2);
              float f = G * m[j] * (inv * inv * inv);
              float t0 = f * ddx, t1 = f * ddy, t2 = f * ddz;
              a0 += (double)t0;
              a1 += (double)t1;
              a2 += (double)t2;
              if (gold || abs_sum) {
                double ex = (double)x[j] - (double)xi, ey = (double)y[j] - (double)yi, ez = (double)z[j] - (double)zi;
                double q = ex * ex + ey * ey + ez * ez + (double)eps2;
                double fg = (double)G * (double)m[j] / (q * sqrt(q));
                g0 += fg * ex; g1 += fg * ey; g2 += fg * ez;
                sabs += sqrt((double)t0 * t0 + (double)t1 * t1 + (double)t2 * t2);
              }
            }
          }
        }
    ax[i] = (float)a0; ay[i] = (float)a1; az[i] = (float)a2;
    if (gold) { gold[3 * i] = g0; gold[3 * i + 1] = g1; gold[3 * i + 2] = g2; }
    if (abs_sum) abs_sum[i] = sabs;
  }
  free(cell_of); free(start); free(order);
  return 0;
}

ORACLE_API int oracle_spatial_hash_forces_grid(size_t n, size_t n_t, const float* x, const float* y,
                                               const float* z, const float* m, float* ax, float* ay,
                                               float* az, float G, float eps2, float cell_size,
                                               float cutoff, const float bmin[3], const int dims[3]) {
  return hash_forces_grid(n, n_t, x, y, z, m, ax, ay, az, G, eps2, cell_size, cutoff, bmin, dims, NULL, NULL);
}

/* the same forces plus, per body, the fp64 sum over the same pair set and the sum of the term magnitudes */
ORACLE_API int oracle_spatial_hash_forces_cond(size_t n, const float* x, const float* y, const float* z,
                                               const float* m, float* ax, float* ay, float* az, float G,
                                               float eps2, float cell_size, float cutoff, double* gold,
                                               double* abs_sum) {
  float bmin[3], bmax[3];
  int dims[3];
  oracle_hash_grid(n, x, y, z, cell_size, bmin, bmax, dims);
  return hash_forces_grid(n, n, x, y, z, m, ax, ay, az, G, eps2, cell_size, cutoff, bmin, dims, gold, abs_sum);
}

ORACLE_API int oracle_spatial_hash_forces(size_t n, const float* x, const float* y,
                                          const float* z, const float* m, float* ax, float* ay,
                                          float* az, float G, float eps2, float cell_size,
                                          float cutoff) {
  float bmin[3], bmax[3];
  int dims[3];
  oracle_hash_grid(n, x, y, z, cell_size, bmin, bmax, dims);
  return oracle_spatial_hash_forces_grid(n, n, x, y, z, m, ax, ay, az, G, eps2, cell_size, cutoff,
                                         bmin, dims);
}

/* Direct sum restricted to pairs with unsoftened r^2 < cutoff^2: what the
 * spatial hash equals when cutoff <= cell_size (27-cell search complete). */
ORACLE_API void oracle_direct_cutoff_forces(size_t n, const float* x, const float* y,
                                            const float* z, const float* m, size_t n_t,
                                            const int64_t* tidx, float* ax, float* ay, float* az,
                                            float G, float eps2, float cutoff) {
  const float cutoff2 = cutoff * cutoff;
#pragma omp parallel for schedule(static)
  for (long long k = 0; k < (long long)n_t; k++) {
    size_t i = (size_t)tidx[k];
    double a0 = 0, a1 = 0, a2 = 0;
    for (size_t j = 0; j < n; j++) {
      if (j == i) continue;
      float dx = x[j] - x[i], dy = y[j] - y[i], dz = z[j] - z[i];
      float r2 = hash_dist2(dx, dy, dz);
      if (r2 < cutoff2) {
        float inv = 1.0f / sqrtf(r2 + eps2);
        float f = G * m[j] * (inv * inv * inv);
        a0 += (double)(f * dx); a1 += (double)(f * dy); a2 += (double)(f * dz);
      }
    }
    ax[k] = (float)a0; ay[k] = (float)a1; az[k] = (float)a2;
  }
}

/* ------------------------------------------------------------------------- */
/* a8: Barnes-Hut.  The reference's host insertion (force_barnes_hut.cu:        */
/*     363-396) never subdivides an occupied leaf, so only   */
/*     its CONTRACT is followed:                                                */
/*       root cube   centre = bbox midpoint, half = max extent/2 + 0.001 (:339-344) */
/*       octree      every cell with more than `leaf_max` bodies is split, down  */
/*                   to `max_depth` levels (cells are addressed on the 2^B       */
/*                   integer grid of the root cube, the Morton quantisation of   */
/*                   :23-38 applied to the cube; B = 10 bits per axis up to      */
/*                   depth 10, 21 bits per axis up to depth 21 -- the reference   */
/*                   caps its insertion at depth 20, :363)                       */
/*       monopoles   mass and centre of mass per node                            */
/*       traversal   :164-195 -- skip massless nodes; accept a node if it is a   */
/*                   leaf or (2 half)^2 / (d^2 + eps^2) < theta^2; a leaf's       */
/*                   bodies interact individually, the body itself is skipped    */
/*     The HIP path builds the same tree from sorted Morton keys, so the two     */
/*     can be compared body by body.                                             */
/* ------------------------------------------------------------------------- */
typedef struct {
  float cx, cy, cz, half;
  double mx, my, mz, mass; /* centre of mass, mass */
  int child[8];
  int first, count; /* range of the Morton-sorted body list */
  int is_leaf;
} ONode;

typedef struct {
  ONode* nodes;
  int count, cap;
  const uint64_t* key;  /* sorted keys, 3 axis_bits wide */
  int axis_bits;        /* 10 or 21 */
  const int* order;     /* sorted position -> body index */
  const float *x, *y, *z, *m;
  float lo[3], half;
  int max_depth, leaf_max;
} OTree;

static uint32_t expand_bits10(uint32_t v) {
  /* force_barnes_hut.cu:23-29 */
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

ORACLE_API void oracle_bh_root(size_t n, const float* x, const float* y, const float* z,
                               float center[3], float* half) {
  float bmin[3], bmax[3];
  oracle_bbox(n, x, y, z, bmin, bmax);
  float ext = 0.0f;
  for (int a = 0; a < 3; a++) {
    center[a] = (bmin[a] + bmax[a]) * 0.5f;
    float e = bmax[a] - bmin[a];
    if (e > ext) ext = e;
  }
  *half = ext * 0.5f + 0.001f;
}

ORACLE_API uint32_t oracle_bh_key(float px, float py, float pz, const float lo[3], float scale) {
  int q[3];
  const float p[3] = {px, py, pz};
  for (int a = 0; a < 3; a++) {
    float f = (p[a] - lo[a]) * scale;
    int v = (int)f;
    q[a] = v < 0 ? 0 : (v > 1023 ? 1023 : v);
  }
  return (expand_bits10((uint32_t)q[0]) << 2) | (expand_bits10((uint32_t)q[1]) << 1) |
         expand_bits10((uint32_t)q[2]);
}

/* the same with `bits` (10 or 21) per axis; `scale` = 2^bits / (2 half).  Plain bit loop: the oracle
 * does not have to be fast here. */
static uint64_t bh_key_bits(float px, float py, float pz, const float lo[3], float scale, int bits) {
  const float p[3] = {px, py, pz};
  const int top = (1 << bits) - 1;
  uint64_t key = 0;
  for (int a = 0; a < 3; a++) {
    float f = (p[a] - lo[a]) * scale;
    int v = (int)f;
    uint64_t q = (uint64_t)(v < 0 ? 0 : (v > top ? top : v));
    for (int b = 0; b < bits; b++) key |= ((q >> b) & 1ull) << (3 * b + (2 - a));
  }
  return key;
}

static int otree_new(OTree* t) {
  if (t->count == t->cap) {
    t->cap = t->cap ? t->cap * 2 : 1024;
    t->nodes = (ONode*)realloc(t->nodes, (size_t)t->cap * sizeof(ONode));
  }
  ONode* nd = &t->nodes[t->count];
  memset(nd, 0, sizeof(*nd));
  for (int k = 0; k < 8; k++) nd->child[k] = -1;
  return t->count++;
}

/* node over sorted range [first, first+count) at `level`; cell = integer cell coords at level */
static int otree_build(OTree* t, int first, int count, int level) {
  int ni = otree_new(t);
  {
    ONode* nd = &t->nodes[ni];
    nd->first = first;
    nd->count = count;
    /* geometry from the key prefix */
    uint64_t k = t->key[first];
    uint32_t ix = 0, iy = 0, iz = 0;
    for (int b = 0; b < t->axis_bits; b++) {
      ix |= (uint32_t)((k >> (3 * b + 2)) & 1ull) << b;
      iy |= (uint32_t)((k >> (3 * b + 1)) & 1ull) << b;
      iz |= (uint32_t)((k >> (3 * b + 0)) & 1ull) << b;
    }
    int sh = t->axis_bits - level;
    float h = ldexpf(t->half, -level);
    nd->half = h;
    nd->cx = t->lo[0] + ((float)(ix >> sh) + 0.5f) * (2.0f * h);
    nd->cy = t->lo[1] + ((float)(iy >> sh) + 0.5f) * (2.0f * h);
    nd->cz = t->lo[2] + ((float)(iz >> sh) + 0.5f) * (2.0f * h);
    nd->is_leaf = (count <= t->leaf_max) || (level >= t->max_depth);
  }
  if (t->nodes[ni].is_leaf) {
    double mx = 0, my = 0, mz = 0, ms = 0;
    for (int k = first; k < first + count; k++) {
      int b = t->order[k];
      double mb = t->m[b];
      mx += mb * (double)t->x[b]; my += mb * (double)t->y[b]; mz += mb * (double)t->z[b];
      ms += mb;
    }
    ONode* nd = &t->nodes[ni];
    nd->mass = ms;
    if (count == 1) {
      int b = t->order[first];
      nd->mx = t->x[b]; nd->my = t->y[b]; nd->mz = t->z[b];
    } else if (ms > 0) {
      nd->mx = mx / ms; nd->my = my / ms; nd->mz = mz / ms;
    } else {
      nd->mx = nd->cx; nd->my = nd->cy; nd->mz = nd->cz;
    }
    return ni;
  }
  int shift = 3 * t->axis_bits - 3 - 3 * level;
  int k = first;
  double mx = 0, my = 0, mz = 0, ms = 0;
  while (k < first + count) {
    uint64_t oct = (t->key[k] >> shift) & 7ull;
    int e = k;
    while (e < first + count && ((t->key[e] >> shift) & 7ull) == oct) e++;
    int c = otree_build(t, k, e - k, level + 1);
    t->nodes[ni].child[oct] = c;
    ONode* cn = &t->nodes[c];
    mx += cn->mx * cn->mass; my += cn->my * cn->mass; mz += cn->mz * cn->mass; ms += cn->mass;
    k = e;
  }
  ONode* nd = &t->nodes[ni];
  nd->mass = ms;
  if (ms > 0) { nd->mx = mx / ms; nd->my = my / ms; nd->mz = mz / ms; }
  else { nd->mx = nd->cx; nd->my = nd->cy; nd->mz = nd->cz; }
  return ni;
}

typedef struct { uint64_t key; int idx; } KeyIdx;
/* |d|^2 of the Barnes-Hut walk as the device code forms it: dx*dx, then two fused multiply-adds
 * (force_barnes_hut.cu:165 `dx*dx + dy*dy + dz*dz` -- nvcc contracts exactly this chain with its
 * default -fmad=true; the HIP walk writes the same chain explicitly, so the opening decisions of
 * both sides are taken on bit-identical distances). */
static inline float bh_dist2(float dx, float dy, float dz) { return fmaf(dz, dz, fmaf(dy, dy, dx * dx)); }

static int cmp_keyidx(const void* a, const void* b) {
  const KeyIdx* p = (const KeyIdx*)a; const KeyIdx* q = (const KeyIdx*)b;
  if (p->key != q->key) return p->key < q->key ? -1 : 1;
  return p->idx < q->idx ? -1 : (p->idx > q->idx ? 1 : 0);
}

/* Accelerations of the bodies listed in tidx (body indices).  Outputs: root mass, node count,
 * and optionally the Morton-sorted order (sorted position -> body index, n ints). */
ORACLE_API int oracle_barnes_hut_forces(size_t n, const float* x, const float* y, const float* z,
                                        const float* m, size_t n_t, const int64_t* tidx,
                                        float* ax, float* ay, float* az, float G, float eps2,
                                        float theta, int max_depth, int leaf_max,
                                        double* root_mass, int* node_count, int* order_out) {
  if (max_depth < 1) max_depth = 1;
  if (max_depth > 21) max_depth = 21;
  if (leaf_max < 1) leaf_max = 1;
  /* 10 bits per axis up to depth 10 (the 30-bit keys of :23-38), 21 beyond; the finer grid refines the
   * coarser one: its scale is the 10-bit scale times 2^11, an exact fp32 scaling */
  const int axis_bits = max_depth <= 10 ? 10 : 21;
  float center[3], half;
  oracle_bh_root(n, x, y, z, center, &half);
  OTree t;
  memset(&t, 0, sizeof(t));
  t.half = half;
  for (int a = 0; a < 3; a++) t.lo[a] = center[a] - half;
  const float scale = (1024.0f / (2.0f * half)) * (axis_bits == 10 ? 1.0f : 2048.0f);
  KeyIdx* ki = (KeyIdx*)malloc(n * sizeof(KeyIdx));
  for (size_t i = 0; i < n; i++) {
    ki[i].key = axis_bits == 10 ? (uint64_t)oracle_bh_key(x[i], y[i], z[i], t.lo, scale)
                                : bh_key_bits(x[i], y[i], z[i], t.lo, scale, axis_bits);
    ki[i].idx = (int)i;
  }
  qsort(ki, n, sizeof(KeyIdx), cmp_keyidx);
  uint64_t* key = (uint64_t*)malloc(n * sizeof(uint64_t));
  int* order = (int*)malloc(n * sizeof(int));
  int* pos_of = (int*)malloc(n * sizeof(int));
  for (size_t k = 0; k < n; k++) { key[k] = ki[k].key; order[k] = ki[k].idx; pos_of[ki[k].idx] = (int)k; }
  free(ki);
  t.key = key; t.axis_bits = axis_bits; t.order = order; t.x = x; t.y = y; t.z = z; t.m = m;
  t.max_depth = max_depth; t.leaf_max = leaf_max;
  otree_build(&t, 0, (int)n, 0);
  if (root_mass) *root_mass = t.nodes[0].mass;
  if (node_count) *node_count = t.count;
  if (order_out) memcpy(order_out, order, n * sizeof(int));
  const float theta2 = theta * theta;
  const ONode* nodes = t.nodes;
#pragma omp parallel for schedule(dynamic, 64)
  for (long long k = 0; k < (long long)n_t; k++) {
    int i = (int)tidx[k];
    float xi = x[i], yi = y[i], zi = z[i];
    double a0 = 0, a1 = 0, a2 = 0;
    int stack[8 * 24];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
      const ONode* nd = &nodes[stack[--sp]];
      if (nd->mass == 0.0) continue; /* :161-162 */
      if (nd->is_leaf) {
        for (int q = nd->first; q < nd->first + nd->count; q++) {
          int j = order[q];
          if (j == i) continue; /* :175 self */
          float dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi;
          float dist2 = bh_dist2(dx, dy, dz) + eps2;
          float inv = 1.0f / sqrtf(dist2);
          float f = G * m[j] * (inv * inv * inv);
          a0 += (double)(f * dx); a1 += (double)(f * dy); a2 += (double)(f * dz);
        }
        continue;
      }
      float dx = (float)nd->mx - xi, dy = (float)nd->my - yi, dz = (float)nd->mz - zi;
      float dist2 = bh_dist2(dx, dy, dz) + eps2; /* :165 */
      float size = 2.0f * nd->half;
      float size2 = size * size;                         /* :168 */
      /* :171-172 `size2 / dist2 < theta2`, evaluated as size2 < theta2 * dist2 (dist2 > 0): the
       * same inequality without a division; the HIP traversal uses the same form, so that both
       * sides take identical decisions */
      if (size2 < theta2 * dist2) {
        float inv = 1.0f / sqrtf(dist2);
        float f = G * (float)nd->mass * (inv * inv * inv);
        a0 += (double)(f * dx); a1 += (double)(f * dy); a2 += (double)(f * dz);
      } else {
        for (int c = 7; c >= 0; c--) if (nd->child[c] >= 0) stack[sp++] = nd->child[c];
      }
    }
    ax[k] = (float)a0; ay[k] = (float)a1; az[k] = (float)a2;
  }
  free(t.nodes); free(key); free(order); free(pos_of);
  return 0;
}
