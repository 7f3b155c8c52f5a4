"""Helpers shared by the -m gpu tests."""
import numpy as np
import torch

from oracle_bind import host_state


def to_device(nb, ic):
    """ParticleData on the device filled from an initial-condition dict (the reference tests'
    allocateDevice / allocateHost / copyToDevice sequence, tests/test_force_calculation.cpp:62-80)."""
    s = host_state(ic)
    n = s["pos_x"].size
    h = nb.ParticleData()
    nb.ParticleDataManager.allocateHost(h, n)
    for k, v in s.items():
        getattr(h, k)[:] = v
    d = nb.ParticleData()
    nb.ParticleDataManager.allocateDevice(d, n)
    nb.ParticleDataManager.copyToDevice(d, h)
    return d, h


def acc_of(d):
    return np.stack([d.acc_x.cpu().numpy(), d.acc_y.cpu().numpy(), d.acc_z.cpu().numpy()], 1)


def rel_err(a, ref):
    """per-body ||a - ref|| / ||ref|| (the parity metric)."""
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    return np.linalg.norm(a - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-300)


def packed(ic, device="cuda"):
    p = np.stack([ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]], 1).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(p)).to(device)


# ---- the a-priori bound of the spatial-hash comparisons ("A-priori bound") --------------------
# A body's acceleration is a sum of fp32 terms t_ij that may nearly cancel; kappa_i = sum_j |t_ij| / |a_i| (computed by
# the oracle) is the condition number of that sum.  Two evaluations of the reference's loop
# (ref: src/cuda/force_spatial_hash.cu:118-146) share dx, dy, dz, r2 and r2 + eps2 BIT FOR BIT (same operands, same
# operations, hash_dist2); they differ, to first order in u = 2^-24, by at most
#   C_TERMS_*  x u x |t_ij|  per term:
#     GPU kernels: v_rsq_f32 (<= 1 ulp = 2u relative), cubed -> 6u; the three roundings of (m inv) (inv inv) -> 3u; the
#                  product f d is exact inside the FMA                                                   => 9
#     oracle     : 1 / sqrtf = two correctly rounded operations -> 2u, cubed -> 6u; G m, inv inv, (inv inv) inv, their
#                  product -> 4u; f d -> 1u                                                               => 11
#   C_SUM_FP32 x u x sum_j |t_ij|  for the accumulation: an fp32 running sum of n terms is within (n - 1) u sum |t| of the
#     exact sum of those terms; the longest fp32 run any kernel keeps is 64 (cell-run kernel; 32 in the wave-per-cell
#     kernels) before it is folded into fp64, whose own error (2^-53 per operation) is negligible        => 63
# and by the final rounding of the result to fp32 (u |a_i|, below 1e-5 by two orders of magnitude).  These are worst
# cases (every rounding at its limit, all with the same sign): independent roundings would give about
# 3.5 u sqrt(sum |t|^2) ~ 0.5-1 x u kappa, and the kernels are measured at <= 2.2 x u kappa over 4,194,304 bodies
# (profiles/r03_full_population_parity.txt).  EVERY comparison of the spatial-hash tests and tools asserts
#     err_i <= max(1e-5, C u kappa_i)
# with C from this table -- nothing is fitted to the data -- and prints the measured margin.
U = 2.0 ** -24
C_TERMS_GPU, C_TERMS_ORACLE, C_SUM_FP32 = 9, 11, 63
HASH_C = {"oracle": C_TERMS_GPU + C_TERMS_ORACLE + C_SUM_FP32,   # a kernel against the oracle (fp64 accumulation)        83
          "gold": C_TERMS_GPU + C_SUM_FP32,                      # a kernel against the fp64 evaluation of the same pairs  72
          "gpu": 2 * C_SUM_FP32}                                 # two kernel evaluations (identical terms, other grouping) 126


def hash_bound(kappa, kind="oracle", tol=1e-5):
    """per-body bound max(tol, C u kappa) of a spatial-hash comparison (see the table above)"""
    return np.maximum(tol, HASH_C[kind] * U * np.asarray(kappa, np.float64))


def hash_margin(err, kappa):
    """largest err / (u kappa) among the bodies above 1e-5 (0 if none): how far inside the worst case the run sits"""
    over = err > 1e-5
    return float((err[over] / (U * kappa[over])).max()) if over.any() else 0.0
