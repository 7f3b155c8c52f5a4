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
