"""Synthetic initial conditions for tests and bench.py (host side, numpy, seeded).

The reference ships only UNIFORM / SPHERICAL / DISK initialisers
(src/cuda/particle_init.cu:286-357) built on std::mt19937 + libstdc++ distributions, whose float
mapping is implementation-defined; BASELINE.json's workloads (Plummer sphere, two-galaxy
collision, uniform box) are therefore generated here with numpy's PCG64 so that the same seed
gives the same bodies on every box.  All outputs are float32 SoA arrays.
"""
from __future__ import annotations

import numpy as np


def plummer(n: int, seed: int = 42, a: float = 1.0, rmax: float = 10.0, total_mass: float = 1.0):
    """Plummer sphere (scale length a, truncated at rmax*a), Aarseth-Henon-Wielen velocities,
    G = 1, equal masses total_mass/n, centred on its centre of mass.  SURVEY.md section 8d (1)."""
    rng = np.random.default_rng(seed)
    r = np.empty(n)
    filled = 0
    while filled < n:
        u = rng.random(n - filled)
        u = u[(u > 1e-12) & (u < 1.0)]
        rr = a / np.sqrt(u ** (-2.0 / 3.0) - 1.0)
        rr = rr[rr <= rmax * a]
        r[filled:filled + rr.size] = rr
        filled += rr.size
    cos_t = rng.uniform(-1.0, 1.0, n)
    sin_t = np.sqrt(1.0 - cos_t * cos_t)
    phi = rng.uniform(0.0, 2.0 * np.pi, n)
    x, y, z = r * sin_t * np.cos(phi), r * sin_t * np.sin(phi), r * cos_t
    # speed: q = v / v_esc drawn from g(q) = q^2 (1-q^2)^(7/2) by rejection
    q = np.empty(n)
    filled = 0
    while filled < n:
        m = n - filled
        qq = rng.random(m)
        yy = rng.random(m) * 0.1
        ok = yy < qq * qq * (1.0 - qq * qq) ** 3.5
        k = int(ok.sum())
        q[filled:filled + k] = qq[ok]
        filled += k
    vesc = np.sqrt(2.0 * total_mass / a) * (1.0 + (r / a) ** 2) ** -0.25
    v = q * vesc
    cos_t = rng.uniform(-1.0, 1.0, n)
    sin_t = np.sqrt(1.0 - cos_t * cos_t)
    phi = rng.uniform(0.0, 2.0 * np.pi, n)
    vx, vy, vz = v * sin_t * np.cos(phi), v * sin_t * np.sin(phi), v * cos_t
    mass = np.full(n, total_mass / n)
    for arr in (x, y, z, vx, vy, vz):
        arr -= arr.mean()
    f = np.float32
    return dict(pos_x=x.astype(f), pos_y=y.astype(f), pos_z=z.astype(f), vel_x=vx.astype(f),
                vel_y=vy.astype(f), vel_z=vz.astype(f), mass=mass.astype(f))


def uniform_box(n: int, seed: int = 42, lo: float = -32.0, hi: float = 32.0,
                min_mass: float = 1.0, max_mass: float = 1.0):
    """Uniform box, zero velocities (recipe of initUniform, particle_init.cu:290-309)."""
    rng = np.random.default_rng(seed)
    f = np.float32
    out = {k: rng.uniform(lo, hi, n).astype(f) for k in ("pos_x", "pos_y", "pos_z")}
    for k in ("vel_x", "vel_y", "vel_z"):
        out[k] = np.zeros(n, f)
    out["mass"] = rng.uniform(min_mass, max_mass, n).astype(f) if max_mass > min_mass \
        else np.full(n, min_mass, f)
    return out


def sphere(n: int, seed: int = 42, radius: float = 10.0, center=(0.0, 0.0, 0.0),
           min_mass: float = 1.0, max_mass: float = 1.0):
    """Uniform-in-volume sphere, zero velocities (recipe of initSpherical,
    particle_init.cu:311-331: r = R cbrt(u), theta = 2 pi u, phi = acos(2u-1))."""
    rng = np.random.default_rng(seed)
    u = rng.random((n, 3))
    r = np.cbrt(u[:, 0]) * radius
    theta = u[:, 1] * 2.0 * np.pi
    phi = np.arccos(2.0 * u[:, 2] - 1.0)
    f = np.float32
    out = dict(pos_x=(center[0] + r * np.sin(phi) * np.cos(theta)).astype(f),
               pos_y=(center[1] + r * np.sin(phi) * np.sin(theta)).astype(f),
               pos_z=(center[2] + r * np.cos(phi)).astype(f))
    for k in ("vel_x", "vel_y", "vel_z"):
        out[k] = np.zeros(n, f)
    out["mass"] = rng.uniform(min_mass, max_mass, n).astype(f) if max_mass > min_mass \
        else np.full(n, min_mass, f)
    return out


def disk(n: int, seed: int = 42, radius: float = 10.0, thickness: float = 1.0,
         center=(0.0, 0.0, 0.0), rotation_speed: float = 1.0, mass: float = 1.0):
    """Flat disk with tangential v = omega sqrt(r) (recipe of initDisk, particle_init.cu:333-357)."""
    rng = np.random.default_rng(seed)
    u = rng.random((n, 3))
    r = np.sqrt(u[:, 0]) * radius
    theta = u[:, 1] * 2.0 * np.pi
    z = (u[:, 2] - 0.5) * thickness
    v = rotation_speed * np.sqrt(r)
    f = np.float32
    return dict(pos_x=(center[0] + r * np.cos(theta)).astype(f),
                pos_y=(center[1] + r * np.sin(theta)).astype(f),
                pos_z=(center[2] + z).astype(f), vel_x=(-v * np.sin(theta)).astype(f),
                vel_y=(v * np.cos(theta)).astype(f), vel_z=np.zeros(n, f),
                mass=np.full(n, mass, f))


def two_galaxies(n: int, seed: int = 42):
    """Two DISK distributions offset +-(15,3,0) with -+(0.5,0,0) bulk velocity
    (SURVEY.md section 8d (4))."""
    n1 = n // 2
    a = disk(n1, seed, center=(-15.0, -3.0, 0.0))
    b = disk(n - n1, seed + 1, center=(15.0, 3.0, 0.0))
    a["vel_x"] += np.float32(0.5)
    b["vel_x"] -= np.float32(0.5)
    return {k: np.concatenate([a[k], b[k]]) for k in a}
