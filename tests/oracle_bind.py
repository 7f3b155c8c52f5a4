"""ctypes binding of oracle/libnbody_oracle.so -- the CPU checker.  Only tests/, smoke() and
bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")

_f = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i64 = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i32 = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f3 = C.c_float * 3
_i3 = C.c_int * 3


class Oracle:
    def __init__(self, path):
        L = C.CDLL(path)
        self.L = L
        L.oracle_num_threads.restype = C.c_int
        L.oracle_set_num_threads.argtypes = [C.c_int]
        L.oracle_pair_force.argtypes = [_f3, _f3, C.c_float, C.c_float, C.c_float, C.c_float, _f3]
        L.oracle_direct_forces.argtypes = [C.c_size_t, _f, _f, _f, _f, C.c_size_t, C.c_size_t, _f, _f,
                                           _f, C.c_float, C.c_float, C.c_int]
        L.oracle_direct_forces_indexed.argtypes = [C.c_size_t, _f, _f, _f, _f, C.c_size_t, _i64, _f,
                                                   _f, _f, C.c_float, C.c_float, C.c_int]
        L.oracle_direct_forces_points.argtypes = [C.c_size_t, _f, _f, _f, _f, C.c_size_t, _f, _f, _f,
                                                  _f, _f, _f, C.c_float, C.c_float]
        L.oracle_update_positions.argtypes = [C.c_size_t] + [_f] * 9 + [C.c_float]
        L.oracle_update_velocities.argtypes = [C.c_size_t] + [_f] * 9 + [C.c_float]
        L.oracle_integrate_direct.argtypes = [C.c_size_t] + [_f] * 13 + [C.c_float, C.c_float,
                                                                         C.c_float, C.c_int, C.c_int]
        L.oracle_kinetic_energy.argtypes = [C.c_size_t, _f, _f, _f, _f, C.c_int, C.c_int]
        L.oracle_kinetic_energy.restype = C.c_double
        L.oracle_potential_energy.argtypes = [C.c_size_t, _f, _f, _f, _f, C.c_float, C.c_float,
                                              C.c_int, C.c_int]
        L.oracle_potential_energy.restype = C.c_double
        L.oracle_bbox.argtypes = [C.c_size_t, _f, _f, _f, _f3, _f3]
        L.oracle_grid_dims.argtypes = [_f3, _f3, C.c_float, _i3]
        L.oracle_cell_index.argtypes = [C.c_float, C.c_float, C.c_float, _f3, C.c_float, _i3]
        L.oracle_cell_index.restype = C.c_int
        L.oracle_assign_cells.argtypes = [C.c_size_t, _f, _f, _f, _f3, C.c_float, _i3, _i32]
        L.oracle_hash_grid.argtypes = [C.c_size_t, _f, _f, _f, C.c_float, _f3, _f3, _i3]
        L.oracle_spatial_hash_forces.argtypes = [C.c_size_t, _f, _f, _f, _f, _f, _f, _f, C.c_float,
                                                 C.c_float, C.c_float, C.c_float]
        L.oracle_spatial_hash_forces.restype = C.c_int
        L.oracle_spatial_hash_forces_grid.argtypes = [C.c_size_t, C.c_size_t, _f, _f, _f, _f, _f, _f, _f,
                                                      C.c_float, C.c_float, C.c_float, C.c_float, _f3, _i3]
        L.oracle_spatial_hash_forces_grid.restype = C.c_int
        _d = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        L.oracle_spatial_hash_forces_cond.argtypes = [C.c_size_t, _f, _f, _f, _f, _f, _f, _f, C.c_float, C.c_float,
                                                      C.c_float, C.c_float, _d, _d]
        L.oracle_spatial_hash_forces_cond.restype = C.c_int
        L.oracle_direct_cutoff_forces.argtypes = [C.c_size_t, _f, _f, _f, _f, C.c_size_t, _i64, _f, _f,
                                                  _f, C.c_float, C.c_float, C.c_float]
        L.oracle_barnes_hut_forces.argtypes = [C.c_size_t, _f, _f, _f, _f, C.c_size_t, _i64, _f, _f, _f,
                                               C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                               C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p]
        L.oracle_bh_root.argtypes = [C.c_size_t, _f, _f, _f, _f3, C.POINTER(C.c_float)]
        L.oracle_barnes_hut_forces.restype = C.c_int

    # -- convenience wrappers (numpy in / numpy out) ------------------------------------------
    def num_threads(self):
        return self.L.oracle_num_threads()

    def pair_force(self, p1, p2, m1, m2, G, eps):
        out = _f3()
        self.L.oracle_pair_force(_f3(*p1), _f3(*p2), m1, m2, G, eps, out)
        return np.array(list(out), dtype=np.float32)

    def direct_forces(self, x, y, z, m, G, eps2, mode=1, t0=0, t1=None):
        n = x.size
        t1 = n if t1 is None else t1
        ax, ay, az = (np.empty(t1 - t0, np.float32) for _ in range(3))
        self.L.oracle_direct_forces(n, x, y, z, m, t0, t1, ax, ay, az, G, eps2, mode)
        return ax, ay, az

    def direct_forces_indexed(self, x, y, z, m, idx, G, eps2, mode=1):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        ax, ay, az = (np.empty(idx.size, np.float32) for _ in range(3))
        self.L.oracle_direct_forces_indexed(x.size, x, y, z, m, idx.size, idx, ax, ay, az, G, eps2, mode)
        return ax, ay, az

    def direct_forces_points(self, sx, sy, sz, sm, tx, ty, tz, G, eps2):
        ax, ay, az = (np.empty(tx.size, np.float32) for _ in range(3))
        self.L.oracle_direct_forces_points(sx.size, sx, sy, sz, sm, tx.size, tx, ty, tz, ax, ay, az, G, eps2)
        return ax, ay, az

    def update_positions(self, s, dt):
        self.L.oracle_update_positions(s["pos_x"].size, s["pos_x"], s["pos_y"], s["pos_z"],
                                       s["vel_x"], s["vel_y"], s["vel_z"], s["acc_x"], s["acc_y"],
                                       s["acc_z"], dt)

    def update_velocities(self, s, dt):
        self.L.oracle_update_velocities(s["vel_x"].size, s["vel_x"], s["vel_y"], s["vel_z"],
                                        s["acc_old_x"], s["acc_old_y"], s["acc_old_z"], s["acc_x"],
                                        s["acc_y"], s["acc_z"], dt)

    def integrate_direct(self, s, G, eps, dt, steps, mode=1):
        self.L.oracle_integrate_direct(s["pos_x"].size, s["pos_x"], s["pos_y"], s["pos_z"], s["vel_x"],
                                       s["vel_y"], s["vel_z"], s["acc_x"], s["acc_y"], s["acc_z"],
                                       s["acc_old_x"], s["acc_old_y"], s["acc_old_z"], s["mass"], G,
                                       eps, dt, steps, mode)

    def kinetic_energy(self, s, block=256, mode=0):
        return self.L.oracle_kinetic_energy(s["vel_x"].size, s["vel_x"], s["vel_y"], s["vel_z"],
                                            s["mass"], block, mode)

    def potential_energy(self, s, G, eps, block=256, mode=0):
        return self.L.oracle_potential_energy(s["pos_x"].size, s["pos_x"], s["pos_y"], s["pos_z"],
                                              s["mass"], G, eps, block, mode)

    def cell_index(self, p, bmin, cell, dims):
        return self.L.oracle_cell_index(p[0], p[1], p[2], _f3(*bmin), cell, _i3(*dims))

    def grid_dims(self, bmin, bmax, cell):
        d = _i3()
        self.L.oracle_grid_dims(_f3(*bmin), _f3(*bmax), cell, d)
        return list(d)

    def bbox(self, x, y, z):
        lo, hi = _f3(), _f3()
        self.L.oracle_bbox(x.size, x, y, z, lo, hi)
        return list(lo), list(hi)

    def hash_grid(self, x, y, z, cell):
        lo, hi, d = _f3(), _f3(), _i3()
        self.L.oracle_hash_grid(x.size, x, y, z, cell, lo, hi, d)
        return list(lo), list(hi), list(d)

    def spatial_hash_forces(self, x, y, z, m, G, eps2, cell, cutoff):
        ax, ay, az = (np.empty(x.size, np.float32) for _ in range(3))
        rc = self.L.oracle_spatial_hash_forces(x.size, x, y, z, m, ax, ay, az, G, eps2, cell, cutoff)
        if rc:
            raise RuntimeError("grid too large")
        return ax, ay, az

    def spatial_hash_forces_cond(self, x, y, z, m, G, eps2, cell, cutoff):
        """(acc [n,3] float32, gold [n,3] float64, kappa [n]): the oracle's forces, the fp64 sum over the same pair
        set, and the condition number sum_j |t_ij| / |a_i| of every body's sum."""
        ax, ay, az = (np.empty(x.size, np.float32) for _ in range(3))
        gold, sabs = np.empty((x.size, 3), np.float64), np.empty(x.size, np.float64)
        if self.L.oracle_spatial_hash_forces_cond(x.size, x, y, z, m, ax, ay, az, G, eps2, cell, cutoff, gold, sabs):
            raise RuntimeError("grid too large")
        acc = np.stack([ax, ay, az], 1)
        kappa = sabs / np.maximum(np.linalg.norm(acc.astype(np.float64), axis=1), 1e-300)
        return acc, gold, kappa

    def spatial_hash_forces_grid(self, x, y, z, m, n_t, G, eps2, cell, cutoff, bmin, dims):
        ax, ay, az = (np.empty(n_t, np.float32) for _ in range(3))
        rc = self.L.oracle_spatial_hash_forces_grid(x.size, n_t, x, y, z, m, ax, ay, az, G, eps2, cell,
                                                    cutoff, _f3(*bmin), _i3(*dims))
        if rc:
            raise RuntimeError("grid too large")
        return ax, ay, az

    def direct_cutoff_forces(self, x, y, z, m, idx, G, eps2, cutoff):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        ax, ay, az = (np.empty(idx.size, np.float32) for _ in range(3))
        self.L.oracle_direct_cutoff_forces(x.size, x, y, z, m, idx.size, idx, ax, ay, az, G, eps2, cutoff)
        return ax, ay, az

    def barnes_hut_forces(self, x, y, z, m, idx, G, eps2, theta, max_depth=20, leaf_max=1,
                          want_order=False):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        ax, ay, az = (np.empty(idx.size, np.float32) for _ in range(3))
        rm, nc = C.c_double(), C.c_int()
        order = np.empty(x.size, np.int32) if want_order else None
        self.L.oracle_barnes_hut_forces(x.size, x, y, z, m, idx.size, idx, ax, ay, az, G, eps2, theta,
                                        max_depth, leaf_max, C.byref(rm), C.byref(nc),
                                        order.ctypes.data if want_order else None)
        if want_order:
            return ax, ay, az, rm.value, nc.value, order
        return ax, ay, az, rm.value, nc.value

    def bh_root(self, x, y, z):
        c, h = _f3(), C.c_float()
        self.L.oracle_bh_root(x.size, x, y, z, c, C.byref(h))
        return list(c), h.value


def host_state(ic: dict) -> dict:
    """13-array host state (numpy float32) from an initial-condition dict."""
    n = ic["pos_x"].size
    s = {k: np.ascontiguousarray(v, dtype=np.float32).copy() for k, v in ic.items()}
    for k in ("vel_x", "vel_y", "vel_z", "acc_x", "acc_y", "acc_z", "acc_old_x", "acc_old_y", "acc_old_z"):
        s.setdefault(k, np.zeros(n, np.float32))
    return s


_cache = {}


def load(fast: bool = False) -> Oracle:
    name = "libnbody_oracle_fast.so" if fast else "libnbody_oracle.so"
    if name in _cache:
        return _cache[name]
    path = os.path.join(ODIR, name)
    src = os.path.join(ODIR, "nbody_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ODIR, name], stdout=subprocess.DEVNULL)
    _cache[name] = Oracle(path)
    return _cache[name]
