"""include/nbody_hip_comm.h without a GPU: the exported partition logic (nbody_hip_shard_bounds,
nbody_hip_pair_schedule -- the C++ twin of distributed.pair_schedule) and argument / no-device behaviour of the
communicator entry points.  The N-rank data path itself runs under -m gpu (tests/test_comm_gpu.py: virtual ranks
on one device, one RCCL rank) and, for the torch.distributed host, on gloo in tests/test_sharded_cpu.py."""
import ctypes as C

import numpy as np
import pytest


@pytest.mark.parametrize("n,world", [(1, 1), (10, 3), (4096, 8), (1000003, 8), (7, 8), (1 << 20, 8), (12345, 32)])
def test_shard_bounds_match_the_python_host(nb, n, world):
    from nbody_amd.distributed import shard_bounds as py
    from nbody_amd.sharded import shard_bounds as c
    got = [c(n, world, r) for r in range(world)]
    assert got == [py(n, world, r) for r in range(world)]
    assert got[0][1] == 0 and got[-1][2] == n
    assert all(a[2] == b[1] for a, b in zip(got, got[1:]))         # contiguous, in rank order
    assert all(hi - lo <= S for S, lo, hi in got)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 31, 32])
@pytest.mark.parametrize("S", [1, 2, 5, 64, 131072])
def test_pair_schedule_covers_every_cross_shard_pair_once(nb, world, S):
    from nbody_amd.distributed import pair_schedule as py
    from nbody_amd.sharded import pair_schedule as c
    cover = np.zeros((world, world), dtype=np.int64)     # body pairs (in units of bodies^2) per ordered shard pair
    work = []
    for r in range(world):
        rows = c(world, r, S)
        assert rows == [tuple(t) for t in py(world, r, S)]
        assert len(rows) <= world // 2 + 1
        w = 0
        for i0, i1, sh, j0, j1 in rows:
            assert sh != r and 0 <= i0 < i1 <= S and 0 <= j0 < j1 <= S
            cover[r, sh] += (i1 - i0) * (j1 - j0)
            w += (i1 - i0) * (j1 - j0)
        work.append(w)
    both = cover + cover.T                                # a pair of shards may be split between its two owners
    off = ~np.eye(world, dtype=bool)
    assert np.all(both[off] == S * S) and np.all(np.diag(cover) == 0)
    if world > 1 and S > 1:
        assert max(work) - min(work) <= S * S // 2 + S   # balanced up to the antipodal half for odd S


def test_pair_schedule_exact_rectangles(nb):
    """beyond the area count: the two halves of an antipodal rectangle are complementary"""
    from nbody_amd.sharded import pair_schedule
    W, S = 8, 7
    seen = np.zeros((W * S, W * S), dtype=np.int8)
    for r in range(W):
        for i0, i1, sh, j0, j1 in pair_schedule(W, r, S):
            seen[r * S + i0:r * S + i1, sh * S + j0:sh * S + j1] += 1
    sym = seen + seen.T
    for a in range(W):
        for b in range(W):
            blk = sym[a * S:(a + 1) * S, b * S:(b + 1) * S]
            assert np.all(blk == (0 if a == b else 1))


def test_comm_entry_points_without_a_device(nb):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = nb._lib.load()
    h = C.c_void_p()
    assert lib.nbody_hip_comm_init_all(2, None, 0, C.byref(h)) == nb._lib.ERR_DEVICE and not h.value
    assert b"no HIP device" in lib.nbody_hip_last_error()
    assert lib.nbody_hip_comm_init_all(0, None, 0, C.byref(h)) == nb._lib.ERR_VALIDATION
    assert lib.nbody_hip_comm_init_all(2, None, 7, C.byref(h)) == nb._lib.ERR_VALIDATION
    uid = C.create_string_buffer(128)
    assert lib.nbody_hip_comm_init_rank(0, 3, 2, uid, C.byref(h)) == nb._lib.ERR_VALIDATION   # rank outside the world
    assert lib.nbody_hip_comm_init_rank(0, 0, 2, uid, C.byref(h)) == nb._lib.ERR_DEVICE
    assert lib.nbody_hip_sharded_direct_create(None, 10, 1.0, 0.1, C.byref(h)) == nb._lib.ERR_STATE
    assert lib.nbody_hip_sharded_direct_step(None, 0.1, 1) == nb._lib.ERR_STATE
    assert lib.nbody_hip_comm_destroy(None) == 0 and lib.nbody_hip_sharded_direct_destroy(None) == 0
    assert lib.nbody_hip_sharded_hash_create(None, 10, 1.0, 0.1, 1.0, 1.0, C.byref(h)) == nb._lib.ERR_STATE
    assert lib.nbody_hip_sharded_hash_step(None, 0.1, 1) == nb._lib.ERR_STATE
    assert lib.nbody_hip_sharded_hash_forces(None) == nb._lib.ERR_STATE
    assert lib.nbody_hip_sharded_hash_destroy(None) == 0
    with pytest.raises(nb.ValidationException):
        nb.sharded.shard_bounds(10, 2, 2)
    with pytest.raises(nb.DeviceException):
        nb.sharded.Comm.init_all(2)


# nbody_hip_slab_layer_owner: the host side of the slab cuts of the sharded spatial hash (the same fma the partition
# kernel evaluates): owner = cuts at or below the layer's centre; monotone, so a rank's layers are contiguous
def test_slab_layer_owner_is_the_count_of_cuts_below_the_layer_centre(nb):
    lib = nb._lib.load()
    cuts = np.array([-3.25, -3.25, 0.5, 7.0], np.float32)          # 5 ranks, one of them without a layer
    lo, cell = np.float32(-10.001), np.float32(0.75)
    owners = [lib.nbody_hip_slab_layer_owner(z, float(lo), float(cell), 5, cuts.ctypes.data) for z in range(40)]
    want = [int(np.sum(cuts <= np.float32(np.float64(np.float32(z) + np.float32(0.5)) * np.float64(cell) + np.float64(lo))))
            for z in range(40)]
    assert owners == want and owners == sorted(owners) and set(owners) == {0, 2, 3, 4}
    assert lib.nbody_hip_slab_layer_owner(3, 0.0, 1.0, 5, None) == -1
