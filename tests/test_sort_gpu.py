"""The radix sort that bins the bodies (ref: thrust::sort_by_key, src/cuda/force_barnes_hut.cu:276-280,
force_spatial_hash.cu:286-288).  Above the crossover sizes the library drives rocPRIM's Onesweep kernels itself
(csrc/onesweep.h: rocprim::detail, a private namespace) -- fenced at compile time to the rocPRIM version it was written
against and at run time by a self-test against the public rocprim::radix_sort_pairs.  Here: the fence's state, and that
trees and grids built on BOTH sides of the fence (NBH_OWN_SORT_FROM moves the crossover) are the same structures bit for
bit -- i.e. the two sorts give the same permutation on real keys."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from gpu_util import acc_of, to_device

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sort_info(nb):
    lib = nb._lib.load()
    a, b, o, c = C.c_int(-1), C.c_int(-1), C.c_int(-1), C.c_int(-1)
    assert lib.nbody_hip_sort_info(C.byref(a), C.byref(b), C.byref(o), C.byref(c)) == 0
    return a.value, b.value, o.value, c.value


def test_sort_info_reports_the_fence(nb):
    compiled, state, own_state, version = _sort_info(nb)
    assert compiled in (0, 1) and state in (0, 1, 2) and own_state in (0, 1, 2) and version > 0
    if version != 400200:          # another rocPRIM than the driver was written against: compiled out
        assert compiled == 0


CHILD = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import numpy as np, torch
import nbody_amd as nb
from gpu_util import acc_of, to_device
torch.cuda.set_device(0)
n = 300000
ic = nb.ic.two_galaxies(n, seed=11)
d, _ = to_device(nb, ic)
tree = nb.BarnesHutTree(n)
tree.build(d); tree.computeForces(d, 0.5, 1.0, 0.05)
a_tree = acc_of(d).copy()
nodes = tree.getNodeCount()
box = nb.ic.uniform_box(n, seed=12, lo=-13.0, hi=13.0)
d2, _ = to_device(nb, box)
grid = nb.SpatialHashGrid(n, 1.0)
grid.build(d2); grid.computeForces(d2, 1.0, 1.0, 0.05)
cs, ce, pc, si = grid.copyCellDataToHost()
np.savez({out!r}, a_tree=a_tree, nodes=nodes, a_grid=acc_of(d2), cs=cs, ce=ce, si=si)
import ctypes as C
a, b, o, c = C.c_int(), C.c_int(), C.c_int(), C.c_int()
nb._lib.load().nbody_hip_sort_info(C.byref(a), C.byref(b), C.byref(o), C.byref(c))
print("SORT", a.value, b.value, o.value, c.value)
"""


@pytest.mark.gpu
def test_both_sides_of_the_fence_give_the_same_tree_and_grid(tmp_path):
    """300,000 bodies: above both crossovers with the default (the Onesweep driver sorts) and with NBH_SORT=own (the
    hand-written sort of csrc/radix_sort.h), below them with NBH_OWN_SORT_FROM = 10^9 (the public sort does): node count,
    accelerations of every body from the tree, the grid's cell ranges and sorted index list and its accelerations must be
    identical."""
    res = {}
    for name, env_from, which in (("driver", "0", "driver"), ("public", "1000000000", "public"), ("own", "0", "own")):
        out = tmp_path / f"{name}.npz"
        script = tmp_path / f"{name}.py"
        script.write_text(CHILD.format(root=ROOT, tests=os.path.join(ROOT, "tests"), out=str(out)))
        env = dict(os.environ, NBH_OWN_SORT_FROM=env_from, NBH_SORT=which)
        r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("SORT")][-1].split()
        res[name] = (np.load(out), [int(v) for v in line[1:]], r.stderr)
    (dr, info_d, err_d), (pu, info_p, _), (ow, info_o, err_o) = res["driver"], res["public"], res["own"]
    if info_d[0] == 1:
        # the driver is compiled in: its self-tests ran when the tree and the grid were made, and passed
        assert info_d[1] == 1, (info_d, err_d[-500:])
    assert info_o[2] == 1, (info_o, err_o[-500:])          # the hand-written sort's self-tests (tree and grid) passed
    assert "does not reproduce" not in err_d and "does not reproduce" not in err_o
    for k in ("nodes", "a_tree", "cs", "ce", "si", "a_grid"):
        assert np.array_equal(dr[k], pu[k]), k
        assert np.array_equal(ow[k], pu[k]), k              # NBH_SORT=own: csrc/radix_sort.h, no rocPRIM in the sort
