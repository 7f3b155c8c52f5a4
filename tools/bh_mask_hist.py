#!/usr/bin/env python3
"""Barnes-Hut walk: how many lanes of a wave take part in / accept each node test
(nbody_hip_tree_visit_histogram).  Usage: python tools/bh_mask_hist.py [two_galaxies|plummer] [N]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb  # noqa: E402
from gpu_util import to_device  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "two_galaxies"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
torch.cuda.set_device(0)
ic = getattr(nb.ic, which)(n, seed=42)
d, _ = to_device(nb, ic)
tree = nb.BarnesHutTree(n)
tree.build(d)
tree.countVisits(True)
tree.computeForces(d, 0.5, 1.0, 0.1)
out = (C.c_ulonglong * 130)()
nb._lib.check(tree.ctx._lib.nbody_hip_tree_visit_histogram(tree._h, C.byref(out)))
h = np.array(list(out), dtype=np.float64)
test, acc = h[:65], h[65:]
tot = test.sum()
waves = (n + 63) // 64
print(f"{which} N={n}: {tot / waves:.0f} internal-node tests per wave")
print("lanes testing  share of tests  cumulative   | lanes accepting  share")
cum = 0.0
for lo, hi in ((0, 0), (1, 4), (5, 8), (9, 16), (17, 32), (33, 48), (49, 63), (64, 64)):
    t = test[lo:hi + 1].sum() / tot
    a = acc[lo:hi + 1].sum() / tot
    cum += t
    print(f"  {lo:2d}..{hi:2d}        {t:6.3f}         {cum:6.3f}     |   {lo:2d}..{hi:2d}          {a:6.3f}")
k = np.arange(65)
print(f"mean lanes testing {(test * k).sum() / tot:.1f}, mean lanes accepting {(acc * k).sum() / tot:.1f}")
