import os, sys
import numpy as np, torch
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb
from gpu_util import to_device
n = 1 << 20
ic = nb.ic.two_galaxies(n, seed=42)
d, _ = to_device(nb, ic)
tree = nb.BarnesHutTree(n); tree.build(d)
nodes = tree.copyNodesToHost()
ch = nodes["children"]
internal = ~nodes["is_leaf"]
cn = (ch >= 0).sum(1)[internal]
c0 = np.where(ch >= 0, ch, 1 << 30).min(1)[internal]
steps_now = ((c0 + cn - 1) // 2 - c0 // 2 + 1)
steps_al = (cn + 1) // 2
print("internal nodes", internal.sum(), "mean children", cn.mean(), "pair-steps now", steps_now.mean(), "aligned", steps_al.mean(), "saving", 1 - steps_al.sum() / steps_now.sum())
print("hist cn", np.bincount(cn, minlength=9))
print("odd groups", (cn % 2 == 1).mean())
