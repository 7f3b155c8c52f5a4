#!/usr/bin/env python3
"""Symmetric direct kernel at N = 2^20: equal / general masses x deterministic slots on / off x bodies per lane."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb  # noqa: E402
from gpu_util import packed  # noqa: E402

torch.cuda.set_device(0)
ctx = nb.default_context(0)
n = 1 << 20
ic = nb.ic.plummer(n, seed=42)
p = packed(ic)
rng = np.random.default_rng(7)
pg = p.clone()
pg[:, 3] *= torch.from_numpy((0.75 + 0.5 * rng.random(n)).astype(np.float32)).cuda()
tpls = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else (16, 8)
splits = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for tpl in tpls:
    ctx.tuning(3, tpl, splits)
    for det in (True, False):
        ctx.deterministic(det)
        a = nb.time_direct_packed(ctx, p, p, 1.0, 1e-6, 3)
        b = nb.time_direct_packed(ctx, pg, pg, 1.0, 1e-6, 3)
        print(f"R={tpl} deterministic={det}: equal mass {a:.2f} ms, general mass {b:.2f} ms", flush=True)
