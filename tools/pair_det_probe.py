#!/usr/bin/env python3
"""Two-set (shard pair) Direct kernel: deterministic slots vs fp64 atomics x bodies per lane, at the 8-GPU shard size.
Usage: python tools/pair_det_probe.py [shard bodies]"""
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
S = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
p = packed(nb.ic.plummer(2 * S, seed=42))
pg = p.clone()
pg[:, 3] *= torch.from_numpy((0.75 + 0.5 * np.random.default_rng(7).random(2 * S)).astype(np.float32)).cuda()
for name, q in (("equal", p), ("general", pg)):
    a, b = q[:S].contiguous(), q[S:].contiguous()
    acc_a, acc_b = torch.zeros_like(a), torch.zeros_like(b)
    for tpl, det, splits in ((8, 1, 0), (8, 0, 0), (16, 1, 0), (16, 0, 0), (16, 1, 96), (16, 1, 64), (16, 1, 48), (16, 1, 32), (16, 0, 64), (8, 1, 64), (8, 1, 32)):
        if True:
            ctx.tuning(-1, tpl, splits)
            ctx.deterministic(det)
            nb.direct_forces_pair_packed(ctx, a, b, 1.0, 1e-6, acc_a, acc_b)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                nb.direct_forces_pair_packed(ctx, a, b, 1.0, 1e-6, acc_a, acc_b)
            e1.record()
            e1.synchronize()
            print(f"pair {S} x {S} {name} masses R={tpl} deterministic={det} splits={splits or 'auto'}: {e0.elapsed_time(e1) / 5:.3f} ms", flush=True)
ctx.tuning()
ctx.deterministic(1)
