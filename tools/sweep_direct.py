#!/usr/bin/env python3
"""Times nbh::direct_kernel for every (variant, targets-per-lane, splits) on one GPU.
Usage: python tools/sweep_direct.py [N ...]   -> one line per configuration, best first."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd as nb  # noqa: E402


def main():
    sizes = [int(s) for s in sys.argv[1:]] or [262144, 1 << 20]
    torch.cuda.set_device(0)
    ctx = nb.default_context(0)
    for n in sizes:
        ic = nb.ic.plummer(n, seed=42)
        p = torch.from_numpy(np.ascontiguousarray(
            np.stack([ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]], 1))).cuda()
        rows = []
        for variant in (0, 1, 2):
            for tpl in (1, 2, 4):
                if variant == 1 and tpl == 1:
                    continue
                for splits in (0, 1, 2, 4):
                    ctx.tuning(variant, tpl, splits)
                    iters = 3 if n <= 300000 else 2
                    ms = nb.time_direct_packed(ctx, p, p, 1.0, 1e-6, iters)
                    rate = float(n) * n / (ms * 1e-3)
                    rows.append((rate, variant, tpl, splits, ms))
                    print(f"N={n} variant={variant} tpl={tpl} splits={splits} ms={ms:.3f} "
                          f"pairs/s={rate:.3e} TFLOP/s(20)={rate * 20 / 1e12:.1f}", flush=True)
        ctx.tuning()
        rows.sort(reverse=True)
        print(f"BEST N={n}: variant={rows[0][1]} tpl={rows[0][2]} splits={rows[0][3]} "
              f"ms={rows[0][4]:.3f} pairs/s={rows[0][0]:.3e}", flush=True)


if __name__ == "__main__":
    main()
