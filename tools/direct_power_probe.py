#!/usr/bin/env python3
"""Direct N^2 kernel at N = 2^20 with 8 / 12 / 16 bodies per lane, each run for ~3 s with the engine clock / package power (sysfs) sampled beside it:
time per launch, engine clock and package power -- is one of the shapes cheaper in energy (higher sustained clock on a
power-limited box)?   Usage: python tools/direct_power_probe.py"""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb  # noqa: E402
from gpu_util import packed  # noqa: E402


from bench import ClockSampler  # noqa: E402  (sysfs reader: a process that holds the GPU starts no rocm-smi child)

_clock = ClockSampler(0)


def smi():
    r = _clock.read() if _clock.src else None
    return {"sclk": r["sclk_mhz"], "power": r.get("power_w", 0.0)} if r else {}


torch.cuda.set_device(0)
ctx = nb.default_context(0)
n = 1 << 20
import numpy as np  # noqa: E402
p_eq = packed(nb.ic.plummer(n, seed=42))
p_gen = p_eq.clone()
p_gen[:, 3] *= torch.from_numpy((0.75 + 0.5 * np.random.default_rng(7).random(n)).astype(np.float32)).cuda()
order = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 12, 16, 12, 16, 12]
for k, tpl in enumerate(order * 2):
    p = p_eq if k < len(order) else p_gen
    if k == len(order):
        print("general masses:")
    ctx.tuning(3, tpl, 0)
    nb.time_direct_packed(ctx, p, p, 1.0, 1e-6, 2)
    samples, stop = [], threading.Event()

    def run():
        while not stop.is_set():
            r = smi()
            if r.get("power", 0) > 600:
                samples.append(r)
            stop.wait(0.1)
    t = threading.Thread(target=run, daemon=True)
    t.start()
    ms = nb.time_direct_packed(ctx, p, p, 1.0, 1e-6, 20)
    stop.set()
    t.join()
    clk = sorted(s["sclk"] for s in samples) or [0]
    pw = [s["power"] for s in samples] or [0]
    print(f"R={tpl:2d}: {ms:7.2f} ms per launch, sclk median {clk[len(clk) // 2]:.0f} MHz, power mean {sum(pw) / len(pw):.0f} W "
          f"({len(samples)} samples) -> {ms * clk[len(clk) // 2] / 1e3:.1f} Mcycles per launch", flush=True)
