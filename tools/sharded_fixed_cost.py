#!/usr/bin/env python3
"""Fixed cost of the sharded Direct step at the 8-GPU per-rank sizes, measured on ONE GPU.
The C-ABI host (csrc/sharded.hip) runs W = 8 VIRTUAL ranks on device 0 (transport P2P): every rank's launches --
drift, own x own at 131,072 bodies, its 3 1/2 shard pairs, the reaction exchange, the fixed-order sum + kick -- are
issued exactly as on 8 GPUs, only that they share one device, so wall time per step / 8 is the per-rank step
(kernels + launch gaps + event waits) that an 8-GPU run cannot beat, and (wall - sum of kernel times) is the
orchestration overhead.  Run under `rocprofv3 --kernel-trace --stats` for the kernel sums.
Usage: python tools/sharded_fixed_cost.py [W] [N] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd as nb  # noqa: E402
from nbody_amd.sharded import Comm, ShardedDirect  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
torch.cuda.set_device(0)
ic = nb.ic.plummer(n, seed=42)
comm = Comm.init_all(W, [0] * W)
s = ShardedDirect(comm, n, 1.0, 1e-3)
s.set_state(ic)
s.forces()
s.step(1e-3, 1)
s.synchronize()
t0 = time.perf_counter()
s.step(1e-3, steps)
t_issue = (time.perf_counter() - t0) / steps * 1e3      # host time to ISSUE a step of all W ranks
s.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
ev = s.time_steps(1e-3, 0, steps)
print(f"W={W} virtual ranks on one GPU, N={n} (shard {n // W}): wall {wall:.2f} ms per step of all ranks = {wall / W:.2f} ms per rank; "
      f"rank-0 stream events {ev:.2f} ms per step; host issue time {t_issue:.3f} ms per step ({t_issue / W * 1e3:.0f} us per rank)")
ke, pe = s.energies()
print(f"energies KE {ke:.6f} PE {pe:.6f}")
