"""Process exit with live handles: an uncaught exception keeps the failing frame's objects -- a
Barnes-Hut tree with its side stream, a grid, a step graph, a sharded system and its communicator -- alive until the
interpreter finalises.  Their `__del__` used to call nbody_hip_*_destroy after the HIP runtime's own static destructors
had run: std::bad_variant_access out of hipStreamSynchronize, SIGABRT, rc 134 instead of the exception's rc 1.  Now ONE
atexit hook (nbody_amd._lib.close_all) closes them in dependency order while the runtime is up, `__del__` is a no-op once
the interpreter finalises, and the C side answers a dead runtime with NBODY_HIP_OK (csrc/common.h runtime_alive).
Destructor it mirrors: ref src/cuda/force_barnes_hut.cu:212-216 (frees while the CUDA runtime lives)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import numpy as np, torch
import nbody_amd as nb
from nbody_amd.sharded import Comm, ShardedDirect, ShardedHash
from gpu_util import to_device
torch.cuda.set_device(0)
ctx = nb.default_context(0)
n = 20000
ic = nb.ic.plummer(n, seed=3)
d, _ = to_device(nb, ic)
tree = nb.BarnesHutTree(n)
tree.build(d); tree.computeForces(d, 0.5, 1.0, 0.01)       # the walk's schedule goes to the tree's side stream
grid = nb.SpatialHashGrid(n, 0.5)
grid.build(d); grid.computeForces(d, 0.5, 1.0, 0.01)
fc = nb.DirectForceCalculator(); integ = nb.Integrator()
fc.computeForces(d)
with ctx.capture() as rec:
    integ.integrate(d, fc, 1e-3)
rec.graph.launch(2)
comm = Comm.init_all(1, [0])
sd = ShardedDirect(comm, n, 1.0, 1e-3); sd.set_state(ic); sd.forces(); sd.step(1e-3, 1)
sh = ShardedHash(comm, n, 1.0, 0.01, 0.5, 0.5); sh.set_state(ic); sh.forces(); sh.step(1e-3, 1)
mode = {mode!r}
if mode == "raise":
    def fail():
        keep = (tree, grid, rec, sd, sh, comm, ctx, d)      # the traceback keeps this frame (and the objects) alive
        raise RuntimeError("deliberate")
    fail()
elif mode == "cycle":
    class Box: pass
    b = Box(); b.self = b; b.t = tree; b.g = grid; b.s = sd; b.h = sh   # a reference cycle: collected at finalisation
    del tree, grid, sd, sh
    sys.exit(3)
print("clean")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("mode,rc", [("raise", 1), ("cycle", 3), ("clean", 0)])
def test_exit_code_with_live_handles(mode, rc, tmp_path):
    script = tmp_path / "child.py"
    script.write_text(CHILD.format(root=ROOT, tests=os.path.join(ROOT, "tests"), mode=mode))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == rc, f"exit code {r.returncode} (134 = abort at teardown):\n{tail}"
    assert "terminate called" not in r.stderr and "bad_variant_access" not in r.stderr, tail
    if mode == "raise":
        assert "RuntimeError: deliberate" in r.stderr


def test_close_all_orders_and_silences_del():
    """CPU: the registry closes in dependency order, survives close() raising, and empties itself."""
    import nbody_amd  # noqa: F401
    from nbody_amd import _lib
    order = []

    class H:
        def __init__(self, kind, boom=False):
            self.kind, self.boom = kind, boom
            _lib.track(self, kind)

        def close(self):
            order.append(self.kind)
            if self.boom:
                raise RuntimeError("close failed")
    saved = {k: list(v) for k, v in _lib._live.items()}
    for v in _lib._live.values():
        v.clear()
    try:
        objs = [H("context"), H("comm"), H("grid", boom=True), H("tree"), H("graph"), H("system"), H("backend")]
        _lib.close_all()
        assert order == ["system", "graph", "tree", "grid", "backend", "comm", "context"]
        assert all(len(v) == 0 for v in _lib._live.values())
        assert not _lib.finalizing()  # only the interpreter's own finalisation silences __del__
        del objs
    finally:
        for k, objs_k in saved.items():
            for o in objs_k:
                _lib._live[k].add(o)
