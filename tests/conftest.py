import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def nb():
    import nbody_amd  # registers the hyphenated package directory as `nbody_amd`
    return nbody_amd


@pytest.fixture(scope="session")
def oracle():
    import oracle_bind
    return oracle_bind.load()


@pytest.fixture(scope="session")
def ctx(nb):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    torch.cuda.set_device(0)
    return nb.default_context(0)


@pytest.fixture(autouse=True)
def _restore_session_context(request):
    """The context of the -m gpu tests is session-wide (nb.default_context is cached): a test that changes its
    determinism mode or its tuning overrides must not leak that into the tests after it.  Snapshot before,
    restore after (the tuning overrides have no getter: they are reset to automatic)."""
    if "ctx" not in request.fixturenames:
        yield
        return
    c = request.getfixturevalue("ctx")
    mode = c.directInfo()["deterministic_mode"]
    yield
    c.tuning()
    if c.directInfo()["deterministic_mode"] != mode:
        c.deterministic(mode)
