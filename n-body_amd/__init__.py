"""n-body_amd -- MI355X-native force/integration engine behind the LessUp/n-body plugin API.

The directory name contains a hyphen (it mirrors the reference's repository name), so the
package is imported under the module name `nbody_amd`; tests/conftest.py, bench.py and
__graft_entry__.py register it with `load_package()` from the repo-root helper `nbody_amd.py`.
"""
from . import _lib, ic  # noqa: F401
from ._lib import (DeviceException, NBodyError, ResourceException, StateException,  # noqa: F401
                   ValidationException)
from .api import *  # noqa: F401,F403
from .api import (BarnesHutCalculator, BarnesHutTree, Context, StepGraph, DirectForceCalculator, ForceCalculator, ForceMethod,  # noqa: F401
                  InitDistribution, Integrator, ParticleData, ParticleDataManager, ParticleInitializer,
                  DiskDistParams, SphericalDistParams, UniformDistParams,
                  SimulationConfig, SpatialHashCalculator, SpatialHashGrid,
                  createForceCalculator, default_context,
                  direct_forces_pair_packed, direct_forces_packed, pack_posm,
                  time_direct_packed)
from .system import (MAX_PARTICLE_COUNT, NBODY_MAGIC, NBODY_VERSION, ParticleSystem,  # noqa: F401,E402
                     Serializer, SimulationState)
from . import observability  # noqa: F401,E402
from . import sharded  # noqa: F401,E402
from ._lib import CommException  # noqa: F401,E402


def __getattr__(name):
    # `cli` and `benchmarks` are also `python -m` entry points: importing them here eagerly makes runpy warn
    # ("found in sys.modules after import of package"), so they are loaded on first use (nb.cli, nb.benchmarks)
    if name in ("cli", "benchmarks"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
