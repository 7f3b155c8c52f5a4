"""Import shim: makes the package in `n-body_amd/` (hyphenated, not a legal module name)
importable as `nbody_amd`.  `import nbody_amd` from the repo root is enough."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "n-body_amd")
_spec = _u.spec_from_file_location("nbody_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["nbody_amd"] = _mod
_spec.loader.exec_module(_mod)
