"""Import alias for the package directory `augmentedgaussianprocesses.jl_amd/` (its name contains a dot, so it
cannot be imported by name).  `import agp_amd as AGP` loads that package under the module name `agp_amd`."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "augmentedgaussianprocesses.jl_amd")
_spec = _ilu.spec_from_file_location("agp_amd", _os.path.join(_pkg_dir, "__init__.py"),
                                     submodule_search_locations=[_pkg_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["agp_amd"] = _mod
_spec.loader.exec_module(_mod)
