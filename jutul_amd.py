"""Import shim: the product package lives in the directory `jutul.jl_amd/` (a name Python's import statement
cannot spell), so `import jutul_amd` loads that directory as the package `jutul_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jutul.jl_amd")
_spec = importlib.util.spec_from_file_location("jutul_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["jutul_amd"] = _mod
_spec.loader.exec_module(_mod)
