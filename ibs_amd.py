"""Import shim: `import ibs_amd` loads the package in ./ideal-ballooning-solver_amd/ (whose
directory name, fixed by the repository layout, is not a Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ideal-ballooning-solver_amd")
_spec = importlib.util.spec_from_file_location("ibs_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ibs_amd"] = _mod
_spec.loader.exec_module(_mod)
