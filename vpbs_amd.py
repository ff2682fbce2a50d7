"""Loader: `import vpbs_amd` gives the package that lives in ./verifiable-fhe-paper_amd/ (hyphenated directory)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "verifiable-fhe-paper_amd")
_spec = importlib.util.spec_from_file_location("vpbs_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vpbs_amd"] = _mod
_spec.loader.exec_module(_mod)
