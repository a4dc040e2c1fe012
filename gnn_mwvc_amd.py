"""Import shim: the package directory is named `gnn-mwvc_amd/` (not a valid
Python identifier), so `import gnn_mwvc_amd` loads it from here."""
import importlib.util
import pathlib
import sys

_pkg = pathlib.Path(__file__).resolve().parent / "gnn-mwvc_amd"
_spec = importlib.util.spec_from_file_location(
    "gnn_mwvc_amd", _pkg / "__init__.py", submodule_search_locations=[str(_pkg)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gnn_mwvc_amd"] = _mod
_spec.loader.exec_module(_mod)
