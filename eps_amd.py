"""Import shim: makes the hyphenated package directory ``edge-proposal-sets_amd/`` importable
as ``eps_amd`` (``import eps_amd``; ``from eps_amd import models``)."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "edge-proposal-sets_amd")
_spec = importlib.util.spec_from_file_location("eps_amd", os.path.join(_PKG_DIR, "__init__.py"),
                                               submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["eps_amd"] = _mod
_spec.loader.exec_module(_mod)
