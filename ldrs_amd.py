"""Importable alias for the package directory `learning-disentangled-representations-under-supervision_amd/`
(its name is not a Python identifier). `import ldrs_amd` gives that package."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_pkg = importlib.import_module("learning-disentangled-representations-under-supervision_amd")
sys.modules[__name__] = _pkg
