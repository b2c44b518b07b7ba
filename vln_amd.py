"""Import shim: `import vln_amd` -> the package in `curriculum-learning-for-vln_amd/`."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("curriculum-learning-for-vln_amd")
sys.modules[__name__] = _pkg
