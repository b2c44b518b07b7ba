"""Import shim: `import vln_amd` -> the package in `curriculum-learning-for-vln_amd/`.

Every submodule is aliased too (`vln_amd.functional` IS `curriculum-learning-for-vln_amd.functional`): without the aliases
`from vln_amd.decoders import X` would execute decoders.py -- and everything it imports relatively -- a SECOND time under the
alias name, and module-level switches (functional.set_grad_in_place, ...) would exist twice."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_real = "curriculum-learning-for-vln_amd"
_pkg = importlib.import_module(_real)
sys.modules[__name__] = _pkg
for _k in [k for k in sys.modules if k.startswith(_real + ".")]:
    sys.modules.setdefault(__name__ + _k[len(_real):], sys.modules[_k])
