"""Import shim: the package directory is named after the reference repo
(`3d-super-resolution-face-reconstruction_amd`, not a valid identifier), so `import sr3_amd`
loads it through importlib and aliases it."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("3d-super-resolution-face-reconstruction_amd")
sys.modules[__name__] = _pkg
