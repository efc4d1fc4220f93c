"""Importable alias of the ``mri-raytracer_amd`` package (a hyphen cannot be written in an
``import`` statement).  ``import mrirt`` / ``from mrirt.camera import OrbitalCamera`` resolve
to the very same module objects as ``importlib.import_module("mri-raytracer_amd")``."""
import importlib
import os
import sys

_REAL = "mri-raytracer_amd"
_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_REAL + "."):
        sys.modules[__name__ + _name[len(_REAL):]] = _mod
sys.modules[__name__] = _pkg
