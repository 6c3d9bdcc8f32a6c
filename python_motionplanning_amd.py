"""Importable alias of the hyphen-named package directory
``python-motionplanning_amd/`` (a hyphen is not valid in an ``import``
statement).  ``import python_motionplanning_amd as vd`` gives the package."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("python-motionplanning_amd")
