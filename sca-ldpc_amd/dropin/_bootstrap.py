"""Locate the hyphen-named package `sca-ldpc_amd` for the drop-in modules."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)


def sub(name=""):
    return importlib.import_module("sca-ldpc_amd" + ("." + name if name else ""))
