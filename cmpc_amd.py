"""Importable alias for the package directory (whose mandated name contains '-').

``import cmpc_amd`` loads
``online-non-linear-centroidal-mpc-with-stability-guarantees-for-robust-locomotion-of-legged-robots-_amd/``
as the package ``cmpc_amd`` so that ``cmpc_amd.solver`` etc. resolve normally.
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(
    os.path.dirname(os.path.abspath(__file__)),
    "online-non-linear-centroidal-mpc-with-stability-guarantees-for-robust-locomotion-of-legged-robots-_amd")

_spec = importlib.util.spec_from_file_location(
    "cmpc_amd", os.path.join(_PKG_DIR, "__init__.py"),
    submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cmpc_amd"] = _mod
_spec.loader.exec_module(_mod)
