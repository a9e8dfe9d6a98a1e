"""
Import shim: the package lives in the directory ``ipp-rl_amd/`` (not a valid Python identifier), so
``import ipp_rl_amd`` resolves to this file, which loads that directory as the package ``ipp_rl_amd``.
"""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ipp-rl_amd")
_spec = importlib.util.spec_from_file_location(
    "ipp_rl_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ipp_rl_amd"] = _mod
_spec.loader.exec_module(_mod)
