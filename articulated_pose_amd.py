"""Import shim: `import articulated_pose_amd` -> the package directory `articulated-pose_amd/`
(a hyphen is not importable as a Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "articulated-pose_amd")
_spec = importlib.util.spec_from_file_location(
    "articulated_pose_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["articulated_pose_amd"] = _mod
_spec.loader.exec_module(_mod)
