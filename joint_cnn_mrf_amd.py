"""Import alias: the package directory is `joint-cnn-mrf_amd/` (a hyphen is not a valid
Python identifier), so `import joint_cnn_mrf_amd` loads that directory as a package."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'joint-cnn-mrf_amd')
_spec = importlib.util.spec_from_file_location(
    'joint_cnn_mrf_amd', os.path.join(_pkg_dir, '__init__.py'),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['joint_cnn_mrf_amd'] = _mod
_spec.loader.exec_module(_mod)
