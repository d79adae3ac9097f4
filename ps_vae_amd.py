"""Import shim: the package directory is named `ps-vae_amd` (not a valid Python
identifier), so `import ps_vae_amd` loads that directory as the package `ps_vae_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ps-vae_amd')
_spec = importlib.util.spec_from_file_location(
    'ps_vae_amd', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules['ps_vae_amd'] = _pkg
_spec.loader.exec_module(_pkg)
