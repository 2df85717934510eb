"""Importable alias of the package directory `graph-unet-traffic-prediction_amd/` (its name is
not a Python identifier): `import gunet_amd` loads that directory as the package `gunet_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'graph-unet-traffic-prediction_amd')
_spec = importlib.util.spec_from_file_location(
    'gunet_amd', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_module = importlib.util.module_from_spec(_spec)
sys.modules['gunet_amd'] = _module
_spec.loader.exec_module(_module)
