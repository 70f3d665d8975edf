"""Import alias for the package directory `control-gic_amd/`.

The repository layout names the package with a hyphen, which Python cannot
import directly; importing `control_gic_amd` loads that directory as a regular
package under this (valid) name, submodules included.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "control-gic_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
