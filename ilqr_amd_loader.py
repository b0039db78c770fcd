"""Import shim: the package directory is named `iterativelqr.jl_amd` (a dot is
not importable), so load it under the module name `iterativelqr_jl_amd`."""
import importlib.util
import os
import sys

_NAME = "iterativelqr_jl_amd"


def load_package():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    root = os.path.dirname(os.path.abspath(__file__))
    pkg_dir = os.path.join(root, "iterativelqr.jl_amd")
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
