"""Import shim: the product package lives in ``feature-point-cnn_amd/`` (the
directory name the build contract fixes; a hyphen is not importable), so this
module loads it under the importable name ``fpc_amd``.

    import fpc_amd                      # package  feature-point-cnn_amd/
    from fpc_amd import synth, arch     # its sub-modules
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "feature-point-cnn_amd")


def _load():
    spec = importlib.util.spec_from_file_location(
        "fpc_amd", os.path.join(_PKG_DIR, "__init__.py"),
        submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["fpc_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


_load()
