"""Import alias: the package directory required by the project layout,
``nested-u-net-based-real-time-speech-enhancement-mobile-app_amd/``, is not a
valid Python identifier, so it is loaded here under the name ``nunet_amd``.

    import nunet_amd                  # the package
    from nunet_amd import NutlsEngine, NutlsRunner
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "nested-u-net-based-real-time-speech-enhancement-mobile-app_amd")
_spec = importlib.util.spec_from_file_location(
    "nunet_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["nunet_amd"] = _mod
_spec.loader.exec_module(_mod)
