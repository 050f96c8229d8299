"""The package directory is `ground-fusion2_amd/` (hyphen: not a Python identifier), so it is
registered under the importable name `ground_fusion2_amd` here. Used by tests/, bench.py and
__graft_entry__.py:  `from _gfbe_import import gf`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_NAME = "ground_fusion2_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    pkg_dir = os.path.join(_ROOT, "ground-fusion2_amd")
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


gf = load()
