"""Import helper: the package directory is `efficient-nerf_amd/` (hyphen, not an
identifier), so register it as the module `efficient_nerf_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, 'efficient-nerf_amd')
NAME = 'efficient_nerf_amd'


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    spec = importlib.util.spec_from_file_location(NAME, os.path.join(PKG_DIR, '__init__.py'),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        sys.modules.pop(NAME, None)
        raise
    return mod
