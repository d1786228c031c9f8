"""Import helper: the package directory is named `incrementalinference.jl_amd` (contains a dot,
so it cannot be named in an `import` statement).  `load()` registers it as module `iif_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(_ROOT, "incrementalinference.jl_amd")


def load():
    if "iif_amd" in sys.modules:
        return sys.modules["iif_amd"]
    spec = importlib.util.spec_from_file_location(
        "iif_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["iif_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
