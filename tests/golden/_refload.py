"""Load a module of the reference by FILE PATH under an alias (the repo has its own `tools` package, which
would shadow the reference's namespace package on sys.path).  Build container only."""
import importlib.util
import sys
import types

REF = "/root/reference"


def load_reference_module(relpath: str, alias: str, stubs=("torchvision", "hdf5storage")):
    sys.dont_write_bytecode = True
    for s in stubs:
        sys.modules.setdefault(s, types.ModuleType(s))
    spec = importlib.util.spec_from_file_location(alias, f"{REF}/{relpath}")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
