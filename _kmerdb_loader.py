"""Import helper: the package directory is named ``kmer-db_amd`` (the reference's name + _amd),
which is not a valid Python identifier, so it is registered under the module name
``kmerdb_amd``."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "kmer-db_amd")


def import_kmerdb_amd():
    if "kmerdb_amd" in sys.modules:
        return sys.modules["kmerdb_amd"]
    spec = importlib.util.spec_from_file_location(
        "kmerdb_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["kmerdb_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
