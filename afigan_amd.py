"""Import shim: the package directory is ``afi-gan_amd/`` (not a valid Python identifier), so this one-file
module loads it under the importable name ``afigan_amd`` and replaces itself in sys.modules."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "afi-gan_amd")
_spec = importlib.util.spec_from_file_location("afigan_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["afigan_amd"] = _mod
_spec.loader.exec_module(_mod)
