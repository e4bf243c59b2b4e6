"""Import alias for the package directory ``seamless-through-breaking-rethinking-image-stitching-for-optimal-alignment_amd``
(its name is not a Python identifier): ``import stitch_amd`` loads it under the name ``stitch_amd``."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "seamless-through-breaking-rethinking-image-stitching-for-optimal-alignment_amd")
_spec = importlib.util.spec_from_file_location("stitch_amd", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["stitch_amd"] = _mod
_spec.loader.exec_module(_mod)
