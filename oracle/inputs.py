"""Synthetic input pairs for the oracle-side tests (test infrastructure).

The generator itself is product-side data code (``<pkg>/data.py``, also used by bench.py); it is loaded here by
path so that the CPU oracle tests do not need the HIP library that importing the package requires."""
import importlib.util
import os

_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                     "seamless-through-breaking-rethinking-image-stitching-for-optimal-alignment_amd", "data.py")
_spec = importlib.util.spec_from_file_location("_stitch_data", _PATH)
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
structured_pair, noise_pair = _mod.structured_pair, _mod.noise_pair
