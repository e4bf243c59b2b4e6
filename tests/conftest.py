import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the HIP library is a build artefact (git-ignored): compile it in-tree if a fresh checkout lacks it
_PKG = os.path.join(ROOT, "seamless-through-breaking-rethinking-image-stitching-for-optimal-alignment_amd")
if not os.path.exists(os.path.join(_PKG, "libstitch_gfx950.so")):
    import importlib.util
    _spec = importlib.util.spec_from_file_location("_stitch_build", os.path.join(_PKG, "build.py"))
    _mod = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(_mod)
    _mod.build()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_ops():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "ops_small.npz"))


@pytest.fixture(scope="session")
def seeded_sd():
    from oracle import spec
    return spec.seeded_state_dict(1234)
