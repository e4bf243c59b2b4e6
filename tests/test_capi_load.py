"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol that include/stitch_gfx950.h declares (no compute calls here)."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import stitch_amd
    from stitch_amd import _lib
    decl = _lib.declared_functions()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(_lib.lib, name), name
    out = subprocess.check_output(["nm", "-D", _lib.LIB_PATH]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if " T st_" in l}
    assert exported == set(decl), exported ^ set(decl)
    assert stitch_amd.ops.lib.st_abi_gemm_desc_size() == ctypes.sizeof(_lib.GemmDesc)


def test_rejects_bad_arguments_without_touching_the_gpu():
    from stitch_amd._lib import lib
    assert lib.st_conv_gemm(None, None) == 1001
    assert lib.st_layernorm(None, 0, None, None, None, 0, 0, 0, 1e-5, None) == 1001
    assert lib.st_homo_warp(None, None, None, None, 0, 0, 0, 0, 0, 0, 0, None) == 1001
