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


def test_corr_volume_both_dispatch_decision():
    """st_corr_volume_both must take its two-product fallback wherever the transposed second store cannot run --
    including volumes beyond the 32-bit buffer offsets (N * N * 4 >= 2^31, flow grids past ~1 200 x 1 200 px / 8), where the
    transposed copy and the row-chunked GEMM path exclude each other -- instead of returning ST_EINVAL."""
    from stitch_amd._lib import lib
    plan = lib.st_corr_volume_both_plan
    assert plan(1, 4096, 256, 1) == 1 and plan(8, 4096, 256, 1) == 1          # the 512 x 512 path
    assert plan(1, 4096, 256, 0) == 0                                         # misaligned operand
    assert plan(1, 4094, 256, 1) == 0 and plan(1, 4096, 96, 1) == 0 and plan(1, 4096, 144, 1) == 0      # N % 4, C < 128, C % 32
    assert plan(1, 23172, 256, 1) == 0 and plan(1, 23168, 256, 1) == 0        # N^2 * 4 >= 2^31 / (N + 256) * N * 4 >= 2^31 (row chunking)
    assert plan(1, 23044, 256, 1) == 0 and plan(1, 23040, 256, 1) == 1        # the boundary: (N + 256) * N * 4 < 2^31 <=> N <= 23042
    assert plan(0, 4096, 256, 1) == 0


def test_chain_and_mlp_descriptors_are_validated_on_the_host():
    """st_linear_chain128 rejects a chain that needs two different saved layer inputs (the kernel keeps one copy) and a
    bias that is not 16-byte aligned (read with 128-bit loads); st_mlp128 rejects in-place use and bad shapes.  Host checks only:
    the pointers are never dereferenced."""
    import ctypes as C
    from stitch_amd._lib import ChainDesc, MlpDesc, lib
    assert lib.st_abi_mlp_desc_size() == C.sizeof(MlpDesc) and lib.st_abi_chain_desc_size() == C.sizeof(ChainDesc)
    base = 0x7f0000000000

    def chain(res_layers, bias_off=0):
        d = ChainDesc()
        d.a, d.out, d.lda, d.ldo, d.M, d.nlayers = base, base + (1 << 30), 128, 128, 64, 3
        for i in range(3):
            ly = d.layer[i]
            ly.w, ly.bias = base + (2 << 30) + i * 65536, base + (3 << 30) + i * 512 + (bias_off if i == 1 else 0)
            if res_layers[i] is not None:
                ly.res, ly.res_layer = 2, res_layers[i]
        return d
    assert lib.st_linear_chain128(C.byref(chain([None, 1, 0])), None) == 1001      # x_1 and x_0 both needed: one copy only
    assert lib.st_linear_chain128(C.byref(chain([None, None, 3])), None) == 1001   # res_layer beyond the layer itself
    assert lib.st_linear_chain128(C.byref(chain([None, None, None], bias_off=4)), None) == 1001
    m = MlpDesc()
    m.a = m.out = base
    m.w1, m.b1, m.w2, m.b2, m.lda, m.ldo, m.M, m.hidden = base + 4096, base + 8192, base + 12288, base + 16384, 128, 128, 64, 512
    assert lib.st_mlp128(C.byref(m), None) == 1001                                 # in place
    m.out = base + (1 << 30)
    m.hidden = 500
    assert lib.st_mlp128(C.byref(m), None) == 1001                                 # hidden % 32
    m.hidden, m.b1 = 512, base + 8196
    assert lib.st_mlp128(C.byref(m), None) == 1001                                 # misaligned bias
