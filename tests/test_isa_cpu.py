"""ISA-level guards (no GPU: hipcc cross-compiles gfx950 here).  The latency-bound kernels of the decoder loop run with one or two
waves per SIMD, so a chain of dependent memory round trips is their whole cost: hipcc emits such chains (load / s_waitcnt vmcnt(0) /
load / ...) for `cond ? *p : 0`, for loads behind a wave-uniform `continue` and for selects on a loaded value.  tools/isa_waits.py
counts them per kernel; these bounds keep the rewritten kernels from sliding back (round 3: token chain 40 -> 2, flow-head conv2
21 -> 7 counted waits at the tail of one batch, CCL 36 -> 4, dwconv3x3 10 -> 1, convex upsampling 9 -> 1)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _summ(fname):
    spec = importlib.util.spec_from_file_location("isa_waits", os.path.join(ROOT, "tools", "isa_waits.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = mod.summarize(os.path.join(mod.CSRC, fname))
    # (file, kernel, vgpr, spills, scratch bytes, loads, vmcnt(0) waits, load -> vmcnt(0) chains)
    return {r[1].replace("void ", ""): r for r in rows}


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("c++filt") is None, reason="needs hipcc + c++filt")
def test_no_dependent_load_chains_in_the_decoder_loop_kernels():
    flow = _summ("flowops.hip")
    tc = next(v for k, v in flow.items() if k.startswith("decoder_token_chain_kernel"))
    assert tc[7] <= 4 and tc[3] == 0 and tc[4] == 0, tc            # two batches of loads, no spills, no stack
    assert flow["convex_upsample_kernel"][7] <= 2, flow["convex_upsample_kernel"]
    nn = _summ("nn.hip")
    assert nn["dwconv3x3_kernel"][7] <= 2, nn["dwconv3x3_kernel"]
    assert nn["ccl_softargmax_kernel"][7] <= 6, nn["ccl_softargmax_kernel"]
    gemm = _summ("gemm.hip")
    nc = gemm["narrow_conv3x3_kernel<2>"]
    assert nc[7] <= 8 and nc[3] == 0, nc
    for k, v in gemm.items():
        if k.startswith(("conv_gemm_dma_kernel", "conv_gemm_dma_pair_kernel", "rowstream_gemm_kernel")):
            assert v[3] == 0 and v[4] == 0, (k, v)                  # the MFMA kernels never spill
