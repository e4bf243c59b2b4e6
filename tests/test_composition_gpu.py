"""the UDIS2 composition stage on the HIP kernels against the CPU oracle and the golden vectors the
reference produced (tests/golden/composition_512x544.npz, oracle/ref_harness/make_composition_golden.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import composition as oc  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def net():
    import stitch_amd
    n = stitch_amd.composition.Network()
    n.load_state_dict(oc.seeded_state_dict(4321), strict=True)
    return n.cuda().eval()


@pytest.mark.parametrize("dil,C,Co", [(2, 32, 64), (3, 64, 24), (5, 256, 130), (4, 4, 32)])
def test_dilated_conv(dil, C, Co):
    """st_conv_gemm with dh/dw (register-staged, DMA-pipelined and 128x32 tiles) vs F.conv2d(padding=1, dilation=d)."""
    import stitch_amd
    ops = stitch_amd.ops
    g = torch.Generator().manual_seed(dil)
    B, H, W = 2, 30, 37
    x = torch.randn(B, C, H, W, generator=g)
    w, b = torch.randn(Co, C, 3, 3, generator=g) / (9 * C) ** 0.5, torch.randn(Co, generator=g)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1, dilation=dil))
    Ho, Wo = ref.shape[2:]
    xr = x.permute(0, 2, 3, 1).reshape(-1, C).cuda().contiguous()
    wp = w.permute(0, 2, 3, 1).reshape(Co, -1).cuda().contiguous()
    out = torch.empty(B * Ho * Wo, Co, device="cuda")
    ops.conv_gemm(xr, wp, out, geom=(B, H, W, 3, 3, 1, 1, 1, 1), dil=(dil, dil), bias=b.cuda(), act="relu")
    got = out.cpu().view(B, Ho, Wo, Co).permute(0, 3, 1, 2).double()
    assert (got - ref).abs().max() < 3e-5


def test_nearest_and_sub_rows():
    import stitch_amd
    ops = stitch_amd.ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 7, 9, generator=g)
    ref = F.interpolate(x, size=(47, 61), mode="nearest")
    xr = x.permute(0, 2, 3, 1).reshape(-1, 8).cuda().contiguous()
    out = torch.empty(2 * 47 * 61, 8, device="cuda")
    ops.resize_nearest_rows(xr, out, 2, 7, 9, 8, 47, 61)
    assert torch.equal(out.cpu().view(2, 47, 61, 8).permute(0, 3, 1, 2), ref)
    a, b = torch.randn(100, 16, generator=g), torch.randn(100, 24, generator=g)
    o = torch.zeros(100, 20, device="cuda")
    ops.sub_rows(a.cuda(), b.cuda()[:, 4:20], o[:, 2:18])
    assert torch.equal(o[:, 2:18].cpu(), a - b[:, 4:20]) and o[:, :2].abs().sum() == 0


def test_composition_vs_reference_golden(net):
    import stitch_amd
    g = np.load(os.path.join(GOLDEN, "composition_512x544.npz"))
    assert list(net.state_dict().keys()) == list(g["keys"])
    o1, o2, m1, m2 = oc.synthetic_inputs(512, 544, 77)
    assert np.allclose(g["in_checksum"], [float(o1.double().sum()), float(o2.double().sum()), float(m1.sum()), float(m2.sum())])
    out = stitch_amd.composition.compose(net, o1, o2, m1, m2)
    assert set(out) == {"learned_mask1", "learned_mask2", "stitched_image"}
    w1, w2 = oc.preprocess(o1, False).cuda(), oc.preprocess(o2, False).cuda()
    mask = net(w1, w2, m1.cuda(), m2.cuda())
    assert tuple(mask.shape) == (1, 1, 512, 544)
    assert np.abs(mask[0, 0, ::2, ::2].cpu().numpy() - g["net_out_sub"]).max() < 2e-4          # sigmoid output, 18 conv layers
    assert np.abs(out["stitched_image"][0, :, ::4, ::4].cpu().numpy() - g["stitched_sub"]).max() < 5e-4
    assert np.abs(out["learned_mask1"][0, :, ::4, ::4].cpu().numpy() - g["lm1_sub"]).max() < 2e-4
    assert np.abs(out["learned_mask2"][0, :, ::4, ::4].cpu().numpy() - g["lm2_sub"]).max() < 2e-4


def test_composition_small_canvas_is_upscaled_like_out_py(net):
    """canvas below 512 px: out.py:278-283 scales it up bilinearly (scale_factor form, align_corners=False)."""
    import stitch_amd
    o1, o2, m1, m2 = oc.synthetic_inputs(300, 340, 5)
    ref = oc.compose(oc.seeded_state_dict(4321), o1, o2, m1, m2)
    got = stitch_amd.composition.compose(net, o1, o2, m1, m2)
    assert got["stitched_image"].shape == ref["stitched_image"].shape
    assert (got["stitched_image"].cpu() - ref["stitched_image"]).abs().max() < 1e-3
    assert (got["learned_mask1"].cpu() - ref["learned_mask1"]).abs().max() < 5e-4
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 512, 512), torch.zeros(1, 3, 512, 512))         # CPU tensors: loud failure, no fallback
