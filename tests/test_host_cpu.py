"""Host-side logic of the drop-in boundary on CPU (no GPU, no compute calls): checkpoint key set,
config plugin surface, weight prepack layouts, loud failure without a GPU."""
import json
import os

import pytest
import torch

import stitch_amd
from oracle import spec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_checkpoint_key_set_matches_reference():
    want = json.load(open(os.path.join(GOLDEN, "state_keys.json")))
    m = stitch_amd.build_model()
    sd = m.state_dict()
    assert set(sd) == set(want)
    for k, (shape, dtype) in want.items():
        assert list(sd[k].shape) == shape and str(sd[k].dtype).replace("torch.", "") == dtype, k
    # product and oracle enumerate the same contract independently
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v) for k, v in spec.state_spec().items()}


def test_strict_load_with_and_without_dataparallel_prefix(seeded_sd):
    m = stitch_amd.build_model()
    m.load_state_dict(seeded_sd, strict=True)
    m.load_state_dict({"module." + k: v for k, v in seeded_sd.items()}, strict=True)
    assert torch.equal(m.state_dict()["flow_backbone.memory_decoder.update_block.aggregator.gamma"], torch.tensor([0.5]))
    bad = dict(seeded_sd)
    bad.pop("homo_backbone.regressNet1_part2.4.bias")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad, strict=True)


def test_config_plugin_surface():
    for name in ("all_img1_with_inpaint_g12_transRef", "inpaint_all_area_g12_diffusion"):
        cfg, tps = stitch_amd.load_inference_config(name)
        assert cfg.transformer == "percostformer3" and cfg.percostformer3.decoder_depth == 12
        assert cfg.test_not_use_combine_h_flow is True and cfg.use_fb_consistency_mask is True
        assert cfg.use_forward is False and cfg.use_whole_resolution is False and cfg.pad_mode == "replicate"
        assert hasattr(cfg, "use_foward") and not hasattr(cfg, "nonexistent_key")
        assert (tps.grid_h, tps.grid_w) == (12, 12) and tps.tps_method == "opencv"
    assert stitch_amd.load_inference_config("inpaint_all_area_g12_diffusion")[1].inpainter == "inpainter"
    with pytest.raises(ModuleNotFoundError):
        stitch_amd.load_inference_config("does_not_exist")


def test_forward_type_dispatch_and_loud_cpu_failure(seeded_sd):
    m = stitch_amd.build_model()
    x = torch.zeros(1, 3, 512, 512)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, x, type="test_eval")
    with pytest.raises(RuntimeError):
        m.flow_backbone(x, x)
    with pytest.raises(RuntimeError):
        m.homo_backbone(x, x)


def test_prepack_layouts(seeded_sd):
    """BN folding and the (ky, kx, c) GEMM layout reproduce the conv they replace (CPU arithmetic only)."""
    import torch.nn.functional as F
    from stitch_amd.homography import fold_bn, pack_conv
    p = {k[len("homo_backbone."):]: v for k, v in seeded_sd.items() if k.startswith("homo_backbone.")}
    x = torch.randn(1, 3, 20, 24, generator=torch.Generator().manual_seed(0))
    ref = F.batch_norm(F.conv2d(x, p["feature_extractor_stage1.0.weight"], stride=2, padding=3),
                       p["feature_extractor_stage1.1.running_mean"], p["feature_extractor_stage1.1.running_var"],
                       p["feature_extractor_stage1.1.weight"], p["feature_extractor_stage1.1.bias"], False, 0.0, 1e-5)
    s, sh = fold_bn(p, "feature_extractor_stage1.1")
    wp = pack_conv(p["feature_extractor_stage1.0.weight"], 4, s)                   # [64, 7*7*4]
    cols = F.unfold(F.pad(x, (0, 0, 0, 0, 0, 1)), 7, padding=3, stride=2)          # [1, 4*49, L] (c, ky, kx)
    cols = cols.view(1, 4, 49, -1).permute(0, 2, 1, 3).reshape(1, 196, -1)         # -> (ky, kx, c)
    got = (wp @ cols[0] + sh[:, None]).view(1, 64, *ref.shape[2:])
    assert (got - ref).abs().max() < 1e-4
    m = stitch_amd.build_model()
    m.load_state_dict(seeded_sd)
    pk = m.flow_backbone.pack()
    c1 = pk["dec"]["convc1"][0]
    w = seeded_sd["flow_backbone.memory_decoder.update_block.encoder.convc1.weight"].reshape(256, 145)
    assert torch.equal(c1[:, :81], w[:, 64:]) and torch.equal(c1[:, 84:148], w[:, :64]) and (c1[:, 81:84] == 0).all() and (c1[:, 148:] == 0).all()


def test_metric_oracle_and_summary_split():
    """f-2 host logic: skimage-0.19 restatement sanity + evaluate.py:76-93 slices (the worst pair is dropped)."""
    import numpy as np
    from oracle import metrics
    from stitch_amd import evaluate as ev
    rs = np.random.RandomState(0)
    a = rs.randint(0, 256, size=(40, 50, 3)).astype(np.uint8)
    assert metrics.ssim(a, a) == pytest.approx(1.0, abs=1e-12)
    b = np.clip(a.astype(np.int32) + rs.randint(-5, 6, size=a.shape), 0, 255).astype(np.uint8)
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    assert metrics.psnr(a, b) == pytest.approx(10 * np.log10(255.0 ** 2 / mse))
    assert 0.5 < metrics.ssim(a, b) < 1.0
    img = rs.rand(3, 20, 20).astype(np.float32) * 300 - 20
    x, y = metrics.masked_uint8_pair(img, img, np.full((1, 20, 20), 0.99, np.float32))
    assert x.max() == 0 and y.max() == 0                       # mask mean < 1 truncates to 0 (evaluate.py:55)
    p = list(range(1106, 0, -1))
    res = ev.summarize(p, p)
    assert res["easy_psnr"] == pytest.approx(np.mean(p[0:331])) and res["mid_psnr"] == pytest.approx(np.mean(p[331:663]))
    assert res["hard_psnr"] == pytest.approx(np.mean(p[663:-1])) and res["avg_psnr"] == pytest.approx(np.mean(p))


def test_composition_checkpoint_key_set():
    """f-4: the composition network exposes the reference module tree (core/UDIS2/Composition/network.py:78-102) and loads
    `checkpoint['model']` strictly, with or without the DataParallel prefix."""
    from oracle import composition as oc
    net = stitch_amd.composition.Network()
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v) for k, v in oc.state_spec().items()}
    sd = oc.seeded_state_dict(1)
    net.load_state_dict(sd, strict=True)
    net.load_state_dict({"module." + k: v for k, v in sd.items()}, strict=True)
    bad = dict(sd)
    bad.pop("up3.conv.2.bias")
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad, strict=True)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 512, 512), torch.zeros(1, 3, 512, 512))      # CPU tensors: no fallback


def test_fold_layernorm_is_exact_algebra():
    """ops.fold_layernorm: Linear(LayerNorm(x)) == (x - mean) * rstd @ (W * gamma)^T + (b + W @ beta) -- the weight-only folding that
    lets the GEMM normalise its A rows in registers (st_gemm_desc.a_ln); checked in fp64 on the host."""
    import stitch_amd
    g = torch.Generator().manual_seed(0)
    x = torch.randn(50, 128, generator=g, dtype=torch.float64) * 3 + 1
    gam, bet = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    w, b = torch.randn(96, 128, generator=g), torch.randn(96, generator=g)
    wf, bf = stitch_amd.ops.fold_layernorm(gam, bet, w, b)
    assert wf.dtype == torch.float32 and wf.shape == w.shape and bf.shape == b.shape
    ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(x, (128,), gam.double(), bet.double(), 1e-5), w.double(), b.double())
    xn = (x - x.mean(1, keepdim=True)) / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    got = xn @ wf.double().t() + bf.double()
    assert (got - ref).abs().max() < 5e-6                    # (the folded weights are rounded to fp32 once)
    wf2, bf2 = stitch_amd.ops.fold_layernorm(gam, bet, w)    # no bias: b' = W @ beta
    assert torch.equal(wf2, wf) and (bf2.double() - (w.double() @ bet.double())).abs().max() < 1e-6


def test_no_memset_call_in_capturable_code():
    """Round 4: a `hipMemsetAsync` captured into a hipGraph did not clear the range map's accumulator on replays (csrc/geom.hip; reproducer:
    tools/probes/graph_memset_probe.py).  Every entry point of the library may be captured, so no source of it may CALL hipMemset* /
    hipMemcpy* (accumulators are zeroed by kernels); comments may mention them."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in glob.glob(os.path.join(root, "seamless-*", "csrc", "*.hip")) + glob.glob(os.path.join(root, "seamless-*", "csrc", "*.h")):
        src = re.sub(r"//[^\n]*", "", open(f).read())
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        assert not re.search(r"\bhipMem(set|cpy)\w*\s*\(", src), f
