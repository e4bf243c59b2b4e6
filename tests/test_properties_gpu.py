"""Size-independent properties of the hot path at BASELINE.json's full sizes (512^2, 1024^2, B=8), plus
the batch / canvas edge cases.  These do not need the oracle to be fast at full size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from _measure import check  # noqa: E402

from oracle import adapter as oadapter  # noqa: E402
from oracle import inputs, spec  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    import stitch_amd
    return stitch_amd.ops


@pytest.fixture(scope="module")
def model(seeded_sd):
    import stitch_amd
    cfg, _ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
    m = stitch_amd.build_model(cfg)
    m.load_state_dict(seeded_sd, strict=True)
    return m.cuda().eval()


def g(seed):
    return torch.Generator().manual_seed(seed)


def test_corr_volume_full_size_transpose_symmetry(ops):
    """config 3 shape (B=8, 4096x4096x256): corr(f1,f2)[i,j] == corr(f2,f1)[j,i] bit for bit (same k order)."""
    f1 = torch.randn(8, 4096, 256, generator=g(1)).cuda()
    f2 = torch.randn(8, 4096, 256, generator=g(2)).cuda()
    a, b = torch.empty(8, 4096, 4096, device="cuda"), torch.empty(8, 4096, 4096, device="cuda")
    ops.corr_volume(f1, f2, a)
    ops.corr_volume(f2, f1, b)
    assert torch.equal(a, b.transpose(1, 2))
    ref = (f1[3, 100:132].double() @ f2[3].double().t()).float()
    assert (a[3, 100:132] - ref).abs().max() < 1e-3          # |corr| ~ 16, K = 256
    del a, b


def test_range_map_conservation_and_zero_flow(ops):
    z = torch.zeros(2, 2, 1024, 1024, device="cuda")
    assert torch.equal(ops.range_map(z), torch.ones(2, 1, 1024, 1024, device="cuda"))        # identity splat
    f = (torch.rand(1, 2, 1024, 1024, generator=g(3)) * 6 - 3)
    f[:, :, 8:-8, 8:-8] = f[:, :, 8:-8, 8:-8]
    rm = ops.range_map(f.cuda())
    inner = torch.zeros(1, 2, 1024, 1024)
    inner[:, :, 8:-8, 8:-8] = f[:, :, 8:-8, 8:-8]                                           # nothing leaves the image
    total = ops.range_map(inner.cuda()).double().sum().item()
    assert abs(total - 1024 * 1024) < 1e-2                                                   # weights of a splat sum to 1
    assert rm.min() >= 0


def test_morph_open_idempotent_and_monotone(ops):
    m = (torch.rand(1, 1, 1024, 1024, generator=g(4)) > 0.001).float().cuda()
    o1 = ops.morph_open(m)
    assert torch.equal(ops.morph_open(o1), o1)                 # opening is idempotent
    assert (o1 <= m).all()                                     # and anti-extensive


def test_homography_translation_is_a_shift(ops):
    """theta = pure integer translation in pixel units -> output equals the shifted image where both taps are
    in range (weights (1,0,0,0) up to rounding) -- exercises the 1024^2 canvas path."""
    W = H = 1024
    img = torch.rand(1, 3, H, W, generator=g(5)) * 255
    # normalised coords: x_src_pix = (x_n + 1) * W / 2 ; output grid x_n = linspace(-1, 1, W)
    out = ops.homo_warp(img.cuda(), torch.eye(3).reshape(1, 9).cuda(), (H, W)).cpu()
    # identity theta is NOT an identity warp (sample position i*W/(W-1)): check against the oracle on a strip
    from oracle import cgeom
    ref, _ = cgeom.homo_warp(img.numpy(), np.eye(3, dtype=np.float32)[None], (H, W), want_idx=False)
    assert np.array_equal(out.numpy(), ref)


def test_batch_independence(model):
    """B=2 == two B=1 evaluations (no cross-sample op on the path).  Checked per stage with inputs held fixed:
    end to end the seeded random-weight flow net amplifies the ~1e-6 change of H (other split-K at 2x rows) to
    ~0.1 px, exactly as the CPU oracle does under the same perturbation."""
    a0, b0 = inputs.structured_pair(512, 512, seed=31)
    a1, b1 = inputs.structured_pair(512, 512, seed=32, shift=(-4, 7))
    A, Bm = torch.cat([a0, a1]).cuda(), torch.cat([b0, b1]).cuda()
    o = model(A, Bm, type="test_eval")
    assert o["final_warp_output"].shape == (2, 6, 512, 512) and o["origin_occlusion_mask"].shape == (2, 512, 512)
    fwd, bwd = model.predict_flow_pair(A, Bm)
    worst = dict(H=0.0, output_H=0.0, flow_fwd_px=0.0, flow_bwd_px=0.0)            # one bound per quantity = 3x its maximum over the samples
    for i, (a, b) in enumerate(((a0, b0), (a1, b1))):
        s = model(a.cuda(), b.cuda(), type="test_eval")
        f1, b1_ = model.predict_flow_pair(a.cuda(), b.cuda())
        for k, v in (("H", (o["H"][i] - s["H"][0]).abs().max()), ("output_H", (o["output_H"][i] - s["output_H"][0]).abs().max()),
                     ("flow_fwd_px", (fwd[i] - f1[0]).abs().max()), ("flow_bwd_px", (bwd[i] - b1_[0]).abs().max())):
            worst[k] = max(worst[k], float(v))
    check("batch2_vs_1_H", worst["H"], 3.6e-6)                     # measured 1.2e-6
    check("batch2_vs_1_output_H", worst["output_H"], 4.2e-2)       # measured 1.4e-2 grey levels
    check("batch2_vs_1_flow_fwd_px", worst["flow_fwd_px"], 4.5e-2)   # measured 1.5e-2
    check("batch2_vs_1_flow_bwd_px", worst["flow_bwd_px"], 4.5e-2)   # measured 1.5e-2; |flow| ~ 30 px; fp32 reorder (other split-K at 2x rows) x 12 iterations


def test_test_out_1024_vs_oracle(model, seeded_sd):
    """BASELINE.json configs[3] shape: 1024x1024 pair through test_out; canvas ints exact vs the CPU oracle."""
    a, b = inputs.structured_pair(1024, 1024, seed=41, shift=(6, -10))
    o = model(a.cuda(), b.cuda(), type="test_out")
    with torch.no_grad():
        r = oadapter.forward_test_out(seeded_sd, a, b)
    for k in ("width_min", "height_min", "out_height", "out_width"):
        assert o[k] == r[k], (k, o[k], r[k])
    assert tuple(o["blend_image"].shape) == tuple(r["blend_image"].shape) and o["blend_image"].dtype == torch.uint8
    d = (o["blend_image"].cpu().int() - r["blend_image"].int()).abs()
    check("out1024_blend_gt2_frac", (d > 2).float().mean(), 1.2e-3)      # measured 0.000401
    check("out1024_H_rel", (o["H"].cpu() - r["H"]).abs().max().item() / max(1.0, r["H"].abs().max().item()), 7.5e-7)      # measured 2.45e-07
    flips = (o["mask1"].cpu() != r["mask1"]).float().mean()
    check("out1024_mask1_flip_frac", flips, 1e-5)      # measured 0
    occ = (o["occlusion_mask"].cpu() != r["occlusion_mask"]).float().mean()
    check("out1024_occlusion_flip_frac", occ, 1e-4)      # measured 0


def test_rejects_unsupported_requests(model):
    x = torch.zeros(2, 3, 512, 512, device="cuda")
    with pytest.raises(NotImplementedError):
        model(x, x, type="test_out")              # shared canvas: batch must be 1
    with pytest.raises(NotImplementedError):
        model(x, x, type="bogus")
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, 500, 500, device="cuda"), torch.zeros(1, 3, 500, 500, device="cuda"), type="test_eval")


def test_masked_psnr_ssim_kernel_vs_metric_oracle(ops):
    """f-2: evaluate.py:53-59 metric on the GPU vs the numpy/scipy restatement of skimage 0.19 (unpinned vs skimage)."""
    from oracle import metrics
    gg = g(80)
    B, H, W = 3, 96, 128
    img = (torch.rand(B, 3, H, W, generator=gg) * 255).round()
    noise = torch.randn(B, 3, H, W, generator=gg) * torch.tensor([2.0, 10.0, 40.0]).view(B, 1, 1, 1)
    warped = (img + noise).clamp(-20, 280)                                    # exercises the uint8 clip
    mask3 = torch.ones(B, 3, H, W)
    mask3[:, :, :, 100:] = 0                                                  # invalid strip
    mask3[1, 0, 10:20, 10:20] = 0.5                                           # mean < 1 -> uint8 truncation to 0
    fwo = torch.cat([warped * (mask3.mean(1, keepdim=True) > 0), mask3], 1)
    got = ops.masked_psnr_ssim(img.cuda(), fwo.cuda().contiguous()).cpu().numpy()
    for b in range(B):
        p, s = metrics.pair_metrics(img[b].numpy(), fwo[b, :3].numpy(), fwo[b, 3:6].mean(0, keepdim=True).numpy())
        assert abs(got[b, 0] - p) < 1e-9 * max(1.0, abs(p)), (got[b, 0], p)       # integer-exact MSE
        assert abs(got[b, 1] - s) < 1e-9, (got[b, 1], s)
    again = ops.masked_psnr_ssim(img.cuda(), fwo.cuda().contiguous()).cpu().numpy()
    assert np.array_equal(got, again)                                             # deterministic reduction


def test_validate_with_model_harness(model, tmp_path):
    """f-2 harness: UDISDataset (PIL) -> sharded validate -> reference summary keys."""
    from PIL import Image
    import stitch_amd
    from stitch_amd import evaluate as ev
    for i in range(3):
        a, b = inputs.structured_pair(512, 512, seed=50 + i)
        for name, t in (("input1", a), ("input2", b)):
            d = tmp_path / "testing" / name
            d.mkdir(parents=True, exist_ok=True)
            Image.fromarray(t[0].permute(1, 2, 0).byte().numpy()).save(str(d / f"{i:06d}.jpg"), quality=95)
    ds = ev.UDISDataset(str(tmp_path) + "/")
    assert len(ds) == 3 and ds[0][0].shape == (3, 512, 512)
    res, table = ev.validate_with_model(model, ds, batch_size=2)
    assert set(res) == {"avg_psnr", "avg_ssim", "easy_psnr", "mid_psnr", "hard_psnr", "easy_ssim", "mid_ssim", "hard_ssim"}
    assert table.shape == (3, 2) and torch.isfinite(table).all() and 0 < res["avg_ssim"] <= 1.0
    # the root evaluate.py CLI (reference flags, evaluate.py:130-152) drives the same harness from a checkpoint file
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec_ = importlib.util.spec_from_file_location("stitch_eval_cli", os.path.join(root, "evaluate.py"))
    cli = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(cli)
    ck = tmp_path / "final_ckpt"
    torch.save({"module." + k: v for k, v in model.state_dict().items()}, str(ck))
    res2 = cli.main(["--ckpt_path", str(ck), "--data_dir", str(tmp_path) + "/", "--batch_size", "3"])
    assert set(res2) == set(res) and abs(res2["avg_psnr"] - res["avg_psnr"]) < 0.5


def test_out_harness_writes_the_reference_file_set(tmp_path, seeded_sd):
    """f-1: out.py flags / demo.txt list / result naming / the 7 JPEGs, on the 256x256 demo1 pair held as fixture data."""
    import os
    from PIL import Image
    sys_path_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("stitch_out_harness", os.path.join(sys_path_root, "out.py"))
    outmod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(outmod)
    g_ = np.load(os.path.join(sys_path_root, "tests", "golden", "e2e_out_256.npz"))
    root = tmp_path / "demo"
    (root / "demo1").mkdir(parents=True)
    Image.fromarray(g_["input1"]).save(str(root / "demo1" / "input1.jpg"), quality=98)
    Image.fromarray(g_["input2"]).save(str(root / "demo1" / "input2.jpg"), quality=98)
    (root / "demo.txt").write_text("demo1/\n")
    ck = tmp_path / "ckpts" / "seeded" / "final_ckpt"
    ck.parent.mkdir(parents=True)
    torch.save({"module." + k: v for k, v in seeded_sd.items()}, str(ck))
    save_root = outmod.main(["--data_root_path", str(root) + "/", "--ckpt_path", str(ck)])       # the reference's flag (out.py:18)
    assert "ours__seeded_advanced_uniform_multi_all_img1_with_inpaint_g12" in save_root
    files = sorted(os.listdir(os.path.join(save_root, "demo1")))
    # the 7 warp-stage JPEGs + the 3 files of the composition stage (out.py:303-312; cfg.use_composition is on in this plugin)
    assert files == sorted(["H_warp.jpg", "flow_warp.jpg", "warp1.jpg", "warp2.jpg", "mask1.jpg", "mask2.jpg", "ave_fusion.jpg",
                            "composition.jpg", "learned_mask1.jpg", "learned_mask2.jpg"])
    assert Image.open(os.path.join(save_root, "demo1", "composition.jpg")).size[0] >= 512      # canvases < 512 are up-scaled
    im = Image.open(os.path.join(save_root, "demo1", "ave_fusion.jpg"))
    assert abs(im.size[0] - 256) < 64 and abs(im.size[1] - 256) < 64


def test_gemm_beyond_32bit_offsets_runs_in_chunks():
    """Operands larger than the 2 GiB a buffer descriptor / 32-bit epilogue offset reaches (whole-batch PatchEmbed maps at
    B >= 4) are processed in row chunks that shift every operand's base: same bits as the small calls."""
    import stitch_amd
    ops = stitch_amd.ops
    g = torch.Generator().manual_seed(0)
    # conv: 160 images of 64x64 inside a 1024-wide row buffer -> A extent 2.7 GB
    B, H, W, C, ld = 160, 64, 64, 32, 1024
    xw = torch.empty(B * H * W, ld, device="cuda")
    xw[:, 8:8 + C] = torch.randn(B * H * W, C, generator=g).cuda()
    w = (torch.randn(24, 9 * C, generator=g) / (9 * C) ** 0.5).cuda()
    res = torch.randn(B * H * W, 24, generator=g).cuda()
    out = torch.empty(B * H * W, 24, device="cuda")
    ops.conv_gemm(xw[:, 8:8 + C], w, out, geom=(B, H, W, 3, 3, 1, 1, 1, 1), act="relu", epi="add", aux1=res)
    for b0 in (0, 77, 159):
        rows = slice(b0 * H * W, (b0 + 1) * H * W)
        ref = torch.empty(H * W, 24, device="cuda")
        ops.conv_gemm(xw[rows, 8:8 + C], w, ref, geom=(1, H, W, 3, 3, 1, 1, 1, 1), act="relu", epi="add", aux1=res[rows])
        assert torch.equal(out[rows], ref)
    del xw, out, res
    # plain matrix with a (row / 8) % 40 addend table: output 4.4M x 128 fp32 = 2.25 GB
    M, K, N = 4_400_000, 64, 128
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / 8).cuda()
    tab = torch.randn(40, N, generator=g).cuda()
    out = torch.empty(M, N, device="cuda")
    ops.conv_gemm(x, w, out, aux0=tab, row_div=8, row_mod=40)
    for m0 in (0, 2_999_680, M - 640):          # multiples of 8*40
        ref = torch.empty(640, N, device="cuda")
        ops.conv_gemm(x[m0:m0 + 640], w, ref, aux0=tab, row_div=8, row_mod=40)
        assert torch.equal(out[m0:m0 + 640], ref)


def test_batch_of_four_pairs_runs_and_matches_single(model):
    """evaluate.py feeds batches (bs 12 in the reference): B=4 pairs = 8 FlowFormer samples = 32768 cost maps, whose
    PatchEmbed activations exceed 2 GiB and go through the chunked GEMM path.  Sample 2 of the batch vs the same pair alone."""
    pairs = [inputs.structured_pair(512, 512, seed=60 + i, shift=(2 * i - 3, 5 - i)) for i in range(4)]
    A, Bm = torch.cat([p[0] for p in pairs]).cuda(), torch.cat([p[1] for p in pairs]).cuda()
    fwd, bwd = model.predict_flow_pair(A, Bm)
    assert fwd.shape == (4, 2, 512, 512) and torch.isfinite(fwd).all() and torch.isfinite(bwd).all()
    f1, b1 = model.predict_flow_pair(pairs[2][0].cuda(), pairs[2][1].cuda())
    check("batch4_vs_1_flow_fwd_px", (fwd[2] - f1[0]).abs().max(), 4.4e-2)      # measured 1.5e-2
    check("batch4_vs_1_flow_bwd_px", (bwd[2] - b1[0]).abs().max(), 4.2e-2)      # measured 1.4e-2
    o = model(A, Bm, type="test_eval")
    assert o["final_warp_output"].shape == (4, 6, 512, 512) and torch.isfinite(o["final_warp_output"]).all()
