"""TPS post-pipeline on the GPU: the HIP stages through the C-ABI against the CPU oracle
(oracle/tps_pipeline.py, pinned bit for bit to the reference's own core/inference functions) and directly against the
reference golden (tests/golden/tps_pipeline.npz).  Bars: point sets, masks and every elementwise stage bit-exact; the TPS
warp itself to tolerance (the reference solves its fp32 system with MKL's blocked LU and sums the kernel terms with a
vectorised reduction, neither reproducible: here the same fp32 system is solved in fp64 and summed in order)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from _measure import check  # noqa: E402

from oracle import tps_pipeline as otp  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
T = torch.from_numpy


def cfg(**kw):
    d = dict(grid_h=12, grid_w=12, pad_num=4, residual_flow_use_forward=False, flow_limit=-1, add_corner=False,
             get_pt_methods=["advanced_uniform_multi"], affine_scale=1.0, kernel_scale=1.0, use_boundary_limit=False,
             tps_method="kornia", output2_is_only_tps=True, do_avg_pooling=True)
    d.update(kw)
    return SimpleNamespace(**d)


@pytest.fixture(scope="module")
def tp():
    import stitch_amd
    return stitch_amd.tps_pipeline


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "tps_pipeline.npz"))


def cuda_case(case):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.items()}


def test_preprocess_sampling_and_pairs_bit_exact(tp, gold):
    ih, iw, wmin, hmin, oh, ow = 200, 264, -21, -13, 236, 300
    case = otp.synthetic_case(5, ih, iw, wmin, hmin, oh, ow)
    fl = tp.preprocess(case["residual_flow"].cuda(), None, True, False, 12, 12)
    ref = otp.preprocess(case["residual_flow"].clone(), None, True, False, 12, 12)
    assert torch.equal(fl.cpu(), ref)
    assert np.array_equal(fl.cpu()[..., ::3, ::3].numpy(), gold["pre_flow_out_sub"])
    valid = (torch.rand(1, 1, ih, iw, generator=torch.Generator().manual_seed(1)) > 0.3).float()
    fl_v = tp.preprocess(case["residual_flow"].cuda(), valid.cuda(), True, True, 12, 12)
    assert torch.equal(fl_v.cpu(), otp.preprocess(case["residual_flow"].clone(), valid, True, True, 12, 12))
    crop = case["H_warp"][:, :, abs(hmin):abs(hmin) + ih, abs(wmin):abs(wmin) + iw].contiguous()
    import stitch_amd
    grad = stitch_amd.ops.sobel_magnitude(crop.cuda())
    assert torch.equal(grad.cpu(), otp.sobel_magnitude(crop)[0, 0])
    for pad in (4, 22, 44):
        pts = tp.advanced_uniform_sample_border_points(crop.cuda(), max(ih, iw) // 12, pad)
        assert np.array_equal(pts.numpy(), gold[f"sample_pts_pad{pad}"]), pad
    bp = tp.advanced_uniform_sample_border_points(crop.cuda(), 22, 4)
    s, t = tp.get_point_pairs(bp, fl, -1)
    assert np.array_equal(s.numpy(), gold["pairs_src"]) and np.array_equal(t.numpy(), gold["pairs_tgt"])
    s2, t2 = tp.get_point_pairs(bp, fl * 8, 20)
    assert np.array_equal(s2.numpy(), gold["pairs_lim_src"]) and np.array_equal(t2.numpy(), gold["pairs_lim_tgt"])
    bs, bd = tp.boundary_src_and_tgt(s.float() * 1.2 - 10, t * 1.2 - 10, t, out_height=oh, out_width=ow)
    assert np.array_equal(bs.numpy(), gold["bound_src"]) and np.array_equal(bd.numpy(), gold["bound_dst"])


def test_tps_solve_and_warp_vs_reference_golden(gold):
    import stitch_amd
    ops = stitch_amd.ops
    case = otp.synthetic_case(5, 200, 264, -21, -13, 236, 300)
    img = case["H_warp"][:, :, ::2, ::2].contiguous()
    ps, pd = T(gold["tps_ps"]), T(gold["tps_pd"])
    kw, aw = ops.tps2_solve(pd[0].cuda(), ps[0].cuda(), ps[0].cuda(), mode=0)     # get_tps_transform(points_dst, points_src)
    check("tps2_kw_rel", (kw.cpu() - T(gold["tps_kw"])[0]).abs().max().item() / max(1.0, np.abs(gold["tps_kw"]).max()), 2.5e-6)      # measured 7.15e-07
    check("tps2_aw_abs", (aw.cpu() - T(gold["tps_aw"])[0]).abs().max(), 7.5e-7)      # measured 2.38e-07
    out = ops.tps2_warp(img.cuda(), pd[0], ps[0], weights=(T(gold["tps_kw"])[0].cuda(), T(gold["tps_aw"])[0].cuda()))
    d = (out.cpu()[..., ::2, ::2] - T(gold["tps_warp_sub"])).abs()
    print(f"[tps2 warp, reference weights] max {d.max():.3e} p99 {np.percentile(d.numpy(), 99):.3e}")
    check("tps2_warp_refweights_max", d.max(), 6e-3)      # measured 0.00179
    check("tps2_warp_refweights_p99", np.percentile(d.numpy(), 99), 6e-4)      # measured 0.000169
    out2 = ops.tps2_warp(img.cuda(), pd[0], ps[0])
    d2 = (out2.cpu()[..., ::2, ::2] - T(gold["tps_warp_sub"])).abs()
    print(f"[tps2 solve + warp] max {d2.max():.3e} p99 {np.percentile(d2.numpy(), 99):.3e}")
    check("tps2_solve_warp_p99", np.percentile(d2.numpy(), 99), 3e-3)      # measured 0.000911
    check("tps2_solve_warp_mean", d2.mean(), 2e-4)      # measured 6.41e-05


def test_rect_filter_and_mix_blend_bit_exact():
    import stitch_amd
    ops = stitch_amd.ops
    g = torch.Generator().manual_seed(3)
    h, w = 83, 121
    m = (torch.rand(1, 1, h, w, generator=g) > 0.2).float()
    m[:, :, 30:50, 40:90] = 0
    got = ops.rect_filter(ops.rect_filter(m.cuda(), 11, False), 11, True)
    assert torch.equal(got.cpu(), otp.erode_dilate(m, 11))
    assert torch.equal(ops.rect_filter(m.cuda(), 5, True).cpu(), otp._rect_filter(m, 5, True))
    both = torch.rand(1, 6, h, w, generator=g) * 255
    both[:, 3:] = torch.rand(1, 3, h, w, generator=g)
    inv = ops.tps_mask_inv(both[:, 3:].contiguous().cuda())
    assert torch.equal(inv.cpu(), 1.0 - (both[:, 3:].mean(dim=1, keepdim=True) >= 0.5).float())


@pytest.mark.parametrize("name", ["a", "b"])
def test_pipeline_vs_oracle_and_reference_golden(tp, gold, name):
    """tps_H_warp end to end (inpaint_fn=None): control points identical to the oracle's, masks exact, images to tolerance."""
    ih, iw, wmin, hmin, oh, ow, seed = (int(v) for v in gold[f"pipe_{name}_dims"])
    case = otp.synthetic_case(seed, ih, iw, wmin, hmin, oh, ow)
    limit = dict(width_min=wmin, height_min=hmin, out_height=oh, out_width=ow)
    ref = otp.tps_H_warp(case, limit, cfg())
    got = tp.tps_H_warp(cuda_case(case), SimpleNamespace(**limit), cfg())
    assert torch.equal(got["points_src"], ref["points_src"]) and torch.equal(got["points_dst"], ref["points_dst"])
    d = (got["tps_output"].cpu() - ref["tps_output"]).abs()
    mask_flips = int((got["mask2"].cpu() != ref["mask2"]).sum())
    mix_flips = int((got["mix_tps_flow_warp_mask"].cpu() != ref["mix_tps_flow_warp_mask"]).sum())
    db = (got["new_blend_image"].cpu().int() - T(gold[f"pipe_{name}_blend"]).int()).abs()
    print(f"[tps pipeline {name}] n_points {got['points_src'].shape[1]} tps |d| max {d.max():.3e} p99 {np.percentile(d.numpy(), 99):.3e} "
          f"mask flips {mask_flips} mix-mask flips {mix_flips} blend: {(db > 0).float().mean():.2e} of bytes differ, max {int(db.max())}")
    check(f"tps_pipe_{name}_mask_flips", mask_flips, 2, inclusive=True)                      # a mask value within rounding of the 0.5 threshold
    check(f"tps_pipe_{name}_mixmask_flips", mix_flips, 2, inclusive=True)           # measured 0
    # bounds = the REFERENCE's own spread between two MKL code paths on these very inputs (tests/golden/tps_floor.npz, written by
    # oracle/ref_harness/make_tps_floor_golden.py: the reference under MKL_ENABLE_INSTRUCTIONS=AVX2 / SSE4_2 against itself under AVX-512;
    # its fp32 vsLn and its fp32 LU both change bits with the instruction set), not a multiple of this build's measurement
    floor = np.load(os.path.join(os.path.dirname(__file__), "golden", "tps_floor.npz"))
    f_p99 = max(float(floor[f"floor_avx2_pipe_{name}_tps_p99"]), float(floor[f"floor_sse4_2_pipe_{name}_tps_p99"]))                      # a 0.0140, b 0.0403
    f_frac = max(float(floor[f"floor_avx2_pipe_{name}_blend_differs_frac"]), float(floor[f"floor_sse4_2_pipe_{name}_blend_differs_frac"]))  # a 2.9e-4, b 7.6e-4
    check(f"tps_pipe_{name}_tps_p99", np.percentile(d.numpy(), 99), f_p99)          # measured 7.5e-3 (a) / 2.3e-2 (b) grey levels
    check(f"tps_pipe_{name}_blend_gt1_frac", (db > 1).float().mean(), 1e-4)        # measured 0: no byte is off by more than one level
    check(f"tps_pipe_{name}_blend_differs_frac", (db > 0).float().mean(), f_frac)    # measured 1e-4 (a) / 2.7e-4 (b) (fp64 solve vs the reference's fp32 LU)
    gm = np.unpackbits(gold[f"pipe_{name}_mask2_bits"])[:oh * ow].reshape(oh, ow)
    assert int((got["mask2"].cpu()[0, 0].numpy() != gm).sum()) <= 4


def test_opencv_mode_is_the_interpolating_spline(tp):
    """tps_method='opencv' (unpinned against OpenCV itself): the pixel-unit r^2 log r^2 spline must interpolate its control
    points -- a dot drawn at each source point lands on its target -- and reduce to the identity for zero flow."""
    import stitch_amd
    ops = stitch_amd.ops
    h, w = 120, 160
    g = torch.Generator().manual_seed(2)
    src = torch.stack([torch.randint(15, w - 15, (1, 14), generator=g), torch.randint(15, h - 15, (1, 14), generator=g)], -1).float()
    src = torch.unique(src[0], dim=0)[None]
    dst = src + torch.randint(-4, 5, src.shape, generator=g).float()
    img = torch.zeros(1, 1, h, w)
    for x, y in src[0].long().tolist():
        img[0, 0, y, x] = 1.0
    out = ops.tps2_warp(img.cuda(), dst[0], src[0], mode=1).cpu()         # backward map: f(dst_i) = src_i
    for (x, y) in dst[0].long().tolist():
        assert out[0, 0, y, x] > 0.99, (x, y, out[0, 0, y, x])
    ident = ops.tps2_warp(img.cuda(), src[0], src[0], mode=1).cpu()
    assert (ident - img).abs().max() < 1e-4


@pytest.mark.parametrize("mname", ["all_img1_with_inpaint", "inpaint_all_area"])
def test_mix_methods_vs_oracle_and_reference_golden(tp, gold, mname):
    """The `mix_fn` plug-ins through tps_H_warp(inpaint_fn=...) with the pass-through inpainter: masks and the filled image
    follow the oracle (= the reference's own functions, golden `mix_*`); the uint8 blend inherits only the TPS-solve tolerance."""
    import importlib
    mix_fn = importlib.import_module(f"stitch_amd.mix_methods.{mname}").mix_fn
    inp = importlib.import_module("stitch_amd.mix_methods.utils.passthrough_inpainter").inpainter
    ih, iw, wmin, hmin, oh, ow = 200, 264, -21, -13, 236, 300
    case = otp.synthetic_case(5, ih, iw, wmin, hmin, oh, ow)
    limit = dict(width_min=wmin, height_min=hmin, out_height=oh, out_width=ow)
    ofn = {"all_img1_with_inpaint": otp.mix_all_img1_with_inpaint, "inpaint_all_area": otp.mix_inpaint_all_area}[mname]
    ref = otp.tps_H_warp_with_inpaint(case, limit, cfg(), ofn)
    fn = lambda **kw: mix_fn(**kw, inpainter=inp, use_composition=False, is_plot=False, resize_to_area_limit_before_inpaint=750 * 750)  # noqa: E731
    got = tp.tps_H_warp(cuda_case(case), SimpleNamespace(**limit), cfg(), inpaint_fn=fn)
    mflips = int(((got["mask2"].cpu() >= 0.5) != (ref["mask2"] >= 0.5)).sum())
    d = (got["output2"].cpu() - ref["output2"]).abs()
    db = (got["new_blend_image"].cpu().int() - T(gold[f"mix_{mname}_blend"]).int()).abs()
    da = (got["inpaint_area_mask"].cpu() - ref["inpaint_area_mask"]).abs()
    print(f"[mix {mname}] mask2 flips {mflips}, output2 |d| max {d.max():.3e} p99 {np.percentile(d.numpy(), 99):.3e}, area-mask |d| max "
          f"{da[:, -1].max():.1e}, blend: {(db > 0).float().mean():.2e} of bytes differ (max {int(db.max())})")
    check(f"mix_{mname}_mask2_flips", mflips, 2, inclusive=True)                    # measured 0
    assert da[:, -1].max() == 0                                    # the binary "left to the inpainter" mask is exact
    check(f"mix_{mname}_output2_p99", np.percentile(d.numpy(), 99), {"all_img1_with_inpaint": 9.7e-3, "inpaint_all_area": 1.2e-2}[mname])         # measured 3.2e-3 / 3.9e-3
    check(f"mix_{mname}_blend_gt1_frac", (db > 1).float().mean(), 1e-4)           # measured 0
    check(f"mix_{mname}_blend_differs_frac", (db > 0).float().mean(), {"all_img1_with_inpaint": 1.2e-3, "inpaint_all_area": 8.4e-4}[mname])     # measured 3.7e-4 / 2.8e-4


def test_mix_stage_kernels_bit_exact():
    """dilate_thin_area, the 7x7 dilate_mask and the elementwise stages on fractional masks: bit-exact vs the oracle."""
    import stitch_amd
    ops = stitch_amd.ops
    g = torch.Generator().manual_seed(8)
    h, w = 97, 141
    m = (torch.rand(1, 1, h, w, generator=g) > 0.35).float()
    m[:, :, 20:60, 30:100] = 1
    m[:, :, 70:75, :] = torch.rand(1, 1, 5, w, generator=g)                         # fractional band
    for tk in (8, 16):
        res, ge1 = ops.dilate_thin_area_plane(m.cuda(), thickening_kernel_size=tk)
        ref = otp.dilate_thin_area(m, thickening_kernel_size=tk)
        assert torch.equal(res.cpu(), ref[:, 0:1]), tk
        assert torch.equal(ops.rect_filter(ge1, 7, True).cpu(), (otp.dilate_mask(ref.repeat(1, 3, 1, 1), 7)[:, 0:1] > 0).float())
    fw, tps, o1 = (torch.rand(1, 3, h, w, generator=g) * 255 for _ in range(3))
    m1 = torch.rand(1, 3, h, w, generator=g).round() * (torch.rand(1, 3, h, w, generator=g) > 0.1) + 0.3 * (torch.rand(1, 3, h, w, generator=g) > 0.9)
    m1 = m1.clip(0, 1)
    occ, tm = (torch.rand(1, 1, h, w, generator=g) > 0.3).float(), (torch.rand(1, 1, h, w, generator=g) > 0.3).float()
    for method in (0, 1):
        tfw, tfwm, iam0 = ops.mix_stage_a(fw.cuda(), occ.cuda(), m1.cuda(), tps.cuda(), tm.cuda(), method)
        if method == 0:
            inv = 1. - (m1 > 0.5).float()
            rt, rm = fw * occ * m1 + tps * inv, occ * m1 + tm * inv
            ri = ((1. - rm) * m1)[:, 0:1]
        else:
            inv = 1. - m1
            rt, rm = fw * occ + tps * inv, occ + tm * inv
            ri = ((1. - rm) * m1 * tm)[:, 0:1]
        assert torch.equal(tfw.cpu(), rt) and torch.equal(tfwm.cpu(), rm) and torch.equal(iam0.cpu(), ri)
    bl = ops.blend_pair(o1.cuda(), m1.cuda(), tps.cuda(), tm.cuda()).cpu()
    rb = torch.nan_to_num(((o1 * m1 + tps * tm) / (m1 + tm)).clip(0, 255), nan=0.0).to(torch.uint8)
    assert torch.equal(bl, rb)


def test_opencv_branch_quantisation_and_points_vs_oracle(tp):
    """tps_method='opencv' as the reference's own Python feeds it (opencv_tps.py:59-68, utils.py:10): image and mask truncated to
    uint8 BEFORE the warp, result rounded / saturated, scales unused -- against the oracle restatement (fp64).  cv2's fixed-point
    remap and its spline fit themselves stay unpinned (OpenCV is not importable)."""
    case = otp.synthetic_case(9, 120, 160, -9, -7, 140, 180)
    H_warp, H_mask = case["H_warp"], case["H_warp_mask"].clone()
    H_mask[:, :, 40:44] = 0.7                                  # a fractional (bilinear-edge) mask strip: truncates to 0
    g = torch.Generator().manual_seed(4)
    src = torch.stack([torch.randint(10, 170, (1, 30), generator=g), torch.randint(10, 130, (1, 30), generator=g)], -1).float()
    src = torch.cat([src, src[:, :3]], 1)                      # three coincident sites, as advanced_uniform_multi can produce
    dst = src + torch.randint(-3, 4, src.shape, generator=g).float()
    dst[:, -3:] = dst[:, :3]
    ref = otp.warp_by_tps_opencv_like(H_warp, H_mask, src, dst)
    got = tp.warp_by_tps(H_warp.cuda(), H_mask.cuda(), src, dst, 140, 180, "opencv", 3.0, 5.0).cpu()      # scales must be ignored
    assert torch.equal(got, got.round()) and got.min() >= 0 and got.max() <= 255          # uint8-valued
    d = (got - ref).abs()
    mflips = int((got[:, 3:] != ref[:, 3:]).sum())
    print(f"[opencv-like branch] |d| max {d.max():.0f}, fraction of values that differ {(d > 0).float().mean():.2e}, mask flips {mflips}")
    check("opencv_like_differs_frac", (d > 0).float().mean(), 3e-3)         # fp32 spline evaluation vs fp64: a rounding tie now and then
    check("opencv_like_max_levels", d.max(), 1.0, inclusive=True)
    assert (got[:, 3:, 40:44] == 0).all() or (got[:, 3:] <= 1).all()


def test_singular_tps_is_detected_not_nan(tp):
    """coincident sites (kornia back-end: torch.linalg.solve raises in the reference) and collinear sites: the solve reports it,
    the pipeline leaves the homography warp unchanged instead of writing NaN over the canvas."""
    import stitch_amd
    ops = stitch_amd.ops
    img = torch.rand(1, 4, 64, 80).cuda() * 255
    pts = torch.tensor([[10.0, 10.0], [30.0, 12.0], [20.0, 40.0], [10.0, 10.0]])
    with pytest.raises(ops.SingularTPSError):
        ops.tps2_solve((pts / 80).cuda(), (pts / 80).cuda(), (pts / 80).cuda(), mode=0)
    line = torch.tensor([[5.0, 5.0], [10.0, 10.0], [20.0, 20.0], [40.0, 40.0]])
    with pytest.raises(ops.SingularTPSError):
        ops.tps2_solve(line.cuda(), line.cuda(), (line + 1).cuda(), mode=1)
    out = tp.warp_by_tps(img[:, :3], img[:, 3:], pts[None], pts[None] + 1, 64, 80, "kornia", 1.0, 1.0)
    assert torch.equal(out, img) and torch.isfinite(out).all()
