"""BASELINE.json configs[4] at its size: the full out.py chain (reference out.py:158-312) on a 512x512 pair --
`test_out` -> `tps_H_warp` (in-tree kornia-style back-end) -> `mix_fn` all_img1_with_inpaint (pass-through inpainter: TransRef
itself is out of scope, weights + mmcv absent) -> composition network -- on the HIP kernels against the same chain through
oracle/.  Two comparisons:
  * post-pipeline held to the ORACLE's `test_out` outputs (identical inputs on both sides): control points exact, masks
    exact, images to the TPS-solve tolerance -- the enforceable criterion;
  * end to end from the images (the HIP `test_out` feeds the HIP post-pipeline): the seeded random-weight flow network
    amplifies rounding differences, so masks / bytes are bounded by small fractions."""
import importlib
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from _measure import check  # noqa: E402

from oracle import adapter as oadapter  # noqa: E402
from oracle import composition as oc  # noqa: E402
from oracle import inputs  # noqa: E402
from oracle import tps_pipeline as otp  # noqa: E402


def _post_inputs(o, tpc, cuda):
    """out.py:218-232: the post-pipeline's inputs from a `test_out` dict."""
    dev = (lambda t: t.cuda()) if cuda else (lambda t: t.cpu())
    bpm = o["occlusion_mask"] if tpc.use_occ_filter else (o["H_warp_mask"].mean(dim=1, keepdim=True) > 0.5).float()
    d = dict(output1=o["output1"], mask1=o["mask1"], H_warp=o["H_warp"], H_warp_mask=o["H_warp_mask"], final_warp=o["final_warp"],
             mask2=o["mask2"], residual_flow=o["residual_flow"], valid=None, occlusion_mask=o["occlusion_mask"],
             border_points_mask=bpm if tpc.use_border_points_mask else None)
    lim = dict(width_min=o["width_min"], height_min=o["height_min"], out_height=o["out_height"], out_width=o["out_width"])
    return {k: (dev(v) if torch.is_tensor(v) else v) for k, v in d.items()}, lim


@pytest.fixture(scope="module")
def chain(seeded_sd):
    import stitch_amd
    cfg, tpc = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
    tpc.tps_method = "kornia"                     # the back-end the reference holds in-tree (the shipped "opencv" one is unpinned)
    tpc.use_occ_filter = False                    # seeded random weights: the occlusion mask keeps no border point at all; filter the
    #                                               control points by the warped-image mask instead (out.py:229-232, the other branch)
    model = stitch_amd.build_model(cfg)
    model.load_state_dict(seeded_sd, strict=True)
    model = model.cuda().eval()
    comp = stitch_amd.composition.Network()
    comp.load_state_dict(oc.seeded_state_dict(4321), strict=True)
    comp = comp.cuda().eval()
    a, b = inputs.structured_pair(512, 512, seed=61, shift=(5, -7))
    with torch.no_grad():
        ref_out = oadapter.forward_test_out(seeded_sd, a, b)
    return dict(cfg=cfg, tpc=tpc, model=model, comp=comp, a=a, b=b, ref_out=ref_out, sd=seeded_sd)


def _hip_post(chain, o):
    import stitch_amd
    tpc = chain["tpc"]
    mix_fn = importlib.import_module(f"stitch_amd.mix_methods.{tpc.mix_method}").mix_fn
    inp = importlib.import_module("stitch_amd.mix_methods.utils.passthrough_inpainter").inpainter
    fn = lambda **kw: mix_fn(**kw, inpainter=inp, use_composition=tpc.use_composition_when_inpaint, is_plot=False,   # noqa: E731
                             resize_to_area_limit_before_inpaint=tpc.resize_to_area_limit_before_inpaint)
    ins, lim = _post_inputs(o, tpc, cuda=True)
    new = stitch_amd.tps_pipeline.tps_H_warp(ins, SimpleNamespace(**lim), tpc, inpaint_fn=fn)
    m2 = new["mask2"] if new["mask2"].shape[1] == 3 else new["mask2"].expand(-1, 3, -1, -1)
    comp = stitch_amd.composition.compose(chain["comp"], ins["output1"], new["output2"], (ins["mask1"] > 0.5).float(), (m2 > 0.5).float())
    return new, comp


def _oracle_post(chain, o, solve_dtype=None):
    tpc = chain["tpc"]
    ins, lim = _post_inputs(o, tpc, cuda=False)
    new = otp.tps_H_warp_with_inpaint(ins, lim, tpc, otp.mix_all_img1_with_inpaint, inpainter=otp.PassthroughInpainter(), solve_dtype=solve_dtype)
    m2 = new["mask2"] if new["mask2"].shape[1] == 3 else new["mask2"].expand(-1, 3, -1, -1)
    comp = oc.compose(oc.seeded_state_dict(4321), ins["output1"], new["output2"], (ins["mask1"] > 0.5).float(), (m2 > 0.5).float())
    return new, comp


def _gap(new, comp, ref_new, ref_comp):
    d2 = (new["output2"].cpu() - ref_new["output2"].cpu()).abs()
    db = (new["new_blend_image"].cpu().int() - ref_new["new_blend_image"].cpu().int()).abs()
    return dict(mask2_flips=int(((new["mask2"].cpu() >= 0.5) != (ref_new["mask2"].cpu() >= 0.5)).sum()),
                area_mask_flips=int((new["inpaint_area_mask"].cpu()[:, -1] != ref_new["inpaint_area_mask"].cpu()[:, -1]).sum()),
                output2_p99=float(np.percentile(d2.numpy(), 99)), blend_differs_frac=(db > 0).float().mean().item(),
                blend_gt1_frac=(db > 1).float().mean().item(),
                stitched_max=(comp["stitched_image"].cpu() - ref_comp["stitched_image"].cpu()).abs().max().item(),
                stitched_p999=float(np.percentile((comp["stitched_image"].cpu() - ref_comp["stitched_image"].cpu()).abs().numpy(), 99.9)),
                learned_mask_p999=float(np.percentile((comp["learned_mask1"].cpu() - ref_comp["learned_mask1"].cpu()).abs().numpy(), 99.9)))


def test_post_pipeline_512_given_oracle_test_out(chain):
    """tps_H_warp + mix_fn + composition on the ORACLE's 512x512 `test_out` outputs, HIP vs oracle.

    This canvas yields 101 control points of which two lie 0.68 px apart: the TPS system has cond = 8.8e6 and the reference's
    fp32 `torch.linalg.solve` is then a property of its LAPACK build (tests/test_oracle_pin.py::test_tps_fp32_solve_is_lapack_
    dependent: MKL, scipy and plain LUs are ~2 % apart, as far as each is from the fp64 solution).  The GPU solves the same fp32
    system in fp64, so the criterion is the oracle with that solve in fp64 (everything else in the reference's arithmetic);
    the reference-arithmetic oracle is the control: the HIP path must not be further from it than the fp64-solve oracle is."""
    ref_new, ref_comp = _oracle_post(chain, chain["ref_out"], solve_dtype=torch.float64)
    ref32_new, ref32_comp = _oracle_post(chain, chain["ref_out"])
    new, comp = _hip_post(chain, chain["ref_out"])
    assert torch.equal(new["points_src"].cpu(), ref_new["points_src"]) and torch.equal(new["points_dst"].cpu(), ref_new["points_dst"])
    n_pts = int(new["points_src"].shape[1])
    g64, g32, ctl = _gap(new, comp, ref_new, ref_comp), _gap(new, comp, ref32_new, ref32_comp), _gap(ref_new, ref_comp, ref32_new, ref32_comp)
    print(f"[chain 512, oracle test_out] canvas {tuple(new['output2'].shape[-2:])} control points {n_pts}")
    print("   HIP vs oracle(fp64 solve):", json.dumps(g64))
    print("   HIP vs oracle(reference fp32 solve):", json.dumps(g32))
    print("   control: oracle(fp64 solve) vs oracle(fp32 solve):", json.dumps(ctl))
    assert n_pts >= 20
    # |w| reaches 154 on this system: the last-bit differences of logf in r^2 log r^2 (ocml vs SLEEF) and the kernel-sum order are
    # amplified to a few hundredths of a grey level and a handful of threshold flips even with identical (fp64) solves
    check("chain512_fixed_mask2_flips", g64["mask2_flips"], 15, inclusive=True)             # measured 5 of 302 211 pixels
    check("chain512_fixed_area_mask_flips", g64["area_mask_flips"], 0, inclusive=True)
    check("chain512_fixed_output2_p99", g64["output2_p99"], 0.13)                           # measured 4.4e-2 grey levels
    check("chain512_fixed_blend_gt1_frac", g64["blend_gt1_frac"], 3e-5)                     # measured 1e-5 (3 bytes: the flipped mask pixels)
    check("chain512_fixed_blend_differs_frac", g64["blend_differs_frac"], 5e-3)             # measured 1.7e-3
    check("chain512_fixed_stitched_p999", g64["stitched_p999"], 7e-3)                       # stitched image in [-1, 1]
    check("chain512_fixed_learned_mask_p999", g64["learned_mask_p999"], 2e-2)               # (a flipped input-mask pixel moves the net's seam mask locally)
    # against the reference's own (LAPACK-dependent) arithmetic: no further than the fp64-solve oracle is from it
    check("chain512_vs_fp32ref_mask2_flips_over_control", g32["mask2_flips"] / max(1.0, ctl["mask2_flips"]), 1.5)
    check("chain512_vs_fp32ref_output2_p99_over_control", g32["output2_p99"] / max(1e-3, ctl["output2_p99"]), 1.5)


def test_chain_512_end_to_end(chain):
    """images -> test_out -> post-pipeline -> composition, all on the HIP kernels, against the oracle chain."""
    o = chain["model"](chain["a"].cuda(), chain["b"].cuda(), type="test_out")
    r = chain["ref_out"]
    for k in ("width_min", "height_min", "out_height", "out_width"):
        assert o[k] == r[k], (k, o[k], r[k])
    new, comp = _hip_post(chain, o)
    ref_new, ref_comp = _oracle_post(chain, r, solve_dtype=torch.float64)     # (see the test above for why the fp64 solve)
    same_pts = new["points_src"].shape == ref_new["points_src"].shape and torch.equal(new["points_src"].cpu(), ref_new["points_src"])
    dpts = (new["points_dst"].cpu() - ref_new["points_dst"]).abs().max().item() if same_pts else float("nan")
    mflip = ((new["mask2"].cpu() >= 0.5) != (ref_new["mask2"] >= 0.5)).float().mean().item()
    db = (new["new_blend_image"].cpu().int() - ref_new["new_blend_image"].int()).abs()
    ds = (comp["stitched_image"].cpu() - ref_comp["stitched_image"]).abs()
    rec = dict(same_control_points=bool(same_pts), points_dst_max_px=dpts, mask2_flip_frac=mflip, blend_gt2_frac=(db > 2).float().mean().item(),
               stitched_p99=float(np.percentile(ds.numpy(), 99)), stitched_mean=ds.mean().item())
    print("[chain 512, end to end]", json.dumps(rec))
    # control, measured in this run (as tests/test_parity_gpu.py does for the evaluation metric): the CPU oracle against ITSELF, started from the HIP
    # path's corner offsets (they differ from the oracle's by ~1e-5 px) -- how far the seeded random-weight chain moves for a perturbation of that size
    motion = chain["model"].predict_homo(chain["a"].cuda(), chain["b"].cuda()).cpu()
    with torch.no_grad():
        r2 = oadapter.forward_test_out(chain["sd"], chain["a"], chain["b"], motion=motion)
    sens = None
    if all(r2[k] == r[k] for k in ("width_min", "height_min", "out_height", "out_width")):
        new2, comp2 = _oracle_post(chain, r2, solve_dtype=torch.float64)
        if new2["points_src"].shape == ref_new["points_src"].shape and torch.equal(new2["points_src"], ref_new["points_src"]):
            d2 = (comp2["stitched_image"] - ref_comp["stitched_image"]).abs()
            sens = dict(points_dst_max_px=(new2["points_dst"] - ref_new["points_dst"]).abs().max().item(),
                        mask2_flip_frac=((new2["mask2"] >= 0.5) != (ref_new["mask2"] >= 0.5)).float().mean().item(),
                        blend_gt2_frac=((new2["new_blend_image"].int() - ref_new["new_blend_image"].int()).abs() > 2).float().mean().item(),
                        stitched_p99=float(np.percentile(d2.numpy(), 99)))
            # the two paths from the SAME corner offsets: what the stages after the homography contribute
            ds2 = (comp["stitched_image"].cpu() - comp2["stitched_image"]).abs()
            same = dict(points_dst_max_px=(new["points_dst"].cpu() - new2["points_dst"]).abs().max().item(),
                        mask2_flip_frac=((new["mask2"].cpu() >= 0.5) != (new2["mask2"] >= 0.5)).float().mean().item(),
                        blend_gt2_frac=((new["new_blend_image"].cpu().int() - new2["new_blend_image"].int()).abs() > 2).float().mean().item(),
                        stitched_p99=float(np.percentile(ds2.numpy(), 99)))
            print("[chain 512, oracle sensitivity]", json.dumps(sens))
            print("[chain 512, same start]", json.dumps(same))
    assert same_pts, "control point sites differ"               # integer sites from the Sobel sampling of H_warp
    check("chain512_e2e_points_dst_max_px", dpts, 0.05)          # measured 1.6e-2
    #          # site + box-averaged flow: inherits the end-to-end flow gap
    # mask / byte fractions: chaotic end-to-end figures of a seeded random-weight network -- a different draw for every summation order of any kernel in
    # front (mask flips: round 3 1.1e-3, rounds 4-5 2.8e-4, round 6 with the block tails on the split3 kernel 1.2e-3).  A constant fitted to one draw is
    # not a bound; the bound is the larger of the round-3..5 constant and 4x what the oracle itself moves by in this run (the factor the parity test allows
    # the end-to-end flow over the oracle's sensitivity; measured here: 3.6x for the mask flips, 1.8x for the bytes, 1.4x for the stitched image, while
    # the flow network's own gap to the oracle is unchanged from round 5 -- profiles/r6_parity.json: same-start flow p99 4.0e-3 px, 105 occlusion flips).
    s_m, s_b, s_s = (4.0 * sens[k] if sens else 0.0 for k in ("mask2_flip_frac", "blend_gt2_frac", "stitched_p99"))
    note = "no oracle control (canvas or control points moved)" if sens is None else f"4x the oracle's own sensitivity in this run: {json.dumps(sens)}"
    check("chain512_e2e_mask2_flip_frac", mflip, max(9e-4, s_m), note=note)
    check("chain512_e2e_blend_gt2_frac", (db > 2).float().mean(), max(3.4e-3, s_b), note=note)
    check("chain512_e2e_stitched_p99", np.percentile(ds.numpy(), 99), max(1e-2, s_s), note=note)
    assert torch.isfinite(comp["stitched_image"]).all()
