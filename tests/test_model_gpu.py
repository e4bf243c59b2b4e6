"""Network- and path-level parity on the GPU: the HIP-backed drop-in modules against the CPU oracle and
against the golden vectors produced by the reference (tests/golden), same seeded weights, same inputs.

End-to-end tolerances are quoted next to the reference's own self-noise (8 vs 1 CPU
threads changes flow by 1.1e-4 px, output_H by 8.8e-3, flips 3 of 262144 occlusion pixels)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from _measure import check  # noqa: E402

from oracle import adapter as oadapter  # noqa: E402
from oracle import inputs, nets, spec  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
T = torch.from_numpy


@pytest.fixture(scope="module")
def model(seeded_sd):
    import stitch_amd
    cfg, _ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
    m = stitch_amd.build_model(cfg)
    m.load_state_dict(seeded_sd, strict=True)
    return m.cuda().eval()


def rows_to_nchw(y, B, H, W):
    return y.cpu().reshape(B, H, W, -1).permute(0, 3, 1, 2)


def _bits(t):
    return np.packbits((t.detach().cpu().numpy() >= 0.5).astype(np.uint8).reshape(-1))


def test_resnet_and_regress(model, golden_ops, seeded_sd):
    import stitch_amd
    ops = stitch_amd.ops
    hb = model.homo_backbone
    hb.pack()
    im = T(golden_ops["res_in"])
    x = torch.empty((64 * 96, 4), device="cuda")
    ops.prep_image(im.cuda(), x, 4, 1.0, 1.0, 0.0)
    f, h, w = hb.features(x, 1, 64, 96)
    ref = T(golden_ops["res_stage2"])
    err = (rows_to_nchw(f, 1, h, w) - ref).abs().max().item()
    check("res_stage2_rel", err / max(1.0, ref.abs().max().item()), 4e-6)      # measured 1.12e-06
    cf = T(golden_ops["regress_in"])
    xr = torch.zeros((32 * 32, 4), device="cuda")
    xr[:, :2] = cf.permute(0, 2, 3, 1).reshape(-1, 2).cuda()
    off = hb.regress(xr, 1, 32, 32)
    check("regress_out_abs_px", (off.cpu() - T(golden_ops["regress_out"])).abs().max(), 2.5e-5)      # measured 7.63e-06


def test_homography_offsets_512(model, seeded_sd):
    a, b = inputs.structured_pair(512, 512, seed=7)
    ref = nets.homo_offsets(nets.W(seeded_sd, "homo_backbone."), a, b)
    got = model.predict_homo(a.cuda(), b.cuda()).cpu()
    check("homo_offsets_512_px", (got - ref).abs().max(), 5e-5)      # measured 1.53e-05


def test_twins_encoder(model, golden_ops):
    import stitch_amd
    ops = stitch_amd.ops
    fb = model.flow_backbone
    pk = fb.pack()
    im = T(golden_ops["twins_in"])
    x = torch.empty((64 * 96, 4), device="cuda")
    ops.prep_image(im.cuda(), x, 4, 1.0, 1.0, 0.0)
    f, h, w = fb._twins(pk["fnet"], x, 1, 64, 96)
    ref = T(golden_ops["twins_out"])
    check("twins_out_rel", (rows_to_nchw(f, 1, h, w) - ref).abs().max().item() / max(1.0, ref.abs().max().item()), 2e-6)      # measured 6.54e-07


def test_cost_encoder_blocks(model, golden_ops, seeded_sd):
    fb = model.flow_backbone
    fb.pack()
    g = golden_ops
    cm = T(g["pe_in"])                                     # [8,1,64,64]
    tok, P = fb._patch_embed(cm.reshape(8, -1).cuda().contiguous(), 8, 64, 64)
    assert P == 64
    check("pe_out_abs", (tok.cpu().view(8, 64, 128) - T(g["pe_out"])).abs().max(), 1e-5)      # measured 3.04e-06
    x = fb._latent_layer(fb._pk["xin"], None, 6, True, T(g["xattn_tokens"]).reshape(-1, 128).cuda().contiguous(), 64)
    check("xattn_out_abs", (x.cpu().view(6, 8, 128) - T(g["xattn_out"])).abs().max(), 5e-6)      # measured 1.43e-06
    y = fb._latent_layer(fb._pk["self"][1], T(g["sattn_in"]).reshape(-1, 128).cuda().contiguous(), 6, False)
    check("sattn_out_abs", (y.cpu().view(6, 8, 128) - T(g["sattn_out"])).abs().max(), 6e-6)      # measured 1.79e-06
    vx, vctx = T(g["vert_x"]), T(g["vert_ctx"])            # [8 latents, 192 px, 128], [1,256,12,16]
    xr = vx.permute(1, 0, 2).reshape(-1, 128).cuda().contiguous()      # rows (n, l)
    ctx = vctx.permute(0, 2, 3, 1).reshape(-1, 256).cuda().contiguous()
    out = fb._vertical(fb._pk["vert"][2], xr, ctx, 1, 12, 16, 8)
    got = out.cpu().view(192, 8, 128).permute(1, 0, 2)
    check("vert_out_rel", (got - T(g["vert_out"])).abs().max().item() / max(1.0, T(g["vert_out"]).abs().max().item()), 3e-6)      # measured 8.54e-07


def test_corr_volume_vs_reference_golden(golden_ops):
    """MemoryEncoder.corr (encoder.py:359-369), un-scaled all-pairs dot products, against the reference's own output."""
    import stitch_amd
    f1, f2, ref = T(golden_ops["corr_f1"]), T(golden_ops["corr_f2"]), T(golden_ops["corr_out"])
    B, C, H, W = f1.shape
    r1 = f1.permute(0, 2, 3, 1).reshape(B, H * W, C).cuda().contiguous()
    r2 = f2.permute(0, 2, 3, 1).reshape(B, H * W, C).cuda().contiguous()
    vol = torch.empty((B, H * W, H * W), device="cuda")
    stitch_amd.ops.corr_volume(r1, r2, vol)
    err = (vol.cpu().reshape(ref.shape) - ref).abs().max().item()
    check("corr_out_rel", err / ref.abs().max().item(), 8e-7)      # measured 2.64e-07


def test_resnet_stage1_vs_reference_golden(model, golden_ops):
    """conv1/bn1/relu/maxpool/layer1/layer2 (network.py:103-118) against the reference's `res_stage1`."""
    import stitch_amd
    ops = stitch_amd.ops
    hb = model.homo_backbone
    pk = hb.pack()
    im = T(golden_ops["res_in"])
    B, H, W = 1, 64, 96
    x = torch.empty((H * W, 4), device="cuda")
    ops.prep_image(im.cuda(), x, 4, 1.0, 1.0, 0.0)
    H2, W2 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    c1 = torch.empty((B * H2 * W2, 64), device="cuda")
    ops.conv_gemm(x, pk["stem"][0], c1, geom=(B, H, W, 7, 7, 2, 2, 3, 3), bias=pk["stem"][1], act="relu")
    h, w = (H2 + 2 - 3) // 2 + 1, (W2 + 2 - 3) // 2 + 1
    y = torch.empty((B * h * w, 64), device="cuda")
    ops.maxpool(c1, y, B, H2, W2, 64, 3, 2, 1)
    for lname, nblocks, stride in (("feature_extractor_stage1.4", 3, 1), ("feature_extractor_stage1.5", 4, 2)):
        for i in range(nblocks):
            y, h, w = hb._bottleneck(pk[f"{lname}.{i}"], y, B, h, w, stride if i == 0 else 1)
    ref = T(golden_ops["res_stage1"])
    err = (rows_to_nchw(y, 1, h, w) - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


def test_update_block_vs_reference_golden(model, golden_ops):
    """One refinement iteration of GMAUpdateBlock (gru.py:322-334: BasicMotionEncoder :246-254, Aggregate gma.py:102-115,
    SepConvGRU :44-59, FlowHead :5-13, mask head :315-318,333) + GMA Attention (gma.py:54-76) + convex upsampling
    (decoder.py:214-225), on the reference's own inputs and outputs (`gma_attn`, `ub_net_out`, `ub_dflow`, `ub_mask`,
    `up_out`)."""
    import stitch_amd
    ops = stitch_amd.ops
    g = golden_ops
    fb = model.flow_backbone
    D = fb.pack()["dec"]
    B, H1, W1 = 1, 12, 16
    N = R = H1 * W1

    def rows(t):
        return t.permute(0, 2, 3, 1).reshape(R, -1).cuda().contiguous()
    inp = rows(T(g["gma_inp"]))
    qk = torch.empty((R, 256), device="cuda")
    attn = torch.empty((B, N, N), device="cuda")
    ops.gma_attention(inp, D["qk"], qk, attn, B, N)
    ref_attn = T(g["gma_attn"]).reshape(B, N, N)
    assert (attn.cpu() - ref_attn).abs().max() < 2e-6, (attn.cpu() - ref_attn).abs().max()      # softmax rows, values <= 1
    S = fb._update_state(R, B, N, torch.device("cuda"))
    S["hxA"][:, :128] = rows(T(g["ub_net"]))
    if S.get("s3"):               # split3 path: the state also lives as bf16 planes (written by proj_net's / the q convs' epilogues in a forward)
        ops.split3_pack(S["hxA"][:, :128], out=S["hxA_p"].cols(0, 128))
    corr = T(g["ub_corr"])                                     # reference order: cat([cost_global 64, cost_forward 81])
    S["corr"][:, :81] = rows(corr[:, 64:])
    S["corr"][:, 84:148] = rows(corr[:, :64])
    coords0 = torch.empty((R, 2), device="cuda")
    ops.coords_grid(coords0, B, H1, W1)
    coords1 = coords0 + rows(T(g["ub_flow"]))
    before = coords1.clone()
    tabs = fb._gru_tables(inp, B, H1, W1)
    fb._update_block(S, coords1, attn, tabs, B, H1, W1)
    net = rows_to_nchw(S["hxA"][:, :128], B, H1, W1)
    dflow = rows_to_nchw(coords1 - before, B, H1, W1)
    e_net = (net - T(g["ub_net_out"])).abs().max().item()
    e_df = (dflow - T(g["ub_dflow"])).abs().max().item()
    mask = fb._mask_head(S, B, H1, W1)
    e_mask = (rows_to_nchw(mask, B, H1, W1) - T(g["ub_mask"])).abs().max().item()
    print(f"[update block] net {e_net:.2e} dflow {e_df:.2e} mask {e_mask:.2e}")
    assert e_net < 2e-5, e_net                                  # |net| <= 1 (GRU state)
    assert e_df < 1e-4 * max(1.0, T(g["ub_dflow"]).abs().max().item()), e_df     # coords1 - before cancels ~10 px
    assert e_mask < 2e-5 * max(1.0, T(g["ub_mask"]).abs().max().item()), e_mask
    # convex upsampling of (flow, reference mask): coords = grid + flow
    up = torch.empty((B, 2, 8 * H1, 8 * W1), device="cuda")
    ops.convex_upsample(before, rows(T(g["ub_mask"])), up, B, H1, W1)
    assert (up.cpu() - T(g["up_out"])).abs().max() < 1e-4


def test_decoder_cross_attention_vs_reference_golden(model, golden_ops):
    """MemoryDecoderLayer / CrossAttentionLayer (decoder.py:62-136) against the reference's `dx_out`.  The fused kernel
    starts one step earlier (flow_token_encoder); the golden's query is injected through it exactly: W0 = [I | 0],
    b0 = +16 puts every pre-activation above the point where fp32 GELU is the identity (erf saturates), W2 = I, b2 = -16
    removes the shift (query + 16 - 16 costs one rounding at 2^-20, far below the tolerance)."""
    import stitch_amd
    ops = stitch_amd.ops
    g = golden_ops
    fb = model.flow_backbone
    D = fb.pack()["dec"]
    R, nl = 12 * 16, 8
    qy = T(g["dx_query"]).reshape(R, 64)
    mem = T(g["dx_mem"]).reshape(R * nl, 128).cuda().contiguous()
    coords = T(g["lookup_coords"]).permute(0, 2, 3, 1).reshape(R, 2).cuda().contiguous()
    kv = torch.empty((R * nl, 128), device="cuda")
    ops.conv_gemm(mem, D["ca"]["kv"][0], kv, bias=D["ca"]["kv"][1])
    w16 = list(D["chain16"])
    w0 = torch.zeros((64, 84))
    w0[:, :64] = torch.eye(64)
    w16[0], w16[1] = w0.cuda(), torch.full((64,), 16.0, device="cuda")
    w16[2], w16[3] = torch.eye(64).cuda().contiguous(), torch.full((64,), -16.0, device="cuda")
    corr = torch.zeros((R, 148), device="cuda")
    corr[:, :64] = qy.cuda()
    ops.decoder_token_chain(corr, coords, kv, w16, R, nl)
    ref = T(g["dx_out"]).permute(0, 2, 3, 1).reshape(R, 64)           # golden is NCHW [1,64,12,16]
    err = (corr[:, 84:].cpu() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


def test_flowformer_small_vs_reference_golden(model, golden_ops):
    a, b = inputs.structured_pair(96, 128, seed=3, shift=(2, -3))
    flow = model.predict_flow(a.cuda(), b.cuda())[0].cpu()
    d = (flow - T(golden_ops["ff_small_flow"])).abs()
    check("ff_small_flow_px", d.max(), 3e-2)      # measured 1.5e-2; the fp32 oracle itself is 9.6e-3 from the fp64 answer here        # full-res flow, |flow| ~ 10 px, 12 recurrent refinements


def test_flowformer_forward_surface(model):
    a, b = inputs.structured_pair(64, 96, seed=5)
    up, low = model.flow_backbone(a.cuda(), b.cuda())
    assert up.shape == (1, 2, 64, 96) and low.shape == (1, 2, 8, 12)
    with pytest.raises(RuntimeError):
        model.flow_backbone(a, b)                         # CPU tensors: loud failure, no fallback


def test_end_to_end_test_eval_512_vs_reference_golden(model):
    g = np.load(os.path.join(GOLDEN, "e2e_eval_512.npz"))
    a, b = inputs.structured_pair(512, 512, seed=7)
    o = model(a.cuda(), b.cuda(), type="test_eval")
    assert set(o) == {"output_H", "output_H_inv", "final_warp_output", "overlap", "flow_predictions", "H",
                      "origin_occlusion_mask"}
    assert o["output_H"].shape == (1, 6, 512, 512) and o["overlap"].shape == (1, 512, 512)
    assert o["origin_occlusion_mask"].shape == (1, 512, 512) and o["H"].shape == (1, 3, 3)
    # End-to-end bounds.  Stage by stage with the oracle's intermediates held fixed the path is bit-exact (homography stage,
    # flow warp, occlusion) or as close to the fp64 answer as the reference's own fp32 evaluation (flow network):
    # tests/test_parity_gpu.py.  End to end the two paths start from corner offsets that differ by ~1e-5 px (H by ~1.5e-6),
    # and the seeded random-weight flow network amplifies that: the CPU oracle itself moves by flow max 0.06 / p99 0.017 px
    # and ~800 occlusion pixels when fed the HIP offsets (profiles/r2_parity.json, oracle_sensitivity), and the HIP path
    # against ITSELF at batch 8 vs batch 1 (another summation order) by 0.12 px / 950 pixels (test_forward_batch8 below).
    H = o["H"].cpu().numpy()
    check("e2e_eval_H_rel", np.abs(H - g["H"]).max() / max(1.0, np.abs(g["H"]).max()), 4e-6)      # measured 1.2e-06
    flow = o["flow_predictions"][0].cpu()
    dflow = np.abs(flow[..., ::8, ::8].numpy() - g["flow_sub"])
    # bounds anchored on the REFERENCE's own spread on this very case (tests/golden/e2e_r5_512.npz `struct_seeded_floor_avx2_*`: the reference
    # under AVX2 MKL kernels against itself under AVX-512: flow 0.199 / 0.043 px, 1 728 occlusion pixels, 11 overlap pixels): <= 3x that
    # floor, and never looser than 3x this build's own measurement
    fl = np.load(os.path.join(GOLDEN, "e2e_r5_512.npz"))
    F = lambda k: float(fl["struct_seeded_floor_avx2_" + k])          # noqa: E731
    check("e2e_eval_flow_max_px", dflow.max(), min(0.25, 3 * F("flow_max_px")))      # measured 0.0837 (reference floor 0.199)
    check("e2e_eval_flow_p99_px", np.percentile(dflow, 99), min(5e-2, 3 * F("flow_p99_px")))      # measured 2.7e-2 (floor 4.3e-2)
    dH = np.abs(o["output_H"][..., ::8, ::8].cpu().numpy() - g["output_H_sub"])
    check("e2e_eval_output_H_p99", np.percentile(dH, 99), min(5e-3, 3 * F("output_H_p99")))      # measured 0.00169 (floor 2.7e-3)
    occ_flip = np.unpackbits(_bits(o["origin_occlusion_mask"]) ^ g["occ_bits"]).sum()
    ov_flip = np.unpackbits(_bits(o["overlap"]) ^ g["overlap_bits"]).sum()
    check("e2e_eval_occ_flips", occ_flip, min(2000, 3 * F("occ_flips")))      # measured 1.18e+03 (floor 1 728)
    check("e2e_eval_overlap_flips", ov_flip, min(8, 3 * F("overlap_flips")))      # measured 2 (floor 11)
    print(f"[e2e eval] H err {np.abs(H - g['H']).max():.2e} flow max {dflow.max():.3e} p99 {np.percentile(dflow, 99):.3e} "
          f"output_H p99 {np.percentile(dH, 99):.3e} occ flips {occ_flip} overlap flips {ov_flip}")


def test_end_to_end_test_out_256_vs_reference_golden(model):
    g = np.load(os.path.join(GOLDEN, "e2e_out_256.npz"))
    a = T(g["input1"]).permute(2, 0, 1)[None].float().cuda()
    b = T(g["input2"]).permute(2, 0, 1)[None].float().cuda()
    o = model(a, b, type="test_out")
    assert sorted(o.keys()) == list(g["keys"])
    assert [o["width_min"], o["height_min"], o["out_height"], o["out_width"]] == list(g["ints"])
    assert o["blend_image"].dtype == torch.uint8 and tuple(o["blend_image"].shape) == tuple(g["blend_image"].shape)
    d = np.abs(o["blend_image"].cpu().numpy().astype(np.int32) - g["blend_image"].astype(np.int32))
    check("e2e_out256_blend_gt2_frac", (d > 2).mean(), 1.2e-3)      # measured 0.000363
    for key, bits in [("mask1", "mask1_bits"), ("warp_input2_mask", "warp_mask_bits"), ("occlusion_mask", "occ_bits")]:
        flips = np.unpackbits(_bits(o[key]) ^ g[bits]).sum()
        check(f"e2e_out256_{key}_flip_frac", flips / o[key].numel(), 1e-4)      # measured 0 on all three masks
    assert o["residual_flow"].shape == (1, 2, 256, 256) and o["I_mat"].shape == (1, 3, 3)
    print(f"[e2e out] blend>2 frac {(d > 2).mean():.2e} mean abs {d.mean():.3f}")


@pytest.fixture(scope="module")
def damped_model():
    import stitch_amd
    cfg, _ = stitch_amd.load_inference_config("all_img1_with_inpaint_g12_transRef")
    m = stitch_amd.build_model(cfg)
    m.load_state_dict(spec.damped_state_dict(1234), strict=True)
    return m.cuda().eval()


def test_end_to_end_damped_eval_512_vs_reference_golden(damped_model):
    """The non-chaotic end-to-end case: `spec.damped_state_dict` takes the refinement loop's gain below 1
    (flow head x 0.15; the network still moves pixels by up to 6.2 px and occludes 8 143 of them), so the whole path -- images in,
    `test_eval` dict out -- can be held against the REFERENCE's output in absolute terms: north_star's warped-pixel L_inf < 1e-3 px,
    and occlusion flips <= 3x the reference's own 8-vs-1-thread floor on this very case (5 flips, flow 4.5e-4 px; recorded in the
    golden by oracle/ref_harness/make_e2e_goldens.py)."""
    g = np.load(os.path.join(GOLDEN, "e2e_eval_damped_512.npz"))
    a, b = inputs.structured_pair(512, 512, seed=7)
    o = damped_model(a.cuda(), b.cuda(), type="test_eval")
    H = o["H"].cpu().numpy()
    check("damped_e2e_H_rel", np.abs(H - g["H"]).max() / max(1.0, np.abs(g["H"]).max()), 3.6e-6)      # measured 1.2e-6
    dflow = np.abs(o["flow_predictions"][0][..., ::4, ::4].cpu().numpy() - g["flow_sub"])
    check("damped_e2e_flow_max_px", dflow.max(), 7.5e-4)                                   # north_star's bound; measured 2.5e-4 (reference vs itself: 4.5e-4)
    check("damped_e2e_flow_p99_px", np.percentile(dflow, 99), 3.6e-4)      # measured 1.2e-4
    occ_flip = np.unpackbits(_bits(o["origin_occlusion_mask"]) ^ g["occ_bits"]).sum()
    check("damped_e2e_occ_flips", occ_flip, 3 * int(g["ref_floor_occ_flips"]), inclusive=True)      # <= 3x the reference's own floor (15); measured 4
    check("damped_e2e_overlap_flips", np.unpackbits(_bits(o["overlap"]) ^ g["overlap_bits"]).sum(), 2, inclusive=True)      # measured 0
    dH = np.abs(o["output_H"][..., ::4, ::4].cpu().numpy() - g["output_H_sub"])
    check("damped_e2e_output_H_max", dH.max(), 0.05)   # grey levels; measured 0.017, the reference's own floor 0.067
    # the stitched image: identical wherever the two occlusion masks agree, up to the flow difference x the image gradient
    got, want = o["final_warp_output"][..., ::4, ::4].cpu().numpy(), g["final_sub"]
    same_mask = (got[:, 3:6] == want[:, 3:6]).all(1, keepdims=True)
    dimg = np.abs(got[:, 0:3] - want[:, 0:3]) * same_mask
    check("damped_e2e_final_max_where_masks_agree", dimg.max(), 0.04)      # measured 0.013 grey levels
    fcs = np.array([float(o["flow_predictions"][0].double().sum()), float((o["flow_predictions"][0].double() ** 2).sum())])
    check("damped_e2e_flow_checksum_rel", np.abs(fcs / g["flow_cs"] - 1).max(), 7e-6)      # measured 2.4e-6
    print(f"[damped e2e] H {np.abs(H - g['H']).max():.2e} flow max {dflow.max():.3e} p99 {np.percentile(dflow, 99):.3e} occ flips {occ_flip} "
          f"output_H max {dH.max():.3e} final max {dimg.max():.3e}")


@pytest.mark.parametrize("name", ["demo1", "demo2"])
def test_end_to_end_reference_demo_pairs_512(model, name):
    """The two real photo pairs the reference ships (demo/demo{1,2}/input{1,2}.jpg, 512x512), native size, `test_eval` and
    `test_out` against the reference's own outputs (tests/golden/e2e_demo_512.npz).  Seeded (chaotic) weights: the bounds are the
    stage bounds of the synthetic-pair tests above, the damped case is the absolute one."""
    g = np.load(os.path.join(GOLDEN, "e2e_demo_512.npz"))
    a = T(g[name + "_input1"]).permute(2, 0, 1)[None].float().cuda()
    b = T(g[name + "_input2"]).permute(2, 0, 1)[None].float().cuda()
    o = model(a, b, type="test_eval")
    p = name + "_eval_"
    H = o["H"].cpu().numpy()
    check(f"{name}_eval_H_rel", np.abs(H - g[p + "H"]).max() / max(1.0, np.abs(g[p + "H"]).max()), 3.3e-6)      # measured 1.0e-6 / 1.1e-6
    dflow = np.abs(o["flow_predictions"][0][..., ::8, ::8].cpu().numpy() - g[p + "flow_sub"])
    # every bound = min(3x the REFERENCE's own spread on this pair, 3x this build's measurement).  The spread (tests/golden/e2e_r5_512.npz,
    # oracle/ref_harness/make_r5_goldens.py + make_r5_isa_floor.py) is the larger of the reference at 8 vs 1 CPU threads and the reference
    # under AVX2 vs AVX-512 MKL kernels, same code, same inputs: demo1 flow 0.050 / 0.0064 px, 289 occlusion pixels, output_H p99 1.7e-3;
    # demo2 flow 0.39 / 0.061 px, 1 944 pixels (its 8-vs-1-thread run alone).  The damped-weights figures for the same pairs are in
    # test_demo_pairs_damped_vs_reference_golden_and_floor.
    fl = np.load(os.path.join(GOLDEN, "e2e_r5_512.npz"))
    F = lambda k: max(float(fl[f"{name}_floor_seeded_eval_{k}"]), float(fl[f"{name}_seeded_floor_avx2_{k}"]))          # noqa: E731

    def why(k):        # the reference's own spread on THIS quantity of THIS pair, quoted when a bound trips
        return (f"seeded (chaotic) weights amplify a last-bit difference ~1e4x: the reference against ITSELF on this pair moves {k} by "
                f"{float(fl[f'{name}_floor_seeded_eval_{k}']):.4g} (8 vs 1 CPU threads) and {float(fl[f'{name}_seeded_floor_avx2_{k}']):.4g} (AVX2 vs AVX-512 MKL); "
                f"bound = min(3x that floor, 3x this build's first measurement).  The gate for a change of summation order is the damped-weights "
                f"test (test_demo_pairs_damped_vs_reference_golden_and_floor: flow 2.1e-4 px, <= 4 flips)")
    check(f"{name}_eval_flow_max_px", dflow.max(), min({"demo1": 0.23, "demo2": 0.055}[name], 3 * F("flow_max_px")), note=why("flow_max_px"))      # measured 0.076 / 0.018 (demo1 / demo2)
    check(f"{name}_eval_flow_p99_px", np.percentile(dflow, 99), min({"demo1": 0.054, "demo2": 0.017}[name], 3 * F("flow_p99_px")), note=why("flow_p99_px"))      # measured 0.018 / 0.0055
    check(f"{name}_eval_output_H_p99", np.percentile(np.abs(o["output_H"][..., ::8, ::8].cpu().numpy() - g[p + "output_H_sub"]), 99), min({"demo1": 1.2e-2, "demo2": 8.4e-3}[name], 3 * F("output_H_p99")), note=why("output_H_p99"))      # measured 4.1e-3 / 2.8e-3
    check(f"{name}_eval_occ_flips", np.unpackbits(_bits(o["origin_occlusion_mask"]) ^ g[p + "occ_bits"]).sum(), min({"demo1": 2400, "demo2": 1370}[name], 3 * F("occ_flips")), note=why("occ_flips"))      # measured 785 / 457
    check(f"{name}_eval_overlap_flips", np.unpackbits(_bits(o["overlap"]) ^ g[p + "overlap_bits"]).sum(), min({"demo1": 15, "demo2": 3}[name], max(3, 3 * F("overlap_flips"))), inclusive=True, note=why("overlap_flips"))      # measured 5 / 1
    o = model(a, b, type="test_out")
    p = name + "_out_"
    assert [o["width_min"], o["height_min"], o["out_height"], o["out_width"]] == list(g[p + "ints"])      # canvas ints exact
    check(f"{name}_out_H_rel", np.abs(o["H"].cpu().numpy() - g[p + "H"]).max() / max(1.0, np.abs(g[p + "H"]).max()), {"demo1": 2.4e-6, "demo2": 1.6e-6}[name])      # measured 8e-7 / 5.3e-7
    assert np.abs(o["I_mat"].cpu().numpy() - g[p + "I_mat"]).max() < 1e-6
    d = np.abs(o["blend_image"][..., ::2, ::2].cpu().numpy().astype(np.int32) - g[p + "blend_sub"].astype(np.int32))
    check(f"{name}_out_blend_gt2_frac", (d > 2).mean(), {"demo1": 2.6e-3, "demo2": 1e-3}[name])      # measured 8.8e-4 / 3.4e-4
    drf = np.abs(o["residual_flow"][..., ::8, ::8].cpu().numpy() - g[p + "residual_flow_sub"])
    check(f"{name}_out_residual_flow_p99_px", np.percentile(drf, 99), {"demo1": 0.054, "demo2": 0.017}[name])      # measured 0.018 / 0.0055
    for key, bits in [("mask1", "mask1_bits"), ("warp_input2_mask", "warp_mask_bits")]:
        check(f"{name}_out_{key}_flip_frac", np.unpackbits(_bits(o[key]) ^ g[p + bits]).sum() / o[key].numel(), 1e-4)
    for key, bits in [("mask2", "mask2_bits"), ("occlusion_mask", "occ_bits"), ("origin_occlusion_mask", "origin_occ_bits")]:
        check(f"{name}_out_{key}_flip_frac", np.unpackbits(_bits(o[key]) ^ g[p + bits]).sum() / o[key].numel(), 1.5e-3)      # measured <= 5.1e-4 (demo1), 0 (demo2)


def _floor(g, prefix, key, floors=("floor_", "floor_avx2_")):
    """the reference's own spread on this case: the larger of its 8-vs-1-thread run and its AVX2-vs-AVX-512 MKL run (tests/golden/e2e_r5_512.npz,
    written by oracle/ref_harness/make_r5_goldens.py and make_r5_isa_floor.py)."""
    vals = [float(g[prefix + f + key]) for f in floors if (prefix + f + key) in g.files]
    assert vals, (prefix, key)
    return max(vals)


def _eval_errors(o, g, p, i, s):
    """errors of sample i of a HIP `test_eval` dict against the golden record with prefix p (sub-sampling s)."""
    sl = slice(i, i + 1)
    H, gH = o["H"][sl].cpu().numpy(), g[p + "H"][sl]
    dflow = np.abs(o["flow_predictions"][0][sl, :, ::s, ::s].cpu().numpy() - g[p + "flow_sub"][sl])
    dH = np.abs(o["output_H"][sl, :, ::s, ::s].cpu().numpy() - g[p + "output_H_sub"][sl])
    B = g[p + "H"].shape[0]
    n = 512 * 512
    occ_g = np.unpackbits(g[p + "occ_bits"])[: B * n].reshape(B, n)[i]
    ov_g = np.unpackbits(g[p + "overlap_bits"])[: B * n].reshape(B, n)[i]
    occ = (o["origin_occlusion_mask"][i].cpu().numpy().reshape(-1) >= 0.5).astype(np.uint8)
    ov = (o["overlap"][i].cpu().numpy().reshape(-1) >= 0.5).astype(np.uint8)
    return dict(H_rel=np.abs(H - gH).max() / max(1.0, np.abs(gH).max()), flow_max=dflow.max(), flow_p99=np.percentile(dflow, 99),
                output_H_p99=np.percentile(dH, 99), occ_flips=int((occ != occ_g).sum()), overlap_flips=int((ov != ov_g).sum()))


def test_batch2_and_batch8_eval_vs_reference_golden(damped_model):
    """BASELINE configs[2] against the ORACLE, not against the path itself: the reference's own `test_eval` on a BATCH OF
    TWO structured pairs with the damped weights (`b2_*` of tests/golden/e2e_r5_512.npz; evaluate.py:34-43 is the caller that batches).
    HIP(B = 2) and samples 0-1 of HIP(B = 8) must both reproduce it.  Bounds: the damped case's absolute bounds
    (test_end_to_end_damped_eval_512_vs_reference_golden) or 3x the reference's own spread on this batch where that is larger -- the reference
    itself moves by 5.7e-3 px / 25 occlusion pixels between ITS batch-2 and batch-1 forwards of the same pair (`b2_vs_b1_*`: MKL picks
    another blocking for the taller matrices), which is the floor for "batched = unbatched" on either side."""
    g = np.load(os.path.join(GOLDEN, "e2e_r5_512.npz"))
    p0 = inputs.structured_pair(512, 512, seed=int(g["b2_seeds"][0]))
    p1 = inputs.structured_pair(512, 512, seed=int(g["b2_seeds"][1]), shift=tuple(int(v) for v in g["b2_shift1"]))
    extra = [inputs.structured_pair(512, 512, seed=40 + i, shift=(2 * i - 7, 5 - i)) for i in range(6)]
    A2, B2 = torch.cat([p0[0], p1[0]]).cuda(), torch.cat([p0[1], p1[1]]).cuda()
    A8, B8 = torch.cat([A2] + [e[0].cuda() for e in extra]), torch.cat([B2] + [e[1].cuda() for e in extra])
    # floor of "a batched forward": the reference's batch-2 forward against ITS OWN batch-1 forward of sample 0 (b2_vs_b1_*: 5.7e-3 px, 25
    # occlusion pixels -- MKL blocks the taller matrices differently), next to its 8-vs-1-thread and AVX2-vs-AVX-512 spread on the batch
    ref_b2_vs_b1 = float(g["b2_vs_b1_flow_max_px"])
    f_flow_max = 3 * max(ref_b2_vs_b1, _floor(g, "b2_", "flow_max_px", floors=("floor_", "damped_floor_avx2_")))
    f_occ = 3 * max(int(g["b2_vs_b1_occ_flips"]), int(_floor(g, "b2_", "occ_flips", floors=("floor_", "damped_floor_avx2_"))))
    g1 = np.load(os.path.join(GOLDEN, "e2e_eval_damped_512.npz"))            # the reference's batch-1 forward of sample 0
    for tag, (A, Bt) in (("b2", (A2, B2)), ("b8", (A8, B8))):
        o = damped_model(A, Bt, type="test_eval")
        for i in (0, 1):
            e = _eval_errors(o, g, "b2_", i, 4)
            print(f"[{tag} sample {i} vs reference B=2 golden] " + " ".join(f"{k} {v:.3g}" for k, v in e.items()))
            check(f"{tag}_vs_ref_s{i}_H_rel", e["H_rel"], 3.6e-6)
            check(f"{tag}_vs_ref_s{i}_flow_max_px", e["flow_max"], f_flow_max)                       # measured 5.7e-3 = the reference's own b2-vs-b1 figure
            check(f"{tag}_vs_ref_s{i}_flow_p99_px", e["flow_p99"], ref_b2_vs_b1)                     # p99 below the reference's own maximum
            check(f"{tag}_vs_ref_s{i}_occ_flips", e["occ_flips"], f_occ, inclusive=True)
            check(f"{tag}_vs_ref_s{i}_overlap_flips", e["overlap_flips"], 3, inclusive=True)
            check(f"{tag}_vs_ref_s{i}_output_H_p99", e["output_H_p99"], max(1e-3, 3 * _floor(g, "b2_", "output_H_p99", floors=("floor_", "damped_floor_avx2_"))))
        # the batched HIP forward reproduces the reference's UNBATCHED forward of sample 0 to the damped case's absolute bounds: the 5.7e-3 px
        # above is the reference's batch dependence, not this path's
        dflow = np.abs(o["flow_predictions"][0][0:1, :, ::4, ::4].cpu().numpy() - g1["flow_sub"])
        occ1 = np.unpackbits(_bits(o["origin_occlusion_mask"][0:1]) ^ g1["occ_bits"]).sum()
        print(f"[{tag} sample 0 vs reference B=1 golden] flow max {dflow.max():.3g} p99 {np.percentile(dflow, 99):.3g} occ flips {occ1}")
        check(f"{tag}_s0_vs_ref_b1_flow_max_px", dflow.max(), 1e-3)                                 # north_star's figure
        check(f"{tag}_s0_vs_ref_b1_occ_flips", occ1, 3 * int(g1["ref_floor_occ_flips"]), inclusive=True)


@pytest.mark.parametrize("name", ["demo1", "demo2"])
def test_demo_pairs_damped_vs_reference_golden_and_floor(damped_model, name):
    """The reference's two photo pairs with the DAMPED weights (the non-chaotic case), `test_eval`, against the reference's own output and
    ITS OWN spread on the same pair: every bound is 3x the reference-vs-itself figure (8 vs 1 threads, and AVX2 vs
    AVX-512 MKL kernels: `demo*_damped_floor_*`), with north_star's absolute figure as the lower limit of a bound where the floor is below it."""
    g = np.load(os.path.join(GOLDEN, "e2e_r5_512.npz"))
    gd = np.load(os.path.join(GOLDEN, "e2e_demo_512.npz"))
    a = T(gd[name + "_input1"]).permute(2, 0, 1)[None].float().cuda()
    b = T(gd[name + "_input2"]).permute(2, 0, 1)[None].float().cuda()
    o = damped_model(a, b, type="test_eval")
    p = name + "_damped_"
    e = _eval_errors(o, g, p, 0, 8)
    print(f"[{name} damped vs reference] " + " ".join(f"{k} {v:.3g}" for k, v in e.items()) + " | reference floor (max of 8v1 threads, AVX2): " +
          " ".join(f"{k} {_floor(g, p, k):.3g}" for k in ("flow_max_px", "flow_p99_px", "occ_flips", "output_H_p99", "H_rel")))
    check(f"{name}_damped_H_rel", e["H_rel"], max(3.6e-6, 3 * _floor(g, p, "H_rel")))
    check(f"{name}_damped_flow_max_px", e["flow_max"], max(1e-3, 3 * _floor(g, p, "flow_max_px")))
    check(f"{name}_damped_flow_p99_px", e["flow_p99"], max(3.6e-4, 3 * _floor(g, p, "flow_p99_px")))
    check(f"{name}_damped_occ_flips", e["occ_flips"], max(15, 3 * int(_floor(g, p, "occ_flips"))), inclusive=True)
    check(f"{name}_damped_overlap_flips", e["overlap_flips"], max(2, 3 * int(_floor(g, p, "overlap_flips"))), inclusive=True)
    check(f"{name}_damped_output_H_p99", e["output_H_p99"], max(1e-3, 3 * _floor(g, p, "output_H_p99")))


def test_forward_batch8_matches_single_pairs(model):
    """BASELINE configs[2]: one `test_eval` forward over 8 pairs.  Samples are independent on the path (no cross-sample
    op), so every sample of the batched forward must reproduce the batch-1 forward of the same pair.  GEMM tiles see other
    rows of the batch but each output row's k order is the same: the homography stage is bit-identical; split-K choices
    depend on M, so the flow may differ by reorder noise that the refinements amplify (bounds as in test_parity_gpu)."""
    pairs = [inputs.structured_pair(512, 512, seed=40 + i, shift=(2 * i - 7, 5 - i)) for i in range(8)]
    A = torch.cat([p[0] for p in pairs]).cuda()
    Bt = torch.cat([p[1] for p in pairs]).cuda()
    o8 = model(A, Bt, type="test_eval")
    assert o8["final_warp_output"].shape == (8, 6, 512, 512) and o8["H"].shape == (8, 3, 3)
    assert torch.isfinite(o8["final_warp_output"]).all()
    worst = dict(H=0.0, flow_max_px=0.0, flow_p99_px=0.0, occ_flips=0.0)          # one bound per quantity = 3x its maximum over the samples
    for i in (0, 3, 7):
        o1 = model(A[i:i + 1], Bt[i:i + 1], type="test_eval")
        dH = (o8["H"][i] - o1["H"][0]).abs().max().item()
        d = (o8["flow_predictions"][0][i] - o1["flow_predictions"][0][0]).abs()
        flips = int((o8["origin_occlusion_mask"][i] != o1["origin_occlusion_mask"][0]).sum())
        print(f"[batch 8 vs 1, sample {i}] H {dH:.2e} flow max {d.max().item():.3e} p99 {np.percentile(d.cpu().numpy(), 99):.3e} occ flips {flips}")
        for k, v in (("H", dH), ("flow_max_px", d.max().item()), ("flow_p99_px", np.percentile(d.cpu().numpy(), 99)), ("occ_flips", flips)):
            worst[k] = max(worst[k], float(v))
    check("batch8_vs_1_H", worst["H"], 3e-6)                       # measured 9.5e-7 (sample 3)
    check("batch8_vs_1_flow_max_px", worst["flow_max_px"], 0.34)   # measured 0.114 (sample 3)
    check("batch8_vs_1_flow_p99_px", worst["flow_p99_px"], 0.06)   # measured 0.0195
    check("batch8_vs_1_occ_flips", worst["occ_flips"], 2800)       # measured 952


def test_graph_replay_and_concurrent_streams_match_eager(model):
    """hipGraph replay (own split-K workspace per graph) == eager, also with 3 pairs in flight."""
    pairs = [inputs.structured_pair(512, 512, seed=20 + i) for i in range(3)]
    eager = [model(a.cuda(), b.cuda(), type="test_eval")["final_warp_output"].clone() for a, b in pairs]
    graphs = [model.graphed("test_eval") for _ in range(3)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    for rep in range(2):
        outs = []
        for i, (a, b) in enumerate(pairs):
            streams[i].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(streams[i]):
                outs.append(graphs[i](a.cuda(non_blocking=True), b.cuda(non_blocking=True))["final_warp_output"])
        torch.cuda.synchronize()
        for e, o in zip(eager, outs):
            assert torch.equal(e, o)


def test_graphed_test_out_matches_eager(model):
    """network part of `test_out` replayed from a hipGraph + eager canvas part == the eager forward, bit for bit, also when
    two pairs are launched before either is finished (the bench's software pipeline)."""
    pairs = [inputs.structured_pair(320, 416, seed=70 + i, shift=(4 - 3 * i, 2 * i - 5)) for i in range(2)]
    eager = [model(a.cuda(), b.cuda(), type="test_out") for a, b in pairs]
    gs = [model.graphed_test_out() for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    for rep in range(2):
        handles = []
        for i, (a, b) in enumerate(pairs):
            streams[i].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(streams[i]):
                handles.append(gs[i].launch(a.cuda(), b.cuda()))
        outs = [gs[i].finish(h) for i, h in enumerate(handles)]
        torch.cuda.synchronize()
        for e, o in zip(eager, outs):
            assert [o[k] for k in ("width_min", "height_min", "out_height", "out_width")] == [e[k] for k in ("width_min", "height_min", "out_height", "out_width")]
            for k in ("blend_image", "output2", "mask2", "residual_flow", "occlusion_mask", "H"):
                assert torch.equal(e[k], o[k]), k


def test_forward_accepts_cpu_input_tensors(model):
    """The reference's callers hand `forward` whatever the loader produced -- CPU float tensors through nn.DataParallel (out.py:80,197,
    evaluate.py:43,119).  The drop-in moves them to the module's device like DataParallel's scatter; the result is the
    result of the same call on device tensors, bit for bit."""
    a, b = inputs.structured_pair(304, 400, seed=91, shift=(2, -3))
    assert not a.is_cuda
    ref = model(a.cuda(), b.cuda(), type="test_out")
    got = model(a, b, type="test_out")                       # out.py:197's call shape
    for k in ("blend_image", "output2", "mask2", "residual_flow", "H"):
        assert got[k].is_cuda and torch.equal(got[k], ref[k]), k
    assert [got[k] for k in ("width_min", "height_min", "out_height", "out_width")] == [ref[k] for k in ("width_min", "height_min", "out_height", "out_width")]
    a5, b5 = inputs.structured_pair(512, 512, seed=92)
    e_ref, e_got = model(a5.cuda(), b5.cuda(), type="test_eval"), model(a5, b5, type="test_eval")      # evaluate.py:43
    assert torch.equal(e_got["final_warp_output"], e_ref["final_warp_output"]) and torch.equal(e_got["H"], e_ref["H"])
