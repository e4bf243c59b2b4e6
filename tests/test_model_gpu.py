"""Network- and path-level parity on the GPU: the HIP-backed drop-in modules against the CPU oracle and
against the golden vectors produced by the reference (tests/golden), same seeded weights, same inputs.

End-to-end tolerances are quoted next to the reference's own self-noise (8 vs 1 CPU
threads changes flow by 1.1e-4 px, output_H by 8.8e-3, flips 3 of 262144 occlusion pixels)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

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
    assert err < 1e-4 * max(1.0, ref.abs().max().item()), err
    cf = T(golden_ops["regress_in"])
    xr = torch.zeros((32 * 32, 4), device="cuda")
    xr[:, :2] = cf.permute(0, 2, 3, 1).reshape(-1, 2).cuda()
    off = hb.regress(xr, 1, 32, 32)
    assert (off.cpu() - T(golden_ops["regress_out"])).abs().max() < 2e-3


def test_homography_offsets_512(model, seeded_sd):
    a, b = inputs.structured_pair(512, 512, seed=7)
    ref = nets.homo_offsets(nets.W(seeded_sd, "homo_backbone."), a, b)
    got = model.predict_homo(a.cuda(), b.cuda()).cpu()
    assert (got - ref).abs().max() < 2e-2, (got - ref).abs().max()      # corner offsets in px (|offset| ~ 10-30)


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
    assert (rows_to_nchw(f, 1, h, w) - ref).abs().max() < 5e-4 * max(1.0, ref.abs().max().item())


def test_cost_encoder_blocks(model, golden_ops, seeded_sd):
    fb = model.flow_backbone
    fb.pack()
    g = golden_ops
    cm = T(g["pe_in"])                                     # [8,1,64,64]
    tok, P = fb._patch_embed(cm.reshape(8, -1).cuda().contiguous(), 8, 64, 64)
    assert P == 64 and (tok.cpu().view(8, 64, 128) - T(g["pe_out"])).abs().max() < 5e-4
    x = fb._latent_layer(fb._pk["xin"], None, 6, True, T(g["xattn_tokens"]).reshape(-1, 128).cuda().contiguous(), 64)
    assert (x.cpu().view(6, 8, 128) - T(g["xattn_out"])).abs().max() < 5e-4
    y = fb._latent_layer(fb._pk["self"][1], T(g["sattn_in"]).reshape(-1, 128).cuda().contiguous(), 6, False)
    assert (y.cpu().view(6, 8, 128) - T(g["sattn_out"])).abs().max() < 5e-4
    vx, vctx = T(g["vert_x"]), T(g["vert_ctx"])            # [8 latents, 192 px, 128], [1,256,12,16]
    xr = vx.permute(1, 0, 2).reshape(-1, 128).cuda().contiguous()      # rows (n, l)
    ctx = vctx.permute(0, 2, 3, 1).reshape(-1, 256).cuda().contiguous()
    out = fb._vertical(fb._pk["vert"][2], xr, ctx, 1, 12, 16, 8)
    got = out.cpu().view(192, 8, 128).permute(1, 0, 2)
    assert (got - T(g["vert_out"])).abs().max() < 1e-3, (got - T(g["vert_out"])).abs().max()


def test_flowformer_small_vs_reference_golden(model, golden_ops):
    a, b = inputs.structured_pair(96, 128, seed=3, shift=(2, -3))
    flow = model.predict_flow(a.cuda(), b.cuda())[0].cpu()
    d = (flow - T(golden_ops["ff_small_flow"])).abs()
    assert d.max() < 2e-2, d.max()        # full-res flow, |flow| ~ 10 px, 12 recurrent refinements


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
    H = o["H"].cpu().numpy()
    assert np.abs(H - g["H"]).max() < 5e-3 * max(1.0, np.abs(g["H"]).max())
    flow = o["flow_predictions"][0].cpu()
    dflow = np.abs(flow[..., ::8, ::8].numpy() - g["flow_sub"])
    assert dflow.max() < 0.5 and np.percentile(dflow, 99) < 5e-2, (dflow.max(), np.percentile(dflow, 99))
    dH = np.abs(o["output_H"][..., ::8, ::8].cpu().numpy() - g["output_H_sub"])
    assert np.percentile(dH, 99) < 0.5, np.percentile(dH, 99)
    occ_flip = np.unpackbits(_bits(o["origin_occlusion_mask"]) ^ g["occ_bits"]).sum()
    ov_flip = np.unpackbits(_bits(o["overlap"]) ^ g["overlap_bits"]).sum()
    assert occ_flip < 0.01 * 512 * 512 and ov_flip < 0.01 * 512 * 512, (occ_flip, ov_flip)
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
    assert (d > 2).mean() < 0.02, (d > 2).mean()
    for key, bits in [("mask1", "mask1_bits"), ("warp_input2_mask", "warp_mask_bits"), ("occlusion_mask", "occ_bits")]:
        flips = np.unpackbits(_bits(o[key]) ^ g[bits]).sum()
        assert flips < 0.02 * o[key].numel(), (key, flips)
    assert o["residual_flow"].shape == (1, 2, 256, 256) and o["I_mat"].shape == (1, 3, 3)
    print(f"[e2e out] blend>2 frac {(d > 2).mean():.2e} mean abs {d.mean():.3f}")


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
