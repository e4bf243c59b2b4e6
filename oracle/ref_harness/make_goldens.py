"""Generate the golden vectors under tests/golden from the REFERENCE's own Python (build container only).

    python -m oracle.ref_harness.make_goldens

The reference (``/root/reference``) is imported on CPU through the stand-ins of ``stubs.py`` and driven
with ``oracle.spec.seeded_state_dict(1234)`` (loaded ``strict=True``).  Only data is written: inputs,
expected outputs, checksums.  Nothing here runs on the GPU box.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from oracle import inputs, spec
from oracle.ref_harness import stubs

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden")


def packbits(t):
    return np.packbits((t.detach().numpy() >= 0.5).astype(np.uint8).reshape(-1))


def sub(t, s=8):
    return t[..., ::s, ::s].contiguous().numpy()


def checksum(t):
    t = t.detach().double()
    return np.array([float(t.sum()), float((t * t).sum())])


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    sd = spec.seeded_state_dict(1234)
    model, cfg = stubs.build_reference(sd, overlay=dict(
        test_not_use_combine_h_flow=True, use_forward=False, use_fb_consistency_mask=True,
        use_whole_resolution=False))
    import importlib
    ref = lambda m: importlib.import_module(m)
    g = torch.Generator().manual_seed(99)
    rn = lambda *s: torch.randn(*s, generator=g)

    # ---- checkpoint key set (out.py:85 strict load) -------------------------------------------
    keys = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()}
    json.dump(keys, open(os.path.join(OUT, "state_keys.json"), "w"), indent=0, sort_keys=True)

    ops = {}
    with torch.no_grad():
        # ---- geometric ops ------------------------------------------------------------------
        dlt = ref("core.udis_utils.torch_DLT")
        src = torch.tensor([[0., 0.], [64., 0.], [0., 48.], [64., 48.]])[None].repeat(5, 1, 1)
        dst = src + 6 * rn(5, 4, 2)
        ops["dlt_src"], ops["dlt_dst"], ops["dlt_H"] = src, dst, dlt.tensor_DLT(src, dst)

        tht = ref("core.udis_utils.torch_homo_transform")
        U = torch.rand(2, 6, 40, 56, generator=g) * 255
        theta = torch.eye(3)[None].repeat(2, 1, 1) + 0.15 * rn(2, 3, 3)
        ops["homo_U"], ops["homo_theta"] = U, theta
        ops["homo_out"] = tht.transformer(U, theta, (33, 47))

        tps = ref("core.udis_utils.torch_tps_transform")
        gy, gx = torch.meshgrid(torch.linspace(-1, 1, 13), torch.linspace(-1, 1, 13), indexing="ij")
        tgt = torch.stack([gx, gy], -1).reshape(1, 169, 2).repeat(2, 1, 1)
        srcp = tgt + 0.03 * rn(2, 169, 2)
        Ut = torch.rand(2, 3, 32, 40, generator=g) * 255
        ops["tps_U"], ops["tps_source"], ops["tps_target"] = Ut, srcp, tgt
        ops["tps_out"] = tps.transformer(Ut, srcp, tgt, (24, 28))

        wu = ref("core.warp_utils")
        x = torch.rand(2, 6, 48, 64, generator=g) * 255
        fij, fji = 4 * rn(2, 2, 48, 64), 4 * rn(2, 2, 48, 64)
        ops["warp_x"], ops["flow_ij"], ops["flow_ji"] = x, fij, fji
        ops["warp_out"] = wu.warp(x, fij)
        ops["resize_flow_out"] = wu.resize_flow(fij.clone(), (60, 100))
        ops["range_map"] = wu.compute_range_map(fji)
        ops["occlusion"] = wu.compute_occlusion(fij, fji, "wang", occlusion_are_zeros=True, boundaries_occluded=True)
        fha = ref("core.flowHomoAdpater")
        msk = (torch.rand(1, 1, 96, 80, generator=g) > 0.02).float()
        msk[:, :, 30:40, 10:70] = 0
        ops["open_in"], ops["open_out"] = msk, fha.preprocess_occlusion_mask(msk)
        Hm = torch.eye(3)[None] + torch.tensor([[[0.05, 0.02, 9.0], [-0.03, 0.04, -6.0], [1e-4, -2e-4, 0.0]]])
        mesh = wu.H2Mesh(Hm, wu.get_rigid_mesh(1, 300, 400))
        ops["mesh_H"] = Hm
        ops["mesh_minmax"] = torch.stack([mesh[..., 0].min(), mesh[..., 0].max(), mesh[..., 1].min(), mesh[..., 1].max()])
        ops["resize512_in"] = torch.rand(1, 3, 40, 72, generator=g) * 255
        ops["resize512_out"] = fha.resize_512(ops["resize512_in"])[..., ::16, ::16].contiguous()

        # ---- network blocks -----------------------------------------------------------------
        hb, fb = model.homo_backbone, model.flow_backbone
        f1, f2 = rn(2, 64, 8, 8), rn(2, 64, 8, 8)
        ops["ccl_f1"], ops["ccl_f2"], ops["ccl_out"] = f1, f2, hb.CCL(f1, f2)
        cf = 3 * rn(1, 2, 32, 32)
        ops["regress_in"] = cf
        ops["regress_out"] = hb.regressNet1_part2(hb.regressNet1_part1(cf).view(1, -1))
        im = torch.rand(1, 3, 64, 96, generator=g) * 2 - 1
        ops["res_in"] = im
        s1 = hb.feature_extractor_stage1(im)
        ops["res_stage1"], ops["res_stage2"] = s1, hb.feature_extractor_stage2(s1)

        enc = fb.memory_encoder
        ops["twins_in"] = im
        ops["twins_out"] = enc.feat_encoder(im)[0]
        a, b = rn(1, 256, 6, 8), rn(1, 256, 6, 8)
        ops["corr_f1"], ops["corr_f2"], ops["corr_out"] = a, b, enc.corr(a, b)
        cpe = enc.cost_perceiver_encoder
        cm = 8 * rn(8, 1, 64, 64)
        ops["pe_in"], ops["pe_out"] = cm, cpe.patch_embed(cm)[0]
        tok = rn(6, 64, 128)
        ops["xattn_tokens"] = tok
        ops["xattn_out"] = cpe.input_layer(cpe.latent_tokens, tok, (8, 8))
        lat = rn(6, 8, 128)
        ops["sattn_in"], ops["sattn_out"] = lat, cpe.encoder_layers[1](lat)
        vx, vctx = rn(8, 12 * 16, 128), rn(1, 256, 12, 16)
        ops["vert_x"], ops["vert_ctx"] = vx, vctx
        ops["vert_out"] = cpe.vertical_encoder_layers[2](vx, (12, 16), vctx)

        dec = fb.memory_decoder
        cmaps = 4 * rn(12 * 16, 1, 12, 16)
        coords = ref("core.utils.utils").coords_grid(1, 12, 16) + 1.5 * rn(1, 2, 12, 16)
        ops["lookup_maps"], ops["lookup_coords"] = cmaps, coords
        ops["lookup_out"] = dec.encode_flow_token(cmaps, coords)
        inp = torch.relu(rn(1, 128, 12, 16))
        ops["gma_inp"], att = inp, dec.att(inp)
        ops["gma_attn"] = att
        net, corr, flow = torch.tanh(rn(1, 128, 12, 16)), rn(1, 145, 12, 16), 2 * rn(1, 2, 12, 16)
        ops["ub_net"], ops["ub_corr"], ops["ub_flow"] = net, corr, flow
        n2, mask, dflow = dec.update_block(net, inp, corr, flow, att)
        ops["ub_net_out"], ops["ub_mask"], ops["ub_dflow"] = n2, mask, dflow
        ops["up_out"] = dec.upsample_flow(flow, mask)
        mem, qy = rn(12 * 16, 8, 128), rn(12 * 16, 1, 64)
        ops["dx_mem"], ops["dx_query"] = mem, qy
        cg, _, _ = dec.decoder_layer(qy, None, None, mem, coords, (1, 128, 12, 16), (1, 1))
        ops["dx_out"] = cg

        # ---- small end-to-end FlowFormer (128x96) -------------------------------------------
        sa, sb = inputs.structured_pair(96, 128, seed=3, shift=(2, -3))
        ops["ff_small_flow"] = model.predict_flow(sa, sb)[0]
    np.savez_compressed(os.path.join(OUT, "ops_small.npz"), **{k: v.detach().numpy() for k, v in ops.items()})

    # ---- end-to-end: test_eval @512 on the structured synthetic pair ----------------------------
    with torch.no_grad():
        a, b = inputs.structured_pair(512, 512, seed=7)
        o = model(a, b, type="test_eval")
    e = dict(H=o["H"].numpy(), flow_sub=sub(o["flow_predictions"][0]), flow_cs=checksum(o["flow_predictions"][0]),
             output_H_sub=sub(o["output_H"]), output_H_cs=checksum(o["output_H"]),
             output_H_inv_cs=checksum(o["output_H_inv"]),
             final_sub=sub(o["final_warp_output"]), final_cs=checksum(o["final_warp_output"]),
             overlap_bits=packbits(o["overlap"]), occ_bits=packbits(o["origin_occlusion_mask"]))
    np.savez_compressed(os.path.join(OUT, "e2e_eval_512.npz"), **e)

    # ---- end-to-end: test_out on demo1 resized to 256x256 (BASELINE.json configs[0]) ---------------
    from PIL import Image

    def load(p):
        im = Image.open(p).convert("RGB").resize((256, 256), Image.BILINEAR)
        return np.asarray(im).copy()
    i1 = load(stubs.REF_ROOT + "/demo/demo1/input1.jpg")
    i2 = load(stubs.REF_ROOT + "/demo/demo1/input2.jpg")
    with torch.no_grad():
        ta = torch.from_numpy(i1).permute(2, 0, 1)[None].float()
        tb = torch.from_numpy(i2).permute(2, 0, 1)[None].float()
        o = model(ta, tb, type="test_out")
    e = dict(input1=i1, input2=i2, blend_image=o["blend_image"].numpy(), H=o["H"].numpy(), I_mat=o["I_mat"].numpy(),
             ints=np.array([o["width_min"], o["height_min"], o["out_height"], o["out_width"]]),
             residual_flow_sub=sub(o["residual_flow"], 4), residual_flow_cs=checksum(o["residual_flow"]),
             mask1_bits=packbits(o["mask1"]), mask2_bits=packbits(o["mask2"]),
             occ_bits=packbits(o["occlusion_mask"]), origin_occ_bits=packbits(o["origin_occlusion_mask"]),
             warp_mask_bits=packbits(o["warp_input2_mask"]),
             output2_cs=checksum(o["output2"]), final_warp_cs=checksum(o["final_warp"]), H_warp_cs=checksum(o["H_warp"]),
             keys=np.array(sorted(o.keys())))
    np.savez_compressed(os.path.join(OUT, "e2e_out_256.npz"), **e)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
