"""Golden vectors of the TPS post-pipeline from the REFERENCE's own functions (build container only).

    python -m oracle.ref_harness.make_tps_goldens

Imports /root/reference/core/inference/{tps_pipline,sample_point_methods,utils}.py and tps_methods/kornia_tps.py on CPU.
Third-party names those files import are stood in for (restatements of published behaviour, not reference code):
  cv2.getStructuringElement / erode / dilate   binary min / max filter over the in-image window (OpenCV default border)
  torchvision.transforms.functional.crop        tensor slicing
  kornia get_tps_transform / warp_points_tps / create_meshgrid     kornia's published definitions (oracle/tps_pipeline.py)
so the fixtures pin everything the reference itself wrote (preprocess, border sampling, point pairs, the "kornia" branch of
warp_by_tps with the in-tree warp_image_tps, mask clean-up / mix / blend of tps_H_warp with inpaint_fn=None); against
kornia / cv2 themselves parity is unpinned.  Inputs are synthetic (seeded); only data is written (tests/golden/tps_pipeline.npz).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

from oracle import tps_pipeline as otp
from oracle.ref_harness import stubs

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden")


def install_inference_stubs():
    stubs.install()
    import scipy.ndimage as ndi
    cv2 = sys.modules["cv2"]
    cv2.MORPH_RECT = 0
    cv2.getStructuringElement = lambda shape, ksize: np.ones((ksize[1], ksize[0]), np.uint8)
    cv2.erode = lambda img, k: ndi.minimum_filter(img, footprint=k.astype(bool), mode="constant", cval=np.inf)
    def _dilate(img, k, iterations=1):
        if img.ndim == 3:
            return np.stack([_dilate(img[..., c], k, iterations) for c in range(img.shape[2])], -1)
        for _ in range(iterations):
            img = ndi.maximum_filter(img, footprint=k.astype(bool), mode="constant", cval=0 if img.dtype == np.uint8 else -np.inf)
        return img
    cv2.dilate = _dilate
    tv = sys.modules["torchvision"]
    tvf = stubs._mod("torchvision.transforms.functional",
                     crop=lambda img, top, left, height, width: img[..., top:top + height, left:left + width])
    tv.transforms.functional = tvf
    k = stubs._mod("kornia")
    k.geometry = stubs._mod("kornia.geometry")
    k.geometry.transform = stubs._mod("kornia.geometry.transform", get_tps_transform=otp.get_tps_transform,
                                      warp_points_tps=otp.warp_points_tps, warp_image_tps=None)
    k.utils = stubs._mod("kornia.utils", create_meshgrid=lambda h, w, device=None, dtype=None: otp.create_meshgrid(h, w))
    k.core = stubs._mod("kornia.core", Tensor=torch.Tensor)


def cs(t):
    t = t.detach().double()
    return np.array([float(t.sum()), float((t * t).sum())])


def main():
    install_inference_stubs()
    import contextlib
    import io
    from core.inference import tps_pipline as ref_tp
    from core.inference import utils as ref_u
    from core.inference.sample_point_methods import advanced_uniform_sample_border_points as ref_sample
    from core.inference.tps_methods import kornia_tps as ref_k
    cfg = stubs.AttrDict(dict(grid_h=12, grid_w=12, pad_num=4, residual_flow_use_forward=False, flow_limit=-1, add_corner=False,
                              get_pt_methods=["advanced_uniform_multi"], add_meshgrid=False, affine_scale=1.0, kernel_scale=1.0,
                              use_boundary_limit=False, tps_method="kornia", output2_is_only_tps=True, do_avg_pooling=True))
    out = {}
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        ih, iw, wmin, hmin, oh, ow = 200, 264, -21, -13, 236, 300
        case = otp.synthetic_case(5, ih, iw, wmin, hmin, oh, ow)
        # --- the pieces
        fl = ref_tp.preprocess(case["residual_flow"].clone(), None, do_avg_pooling=True, residual_flow_use_forward=False, grid_h=12, grid_w=12)
        out["pre_flow_out_sub"], out["pre_flow_out_cs"] = fl[..., ::3, ::3].contiguous().numpy(), cs(fl)
        crop = case["H_warp"][:, :, abs(hmin):abs(hmin) + ih, abs(wmin):abs(wmin) + iw]
        for pad in (4, 22, 44):
            out[f"sample_pts_pad{pad}"] = ref_sample(crop, step=max(ih, iw) // 12, pad_num=pad).numpy()
        bp = ref_sample(crop, step=22, pad_num=4)
        s, t = ref_u.get_point_pairs(bp, fl, -1)
        out["pairs_src"], out["pairs_tgt"] = s.numpy(), t.numpy()
        s2, t2 = ref_u.get_point_pairs(bp, fl * 8, 20)                     # exercises the flow-limit filter
        out["pairs_lim_src"], out["pairs_lim_tgt"] = s2.numpy(), t2.numpy()
        bs, bd = ref_u.boundary_src_and_tgt(s.float() * 1.2 - 10, t * 1.2 - 10, t, out_height=oh, out_width=ow)
        out["bound_src"], out["bound_dst"] = bs.numpy(), bd.numpy()
        # --- the in-tree TPS warp on a small canvas
        img = case["H_warp"][:, :, ::2, ::2].contiguous()                 # tests regenerate it from the seed
        ps = torch.tensor([[[0.1, 0.1], [0.8, 0.15], [0.2, 0.7], [0.75, 0.8], [0.5, 0.45], [0.35, 0.3]]])
        pd = ps + torch.tensor([[[0.02, -0.01], [-0.015, 0.02], [0.01, 0.015], [-0.02, -0.01], [0.0, 0.02], [0.012, -0.02]]])
        kw, aw = ref_k.get_tps_transform(pd, ps)
        out["tps_ps"], out["tps_pd"] = ps.numpy(), pd.numpy()
        out["tps_kw"], out["tps_aw"] = kw.numpy(), aw.numpy()
        tw = ref_k.warp_image_tps(img, ps, kw, aw, align_corners=False)
        out["tps_warp_sub"], out["tps_warp_cs"] = tw[..., ::2, ::2].contiguous().numpy(), cs(tw)
        # --- the whole pipeline, inpaint_fn=None
        for name, seed, dims in (("a", 5, (200, 264, -21, -13, 236, 300)), ("b", 9, (160, 176, 0, -30, 211, 190))):
            ih, iw, wmin, hmin, oh, ow = dims
            case = otp.synthetic_case(seed, ih, iw, wmin, hmin, oh, ow)
            inputs = types.SimpleNamespace(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in case.items()})
            limit = types.SimpleNamespace(width_min=wmin, height_min=hmin, out_height=oh, out_width=ow)
            res = ref_tp.tps_H_warp(inputs, limit, cfg, inpaint_fn=None, is_plot=False)
            out[f"pipe_{name}_dims"] = np.array(dims + (seed,))
            out[f"pipe_{name}_blend"] = res["new_blend_image"].numpy()
            out[f"pipe_{name}_tps_sub"] = res["tps_output"][..., ::4, ::4].contiguous().numpy()
            out[f"pipe_{name}_tps_cs"] = np.array([float(res["tps_output"].double().sum()), float((res["tps_output"].double() ** 2).sum())])
            out[f"pipe_{name}_mask2_bits"] = np.packbits((res["mask2"].numpy() >= 0.5).astype(np.uint8).reshape(-1))
            out[f"pipe_{name}_mixmask_bits"] = np.packbits((res["mix_tps_flow_warp_mask"].numpy() >= 0.5).astype(np.uint8).reshape(-1))
            out[f"pipe_{name}_output2_sub"] = res["output2"][..., ::4, ::4].contiguous().numpy()
        # --- the mix_fn plug-ins (inpaint_fn of tps_H_warp) with a pass-through inpainter in place of the neural ones
        import importlib
        for mname, ofn in (("all_img1_with_inpaint", otp.mix_all_img1_with_inpaint), ("inpaint_all_area", otp.mix_inpaint_all_area)):
            ref_mix = importlib.import_module(f"core.inference.mix_methods.{mname}").mix_fn
            ih, iw, wmin, hmin, oh, ow = 200, 264, -21, -13, 236, 300
            case = otp.synthetic_case(5, ih, iw, wmin, hmin, oh, ow)
            inputs = types.SimpleNamespace(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in case.items()})
            limit = types.SimpleNamespace(width_min=wmin, height_min=hmin, out_height=oh, out_width=ow)
            inp = otp.PassthroughInpainter()
            fn = lambda **kw: ref_mix(**kw, inpainter=inp, use_composition=False, is_plot=False, resize_to_area_limit_before_inpaint=750 * 750)  # noqa: E731
            res = ref_tp.tps_H_warp(inputs, limit, cfg, inpaint_fn=fn, is_plot=False)
            out[f"mix_{mname}_blend"] = res["new_blend_image"].numpy()
            out[f"mix_{mname}_output2_sub"] = res["output2"][..., ::4, ::4].contiguous().numpy()
            out[f"mix_{mname}_output2_cs"] = cs(res["output2"])
            out[f"mix_{mname}_mask2_bits"] = np.packbits((res["mask2"].numpy() >= 0.5).astype(np.uint8).reshape(-1))
            out[f"mix_{mname}_area_cs"] = cs(res["inpaint_area_mask"])
    np.savez_compressed(os.path.join(OUT, "tps_pipeline.npz"), **out)
    print({k: v.shape for k, v in out.items()})
    print(os.path.getsize(os.path.join(OUT, "tps_pipeline.npz")), "bytes")


if __name__ == "__main__":
    main()
