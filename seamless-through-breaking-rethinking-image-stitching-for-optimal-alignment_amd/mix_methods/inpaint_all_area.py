"""`inpaint_all_area` mix method (reference: core/inference/mix_methods/inpaint_all_area.py:8-73): every hole goes to the
inpainter."""
from __future__ import annotations

import torch

from .. import ops
from .utils.passthrough_inpainter import inpainter as _default_inpainter


def mix_fn(tps_H_warp, tps_H_warp_mask, output1, mask1, final_warp, occlusion_mask, padding=None, residual_flow=None,
           use_composition=False, is_plot=False, resize_to_area_limit_before_inpaint=950 * 950, inpainter=None):
    if use_composition:
        print("[Warning]: use_composition is not implemented")
    inpainter = inpainter or _default_inpainter
    f = lambda t: t.float().contiguous()                                                          # noqa: E731
    tps, tmask, o1, m1, fw, occ = map(f, (tps_H_warp, tps_H_warp_mask[:, 0:1], output1, mask1, final_warp, occlusion_mask[:, 0:1]))
    tfw, tfwm, iam0 = ops.mix_stage_a(fw, occ, m1, tps, tmask, method=1)                          # :43-51
    iam, _ = ops.dilate_thin_area_plane(iam0, thickening_kernel_size=16)                          # :52
    iam3 = iam.repeat(1, 3, 1, 1)
    if inpainter.name == "transref_inpainter":                                                    # :54-61
        inpaint_img = inpainter.inpaint(tfw, iam3, control_image_tensor=ops.mix_mul_mask(o1, clip=True),
                                        resize_to_area_limit_before_inpaint=False)
    else:
        big = iam.shape[2] * iam.shape[3] > resize_to_area_limit_before_inpaint or inpainter.name == "gan_inpainter"
        inpaint_img = inpainter.inpaint(tfw, iam3, resize_to_area_limit_before_inpaint=resize_to_area_limit_before_inpaint if big else False)
    inpaint_img = f(inpaint_img).to(tps.device)
    inpaint_img_mask = tps_H_warp_mask.clone()
    if int(torch.count_nonzero(inpaint_img)) == 0:                                                # :65-69
        print("Warning: inpaint_img is all zero, not use!!")
    else:
        tfw, tfwm = inpaint_img.clone(), inpaint_img_mask.clone()
    return tfw, tfwm, inpaint_img, inpaint_img_mask, iam3
