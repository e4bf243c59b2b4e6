"""`all_img1_with_inpaint` mix method (reference: core/inference/mix_methods/all_img1_with_inpaint.py:8-113): image 1 fills
most of the holes of the TPS-warped image 2, a thin border is left to the inpainter."""
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
    tfw, tfwm, iam0 = ops.mix_stage_a(fw, occ, m1, tps, tmask, method=0)                          # :44-53
    iam, iam_ge1 = ops.dilate_thin_area_plane(iam0)                                               # :54
    dil = ops.rect_filter(iam_ge1, 7, True)                                                       # :56-57 dilate_mask(kernel 7) > 0
    only_img1, other0 = ops.mix_stage_b(iam, dil, m1, tfw, o1)                                    # :58-77
    other, _ = ops.dilate_thin_area_plane(other0, thickening_kernel_size=8)                       # :78
    other = ops.plane_threshold(other, 0.05)                                                      # :80-81
    if inpainter.name == "transref_inpainter":                                                    # :85-99
        control = ops.mix_mul_mask(only_img1, clip=True)
        inpaint_img = inpainter.inpaint(control, other.repeat(1, 3, 1, 1), control_image_tensor=control, resize_to_area_limit_before_inpaint=False)
    else:
        masked = ops.mix_mul_mask(only_img1, other, invert=True)                                  # :82
        big = other.shape[2] * other.shape[3] > resize_to_area_limit_before_inpaint or inpainter.name == "gan_inpainter"
        inpaint_img = inpainter.inpaint(masked, other.repeat(1, 3, 1, 1),
                                        resize_to_area_limit_before_inpaint=resize_to_area_limit_before_inpaint if big else False)
    inpaint_img = ops.mix_mul_mask(f(inpaint_img).to(tps.device), tmask)                          # :101-103
    inpaint_img_mask = tps_H_warp_mask
    if int(torch.count_nonzero(inpaint_img)) == 0:                                                # :106-110
        print("Warning: inpaint_img is all zero, not use!!")
    else:
        tfw, tfwm = inpaint_img.clone(), inpaint_img_mask.clone()
    inpaint_area_mask = torch.cat((only_img1, other), dim=1)                                      # :111
    return tfw, tfwm, inpaint_img, inpaint_img_mask, inpaint_area_mask
