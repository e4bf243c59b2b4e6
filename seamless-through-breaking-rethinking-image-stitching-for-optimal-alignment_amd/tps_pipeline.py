"""TPS post-pipeline on the MI355X kernels -- drop-in for ``core.inference.tps_pipline.tps_H_warp`` of the reference
(core/inference/tps_pipline.py:20-205) with its helpers (``preprocess`` :213-244, ``sample_init_points`` :247-336,
``warp_by_tps`` :339-426; sample_point_methods.py:5-128; core/inference/utils.py:61-121).

Same call signature and result keys.  What runs where:
  * every per-pixel stage is a HIP kernel behind the C-ABI (csrc/tps_pipeline.hip): flow smoothing, Sobel magnitude and
    per-range arg-max, TPS solve + warp, mask clean-up, mix and uint8 blend.  The canvases never leave the GPU (the
    reference moves everything to the CPU first, out.py:204-216);
  * the O(100) control points are compacted on the host (unique / flow-limit / mask filters need their count anyway), as
    plain numpy on values the kernels produced.
Back-ends: ``tps_method="kornia"`` is the reference's in-tree back-end (pinned by tests/golden/tps_pipeline.npz).
``tps_method="opencv"`` (the shipped default) is served by the same kernels in pixel units: the exact r^2 log r^2
interpolating spline that OpenCV's ThinPlateSplineShapeTransformer fits, sampled bilinearly -- OpenCV itself is not
installable here, so that path is UNPINNED against OpenCV.  ``"other"`` and the neural inpainters are out of
scope.  ``inpaint_fn`` is the reference's ``mix_fn`` plug-in hook (out.py:235-236): `stitch_amd.mix_methods.<name>.mix_fn`
restates both shipped mix methods on the GPU and takes any object with the reference's inpainter protocol.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def _get(obj, name):
    return obj[name] if isinstance(obj, dict) else getattr(obj, name)


def preprocess(residual_flow, valid, do_avg_pooling, residual_flow_use_forward, grid_h, grid_w, is_plot=False):
    """tps_pipline.py:213-244 on the GPU (one kernel)."""
    k = (min(grid_h, grid_w) // 2 * 2 - 1) if do_avg_pooling else 1
    return ops.flow_boxavg(residual_flow.float().contiguous(), valid, k, negate=not residual_flow_use_forward)


def border_ranges(H, W, step, pad_num):
    """ranges between consecutive uniform border samples (sample_point_methods.py:38-69)."""
    out = []
    for y in (pad_num, H - 1 - pad_num):
        i_old = 0
        for i in range(pad_num, W - pad_num, step):
            if i_old != 0:
                out.append((i_old, y, i, y))
            i_old = i
    for x in (pad_num, W - 1 - pad_num):
        i_old = 0
        for i in range(pad_num, H - pad_num, step):
            if i_old != 0:
                out.append((x, i_old, x, i))
            i_old = i
    return out


def advanced_uniform_sample_border_points(image, step, pad_num, is_plot=False, _grad=None):
    """sample_point_methods.py:5-128 -> unique-sorted [n,2] (x, y) int64 (host tensor, like the reference's)."""
    _, _, H, W = image.shape
    if pad_num < 2:
        raise NotImplementedError("pad_num < 2 makes the reference's slice starts negative (wrap-around); not supported")
    ranges = border_ranges(H, W, step, pad_num)
    if not ranges:
        return torch.zeros((0, 2), dtype=torch.int64)
    grad = ops.sobel_magnitude(image) if _grad is None else _grad
    flat = ops.range_argmax(grad, torch.tensor(ranges, dtype=torch.int32, device=image.device)).cpu().numpy().astype(np.int64)
    pts = np.stack([flat % W, flat // W], 1)
    return torch.from_numpy(np.unique(pts, axis=0))


def get_point_pairs(border_points, flow, flow_limit):
    """core/inference/utils.py:61-93; flow stays on the GPU, the looked-up vectors come back (n x 2 floats)."""
    B, _, H, W = flow.shape
    assert B == 1
    fl = ops.gather_points(flow[0], border_points.to(torch.int32).to(flow.device)).cpu()          # [n, 2]
    src = border_points.unsqueeze(0)
    if flow_limit == -1:
        flow_limit = (H + W) // 2 // 8
    fl = fl.unsqueeze(0)
    if flow_limit is not None:
        a = fl.abs()
        sel = (a[:, :, 0] < flow_limit) & (a[:, :, 1] < flow_limit)
        src, fl = src[sel].view(1, -1, 2), fl[sel].view(1, -1, 2)
    return src, src + fl


def shift_points(points, width_min, width_max, height_min, height_max, H, W, pad_num):
    out = points.clone()
    out[:, :, 0] = out[:, :, 0] + int(abs(width_min))
    out[:, :, 1] = out[:, :, 1] + int(abs(height_min))
    return out


def boundary_src_and_tgt(points_src, points_dst, target_points, out_height, out_width):
    B = points_src.shape[0]
    m = ((points_dst[:, :, 0] >= 0) & (points_dst[:, :, 0] < out_width) & (points_dst[:, :, 1] >= 0) & (points_dst[:, :, 1] < out_height)
         & (points_src[:, :, 0] >= 0) & (points_src[:, :, 0] < out_width) & (points_src[:, :, 1] >= 0) & (points_src[:, :, 1] < out_height))
    m = m.unsqueeze(-1).expand_as(points_dst)
    return points_src[m].view(B, -1, 2), points_dst[m].view(B, -1, 2)


def sample_init_points(residual_flow, out_height, out_width, width_min, height_min, grid_h, grid_w, pad_num, get_pt_methods,
                       flow_limit, H_warp, occlusion_mask=None, valid=None, is_plot=False):
    """tps_pipline.py:247-336."""
    W, H = residual_flow.shape[-1], residual_flow.shape[-2]
    left, top = int(abs(width_min)), int(abs(height_min))
    step = max(H, W) // min(grid_h, grid_w)
    crop = H_warp[:, :, top:top + H, left:left + W].contiguous()
    grad = ops.sobel_magnitude(crop)
    # ONE device -> host round trip for the whole sampling stage (was one per border-sampling call + one for the flow lookup, 4-5 per pair:
    # VERDICT r5 item 8): every range_argmax of every method is enqueued first, the flow vectors of ALL candidate sites are gathered on the
    # stream behind them, and both come back together; the per-call de-duplication (np.unique, as sample_point_methods.py does it) and the
    # flow-limit selection then run on the host on those few hundred rows -- same points, same order as the call-by-call form.
    dev = crop.device
    calls = []                                   # (method index, device tensor of flat argmax positions)
    for mi, method in enumerate(get_pt_methods):
        if method not in ("advanced_uniform", "advanced_uniform_multi"):
            raise NotImplementedError(method)
        pads = [pad_num]
        if method == "advanced_uniform_multi":
            p = step
            while p <= max(H, W) // 4:
                pads.append(p)
                p *= 2
        for pad in pads:
            if pad < 2:
                raise NotImplementedError("pad_num < 2 makes the reference's slice starts negative (wrap-around); not supported")
            ranges = border_ranges(H, W, step, pad)
            if ranges:
                calls.append((mi, ops.range_argmax(grad, torch.tensor(ranges, dtype=torch.int32, device=dev))))
    src = tgt = None
    if calls:
        flat_dev = torch.cat([c[1].reshape(-1) for c in calls]).to(torch.int64)
        pts_dev = torch.stack([flat_dev % W, flat_dev // W], 1).to(torch.int32).contiguous()
        fl_dev = ops.gather_points(residual_flow[0], pts_dev)                                    # [n_candidates, 2]
        both = torch.cat([flat_dev.to(torch.float64).unsqueeze(1), fl_dev.to(torch.float64)], 1).cpu().numpy()      # the round trip
        flat_all, fl_all = both[:, 0].astype(np.int64), both[:, 1:].astype(np.float32)
        off = 0
        per_method = {}
        for mi, t in calls:
            n = t.numel()
            flat = flat_all[off:off + n]
            pts = np.stack([flat % W, flat // W], 1)
            uniq, first = np.unique(pts, axis=0, return_index=True)
            per_method.setdefault(mi, []).append((torch.from_numpy(uniq), torch.from_numpy(fl_all[off:off + n][first])))
            off += n
        lim = (H + W) // 2 // 8 if flow_limit == -1 else flow_limit
        for mi in sorted(per_method):
            bp = torch.cat([u for u, _ in per_method[mi]], 0).unsqueeze(0)
            fl = torch.cat([f for _, f in per_method[mi]], 0).unsqueeze(0)
            if lim is not None:
                a = fl.abs()
                sel = (a[:, :, 0] < lim) & (a[:, :, 1] < lim)
                bp, fl = bp[sel].view(1, -1, 2), fl[sel].view(1, -1, 2)
            s, t = bp, bp + fl
            src = s if src is None else torch.cat((src, s), 1)
            tgt = t if tgt is None else torch.cat((tgt, t), 1)
    if src is None:
        raise Exception("src_points is None and non_shifted_src_points is None")
    sh = lambda x: shift_points(x, width_min, None, height_min, None, H, W, pad_num)          # noqa: E731
    return src, tgt, sh(src), sh(tgt)


def warp_by_tps(H_warp, H_warp_mask, points_src, points_dst, out_height, out_width, tps_method, kernel_scale, affine_scale,
                is_plot=False):
    """tps_pipline.py:339-426 -> warped [1, 3 + C_mask, h, w] on the GPU."""
    x = torch.cat((H_warp, H_warp_mask), dim=1).float().contiguous()
    if points_src.shape[1] < 3:
        # fewer control points than the affine part of the spline has unknowns (every border sample was rejected by the flow
        # limit / occlusion filter): the reference's solvers are singular here; leave the homography warp as it is
        print(f"[tps_pipeline] only {points_src.shape[1]} control point(s) left: TPS warp skipped (identity)")
        return x
    try:
        if tps_method == "kornia":
            ps, pd = points_src.to(torch.float64), points_dst.to(torch.float64)
            ps = torch.stack([ps[:, :, 0] / out_width, ps[:, :, 1] / out_height], 2).to(torch.float32)
            pd = torch.stack([pd[:, :, 0] / out_width, pd[:, :, 1] / out_height], 2).to(torch.float32)
            return ops.tps2_warp(x, pd[0], ps[0], kernel_scale, affine_scale, mode=0)      # get_tps_transform(dst, src), centres = src
        if tps_method == "opencv":
            # tensor2WarpImage_TPS (opencv_tps.py): `to_pillow_fn` truncates image AND mask to uint8 before cv2 sees them (a
            # bilinear mask edge < 1 becomes 0), estimateTransformation(target, source) + warpImage return uint8, and
            # kernel_scale / affine_scale are not used on this branch: mode 3 = the pixel-unit spline on uint8-quantised data.
            # Coincident sites (advanced_uniform_multi concatenates point sets) make the spline singular: keep the first of each
            a, b = points_dst[0].float(), points_src[0].float()
            _, first = np.unique(a.cpu().numpy(), axis=0, return_index=True)
            if len(first) < a.shape[0]:
                keep = torch.from_numpy(np.sort(first))
                a, b = a[keep], b[keep]
            return ops.tps2_warp(x, a, b, 1.0, 1.0, mode=3)
    except ops.SingularTPSError as e:
        # the reference's solvers raise (kornia: torch.linalg.solve) or return garbage (cv2) on a singular system; a NaN canvas
        # would silently blank the blend, so leave the homography warp as it is and say so
        print(f"[tps_pipeline] {e}: TPS warp skipped (identity)")
        return x
    raise NotImplementedError(f"tps_method={tps_method!r}: only 'kornia' (pinned) and 'opencv' (native pixel-unit spline) are built")


def tps_H_warp(inputs, image_limit, tps_pipeline_config, inpaint_fn=None, is_plot=False):
    """tps_pipline.py:20-205.  ``inputs`` / ``image_limit`` / config: objects or dicts with the reference's field names."""
    cfg = tps_pipeline_config
    dev = _get(inputs, "H_warp").device
    if dev.type != "cuda":
        raise RuntimeError("tps_H_warp (gfx950) needs CUDA/HIP tensors: there is no CPU fallback")
    g = lambda n: _get(inputs, n)                                                                   # noqa: E731
    output1, mask1, H_warp, H_warp_mask, final_warp = (g(n).float().contiguous() for n in
                                                       ("output1", "mask1", "H_warp", "H_warp_mask", "final_warp"))
    valid, border_points_mask = g("valid"), g("border_points_mask")
    wmin, hmin = _get(image_limit, "width_min"), _get(image_limit, "height_min")
    out_h, out_w = _get(image_limit, "out_height"), _get(image_limit, "out_width")
    flow = preprocess(g("residual_flow"), valid, _get(cfg, "do_avg_pooling"), _get(cfg, "residual_flow_use_forward"),
                      _get(cfg, "grid_h"), _get(cfg, "grid_w"))
    src, tgt, ps, pd = sample_init_points(flow, out_h, out_w, wmin, hmin, _get(cfg, "grid_h"), _get(cfg, "grid_w"),
                                          _get(cfg, "pad_num"), _get(cfg, "get_pt_methods"), _get(cfg, "flow_limit"), H_warp)
    if _get(cfg, "use_boundary_limit"):
        ps, pd = boundary_src_and_tgt(ps, pd, tgt, out_height=out_h, out_width=out_w)
    if _get(cfg, "add_corner"):
        corners = torch.tensor([[[0, 0], [0, out_h - 1], [out_w - 1, 0], [out_w - 1, out_h - 1]]])
        ps, pd = torch.cat((ps, corners.to(ps.dtype)), 1), torch.cat((pd, corners.to(pd.dtype)), 1)
    if border_points_mask is not None:                                     # :111-128, keep points whose mask value is 1
        assert border_points_mask.shape[0] == 1 and border_points_mask.shape[1] == 1
        assert tuple(border_points_mask.shape[-2:]) == (out_h, out_w)
        n0 = src.shape[1]
        vals = ops.gather_points(border_points_mask[0].float().contiguous().to(dev), ps[0, :n0].to(torch.int32).to(dev)).cpu()[:, 0]
        keep = torch.nonzero(vals == 1)[:, 0]
        ps, pd = ps[:, keep, :], pd[:, keep, :]
    both = warp_by_tps(H_warp, H_warp_mask, ps, pd, out_h, out_w, _get(cfg, "tps_method"), _get(cfg, "kernel_scale"),
                       _get(cfg, "affine_scale"))
    tps = both[:, 0:3].contiguous()
    inv = ops.tps_mask_inv(both[:, 3:].contiguous())                       # :139-141
    inv = ops.rect_filter(ops.rect_filter(inv, 11, False), 11, True)       # :143-148 erode then dilate, 11 x 11
    tmask, mix, mixmask, blend = ops.tps_mix_blend(tps, inv, final_warp, output1, mask1)           # :150-172 (tps *= tmask in place)
    out = {"new_blend_image": blend, "tps_output": tps, "mix_tps_flow_warp": mix, "mix_tps_flow_warp_mask": mixmask,
           "points_src": ps, "points_dst": pd}
    if _get(cfg, "output2_is_only_tps"):
        out.update(output2=tps, mask2=tmask)                                # :174-176 (tmask is binary: tps * tmask == tps)
    else:
        out.update(output2=mix, mask2=mixmask)
    if inpaint_fn is not None:                                              # :178-188: the mix_fn plug-in (+ its inpainter)
        assert _get(cfg, "output2_is_only_tps") is True
        W, H = flow.shape[-1], flow.shape[-2]
        width_max, height_max = out_w - abs(wmin), out_h - abs(hmin)
        padding = (int(abs(wmin)), int(abs(width_max - W)), int(abs(hmin)), int(abs(height_max - H)))
        tfw, tfwm, inpaint_img, inpaint_img_mask, inpaint_area_mask = inpaint_fn(
            tps_H_warp=out["output2"].clone(), tps_H_warp_mask=out["mask2"].clone(), output1=output1, mask1=mask1,
            final_warp=final_warp, occlusion_mask=g("occlusion_mask"), padding=padding, residual_flow=flow)
        tfw, tfwm = tfw.float().contiguous(), tfwm.float().contiguous()
        out.update(output2=tfw, mask2=tfwm, new_blend_image=ops.blend_pair(output1, mask1, tfw, tfwm),
                   inpaint_img=inpaint_img, inpaint_area_mask=inpaint_area_mask)
    return out
