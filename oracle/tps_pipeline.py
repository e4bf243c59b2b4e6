"""CPU oracle of the TPS post-pipeline -- test infrastructure, torch-CPU fp32.

Restates, function by function, the reference's ``core/inference`` package for the in-tree ("kornia") TPS back-end
and with no inpainter (``inpaint_fn=None``):

  preprocess                              core/inference/tps_pipline.py:213-244
  advanced_uniform_sample_border_points   core/inference/sample_point_methods.py:5-128
  get_point_pairs / shift_points / boundary_src_and_tgt          core/inference/utils.py:61-121
  sample_init_points                      core/inference/tps_pipline.py:247-336
  get_tps_transform / warp_points_tps / warp_image_tps           core/inference/tps_methods/kornia_tps.py:26-176
                                          (the in-tree file re-exports kornia's get_tps_transform / warp_points_tps /
                                          create_meshgrid; those three are restated from kornia's published definitions:
                                          PARITY AGAINST KORNIA ITSELF IS UNPINNED -- kornia is not installable here)
  warp_by_tps ("kornia" branch)           core/inference/tps_pipline.py:339-378
  tps_H_warp (mask clean-up, mix, blend)  core/inference/tps_pipline.py:20-205
                                          (cv2.erode / cv2.dilate with an 11x11 rectangle are restated as binary
                                          min / max filters over the in-image part of the window = OpenCV's default
                                          morphology border; cv2 is absent: unpinned against cv2 itself)

Pinned to the reference's own functions (imported with kornia / cv2 / torchvision stand-ins) by
oracle/ref_harness/make_tps_goldens.py -> tests/golden/tps_pipeline.npz, tests/test_oracle_pin.py.
The reference's default back-end (OpenCV ThinPlateSplineShapeTransformer) and its inpainters are out of scope.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ synthetic inputs of the fixtures / tests
def synthetic_case(seed, ih, iw, wmin, hmin, out_h, out_w):
    """A `test_out`-shaped input set: canvases [1,3,out_h,out_w], flow at image size, binary masks."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(out_h).float(), torch.arange(out_w).float(), indexing="ij")
    tex = (torch.rand(1, 3, out_h // 8 + 2, out_w // 8 + 2, generator=g) * 255)
    tex = torch.nn.functional.interpolate(tex, size=(out_h, out_w), mode="bilinear", align_corners=False)
    tex = (tex + 20 * torch.sin(xx / 7.0) * torch.cos(yy / 5.0)).clip(0, 255)
    left, top = abs(wmin), abs(hmin)
    in_img = ((xx >= left + 6) & (xx < left + iw - 9) & (yy >= top + 4) & (yy < top + ih - 7)).float()
    H_warp_mask = in_img[None, None].repeat(1, 3, 1, 1)
    H_warp = tex * H_warp_mask
    mask1 = ((xx >= left) & (xx < left + iw) & (yy >= top) & (yy < top + ih)).float()[None, None].repeat(1, 3, 1, 1)
    mask1[:, :, :, left + iw // 2:] = 0                                   # image 1 covers the left half
    output1 = (255 - tex) * mask1
    fy, fx = torch.meshgrid(torch.arange(ih).float(), torch.arange(iw).float(), indexing="ij")
    flow = torch.stack([4 * torch.sin(fx / 40.0) + 2 * torch.cos(fy / 23.0), 3 * torch.cos(fx / 31.0) - 2 * torch.sin(fy / 17.0)])[None]
    flow = flow + 0.5 * torch.randn(1, 2, ih, iw, generator=g)
    occ = (torch.rand(1, 1, out_h, out_w, generator=g) > 0.15).float()
    final_warp = tex.roll(3, -1) * H_warp_mask
    final_warp[:, :, : top + 10] = 0
    return dict(output1=output1, mask1=mask1, H_warp=H_warp, H_warp_mask=H_warp_mask, final_warp=final_warp,
                mask2=H_warp_mask.clone(), residual_flow=flow, valid=None, occlusion_mask=occ, border_points_mask=occ)



# ------------------------------------------------------------------ preprocess (tps_pipline.py:213-244)
def preprocess(residual_flow, valid, do_avg_pooling, residual_flow_use_forward, grid_h, grid_w):
    if do_avg_pooling:
        size = residual_flow.shape[-2:]
        k = min(grid_h, grid_w) // 2 * 2 - 1
        p = (k - 1) // 2
        residual_flow = F.pad(residual_flow, (p, p, p, p), mode="constant")
        residual_flow = F.avg_pool2d(residual_flow, kernel_size=k, stride=1, padding=0)
        residual_flow = F.interpolate(residual_flow, size=size, mode="bilinear", align_corners=False)
    if not residual_flow_use_forward:
        residual_flow = -residual_flow.clone()
    if valid is not None:
        residual_flow = residual_flow * valid
    return residual_flow


# ------------------------------------------------------------------ border sampling (sample_point_methods.py:5-128)
def border_ranges(H, W, step, pad_num):
    """the (x1, y1, x2, y2) ranges between consecutive uniform border samples (:38-69); a range needs i_old != 0."""
    out = []
    for y in (pad_num, H - 1 - pad_num):                     # top, bottom
        i_old = 0
        for i in range(pad_num, W - pad_num, step):
            if i_old != 0:
                out.append((i_old, y, i, y))
            i_old = i
    for x in (pad_num, W - 1 - pad_num):                     # left, right
        i_old = 0
        for i in range(pad_num, H - pad_num, step):
            if i_old != 0:
                out.append((x, i_old, x, i))
            i_old = i
    return out


def sobel_magnitude(image):
    """|Sobel_x| and |Sobel_y| per channel, channel mean of each, summed (:70-90) -> [B,1,H,W]."""
    C = image.shape[1]
    kx = torch.tensor([[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]]).repeat(C, 1, 1, 1)
    ky = torch.tensor([[-1., -2., -1.], [0., 0., 0.], [1., 2., 1.]]).repeat(C, 1, 1, 1)
    gx = torch.abs(F.conv2d(image, kx, padding=1, groups=C)).mean(dim=1, keepdim=True)
    gy = torch.abs(F.conv2d(image, ky, padding=1, groups=C)).mean(dim=1, keepdim=True)
    return gx.abs() + gy.abs()


def advanced_uniform_sample_border_points(image, step, pad_num):
    """arg-max of the gradient magnitude inside every range window [y1-2, y2+2) x [x1-2, x2+2) (:93-113); first maximum in
    row-major order (torch.argmax); result unique-sorted (:114-117) -> [n, 2] (x, y) int64."""
    _, _, H, W = image.shape
    grad = sobel_magnitude(image)
    pts = []
    for (x1, y1, x2, y2) in border_ranges(H, W, step, pad_num):
        mask = torch.zeros_like(grad)
        mask[:, :, y1 - 2:y2 + 2, x1 - 2:x2 + 2] = 1
        rg = grad * mask + (-1 * torch.ones_like(grad)) * (1 - mask)
        idx = int(torch.argmax(rg))
        pts.append([idx % W, (idx // W) % H])
    if not pts:
        return torch.zeros((0, 2), dtype=torch.int64)
    return torch.unique(torch.tensor(pts), dim=0)


# ------------------------------------------------------------------ point pairs (core/inference/utils.py:61-121)
def get_point_pairs(border_points, flow, flow_limit):
    B = flow.shape[0]
    src = border_points.unsqueeze(0).repeat(B, 1, 1)
    fl = flow[:, :, border_points[:, 1], border_points[:, 0]].permute(0, 2, 1)
    if flow_limit == -1:
        flow_limit = (flow.shape[2] + flow.shape[3]) // 2 // 8
    if flow_limit is not None:
        a = fl.abs()
        sel = ((a[:, :, 0] < flow_limit) & (a[:, :, 1] < flow_limit)).unsqueeze(-1).expand_as(fl)
        src = src[sel].view(B, -1, 2)
        fl = fl[sel].view(B, -1, 2)
    return src, src + fl


def shift_points(points, width_min, height_min):
    out = points.clone()
    out[:, :, 0] = out[:, :, 0] + int(abs(width_min))
    out[:, :, 1] = out[:, :, 1] + int(abs(height_min))
    return out


def boundary_src_and_tgt(points_src, points_dst, out_height, out_width):
    B = points_src.shape[0]
    m = ((points_dst[:, :, 0] >= 0) & (points_dst[:, :, 0] < out_width) & (points_dst[:, :, 1] >= 0) & (points_dst[:, :, 1] < out_height)
         & (points_src[:, :, 0] >= 0) & (points_src[:, :, 0] < out_width) & (points_src[:, :, 1] >= 0) & (points_src[:, :, 1] < out_height))
    m = m.unsqueeze(-1).expand_as(points_dst)
    return points_src[m].view(B, -1, 2), points_dst[m].view(B, -1, 2)


def sample_init_points(residual_flow, H_warp, width_min, height_min, grid_h, grid_w, pad_num, get_pt_methods, flow_limit):
    """tps_pipline.py:247-336 -> (src_points, target_points, points_src, points_dst); the last two shifted onto the canvas."""
    W, H = residual_flow.shape[-1], residual_flow.shape[-2]
    left, top = int(abs(width_min)), int(abs(height_min))
    step = max(H, W) // min(grid_h, grid_w)
    crop = H_warp[:, :, top:top + H, left:left + W]                      # torchvision crop (:277,:285)
    src = tgt = None
    for method in get_pt_methods:
        bp = advanced_uniform_sample_border_points(crop, step, pad_num)
        if method == "advanced_uniform_multi":
            p = step
            while p <= max(H, W) // 4:
                bp = torch.cat((bp, advanced_uniform_sample_border_points(crop, step, p)), dim=0)
                p *= 2
        elif method != "advanced_uniform":
            raise NotImplementedError(method)
        s, t = get_point_pairs(bp, residual_flow, flow_limit)
        src = s if src is None else torch.cat((src, s), 1)
        tgt = t if tgt is None else torch.cat((tgt, t), 1)
    return src, tgt, shift_points(src, width_min, height_min), shift_points(tgt, width_min, height_min)


# ------------------------------------------------------------------ TPS (kornia_tps.py:26-176 + kornia's published functions)
def _pair_square_euclidean(t1, t2):
    t1_sq = t1.mul(t1).sum(dim=-1, keepdim=True)
    t2_sq = t2.mul(t2).sum(dim=-1, keepdim=True).transpose(1, 2)
    return (-2 * t1.matmul(t2.transpose(1, 2)) + t1_sq + t2_sq).clamp(min=0)


def _kernel_distance(sq, eps=1e-8):
    return 0.5 * sq * sq.add(eps).log()


def get_tps_transform(points_src, points_dst, solve_dtype=None):
    """kornia.geometry.transform.get_tps_transform: [K P; P^T 0] [w; a] = [dst; 0], torch.linalg.solve.

    solve_dtype=torch.float64 solves the SAME fp32 system in fp64 (a control, not the reference's arithmetic): near-coincident
    control points make the system ill-conditioned (cond ~ 1e7 on real canvases), and then the reference's fp32 LU result
    depends on the LAPACK build (tests/test_oracle_pin.py::test_tps_fp32_solve_is_lapack_dependent)."""
    B, N = points_src.shape[:2]
    K = _kernel_distance(_pair_square_euclidean(points_src, points_dst))
    zero = torch.zeros(B, 3, 3, dtype=points_src.dtype)
    one = torch.ones(B, N, 1, dtype=points_src.dtype)
    dest = torch.cat((points_dst, zero[:, :, :2]), 1)
    P = torch.cat((one, points_src), -1)
    Pt = torch.cat((P, zero), 1).transpose(1, 2)
    L = torch.cat((torch.cat((K, P), -1), Pt), 1)
    w = torch.linalg.solve(L, dest) if solve_dtype is None else torch.linalg.solve(L.to(solve_dtype), dest.to(solve_dtype)).to(L.dtype)
    return w[:, :-3], w[:, -3:]


def warp_points_tps(points, centers, kernel_weights, affine_weights):
    """kornia.geometry.transform.warp_points_tps: a_0 + [a_x a_y] . v + sum_i w_i U(|v - u_i|)."""
    k = _kernel_distance(_pair_square_euclidean(points, centers))
    return (k[..., None].mul(kernel_weights[:, None]).sum(-2) + points[..., None].mul(affine_weights[:, None, 1:]).sum(-2)
            + affine_weights[:, None, 0])


def create_meshgrid(h, w):
    """kornia.utils.create_meshgrid(normalized_coordinates=True): [1,h,w,2] (x, y) in [-1, 1]."""
    xs = (torch.linspace(0, w - 1, w) / (w - 1) - 0.5) * 2
    ys = (torch.linspace(0, h - 1, h) / (h - 1) - 0.5) * 2
    gx, gy = torch.meshgrid(xs, ys, indexing="ij")
    return torch.stack([gx, gy], -1).permute(1, 0, 2).unsqueeze(0)


def warp_image_tps(image, centers, kernel_weights, affine_weights, align_corners=False):
    B, _, h, w = image.shape
    coords = create_meshgrid(h, w).reshape(-1, 2).expand(B, -1, -1)
    warped = warp_points_tps(coords, centers, kernel_weights, affine_weights).view(-1, h, w, 2)
    return F.grid_sample(image, warped, align_corners=align_corners)


def warp_by_tps(H_warp, H_warp_mask, points_src, points_dst, out_height, out_width, kernel_scale=1.0, affine_scale=1.0, solve_dtype=None):
    """'kornia' branch of tps_pipline.py:362-378: points / (out_width, out_height) through float64 and back."""
    x = torch.cat((H_warp, H_warp_mask), dim=1)
    ps, pd = points_src.to(torch.float64), points_dst.to(torch.float64)
    ps = torch.stack([ps[:, :, 0] / out_width, ps[:, :, 1] / out_height], 2).to(torch.float32)
    pd = torch.stack([pd[:, :, 0] / out_width, pd[:, :, 1] / out_height], 2).to(torch.float32)
    kw, aw = get_tps_transform(pd, ps, solve_dtype)          # the reverse transform: dst -> src
    return warp_image_tps(x, ps, kw * kernel_scale, aw * affine_scale, align_corners=False)


# ------------------------------------------------------------------ binary morphology (cv2.erode / cv2.dilate, 11x11 rectangle)
def _rect_filter(x, k, take_max):
    p = k // 2
    if take_max:
        return F.max_pool2d(F.pad(x, (p, p, p, p), value=float("-inf")), k, stride=1)
    return -F.max_pool2d(F.pad(-x, (p, p, p, p), value=float("-inf")), k, stride=1)


def erode_dilate(x, k=11):
    return _rect_filter(_rect_filter(x, k, False), k, True)


def warp_by_tps_opencv_like(H_warp, H_warp_mask, points_src, points_dst):
    """'opencv' branch of warp_by_tps (tps_pipline.py:380-385 -> tps_methods/opencv_tps.py:8-18,59-68) as far as it can be
    restated without OpenCV (NOT importable here: the spline fit and cv2's fixed-point remap are unpinned against cv2):
      * `to_pillow_fn` (core/inference/utils.py:10) truncates image AND mask to uint8 before cv2 sees them;
      * `estimateTransformation(target, source)` + `warpImage`: the r^2 log r^2 thin-plate spline in pixel units that maps an
        output pixel (a target/dst site) to its source position, sampled bilinearly with a constant 0 border; coincident sites
        keep their first occurrence; kernel_scale / affine_scale are not used on this branch;
      * the result is uint8 (rounded half-to-even, saturated) and comes back as float.
    fp64 throughout (exact reference for the HIP kernel's fp32 evaluation)."""
    x = torch.cat((H_warp, H_warp_mask), dim=1).to(torch.uint8).double()
    a, b = points_dst[0].double().numpy(), points_src[0].double().numpy()
    _, first = np.unique(a.astype(np.float32), axis=0, return_index=True)
    first = np.sort(first)
    a, b = a[first], b[first]
    n = a.shape[0]

    def U(p, q):
        d2 = ((p[:, None, :] - q[None, :, :]) ** 2).sum(-1)
        return d2 * np.log(d2 + 1.1920929e-7)
    L = np.zeros((n + 3, n + 3))
    L[:n, :n] = U(a.astype(np.float32).astype(np.float64), a.astype(np.float32).astype(np.float64))
    L[:n, n], L[:n, n + 1:] = 1.0, a
    L[n, :n], L[n + 1:, :n] = 1.0, a.T
    w = np.linalg.solve(L, np.concatenate([b, np.zeros((3, 2))], 0))
    _, C, H, W = x.shape
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    g = np.stack([xs.ravel(), ys.ravel()], 1)
    m = U(g, a) @ w[:n] + w[n][None] + g @ w[n + 1:]
    ix, iy = m[:, 0], m[:, 1]
    x0, y0 = np.floor(ix), np.floor(iy)
    out = np.zeros((C, H * W))
    im = x[0].numpy().reshape(C, -1)
    for dx, dy, wt in ((0, 0, (x0 + 1 - ix) * (y0 + 1 - iy)), (1, 0, (ix - x0) * (y0 + 1 - iy)), (0, 1, (x0 + 1 - ix) * (iy - y0)), (1, 1, (ix - x0) * (iy - y0))):
        xx, yy = x0 + dx, y0 + dy
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        idx = (np.where(ok, yy, 0) * W + np.where(ok, xx, 0)).astype(np.int64)
        out += np.where(ok, wt, 0.0)[None] * im[:, idx]
    out = np.clip(np.rint(out), 0, 255)
    return torch.from_numpy(out.reshape(1, C, H, W)).float()


# ------------------------------------------------------------------ pipeline (tps_pipline.py:20-205, inpaint_fn=None)
def tps_H_warp(inputs, image_limit, cfg, solve_dtype=None):
    """inputs: dict(output1, mask1, H_warp, H_warp_mask, final_warp, mask2, residual_flow, valid, occlusion_mask,
    border_points_mask); image_limit: dict(width_min, height_min, out_height, out_width); cfg: TPS_PIPELINE_CONFIG."""
    out_h, out_w = image_limit["out_height"], image_limit["out_width"]
    wmin, hmin = image_limit["width_min"], image_limit["height_min"]
    flow = preprocess(inputs["residual_flow"], inputs["valid"], cfg.do_avg_pooling, cfg.residual_flow_use_forward, cfg.grid_h, cfg.grid_w)
    src, tgt, ps, pd = sample_init_points(flow, inputs["H_warp"], wmin, hmin, cfg.grid_h, cfg.grid_w, cfg.pad_num,
                                          cfg.get_pt_methods, cfg.flow_limit)
    if cfg.use_boundary_limit:
        ps, pd = boundary_src_and_tgt(ps, pd, out_h, out_w)
    if cfg.add_corner:
        corners = torch.tensor([[[0, 0], [0, out_h - 1], [out_w - 1, 0], [out_w - 1, out_h - 1]]]).repeat(ps.shape[0], 1, 1)
        ps, pd = torch.cat((ps, corners.to(ps.dtype)), 1), torch.cat((pd, corners.to(pd.dtype)), 1)
    bpm = inputs.get("border_points_mask")
    if bpm is not None:                                                   # :111-128 (loops over src_points.shape[1] entries)
        m = bpm[0, 0]
        keep = [i for i in range(src.shape[1]) if m[int(ps[0, i, 1]), int(ps[0, i, 0])] == 1]
        keep = torch.tensor(keep, dtype=torch.long)
        ps, pd = ps[:, keep, :], pd[:, keep, :]
    both = warp_by_tps(inputs["H_warp"], inputs["H_warp_mask"], ps, pd, out_h, out_w, cfg.kernel_scale, cfg.affine_scale, solve_dtype)
    tps, tmask = both[:, 0:3], both[:, 3:]
    tmask = (tmask.mean(dim=1, keepdim=True) >= 0.5).float()
    tmask = 1.0 - erode_dilate(1.0 - tmask, 11)                           # :143-150
    tps = tps * tmask
    final_warp, mask1, output1 = inputs["final_warp"], inputs["mask1"], inputs["output1"]
    fmask = ((final_warp >= 3).float().mean(dim=1, keepdim=True) >= 0.5).float()            # :154-155
    inv1 = ((1 - mask1).float().mean(dim=1, keepdim=True) >= 0.5).float()
    mix = final_warp * fmask + tps * (1 - fmask) * inv1
    mix_mask = fmask + (1 - fmask) * tmask * inv1
    output2, mask2 = mix * mix_mask, mix_mask
    blend = ((output1 * mask1 + output2 * mask2) / (mask1 + mask2)).clip(0, 255)
    blend = torch.nan_to_num(blend, nan=0.0).to(torch.uint8)              # CPU cast of NaN (0/0) is 0
    res = dict(new_blend_image=blend, tps_output=tps, mix_tps_flow_warp=output2, mix_tps_flow_warp_mask=mask2,
               points_src=ps, points_dst=pd)
    if cfg.output2_is_only_tps:
        output2, mask2 = tps * tmask, tmask
    res.update(output2=output2, mask2=mask2)
    return res


# ================================================================== mix methods (core/inference/mix_methods/*.py)
# The `mix_fn` plug-ins that `tps_H_warp` calls as `inpaint_fn` (out.py:235-236): everything except the neural inpainter
# itself is restated here; `inpainter` is any object with `.name` and `.inpaint(img, mask, control_image_tensor=None,
# prompt="", resize_to_area_limit_before_inpaint=False)` (transref_inpainter.py:16,37).
class PassthroughInpainter:
    """Stand-in for the out-of-scope neural inpainters: returns the control image (or the input) unchanged."""
    name = "passthrough_inpainter"

    def inpaint(self, init_image_tensor, mask_image_tensor, control_image_tensor=None, prompt="", resize_to_area_limit_before_inpaint=False):
        return (control_image_tensor if control_image_tensor is not None else init_image_tensor).clone()


def dilate_thin_area(mask, dilation_kernel_size=8, thickening_kernel_size=8):
    """core/inference/utils.py:125-160 (channel 0 only; even kernels with padding k//2, results cropped to H x W)."""
    _, origin_channel, H, W = mask.shape
    mask = mask[:, 0:1]
    k = torch.ones((1, 1, dilation_kernel_size, dilation_kernel_size), dtype=mask.dtype)
    p = dilation_kernel_size // 2
    erosion = (F.conv2d(mask, k, padding=(p, p)) == k.numel()).float()
    dilation = (F.conv2d(erosion, k, padding=(p, p)) >= 1).float()[:, :, :H, :W]
    thick = (mask * dilation).clamp(0, 1)
    thin = mask * (1 - thick)
    k2 = torch.ones((1, 1, thickening_kernel_size, thickening_kernel_size), dtype=mask.dtype)
    p2 = thickening_kernel_size // 2
    dthin = (F.conv2d(thin, k2, padding=(p2, p2)) >= 1).float()[:, :, :H, :W]
    return (thick + dthin).clamp(0, 1).repeat(1, origin_channel, 1, 1)


def dilate_mask(mask, kernel_size=3):
    """core/inference/utils.py:163-170: uint8 truncation of the mask (to_pillow_fn), cv2.dilate, channel 0 / 255."""
    u8 = mask[0].permute(1, 2, 0).to(torch.uint8).float()[None].permute(0, 3, 1, 2)             # values 0 / 1 for masks in [0, 1]
    d = _rect_filter(u8, kernel_size, True)[:, 0:1] / 255.0
    return d.repeat(1, 3, 1, 1).to(mask.dtype)


def mix_all_img1_with_inpaint(tps_H_warp, tps_H_warp_mask, output1, mask1, final_warp, occlusion_mask, padding=None, residual_flow=None,
                              inpainter=None, resize_to_area_limit_before_inpaint=950 * 950):
    """core/inference/mix_methods/all_img1_with_inpaint.py:8-113."""
    inpainter = inpainter or PassthroughInpainter()
    inv_mask1 = 1. - torch.where(mask1 > 0.5, torch.ones_like(mask1), torch.zeros_like(mask1))
    tfw = final_warp * occlusion_mask * mask1 + tps_H_warp * inv_mask1
    tfwm = occlusion_mask * mask1 + tps_H_warp_mask * inv_mask1
    iam = dilate_thin_area((1. - tfwm) * mask1)
    dil = dilate_mask(iam, kernel_size=7)
    dil = torch.where(dil > 0, torch.ones_like(dil), torch.zeros_like(dil))
    border = torch.abs(iam - dil)
    iam = dil
    by1 = (1 - border) * iam * mask1
    inpaint_img = tfw * (1 - by1) + (output1 * by1) * by1
    only_img1 = inpaint_img.clone()
    other = dilate_thin_area((1. - by1) * border, thickening_kernel_size=8)
    other = torch.where(other > 0.05, torch.ones_like(other), torch.zeros_like(other))
    inpaint_img = inpaint_img * (1 - other)
    if inpainter.name == "transref_inpainter":
        control = only_img1.clip(0, 255)
        inpaint_img = inpainter.inpaint(control, other, control_image_tensor=control, resize_to_area_limit_before_inpaint=False)
    else:
        big = other.shape[2] * other.shape[3] > resize_to_area_limit_before_inpaint or inpainter.name == "gan_inpainter"
        inpaint_img = inpainter.inpaint(inpaint_img, other, resize_to_area_limit_before_inpaint=resize_to_area_limit_before_inpaint if big else False)
    inpaint_img = inpaint_img.float() * tps_H_warp_mask
    inpaint_img_mask = tps_H_warp_mask
    if torch.count_nonzero(inpaint_img) != 0:
        tfw, tfwm = inpaint_img.clone(), inpaint_img_mask.clone()
    return tfw, tfwm, inpaint_img, inpaint_img_mask, torch.cat((only_img1, other[:, 0:1]), dim=1)


def mix_inpaint_all_area(tps_H_warp, tps_H_warp_mask, output1, mask1, final_warp, occlusion_mask, padding=None, residual_flow=None,
                         inpainter=None, resize_to_area_limit_before_inpaint=950 * 950):
    """core/inference/mix_methods/inpaint_all_area.py:8-73."""
    inpainter = inpainter or PassthroughInpainter()
    inv_mask1 = 1. - mask1
    tfw = final_warp * occlusion_mask + tps_H_warp * inv_mask1
    tfwm = occlusion_mask + tps_H_warp_mask * inv_mask1
    iam = dilate_thin_area((1. - tfwm) * mask1 * tps_H_warp_mask, thickening_kernel_size=16)
    if inpainter.name == "transref_inpainter":
        inpaint_img = inpainter.inpaint(tfw, iam, control_image_tensor=output1.clip(0, 255), resize_to_area_limit_before_inpaint=False)
    else:
        big = iam.shape[2] * iam.shape[3] > resize_to_area_limit_before_inpaint or inpainter.name == "gan_inpainter"
        inpaint_img = inpainter.inpaint(tfw, iam, resize_to_area_limit_before_inpaint=resize_to_area_limit_before_inpaint if big else False)
    inpaint_img_mask = tps_H_warp_mask.clone()
    if torch.count_nonzero(inpaint_img) != 0:
        tfw, tfwm = inpaint_img.clone(), inpaint_img_mask.clone()
    return tfw, tfwm, inpaint_img, inpaint_img_mask, iam


def tps_H_warp_with_inpaint(inputs, image_limit, cfg, mix_fn, inpainter=None, solve_dtype=None):
    """tps_pipline.py:20-205 with inpaint_fn = mix_fn(..., inpainter=...) (out.py:235-236): :178-188."""
    res = tps_H_warp(inputs, image_limit, cfg, solve_dtype)
    assert cfg.output2_is_only_tps
    output1, mask1 = inputs["output1"], inputs["mask1"]
    tfw, tfwm, inpaint_img, inpaint_img_mask, inpaint_area_mask = mix_fn(
        tps_H_warp=res["output2"].clone(), tps_H_warp_mask=res["mask2"].clone(), output1=output1, mask1=mask1,
        final_warp=inputs["final_warp"], occlusion_mask=inputs["occlusion_mask"], inpainter=inpainter)
    blend = ((output1 * mask1 + tfw * tfwm) / (mask1 + tfwm)).clip(0, 255)
    res.update(output2=tfw, mask2=tfwm, new_blend_image=torch.nan_to_num(blend, nan=0.0).to(torch.uint8),
               inpaint_img=inpaint_img, inpaint_area_mask=inpaint_area_mask)
    return res
