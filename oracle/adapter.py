"""CPU oracle: ``FlowHomoAdpater.forward`` orchestration (test infrastructure).

Follows core/flowHomoAdpater.py: ``train_eval_foward`` (:83-191, live branch :165-186) and
``test_out_forward`` (:197-377, live branch :299-360) for the shipped config
(only_homo=False, use_forward=False, use_combine_h_flow=False, use_fb_consistency_mask=True,
test_not_use_combine_h_flow=True, use_whole_resolution=False).
"""
from __future__ import annotations

import torch

from . import geom
from .nets import W, flowformer, homo_offsets


def _conj(H, M):
    Minv = geom.inverse(M)
    return geom.matmul3(geom.matmul3(Minv.expand_as(H), H), M.expand_as(H))


def _scale_mat(w, h):
    return torch.tensor([[w / 2.0, 0., w / 2.0], [0., h / 2.0, h / 2.0], [0., 0., 1.]])[None]


def _corners(B, w, h):
    return torch.tensor([[0., 0.], [w, 0.], [0., h], [w, h]])[None].expand(B, -1, -1)


def forward_test_eval(sd, img1, img2, iters=12, stages=None, motion=None):
    """type='test_eval' (flowHomoAdpater.py:83-191).  sd: flat state dict (no 'module.' prefix).
    ``motion`` (test hook): corner offsets [B,4,2] to use instead of the homography net's own, to separate the path's
    sensitivity to its first stage from the later stages' arithmetic."""
    B, _, h, w = img1.shape
    hw, fw = W(sd, "homo_backbone."), W(sd, "flow_backbone.")
    if motion is None:
        motion = homo_offsets(hw, img1, img2)
    src = _corners(B, float(w), float(h))
    H = geom.dlt4(src / 8, (src + motion) / 8)                                       # :96
    M = _scale_mat(w / 8, h / 8)
    ones = torch.ones_like(img2)
    output_H = geom.homo_transformer(torch.cat([img2, ones], 1), _conj(H, M), (h, w))   # :111
    output_H_inv = geom.homo_transformer(torch.cat([img1, ones], 1), _conj(geom.inverse(H), M), (h, w))
    warp2 = output_H[:, 0:3]
    flow_ij = flowformer(fw, img1, warp2, iters)[0]                                 # :167
    final = geom.warp(output_H, flow_ij)                                            # :170
    ov = final[:, 3:6].mean(1)
    overlap = torch.where(ov < 0.9, torch.ones_like(ov), torch.zeros_like(ov))      # :171-174
    flow_ji = flowformer(fw, warp2, img1, iters)[0]                                 # :178
    occ = geom.occlusion_wang(flow_ij, flow_ji)
    occ = torch.where(occ >= 0.5, torch.ones_like(occ), torch.zeros_like(occ))      # :181
    final = final * occ
    if stages is not None:
        stages.update(motion=motion, flow_ji=flow_ji)
    return dict(output_H=output_H, output_H_inv=output_H_inv, final_warp_output=final, overlap=overlap,
                flow_predictions=[flow_ij], H=H, origin_occlusion_mask=occ.squeeze(1))


def blend_canvas(homo1, homo2, final):
    """Mask algebra + uint8 blend of test_out_forward (core/flowHomoAdpater.py:339-360): homo1 / homo2 = [image | mask] of the
    reference / warped view on the canvas, final = flow-warped view already multiplied by the occlusion mask.
    Returns (output1, output2, mask1, mask2, blend_image uint8)."""
    o1, m1 = homo1[:, 0:3], homo1[:, 3:6]
    o2, m2 = final[:, 0:3], final[:, 3:6]
    nov = 1 - m1
    o2 = homo2[:, 0:3] * (1 - m2) * nov + o2 * m2                                    # :345
    m2 = homo2[:, 3:6] * (1 - m2) * nov + m2 * m2                                    # :346
    blend = ((o1 * m1 + o2 * m2) / (m1 + m2)).clip(0, 255)                           # :355-356
    blend = torch.nan_to_num(blend, nan=0.0).to(torch.uint8)                         # CPU NaN -> 0
    m1o = m1.mean(1, keepdim=True).clip(0, 1).repeat(1, 3, 1, 1)
    m2o = m2.mean(1, keepdim=True).clip(0, 1).repeat(1, 3, 1, 1)
    return o1, o2, m1o, m2o, blend


def forward_test_out(sd, img1, img2, iters=12, motion=None):
    """type='test_out' (flowHomoAdpater.py:197-377); batch must be 1 (shared canvas).

    ``motion`` (test hook, as in forward_test_eval): corner offsets [B,4,2] of the 512 x 512 pair to use instead of the homography net's own."""
    B, _, ih, iw = img1.shape
    hw, fw = W(sd, "homo_backbone."), W(sd, "flow_backbone.")
    a512, b512 = geom.resize512(img1), geom.resize512(img2)                          # :204-205
    if motion is None:
        motion = homo_offsets(hw, a512, b512)
    src = _corners(B, 512., 512.)
    H512 = geom.dlt4(src, src + motion)                                              # :216
    out_H = geom.homo_transformer(torch.cat([b512, torch.ones_like(b512)], 1),
                                  _conj(H512, _scale_mat(512, 512)), (512, 512))     # :230
    warp2_512 = out_H[:, 0:3]
    m512 = (out_H[:, 3:6].mean(1, keepdim=True) > 0.5).float()                       # :233-234
    flow512 = flowformer(fw, a512, warp2_512, iters)[0]                              # :236
    residual = geom.resize_flow(flow512, (ih, iw))                                   # :241
    motion_n = torch.stack([motion[..., 0] * iw / 512, motion[..., 1] * ih / 512], 2)  # :244
    srcn = _corners(B, float(iw), float(ih))
    H = geom.dlt4(srcn, srcn + motion_n)                                             # :253
    mesh = geom.h2mesh(H, geom.rigid_mesh(B, ih, iw))                                # :254-255
    wmax = torch.maximum(torch.tensor(float(iw)), mesh[..., 0].max()).int()          # :259-268
    wmin = torch.minimum(torch.tensor(0.), mesh[..., 0].min()).int()
    hmax = torch.maximum(torch.tensor(float(ih)), mesh[..., 1].max()).int()
    hmin = torch.minimum(torch.tensor(0.), mesh[..., 1].min()).int()
    ow, oh = int(wmax - wmin), int(hmax - hmin)                                      # :270-271
    M = _scale_mat(float(ow), float(oh))[0]                                          # :274-276
    N = _scale_mat(float(iw), float(ih))[0]
    Ninv = geom.inverse(N)
    I_ = torch.tensor([[1., 0., float(wmin)], [0., 1., float(hmin)], [0., 0., 1.]])
    I_mat = geom.matmul3(geom.matmul3(Ninv, I_), M)[None]                            # :291
    homo1 = geom.homo_transformer(torch.cat([img1, torch.ones_like(img1)], 1), I_mat, (oh, ow))
    Hc = geom.matmul3(H, I_[None])                                                   # :306
    H_mat = geom.matmul3(geom.matmul3(Ninv[None].expand(B, -1, -1), Hc), M[None].expand(B, -1, -1))
    homo2 = geom.homo_transformer(torch.cat([img2, torch.ones_like(img2)], 1), H_mat, (oh, ow))
    fmask = torch.ones_like(residual).mean(1, keepdim=True)
    rf = geom.homo_transformer(torch.cat([residual, fmask], 1), I_mat, (oh, ow))     # :314
    final = geom.warp(homo2, rf[:, 0:2]) * rf[:, 2:3]                                # :316-317
    back512 = flowformer(fw, warp2_512, a512, iters)[0]                              # :326
    back = geom.resize_flow(back512, (ih, iw))
    occ = geom.morph_open19(geom.occlusion_wang(residual, back))                     # :332-333
    origin_occ = occ.clone()
    occ = geom.morph_open19(geom.homo_transformer(occ, I_mat, (oh, ow)))             # :335-336
    final = final * occ
    o1, o2, m1o, m2o, blend = blend_canvas(homo1, homo2, final)
    return dict(H_warp=homo2[:, 0:3], final_warp=final[:, 0:3], output1=o1, output2=o2, mask1=m1o, mask2=m2o,
                blend_image=blend, residual_flow=residual, width_min=int(wmin), height_min=int(hmin),
                out_height=oh, out_width=ow, H=Hc, warp_input2_mask=m512, warp_input2_tensor_512=warp2_512,
                I_mat=I_mat, H_warp_mask=homo2[:, 3:6], occlusion_mask=occ, origin_occlusion_mask=origin_occ)
