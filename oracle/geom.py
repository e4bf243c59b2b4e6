"""CPU oracle: geometric stage of the stitching path (test infrastructure, torch-CPU fp32).

Index / integer work delegates to the plain-C oracle (``oracle/c/geom_oracle.c``); floating-point
resampling uses the torch-CPU op the reference itself calls.  All tensors NCHW like the reference.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import cgeom


def dlt4(src, dst):
    """4-point DLT, ``h = A^-1 b`` (reference: core/udis_utils/torch_DLT.py:17-45).

    src, dst: [B,4,2] -> H [B,3,3] with H[2,2]=1.  Evaluated by the plain-C restatement (``orc_dlt4``): the
    reference's ``torch.inverse`` + ``torch.matmul`` in the exact operation order of its torch-CPU / MKL
    build (bit-identical to the reference golden ``dlt_H``), so the oracle does not depend on which MKL
    code path the host CPU selects.  ``dlt4_torch`` is the same computation through torch itself.
    """
    return torch.from_numpy(cgeom.dlt4(src.detach().numpy(), dst.detach().numpy()))


def dlt4_torch(src, dst):
    """tensor_DLT through torch's own inverse/matmul (used to pin ``orc_dlt4`` in the build container)."""
    B = src.shape[0]
    x, y = src[..., 0], src[..., 1]
    u, v = dst[..., 0], dst[..., 1]
    one, zero = torch.ones_like(x), torch.zeros_like(x)
    r0 = torch.stack([x, y, one, zero, zero, zero, -(u * x), -(u * y)], -1)
    r1 = torch.stack([zero, zero, zero, x, y, one, -(v * x), -(v * y)], -1)
    A = torch.stack([r0, r1], 2).reshape(B, 8, 8)
    b = dst.reshape(B, 8, 1)
    h8 = torch.matmul(torch.inverse(A), b).reshape(B, 8)
    return torch.cat([h8, torch.ones(B, 1, dtype=h8.dtype)], 1).reshape(B, 3, 3)


def inverse(A):
    """``torch.inverse`` of [.., 3, 3] / [.., 8, 8] fp32 in the reference's operation order (``orc_inv3/8``)."""
    return torch.from_numpy(cgeom.inverse(A.detach().numpy()))


def matmul3(A, B):
    """small ``torch.matmul`` ([..,3,3]@[..,3,3]) in the reference's operation order (``orc_matmul_small``)."""
    A, B = torch.broadcast_tensors(A, B)
    return torch.from_numpy(cgeom.matmul_small(A.contiguous().numpy(), B.contiguous().numpy()))


def homo_transformer(U, theta, out_hw, return_idx=False):
    """Homography spatial transformer (reference: core/udis_utils/torch_homo_transform.py:5-151)."""
    out, idx = cgeom.homo_warp(U.detach().numpy(), theta.detach().numpy(), out_hw, want_idx=return_idx)
    out = torch.from_numpy(out)
    return (out, torch.from_numpy(idx)) if return_idx else out


def rigid_mesh(batch, height, width, grid_h=511, grid_w=511):
    """reference: core/warp_utils.py:10-18 -> [B, gh+1, gw+1, 2] (x, y)."""
    xs = torch.from_numpy(cgeom.linspace(0.0, float(width), grid_w + 1))
    ys = torch.from_numpy(cgeom.linspace(0.0, float(height), grid_h + 1))
    ww = xs[None, :].expand(grid_h + 1, -1)
    hh = ys[:, None].expand(-1, grid_w + 1)
    return torch.stack([ww, hh], 2)[None].expand(batch, -1, -1, -1)


def h2mesh(H, mesh):
    """reference: core/warp_utils.py:20-34: mesh points through H^-1 with perspective divide."""
    B, gh, gw, _ = mesh.shape
    Hinv = inverse(H)
    pts = torch.cat([mesh.reshape(B, -1, 2), torch.ones(B, gh * gw, 1)], 2)
    tar = torch.matmul(Hinv, pts.permute(0, 2, 1))
    mx = tar[:, 0] / tar[:, 2]
    my = tar[:, 1] / tar[:, 2]
    return torch.stack([mx, my], 2).reshape(B, gh, gw, 2)


def resize_flow(flow, new_hw):
    """reference: core/warp_utils.py:38-46."""
    _, _, h, w = flow.shape
    nh, nw = new_hw
    out = F.interpolate(flow, (nh, nw), mode="bilinear", align_corners=True)
    sh, sw = h / float(nh), w / float(nw)
    out = out.clone()
    out[:, 0] = out[:, 0] / sw
    out[:, 1] = out[:, 1] / sh
    return out


def pixel_grid(B, H, W):
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    return torch.stack([xs, ys], 0).float()[None].repeat(B, 1, 1, 1)


def warp(x, flow):
    """Backward warp = grid_sample(bilinear, zeros, align_corners=True) (core/warp_utils.py:71-80)."""
    B, _, H, W = flow.shape
    g = (pixel_grid(B, H, W) + flow).permute(0, 2, 3, 1).clone()
    g[..., 0] = 2.0 * g[..., 0] / max(W - 1, 1) - 1.0
    g[..., 1] = 2.0 * g[..., 1] / max(H - 1, 1) - 1.0
    return F.grid_sample(x, g, mode="bilinear", align_corners=True)


def range_map(flow):
    """reference: core/warp_utils.py:114-175 (sum taken order-free in double)."""
    return torch.from_numpy(cgeom.range_map(flow.detach().numpy()))


def occlusion_wang(flow_ij, flow_ji):
    """compute_occlusion(..., 'wang', occlusion_are_zeros=True, boundaries_occluded=True)
    (reference: core/warp_utils.py:185-221): 1 - (1 - clamp(range_map(flow_ji), 0, 1))."""
    rm = range_map(flow_ji)
    occ = 1 - torch.clamp(rm, min=0.0, max=1.0)
    return 1 - occ


def morph_open19(mask):
    """reference: core/flowHomoAdpater.py:18-35."""
    return torch.from_numpy(cgeom.morph_open(mask.detach().numpy(), 19))


def resize512(x):
    """torchvision-0.13 tensor Resize((512,512)): bilinear, align_corners=False, no antialias
    (reference call site: core/flowHomoAdpater.py:14,204-205)."""
    return F.interpolate(x, size=(512, 512), mode="bilinear", align_corners=False, antialias=False)


def tps_transformer(U, source, target, out_hw):
    """UDIS2 TPS spatial transformer (reference: core/udis_utils/torch_tps_transform.py:7-190).

    source/target: [B,N,2] control points in [-1,1]; fp64 solve, fp32 grid + 4-tap gather
    (same `_interpolate` as the homography transformer)."""
    B, N, _ = source.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    # _solve_system :149-185
    p = torch.cat([torch.ones(B, N, 1), source], 2)
    d2 = ((p[:, :, None, :] - p[:, None, :, :]) ** 2).sum(3)
    r = d2 * torch.log(d2 + 1e-6)
    W0 = torch.cat([p, r], 2)
    W1 = torch.cat([torch.zeros(B, 3, 3), p.permute(0, 2, 1)], 2)
    Wm = torch.cat([W0, W1], 1).double()
    tp = torch.cat([target, torch.zeros(B, 3, 2)], 1).double()
    # torch.inverse (fp64) restated in plain C (cgeom.inverse_f64): MKL's threaded batched dgetrf proved non-deterministic on a
    # GPU box in round 2 (DLASWP parameter errors -> garbage pivots), and a checker must not be flaky
    Winv = torch.from_numpy(cgeom.inverse_f64(Wm.numpy()))
    T = torch.matmul(Winv, tp).permute(0, 2, 1).float()                   # [B,2,N+3]
    # _meshgrid :96-125
    xt = torch.from_numpy(cgeom.linspace(-1.0, 1.0, ow))[None, :].expand(oh, -1).reshape(1, 1, -1)
    yt = torch.from_numpy(cgeom.linspace(-1.0, 1.0, oh))[:, None].expand(-1, ow).reshape(1, 1, -1)
    px, py = source[:, :, 0:1], source[:, :, 1:2]
    dd = (xt - px) ** 2 + (yt - py) ** 2
    rr = dd * torch.log(dd + 1e-6)
    grid = torch.cat([torch.ones(B, 1, oh * ow), xt.expand(B, -1, -1), yt.expand(B, -1, -1), rr], 1)
    Tg = torch.matmul(T, grid)
    return tps_interpolate(U, Tg[:, 0], Tg[:, 1], (oh, ow)), T


def tps_indices(source, T, in_hw, out_hw, return_frac=False):
    """Sample indices (x0, x1, y0, y1) [B,oh,ow,4] int32 of the TPS transformer for control points `source` and solved
    coefficients T [B,2,N+3] (torch_tps_transform.py:96-147 grid + T@grid, :29-41 index arithmetic)."""
    B, N, _ = source.shape
    H, W = in_hw
    oh, ow = int(out_hw[0]), int(out_hw[1])
    xt = torch.from_numpy(cgeom.linspace(-1.0, 1.0, ow))[None, :].expand(oh, -1).reshape(1, 1, -1)
    yt = torch.from_numpy(cgeom.linspace(-1.0, 1.0, oh))[:, None].expand(-1, ow).reshape(1, 1, -1)
    px, py = source[:, :, 0:1], source[:, :, 1:2]
    dd = (xt - px) ** 2 + (yt - py) ** 2
    rr = dd * torch.log(dd + 1e-6)
    grid = torch.cat([torch.ones(B, 1, oh * ow), xt.expand(B, -1, -1), yt.expand(B, -1, -1), rr], 1)
    Tg = torch.matmul(T, grid)
    f32 = np.float32
    x = ((Tg[:, 0].numpy().astype(f32) + f32(1.0)) * f32(W) / f32(2.0)).astype(f32)
    y = ((Tg[:, 1].numpy().astype(f32) + f32(1.0)) * f32(H) / f32(2.0)).astype(f32)

    def toint(v):
        fl = np.floor(v)
        bad = ~((fl >= -2147483648.0) & (fl < 2147483648.0))
        return np.where(bad, -2147483648, np.where(bad, 0, fl).astype(np.int64))
    x0, y0 = toint(x), toint(y)
    out = np.stack([np.clip(x0, 0, W - 1), np.clip(x0 + 1, 0, W - 1), np.clip(y0, 0, H - 1), np.clip(y0 + 1, 0, H - 1)], -1)
    idx = torch.from_numpy(out.astype(np.int32).reshape(B, oh, ow, 4))
    if not return_frac:
        return idx
    # distance of each sample coordinate to the nearest integer (where floor() is decided by the last bit)
    fx, fy = np.abs(x - np.round(x)), np.abs(y - np.round(y))
    return idx, torch.from_numpy(np.minimum(fx, fy).reshape(B, oh, ow))


def tps_interpolate(U, xs, ys, out_hw):
    """`_interpolate` of the TPS/homography transformers (torch_tps_transform.py:18-94) in numpy fp32."""
    B, C, H, W = U.shape
    oh, ow = out_hw
    im = U.detach().numpy()
    x = ((xs.numpy().astype(np.float32) + np.float32(1.0)) * np.float32(W) / np.float32(2.0)).astype(np.float32)
    y = ((ys.numpy().astype(np.float32) + np.float32(1.0)) * np.float32(H) / np.float32(2.0)).astype(np.float32)

    def toint(v):
        f = np.floor(v)
        bad = ~((f >= -2147483648.0) & (f < 2147483648.0))
        i = np.where(bad, 0, f).astype(np.int64)
        return np.where(bad, -2147483648, i)
    x0 = toint(x); y0 = toint(y)
    x1 = np.clip(x0 + 1, 0, W - 1); y1 = np.clip(y0 + 1, 0, H - 1)
    x0 = np.clip(x0, 0, W - 1); y0 = np.clip(y0, 0, H - 1)
    f32 = np.float32
    wa = ((x1.astype(f32) - x) * (y1.astype(f32) - y)).astype(f32)
    wb = ((x1.astype(f32) - x) * (y - y0.astype(f32))).astype(f32)
    wc = ((x - x0.astype(f32)) * (y1.astype(f32) - y)).astype(f32)
    wd = ((x - x0.astype(f32)) * (y - y0.astype(f32))).astype(f32)
    out = np.empty((B, C, oh * ow), f32)
    for b in range(B):
        fl = im[b].reshape(C, -1)
        Ia = fl[:, y0[b] * W + x0[b]]; Ib = fl[:, y1[b] * W + x0[b]]
        Ic = fl[:, y0[b] * W + x1[b]]; Id = fl[:, y1[b] * W + x1[b]]
        v = (wa[b] * Ia).astype(f32)
        v = (v + (wb[b] * Ib).astype(f32)).astype(f32)
        v = (v + (wc[b] * Ic).astype(f32)).astype(f32)
        v = (v + (wd[b] * Id).astype(f32)).astype(f32)
        out[b] = v
    return torch.from_numpy(out.reshape(B, C, oh, ow))
