"""UDIS2 homography regression on the MI355X kernels (drop-in for ``UDIS2Network(only_homo=True)``,
reference: core/UDIS2/Homography/network.py:12-199).

Layout is channels-last end to end, both images go through the backbone as one batch, BatchNorm
(eval) is folded into the convolutions, and the contextual correlation layer is restructured:
instead of materialising 1024 3x3x1024 filters per sample (network.py:153-162) the kernel takes the
plain all-pairs product G = n1 . n2^T (one fp32-MFMA GEMM) and sums the 9 diagonal taps
G[p+d, q+d] on the fly (st_ccl_softargmax) -- 9x fewer FLOPs for the same volume.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .checkpoint import ParamTree, flat_params, homo_spec


def pack_conv(w, cin_pad=None, scale=None):
    """[Co,Ci,kh,kw] -> [Co, kh*kw*Ci_pad] with K ordered (ky, kx, c); optional per-Co scale (BN fold)."""
    Co, Ci, kh, kw = w.shape
    cp = Ci if cin_pad is None else cin_pad
    if scale is not None:
        w = w * scale.view(-1, 1, 1, 1)
    out = torch.zeros((Co, kh, kw, cp), device=w.device, dtype=torch.float32)
    out[..., :Ci] = w.permute(0, 2, 3, 1)
    return out.reshape(Co, kh * kw * cp).contiguous()


def fold_bn(p, name, eps=1e-5):
    scale = p[name + ".weight"] / torch.sqrt(p[name + ".running_var"] + eps)
    shift = p[name + ".bias"] - p[name + ".running_mean"] * scale
    return scale, shift.contiguous()


class UDIS2Network(ParamTree):
    def __init__(self, only_homo=False):
        super().__init__(homo_spec())
        self.only_homo = only_homo
        self._pk = None
        self._gen = 0                    # see FlowFormer._gen
        self.register_load_state_dict_post_hook(lambda m, k: m._invalidate())

    def _invalidate(self):
        self._pk = None
        self._gen += 1

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    # ------------------------------------------------------------------ weight prepack
    def pack(self):
        p = flat_params(self)
        dev = next(iter(p.values())).device
        pk = {}

        def conv_bn(cname, bname, cin_pad=None):
            s, sh = fold_bn(p, bname)
            return pack_conv(p[cname + ".weight"], cin_pad, s), sh

        pk["stem"] = conv_bn("feature_extractor_stage1.0", "feature_extractor_stage1.1", 4)
        for lname, nblocks in (("feature_extractor_stage1.4", 3), ("feature_extractor_stage1.5", 4),
                               ("feature_extractor_stage2.0", 6)):
            for i in range(nblocks):
                b = f"{lname}.{i}"
                blk = {k: conv_bn(f"{b}.conv{k}", f"{b}.bn{k}") for k in (1, 2, 3)}
                if f"{b}.downsample.0.weight" in p:
                    blk["ds"] = conv_bn(f"{b}.downsample.0", f"{b}.downsample.1")
                pk[b] = blk
        for idx in (0, 2, 5, 7, 10, 12):
            pk[f"reg{idx}"] = pack_conv(p[f"regressNet1_part1.{idx}.weight"], 4 if idx == 0 else None)
        # FC1 consumes the NCHW flatten c*16+p of [256,4,4]; our rows are NHWC p*256+c
        w1 = p["regressNet1_part2.0.weight"]
        pk["fc0"] = (w1.view(w1.shape[0], 256, 16).permute(0, 2, 1).reshape(w1.shape[0], 4096).contiguous(),
                     p["regressNet1_part2.0.bias"].contiguous())
        pk["fc2"] = (p["regressNet1_part2.2.weight"].contiguous(), p["regressNet1_part2.2.bias"].contiguous())
        pk["fc4"] = (p["regressNet1_part2.4.weight"].contiguous(), p["regressNet1_part2.4.bias"].contiguous())
        pk["dev"] = dev
        self._pk = pk
        return pk

    # ------------------------------------------------------------------ forward pieces
    @staticmethod
    def _bottleneck(blk, x, B, H, W, stride):
        dev = x.device
        planes = blk[1][0].shape[0]
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        o1 = torch.empty((B * H * W, planes), device=dev)
        ops.conv_gemm(x, blk[1][0], o1, bias=blk[1][1], act="relu")
        o2 = torch.empty((B * Ho * Wo, planes), device=dev)
        ops.conv_gemm(o1, blk[2][0], o2, geom=(B, H, W, 3, 3, stride, stride, 1, 1), bias=blk[2][1], act="relu")
        if "ds" in blk:
            idt = torch.empty((B * Ho * Wo, planes * 4), device=dev)
            ops.conv_gemm(x, blk["ds"][0], idt, geom=(B, H, W, 1, 1, stride, stride, 0, 0), bias=blk["ds"][1])
        else:
            idt = x
        o3 = torch.empty((B * Ho * Wo, planes * 4), device=dev)
        ops.conv_gemm(o2, blk[3][0], o3, bias=blk[3][1], aux0=idt, act="relu")
        return o3, Ho, Wo

    def features(self, x, B, H, W):
        """rows [B*H*W, 4] (prepped image) -> stage-2 feature rows [B*(H/16)*(W/16), 1024]."""
        pk = self._pk or self.pack()
        dev = x.device
        H2, W2 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        c1 = torch.empty((B * H2 * W2, 64), device=dev)
        ops.conv_gemm(x, pk["stem"][0], c1, geom=(B, H, W, 7, 7, 2, 2, 3, 3), bias=pk["stem"][1], act="relu")
        H3, W3 = (H2 + 2 - 3) // 2 + 1, (W2 + 2 - 3) // 2 + 1
        y = torch.empty((B * H3 * W3, 64), device=dev)
        ops.maxpool(c1, y, B, H2, W2, 64, 3, 2, 1)
        h, w = H3, W3
        for lname, nblocks, stride in (("feature_extractor_stage1.4", 3, 1), ("feature_extractor_stage1.5", 4, 2),
                                       ("feature_extractor_stage2.0", 6, 2)):
            for i in range(nblocks):
                y, h, w = self._bottleneck(pk[f"{lname}.{i}"], y, B, h, w, stride if i == 0 else 1)
        return y, h, w

    def ccl(self, f, B, h, w):
        """feature rows of [img1 batch | img2 batch] -> soft-argmax feature flow rows [B*h*w, 4]."""
        dev = f.device
        n = torch.empty_like(f)
        ops.l2norm_rows(f, n)
        P, C = h * w, f.shape[1]
        n = n.view(2, B, P, C)
        G = torch.empty((B, P, P), device=dev)
        ops.corr_volume(n[0], n[1], G)
        out = torch.empty((B * P, 4), device=dev)
        ops.ccl_softargmax(G, out, B, h, w)
        return out

    def regress(self, x, B, h, w):
        pk = self._pk or self.pack()
        dev = x.device
        for i, idx in enumerate((0, 2, 5, 7, 10, 12)):
            wt = pk[f"reg{idx}"]
            y = torch.empty((B * h * w, wt.shape[0]), device=dev)
            ops.conv_gemm(x, wt, y, geom=(B, h, w, 3, 3, 1, 1, 1, 1), act="relu")
            x = y
            if i % 2 == 1:
                y = torch.empty((B * (h // 2) * (w // 2), x.shape[1]), device=dev)
                ops.maxpool(x, y, B, h, w, x.shape[1], 2, 2, 0)
                x, h, w = y, h // 2, w // 2
        x = x.view(B, h * w * x.shape[1])
        if x.shape[1] != 4096:
            raise RuntimeError(f"regressNet1_part2 expects 4096 features (512x512 input), got {x.shape[1]}")
        for name, act in (("fc0", "relu"), ("fc2", "relu"), ("fc4", "none")):
            wt, b = pk[name]
            y = torch.empty((B, wt.shape[0]), device=dev)
            ops.conv_gemm(x, wt, y, bias=b, act=act)
            x = y
        return x

    def offsets_from_images(self, img1, img2, mul=1.0, div=1.0, sub=0.0):
        """NCHW images -> corner offsets [B,8]; the input scaling v = mul*(x/div) - sub is fused into
        the layout change (flowHomoAdpater.py:55-56 passes x/127.5 - 1)."""
        if not img1.is_cuda:
            raise RuntimeError("UDIS2Network runs on the MI355X HIP kernels only: move the module and inputs to cuda")
        self._pk or self.pack()
        B, _, H, W = img1.shape
        x = torch.empty((2 * B * H * W, 4), device=img1.device)
        ops.prep_image(img1.contiguous(), x[:B * H * W], 4, mul, div, sub)
        ops.prep_image(img2.contiguous(), x[B * H * W:], 4, mul, div, sub)
        f, h, w = self.features(x, 2 * B, H, W)
        return self.regress(self.ccl(f, B, h, w), B, h, w)

    def forward(self, input1_tesnor, input2_tesnor):
        """Same surface as the reference (network.py:121-137): normalised inputs -> (offset [B,8], zeros [B,338])."""
        off = self.offsets_from_images(input1_tesnor, input2_tesnor)
        if self.only_homo:
            return off, torch.zeros((off.shape[0], 338), device=off.device)
        raise NotImplementedError
