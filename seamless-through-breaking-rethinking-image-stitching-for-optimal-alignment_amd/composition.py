"""UDIS2 composition stage on the MI355X kernels.

Drop-in for ``core/UDIS2/Composition/network.py`` (``Network`` :78-127, ``build_model`` :8-22) and for the caller's
preprocessing in ``out.py:277-291`` (``compose``).  Same constructor, same ``state_dict`` keys (``down1.layer.0.weight`` ...,
loaded strictly from ``checkpoint['model']``, out.py:100), same output dict.  All convolutions are ``st_conv_gemm``
launches (implicit GEMM with dilation, fused bias + ReLU / sigmoid); both images go through the shared encoder as one batch.
"""
from __future__ import annotations

from collections import OrderedDict

import torch

from . import ops
from .checkpoint import ParamTree, flat_params
from .homography import pack_conv

# (name, cin, cout, dilation, pool) -- network.py:84-92
_DOWN = [("down1", 3, 32, 1, False), ("down2", 32, 64, 2, True), ("down3", 64, 128, 3, True), ("down4", 128, 256, 4, True),
         ("down5", 256, 512, 5, True)]
_UP = [("up1", 512, 256, 4), ("up2", 256, 128, 3), ("up3", 128, 64, 2), ("up4", 64, 32, 1)]


def composition_spec():
    d = OrderedDict()
    for name, cin, cout, _, pool in _DOWN:
        o = 1 if pool else 0                                  # MaxPool2d is layer.0 of a pooled block (network.py:29-34)
        d[f"{name}.layer.{o}.weight"], d[f"{name}.layer.{o}.bias"] = (cout, cin, 3, 3), (cout,)
        d[f"{name}.layer.{o + 2}.weight"], d[f"{name}.layer.{o + 2}.bias"] = (cout, cout, 3, 3), (cout,)
    for name, cin, cout, _ in _UP:
        d[f"{name}.halfChanelConv.0.weight"], d[f"{name}.halfChanelConv.0.bias"] = (cout, cin, 3, 3), (cout,)
        d[f"{name}.conv.0.weight"], d[f"{name}.conv.0.bias"] = (cout, cin, 3, 3), (cout,)
        d[f"{name}.conv.2.weight"], d[f"{name}.conv.2.bias"] = (cout, cout, 3, 3), (cout,)
    d["out.0.weight"], d["out.0.bias"] = (1, 32, 1, 1), (1,)
    return d


def _new(rows, cols, dev):
    return torch.empty((rows, cols), device=dev, dtype=torch.float32)


class Network(ParamTree):
    """reference: core/UDIS2/Composition/network.py:78-127 (predicts the composition mask of image 1)."""

    def __init__(self, nclasses=1):
        if nclasses != 1:
            raise NotImplementedError("the stitching pipeline uses the single-mask network (network.py:79)")
        super().__init__(composition_spec())
        self._pk = None

    def load_state_dict(self, state_dict, strict=True):
        sd = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in state_dict.items())
        r = super().load_state_dict(sd, strict=strict)
        self._pk = None
        return r

    def _apply(self, fn, *a, **k):
        self._pk = None
        return super()._apply(fn, *a, **k)

    def pack(self):
        p = flat_params(self)

        def cv(name, cpad=None):
            return pack_conv(p[name + ".weight"], cpad), p[name + ".bias"].contiguous()
        pk = {}
        for name, cin, _, _, pool in _DOWN:
            o = 1 if pool else 0
            pk[name] = (cv(f"{name}.layer.{o}", 4 if cin == 3 else None), cv(f"{name}.layer.{o + 2}"))
        for name, _, _, _ in _UP:
            pk[name] = (cv(f"{name}.halfChanelConv.0"), cv(f"{name}.conv.0"), cv(f"{name}.conv.2"))
        pk["out"] = cv("out.0")
        self._pk = pk
        return pk

    @staticmethod
    def _conv(x, wb, B, H, W, dil, act="relu", out=None):
        """3x3 conv, padding 1, dilation `dil` (the map shrinks by 2*(dil-1), network.py:32-35) + activation."""
        Ho, Wo = H + 2 - 2 * dil, W + 2 - 2 * dil
        if Ho <= 0 or Wo <= 0:
            raise RuntimeError(f"composition network: a {H}x{W} map is too small for dilation {dil} (inputs must be >= 512 px, out.py:278-283)")
        if out is None:
            out = _new(B * Ho * Wo, wb[0].shape[0], x.device)
        ops.conv_gemm(x, wb[0], out, geom=(B, H, W, 3, 3, 1, 1, 1, 1), dil=(dil, dil), bias=wb[1], act=act)
        return out, Ho, Wo

    def mask_rows(self, x, y):
        """images NCHW in [-1, 1] -> (sigmoid mask rows [B*H*W, 1], (B, H, W)); Network.forward, network.py:104-127."""
        if not x.is_cuda:
            raise RuntimeError("the composition network runs on the MI355X HIP kernels only: move the module and inputs to cuda")
        pk = self._pk or self.pack()
        B, _, H, W = x.shape
        dev = x.device
        B2 = 2 * B                                               # shared encoder: both images as one batch
        cur = _new(B2 * H * W, 4, dev)
        ops.prep_image(x.contiguous(), cur[:B * H * W], 4, 1.0, 1.0, 0.0)
        ops.prep_image(y.contiguous(), cur[B * H * W:], 4, 1.0, 1.0, 0.0)
        feats = []
        h, w = H, W
        for name, _, cout, dil, pool in _DOWN:
            if pool:
                hp, wp = h // 2, w // 2
                pooled = _new(B2 * hp * wp, cur.shape[1], dev)
                ops.maxpool(cur, pooled, B2, h, w, cur.shape[1], 2, 2, 0)
                cur, h, w = pooled, hp, wp
            cur, h, w = self._conv(cur, pk[name][0], B2, h, w, dil)
            cur, h, w = self._conv(cur, pk[name][1], B2, h, w, dil)
            feats.append((cur, h, w, cout))
        f5, h, w, c = feats[4]
        res = _new(B * h * w, c, dev)
        ops.sub_rows(f5[:B * h * w], f5[B * h * w:], res)
        for (name, cin, cout, dil), k in zip(_UP, (3, 2, 1, 0)):
            fk, hk, wk, ck = feats[k]
            R = B * hk * wk
            up = _new(R, cin, dev)
            ops.resize_nearest_rows(res, up, B, h, w, cin, hk, wk)             # F.interpolate(nearest), network.py:70
            cat = _new(R, 2 * cout, dev)                                      # torch.cat([x2, x1], dim=1): column halves
            ops.sub_rows(fk[:R], fk[R:], cat[:, :cout])
            ops.conv_gemm(up, pk[name][0][0], cat[:, cout:], geom=(B, hk, wk, 3, 3, 1, 1, 1, 1), bias=pk[name][0][1], act="relu")
            res, h, w = self._conv(cat, pk[name][1], B, hk, wk, dil)
            res, h, w = self._conv(res, pk[name][2], B, h, w, dil)
        out = _new(B * h * w, 1, dev)
        ops.conv_gemm(res, pk["out"][0], out, bias=pk["out"][1], act="sigmoid")
        return out, (B, h, w)

    def forward(self, x, y, m1=None, m2=None):
        out, (B, h, w) = self.mask_rows(x, y)
        return out.view(B, h, w, 1).permute(0, 3, 1, 2)                         # [B,1,H,W] (layout only)


def build_model(net, warp1_tensor, warp2_tensor, mask1_tensor, mask2_tensor):
    """reference: core/UDIS2/Composition/network.py:8-22."""
    out, (B, h, w) = net.mask_rows(warp1_tensor, warp2_tensor)
    if (h, w) != tuple(warp1_tensor.shape[2:]):
        raise RuntimeError("composition network output size differs from its input (dilated block without skip?)")
    lm1, lm2, st = (torch.empty_like(warp1_tensor) for _ in range(3))
    ops.compose_blend(warp1_tensor.contiguous(), warp2_tensor.contiguous(), mask1_tensor.contiguous(), mask2_tensor.contiguous(),
                      out, lm1, lm2, st)
    return dict(learned_mask1=lm1, learned_mask2=lm2, stitched_image=st)


def compose(net, output1, output2, mask1, mask2):
    """The composition stage of out.py:277-301 on test_out_forward's canvases: bilinear up-scaling to a short side of 512
    (align_corners=False), images to [-1, 1], then build_model."""
    def resize(x):
        if min(x.shape[2], x.shape[3]) < 512:
            s = 512 / min(x.shape[2], x.shape[3])
            x = ops.resize_bilinear(x.contiguous(), int(x.shape[2] * s), int(x.shape[3] * s), 2, div=(1.0 / s, 1.0 / s))
        return x
    o1, o2 = resize(output1.cuda().float()), resize(output2.cuda().float())
    w1, w2 = torch.empty_like(o1), torch.empty_like(o2)
    ops.compose_normalize(o1.contiguous(), w1)
    ops.compose_normalize(o2.contiguous(), w2)
    return build_model(net, w1, w2, resize(mask1.cuda().float()), resize(mask2.cuda().float()))


def load_com_model(composition_model_path):
    """reference: out.py:95-103."""
    net = Network()
    ckpt = torch.load(composition_model_path, map_location="cpu", weights_only=True)
    net.load_state_dict(ckpt["model"] if "model" in ckpt else ckpt)
    return net.cuda().eval(), build_model
