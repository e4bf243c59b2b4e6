"""CPU restatement of the UDIS2 composition stage -- test infrastructure only.

Follows core/UDIS2/Composition/network.py:
  * ``Network.forward`` (:104-127): a shared dilated-conv encoder applied to both warped images, feature
    differences decoded by nearest-upsampling blocks, 1x1 conv + sigmoid -> mask of image 1;
  * ``DownBlock`` (:25-44): [MaxPool2d(2,2)] + 2 x (Conv2d 3x3, padding=1, dilation=d) + ReLU.  padding stays 1
    while the dilation grows, so every conv of a dilated block shrinks the map by 2*(d-1);
  * ``UpBlock`` (:46-75): F.interpolate(nearest) to the skip's size, 3x3 conv + ReLU (halves the channels),
    concat [skip, up], 2 x dilated 3x3 conv + ReLU;
  * ``build_model`` (:8-22): learned masks and the stitched image.
and the caller's preprocessing, out.py:277-291.  Checkpoint keys = the reference module tree
(``down1.layer.0.weight`` ...), loaded by out.py:100 from ``checkpoint['model']``.
Pinned by tests/golden/composition_512x544.npz (reference run in the build container, seeded weights)."""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn.functional as F

DOWN = [("down1", 3, 32, 1, False), ("down2", 32, 64, 2, True), ("down3", 64, 128, 3, True), ("down4", 128, 256, 4, True),
        ("down5", 256, 512, 5, True)]
UP = [("up1", 512, 256, 4), ("up2", 256, 128, 3), ("up3", 128, 64, 2), ("up4", 64, 32, 1)]


def state_spec():
    """{key: shape} of core/UDIS2/Composition/network.py:Network (nn.Sequential indices as in the reference)."""
    d = OrderedDict()
    for name, cin, cout, _, pool in DOWN:
        o = 1 if pool else 0                                  # MaxPool2d occupies layer.0 when present (:29-34)
        d[f"{name}.layer.{o}.weight"], d[f"{name}.layer.{o}.bias"] = (cout, cin, 3, 3), (cout,)
        d[f"{name}.layer.{o + 2}.weight"], d[f"{name}.layer.{o + 2}.bias"] = (cout, cout, 3, 3), (cout,)
    for name, cin, cout, _ in UP:
        d[f"{name}.halfChanelConv.0.weight"], d[f"{name}.halfChanelConv.0.bias"] = (cout, cin, 3, 3), (cout,)
        d[f"{name}.conv.0.weight"], d[f"{name}.conv.0.bias"] = (cout, cin, 3, 3), (cout,)
        d[f"{name}.conv.2.weight"], d[f"{name}.conv.2.bias"] = (cout, cout, 3, 3), (cout,)
    d["out.0.weight"], d["out.0.bias"] = (1, 32, 1, 1), (1,)
    return d


def seeded_state_dict(seed=4321):
    """He-style random weights (the reference initialises with kaiming_normal_, :36-41); biases small."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shape in state_spec().items():
        if k.endswith(".weight"):
            fan_in = shape[1] * shape[2] * shape[3]
            sd[k] = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
        else:
            sd[k] = torch.randn(shape, generator=g) * 0.05
    return sd


def _down(sd, name, x, dil, pool):
    o = 1 if pool else 0
    if pool:
        x = F.max_pool2d(x, 2, 2)
    x = F.relu(F.conv2d(x, sd[f"{name}.layer.{o}.weight"], sd[f"{name}.layer.{o}.bias"], padding=1, dilation=dil))
    return F.relu(F.conv2d(x, sd[f"{name}.layer.{o + 2}.weight"], sd[f"{name}.layer.{o + 2}.bias"], padding=1, dilation=dil))


def _up(sd, name, x1, x2, dil):
    x1 = F.interpolate(x1, size=(x2.shape[2], x2.shape[3]), mode="nearest")
    x1 = F.relu(F.conv2d(x1, sd[f"{name}.halfChanelConv.0.weight"], sd[f"{name}.halfChanelConv.0.bias"], padding=1))
    x = torch.cat([x2, x1], dim=1)
    x = F.relu(F.conv2d(x, sd[f"{name}.conv.0.weight"], sd[f"{name}.conv.0.bias"], padding=1, dilation=dil))
    return F.relu(F.conv2d(x, sd[f"{name}.conv.2.weight"], sd[f"{name}.conv.2.bias"], padding=1, dilation=dil))


def network(sd, x, y):
    """Network.forward (:104-127); the masks m1, m2 are unused by the reference forward."""
    fx, fy = [], []
    for name, _, _, dil, pool in DOWN:
        x, y = _down(sd, name, x, dil, pool), _down(sd, name, y, dil, pool)
        fx.append(x)
        fy.append(y)
    res = fx[4] - fy[4]
    for (name, _, _, dil), k in zip(UP, (3, 2, 1, 0)):
        res = _up(sd, name, res, fx[k] - fy[k], dil)
    return torch.sigmoid(F.conv2d(res, sd["out.0.weight"], sd["out.0.bias"]))


def build_model(sd, warp1, warp2, mask1, mask2):
    """build_model (:8-22)."""
    out = network(sd, warp1, warp2)
    lm1 = (mask1 - mask1 * mask2) + mask1 * mask2 * out
    lm2 = (mask2 - mask1 * mask2) + mask1 * mask2 * (1 - out)
    stitched = (warp1 + 1.0) * lm1 + (warp2 + 1.0) * lm2 - 1.0
    return dict(learned_mask1=lm1, learned_mask2=lm2, stitched_image=stitched)


def preprocess(x, is_mask):
    """out.py:278-287: bilinear (align_corners=False) up-scaling to a short side of 512, images -> clip/127.5 - 1."""
    if min(x.shape[2], x.shape[3]) < 512:
        x = F.interpolate(x, scale_factor=512 / min(x.shape[2], x.shape[3]), mode="bilinear", align_corners=False)
    return x if is_mask else x.clip(0, 255) / 127.5 - 1.0


def compose(sd, output1, output2, mask1, mask2):
    """out.py:289-301: the composition stage on the canvases of test_out_forward."""
    return build_model(sd, preprocess(output1, False), preprocess(output2, False), preprocess(mask1, True), preprocess(mask2, True))


def synthetic_inputs(H=512, W=544, seed=77):
    """Deterministic canvases shaped like test_out_forward's output1 / output2 / mask1 / mask2 (0..255 images that are
    zero outside their masks, binary 3-channel masks overlapping in a band).  Used by the golden generator and the tests."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    base = torch.stack([127 + 100 * torch.sin(xx / 23 + c) * torch.cos(yy / 17 - c) for c in range(3)])[None]
    img1 = (base + 12 * torch.randn(1, 3, H, W, generator=g)).clip(0, 255).round()
    img2 = (base.roll(5, 3) * 0.9 + 10 + 12 * torch.randn(1, 3, H, W, generator=g)).clip(0, 255).round()
    m1 = ((xx < 0.7 * W) & (yy > 10)).float()[None, None].expand(1, 3, H, W).contiguous()
    m2 = ((xx > 0.25 * W) & (yy < H - 20)).float()[None, None].expand(1, 3, H, W).contiguous()
    return img1 * m1, img2 * m2, m1, m2
