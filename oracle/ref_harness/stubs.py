"""Third-party stand-ins so the reference's own Python can be imported on CPU *in the build
container only* (test infrastructure; never runs on the GPU box, never imported by the product).

The reference depends on packages that are not installable here (torchvision 0.13, timm 0.4.12,
yacs, loguru, cv2, skimage, wandb).  On the hot path they contribute:
  * imported-but-unused modules (cv2, skimage.io, wandb, loguru)      -> empty modules
  * torchvision ``Resize((512,512))`` on tensors                     -> bilinear, align_corners=False, no antialias
  * torchvision ``resnet50``                                          -> canonical ResNet-50 v1.5 module tree
  * timm ``Mlp`` / ``DropPath`` / ``create_model('twins_svt_large')`` -> the reference's vendored Twins
    (core/FlowFormer/PerCostFormer3/twins.py:841) with the published twins_svt_large hyper-parameters
  * yacs ``CfgNode``                                                  -> attribute dict
These are restatements of published third-party behaviour, not reference code.
"""
from __future__ import annotations

import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class AttrDict(dict):
    """yacs.CfgNode stand-in: dict with attribute access, nested."""

    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class _Resize:
    def __init__(self, size):
        self.size = tuple(size)

    def __call__(self, x):
        return F.interpolate(x, size=self.size, mode="bilinear", align_corners=False, antialias=False)


# ---- canonical ResNet-50 v1.5 (torchvision naming) ------------------------------------------
class _Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x
        o = self.relu(self.bn1(self.conv1(x)))
        o = self.relu(self.bn2(self.conv2(o)))
        o = self.bn3(self.conv3(o))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(o + idt)


class _ResNet50(nn.Module):
    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(64, 3, 1)
        self.layer2 = self._make(128, 4, 2)
        self.layer3 = self._make(256, 6, 2)

    def _make(self, planes, blocks, stride):
        ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False),
                           nn.BatchNorm2d(planes * 4))
        layers = [_Bottleneck(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(_Bottleneck(self.inplanes, planes, 1, None))
        return nn.Sequential(*layers)


def _resnet50(pretrained=False, **kw):
    return _ResNet50()


# ---- timm bits --------------------------------------------------------------------------------
class _Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class _DropPath(nn.Module):
    def __init__(self, p=0.):
        super().__init__()

    def forward(self, x):
        return x


class _VitAttention(nn.Module):  # placeholder for timm.models.vision_transformer.Attention (unused)
    def __init__(self, *a, **k):
        super().__init__()


def _to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def install():
    """Insert the stand-ins into ``sys.modules`` and put the reference on ``sys.path``."""
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _mod("cv2")
    _mod("wandb")
    sk = _mod("skimage")
    sk.io = _mod("skimage.io")
    _mod("loguru", logger=types.SimpleNamespace(info=print, warning=print, error=print))
    tv = _mod("torchvision")
    tv.transforms = _mod("torchvision.transforms", Resize=_Resize)
    tvm = _mod("torchvision.models")
    tvm.resnet = _mod("torchvision.models.resnet", resnet50=_resnet50)
    tvm.resnet50 = _resnet50
    tv.models = tvm
    yc = _mod("yacs")
    yc.config = _mod("yacs.config", CfgNode=AttrDict)

    timm = _mod("timm")
    timm.data = _mod("timm.data", IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406),
                     IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))
    timm.models = _mod("timm.models")
    timm.models.layers = _mod(
        "timm.models.layers", Mlp=_Mlp, DropPath=_DropPath, to_2tuple=_to_2tuple,
        trunc_normal_=lambda t, std=.02, **k: nn.init.trunc_normal_(t, std=std),
        activations=types.SimpleNamespace())
    timm.models.registry = _mod("timm.models.registry", register_model=lambda f: f)
    timm.models.vision_transformer = _mod("timm.models.vision_transformer", Attention=_VitAttention)

    def create_model(name, pretrained=False, **kw):
        assert name == "twins_svt_large"
        from functools import partial
        tw = __import__("core.FlowFormer.PerCostFormer3.twins", fromlist=["Twins", "Block"])

        class PlainBlock(tw.Block):  # timm's Block.forward(x, size): no context argument
            def forward(self, x, size):
                x = x + self.drop_path(self.attn(self.norm1(x), size))
                return x + self.drop_path(self.mlp(self.norm2(x)))

        return tw.Twins(patch_size=4, embed_dims=[128, 256, 512, 1024], num_heads=[4, 8, 16, 32],
                        mlp_ratios=[4, 4, 4, 4], depths=[2, 2, 18, 2], wss=[7, 7, 7, 7],
                        sr_ratios=[8, 4, 2, 1], block_cls=PlainBlock,
                        norm_layer=partial(nn.LayerNorm, eps=1e-6))

    timm.create_model = create_model
    # test_out_forward calls .cuda() unconditionally (flowHomoAdpater.py:260-266)
    torch.Tensor.cuda = lambda self, *a, **k: self


def load_model_cfg():
    import ast
    src = open(REF_ROOT + "/configs/last_config.py").read()
    cfg = AttrDict(ast.literal_eval(src.split("=", 1)[1].strip()))
    cfg.percostformer3.pretrain = False
    return cfg


def build_reference(state_dict=None, overlay=None):
    """Reference ``FlowHomoAdpater`` on CPU, eval mode, optionally loaded (strict) with ``state_dict``."""
    install()
    import contextlib
    import io
    cfg = load_model_cfg()
    for k, v in (overlay or {}).items():
        cfg[k] = v
    with contextlib.redirect_stdout(io.StringIO()):
        from core.UDIS2.Homography.network import UDIS2Network
        from core.FlowFormer import build_flowformer
        from core.flowHomoAdpater import FlowHomoAdpater
        model = FlowHomoAdpater(UDIS2Network(only_homo=True), build_flowformer(cfg), cfg)
    if state_dict is not None:
        model.load_state_dict(state_dict, strict=True)
    return model.eval(), cfg
