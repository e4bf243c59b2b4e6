"""MI355X-native (gfx950) implementation of the image-stitching hot path
``FlowHomoAdpater.forward`` of gargatik/Seamless-Through-Breaking-Rethinking-Image-Stitching-for-Optimal-Alignment.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all compute runs in
hand-written HIP kernels behind the C-ABI of ``include/stitch_gfx950.h`` (``libstitch_gfx950.so``).
Importing this package requires the built library: there is no CPU or eager-PyTorch fallback.
"""
from . import _lib  # noqa: F401  (fails loudly if the HIP extension is missing)
from . import ops  # noqa: F401
from .adapter import FlowHomoAdpater, preprocess_occlusion_mask  # noqa: E402,F401
from .config import CfgNode, load_inference_config, load_model_config  # noqa: E402,F401
from .flowformer import FlowFormer, build_flowformer  # noqa: E402,F401
from .homography import UDIS2Network  # noqa: E402,F401
from . import composition  # noqa: E402,F401
from . import tps_pipeline  # noqa: E402,F401
from . import mix_methods  # noqa: E402,F401


def build_model(cfg=None):
    """``load_warp_model`` of out.py:63-91 without the checkpoint I/O: FlowHomoAdpater(UDIS2Network, FlowFormer)."""
    cfg = load_model_config() if cfg is None else cfg
    return FlowHomoAdpater(UDIS2Network(only_homo=True), build_flowformer(cfg), cfg)
