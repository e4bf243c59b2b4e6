"""MI355X-native (gfx950) implementation of the image-stitching hot path
``FlowHomoAdpater.forward`` of gargatik/Seamless-Through-Breaking-Rethinking-Image-Stitching-for-Optimal-Alignment.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all compute runs in
hand-written HIP kernels behind the C-ABI of ``include/stitch_gfx950.h`` (``libstitch_gfx950.so``).
Importing this package requires the built library: there is no CPU or eager-PyTorch fallback.
"""
from . import _lib  # noqa: F401  (fails loudly if the HIP extension is missing)
from . import ops  # noqa: F401
