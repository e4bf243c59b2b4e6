"""The two decoder shapes of conv_gemm_dma_kernel<2,2,1,1,4> VERDICT r4 item 4 names, a few launches each, for the SQ-counter passes of
tools/dma_gemm_pmc.sh: SepConvGRU 1x5 conv (gru.py:44-59; M = 8192, N = 256 (z|r), K = 5 * 384 = 1920) and a 3x3 conv with Cin = 128 (K = 1152)."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import stitch_amd
ops = stitch_amd.ops
B, H, W = 2, 64, 64
for (Cin, N, kh, kw) in ((384, 256, 1, 5), (128, 256, 3, 3)):
    x = torch.randn(B * H * W, Cin, device="cuda")
    w = torch.randn(N, kh * kw * Cin, device="cuda") * 0.02
    b = torch.randn(N, device="cuda")
    c = torch.empty(B * H * W, N, device="cuda")
    for _ in range(5):
        ops.conv_gemm(x, w, c, geom=(B, H, W, kh, kw, 1, 1, kh // 2, kw // 2), bias=b, act="relu")
torch.cuda.synchronize()
