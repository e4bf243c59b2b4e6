"""A few launches of the K = 128 streaming GEMM shapes for SQ-counter passes (rocprofv3 --pmc ...)."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import stitch_amd
ops = stitch_amd.ops
for (M, N, K) in ((65536, 128, 128), (65536, 512, 128), (65536, 128, 512), (8192, 256, 1920)):
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; c = torch.empty(M, N, device="cuda")
    b = torch.randn(N, device="cuda")
    for _ in range(4):
        ops.conv_gemm(a, w, c, bias=b)
torch.cuda.synchronize()
