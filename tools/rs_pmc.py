"""A few launches of row-streaming / LDS-DMA GEMM shapes for SQ-counter passes (rocprofv3 --pmc ...): argv[1] = tile."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import stitch_amd
ops = stitch_amd.ops
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for (M, N, K) in ((524288, 128, 128), (65536, 128, 128), (65536, 512, 128)):
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; c = torch.empty(M, N, device="cuda")
    b = torch.randn(N, device="cuda")
    for _ in range(4):
        ops.conv_gemm(a, w, c, bias=b, tile=tile)
torch.cuda.synchronize()
