"""5 launches each of the split3 kernel on the SepConvGRU 1x5 conv (8192 x 256 x 1920), tile 34 (64x64) then tile 32 (128x64): the program
tools/split3_pmc.sh profiles (one rocprofv3 --pmc group per pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools.split3_probe as P
ops = P.ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
B, H, W, Cin, N = 2, 64, 64, 384, 256
x = torch.randn(B * H * W, Cin, generator=g).to(dev)
w = (torch.randn(N, 5 * Cin, generator=g) / (5 * Cin) ** 0.5).to(dev)
out = torch.empty(B * H * W, N, device=dev)
xp, wp = P.pack(x), P.pack(w)
geom = (B, H, W, 1, 5, 1, 1, 0, 2)
for tile in (34, 32):
    d = P.split3_desc(xp, wp, out, geom=geom, Cin=Cin, N=N, tile=tile, split_k=1)
    for _ in range(5):
        P.launch(d)
    torch.cuda.synchronize()
