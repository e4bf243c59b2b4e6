"""In-kernel clock of the split3 K loop (diagnostic build path ST_SPLIT3_DIAG=4: s_memtime / s_memrealtime stamps around the loop of every
workgroup's wave 0, written to the split-K workspace): shader cycles per 100 MHz tick, after >= 2 s of back-to-back launches on random data.
    ST_SPLIT3_DIAG=4 python tools/split3_clock.py"""
import os, sys, time
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
d = P.split3_desc(xp, wp, out, geom=(B, H, W, 1, 5, 1, 1, 0, 2), Cin=Cin, N=N, tile=34, split_k=1)
ws = torch.zeros(16 * 1024 * 1024, device=dev)
d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
t0 = time.time()
while time.time() - t0 < 2.5:
    for _ in range(200):
        P.launch(d)
    torch.cuda.synchronize()
# event timing of the same launches, eager and as a hipGraph of 50 launches
def ev(fn, n):
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
print(f"eager: {ev(lambda: P.launch(d), 200):.1f} us per launch")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3): P.launch(d)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        for _ in range(50): P.launch(d)
    torch.cuda.synchronize()
    print(f"hipGraph of 50 launches: {ev(gr.replay, 20) / 50:.1f} us per launch")
torch.cuda.synchronize()
st = ws[:512 * 16].view(torch.int64).view(512, 8).cpu().double()
clk = st[:, 0] / st[:, 1] * 0.1
print(f"K-loop of conv_gemm_split3_kernel64 (8192x256x1920): median in-kernel clock {clk.median():.3f} GHz (min {clk.min():.3f}, max {clk.max():.3f}); "
      f"loop {st[:, 1].median() * 10:.0f} ns = {st[:, 0].median():.0f} cycles; 60 K steps x 12 MFMAs x 32 = 23040 MFMA cycles per wave, two waves per SIMD")
t0 = st[:, 2].min()
def q(c):
    v = (st[:, c] - t0) * 0.01
    return f"min {v.min():6.2f} median {v.median():6.2f} max {v.max():6.2f} us"
print("workgroup timeline of the LAST launch, relative to the first workgroup's entry (s_memrealtime, 10 ns ticks):")
print("  entry       ", q(2)); print("  loop start  ", q(3)); print("  loop end    ", q(4)); print("  stores done ", q(5))
print("  loop length ", f"min {(st[:,4]-st[:,3]).min()*0.01:.2f} median {(st[:,4]-st[:,3]).median()*0.01:.2f} max {(st[:,4]-st[:,3]).max()*0.01:.2f} us")
