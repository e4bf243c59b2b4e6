"""PatchEmbed's three conv launches stand-alone (M = 8192 cost maps of 64x64: one pair): ms per call of ops.patch_embed's conv part.
    ST_PERSIST_CONV=0|1 python tools/patch_embed_bench.py"""
import os, sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import stitch_amd
ops = stitch_amd.ops
M, H, W = 8192, 64, 64
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(0)
cost = torch.randn(M, H * W, generator=g).to(dev)
s1 = torch.empty(M * 32 * 32, 16, device=dev); s2 = torch.empty(M * 16 * 16, 32, device=dev); s3 = torch.empty(M * 64, 64, device=dev)
w2 = (torch.randn(32, 576, generator=g) * 0.04).to(dev); b2 = torch.randn(32, generator=g).to(dev)
w4 = (torch.randn(64, 1152, generator=g) * 0.03).to(dev); b4 = torch.randn(64, generator=g).to(dev)
torch.randn(1)
s1.normal_()
def c2():
    ops.conv_gemm(s1.view(-1, 32), w2, s2, geom=(M, 32, 16, 6, 3, 2, 1, 2, 1), bias=b2, act="relu")
def c4():
    ops.conv_gemm(s2, w4, s3, geom=(M, 16, 16, 6, 6, 2, 2, 2, 2), bias=b4)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
tag = f"ST_PERSIST_CONV={os.environ.get('ST_PERSIST_CONV', '0')} ST_PERSIST_SLOTS={os.environ.get('ST_PERSIST_SLOTS', '512')}"
t2, t4 = timeit(c2), timeit(c4)
print(f"{tag}: c2 (2097152x32x576) {t2:.1f} us = {2 * 2097152 * 32 * 576 / t2 / 1e6 / 157.3:.3f} of peak; c4 (524288x64x1152) {t4:.1f} us = {2 * 524288 * 64 * 1152 / t4 / 1e6 / 157.3:.3f} of peak")
# ---- the fused c0 + c2 launch (st_patch_conv12) against patch_conv1 + c2
w0 = (torch.randn(36, 16, generator=g) * 0.1).to(dev); b0 = torch.randn(16, generator=g).to(dev)
def c0():
    ops.patch_conv1(cost, w0, b0, s1, M, 64, 64, 32, 32)
def fused():
    ops.patch_conv12(cost, w0, b0, w2, b2, s2, M)
t0, tf = timeit(c0), timeit(fused)
c0(); c2(); ref = s2.clone(); s2.zero_(); fused()
print(f"c0 {t0:.1f} us + c2 {t2:.1f} us = {t0 + t2:.1f} us unfused; fused st_patch_conv12 {tf:.1f} us ({(2 * 2097152 * 32 * 576 + 2 * 8388608 * 16 * 36) / tf / 1e6 / 157.3:.3f} of peak on c0 + c2 FLOPs); bit-identical: {bool(torch.equal(ref, s2))}")
