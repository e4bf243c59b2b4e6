"""Per-kernel parity on the GPU: every HIP entry point (called through the C-ABI) against the CPU
oracle on the same seeded inputs.  Tolerances: integer / index / mask data bit-exact; fp32
contractions 1e-4 relative (different accumulation order); resampling < 1e-3 (north_star)."""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import adapter as oadapter  # noqa: E402
from oracle import cgeom, geom, nets  # noqa: E402
from _measure import check  # noqa: E402

T = torch.from_numpy


@pytest.fixture(scope="module")
def ops():
    import stitch_amd
    assert torch.cuda.is_available()
    return stitch_amd.ops


def dev(t):
    return t.cuda().contiguous()


def g(seed=0):
    return torch.Generator().manual_seed(seed)


def nhwc(x):
    """NCHW cpu -> [B*H*W, C] cuda rows"""
    B, C, H, W = x.shape
    return dev(x.permute(0, 2, 3, 1).reshape(B * H * W, C))


def from_rows(y, B, H, W):
    return y.cpu().reshape(B, H, W, -1).permute(0, 3, 1, 2)


def pack_conv_w(w, cpad=None):
    Co, Ci, kh, kw = w.shape
    cpad = Ci if cpad is None else cpad
    wp = torch.zeros(Co, kh, kw, cpad)
    wp[..., :Ci] = w.permute(0, 2, 3, 1)
    return dev(wp.reshape(Co, kh * kw * cpad))


@pytest.mark.parametrize("M,N,K", [(4096, 128, 256), (300, 70, 36), (129, 2, 1152), (4096, 4096, 256), (5, 4096, 4096),
                                   (1000, 257, 100)])
def test_gemm_linear(ops, M, N, K):
    a, w, b = torch.randn(M, K, generator=g(1)), torch.randn(N, K, generator=g(2)) / K ** 0.5, torch.randn(N, generator=g(3))
    ref = F.linear(a.double(), w.double(), b.double())
    out = torch.empty(M, N, device="cuda")
    ops.conv_gemm(dev(a), dev(w), out, bias=dev(b))
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("tile", [1, 2, 3, 4])
def test_gemm_tiles_and_epilogues(ops, tile):
    M, N, K = 777, 130, 200
    a, w = torch.randn(M, K, generator=g(4)), torch.randn(N, K, generator=g(5)) / K ** 0.5
    bias, x0 = torch.randn(N, generator=g(6)), torch.randn(M // 7 + 1, N, generator=g(7))
    z, h = torch.rand(M, N, generator=g(8)), torch.randn(M, N, generator=g(9))
    base = F.linear(a, w) * 0.5 + bias + x0[torch.arange(M) // 7]
    want = (1 - z) * h + z * torch.tanh(base)
    out = torch.empty(M, N + 6, device="cuda")
    ops.conv_gemm(dev(a), dev(w), out[:, 3:3 + N], bias=dev(bias), alpha=0.5, aux0=dev(x0), row_div=7, act="tanh",
                  epi="gru", aux1=dev(z), aux2=dev(h), tile=tile)
    assert (out[:, 3:3 + N].cpu() - want).abs().max() < 2e-5
    gam = torch.tensor([0.37])
    ops.conv_gemm(dev(a), dev(w), out[:, 3:3 + N], act="gelu", epi="axpy", aux1=dev(h), scale_ptr=dev(gam), tile=tile)
    assert (out[:, 3:3 + N].cpu() - (h + 0.37 * F.gelu(F.linear(a, w)))).abs().max() < 2e-5
    ops.conv_gemm(dev(a), dev(w), out[:, 3:3 + N], act="sigmoid", epi="mul", aux1=dev(h), aux0=dev(x0[:5]), row_mod=5, tile=tile)
    want = torch.sigmoid(F.linear(a, w) + x0[:5][torch.arange(M) % 5]) * h
    assert (out[:, 3:3 + N].cpu() - want).abs().max() < 2e-5


@pytest.mark.parametrize("split", [0, 1, 3, 8])
def test_gemm_split_k(ops, split):
    """split-K slabs + deterministic reducer == single-pass kernel (same epilogue)."""
    M, N, K = 4096, 128, 2560
    a, w = torch.randn(M, K, generator=g(50)), torch.randn(N, K, generator=g(51)) / K ** 0.5
    bias, h, z = torch.randn(N, generator=g(52)), torch.randn(M, N, generator=g(53)), torch.rand(M, N, generator=g(54))
    want = (1 - z) * h + z * torch.tanh(F.linear(a, w, bias))
    out = torch.empty(M, N, device="cuda")
    ops.conv_gemm(dev(a), dev(w), out, bias=dev(bias), act="tanh", epi="gru", aux1=dev(z), aux2=dev(h), split_k=split)
    assert (out.cpu() - want).abs().max() < 3e-5
    out2 = torch.empty(M, N, device="cuda")
    ops.conv_gemm(dev(a), dev(w), out2, bias=dev(bias), act="tanh", epi="gru", aux1=dev(z), aux2=dev(h), split_k=split)
    assert torch.equal(out, out2)            # bit-reproducible


@pytest.mark.parametrize("cfg", [
    dict(B=2, C=8, H=17, W=23, Co=24, kh=3, kw=3, s=1, p=1),
    dict(B=1, C=512, H=12, W=16, Co=128, kh=1, kw=5, s=1, p=(0, 2)),
    dict(B=1, C=512, H=12, W=16, Co=128, kh=5, kw=1, s=1, p=(2, 0)),
    dict(B=3, C=1, H=64, W=64, Co=16, kh=6, kw=6, s=2, p=2),          # scalar-gather path (PatchEmbed conv1)
    dict(B=2, C=16, H=32, W=32, Co=32, kh=6, kw=6, s=2, p=2),
    dict(B=1, C=3, H=64, W=96, Co=64, kh=7, kw=7, s=2, p=3, cpad=4),   # ResNet conv1, channels padded to 4
    dict(B=1, C=3, H=64, W=96, Co=128, kh=4, kw=4, s=4, p=0, cpad=4),  # Twins patch embed
    dict(B=1, C=128, H=12, W=16, Co=128, kh=4, kw=4, s=4, p=0),        # sr conv
    dict(B=2, C=64, H=20, W=20, Co=256, kh=1, kw=1, s=2, p=0),         # strided 1x1 (ResNet downsample)
])
def test_conv_nhwc(ops, cfg):
    B, C, H, W, Co, kh, kw, s = (cfg[k] for k in ("B", "C", "H", "W", "Co", "kh", "kw", "s"))
    p = cfg["p"] if isinstance(cfg["p"], tuple) else (cfg["p"], cfg["p"])
    cpad = cfg.get("cpad", C)
    x = torch.randn(B, C, H, W, generator=g(10))
    w = torch.randn(Co, C, kh, kw, generator=g(11)) / (C * kh * kw) ** 0.5
    b = torch.randn(Co, generator=g(12))
    ref = F.relu(F.conv2d(x, w, b, stride=s, padding=p))
    xr = torch.zeros(B * H * W, cpad)
    xr[:, :C] = x.permute(0, 2, 3, 1).reshape(-1, C)
    Ho, Wo = ref.shape[2:]
    out = torch.empty(B * Ho * Wo, Co, device="cuda")
    ops.conv_gemm(dev(xr), pack_conv_w(w, cpad), out, geom=(B, H, W, kh, kw, s, s, p[0], p[1]), bias=dev(b), act="relu")
    assert (from_rows(out, B, Ho, Wo) - ref).abs().max() < 2e-5


def test_fused_gru_gates_epilogue(ops):
    """epi='zr': one GEMM produces z (cols < N/2) and r*h (cols >= N/2) -- gru.py:47-49."""
    M, K = 500, 96
    a, w = torch.randn(M, K, generator=g(60)), torch.randn(256, K, generator=g(61)) / K ** 0.5
    tab, h = torch.randn(M, 256, generator=g(62)), torch.randn(M, 128, generator=g(63))
    pre = torch.sigmoid(F.linear(a, w) + tab)
    z, rh = torch.empty(M, 128, device="cuda"), torch.empty(M, 130, device="cuda")
    for split in (1, 3):
        ops.conv_gemm(dev(a), dev(w), z, aux0=dev(tab), act="sigmoid", epi="zr", aux1=dev(h), out2=rh[:, 1:129], split_k=split)
        assert (z.cpu() - pre[:, :128]).abs().max() < 1e-5
        assert (rh[:, 1:129].cpu() - pre[:, 128:] * h).abs().max() < 1e-5


def test_patch_conv1_direct(ops):
    maps = torch.randn(5, 1, 12, 16, generator=g(64)) * 4
    w, b = torch.randn(16, 1, 6, 6, generator=g(65)) / 6, torch.randn(16, generator=g(66))
    ref = F.relu(F.conv2d(F.pad(maps, (0, 0, 0, 4)), w, b, stride=2, padding=2))          # H padded 12 -> 16
    out = torch.empty(5 * 8 * 8, 16, device="cuda")
    ops.patch_conv1(dev(maps.reshape(5, -1)), dev(w.reshape(16, 36).t()), dev(b), out, 5, 12, 16, 8, 8)
    assert (from_rows(out, 5, 8, 8) - ref).abs().max() < 1e-5


def test_corr_volume(ops):
    f1, f2 = torch.randn(2, 256, 16, 16, generator=g(13)), torch.randn(2, 256, 16, 16, generator=g(14))
    ref = nets.corr_volume(f1, f2)
    a = dev(f1.reshape(2, 256, -1).transpose(1, 2))
    b = dev(f2.reshape(2, 256, -1).transpose(1, 2))
    out = torch.empty(2, 256, 256, device="cuda")
    ops.corr_volume(a, b, out)
    assert (out.cpu() - ref).abs().max() < 1e-4


def test_rowwise_ops(ops):
    x = torch.randn(1000, 192, generator=g(15)) * 3 + 1
    w, b = torch.randn(192, generator=g(16)), torch.randn(192, generator=g(17))
    out = torch.empty(1000, 200, device="cuda")
    for eps in (1e-5, 1e-6):
        ops.layernorm(dev(x), dev(w), dev(b), out[:, :192], eps)
        assert (out[:, :192].cpu() - F.layer_norm(x, (192,), w, b, eps)).abs().max() < 2e-5
    s = torch.randn(300, 4096, generator=g(18)) * 4
    sd = dev(s)
    ops.softmax_rows(sd)
    assert (sd.cpu() - torch.softmax(s, -1)).abs().max() < 5e-6
    s2 = torch.randn(7, 100, generator=g(19))
    sd2 = dev(s2)
    ops.softmax_rows(sd2)
    assert (sd2.cpu() - torch.softmax(s2, -1)).abs().max() < 5e-6
    f = torch.randn(500, 1024, generator=g(20))
    o = torch.empty(500, 1024, device="cuda")
    ops.l2norm_rows(dev(f), o)
    assert (o.cpu() - F.normalize(f, p=2, dim=1)).abs().max() < 1e-6


def test_maxpool_and_peg(ops):
    x = torch.randn(2, 32, 19, 23, generator=g(21))
    for k, s, p in ((3, 2, 1), (2, 2, 0)):
        ref = F.max_pool2d(x, k, s, p)
        out = torch.empty(2 * ref.shape[2] * ref.shape[3], 32, device="cuda")
        ops.maxpool(nhwc(x), out, 2, 19, 23, 32, k, s, p)
        assert torch.equal(from_rows(out, 2, ref.shape[2], ref.shape[3]), ref)
    w, b = torch.randn(32, 1, 3, 3, generator=g(22)), torch.randn(32, generator=g(23))
    ref = F.conv2d(x, w, b, padding=1, groups=32) + x
    out = torch.empty(2 * 19 * 23, 32, device="cuda")
    ops.dwconv3x3_residual(nhwc(x), dev(w.reshape(32, 9).t()), dev(b), out, 2, 19, 23, 32)
    assert (from_rows(out, 2, 19, 23) - ref).abs().max() < 1e-5


def test_sine_pe(ops):
    c = torch.rand(77, 2, generator=g(24)) * 60
    out = torch.zeros(77, 70, device="cuda")
    ops.sine_pe(out[:, :64], 64, coords=dev(c))
    assert (out[:, :64].cpu() - nets.sine_pe(c, 64)).abs().max() < 2e-5
    grid = nets.coords_grid(1, 5, 9).view(1, 2, -1).permute(0, 2, 1)[0]
    o2 = torch.ones(45, 192, device="cuda")
    ops.sine_pe(o2, 192, Wg=9, cscale=8.0, coff=4.0, accumulate=True)
    assert (o2.cpu() - (1 + nets.sine_pe(grid * 8 + 4, 192))).abs().max() < 3e-5
    o4 = torch.full((45, 192), 5.0, device="cuda")                      # columns < 128 written, columns >= 128 accumulated
    ops.sine_pe(o4, 192, Wg=9, cscale=8.0, coff=4.0, accumulate=128)
    pe = nets.sine_pe(grid * 8 + 4, 192)
    assert (o4[:, :128].cpu() - pe[:, :128]).abs().max() < 3e-5 and (o4[:, 128:].cpu() - (5 + pe[:, 128:])).abs().max() < 3e-5
    o3 = torch.empty(14 * 14, 128, device="cuda")
    ops.sine_pe(o3, 128, Wg=14, ws=7)
    gx = nets.coords_grid(1, 14, 14).view(1, 2, -1).permute(0, 2, 1)[0] % 7
    assert (o3.cpu() - nets.sine_pe(gx, 128)).abs().max() < 2e-5


@pytest.mark.parametrize("B,heads,Nq,Nk,D,bq", [(50, 8, 8, 64, 16, True), (50, 8, 8, 8, 16, False), (300, 8, 1, 8, 8, False)])
def test_attention_small(ops, B, heads, Nq, Nk, D, bq):
    C = heads * D
    q = torch.randn(1 if bq else B, Nq, C, generator=g(25))
    k, v = torch.randn(B, Nk, C, generator=g(26)), torch.randn(B, Nk, C, generator=g(27))
    ref = nets.mha(q, k, v, heads, D ** -0.5)
    out = torch.empty(B, Nq, C, device="cuda")
    ops.attention_small(dev(q), (0 if bq else Nq * C, C), dev(k), (Nk * C, C), dev(v), (Nk * C, C), out, (Nq * C, C),
                        B, heads, Nq, Nk, D, D ** -0.5)
    assert (out.cpu() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("Nq,Nk,D,heads", [(1000, 256, 32, 4), (4096, 256, 16, 8), (192, 12, 16, 8)])
def test_attention_kvlds(ops, Nq, Nk, D, heads):
    C, B = heads * D, 2
    q, k, v = (torch.randn(B, n, C, generator=g(s)) for n, s in ((Nq, 28), (Nk, 29), (Nk, 30)))
    ref = nets.mha(q, k, v, heads, D ** -0.5)
    out = torch.empty(B, Nq, C, device="cuda")
    ops.attention_kvlds(dev(q), (Nq * C, C), dev(k), (Nk * C, C), dev(v), (Nk * C, C), out, (Nq * C, C), B, heads, Nq, Nk,
                        D, D ** -0.5)
    assert (out.cpu() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("H,W,heads,D", [(16, 24, 4, 32), (12, 16, 8, 16), (14, 7, 8, 32)])
def test_window_attention(ops, H, W, heads, D):
    """LSA core vs the oracle's padded-window attention (twins.py:587-631) with q/k/v given."""
    B, C, ws = 2, heads * D, 7
    q, k, v = (torch.randn(B, H * W, C, generator=g(s)) for s in (31, 32, 33))
    qpad, kpad, vpad = (torch.randn(49, C, generator=g(s)) for s in (34, 35, 36))

    def windows(t, pad):
        t = t.view(B, H, W, C)
        pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws
        Hp, Wp = H + pb, W + pr
        full = pad.view(1, 1, ws, 1, ws, C).expand(B, Hp // ws, ws, Wp // ws, ws, C).reshape(B, Hp, Wp, C).clone()
        full[:, :H, :W] = t
        return full.reshape(B, Hp // ws, ws, Wp // ws, ws, C).transpose(2, 3).reshape(-1, 49, C), Hp, Wp
    qw, Hp, Wp = windows(q, qpad)
    kw, _, _ = windows(k, kpad)
    vw, _, _ = windows(v, vpad)
    o = nets.mha(qw, kw, vw, heads, D ** -0.5)
    ref = o.reshape(B, Hp // ws, Wp // ws, ws, ws, C).transpose(2, 3).reshape(B, Hp, Wp, C)[:, :H, :W].reshape(B, H * W, C)
    out = torch.empty(B, H * W, C, device="cuda")
    ops.window_attention(dev(q), dev(k), dev(v), H * W * C, C, dev(qpad), dev(kpad), dev(vpad), out, H * W * C, C, B, H, W,
                         heads, D, ws, D ** -0.5)
    assert (out.cpu() - ref).abs().max() < 2e-5


def test_ccl(ops, golden_ops):
    f1, f2 = T(golden_ops["ccl_f1"]), T(golden_ops["ccl_f2"])
    B, C, h, w = f1.shape
    n1, n2 = torch.empty(B * h * w, C, device="cuda"), torch.empty(B * h * w, C, device="cuda")
    ops.l2norm_rows(nhwc(f1), n1)
    ops.l2norm_rows(nhwc(f2), n2)
    G = torch.empty(B, h * w, h * w, device="cuda")
    ops.corr_volume(n1.view(B, h * w, C), n2.view(B, h * w, C), G)
    out = torch.empty(B * h * w, 4, device="cuda")
    ops.ccl_softargmax(G, out, B, h, w)
    got = from_rows(out, B, h, w)
    assert (got[:, :2] - T(golden_ops["ccl_out"])).abs().max() < 1e-4
    assert (got[:, 2:] == 0).all()


def test_cost_lookup_and_upsample(ops, golden_ops):
    gd = golden_ops
    maps, coords = T(gd["lookup_maps"]), T(gd["lookup_coords"])
    Nq = maps.shape[0]
    c = nhwc(coords)
    out = torch.zeros(Nq, 84, device="cuda")
    ops.cost_lookup(dev(maps.reshape(Nq, -1)), c, out, Nq, 12, 16)
    assert (from_rows(out[:, :81], 1, 12, 16) - T(gd["lookup_out"])).abs().max() < 1e-4
    flow, mask = T(gd["ub_flow"]), T(gd["ub_mask"])
    c1 = nhwc(flow + nets.coords_grid(1, 12, 16))
    up = torch.empty(1, 2, 96, 128, device="cuda")
    ops.convex_upsample(c1, nhwc(mask), up, 1, 12, 16)
    assert (up.cpu() - T(gd["up_out"])).abs().max() < 1e-4
    f4 = torch.empty(12 * 16, 4, device="cuda")
    ops.flow_from_coords(c1, f4, None, 1, 12, 16)
    assert (from_rows(f4[:, :2], 1, 12, 16) - flow).abs().max() < 1e-5 and (f4[:, 2:] == 0).all()
    cg = torch.empty(2 * 12 * 16, 2, device="cuda")
    ops.coords_grid(cg, 2, 12, 16)
    assert torch.equal(from_rows(cg, 2, 12, 16), nets.coords_grid(2, 12, 16))


# ---------------------------------------------------------------- geometric stage
def test_homo_warp_indices_bit_exact(ops, golden_ops):
    U, theta = T(golden_ops["homo_U"]), T(golden_ops["homo_theta"])
    out, idx = ops.homo_warp(dev(U), dev(theta.reshape(2, 9)), (33, 47), want_idx=True)
    ref_out, ref_idx = cgeom.homo_warp(U.numpy(), theta.numpy(), (33, 47))
    assert np.array_equal(idx.cpu().numpy(), ref_idx)                      # grid indices bit-exact
    assert np.array_equal(out.cpu().numpy(), golden_ops["homo_out"])       # == reference output, bit for bit
    assert np.array_equal(out.cpu().numpy(), ref_out)


@pytest.mark.parametrize("hw,ohw", [((512, 512), (512, 512)), ((300, 400), (315, 439)), ((64, 64), (1025, 1026))])
def test_homo_warp_sizes_and_ones(ops, hw, ohw):
    gg = g(40)
    U = torch.rand(1, 3, *hw, generator=gg) * 255
    theta = (torch.eye(3) + torch.tensor([[0.08, 0.03, 0.05], [-0.02, 0.1, -0.04], [0.02, -0.03, 0.0]])).reshape(1, 9)
    out, idx = ops.homo_warp(dev(U), dev(theta), ohw, n_ones=3, want_idx=True)
    ref_out, ref_idx = cgeom.homo_warp(torch.cat([U, torch.ones_like(U)], 1).numpy(), theta.numpy(), ohw)
    assert np.array_equal(idx.cpu().numpy(), ref_idx)
    assert np.array_equal(out.cpu().numpy(), ref_out)


def test_homo_warp_horizon_and_identity(ops):
    """edge cases: t ~ 0 (divide guard, INT_MIN cast) and identity theta (not an identity warp)."""
    U = torch.rand(1, 1, 32, 32, generator=g(41))
    for th in ([1, 0, 0, 0, 1, 0, 0, 0, 1], [1, 0, 0, 0, 1, 0, 1, 0, 0], [1, 0, 0, 0, 1, 0, 5, 5, 1e-8], [0] * 9):
        theta = torch.tensor([th], dtype=torch.float32)
        out, idx = ops.homo_warp(dev(U), dev(theta), (32, 32), want_idx=True)
        ref_out, ref_idx = cgeom.homo_warp(U.numpy(), theta.numpy(), (32, 32))
        assert np.array_equal(idx.cpu().numpy(), ref_idx), th
        a, b = out.cpu().numpy(), ref_out
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b)), th


def test_dlt_and_mat3(ops, golden_ops):
    """DLT solve and the 3x3 chain follow the reference's fp32 arithmetic operation for operation
    (torch_DLT.py:17-45, flowHomoAdpater.py:105-112): bit-identical to the reference golden and to the C oracle."""
    src, dst = T(golden_ops["dlt_src"]), T(golden_ops["dlt_dst"])
    H = torch.empty(5, 3, 3, device="cuda")
    ops.dlt4(dev(src[0]), dev(dst - src), H, 5, 1.0, 1.0, 1.0)
    gH = T(golden_ops["dlt_H"])
    # two bars: within rounding of the golden on ANY host (the golden's bits come from the MKL / AVX dispatch of the machine that
    # generated it: oracle/ref_harness/make_goldens.py) ...
    assert (H.cpu() - gH).abs().max() <= 2e-6 * max(1.0, gH.abs().max().item())
    # ... and bit for bit against THIS committed golden (generated in the build container, AVX-512 MKL code path), which is what
    # the C oracle and the kernel restate operation for operation
    assert torch.equal(H.cpu(), gH)
    ops.dlt4(dev(src[0]), dev(dst - src), H, 5, 1.0, 1.0, 8.0)
    assert torch.equal(H.cpu(), geom.dlt4(src / 8, dst / 8))
    gen = torch.Generator().manual_seed(11)
    for (w, h) in ((512., 512.), (400., 304.), (1000., 760.)):
        s = torch.tensor([[0., 0.], [w, 0.], [0., h], [w, h]])
        mo = (torch.rand(64, 4, 2, generator=gen) - 0.5) * 60
        Hb = torch.empty(64, 3, 3, device="cuda")
        ops.dlt4(dev(s), dev(mo), Hb, 64, w / 512.0, h / 512.0, 1.0)
        mn = torch.stack([mo[..., 0] * w / 512, mo[..., 1] * h / 512], 2)          # flowHomoAdpater.py:244
        assert torch.equal(Hb.cpu(), geom.dlt4(s[None].expand(64, -1, -1), s[None] + mn))
    M = torch.tensor([[32., 0, 32], [0, 24, 24], [0, 0, 1]])
    Minv = geom.inverse(M)
    Hc = T(golden_ops["dlt_H"])
    out = torch.empty(5, 3, 3, device="cuda")
    ops.mat3_sandwich(dev(Minv), dev(Hc), dev(M), out)
    assert torch.equal(out.cpu(), geom.matmul3(geom.matmul3(Minv.expand_as(Hc), Hc), M.expand_as(Hc)))
    ops.mat3_sandwich(dev(Minv), dev(Hc), dev(M), out, invert=True)
    assert torch.equal(out.cpu(), geom.matmul3(geom.matmul3(Minv.expand_as(Hc), geom.inverse(Hc)), M.expand_as(Hc)))
    R = torch.randn(200, 3, 3, generator=gen)                                         # general matrices: pivoting paths
    eye = torch.eye(3)
    outR = torch.empty(200, 3, 3, device="cuda")
    ops.mat3_sandwich(dev(eye), dev(R), dev(eye), outR, invert=True)
    assert torch.equal(outR.cpu(), geom.inverse(R))


def test_mesh_bounds(ops, golden_ops):
    out = torch.empty(4, device="cuda")
    ops.mesh_bounds(dev(T(golden_ops["mesh_H"])), out, 400, 300)
    assert (out.cpu() - T(golden_ops["mesh_minmax"])).abs().max() < 1e-3
    # the multi-workgroup reduction (integer atomics on float bit patterns): extremes of either sign, batches, small meshes, and a
    # second call into the same buffer (the init kernel, not the previous result, seeds the atomics); min / max are exact, so the
    # int()-truncated canvas bounds the adapter derives (flowHomoAdpater.py:259-268) must equal the oracle's
    gen = g(77)
    for trial in range(6):
        B = 1 + trial % 3
        Hm = torch.eye(3)[None].repeat(B, 1, 1) + torch.randn(B, 3, 3, generator=gen) * torch.tensor([[0.05, 0.05, 60.0], [0.05, 0.05, 60.0], [1e-4, 1e-4, 0.0]])
        w, h, gw, gh = (512, 512, 511, 511) if trial < 3 else (37 + 10 * trial, 29, 12, 7)
        mesh = geom.h2mesh(Hm, geom.rigid_mesh(B, h, w, gh, gw))
        want = torch.stack([mesh[..., 0].min(), mesh[..., 0].max(), mesh[..., 1].min(), mesh[..., 1].max()])
        for rep in range(2):
            ops.mesh_bounds(dev(Hm), out, w, h, gw, gh)
            got = out.cpu()
            assert (got - want).abs().max() <= 1e-3 * max(1.0, want.abs().max().item()), (trial, got, want)
            assert torch.equal(got.int(), want.int()) or (got - want).abs().max() < 1e-4, (trial, got, want)


def test_flow_warp_resize(ops, golden_ops):
    x, fij = T(golden_ops["warp_x"]), T(golden_ops["flow_ij"])
    out = ops.flow_warp(dev(x), dev(fij))
    assert (out.cpu() - T(golden_ops["warp_out"])).abs().max() < 1e-3      # warped-pixel L_inf < 1e-3
    mul = torch.rand(2, 1, 48, 64, generator=g(42))
    out2 = ops.flow_warp(dev(x), dev(fij), dev(mul))
    assert (out2.cpu() - T(golden_ops["warp_out"]) * mul).abs().max() < 1e-3
    big = fij * 50                                                          # mostly out of bounds
    assert (ops.flow_warp(dev(x), dev(big)).cpu() - geom.warp(x, big)).abs().max() < 1e-3
    # non-finite flow (never produced by the path itself): ATen multiplies its zero-masked taps by NaN weights and writes NaN;
    # the kernel deliberately writes 0 there (no tap is in range) so that one bad pixel cannot blank a canvas downstream --
    # everywhere else the two agree
    bad = fij.clone()
    bad[0, 0, 5, 7], bad[1, 1, 9, 3], bad[0, 1, 20, 20] = float("nan"), float("inf"), -float("inf")
    ob, rb = ops.flow_warp(dev(x), dev(bad)).cpu(), geom.warp(x, bad)
    nanpos = ~torch.isfinite(rb)
    assert torch.isfinite(ob).all() and nanpos.sum() == 3 * x.shape[1] and (ob[nanpos] == 0).all()
    assert (ob[~nanpos] - rb[~nanpos]).abs().max() < 1e-3
    r = ops.resize_bilinear(dev(fij), 60, 100, True, div=(64 / 100.0, 48 / 60.0))
    assert (r.cpu() - T(golden_ops["resize_flow_out"])).abs().max() < 1e-5
    rin = T(golden_ops["resize512_in"])
    r5 = ops.resize_bilinear(dev(rin), 512, 512, False)
    assert (r5.cpu() - geom.resize512(rin)).abs().max() < 1e-4


def test_range_map_occlusion_open(ops, golden_ops):
    fji = T(golden_ops["flow_ji"])
    rm = ops.range_map(dev(fji))
    assert (rm.cpu() - T(golden_ops["range_map"])).abs().max() < 1e-5
    rm2 = ops.range_map(dev(fji))
    assert torch.equal(rm, rm2)                                             # deterministic splat
    occ = ops.occlusion_from_range(rm, False)
    assert (occ.cpu() - T(golden_ops["occlusion"])).abs().max() < 1e-5
    hard = ops.occlusion_from_range(rm, True).cpu()
    want = (T(golden_ops["occlusion"]) >= 0.5).float()
    near = (T(golden_ops["occlusion"]) - 0.5).abs() < 1e-5
    assert torch.equal(hard[~near], want[~near])
    m = T(golden_ops["open_in"])
    assert torch.equal(ops.morph_open(dev(m)).cpu(), T(golden_ops["open_out"]))
    m2 = (torch.rand(2, 3, 70, 45, generator=g(43)) > 0.004).float()
    assert torch.equal(ops.morph_open(dev(m2)).cpu(), geom.morph_open19(m2))


def test_tps(ops, golden_ops):
    """UDIS2 TPS transformer (torch_tps_transform.py:7-190) against the REFERENCE's own T and output (tests/golden/ops_small.npz `tps_out`,
    tests/golden/tps_floor.npz `tps_T`; generator: oracle/ref_harness/make_tps_floor_golden.py).
    The contraction T @ grid is the same ascending-k fused chain torch.matmul uses for this shape, the fp64 solve agrees with
    torch.inverse to the last fp32 bit of T (cond(W) = 4e4), linspace / floor / gather are restated operation for operation: the ONE
    thing left is the fp32 log inside r^2 log r^2.  torch-CPU's is MKL VML vsLn, a closed algorithm whose bits depend on the instruction set
    MKL dispatches to (AVX-512: correctly rounded on 99.97 % of inputs, AVX2: 92.8 %); the kernel computes the correctly rounded log
    (csrc/common.h st_logf_cr).  Against the AVX-512 golden 28 of 57 122 kernel entries of W differ by one ulp, which the solve turns into
    1.8e-7 of T and 1.7e-4 px of sample coordinate -- 0.027 grey levels on this 0..255 NOISE image (gradient up to 255 / px).  The reference
    differs from ITSELF by more between two hosts (floor_avx2_*, floor_sse4_2_*: T 1e-5, output 0.05-0.10): the bounds below are that floor,
    not a multiple of this build's own measurement; north_star's 1e-3 is below it for this op."""
    U, src, tgt = T(golden_ops["tps_U"]), T(golden_ops["tps_source"]), T(golden_ops["tps_target"])
    floor = np.load(os.path.join(os.path.dirname(__file__), "golden", "tps_floor.npz"))
    gold_T = T(floor["tps_T"])
    out, Tm, idx = ops.tps_transform(dev(U), dev(src), dev(tgt), (24, 28), want_idx=True)
    relT = (Tm.cpu() - gold_T).abs().max().item() / max(1.0, gold_T.abs().max().item())
    check("tps_T_rel", relT, 1e-6)               # measured 1.8e-7; reference vs itself (AVX2 / SSE4.2 MKL): 1.0e-5
    assert relT < 0.1 * min(float(floor["floor_avx2_tps_T_rel"]), float(floor["floor_sse4_2_tps_T_rel"]))
    # informational: the oracle on THIS host (its torch.log is this host's MKL path)
    ref_out, ref_T = geom.tps_transformer(U, src, tgt, (24, 28))
    host_rel = (ref_T - gold_T).abs().max().item() / gold_T.abs().max().item()
    # sample indices from the reference's T (same _interpolate arithmetic as the homography transformer)
    ref_idx, frac = geom.tps_indices(src, gold_T, U.shape[-2:], (24, 28), return_frac=True)
    mism = (idx.cpu() != ref_idx).any(-1)
    near = frac < 1e-4                            # a sample coordinate within 1e-4 px of an integer may floor either way
    check("tps_idx_mismatches_off_integer", int((mism & ~near).sum()), 0, inclusive=True)
    d = (out.cpu() - T(golden_ops["tps_out"])).abs()
    same = ~mism[:, None].expand_as(d)
    print(f"[tps] T rel vs reference golden {relT:.2e} (this host's oracle vs the golden: {host_rel:.2e}); idx mismatches {int(mism.sum())} / "
          f"{mism.numel()}; |out - golden| max {d.max():.3e} (where idx agree: {d[same].max():.3e})")
    check("tps_idx_mismatches_total", int(mism.sum()), int(near.sum()), inclusive=True)
    ref_floor_max = min(float(floor["floor_avx2_tps_out_max"]), float(floor["floor_sse4_2_tps_out_max"]))        # 0.0515
    ref_floor_p99 = min(float(floor["floor_avx2_tps_out_p99"]), float(floor["floor_sse4_2_tps_out_p99"]))        # 0.0183
    check("tps_out_same_taps_max", d[same].max(), ref_floor_max)      # measured 0.0265 (round 4, ocml logf: 0.0768)
    check("tps_out_p99", np.percentile(d.numpy(), 99), ref_floor_p99)


def test_tps_on_a_photograph(ops):
    """a-18 on IMAGE CONTENT: the UDIS2 TPS transformer (torch_tps_transform.py:7-190) of the reference's own demo photograph
    (demo/demo1/input1.jpg, 512 x 512) through its 169-point mesh with a smooth <= 3 px perturbation, against the reference's output
    (tests/golden/tps_photo_512.npz, oracle/ref_harness/make_r6_goldens.py).  The noise-image case above bounds the worst case (gradient 255 grey
    levels / px: 0.0265); this one says what "warped-pixel L-inf" is on a photograph.  Floor = the reference against itself under AVX2 / SSE4.2 MKL
    on the same input: max 0.0148 / 0.0186, p99 0.0029 / 0.0030 grey levels."""
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "tps_photo_512.npz"))
    img = np.load(os.path.join(os.path.dirname(__file__), "golden", "e2e_demo_512.npz"))["demo1_input1"]
    U = T(img).permute(2, 0, 1).float()[None].contiguous()
    src, tgt = T(G["photo_source"]), T(G["photo_target"])
    out, Tm, _ = ops.tps_transform(dev(U), dev(src), dev(tgt), (512, 512), want_idx=True)
    relT = (Tm.cpu() - T(G["photo_T"])).abs().max().item() / max(1.0, float(np.abs(G["photo_T"]).max()))
    check("tps_photo_T_rel", relT, 1e-6)
    d = (out.cpu()[..., ::4, ::4] - T(G["photo_out_sub"])).abs().numpy()
    fl_max = min(float(G["floor_avx2_photo_out_max"]), float(G["floor_sse4_2_photo_out_max"]))
    fl_p99 = min(float(G["floor_avx2_photo_out_p99"]), float(G["floor_sse4_2_photo_out_p99"]))
    print(f"[tps photo] |out - reference| max {d.max():.3e} p99 {np.percentile(d, 99):.3e} grey levels of 0..255 (reference's own host-to-host floor: "
          f"max {fl_max:.3e} p99 {fl_p99:.3e}; the CPU oracle with the correctly rounded log: {float(G['oracle_vs_reference_out_max']):.3e}); T rel {relT:.2e}")
    check("tps_photo_out_max_grey_levels", d.max(), fl_max)
    check("tps_photo_out_p99_grey_levels", np.percentile(d, 99), fl_p99)
    check("tps_photo_out_sum_rel", abs(out.double().sum().item() - float(G["photo_out_sum"])) / float(G["photo_out_sum"]), 1e-7)


def test_blend_and_eval_finish(ops):
    """Mask algebra + uint8 blend (flowHomoAdpater.py:339-360): every intermediate is one fp32 rounding per torch op in
    the reference, and the kernel keeps exactly those roundings (no contraction): outputs are BIT-EXACT, including the
    uint8 bytes, for binary masks and for the fractional masks the bilinear warps really produce."""
    gg = g(44)
    h, w = 67, 131
    for fractional in (False, True):
        homo1, homo2 = torch.rand(1, 6, h, w, generator=gg) * 255, torch.rand(1, 6, h, w, generator=gg) * 255
        fin = torch.rand(1, 6, h, w, generator=gg) * 255
        for t, thr in ((homo1, 100), (homo2, 100), (fin, 60)):
            t[:, 3:] = torch.rand(1, 1, h, w, generator=gg).expand(-1, 3, -1, -1) if fractional else (t[:, 3:] > thr).float()
        if fractional:                                           # zones of exact 0 / 1 as in real canvases (0/0 -> NaN -> 0)
            homo1[:, 3:, :20] = 0; homo2[:, 3:, :30] = 0; fin[:, 3:, :25] = 0; homo1[:, 3:, 40:] = 1
        occ = (torch.rand(1, 1, h, w, generator=gg) > 0.3).float()
        f = fin * occ
        _, o2r, m1r, m2r, bl = oadapter.blend_canvas(homo1, homo2, f)              # the oracle's restatement of :339-360
        find = dev(fin)
        go2, gm1, gm2, gbl = ops.blend(dev(homo1), dev(homo2), find, dev(occ))
        assert torch.equal(find.cpu(), f)
        assert torch.equal(go2.cpu(), o2r)
        assert torch.equal(gbl.cpu(), bl), (int((gbl.cpu() != bl).sum()), fractional)
        assert torch.equal(gm1.cpu(), m1r)
        assert torch.equal(gm2.cpu(), m2r)
    fin6 = torch.rand(2, 6, h, w, generator=gg)
    occ2 = (torch.rand(2, 1, h, w, generator=gg) > 0.5).float()
    fd = dev(fin6)
    ov = ops.eval_finish(fd, dev(occ2))
    assert torch.equal(ov.cpu(), (fin6[:, 3:6].mean(1) < 0.9).float())
    assert torch.equal(fd.cpu(), fin6 * occ2)
    mt = ops.mean_threshold(dev(fin6[:, 3:6].contiguous()), 0.5)
    assert torch.equal(mt.cpu(), (fin6[:, 3:6].mean(1, keepdim=True) > 0.5).float())


@pytest.mark.parametrize("H1,W1,nl", [(15, 20, 8), (7, 11, 5), (8, 8, 1)])
def test_decoder_token_chain_fused(ops, seeded_sd, H1, W1, nl):
    """fused flow_token_encoder + decoder cross-attention layer vs the oracle's unfused chain (decoder.py:305-312); ragged row counts
    (300 / 77 rows against 64-row workgroups) and fewer than 8 memory tokens (the kernel loads 8 clamped token slots and masks the rest)."""
    R = H1 * W1
    w = nets.W(seeded_sd, "flow_backbone.memory_decoder.")
    ca = w.sub("decoder_layer.cross_attend.")
    gg = g(70)
    cf = torch.randn(1, 81, H1, W1, generator=gg) * 3                       # cost_forward [B,81,H1,W1]
    coords = nets.coords_grid(1, H1, W1) + 2 * torch.randn(1, 2, H1, W1, generator=gg)
    mem = torch.randn(R, nl, 128, generator=gg)
    with torch.no_grad():
        qy = nets.conv(w, "flow_token_encoder.2", F.gelu(nets.conv(w, "flow_token_encoder.0", cf)))
        qy = qy.permute(0, 2, 3, 1).reshape(R, 1, 64)
        k, v = nets.linear(ca, "k", mem), nets.linear(ca, "v", mem)
        ref = nets.decoder_cross_attn(ca, qy, k, v, coords).reshape(R, 64)
    sd = seeded_sd
    p = "flow_backbone.memory_decoder."
    c = p + "decoder_layer.cross_attend."
    names = [(p + "flow_token_encoder.0.weight", (64, 81)), (p + "flow_token_encoder.0.bias", None),
             (p + "flow_token_encoder.2.weight", (64, 64)), (p + "flow_token_encoder.2.bias", None),
             (c + "norm1.weight", None), (c + "norm1.bias", None), (c + "q.weight", None), (c + "q.bias", None),
             (c + "proj.weight", None), (c + "proj.bias", None), (c + "norm2.weight", None), (c + "norm2.bias", None),
             (c + "ffn.0.weight", None), (c + "ffn.0.bias", None), (c + "ffn.3.weight", None), (c + "ffn.3.bias", None)]
    ws = []
    for name, shp in names:
        t = sd[name].reshape(shp) if shp else sd[name]
        if name.endswith("flow_token_encoder.0.weight"):
            t = torch.cat([t, torch.zeros(64, 3)], 1)                            # 81 -> 84 zero-padded columns
        ws.append(dev(t))
    corr = torch.zeros(R, 148, device="cuda")
    corr[:, :81] = dev(cf.permute(0, 2, 3, 1).reshape(R, 81))
    kv = dev(torch.cat([k, v], -1).reshape(R * nl, 128))
    ops.decoder_token_chain(corr, nhwc(coords), kv, ws, R, nl)
    assert (corr[:, 84:].cpu() - ref).abs().max() < 1e-4, (corr[:, 84:].cpu() - ref).abs().max()
    assert torch.equal(corr[:, :81].cpu(), cf.permute(0, 2, 3, 1).reshape(R, 81))


# ---------------------------------------------------------------- operator-level entry points (csrc/operators.hip)
def test_gma_attention_and_aggregate_operators(ops):
    """st_gma_attention / st_gma_aggregate against gma.py:54-76,102-115 written with stock fp64 torch ops."""
    B, H, W = 2, 12, 16
    N = H * W
    inp = torch.randn(B, 128, H, W, generator=g(1))
    mf = torch.randn(B, 128, H, W, generator=g(2))
    w_qk = torch.randn(256, 128, generator=g(3)) / 128 ** 0.5
    w_v = torch.randn(128, 128, generator=g(4)) / 128 ** 0.5
    gamma = torch.tensor([0.37])
    x = inp.double().flatten(2).transpose(1, 2)                              # [B,N,128]
    q, k = (x @ w_qk.double().t()).chunk(2, dim=-1)
    attn_ref = torch.softmax(128 ** -0.5 * q @ k.transpose(1, 2), dim=-1)
    m = mf.double().flatten(2).transpose(1, 2)
    out_ref = m + gamma.double() * (attn_ref @ (m @ w_v.double().t()))
    wide = torch.zeros(B * N, 384, device="cuda")                            # [h | mf | mf_global] like the GRU input rows
    wide[:, 128:256] = nhwc(mf)
    qk = torch.empty(B * N, 256, device="cuda")
    attn = torch.empty(B, N, N, device="cuda")
    ops.gma_attention(nhwc(inp), dev(w_qk), qk, attn, B, N)
    assert (attn.cpu().double() - attn_ref).abs().max() < 1e-6
    vT = torch.empty(B, 128, N, device="cuda")
    ops.gma_aggregate(attn, wide[:, 128:256], dev(w_v), dev(gamma), vT, wide[:, 256:], B, N)
    got = wide[:, 256:].cpu().double().view(B, N, 128)
    assert (got - out_ref).abs().max() < 2e-5 * out_ref.abs().max()
    assert torch.equal(wide[:, :128].cpu(), torch.zeros(B * N, 128))         # neighbours untouched


def test_sepconv_gru_operator(ops):
    """st_sepconv_gru against SepConvGRU.forward (gru.py:44-59) written with stock torch convs (fp64)."""
    B, H, W = 2, 12, 16
    R = B * H * W
    h0 = torch.randn(B, 128, H, W, generator=g(1)).tanh()
    inp = torch.randn(B, 128, H, W, generator=g(2)).relu()
    x = torch.randn(B, 256, H, W, generator=g(3))
    wts = {}
    for i, name in enumerate(("z1", "r1", "q1", "z2", "r2", "q2")):
        k = (1, 5) if name.endswith("1") else (5, 1)
        wts[name] = (torch.randn(128, 512, *k, generator=g(10 + i)) * 0.02, torch.randn(128, generator=g(20 + i)) * 0.1)
    h = h0.double()
    for s, pad in (("1", (0, 2)), ("2", (2, 0))):
        cv = lambda t, n: F.conv2d(t, wts[n][0].double(), wts[n][1].double(), padding=pad)   # noqa: E731
        hx = torch.cat([h, inp.double(), x.double()], 1)
        z, r = torch.sigmoid(cv(hx, "z" + s)), torch.sigmoid(cv(hx, "r" + s))
        q = torch.tanh(cv(torch.cat([r * h, inp.double(), x.double()], 1), "q" + s))
        h = (1 - z) * h + z * q

    def pack_hx(w):                      # drop the constant `inp` channels, K ordered (tap, channel)
        return pack_conv_w(torch.cat([w[:, :128], w[:, 256:]], 1))

    def table(s, pad):                   # conv over `inp` + bias for [z | r | q]
        return nhwc(torch.cat([F.conv2d(inp, wts[n + s][0][:, 128:256], wts[n + s][1], padding=pad) for n in "zrq"], 1))

    hxA, hxB = torch.zeros(R, 384, device="cuda"), torch.zeros(R, 384, device="cuda")
    hxA[:, :128], hxA[:, 128:] = nhwc(h0), nhwc(x)
    hxB[:, 128:] = float("nan")             # never read: the q conv takes its x channels from hxA (second A source)
    zbuf = torch.empty(R, 128, device="cuda")
    ops.sepconv_gru(hxA, hxB, zbuf, table("1", (0, 2)), table("2", (2, 0)),
                    torch.cat([pack_hx(wts["z1"][0]), pack_hx(wts["r1"][0])]), pack_hx(wts["q1"][0]),
                    torch.cat([pack_hx(wts["z2"][0]), pack_hx(wts["r2"][0])]), pack_hx(wts["q2"][0]), B, H, W)
    got = from_rows(hxA[:, :128], B, H, W).double()
    assert (got - h).abs().max() < 2e-5, (got - h).abs().max()
    assert torch.equal(hxA[:, 128:].cpu(), nhwc(x).cpu())


def test_patch_embed_operator_ragged(ops):
    """st_patch_embed on a cost map whose side is not a multiple of 8 (zero pad, encoder.py:63-66) vs torch fp64."""
    M, H, W = 5, 12, 20
    cm = torch.randn(M, 1, H, W, generator=g(1))
    mk = lambda *s, sc=0.1, seed=0: torch.randn(*s, generator=g(seed)) * sc          # noqa: E731
    c0, b0, c2, b2, c4, b4 = mk(16, 1, 6, 6, seed=2), mk(16, seed=3), mk(32, 16, 6, 6, seed=4, sc=0.05), mk(32, seed=5), \
        mk(64, 32, 6, 6, seed=6, sc=0.03), mk(64, seed=7)
    f0, fb0, f2, fb2, lw, lb = mk(128, 128, seed=8), mk(128, seed=9), mk(128, 128, seed=10), mk(128, seed=11), \
        1 + mk(128, seed=12), mk(128, seed=13)
    Hp, Wp = (H + 7) // 8 * 8, (W + 7) // 8 * 8
    x = F.pad(cm.double(), (0, Wp - W, 0, Hp - H))
    x = F.relu(F.conv2d(x, c0.double(), b0.double(), stride=2, padding=2))
    x = F.relu(F.conv2d(x, c2.double(), b2.double(), stride=2, padding=2))
    x = F.conv2d(x, c4.double(), b4.double(), stride=2, padding=2)
    h, w = x.shape[-2:]
    P = h * w
    pe_in = torch.zeros(P, 64, device="cuda")
    ops.sine_pe(pe_in, 64, Wg=w, cscale=8.0, coff=4.0)                               # encoder.py:77-80 (own parity test above)
    tok = torch.cat([x.flatten(2).transpose(1, 2), pe_in.cpu().double()[None].expand(M, P, 64)], -1)
    tok = F.relu(tok @ f0.double().t() + fb0.double()) @ f2.double().t() + fb2.double()
    ref = F.layer_norm(tok, (128,), lw.double(), lb.double(), 1e-5)
    pe_bias = (pe_in.cpu() @ f0[:, 64:].t() + fb0).cuda().contiguous()
    w11 = [dev(c0.reshape(16, 36).t()), dev(b0), pack_conv_w(c2), dev(b2), pack_conv_w(c4), dev(b4), dev(f0), dev(f2), dev(fb2),
           dev(lw), dev(lb)]
    s1, s2 = torch.empty(M * (Hp // 2) * (Wp // 2), 16, device="cuda"), torch.empty(M * (Hp // 4) * (Wp // 4), 32, device="cuda")
    s3, s4, out = torch.empty(M * P, 64, device="cuda"), torch.empty(M * P, 128, device="cuda"), torch.empty(M * P, 128, device="cuda")
    ops.patch_embed(dev(cm.reshape(M, H * W)), w11, 128, pe_bias, s1, s2, s3, s4, out, M, H, W)
    assert (out.cpu().double().view(M, P, 128) - ref).abs().max() < 1e-4


def test_patch_conv12_fused_is_bit_identical(ops):
    """st_patch_conv12 (PatchEmbed's c0 + ReLU + c2 + ReLU per 64x64 cost map in one launch, the first feature map kept in LDS) against the two
    launches it replaces -- st_patch_conv1 + the (6x3 pixel-pair) st_conv_gemm of st_patch_embed on the LDS-DMA kernel: same products, same k
    order, same K-block folds, so the [M*256, 32] result must be torch.equal (M = 300: 600 half-map units on 512 workgroups, every workgroup walks
    more than one); both against torch fp64 (encoder.py:36-39,68-72).  A map's result does not depend on its neighbours: M = 1 and M = 3 (fewer
    units than workgroup slots, odd counts) must reproduce the first maps of the big launch bit for bit (the unfused GEMM picks other kernels
    at such row counts, so it is not the yardstick there)."""
    M = 300
    cm = torch.randn(M, 1, 64, 64, generator=g(21)) * 3
    mk = lambda *s, sc=0.1, seed=0: torch.randn(*s, generator=g(seed)) * sc          # noqa: E731
    c0, b0, c2, b2 = mk(16, 1, 6, 6, seed=2), mk(16, seed=3), mk(32, 16, 6, 6, seed=4, sc=0.05), mk(32, seed=5)
    ref = F.relu(F.conv2d(F.relu(F.conv2d(cm.double(), c0.double(), b0.double(), stride=2, padding=2)), c2.double(), b2.double(), stride=2, padding=2))
    ref = ref.permute(0, 2, 3, 1).reshape(M * 256, 32)
    w0, w2 = dev(c0.reshape(16, 36).t()), pack_conv_w(c2)
    maps = dev(cm.reshape(M, 4096))
    s1 = torch.empty(M * 32 * 32, 16, device="cuda")
    ops.patch_conv1(maps, w0, dev(b0), s1, M, 64, 64, 32, 32)
    two = torch.empty(M * 256, 32, device="cuda")
    ops.conv_gemm(s1.view(-1, 32), w2, two, geom=(M, 32, 16, 6, 3, 2, 1, 2, 1), bias=dev(b2), act="relu")
    one = torch.full((M * 256, 32), float("nan"), device="cuda")
    ops.patch_conv12(maps, w0, dev(b0), w2, dev(b2), one, M)
    assert (two.cpu().double() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())
    assert torch.isfinite(one).all()
    assert torch.equal(one, two), (one - two).abs().max().item()
    for m in (1, 3):
        few = torch.full((m * 256, 32), float("nan"), device="cuda")
        ops.patch_conv12(maps[:m].contiguous(), w0, dev(b0), w2, dev(b2), few, m)
        assert torch.equal(few, one[:m * 256]), m
    again = torch.empty_like(one)
    ops.patch_conv12(maps, w0, dev(b0), w2, dev(b2), again, M)
    assert torch.equal(again, one)
    with pytest.raises(ops.StitchErrorBase):
        ops.patch_conv12(maps, w0, dev(b0), w2, dev(b2), one, M, 32, 64)              # only the 64x64 configuration is built


# ---------------------------------------------------------------- LDS-DMA pipelined GEMM kernels (tile 12 / 13 / 14)
@pytest.mark.parametrize("tile,Co", [(12, 130), (13, 130), (14, 24)])
@pytest.mark.parametrize("geo", [dict(kh=3, kw=3, p=(1, 1), C=64, split=0), dict(kh=1, kw=5, p=(0, 2), C=384, split=3),
                                 dict(kh=4, kw=4, p=(0, 0), C=32, split=1, s=4), dict(kh=6, kw=3, p=(2, 1), C=32, split=0, s=(2, 1))])
def test_gemm_dma_pipelined_conv(ops, tile, Co, geo):
    """global -> LDS DMA ring, swizzled LDS image, zero-fill of padded taps, ragged M / N, split-K slabs."""
    B, H, W, C, kh, kw = 2, 20, 24, geo["C"], geo["kh"], geo["kw"]
    st = geo.get("s", 1)
    st = st if isinstance(st, tuple) else (st, st)
    x = torch.randn(B, C, H, W, generator=g(10))
    w = torch.randn(Co, C, kh, kw, generator=g(11)) / (C * kh * kw) ** 0.5
    b = torch.randn(Co, generator=g(12))
    conv = F.conv2d(x.double(), w.double(), b.double(), stride=st, padding=geo["p"])
    Ho, Wo = conv.shape[2:]
    res = torch.randn(B * Ho * Wo, Co, generator=g(13))
    ref = F.relu(conv).permute(0, 2, 3, 1).reshape(-1, Co) + res.double()
    out = torch.empty(B * Ho * Wo, Co + 5, device="cuda")[:, :Co]
    ops.conv_gemm(nhwc(x), pack_conv_w(w), out, geom=(B, H, W, kh, kw, st[0], st[1], geo["p"][0], geo["p"][1]), bias=dev(b),
                  act="relu", epi="add", aux1=dev(res), tile=tile, split_k=geo["split"])
    assert (out.cpu().double() - ref).abs().max() < 3e-5


@pytest.mark.parametrize("M,N,K", [(70001, 130, 256), (65536, 128, 128), (33000, 24, 512)])
def test_gemm_dma_persistent_walk(ops, M, N, K):
    """plain matrices with many M tiles: one workgroup walks tiles g, g+G, ... with a continuous DMA ring."""
    xw = torch.randn(M, K + 32, generator=g(1))
    w, b = torch.randn(N, K, generator=g(2)) / K ** 0.5, torch.randn(N, generator=g(3))
    res = torch.randn(M, N, generator=g(4))
    ref = F.gelu(F.linear(xw[:, 16:16 + K].double(), w.double(), b.double())) + res.double()
    out = torch.empty(M, N, device="cuda")
    ops.conv_gemm(dev(xw)[:, 16:16 + K], dev(w), out, bias=dev(b), act="gelu", epi="add", aux1=dev(res))
    assert (out.cpu().double() - ref).abs().max() < 3e-5
    out2 = torch.empty(M, N, device="cuda")
    ops.conv_gemm(dev(xw)[:, 16:16 + K], dev(w), out2, bias=dev(b), act="gelu", epi="add", aux1=dev(res), tile=3)
    assert torch.equal(out, out2)            # same k order as the register-staged kernel: bit-identical


@pytest.mark.parametrize("H,W,C", [(32, 32, 256), (9, 13, 256), (16, 16, 128)])
def test_narrow_conv(ops, H, W, C):
    """N <= 4 (flow head conv2, gru.py:5-13), coords += delta epilogue: the 3x3 / 256-channel kernel (8 pixels per wave, weights in
    registers), ragged rows, and the generic one-wave-per-pixel kernel (C = 128)."""
    B = 2
    x = torch.randn(B, C, H, W, generator=g(1))
    w, b = torch.randn(2, C, 3, 3, generator=g(2)) / (9 * C) ** 0.5, torch.randn(2, generator=g(3))
    coords = torch.randn(B * H * W, 2, generator=g(4))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, 2) + coords.double()
    c = dev(coords)
    ops.conv_gemm(nhwc(x), pack_conv_w(w), c, geom=(B, H, W, 3, 3, 1, 1, 1, 1), bias=dev(b), epi="add", aux1=c)
    assert (c.cpu().double() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("M,P", [(37, 64), (50, 4), (9, 30)])
def test_latent_pool(ops, M, P):
    """st_latent_pool: per pixel softmax over its P tokens for 64 score rows, then z = softmax(S)^T . T (fp64 torch check)."""
    S = torch.randn(M * P, 64, generator=g(1)) * 3
    T_ = torch.randn(M * P, 128, generator=g(2))
    p = torch.softmax(S.double().view(M, P, 64), dim=1)                       # over tokens
    ref = torch.einsum("mtr,mtc->mrc", p, T_.double().view(M, P, 128))       # [M, 64, 128]
    Sw = torch.zeros(M * P, 72, device="cuda"); Sw[:, 4:68] = S.cuda()        # column slices of wider buffers
    z = torch.empty(M * 64, 128, device="cuda")
    ops.latent_pool(Sw[:, 4:68], dev(T_), z, M, P)
    assert (z.cpu().double().view(M, 64, 128) - ref).abs().max() < 2e-5


@pytest.mark.parametrize("M,N,K,mode", [(70001, 130, 128, "res"), (65536, 384, 128, "div8"), (40000, 128, 64, "mod64"), (33, 512, 128, "gelu"),
                                        (300000, 64, 128, "none"), (4100, 96, 64, "axpy")])
def test_gemm_rowstream(ops, M, N, K, mode):
    """row-streaming kernel (weights resident in LDS, A rows in registers; tile 20): every epilogue form it accepts, ragged M and N,
    column-slice operands; bit-identical to the register-staged kernel (same k pairing and summation order)."""
    xw = torch.randn(M, K + 32, generator=g(1))
    x = dev(xw)[:, 16:16 + K]
    w, b = torch.randn(N, K, generator=g(2)) / K ** 0.5, torch.randn(N, generator=g(3))
    kw, ref = dict(bias=dev(b)), F.linear(xw[:, 16:16 + K].double(), w.double(), b.double())
    if mode == "res":
        res = torch.randn(M, N, generator=g(4))
        kw.update(aux0=dev(res), act="relu")
        ref = F.relu(ref + res.double())
    elif mode == "div8":
        tab = torch.randn((M + 7) // 8, N, generator=g(4))
        kw.update(aux0=dev(tab), row_div=8)
        ref = ref + tab.double()[torch.arange(M) // 8]
    elif mode == "mod64":
        tab = torch.randn(64, N, generator=g(4))
        kw.update(aux0=dev(tab), row_mod=64, act="relu")
        ref = F.relu(ref + tab.double()[torch.arange(M) % 64])
    elif mode == "gelu":
        kw.update(act="gelu")
        ref = F.gelu(ref)
    elif mode == "axpy":
        h, gam = torch.randn(M, N, generator=g(4)), torch.tensor([0.37])
        kw.update(epi="axpy", aux1=dev(h), scale_ptr=dev(gam), alpha=0.5)
        ref = h.double() + 0.37 * (0.5 * F.linear(xw[:, 16:16 + K].double(), w.double()) + b.double())
    out = torch.full((M, N + 8), 7.0, device="cuda")
    ops.conv_gemm(x, dev(w), out[:, 4:4 + N], tile=20, **kw)
    assert (out[:, 4:4 + N].cpu().double() - ref).abs().max() < 3e-5
    assert (out[:, :4] == 7.0).all() and (out[:, 4 + N:] == 7.0).all()              # nothing written outside the column slice
    out2 = torch.empty(M, N, device="cuda")
    ops.conv_gemm(x, dev(w), out2, tile=3, **kw)
    assert torch.equal(out[:, 4:4 + N], out2)


@pytest.mark.parametrize("M,N,K,eps", [(65536, 512, 128, 1e-5), (777, 384, 128, 1e-6), (40000, 128, 64, 1e-5)])
def test_gemm_layernorm_prologue(ops, M, N, K, eps):
    """Linear(LayerNorm(x)) with the normalisation done on the A rows inside the row-streaming kernel (twins.py:787-790,
    encoder.py:156-172) and gamma / beta folded into the weights, against fp64 torch."""
    x = torch.randn(M, K, generator=g(1)) * 3 + 1
    gam, bet = torch.rand(K, generator=g(2)) + 0.5, torch.randn(K, generator=g(3))
    w, b = torch.randn(N, K, generator=g(4)) / K ** 0.5, torch.randn(N, generator=g(5))
    ref = F.gelu(F.linear(F.layer_norm(x.double(), (K,), gam.double(), bet.double(), eps), w.double(), b.double()))
    wf, bf = ops.fold_layernorm(dev(gam), dev(bet), dev(w), dev(b))
    out = torch.empty(M, N, device="cuda")
    ops.conv_gemm(dev(x), wf, out, bias=bf, act="gelu", ln_eps=eps)
    assert (out.cpu().double() - ref).abs().max() < 3e-5
    # the unfused chain (LayerNorm kernel + GEMM) agrees to rounding
    y, out2 = torch.empty(M, K, device="cuda"), torch.empty(M, N, device="cuda")
    ops.layernorm(dev(x), dev(gam), dev(bet), y, eps)
    ops.conv_gemm(y, dev(w), out2, bias=dev(b), act="gelu")
    assert (out - out2).abs().max() < 2e-5
    with pytest.raises(Exception):                                                 # a shape the kernel does not take is rejected, not mis-computed
        ops.conv_gemm(torch.empty(64, 256, device="cuda"), torch.empty(32, 256, device="cuda"), torch.empty(64, 32, device="cuda"), ln_eps=eps)


def test_gelu_epilogue_accuracy(ops):
    """st_gelu (erfc polynomial, branch-free) against fp64 x * Phi(x) over [-9, 9]: as close as torch's own fp32 CPU GELU."""
    n = 1 << 16
    xs = torch.linspace(-9, 9, n)
    a = torch.zeros(n, 32); a[:, 0] = xs
    w = torch.zeros(32, 32); w[:, 0] = 1.0
    out = torch.empty(n, 32, device="cuda")
    ops.conv_gemm(dev(a), dev(w), out, act="gelu", tile=3)
    ref = xs.double() * 0.5 * (1 + torch.erf(xs.double() / 2 ** 0.5))
    err = (out[:, 0].cpu().double() - ref).abs()
    cpu = (F.gelu(xs).double() - ref).abs()
    assert err.max() < 6e-7, err.max()
    assert err.max() <= 1.5 * cpu.max() + 1e-7 and err.mean() <= 1.5 * cpu.mean() + 1e-9


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 13, 22)])
def test_flow_encode(ops, B, H, W):
    """flow = coords1 - coords0 and relu(convf1(flow)) (Conv2d(2, 128, 7, padding=3); gru.py:251, decoder.py:321) in one kernel,
    against fp64 torch; ragged tiles; the flow lands in two columns of a wider buffer."""
    flow = torch.randn(B, 2, H, W, generator=g(1)) * 4
    wt, b = torch.randn(128, 2, 7, 7, generator=g(2)) / 98 ** 0.5, torch.randn(128, generator=g(3))
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    coords1 = (flow + torch.stack([xs, ys])[None]).permute(0, 2, 3, 1).reshape(-1, 2)
    ref = F.relu(F.conv2d(flow.double(), wt.double(), b.double(), padding=3)).permute(0, 2, 3, 1).reshape(-1, 128)
    out = torch.full((B * H * W, 136), 7.0, device="cuda")
    wide = torch.full((B * H * W, 8), 7.0, device="cuda")
    ops.flow_encode(dev(coords1), dev(wt.permute(2, 3, 1, 0).reshape(98, 128)), dev(b), out[:, 4:132], wide[:, 5:7], B, H, W)
    assert (out[:, 4:132].cpu().double() - ref).abs().max() < 2e-5
    assert (out[:, :4] == 7.0).all() and (out[:, 132:] == 7.0).all()
    assert (wide[:, 5:7].cpu() - flow.permute(0, 2, 3, 1).reshape(-1, 2)).abs().max() < 1e-5 and (wide[:, :5] == 7.0).all() and (wide[:, 7] == 7.0).all()


@pytest.mark.parametrize("M", [65536, 4100, 33])
def test_linear_chain128(ops, M):
    """st_linear_chain128: proj + residual -> LayerNorm -> ffn.0 + GELU -> ffn.3 + residual (crossattentionlayer.py:46-56,
    encoder.py:163-172) in one launch, against fp64 torch and BIT-IDENTICAL to the unfused chain of st_conv_gemm launches."""
    gg = g(5)
    att, x = torch.randn(M, 128, generator=gg), torch.randn(M, 128, generator=gg)
    lin = lambda: (torch.randn(128, 128, generator=gg) / 128 ** 0.5, torch.randn(128, generator=gg) * 0.1)      # noqa: E731
    (wp, bp), (w0, b0), (w3, b3) = lin(), lin(), lin()
    gam, bet = torch.rand(128, generator=gg) + 0.5, torch.randn(128, generator=gg) * 0.1
    x1 = F.linear(att.double(), wp.double(), bp.double()) + x.double()
    h = F.gelu(F.linear(F.layer_norm(x1, (128,), gam.double(), bet.double(), 1e-5), w0.double(), b0.double()))
    ref = F.linear(h, w3.double(), b3.double()) + x1
    w0f, b0f = ops.fold_layernorm(dev(gam), dev(bet), dev(w0), dev(b0))
    attw = torch.zeros(M, 136, device="cuda")
    attw[:, 4:132] = att.cuda()
    out = torch.full((M, 136), 7.0, device="cuda")
    ops.linear_chain128(attw[:, 4:132], out[:, 4:132], [dict(w=dev(wp), bias=dev(bp), res=dev(x)),
                                                         dict(w=w0f, bias=b0f, act="gelu", ln_eps=1e-5),
                                                         dict(w=dev(w3), bias=dev(b3), res=1)])
    assert (out[:, 4:132].cpu().double() - ref).abs().max() < 5e-5
    assert (out[:, :4] == 7.0).all() and (out[:, 132:] == 7.0).all()
    # the unfused chain: three launches
    x1u, hu, ou = (torch.empty(M, 128, device="cuda") for _ in range(3))
    ops.conv_gemm(att.cuda(), dev(wp), x1u, bias=dev(bp), aux0=dev(x))
    ops.conv_gemm(x1u, w0f, hu, bias=b0f, act="gelu", ln_eps=1e-5)
    ops.conv_gemm(hu, dev(w3), ou, bias=dev(b3), aux0=x1u)
    assert torch.equal(out[:, 4:132], ou)
    # two-layer form (LayerNorm -> ffn.0 -> ffn.3 + input)
    o2 = torch.empty(M, 128, device="cuda")
    ops.linear_chain128(x1u, o2, [dict(w=w0f, bias=b0f, act="gelu", ln_eps=1e-5), dict(w=dev(w3), bias=dev(b3), res=0)])
    assert torch.equal(o2, ou)


@pytest.mark.parametrize("M,hidden", [(2048, 512), (4099, 512), (96, 256), (17, 32), (70001, 512)])      # 70001 rows: a wave walks two row blocks (grid capped at 512 workgroups)
def test_mlp128_fused(ops, M, hidden):
    """st_mlp128: x + fc2(GELU(fc1(LN(x)))) [+ second residual] of the C = 128 Twins / vertical-layer MLPs (twins.py:785-790) in ONE
    launch, the hidden activations staying on the CU: against fp64 torch, against the unfused launches (fc1 + GELU bit-identical by
    construction; fc2 sums its k in one chain where st_conv_gemm folds every 256: last-bit differences only), ragged M, column
    slices of wider buffers, no-LayerNorm form, and the rejected descriptors."""
    gg = g(11)
    x, extra = torch.randn(M, 128, generator=gg) * 1.5, torch.randn(M, 128, generator=gg)
    w1, b1 = torch.randn(hidden, 128, generator=gg) / 128 ** 0.5, torch.randn(hidden, generator=gg) * 0.1
    w2, b2 = torch.randn(128, hidden, generator=gg) / hidden ** 0.5, torch.randn(128, generator=gg) * 0.1
    gam, bet = torch.rand(128, generator=gg) + 0.5, torch.randn(128, generator=gg) * 0.1
    xd = x.double()
    h = F.gelu(F.linear(F.layer_norm(xd, (128,), gam.double(), bet.double(), 1e-6), w1.double(), b1.double()))
    ref = F.linear(h, w2.double(), b2.double()) + xd
    w1f, b1f = ops.fold_layernorm(dev(gam), dev(bet), dev(w1), dev(b1))
    xw = torch.zeros(M, 136, device="cuda")
    xw[:, 4:132] = x.cuda()
    out = torch.full((M, 136), 7.0, device="cuda")
    ops.mlp128(xw[:, 4:132], out[:, 4:132], w1f, b1f, dev(w2), dev(b2), ln_eps=1e-6)
    scale = ref.abs().max().item()
    check(f"mlp128_vs_fp64_rel_{M}_{hidden}", (out[:, 4:132].cpu().double() - ref).abs().max() / scale, {512: 1.6e-6, 256: 1.1e-6, 32: 3.5e-7}[hidden])      # measured 5.1e-7 / 3.5e-7 / 1.1e-7 (hidden 512 / 256 / 32)
    assert (out[:, :4] == 7.0).all() and (out[:, 132:] == 7.0).all()
    # the unfused launches
    hu, ou = torch.empty(M, hidden, device="cuda"), torch.empty(M, 128, device="cuda")
    ops.conv_gemm(dev(x), w1f, hu, bias=b1f, act="gelu", ln_eps=1e-6)
    ops.conv_gemm(hu, dev(w2), ou, bias=dev(b2), aux0=dev(x))
    if hidden <= 256:
        assert torch.equal(out[:, 4:132], ou)              # no fold below 256 k: the same chain, the same bits
    else:
        check(f"mlp128_vs_unfused_rel_{M}_{hidden}", (out[:, 4:132] - ou).abs().max().item() / scale, 1e-6)
    # second residual (the cost-memory short-cut in the last vertical layer, encoder.py:281) + the no-LayerNorm form
    o2, o2u = torch.empty(M, 128, device="cuda"), torch.empty(M, 128, device="cuda")
    ops.mlp128(dev(x), o2, dev(w1), dev(b1), dev(w2), dev(b2), res=dev(extra))
    ops.conv_gemm(dev(x), dev(w1), hu, bias=dev(b1), act="gelu")
    ops.conv_gemm(hu, dev(w2), o2u, bias=dev(b2), aux0=dev(x), epi="add", aux1=dev(extra))
    ref2 = F.linear(F.gelu(F.linear(xd, w1.double(), b1.double())), w2.double(), b2.double()) + xd + extra.double()
    assert (o2.cpu().double() - ref2).abs().max() / ref2.abs().max() < 6e-6
    assert (o2 - o2u).abs().max().item() / scale < 1e-6
    # the Block's attention output projection + residual in front (twins.py:622-623): x = att @ wp^T + bp + x0 inside the same launch
    att, x0 = torch.randn(M, 128, generator=gg), torch.randn(M, 128, generator=gg)
    wp, bp = torch.randn(128, 128, generator=gg) / 128 ** 0.5, torch.randn(128, generator=gg) * 0.1
    xu, o3, o3u = (torch.empty(M, 128, device="cuda") for _ in range(3))
    ops.conv_gemm(dev(att), dev(wp), xu, bias=dev(bp), aux0=dev(x0))
    ops.mlp128(xu, o3u, w1f, b1f, dev(w2), dev(b2), ln_eps=1e-6, res=dev(extra))
    ops.mlp128(dev(att), o3, w1f, b1f, dev(w2), dev(b2), ln_eps=1e-6, res=dev(extra), proj=(dev(wp), dev(bp), dev(x0)))
    assert torch.equal(o3, o3u)                              # same products, same order: the same bits as projection launch + MLP launch
    o4, o4u = torch.empty(M, 128, device="cuda"), torch.empty(M, 128, device="cuda")
    ops.conv_gemm(dev(att), dev(wp), xu)                     # no bias, no residual
    ops.mlp128(xu, o4u, dev(w1), dev(b1), dev(w2), dev(b2))
    ops.mlp128(dev(att), o4, dev(w1), dev(b1), dev(w2), dev(b2), proj=(dev(wp), None, None))
    assert torch.equal(o4, o4u)
    if M == 96:
        with pytest.raises(ops.StitchErrorBase):
            xc = dev(x)
            ops.mlp128(xc, xc, dev(w1), dev(b1), dev(w2), dev(b2))                            # in place: rejected
        with pytest.raises(ops.StitchErrorBase):
            ops.mlp128(dev(x), o2, dev(w1), torch.zeros(hidden + 4, device="cuda")[1:hidden + 1], dev(w2), dev(b2))     # bias not 16-byte aligned


@pytest.mark.parametrize("kh,kw,ph,pw", [(1, 5, 0, 2), (3, 3, 1, 1)])
def test_gemm_second_a_source(ops, kh, kw, ph, pw):
    """st_gemm_desc.a2: input channels < a2_channels come from a second buffer of the same geometry (SepConvGRU's q conv reads
    [r*h | x] without x being copied, gru.py:50) -- identical to the conv over the concatenated buffer; rejected where the
    LDS-DMA kernel cannot run."""
    B, H, W, C, Co = 2, 24, 20, 384, 128
    gg = g(9)
    xa, xb = torch.randn(B * H * W, C, generator=gg), torch.randn(B * H * W, C, generator=gg)
    w = torch.randn(Co, kh * kw * C, generator=gg) / (kh * kw * C) ** 0.5
    cat = xa.clone()
    cat[:, :128] = xb[:, :128]
    geom = (B, H, W, kh, kw, 1, 1, ph, pw)
    ref, out = torch.empty(B * H * W, Co, device="cuda"), torch.empty(B * H * W, Co, device="cuda")
    ops.conv_gemm(dev(cat), dev(w), ref, geom=geom, act="tanh")
    xbn = xb.clone()
    xbn[:, 128:] = float("nan")                                  # the part of the second buffer that must never be read
    ops.conv_gemm(dev(xa), dev(w), out, geom=geom, act="tanh", a2=dev(xbn), a2_channels=128)
    assert torch.equal(out, ref)
    with pytest.raises(Exception):
        ops.conv_gemm(dev(xa), dev(w), out, geom=geom, a2=dev(xbn), a2_channels=100)         # not a whole number of 32-channel steps
    with pytest.raises(Exception):
        ops.conv_gemm(dev(xa), dev(w), out, geom=geom, a2=dev(xbn), a2_channels=128, tile=3)  # register-staged kernel: no second source


def test_conv_gemm_pair(ops):
    """st_conv_gemm_pair: two independent convs in one launch (BasicMotionEncoder's convc2 + convf2, gru.py:252-253): bit-identical
    to the two separate launches, column-slice outputs, different K / N / epilogues; shapes the LDS-DMA kernel cannot take are rejected."""
    B, H, W = 2, 24, 20
    gg = g(11)
    xa, xb = torch.randn(B * H * W, 256, generator=gg), torch.randn(B * H * W, 128, generator=gg)
    wa, ba = torch.randn(192, 9 * 256, generator=gg) / 48, torch.randn(192, generator=gg)
    wb, bb = torch.randn(64, 9 * 128, generator=gg) / 34, torch.randn(64, generator=gg)
    res = torch.randn(B * H * W, 64, generator=gg)
    geom = (B, H, W, 3, 3, 1, 1, 1, 1)
    ref, out = torch.zeros(B * H * W, 256, device="cuda"), torch.zeros(B * H * W, 256, device="cuda")
    ops.conv_gemm(dev(xa), dev(wa), ref[:, :192], geom=geom, bias=dev(ba), act="relu", split_k=1)
    ops.conv_gemm(dev(xb), dev(wb), ref[:, 192:], geom=geom, bias=dev(bb), act="relu", epi="add", aux1=dev(res), split_k=1)
    ops.conv_gemm_pair((dev(xa), dev(wa), out[:, :192], dict(geom=geom, bias=dev(ba), act="relu")),
                       (dev(xb), dev(wb), out[:, 192:], dict(geom=geom, bias=dev(bb), act="relu", epi="add", aux1=dev(res))))
    assert torch.equal(out, ref)
    want = F.relu(F.conv2d(xa.view(B, H, W, 256).permute(0, 3, 1, 2).double(), wa.view(192, 3, 3, 256).permute(0, 3, 1, 2).double(), ba.double(), padding=1))
    assert (out[:, :192].cpu().double() - want.permute(0, 2, 3, 1).reshape(-1, 192)).abs().max() < 3e-5
    with pytest.raises(Exception):                                                  # Cin = 100: not a whole number of 32-channel steps
        ops.conv_gemm_pair((dev(xa), dev(wa), out[:, :192], dict(geom=geom)),
                           (torch.zeros(B * H * W, 100, device="cuda"), torch.zeros(64, 900, device="cuda"), out[:, 192:], dict(geom=geom)))


@pytest.mark.parametrize("B,N,C", [(2, 4096, 256), (1, 192, 256), (3, 100, 64)])
def test_corr_volume_both(ops, B, N, C):
    """st_corr_volume_both: the reverse direction's all-pairs volume (encoder.py:359-369 for (f2, f1)) is the transpose of the forward
    one and comes out of the same launch through a transposed second store -- bit-identical to two st_corr_volume launches."""
    gg = g(21)
    f1, f2 = torch.randn(B, N, C, generator=gg).cuda(), torch.randn(B, N, C, generator=gg).cuda()
    r12, r21 = torch.empty(B, N, N, device="cuda"), torch.empty(B, N, N, device="cuda")
    ops.corr_volume(f1, f2, r12)
    ops.corr_volume(f2, f1, r21)
    both = torch.full((2 * B, N, N), 7.0, device="cuda")
    ops.corr_volume_both(f1, f2, both[:B], both[B:])
    assert torch.equal(both[:B], r12) and torch.equal(both[B:], r21)
    assert torch.equal(both[B:], r12.transpose(1, 2))
