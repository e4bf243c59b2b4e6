"""The exact-split contraction (st_gemm_desc.split3, csrc/gemm_split3.h) through the C-ABI: plane format, the six-product kernel against
the fp32-MFMA kernel and the fp64 product, plane emission by every epilogue mode, the split3 operators against their fp32 twins.

Bar: the split is EXACT (hi + mid + lo == x bit for bit); the contraction's error against fp64 is no larger than 1.25x the fp32 kernel's
(measured 0.77-0.87x, profiles/r6_split3_probe.json); emitted planes equal st_split3_pack of the fp32 result bit for bit."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from _measure import check  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    import stitch_amd
    assert torch.cuda.is_available()
    return stitch_amd.ops


def g(seed=0):
    return torch.Generator().manual_seed(seed)


def dev(t):
    return t.cuda().contiguous()


def unpack(planes, rows=None):
    """Planes -> fp64 [rows, ncols] = hi + mid + lo (exact in fp64)"""
    t = planes.t[:, planes.c0 // 32:(planes.c0 + planes.ncols + 31) // 32].double().sum(0)       # [chunks, rows, 32]
    x = t.permute(1, 0, 2).reshape(planes.rows, -1)[:, :planes.ncols]
    return x if rows is None else x[:rows]


def test_split3_pack_is_exact(ops):
    x = torch.randn(300, 96, generator=g(1)) * torch.logspace(-20, 20, 96)[None, :]
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.4e38, -3.4e38, 1e-30, 2.0 ** -100])
    p = ops.split3_pack(dev(x))
    assert torch.equal(unpack(p).cpu(), x.double())
    # the three parts are bf16 with decreasing magnitude: |mid| <= 2^-8 |hi|, |lo| <= 2^-16 |hi| (+ a rounding ulp)
    hi, mid, lo = (p.t[i].float().abs() for i in range(3))
    assert bool((mid <= hi * 2.0 ** -7).all()) and bool((lo <= hi * 2.0 ** -15).all())
    # non-finite values stay in hi alone
    y = torch.tensor([[float("inf"), float("-inf"), float("nan")] + [1.0] * 29])
    q = ops.split3_pack(dev(y))
    assert torch.isinf(q.t[0, 0, 0, :2]).all() and torch.isnan(q.t[0, 0, 0, 2]) and bool((q.t[1:, 0, 0, :3] == 0).all())


def conv_ref64(x, w, B, H, W, Cin, kh, kw):
    N = w.shape[0]
    x64 = x.double().view(B, H, W, Cin)
    w64 = w.double().view(N, kh, kw, Cin)
    ref = torch.zeros(B, H, W, N, dtype=torch.float64, device=x.device)
    for ky in range(kh):
        for kx in range(kw):
            dy, dx = ky - kh // 2, kx - kw // 2
            ylo, yhi, xlo, xhi = max(0, -dy), min(H, H - dy), max(0, -dx), min(W, W - dx)
            ref[:, ylo:yhi, xlo:xhi] += x64[:, ylo + dy:yhi + dy, xlo + dx:xhi + dx] @ w64[:, ky, kx].t()
    return ref.view(-1, N)


@pytest.mark.parametrize("B,H,W,Cin,N,kh,kw,tile,split", [
    (2, 16, 16, 128, 128, 1, 5, 0, 0), (2, 16, 16, 128, 128, 5, 1, 34, 1), (1, 24, 40, 64, 126, 3, 3, 34, 1), (1, 24, 40, 64, 126, 3, 3, 32, 1),
    (2, 32, 32, 96, 256, 3, 3, 33, 1), (1, 16, 16, 256, 64, 3, 3, 34, 3), (1, 64, 64, 384, 256, 1, 5, 32, 1), (1, 8, 12, 32, 40, 1, 1, 31, 1),
    (1, 64, 64, 384, 256, 5, 1, 36, 1), (1, 16, 32, 128, 192, 3, 3, 35, 2),
    # two consumer groups per workgroup (K halves of every step, partial tiles summed through LDS): 128x64 and 64x64 tiles; ragged M / N; an odd number of
    # K steps (the last step has no successor in either fragment buffer); the shape the launcher picks tile 39 for by itself (one 64x64 tile per CU, K >= 1024)
    (2, 24, 40, 128, 256, 1, 5, 38, 0), (1, 16, 24, 64, 96, 3, 3, 39, 0), (2, 19, 23, 96, 136, 5, 1, 38, 0), (1, 19, 23, 96, 72, 5, 1, 39, 0),
    (2, 64, 64, 384, 128, 1, 5, 0, 0)])
def test_split3_conv_vs_fp32_kernel_and_fp64(ops, B, H, W, Cin, N, kh, kw, tile, split):
    x = dev(torch.randn(B * H * W, Cin, generator=g(2)))
    w = dev(torch.randn(N, kh * kw * Cin, generator=g(3)) / (kh * kw * Cin) ** 0.5)
    bias = dev(torch.randn(N, generator=g(4)))
    geom = (B, H, W, kh, kw, 1, 1, kh // 2, kw // 2)
    ref = conv_ref64(x, w, B, H, W, Cin, kh, kw) + bias.double()
    oe, os_ = torch.empty(B * H * W, N, device="cuda"), torch.empty(B * H * W, N, device="cuda")
    ops.conv_gemm(x, w, oe, geom=geom, bias=bias)
    ops.conv_gemm(ops.split3_pack(x), ops.split3_pack(w), os_, geom=geom, bias=bias, tile=tile, split_k=split)
    torch.cuda.synchronize()
    scale = ref.pow(2).mean().sqrt()
    e_exact, e_split = ((oe.double() - ref).pow(2).mean().sqrt() / scale).item(), ((os_.double() - ref).pow(2).mean().sqrt() / scale).item()
    check(f"split3_conv_{kh}x{kw}_{Cin}_{N}_t{tile}_rms_vs_fp64", e_split, 1.25 * e_exact)
    assert ((os_.double() - ref).abs().max() / scale).item() < 1.5 * ((oe.double() - ref).abs().max() / scale).item() + 1e-7


def test_split3_batched_plain_matrix_and_second_source(ops):
    # batched A . W^T (the aggregate's form) and the q conv's two-source A
    Bb, M, N, K = 3, 160, 96, 512
    a, w = dev(torch.rand(Bb, M, K, generator=g(5))), dev(torch.randn(Bb, N, K, generator=g(6)))
    out = torch.empty(Bb, M, N, device="cuda")
    ap, wp = ops.split3_pack(a.view(Bb * M, K)), ops.split3_pack(w.view(Bb * N, K))
    ops.conv_gemm(ap, wp, out.view(Bb * M, N)[:M], M=M, N=N, batch=Bb, bsa=M * 32, bsw=N * 32, bsc=M * N)
    torch.cuda.synchronize()
    ref = a.double() @ w.double().transpose(1, 2)
    assert ((out.double() - ref).abs().max() / ref.abs().max()).item() < 2e-6
    B, H, W, Cin, N2 = 1, 16, 16, 256, 128
    x1, x2 = dev(torch.randn(B * H * W, Cin, generator=g(7))), dev(torch.randn(B * H * W, Cin, generator=g(8)))
    w2 = dev(torch.randn(N2, 5 * Cin, generator=g(9)) / 36.0)
    o1, o2 = torch.empty(B * H * W, N2, device="cuda"), torch.empty(B * H * W, N2, device="cuda")
    geom = (B, H, W, 1, 5, 1, 1, 0, 2)
    ops.conv_gemm(x1, w2, o1, geom=geom, a2=x2, a2_channels=128)
    ops.conv_gemm(ops.split3_pack(x1), ops.split3_pack(w2), o2, geom=geom, a2=ops.split3_pack(x2), a2_channels=128)
    torch.cuda.synchronize()
    assert ((o1 - o2).abs().max() / o1.abs().max()).item() < 2e-6


@pytest.mark.parametrize("mode", ["store_fp32_kernel", "store_split3", "splitk", "gru", "zr", "axpy_batched", "pair"])
def test_epilogues_emit_the_planes_of_their_result(ops, mode):
    """st_gemm_desc.c_planes: whatever kernel and epilogue mode, the planes equal st_split3_pack(fp32 result) bit for bit."""
    B, H, W, Cin = 2, 16, 16, 128
    R = B * H * W
    geom = (B, H, W, 3, 3, 1, 1, 1, 1)
    x = dev(torch.randn(R, Cin, generator=g(10)))
    xp = ops.split3_pack(x)
    wide = ops.Planes(R, 384, "cuda")
    wide.t.zero_()

    def same(planes_slice, fp32_2d):
        torch.cuda.synchronize()
        assert torch.equal(unpack(planes_slice).cpu(), fp32_2d.double().cpu())
        want = ops.split3_pack(fp32_2d.contiguous())
        assert torch.equal(planes_slice.t[:, planes_slice.c0 // 32:(planes_slice.c0 + planes_slice.ncols) // 32].cpu(), want.t.cpu())

    if mode in ("store_fp32_kernel", "store_split3", "splitk"):
        N = 96
        w = dev(torch.randn(N, 9 * Cin, generator=g(11)) / 34.0)
        out = torch.empty(R, N, device="cuda")
        if mode == "store_fp32_kernel":
            ops.conv_gemm(x, w, out, geom=geom, act="relu", out_planes=wide.cols(128, 224))
        else:
            ops.conv_gemm(xp, ops.split3_pack(w), out, geom=geom, act="relu", out_planes=wide.cols(128, 224), tile=34,
                          split_k=3 if mode == "splitk" else 1)
        same(wide.cols(128, 224), out)
        assert bool((wide.t[:, :4] == 0).all()) and bool((wide.t[:, 7:] == 0).all())          # neighbours untouched
    elif mode == "gru":
        N = 128
        w = dev(torch.randn(N, 9 * Cin, generator=g(12)) / 34.0)
        z, h = dev(torch.rand(R, N, generator=g(13))), dev(torch.randn(R, 384, generator=g(14)))
        h0 = h.clone()
        ops.conv_gemm(xp, ops.split3_pack(w), h[:, :128], geom=geom, act="tanh", epi="gru", aux1=z, aux2=h[:, :128], out_planes=wide.cols(0, 128))
        same(wide.cols(0, 128), h[:, :128])
        ref = torch.empty(R, N, device="cuda")
        ops.conv_gemm(x, w, ref, geom=geom, act="tanh", epi="gru", aux1=z, aux2=h0[:, :128].contiguous())
        torch.cuda.synchronize()
        assert (ref - h[:, :128]).abs().max().item() < 2e-6
    elif mode == "zr":
        w = dev(torch.randn(256, 9 * Cin, generator=g(15)) / 34.0)
        hh = dev(torch.randn(R, 128, generator=g(16)))
        zb, rh, rh_ref, zb_ref = (torch.full((R, 128), 7.0, device="cuda") for _ in range(4))
        ops.conv_gemm(xp, ops.split3_pack(w), zb, geom=geom, act="sigmoid", epi="zr", aux1=hh, out2=rh, out_planes=wide.cols(0, 128), no_f32=True)
        ops.conv_gemm(x, w, zb_ref, geom=geom, act="sigmoid", epi="zr", aux1=hh, out2=rh_ref)
        torch.cuda.synchronize()
        assert bool((rh == 7.0).all())                                  # c_no_f32: the fp32 r*h is not written ...
        assert (zb - zb_ref).abs().max().item() < 2e-6                  # ... z is
        assert (unpack(wide.cols(0, 128)) - rh_ref.double()).abs().max().item() < 2e-6
    elif mode == "axpy_batched":
        Bb, M, N, K = 2, 128, 128, 256
        a, vt = dev(torch.rand(Bb, M, K, generator=g(17))), dev(torch.randn(Bb, N, K, generator=g(18)))
        mf, gamma = dev(torch.randn(Bb * M, 128, generator=g(19))), dev(torch.tensor([0.3]))
        out = torch.empty(Bb * M, N, device="cuda")
        big = ops.Planes(Bb * M, 384, "cuda")
        ops.conv_gemm(ops.split3_pack(a.view(Bb * M, K)), ops.split3_pack(vt.view(Bb * N, K)), out[:M], M=M, N=N, batch=Bb, bsa=M * 32, bsw=N * 32,
                      bsc=M * N, bsx1=M * 128, epi="axpy", aux1=mf[:M], scale_ptr=gamma, out_planes=big.cols(256, 384), plane_batch_rows=M)
        same(big.cols(256, 384), out)
        ref = mf.double().view(Bb, M, N) + 0.3 * (a.double() @ vt.double().transpose(1, 2))
        assert ((out.double().view(Bb, M, N) - ref).abs().max() / ref.abs().max()).item() < 2e-6
    else:
        w0, w1 = dev(torch.randn(192, 9 * Cin, generator=g(20)) / 34.0), dev(torch.randn(64, 9 * Cin, generator=g(21)) / 34.0)
        x1 = dev(torch.randn(R, Cin, generator=g(22)))
        out = torch.zeros(R, 256, device="cuda")
        ops.conv_gemm_pair((xp, ops.split3_pack(w0), out[:, :192], dict(geom=geom, act="relu", out_planes=wide.cols(0, 192))),
                           (ops.split3_pack(x1), ops.split3_pack(w1), out[:, 192:], dict(geom=geom, act="relu", out_planes=wide.cols(192, 256))))
        same(wide.cols(0, 256), out)
        ref = torch.empty(R, 256, device="cuda")
        ops.conv_gemm(x, w0, ref[:, :192], geom=geom, act="relu")
        ops.conv_gemm(x1, w1, ref[:, 192:], geom=geom, act="relu")
        torch.cuda.synchronize()
        assert ((ref - out).abs().max() / ref.abs().max()).item() < 2e-6


def test_flow_encode_split3_planes(ops):
    B, H, W = 2, 12, 16
    R = B * H * W
    coords = dev(torch.randn(R, 2, generator=g(23)) * 3 + 5)
    w98, b = dev(torch.randn(98, 128, generator=g(24)) * 0.1), dev(torch.randn(128, generator=g(25)) * 0.1)
    o0, o1 = torch.empty(R, 128, device="cuda"), torch.empty(R, 128, device="cuda")
    wide0, wide1 = torch.zeros(R, 384, device="cuda"), torch.zeros(R, 384, device="cuda")
    ops.flow_encode(coords, w98, b, o0, wide0[:, 254:256], B, H, W)
    pl, widep = ops.Planes(R, 128, "cuda"), ops.Planes(R, 384, "cuda")
    widep.t.zero_()
    ops.flow_encode_split3(coords, w98, b, o1, wide1[:, 254:256], B, H, W, pl, (widep, 254))
    torch.cuda.synchronize()
    assert torch.equal(o0, o1) and torch.equal(wide0, wide1)
    assert torch.equal(unpack(pl).cpu(), o1.double().cpu())
    assert torch.equal(unpack(widep).cpu(), wide1.double().cpu())


def test_split3_operators_against_their_fp32_twins(ops):
    """st_sepconv_gru_split3 / st_gma_aggregate_split3 (gru.py:44-59, gma.py:102-115) against st_sepconv_gru / st_gma_aggregate."""
    B, H, W = 2, 16, 16
    N = H * W
    R = B * N
    hx = dev(torch.randn(R, 384, generator=g(30)))
    hx[:, :128] = hx[:, :128].tanh()
    tabs = [dev(torch.randn(R, 384, generator=g(31 + i)) * 0.3) for i in range(2)]
    wzr = [dev(torch.randn(256, 5 * 384, generator=g(33 + i)) * 0.02) for i in range(2)]
    wq = [dev(torch.randn(128, 5 * 384, generator=g(35 + i)) * 0.02) for i in range(2)]
    hA, hB, zb = hx.clone(), torch.zeros(R, 384, device="cuda"), torch.empty(R, 128, device="cuda")
    ops.sepconv_gru(hA, hB, zb, tabs[0], tabs[1], wzr[0], wq[0], wzr[1], wq[1], B, H, W)
    hA3, zb3 = hx.clone(), torch.empty(R, 128, device="cuda")
    pA, pB = ops.split3_pack(hx), ops.Planes(R, 384, "cuda")
    ops.sepconv_gru_split3(hA3, pA, pB, zb3, tabs[0], tabs[1], *(ops.split3_pack(t) for t in (wzr[0], wq[0], wzr[1], wq[1])), B, H, W)
    torch.cuda.synchronize()
    check("split3_sepconv_gru_vs_fp32_operator", (hA3[:, :128] - hA[:, :128]).abs().max().item(), 3e-6)
    assert torch.equal(hA3[:, 128:], hx[:, 128:])
    assert torch.equal(unpack(pA.cols(0, 128)).cpu(), hA3[:, :128].double().cpu())              # the new state's planes
    assert torch.equal(unpack(pA.cols(128, 384)).cpu(), hx[:, 128:].double().cpu())              # x untouched
    # aggregate
    attn = torch.softmax(dev(torch.randn(B, N, N, generator=g(40))) * 2, -1)
    w_v, gamma = dev(torch.randn(128, 128, generator=g(41)) / 11.3), dev(torch.tensor([0.37]))
    wide, wide3 = hx.clone(), hx.clone()
    vT, vT3 = torch.empty(B, 128, N, device="cuda"), torch.empty(B, 128, N, device="cuda")
    ops.gma_aggregate(attn, wide[:, 128:256], w_v, gamma, vT, wide[:, 256:], B, N)
    pW = ops.split3_pack(hx)
    ops.gma_aggregate_split3(ops.split3_pack(attn.view(B * N, N)), wide3[:, 128:256], w_v, gamma, vT3, ops.Planes(B * 128, N, "cuda"), wide3[:, 256:],
                             pW.cols(256, 384), B, N)
    torch.cuda.synchronize()
    assert torch.equal(vT, vT3)
    check("split3_gma_aggregate_vs_fp32_operator", ((wide3[:, 256:] - wide[:, 256:]).abs().max() / wide[:, 256:].abs().max()).item(), 2e-6)
    assert torch.equal(unpack(pW.cols(256, 384)).cpu(), wide3[:, 256:].double().cpu())


def test_split3_rejects_what_it_cannot_run(ops):
    x, w = dev(torch.randn(64, 48)), dev(torch.randn(32, 48))
    out = torch.empty(64, 32, device="cuda")
    with pytest.raises(Exception):
        ops.split3_pack(x)                                    # C % 32 != 0
    xp, wp = ops.split3_pack(dev(torch.randn(64, 64))), ops.split3_pack(dev(torch.randn(32, 64)))
    with pytest.raises(ops.StitchErrorBase):
        ops.conv_gemm(xp, wp, out, tile=13)                   # an fp32-kernel tile
    with pytest.raises(ops.StitchErrorBase):                      # planes of a result whose rows are not whole 32-row tiles
        ops.conv_gemm(dev(torch.randn(40, 64)), dev(torch.randn(32, 64)), torch.empty(40, 32, device="cuda"), out_planes=ops.Planes(40, 32, "cuda"))


def test_split3_kernels_do_not_corrupt_their_neighbours(ops):
    """Round-6 finding (tools/neighbour_stress.py): beside waves that issue bf16 MFMAs back to back, packed-fp32 VALU instructions of OTHER
    waves returned wrong lanes 48-63 (the path's bilinear resize, run on a second stream beside a split3 GEMM, was wrong in 80 % of its
    launches).  The library is therefore built without packed-fp32 instructions (build.py); this guards the build flag."""
    g0 = g(50)
    B, H, W, Cin, N = 2, 64, 64, 256, 192
    x, w = dev(torch.randn(B * H * W, Cin, generator=g0)), dev(torch.randn(N, 9 * Cin, generator=g0) / 48)
    xp, wp = ops.split3_pack(x), ops.split3_pack(w)
    out = torch.empty(B * H * W, N, device="cuda")
    flow = dev(torch.randn(2, 2, 512, 512, generator=g0) * 5)
    ref = ops.resize_bilinear(flow, 320, 416, True, div=(512 / 416.0, 512 / 320.0)).clone()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ws = ops.new_workspace(torch.device("cuda"))
    bad = 0
    for _ in range(40):
        with torch.cuda.stream(sb), ops.workspace_scope(ws):
            for _ in range(12):
                ops.conv_gemm(xp, wp, out, geom=(B, H, W, 3, 3, 1, 1, 1, 1), act="relu")
        with torch.cuda.stream(sa):
            rs = [ops.resize_bilinear(flow, 320, 416, True, div=(512 / 416.0, 512 / 320.0)) for _ in range(12)]
        torch.cuda.synchronize()
        bad += sum(not torch.equal(r, ref) for r in rs)
    assert bad == 0, f"{bad} of 480 resize launches beside a split3 GEMM were corrupted"


@pytest.mark.parametrize("case", ["corr_both", "conv_stride2", "plain_many_tiles"])
def test_split3_persistent_walk(ops, case):
    """tile_cfg 37: a workgroup walks several output tiles with one continuous DMA ring (the all-pairs volume with its transposed second store,
    PatchEmbed's stride-2 6x6 convolution, a plain product with more tiles than workgroup slots): against the fp32 kernel and the fp64 product."""
    if case == "corr_both":
        B, N, Cc = 3, 1024, 256
        f1, f2 = dev(torch.randn(B, N, Cc, generator=g(60))), dev(torch.randn(B, N, Cc, generator=g(61)))
        v12, v21 = torch.empty(B, N, N, device="cuda"), torch.empty(B, N, N, device="cuda")
        e12, e21 = torch.empty(B, N, N, device="cuda"), torch.empty(B, N, N, device="cuda")
        ops.corr_volume_split3(ops.split3_pack(f1.view(B * N, Cc)), ops.split3_pack(f2.view(B * N, Cc)), v12, v21, B, N, Cc)
        ops.corr_volume_both(f1, f2, e12, e21)
        torch.cuda.synchronize()
        ref = f1.double() @ f2.double().transpose(1, 2)
        scale = ref.pow(2).mean().sqrt()
        es, ee = (v12.double() - ref).pow(2).mean().sqrt() / scale, (e12.double() - ref).pow(2).mean().sqrt() / scale
        check("split3_persist_corr_rms_vs_fp64", es.item(), 1.25 * ee.item())
        assert torch.equal(v21, v12.transpose(1, 2).contiguous())                 # the transposed store carries the same bits
        v1 = torch.empty(B, N, N, device="cuda")
        ops.corr_volume_split3(ops.split3_pack(f1.view(B * N, Cc)), ops.split3_pack(f2.view(B * N, Cc)), v1, None, B, N, Cc)
        torch.cuda.synchronize()
        assert torch.equal(v1, v12)
    elif case == "conv_stride2":
        M, H, W, Cin, N = 40, 16, 16, 32, 64                                       # PatchEmbed's Conv2d(32, 64, 6, 2, 2) on 40 maps: 40 * 64 rows
        x = dev(torch.randn(M * H * W, Cin, generator=g(62)))
        w = dev(torch.randn(N, 36 * Cin, generator=g(63)) / 34.0)
        bias = dev(torch.randn(N, generator=g(64)))
        geom = (M, H, W, 6, 6, 2, 2, 2, 2, 8, 8)
        oe, os_ = torch.empty(M * 64, N, device="cuda"), torch.empty(M * 64, N, device="cuda")
        ops.conv_gemm(x, w, oe, geom=geom, bias=bias)
        ops.conv_gemm(ops.split3_pack(x), ops.split3_pack(w), os_, geom=geom, bias=bias, tile=37)
        torch.cuda.synchronize()
        ref = F.conv2d(x.double().view(M, H, W, Cin).permute(0, 3, 1, 2), w.double().view(N, 6, 6, Cin).permute(0, 3, 1, 2), bias.double(), stride=2, padding=2)
        ref = ref.permute(0, 2, 3, 1).reshape(-1, N)
        scale = ref.pow(2).mean().sqrt()
        check("split3_persist_conv_s2_rms_vs_fp64", ((os_.double() - ref).pow(2).mean().sqrt() / scale).item(), 1.25 * ((oe.double() - ref).pow(2).mean().sqrt() / scale).item())
    else:
        M, N, K = 64 * 40, 64 * 30, 160                                            # 1 200 tiles on 512 slots, 5 K steps per tile, ragged walk lengths
        a, w = dev(torch.randn(M, K, generator=g(65))), dev(torch.randn(N, K, generator=g(66)))
        oe, os_ = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        ops.conv_gemm(a, w, oe)
        ops.conv_gemm(ops.split3_pack(a), ops.split3_pack(w), os_, tile=37, act="relu")
        ops.conv_gemm(a, w, oe, act="relu")
        torch.cuda.synchronize()
        ref = (a.double() @ w.double().t()).clamp_min(0)
        assert ((os_.double() - ref).abs().max() / ref.abs().max()).item() < 2e-6


@pytest.mark.parametrize("M,hidden", [(2048, 512), (4099, 512), (96, 256), (17, 32), (5, 64), (40001, 512)])      # 40001 rows: a wave walks two row blocks (256 workgroups)
def test_mlp128_split3(ops, M, hidden):
    """st_mlp128_split3 (csrc/mlp_split3.h): the C = 128 block tail -- [projection + residual ->] LayerNorm -> fc1 + GELU -> fc2 + residual(s),
    twins.py:622-623, 785-790 -- with every product as six bf16 MFMA products of planes split in registers, weights streamed from the packed
    image.  Against fp64 torch: the error is bounded by 1.25x the fp32 kernel's (st_mlp128) on the same inputs; ragged M, column slices of
    wider buffers, the no-LayerNorm form, the second residual, the projection with and without bias / residual."""
    gg = g(11)
    x, extra = torch.randn(M, 128, generator=gg) * 1.5, torch.randn(M, 128, generator=gg)
    w1, b1 = torch.randn(hidden, 128, generator=gg) / 128 ** 0.5, torch.randn(hidden, generator=gg) * 0.1
    w2, b2 = torch.randn(128, hidden, generator=gg) / hidden ** 0.5, torch.randn(128, generator=gg) * 0.1
    gam, bet = torch.rand(128, generator=gg) + 0.5, torch.randn(128, generator=gg) * 0.1
    att, x0 = torch.randn(M, 128, generator=gg), torch.randn(M, 128, generator=gg)
    wp, bp = torch.randn(128, 128, generator=gg) / 128 ** 0.5, torch.randn(128, generator=gg) * 0.1
    xd = x.double()
    w1f, b1f = ops.fold_layernorm(dev(gam), dev(bet), dev(w1), dev(b1))

    def both(ref, a, **kw):
        """(rms, max) error of the fp32 kernel and of the split3 kernel against ref, relative to ref's scale"""
        proj = kw.get("proj")
        img = ops.mlp128_split3_pack(kw["w1"], kw["b1"], dev(w2), proj=None if proj is None else (proj[0], proj[1]))
        oe, os_ = torch.empty(M, 128, device="cuda"), torch.full((M, 136), 7.0, device="cuda")
        args = dict(ln_eps=kw.get("ln_eps"), res=kw.get("res"), proj=proj)
        ops.mlp128(a, oe, kw["w1"], kw["b1"], dev(w2), dev(b2), **args)
        ops.mlp128(a, os_[:, 4:132], kw["w1"], kw["b1"], dev(w2), dev(b2), image=img, **args)
        assert (os_[:, :4] == 7.0).all() and (os_[:, 132:] == 7.0).all()
        scale = ref.abs().max().item()
        ee, es = (oe.cpu().double() - ref), (os_[:, 4:132].cpu().double() - ref)
        return (ee.pow(2).mean().sqrt().item() / scale, ee.abs().max().item() / scale), (es.pow(2).mean().sqrt().item() / scale, es.abs().max().item() / scale)

    # LayerNorm form, input a column slice of a wider buffer
    h = F.gelu(F.linear(F.layer_norm(xd, (128,), gam.double(), bet.double(), 1e-6), w1.double(), b1.double()))
    ref = F.linear(h, w2.double(), b2.double()) + xd
    xw = torch.zeros(M, 136, device="cuda")
    xw[:, 4:132] = x.cuda()
    (er, em), (sr, sm) = both(ref, xw[:, 4:132], w1=w1f, b1=b1f, ln_eps=1e-6)
    check(f"mlp128_split3_ln_rms_vs_fp64_{M}_{hidden}", sr, 1.25 * er)
    check(f"mlp128_split3_ln_max_vs_fp64_{M}_{hidden}", sm, max(2.0 * em, 3e-7))      # a maximum over few samples: 2x, floor = 2.5 fp32 ulps of the scale
    # no LayerNorm + second residual
    ref2 = F.linear(F.gelu(F.linear(xd, w1.double(), b1.double())), w2.double(), b2.double()) + xd + extra.double()
    (er, em), (sr, sm) = both(ref2, dev(x), w1=dev(w1), b1=dev(b1), res=dev(extra))
    check(f"mlp128_split3_res_rms_vs_fp64_{M}_{hidden}", sr, 1.25 * er)
    # projection + bias + residual in front, LayerNorm, second residual
    xpd = F.linear(att.double(), wp.double(), bp.double()) + x0.double()
    h = F.gelu(F.linear(F.layer_norm(xpd, (128,), gam.double(), bet.double(), 1e-6), w1.double(), b1.double()))
    ref3 = F.linear(h, w2.double(), b2.double()) + xpd + extra.double()
    (er, em), (sr, sm) = both(ref3, dev(att), w1=w1f, b1=b1f, ln_eps=1e-6, res=dev(extra), proj=(dev(wp), dev(bp), dev(x0)))
    check(f"mlp128_split3_proj_rms_vs_fp64_{M}_{hidden}", sr, 1.25 * er)
    check(f"mlp128_split3_proj_max_vs_fp64_{M}_{hidden}", sm, max(2.0 * em, 3e-7))
    # projection without bias / residual, no LayerNorm
    xq = F.linear(att.double(), wp.double())
    ref4 = F.linear(F.gelu(F.linear(xq, w1.double(), b1.double())), w2.double(), b2.double()) + xq
    (er, em), (sr, sm) = both(ref4, dev(att), w1=dev(w1), b1=dev(b1), proj=(dev(wp), None, None))
    check(f"mlp128_split3_proj_plain_rms_vs_fp64_{M}_{hidden}", sr, 1.25 * er)
    if M == 96:
        img = ops.mlp128_split3_pack(dev(w1), dev(b1), dev(w2))
        with pytest.raises(ops.StitchErrorBase):
            xc = dev(x)
            ops.mlp128(xc, xc, dev(w1), dev(b1), dev(w2), dev(b2), image=img)                                  # in place: rejected
        with pytest.raises(ops.StitchErrorBase):
            ops.mlp128(dev(att), torch.empty(M, 128, device="cuda"), dev(w1), dev(b1), dev(w2), dev(b2), image=img, proj=(dev(wp), None, None))   # image packed without the projection: too short


@pytest.mark.parametrize("M,N,ln", [(4099, 384, True), (96, 128, False), (17, 32, True), (40001, 384, True), (2048, 256, False)])
def test_rowlin128_split3(ops, M, N, ln):
    """st_rowlin128_split3: LayerNorm -> Linear(128 -> N) + bias (the q | k | v projections behind norm1, twins.py:598-600, encoder.py:156-160) with the
    product as six bf16 MFMA products of planes split in registers: against fp64 the error is bounded by 1.25x that of st_conv_gemm(a_ln) on the same
    inputs; ragged M (a wave walks several blocks at 40 001 rows), column slices of wider buffers on both sides, no bias."""
    gg = g(21)
    x = torch.randn(M, 128, generator=gg) * 1.5 + 0.3
    w, b = torch.randn(N, 128, generator=gg) / 128 ** 0.5, torch.randn(N, generator=gg) * 0.1
    gam, bet = torch.rand(128, generator=gg) + 0.5, torch.randn(128, generator=gg) * 0.1
    xd = x.double()
    if ln:
        ref = F.linear(F.layer_norm(xd, (128,), gam.double(), bet.double(), 1e-5), w.double(), b.double())
        wf, bf = ops.fold_layernorm(dev(gam), dev(bet), dev(w), dev(b))
    else:
        ref = F.linear(xd, w.double(), b.double())
        wf, bf = dev(w), dev(b)
    xw = torch.zeros(M, 136, device="cuda")
    xw[:, 4:132] = x.cuda()
    oe, os_ = torch.empty(M, N, device="cuda"), torch.full((M, N + 8), 7.0, device="cuda")
    ops.conv_gemm(xw[:, 4:132], wf, oe, bias=bf, ln_eps=1e-5 if ln else None)
    img = ops.rowlin128_split3_pack(wf, bf)
    ops.rowlin128_split3(xw[:, 4:132], os_[:, 4:4 + N], img, ln_eps=1e-5 if ln else None)
    assert (os_[:, :4] == 7.0).all() and (os_[:, 4 + N:] == 7.0).all()
    scale = ref.pow(2).mean().sqrt().item()
    ee, es = (oe.cpu().double() - ref), (os_[:, 4:4 + N].cpu().double() - ref)
    check(f"rowlin128_split3_rms_vs_fp64_{M}_{N}", es.pow(2).mean().sqrt().item() / scale, 1.25 * ee.pow(2).mean().sqrt().item() / scale)
    check(f"rowlin128_split3_max_vs_fp64_{M}_{N}", es.abs().max().item() / scale, max(2.0 * ee.abs().max().item() / scale, 3e-7))
    if M == 4099:
        # + a per-position table shared by 8 consecutive rows (the context / position part of q | k in the vertical layers): against the fp32 kernel's aux0 / row_div
        T = torch.randn((M + 7) // 8, N, generator=gg)
        o3, o3e = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        ops.rowlin128_split3(dev(x), o3, img, ln_eps=1e-5 if ln else None, aux=dev(T), row_div=8)
        ops.conv_gemm(dev(x), wf, o3e, bias=bf, aux0=dev(T), row_div=8, ln_eps=1e-5 if ln else None)
        ref3 = ref + T.double().repeat_interleave(8, 0)[:M]
        e3, e3e = (o3.cpu().double() - ref3).pow(2).mean().sqrt().item(), (o3e.cpu().double() - ref3).pow(2).mean().sqrt().item()
        check(f"rowlin128_split3_aux_rms_vs_fp64_{M}_{N}", e3 / scale, 1.25 * e3e / scale)
    if M == 96:
        o2 = torch.empty(M, N, device="cuda")
        ops.rowlin128_split3(dev(x), o2, ops.rowlin128_split3_pack(dev(w), None))                     # no bias, no LayerNorm
        assert ((o2.cpu().double() - F.linear(xd, w.double())).abs().max() / scale).item() < 3e-6      # (measured 1.2e-6: a maximum over 12 288 values against the rms scale)
        with pytest.raises(ops.StitchErrorBase):
            ops.rowlin128_split3(dev(x), torch.empty(M, N + 32, device="cuda"), img)                    # image packed for fewer output features


@pytest.mark.parametrize("R,P", [(64 * 70, 64), (40001, 64), (96, 32), (17, 5)])
def test_pe_tail_split3(ops, R, P):
    """st_pe_tail_split3: PatchEmbed's tail (encoder.py:77-95) -- ffn_with_coord.0 on [x | pe(pos)] (the position half folded into a per-position
    table), ReLU, ffn_with_coord.2, LayerNorm -- as one launch with both products as six-product bf16 contractions, the weights held in LDS.
    Against fp64 torch and against the three fp32 launches it replaces (error bounded by 1.25x theirs); ragged row counts, table periods that do
    not divide the 32-row blocks."""
    gg = g(31)
    x = torch.randn(R, 64, generator=gg)
    w1, tab = torch.randn(128, 128, generator=gg) / 128 ** 0.5, torch.randn(P, 128, generator=gg) * 0.5
    w2, b2 = torch.randn(128, 128, generator=gg) / 128 ** 0.5, torch.randn(128, generator=gg) * 0.1
    gam, bet = torch.rand(128, generator=gg) + 0.5, torch.randn(128, generator=gg) * 0.1
    rows = torch.arange(R) % P
    h = torch.relu(x.double() @ w1[:, :64].double().t() + tab.double()[rows])
    ref = F.layer_norm(h @ w2.double().t() + b2.double(), (128,), gam.double(), bet.double(), 1e-5)
    # the three fp32 launches
    s4, tok = torch.empty(R, 128, device="cuda"), torch.empty(R, 128, device="cuda")
    ops.conv_gemm(dev(x), dev(w1)[:, :64], s4, aux0=dev(tab), row_mod=P, act="relu")
    ops.conv_gemm(s4, dev(w2), tok, bias=dev(b2))
    ops.layernorm(tok, dev(gam), dev(bet), tok, 1e-5)
    out = torch.full((R + 1, 128), 7.0, device="cuda")
    img = ops.pe_tail_split3_pack(dev(w1), dev(w2))
    ops.pe_tail_split3(dev(x), dev(tab), img, dev(b2), dev(gam), dev(bet), out[:R])
    assert (out[R] == 7.0).all()
    scale = ref.pow(2).mean().sqrt().item()
    ee, es = (tok.cpu().double() - ref), (out[:R].cpu().double() - ref)
    check(f"pe_tail_split3_rms_vs_fp64_{R}_{P}", es.pow(2).mean().sqrt().item() / scale, 1.25 * ee.pow(2).mean().sqrt().item() / scale)
    check(f"pe_tail_split3_max_vs_fp64_{R}_{P}", es.abs().max().item() / scale, max(2.0 * ee.abs().max().item() / scale, 5e-7))
