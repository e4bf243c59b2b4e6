"""Thin host wrappers over the C-ABI: torch tensors are only device memory here -- every call
passes raw device pointers + sizes to libstitch_gfx950.so on torch's current HIP stream."""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import ChainDesc, GemmDesc, MlpDesc, check, lib
from ._lib import StitchError as StitchErrorBase

ACT = dict(none=0, relu=1, gelu=2, sigmoid=3, tanh=4)
EPI = dict(store=0, add=1, mul=2, gru=3, axpy=4, zr=5)

assert lib.st_abi_gemm_desc_size() == C.sizeof(GemmDesc), "st_gemm_desc ABI mismatch between header and binding"
assert lib.st_abi_chain_desc_size() == C.sizeof(ChainDesc), "st_chain_desc ABI mismatch between header and binding"
assert lib.st_abi_mlp_desc_size() == C.sizeof(MlpDesc), "st_mlp_desc ABI mismatch between header and binding"


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, "device tensor expected (no CPU fallback)"
    return C.c_void_p(t.data_ptr())


def _pc(t):
    """pointer to a tensor the kernel reads as dense row-major"""
    if t is not None and not t.is_contiguous():
        raise ValueError(f"contiguous tensor expected, got shape {tuple(t.shape)} strides {t.stride()}")
    return _p(t)


def _ld(t):
    """row stride of a 2-D row-major view (last dim contiguous)."""
    assert t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1), (t.shape, t.stride())
    return t.stride(0)


class _Stub:
    """shape-only stand-in for an operand that conv_gemm addresses through Planes"""

    def __init__(self, shape):
        self.shape = tuple(shape)

    def dim(self):
        return len(self.shape)

    def stride(self, i):
        return self.shape[1] if i == 0 else 1

    def data_ptr(self):
        return 0


class Planes:
    """Three blocked bf16 planes (hi, mid, lo; x == hi + mid + lo exactly) of a channels-last fp32 tensor [rows, C], C % 32 == 0: the operand
    format of the exact-split contraction (``st_gemm_desc.split3``, csrc/gemm_split3.h).  Storage ``t`` = [3, C / 32, rows, 32] bf16.
    ``cols(c0, c1)`` is a column slice (whole 32-channel chunks) sharing the storage -- concatenation stays a column offset."""

    def __init__(self, rows, C, dev, t=None, c0=0, ncols=None):
        assert C % 32 == 0, C
        self.rows, self.C = rows, C
        self.t = t if t is not None else torch.empty((3, C // 32, rows, 32), device=dev, dtype=torch.bfloat16)
        self.c0, self.ncols = c0, (C if ncols is None else ncols)

    def cols(self, c0, c1):
        assert c0 % 32 == 0 and 0 <= c0 < c1 <= self.ncols, (c0, c1, self.ncols)
        return Planes(self.rows, self.C, None, t=self.t, c0=self.c0 + c0, ncols=c1 - c0)

    @property
    def pstride(self):
        return self.t.stride(0)

    def ptr(self):
        """address of (plane 0, first chunk of this slice)"""
        return self.t.data_ptr() + (self.c0 // 32) * self.rows * 64


def split3_pack(x, out=None, chunk_rows=None):
    """fp32 rows [rows, C] (row stride >= C) -> Planes (st_split3_pack): weights at pack time, activations no split3-emitting kernel produced."""
    rows, Cc = x.shape
    if out is None:
        out = Planes(chunk_rows or rows, Cc, x.device)
    assert out.ncols == Cc and out.rows >= rows
    check(lib.st_split3_pack(_p(x), C.c_void_p(out.ptr()), rows, Cc, _ld(x), out.pstride, out.rows, _stream()), "st_split3_pack")
    return out


_WS = {}
_WS_OVERRIDE = []


def _workspace(dev):
    """split-K slab buffer (64 MiB): one per (device, HIP stream) so concurrent streams never share
    slabs; an execution context that is replayed on arbitrary streams (a captured hipGraph) brings its
    own through ``workspace_scope``."""
    if _WS_OVERRIDE:
        return _WS_OVERRIDE[-1]
    k = (dev.type, dev.index, torch.cuda.current_stream().cuda_stream)
    if k not in _WS:
        _WS[k] = new_workspace(dev)
    return _WS[k]


def new_workspace(dev):
    return torch.empty((16 * 1024 * 1024,), device=dev, dtype=torch.float32)


_WS_SIDE = {}


def side_workspace(dev):
    """the companion of the CURRENT workspace for a forked branch of the same forward (flowformer.FORK): two branches that run
    concurrently must not share split-K slabs.  Call it before switching to the side stream."""
    base = _workspace(dev)
    k = (dev.type, dev.index, base.data_ptr())
    if k not in _WS_SIDE:
        _WS_SIDE[k] = new_workspace(dev)
    return _WS_SIDE[k]


class workspace_scope:
    """``with ops.workspace_scope(ws):`` -- every split-K GEMM launched inside uses ``ws``."""

    def __init__(self, ws):
        self.ws = ws

    def __enter__(self):
        _WS_OVERRIDE.append(self.ws)
        return self.ws

    def __exit__(self, *exc):
        _WS_OVERRIDE.pop()
        return False


def conv_gemm(x, w, out, *, geom=None, bias=None, act="none", alpha=1.0, aux0=None, row_div=0, row_mod=0,
              epi="store", aux1=None, aux2=None, scale_ptr=None, batch=1, bsa=0, bsw=0, bsc=0, bsx1=0, dil=(1, 1), M=None, tile=0, split_k=0, out2=None,
              ln_eps=None, a2=None, a2_channels=0, _desc_only=False, out_planes=None, plane_row0=0, plane_batch_rows=0, no_f32=False, N=None):
    """out[M,N] = epilogue(alpha * conv(x) @ w^T + bias).

    _desc_only=True returns the filled descriptor without launching (``conv_gemm_pair``).
    a2 / a2_channels: input channels below a2_channels are read from ``a2`` (same shape and row stride as x) instead of x.

    ln_eps: the rows of x are layer-normalised (no affine: fold gamma / beta into w / bias, ``fold_layernorm``) inside
    the kernel before the product (row-streaming kernel only: plain matrix, K = 64 / 128).

    x: 2-D view [rows, Cin] of a channels-last activation; geom=(B,H,W,kh,kw,sh,sw,ph,pw) or None (1x1).
    w: [N, kh*kw*Cin] view.  out/aux*: 2-D views (column slices of wider buffers are fine).

    Exact-split operands (``st_gemm_desc.split3``): x, w (and a2) given as ``Planes`` -- x a column slice of the activation's planes, w the
    ``split3_pack`` image of the [N, K] weight matrix.  out_planes (a ``Planes`` column slice, any kernel of the family): the result also
    leaves as planes, rows plane_row0 + batch * plane_batch_rows + m; no_f32 skips the fp32 store of that output."""
    d = GemmDesc()
    split3 = isinstance(x, Planes)
    if split3:
        assert isinstance(w, Planes) and (a2 is None or isinstance(a2, Planes))
        xp, wp, a2p = x, w, a2
        Cin = xp.ncols
        d.split3 = 1
        d.a_plane_stride, d.a_rows = xp.pstride, xp.rows
        d.w_plane_stride, d.w_rows = wp.pstride, wp.rows
        # stand-ins with the shapes the fp32 path reads below (pointers are replaced afterwards)
        x = _Stub((xp.rows, Cin))
        w = _Stub((wp.rows, wp.ncols))
        a2 = None
    Cin = x.shape[1]
    if geom is None:
        rows = x.shape[0] if M is None else M
        B, H, W, kh, kw, sh, sw, ph, pw = 1, 1, rows, 1, 1, 1, 1, 0, 0
        Ho, Wo = 1, rows
    else:
        B, H, W, kh, kw, sh, sw, ph, pw = geom[:9]
        Ho = (H + 2 * ph - dil[0] * (kh - 1) - 1) // sh + 1
        Wo = (W + 2 * pw - dil[1] * (kw - 1) - 1) // sw + 1
        if len(geom) == 11:            # explicit output size (asymmetric zero padding on the right/bottom)
            Ho, Wo = geom[9], geom[10]
    d.a, d.w, d.c = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.aux0 = aux0.data_ptr() if aux0 is not None else None
    d.aux1 = aux1.data_ptr() if aux1 is not None else None
    d.aux2 = aux2.data_ptr() if aux2 is not None else None
    d.scale_ptr = scale_ptr.data_ptr() if scale_ptr is not None else None
    d.M, d.N, d.K = B * Ho * Wo, (w.shape[0] if N is None else N), kh * kw * Cin          # (N: batched W planes hold every batch's rows)
    assert w.shape[1] == d.K, (tuple(w.shape), d.K)
    d.H, d.W, d.Cin, d.ldx = H, W, Cin, _ld(x)
    d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.Ho, d.Wo = kh, kw, sh, sw, ph, pw, Ho, Wo
    d.ldw, d.ldc = _ld(w), _ld(out)
    d.ld_aux0 = _ld(aux0) if aux0 is not None else 0
    d.ld_aux1 = _ld(aux1) if aux1 is not None else 0
    d.ld_aux2 = _ld(aux2) if aux2 is not None else 0
    d.aux0_row_div, d.aux0_row_mod = row_div, row_mod
    d.act, d.epi, d.alpha = ACT[act], EPI[epi], alpha
    d.batch, d.batch_stride_a, d.batch_stride_w, d.batch_stride_c = batch, bsa, bsw, bsc
    d.batch_stride_aux1 = bsx1
    d.dh, d.dw = dil
    d.tile_cfg = tile
    d.split_k = split_k
    if ln_eps is not None:
        d.a_ln, d.a_ln_eps = 1, float(ln_eps)
    if a2 is not None:
        assert a2.shape == x.shape and _ld(a2) == _ld(x), "second A source: same geometry and row stride as x"
        d.a2, d.a2_channels = a2.data_ptr(), int(a2_channels)
    if split3:
        d.a, d.w = xp.ptr(), wp.ptr()
        d.ldx, d.ldw = Cin, d.K
        if a2p is not None:
            assert a2p.pstride == xp.pstride and a2p.rows == xp.rows, "second A source: planes of the same geometry"
            d.a2, d.a2_channels = a2p.ptr(), int(a2_channels)
    if out_planes is not None:
        d.c_planes, d.c_plane_stride, d.c_plane_rows = out_planes.t.data_ptr(), out_planes.pstride, out_planes.rows
        d.c_plane_col0, d.c_plane_row0, d.c_plane_batch_rows = out_planes.c0, int(plane_row0), int(plane_batch_rows)
        d.c_no_f32 = 1 if no_f32 else 0
    if out2 is not None:
        d.c2, d.ldc2 = out2.data_ptr(), _ld(out2)
    if _desc_only:
        return d
    if batch <= 1 and split_k != 1:
        ws = _workspace(out.device)
        d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
    check(lib.st_conv_gemm(C.byref(d), _stream()), "st_conv_gemm")
    return out


def conv_gemm_pair(call0, call1):
    """Two independent conv_gemm calls as ONE launch (st_conv_gemm_pair): each argument is (x, w, out, kwargs)."""
    d0 = conv_gemm(call0[0], call0[1], call0[2], _desc_only=True, **call0[3])
    d1 = conv_gemm(call1[0], call1[1], call1[2], _desc_only=True, **call1[3])
    check(lib.st_conv_gemm_pair(C.byref(d0), C.byref(d1), _stream()), "st_conv_gemm_pair")


def fold_layernorm(gamma, beta, w, bias=None):
    """Weights of ``Linear(LayerNorm(x))`` with the affine folded in, so that the GEMM can normalise its A rows in
    registers (``ln_eps=``): W' = W * gamma (per input column), b' = b + W @ beta.  Weight-only constant folding at
    load time, fp64 on the host."""
    dev = w.device
    w64, g64, b64 = w.detach().double().cpu(), gamma.detach().double().cpu(), beta.detach().double().cpu()
    wf = (w64 * g64[None, :]).float().contiguous().to(dev)
    bf = w64 @ b64
    if bias is not None:
        bf = bf + bias.detach().double().cpu()
    return wf, bf.float().contiguous().to(dev)


def linear_chain128(a, out, layers):
    """out = chain of up to 3 Linear(128 -> 128) layers over the rows of ``a`` without intermediate round trips (st_linear_chain128).
    layers: dicts with w [128,128], bias [128] or None, act (name), ln_eps (None = no LayerNorm of the layer's input; the affine is
    folded into w / bias: ``fold_layernorm``), res: None | a [M, >=128] tensor added to the layer's output | int l = the input of layer l."""
    d = ChainDesc()
    d.a, d.out, d.lda, d.ldo, d.M, d.nlayers = a.data_ptr(), out.data_ptr(), _ld(a), _ld(out), a.shape[0], len(layers)
    for i, y in enumerate(layers):
        ly = d.layer[i]
        w = y["w"]
        assert w.shape == (128, 128) and w.is_contiguous()
        ly.w, ly.bias = w.data_ptr(), (y["bias"].data_ptr() if y.get("bias") is not None else None)
        ly.act = ACT[y.get("act", "none")]
        ly.ln, ly.ln_eps = (1, float(y["ln_eps"])) if y.get("ln_eps") is not None else (0, 0.0)
        r = y.get("res")
        if r is None:
            ly.res = 0
        elif isinstance(r, int):
            ly.res, ly.res_layer = 2, r
        else:
            ly.res, ly.res_ptr, ly.ld_res = 1, r.data_ptr(), _ld(r)
    check(lib.st_linear_chain128(C.byref(d), _stream()), "st_linear_chain128")
    return out


def rowlin128_split3_pack(w, b):
    """weights of a Linear(128 -> N) -> the image st_rowlin128_split3 streams (uint8 tensor; pack once per weight load)"""
    N = w.shape[0]
    assert w.shape == (N, 128) and w.is_contiguous() and (b is None or b.shape == (N,))
    nb = C.c_int64(0)
    check(lib.st_rowlin128_split3_image_bytes(N, C.byref(nb)), "st_rowlin128_split3_image_bytes")
    img = torch.empty(nb.value, dtype=torch.uint8, device=w.device)
    check(lib.st_rowlin128_split3_pack(_p(w), _p(b), N, C.c_void_p(img.data_ptr()), nb.value, _stream()), "st_rowlin128_split3_pack")
    return img


def rowlin128_split3(a, out, image, ln_eps=None, aux=None, row_div=1):
    """out = LayerNorm(a) @ w^T + b [+ aux[row // row_div]] (ln_eps None: no LayerNorm) over 128-wide rows with the weights of `image` (rowlin128_split3_pack)"""
    assert a.shape[1] == 128 and out.shape[0] == a.shape[0] and out.shape[1] % 32 == 0
    if aux is not None:
        assert aux.shape[1] == out.shape[1] and aux.shape[0] * row_div >= a.shape[0]
    check(lib.st_rowlin128_split3(_p(a), _ld(a), _p(out), _ld(out), a.shape[0], out.shape[1], 1 if ln_eps is not None else 0,
                                  float(ln_eps or 0.0), C.c_void_p(image.data_ptr()), image.numel(), _p(aux), _ld(aux) if aux is not None else 0, row_div,
                                  _stream()), "st_rowlin128_split3")
    return out


def mlp128_split3_pack(w1, b1, w2, proj=None):
    """weights of one st_mlp128 -> the image st_mlp128_split3 streams (uint8 tensor; pack once per weight load).  proj = (wp, bp or None)."""
    hidden = w1.shape[0]
    assert w1.shape == (hidden, 128) and w2.shape == (128, hidden) and w1.is_contiguous() and w2.is_contiguous() and b1.shape == (hidden,)
    nb = C.c_int64(0)
    check(lib.st_mlp128_split3_image_bytes(hidden, 1 if proj is not None else 0, C.byref(nb)), "st_mlp128_split3_image_bytes")
    img = torch.empty(nb.value, dtype=torch.uint8, device=w1.device)
    wp, bp = proj if proj is not None else (None, None)
    if wp is not None:
        assert wp.shape == (128, 128) and wp.is_contiguous()
    check(lib.st_mlp128_split3_pack(_p(w1), _p(b1), _p(w2), _p(wp) if wp is not None else None, _p(bp) if bp is not None else None, hidden,
                                    C.c_void_p(img.data_ptr()), nb.value, _stream()), "st_mlp128_split3_pack")
    return img


def mlp128(a, out, w1, b1, w2, b2, ln_eps=None, res=None, proj=None, image=None):
    """out = x + (GELU(LN(x) @ w1^T + b1) @ w2^T + b2) [+ res] over 128-wide rows in one launch (st_mlp128: the hidden activations stay
    on the CU).  x = a, or with proj = (wp [128,128], bp or None, res0 or None): x = a @ wp^T + bp + res0 (the Block's attention output
    projection + residual in the same launch).  ln_eps None = no LayerNorm (otherwise without affine: fold gamma / beta into w1 / b1)."""
    hidden = w1.shape[0]
    assert w1.shape == (hidden, 128) and w2.shape == (128, hidden) and w1.is_contiguous() and w2.is_contiguous()
    assert b1.shape == (hidden,) and b2.shape == (128,) and a.shape[1] == 128 and out.shape == a.shape
    d = MlpDesc()
    d.a, d.out, d.lda, d.ldo, d.M, d.hidden = a.data_ptr(), out.data_ptr(), _ld(a), _ld(out), a.shape[0], hidden
    d.w1, d.b1, d.w2, d.b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
    d.ln, d.ln_eps = (1, float(ln_eps)) if ln_eps is not None else (0, 0.0)
    if res is not None:
        assert res.shape == a.shape
        d.res, d.ld_res = res.data_ptr(), _ld(res)
    if proj is not None:
        wp, bp, res0 = proj
        assert wp.shape == (128, 128) and wp.is_contiguous()
        d.wp = wp.data_ptr()
        if bp is not None:
            d.bp = bp.data_ptr()
        if res0 is not None:
            assert res0.shape == a.shape
            d.res0, d.ld_res0 = res0.data_ptr(), _ld(res0)
    if image is not None:                                       # the exact-split kernel on the packed image (mlp128_split3_pack)
        check(lib.st_mlp128_split3(C.byref(d), C.c_void_p(image.data_ptr()), image.numel(), _stream()), "st_mlp128_split3")
        return out
    check(lib.st_mlp128(C.byref(d), _stream()), "st_mlp128")
    return out


def corr_volume(f1, f2, out):
    B, N1, Cc = f1.shape
    N2 = f2.shape[1]
    check(lib.st_corr_volume(_p(f1), _p(f2), _p(out), B, N1, N2, Cc, _stream()), "st_corr_volume")
    return out


def corr_volume_split3(f1p, f2p, out12, out21, B, N, Cc):
    """all-pairs volume(s) from the feature planes (st_corr_volume_split3): f1p / f2p = split3_pack of the [B*N, C] feature rows; out21 may be None."""
    assert f1p.pstride == f2p.pstride and f1p.rows == f2p.rows
    check(lib.st_corr_volume_split3(C.c_void_p(f1p.ptr()), C.c_void_p(f2p.ptr()), f1p.pstride, f1p.rows, _pc(out12), _pc(out21) if out21 is not None else None,
                                    B, N, Cc, _stream()), "st_corr_volume_split3")
    return out12


def corr_volume_both(f1, f2, out12, out21):
    """out12 = f1 . f2^T and out21 = f2 . f1^T (= out12^T) from one launch."""
    B, N, Cc = f1.shape
    check(lib.st_corr_volume_both(_p(f1), _p(f2), _pc(out12), _pc(out21), B, N, Cc, _stream()), "st_corr_volume_both")
    return out12, out21


def layernorm(x, w, b, out, eps):
    check(lib.st_layernorm(_p(x), _ld(x), _p(w), _p(b), _p(out), _ld(out), x.shape[0], x.shape[1], eps, _stream()),
          "st_layernorm")
    return out


def softmax_rows(x):
    check(lib.st_softmax_rows(_p(x), _ld(x), x.shape[0], x.shape[1], _stream()), "st_softmax_rows")
    return x


def l2norm_rows(x, out):
    check(lib.st_l2norm_rows(_p(x), _p(out), x.shape[0], x.shape[1], _stream()), "st_l2norm_rows")
    return out


def maxpool(x, out, B, H, W, Cc, k, s, p):
    check(lib.st_maxpool_nhwc(_p(x), _p(out), B, H, W, Cc, k, s, p, _stream()), "st_maxpool_nhwc")
    return out


def dwconv3x3_residual(x, w9c, bias, out, B, H, W, Cc):
    check(lib.st_dwconv3x3_residual(_p(x), _p(w9c), _p(bias), _p(out), B, H, W, Cc, _stream()), "st_dwconv3x3_residual")
    return out


def sine_pe(out, dim, *, coords=None, Wg=0, ws=0, period=0, cscale=1.0, coff=0.0, accumulate=False):
    check(lib.st_sine_pe(_p(out), _ld(out), out.shape[0], dim, _p(coords), _ld(coords) if coords is not None else 0,
                         Wg, ws, period, cscale, coff, int(accumulate), _stream()), "st_sine_pe")       # accumulate: False / True / first added column
    return out


def attention_small(q, qs, k, ks, v, vs, out, os_, B, heads, Nq, Nk, D, scale):
    check(lib.st_attention_small(_p(q), qs[0], qs[1], _p(k), ks[0], ks[1], _p(v), vs[0], vs[1], _p(out), os_[0], os_[1],
                                 B, heads, Nq, Nk, D, scale, _stream()), "st_attention_small")
    return out


def latent_pool(scores, tokens, z, pixels, P):
    check(lib.st_latent_pool(_p(scores), _ld(scores), _p(tokens), _ld(tokens), _pc(z), pixels, P, _stream()), "st_latent_pool")
    return z


def attention_kvlds(q, qs, k, ks, v, vs, out, os_, B, heads, Nq, Nk, D, scale):
    check(lib.st_attention_kvlds(_p(q), qs[0], qs[1], _p(k), ks[0], ks[1], _p(v), vs[0], vs[1], _p(out), os_[0], os_[1],
                                 B, heads, Nq, Nk, D, scale, _stream()), "st_attention_kvlds")
    return out


def window_attention(q, k, v, bs, ts, qpad, kpad, vpad, out, o_bs, o_ts, B, H, W, heads, D, ws, scale):
    check(lib.st_window_attention(_p(q), _p(k), _p(v), bs, ts, _p(qpad), _p(kpad), _p(vpad), _p(out), o_bs, o_ts, B, H, W,
                                  heads, D, ws, scale, _stream()), "st_window_attention")
    return out


def ccl_softargmax(G, out, B, h, w):
    check(lib.st_ccl_softargmax(_p(G), _p(out), _ld(out), B, h, w, _stream()), "st_ccl_softargmax")
    return out


def patch_conv1(maps, w36x16, bias, out, M, H, W, Ho, Wo):
    check(lib.st_patch_conv1(_pc(maps), _pc(w36x16), _pc(bias), _p(out), M, H, W, Ho, Wo, _stream()), "st_patch_conv1")
    return out


def copy2d(src, dst):
    check(lib.st_copy2d(_p(src), _ld(src), _p(dst), _ld(dst), src.shape[0], src.shape[1], _stream()), "st_copy2d")
    return dst


def prep_image(src, dst, ldo, mul, div, sub):
    B, Cc, H, W = src.shape
    check(lib.st_prep_image(_pc(src), _p(dst), B, Cc, H, W, ldo, mul, div, sub, _stream()), "st_prep_image")
    return dst


def coords_grid(out, B, H, W):
    check(lib.st_coords_grid(_p(out), B, H, W, _stream()), "st_coords_grid")
    return out


def flow_from_coords(coords1, flow4, dst2, B, H, W):
    check(lib.st_flow_from_coords(_p(coords1), _p(flow4), _ld(flow4) if flow4 is not None else 0, _p(dst2),
                                  _ld(dst2) if dst2 is not None else 0, B, H, W, _stream()), "st_flow_from_coords")


def flow_encode(coords1, w98, bias, out, flow2, B, H, W):
    """out[:, :Co] = relu(conv7x7(coords1 - grid)); flow2[:, :2] = the flow (gru.py:251,254, decoder.py:321)."""
    check(lib.st_flow_encode(_pc(coords1), _pc(w98), _pc(bias), _p(out), _ld(out), _p(flow2), _ld(flow2) if flow2 is not None else 0,
                             B, H, W, w98.shape[1], _stream()), "st_flow_encode")
    return out


def flow_encode_split3(coords1, w98, bias, out, flow2, B, H, W, out_planes, flow_planes):
    """flow_encode that also writes the planes a split3 consumer reads: out_planes = image of out[:, :Co]; flow_planes = the column slice
    of the GRU input's planes that starts at the flow's two channels' 32-channel chunk (its offset inside the chunk from flow2's column)."""
    fp, fcol = flow_planes
    check(lib.st_flow_encode_split3(_pc(coords1), _pc(w98), _pc(bias), _p(out), _ld(out), _p(flow2), _ld(flow2), B, H, W, w98.shape[1],
                                    C.c_void_p(out_planes.ptr()), out_planes.pstride, out_planes.rows,
                                    C.c_void_p(fp.t.data_ptr()), fp.pstride, fp.rows, int(fcol), _stream()), "st_flow_encode_split3")
    return out


def cost_lookup(maps, coords, out, Nq, H2, W2, r=4):
    check(lib.st_cost_lookup(_p(maps), _p(coords), _p(out), _ld(out), Nq, H2, W2, r, _stream()), "st_cost_lookup")
    return out


def decoder_token_chain(corr, coords1, kv, weights16, rows, ntok):
    """weights16: list of 16 contiguous device tensors (see include/stitch_gfx950.h)."""
    arr = (C.c_void_p * 16)(*[w.data_ptr() for w in weights16])
    check(lib.st_decoder_token_chain(_p(corr), _ld(corr), _pc(coords1), _p(kv), arr, rows, ntok, _stream()), "st_decoder_token_chain")
    return corr


# ---- operator-level entry points (csrc/operators.hip) -------------------------------------------
def _ws(dev):
    ws = _workspace(dev)
    return ws.data_ptr(), ws.numel()


def patch_conv12(maps, c0_w, c0_b, c2_w, c2_b, out, M, H=64, W=64):
    """ReLU(c2(ReLU(c0(map)))) per 64x64 cost map in one launch (st_patch_conv12): maps [M, 4096] -> out [M*256, 32]."""
    check(lib.st_patch_conv12(_pc(maps), _pc(c0_w), _pc(c0_b), _pc(c2_w), _pc(c2_b), _p(out), M, H, W, _stream()), "st_patch_conv12")
    return out


def cost_lookup9x9(maps, coords, out, Nq, H2, W2):
    check(lib.st_cost_lookup9x9(_p(maps), _pc(coords), _p(out), _ld(out), Nq, H2, W2, _stream()), "st_cost_lookup9x9")
    return out


def patch_embed(cost_maps, weights11, ld_f0, pe_bias, s1, s2, s3, s4, tokens, M, H, W):
    """s1 = None: the caller expects the fused c0 + c2 launch (64 x 64 maps).  Whether it fuses is the LIBRARY's decision (its own reading of
    ST_FUSE_PE, operand alignment): if it declines (ST_EINVAL) the scratch is allocated here and the call repeated -- the two sides cannot
    disagree into a hard failure."""
    arr = (C.c_void_p * 11)(*[w.data_ptr() for w in weights11])

    def call(s1_):
        return lib.st_patch_embed(_pc(cost_maps), arr, ld_f0, _pc(pe_bias), _pc(s1_) if s1_ is not None else None, _pc(s2), _pc(s3), _pc(s4), _pc(tokens),
                                  M, H, W, *_ws(cost_maps.device), _stream())
    rc = call(s1)
    if rc == 1001 and s1 is None:
        Hp, Wp = (H + 7) // 8 * 8, (W + 7) // 8 * 8
        rc = call(torch.empty((M * (Hp // 2) * (Wp // 2), 16), device=cost_maps.device, dtype=torch.float32))
    check(rc, "st_patch_embed")
    return tokens


def pe_tail_split3_pack(w_f0, w_f2):
    """ffn_with_coord.0's first 64 input columns [128, ld] + ffn_with_coord.2 [128, 128] -> the LDS image of st_pe_tail_split3 (uint8 tensor; pack once per weight load)"""
    assert w_f0.shape[0] == 128 and w_f0.shape[1] >= 64 and w_f0.stride(1) == 1 and w_f2.shape == (128, 128) and w_f2.is_contiguous()
    nb = C.c_int64(0)
    check(lib.st_pe_tail_split3_image_bytes(C.byref(nb)), "st_pe_tail_split3_image_bytes")
    img = torch.empty(nb.value, dtype=torch.uint8, device=w_f0.device)
    check(lib.st_pe_tail_split3_pack(_p(w_f0), w_f0.stride(0), _p(w_f2), C.c_void_p(img.data_ptr()), nb.value, _stream()), "st_pe_tail_split3_pack")
    return img


def pe_tail_split3(x, tab, image, b2, gamma, beta, out, eps=1e-5):
    """out[R,128] = LayerNorm(ReLU(x[R,64] @ w1^T + tab[r % P]) @ w2^T + b2) (st_pe_tail_split3)"""
    assert x.shape[1] == 64 and x.is_contiguous() and out.shape == (x.shape[0], 128) and out.is_contiguous() and tab.shape[1] == 128 and tab.is_contiguous()
    check(lib.st_pe_tail_split3(_p(x), _p(tab), tab.shape[0], C.c_void_p(image.data_ptr()), image.numel(), _p(b2), _p(gamma), _p(beta), float(eps), _p(out),
                                x.shape[0], _stream()), "st_pe_tail_split3")
    return out


def patch_embed_split3(cost_maps, weights11, ld_f0, pe_bias, s2_planes, c4_w_planes, s3, s4, tokens, M, H, W, tail_image=None):
    """patch_embed for 64 x 64 maps with the third convolution on exact-split operands (st_patch_embed_split3); s2_planes = Planes(M*256, 32);
    tail_image (pe_tail_split3_pack): the three launches behind it as one (s4 may then be None)."""
    arr = (C.c_void_p * 11)(*[w.data_ptr() for w in weights11])
    check(lib.st_patch_embed_split3(_pc(cost_maps), arr, ld_f0, _pc(pe_bias), C.c_void_p(s2_planes.ptr()), s2_planes.pstride, C.c_void_p(c4_w_planes.ptr()),
                                    c4_w_planes.pstride, _pc(s3), _pc(s4) if s4 is not None else None, _pc(tokens), M, H, W,
                                    C.c_void_p(tail_image.data_ptr()) if tail_image is not None else None, tail_image.numel() if tail_image is not None else 0,
                                    *_ws(cost_maps.device), _stream()), "st_patch_embed_split3")
    return tokens


def gma_attention(inp, w_qk, qk, attn, B, N):
    check(lib.st_gma_attention(_p(inp), _ld(inp), _pc(w_qk), _pc(qk), _pc(attn), B, N, *_ws(inp.device), _stream()),
          "st_gma_attention")
    return attn


def gma_aggregate(attn, mf, w_v, gamma, vT, out, B, N, out_planes=None):
    """out = mf + gamma * attn @ (mf . w_v^T) (st_gma_aggregate); out_planes (a Planes column slice): the result also leaves as planes
    (st_gma_aggregate_planes: the same two fp32-MFMA launches, the second one's epilogue emits them)."""
    if out_planes is not None:
        check(lib.st_gma_aggregate_planes(_pc(attn), _p(mf), _ld(mf), _pc(w_v), _p(gamma), _pc(vT), _p(out), _ld(out), C.c_void_p(out_planes.t.data_ptr()),
                                          out_planes.pstride, out_planes.rows, out_planes.c0, B, N, *_ws(mf.device), _stream()), "st_gma_aggregate_planes")
        return out
    check(lib.st_gma_aggregate(_pc(attn), _p(mf), _ld(mf), _pc(w_v), _p(gamma), _pc(vT), _p(out), _ld(out), B, N,
                               *_ws(mf.device), _stream()), "st_gma_aggregate")
    return out


def gma_aggregate_split3(attn_planes, mf, w_v, gamma, vT, vT_planes, out, out_planes, B, N):
    """gma_aggregate on planes: attn_planes = split3_pack(attn [B*N, N]); vT_planes scratch Planes(B*128, N); out also -> out_planes (column slice)."""
    check(lib.st_gma_aggregate_split3(C.c_void_p(attn_planes.ptr()), attn_planes.pstride, _p(mf), _ld(mf), _pc(w_v), _p(gamma), _pc(vT),
                                      C.c_void_p(vT_planes.ptr()), vT_planes.pstride, _p(out), _ld(out), C.c_void_p(out_planes.t.data_ptr()),
                                      out_planes.pstride, out_planes.rows, out_planes.c0, B, N, *_ws(mf.device), _stream()), "st_gma_aggregate_split3")
    return out


def sepconv_gru_split3(hxA, hxA_planes, hxB_planes, zbuf, tab1, tab2, w_zr1, w_q1, w_zr2, w_q2, B, H, W):
    """sepconv_gru on planes (st_sepconv_gru_split3): w_* are split3_pack images of the fp32 operator's weights."""
    assert hxA_planes.pstride == hxB_planes.pstride and hxA_planes.rows == hxB_planes.rows and hxA_planes.C == _ld(hxA)
    assert w_zr1.pstride == w_zr2.pstride and w_q1.pstride == w_q2.pstride
    check(lib.st_sepconv_gru_split3(_p(hxA), _ld(hxA), C.c_void_p(hxA_planes.t.data_ptr()), C.c_void_p(hxB_planes.t.data_ptr()),
                                    hxA_planes.pstride, hxA_planes.rows, _pc(zbuf), _p(tab1), _p(tab2), _ld(tab1),
                                    C.c_void_p(w_zr1.ptr()), C.c_void_p(w_q1.ptr()), C.c_void_p(w_zr2.ptr()), C.c_void_p(w_q2.ptr()),
                                    w_zr1.pstride, w_q1.pstride, B, H, W, *_ws(hxA.device), _stream()), "st_sepconv_gru_split3")
    return hxA


def sepconv_gru(hxA, hxB, zbuf, tab1, tab2, w_zr1, w_q1, w_zr2, w_q2, B, H, W):
    assert _ld(hxA) == _ld(hxB) and w_zr1.shape[1] == 5 * _ld(hxA)
    check(lib.st_sepconv_gru(_p(hxA), _p(hxB), _ld(hxA), _pc(zbuf), _p(tab1), _p(tab2), _ld(tab1), _pc(w_zr1), _pc(w_q1),
                             _pc(w_zr2), _pc(w_q2), B, H, W, *_ws(hxA.device), _stream()), "st_sepconv_gru")
    return hxA


def grid_sample_blend(x, flow, mul, out):
    B, Cc, H, W = x.shape
    check(lib.st_grid_sample_blend(_pc(x), _pc(flow), _pc(mul) if mul is not None else None, _pc(out), B, Cc, H, W,
                                   _stream()), "st_grid_sample_blend")
    return out


def convex_upsample(coords1, mask, out, B, H, W):
    check(lib.st_convex_upsample(_p(coords1), _p(mask), _ld(mask), _p(out), B, H, W, _stream()), "st_convex_upsample")
    return out


# ---- geometric stage (NCHW) ------------------------------------------------------------------
def dlt4(src4x2, motion, H_out, B, mscale_x=1.0, mscale_y=1.0, div=1.0):
    check(lib.st_dlt4(_pc(src4x2), _pc(motion), _pc(H_out), B, mscale_x, mscale_y, div, _stream()), "st_dlt4")
    return H_out


def mat3_sandwich(L, X, R, out, invert=False):
    check(lib.st_mat3_sandwich(_pc(L), _pc(X), _pc(R), _pc(out), X.shape[0], int(invert), _stream()), "st_mat3_sandwich")
    return out


def homo_warp(U, theta, out_hw, n_ones=0, want_idx=False, want_out=True):
    """U [B,C,H,W] (or None with C=0) -> out [B,C+n_ones,oh,ow] (+ idx [B,oh,ow,4] int32)."""
    B = theta.shape[0]
    if U is not None:
        _, Cc, H, W = U.shape
    else:
        raise ValueError("U required (pass n_ones for the implicit mask channels)")
    oh, ow = int(out_hw[0]), int(out_hw[1])
    out = torch.empty((B, Cc + n_ones, oh, ow), device=theta.device, dtype=torch.float32) if want_out else None
    idx = torch.empty((B, oh, ow, 4), device=theta.device, dtype=torch.int32) if want_idx else None
    check(lib.st_homo_warp(_pc(U), _pc(theta), _p(out), _p(idx), B, Cc, n_ones, H, W, oh, ow, _stream()), "st_homo_warp")
    return (out, idx) if want_idx else out


def mesh_bounds(H, out4, width, height, gw=511, gh=511):
    check(lib.st_mesh_bounds(_pc(H), _p(out4), H.shape[0], float(width), float(height), gw, gh, _stream()), "st_mesh_bounds")
    return out4


def flow_warp(x, flow, mul=None):
    B, Cc, H, W = x.shape
    out = torch.empty_like(x)
    check(lib.st_grid_sample_blend(_pc(x), _pc(flow), _pc(mul), _p(out), B, Cc, H, W, _stream()), "st_grid_sample_blend")
    return out


def resize_bilinear(x, oh, ow, align_corners, div=None, out=None):
    B, Cc, H, W = x.shape
    if out is None:
        out = torch.empty((B, Cc, oh, ow), device=x.device, dtype=torch.float32)
    d0, d1, nd = (div[0], div[1], 2) if div is not None else (1.0, 1.0, 0)
    check(lib.st_resize_bilinear(_pc(x), _p(out), B * Cc, H, W, oh, ow, int(align_corners), d0, d1, nd, _stream()),
          "st_resize_bilinear")
    return out


def range_map(flow):
    B, _, H, W = flow.shape
    scratch = torch.empty((B * H * W,), device=flow.device, dtype=torch.int64)
    out = torch.empty((B, 1, H, W), device=flow.device, dtype=torch.float32)
    check(lib.st_range_map(_pc(flow), _p(scratch), _p(out), B, H, W, _stream()), "st_range_map")
    return out


def occlusion_from_range(rng, threshold):
    out = torch.empty_like(rng)
    check(lib.st_occlusion_from_range(_pc(rng), _p(out), rng.numel(), int(threshold), _stream()), "st_occlusion_from_range")
    return out


def morph_open(mask, ksz=19):
    B, Cc, H, W = mask.shape
    scratch = torch.empty((2 * B * Cc * H * W,), device=mask.device, dtype=torch.uint8)
    out = torch.empty_like(mask)
    if ksz == 19:
        check(lib.st_morph_open19(_pc(mask), _p(out), _p(scratch), B * Cc, H, W, _stream()), "st_morph_open19")
    else:
        check(lib.st_morph_open(_pc(mask), _p(out), _p(scratch), B * Cc, H, W, ksz, _stream()), "st_morph_open")
    return out


def eval_finish(final6, occ):
    B, _, H, W = final6.shape
    overlap = torch.empty((B, H, W), device=final6.device, dtype=torch.float32)
    check(lib.st_eval_finish(_pc(final6), _pc(occ), _p(overlap), B, H, W, _stream()), "st_eval_finish")
    return overlap


def blend(homo1, homo2, fin, occ):
    _, _, h, w = homo1.shape
    dev = homo1.device
    o2 = torch.empty((1, 3, h, w), device=dev)
    m1 = torch.empty((1, 3, h, w), device=dev)
    m2 = torch.empty((1, 3, h, w), device=dev)
    bl = torch.empty((1, 3, h, w), device=dev, dtype=torch.uint8)
    check(lib.st_blend(_pc(homo1), _pc(homo2), _pc(fin), _pc(occ), _p(o2), _p(m1), _p(m2), _p(bl), h, w, _stream()), "st_blend")
    return o2, m1, m2, bl


def mean_threshold(x, thr):
    B, Cc, H, W = x.shape
    out = torch.empty((B, 1, H, W), device=x.device, dtype=torch.float32)
    check(lib.st_mean_threshold(_pc(x), _p(out), B, Cc, H, W, thr, _stream()), "st_mean_threshold")
    return out


def tps_transform(U, source, target, out_hw, want_idx=False):
    """UDIS2 TPS transformer: U [B,C,H,W], source/target [B,N,2] in [-1,1] -> [B,C,oh,ow] (+T, idx)."""
    B, Cc, H, W = U.shape
    N = source.shape[1]
    oh, ow = int(out_hw[0]), int(out_hw[1])
    dev = U.device
    work = torch.empty((B * (N + 3) * (N + 5),), device=dev, dtype=torch.float64)
    T = torch.empty((B, 2, N + 3), device=dev, dtype=torch.float32)
    out = torch.empty((B, Cc, oh, ow), device=dev, dtype=torch.float32)
    idx = torch.empty((B, oh, ow, 4), device=dev, dtype=torch.int32) if want_idx else None
    check(lib.st_tps_solve_grid(_p(U), _p(source.contiguous()), _p(target.contiguous()), _p(work), _p(T), _p(out), _p(idx),
                                B, Cc, H, W, N, oh, ow, _stream()), "st_tps_solve_grid")
    return (out, T, idx) if want_idx else (out, T)


# ---- evaluation metric --------------------------------------------------------------------------
def load_rgb8(src_u8_hwc, out=None):
    """core/datasets.py:383-386 / out.py:137-143 on the GPU: uint8 [B,H,W,3] (decoder layout) -> float32 [B,3,H,W]."""
    B, H, W, c = src_u8_hwc.shape
    if c != 3 or src_u8_hwc.dtype != torch.uint8:
        raise ValueError(f"uint8 [B,H,W,3] expected, got {src_u8_hwc.dtype} {tuple(src_u8_hwc.shape)}")
    if out is None:
        out = torch.empty((B, 3, H, W), device=src_u8_hwc.device, dtype=torch.float32)
    check(lib.st_load_rgb8(_pc(src_u8_hwc), _pc(out), B, H, W, _stream()), "st_load_rgb8")
    return out


def masked_psnr_ssim(image1, final_warp_output, out=None):
    """evaluate.py:44-59 on the GPU: image1 [B,3,H,W], final_warp_output [B,6,H,W] -> fp64 [B,2] (psnr, ssim)."""
    B, _, H, W = image1.shape
    dev = image1.device
    valid = torch.empty((B, H, W), device=dev, dtype=torch.float32)
    mask = final_warp_output[:, 3:6]
    check(lib.st_channel_mean(_p(mask), final_warp_output.stride(0), _p(valid), B, 3, H, W, _stream()), "st_channel_mean")
    nblk = (3 * H * W + 255) // 256
    partial = torch.empty((2 * B * nblk,), device=dev, dtype=torch.float64)
    if out is None:
        out = torch.empty((B, 2), device=dev, dtype=torch.float64)
    assert out.dtype == torch.float64 and out.shape == (B, 2) and out.is_contiguous()
    check(lib.st_masked_psnr_ssim(_pc(image1), _p(final_warp_output), final_warp_output.stride(0), _p(valid), _p(partial), _p(out),
                                  B, H, W, _stream()), "st_masked_psnr_ssim")
    return out


# ---- composition stage (csrc/composition.hip) --------------------------------------------------
def resize_nearest_rows(x, out, B, H, W, Cc, oh, ow):
    check(lib.st_resize_nearest_rows(_p(x), _ld(x), _p(out), _ld(out), B, H, W, Cc, oh, ow, _stream()), "st_resize_nearest_rows")
    return out


def sub_rows(a, b, out):
    check(lib.st_sub_rows(_p(a), _ld(a), _p(b), _ld(b), _p(out), _ld(out), a.shape[0], a.shape[1], _stream()), "st_sub_rows")
    return out


def compose_blend(warp1, warp2, mask1, mask2, net_out, lm1, lm2, stitched):
    B, _, H, W = warp1.shape
    check(lib.st_compose_blend(_pc(warp1), _pc(warp2), _pc(mask1), _pc(mask2), _p(net_out), _ld(net_out), _pc(lm1), _pc(lm2),
                               _pc(stitched), B, H, W, _stream()), "st_compose_blend")
    return stitched


def compose_normalize(x, out):
    check(lib.st_compose_normalize(_pc(x), _pc(out), x.numel(), _stream()), "st_compose_normalize")
    return out


# ---- TPS post-pipeline (csrc/tps_pipeline.hip) ---------------------------------------------------
def flow_boxavg(flow, valid=None, k=11, negate=True):
    B, Cc, H, W = flow.shape
    out = torch.empty_like(flow)
    v = valid.float().contiguous() if valid is not None else None
    check(lib.st_flow_boxavg(_pc(flow), _p(v), _p(out), B, Cc, H, W, k, int(negate), _stream()), "st_flow_boxavg")
    return out


def sobel_magnitude(image):
    """image [1,C,H,W] -> [H,W]."""
    B, Cc, H, W = image.shape
    assert B == 1
    grad = torch.empty((H, W), device=image.device, dtype=torch.float32)
    check(lib.st_sobel_magnitude(_pc(image), _p(grad), Cc, H, W, _stream()), "st_sobel_magnitude")
    return grad


def range_argmax(grad, ranges):
    H, W = grad.shape
    out = torch.empty((ranges.shape[0],), device=grad.device, dtype=torch.int32)
    check(lib.st_range_argmax(_pc(grad), _pc(ranges), _p(out), ranges.shape[0], H, W, _stream()), "st_range_argmax")
    return out


def gather_points(planes, points_xy):
    """planes [P,H,W] fp32, points [n,2] int32 (x, y) -> [n,P]."""
    P, H, W = planes.shape
    n = points_xy.shape[0]
    out = torch.empty((n, P), device=planes.device, dtype=torch.float32)
    if n:
        check(lib.st_gather_points(_pc(planes), _pc(points_xy.contiguous()), _p(out), n, P, H, W, _stream()), "st_gather_points")
    return out


class SingularTPSError(StitchErrorBase):
    """the TPS system of the control points has no unique solution (the reference's torch.linalg.solve raises there)"""


def tps2_solve(sites, centers, values, mode=0):
    n = sites.shape[0]
    dev = sites.device
    work = torch.empty(((n + 3) * (n + 6),), device=dev, dtype=torch.float64)
    kw = torch.empty((n, 2), device=dev, dtype=torch.float32)
    aw = torch.empty((3, 2), device=dev, dtype=torch.float32)
    status = torch.zeros((1,), device=dev, dtype=torch.int32)
    check(lib.st_tps2_solve(_pc(sites), _pc(centers), _pc(values), _p(work), _p(kw), _p(aw), n, mode, _p(status), _stream()), "st_tps2_solve")
    if int(status.item()):                     # (the post-pipeline already round-trips to the host for its control points)
        raise SingularTPSError(f"singular TPS system: {n} control points with coincident or collinear sites")
    return kw, aw


def tps2_warp(img, points_a, points_b, kernel_scale=1.0, affine_scale=1.0, mode=0, align_corners=False, weights=None):
    """TPS warp of img [1,C,H,W] by the spline with f(points_a_i) = points_b_i.  mode 0 (kornia): (kw, aw) =
    get_tps_transform(points_a, points_b), kernel centres = points_b; mode 1 (pixel units): centres = points_a; mode 3: mode 1 on
    uint8-quantised data (taps truncated to 0..255 integers, result rounded and saturated: what cv2 sees in the reference)."""
    B, Cc, H, W = img.shape
    assert B == 1
    dev = img.device
    a, b = points_a.float().contiguous().to(dev), points_b.float().contiguous().to(dev)
    centers = b if (mode & 1) == 0 else a
    kw, aw = weights if weights is not None else tps2_solve(a, centers, b, mode & 1)
    out = torch.empty_like(img)
    check(lib.st_tps2_warp(_pc(img), _pc(centers), _pc(kw), _pc(aw), _p(out), Cc, H, W, b.shape[0], float(kernel_scale),
                           float(affine_scale), int(align_corners), mode, _stream()), "st_tps2_warp")
    return out


def rect_filter(x, k, is_max):
    """cv2.erode (is_max False) / cv2.dilate (True) with a k x k rectangle on [.., H, W] planes."""
    H, W = x.shape[-2:]
    planes = x.numel() // (H * W)
    tmp, out = torch.empty_like(x), torch.empty_like(x)
    check(lib.st_minmax_filter(_pc(x), _p(tmp), planes, H, W, k, int(is_max), 0, _stream()), "st_minmax_filter")
    check(lib.st_minmax_filter(_p(tmp), _p(out), planes, H, W, k, int(is_max), 1, _stream()), "st_minmax_filter")
    return out


def tps_mask_inv(warped_mask):
    B, Cc, H, W = warped_mask.shape
    assert B == 1
    inv = torch.empty((1, 1, H, W), device=warped_mask.device, dtype=torch.float32)
    check(lib.st_tps_mask_inv(_pc(warped_mask), _p(inv), Cc, H, W, _stream()), "st_tps_mask_inv")
    return inv


def tps_mix_blend(tps3, inv_clean, final_warp3, output1_3, mask1_3):
    _, _, H, W = tps3.shape
    dev = tps3.device
    tmask = torch.empty((1, 1, H, W), device=dev, dtype=torch.float32)
    mixmask = torch.empty((1, 1, H, W), device=dev, dtype=torch.float32)
    mix = torch.empty((1, 3, H, W), device=dev, dtype=torch.float32)
    blend = torch.empty((1, 3, H, W), device=dev, dtype=torch.uint8)
    check(lib.st_tps_mix_blend(_pc(tps3), _pc(inv_clean), _pc(final_warp3), _pc(output1_3), _pc(mask1_3), _p(tmask), _p(mix),
                               _p(mixmask), _p(blend), H, W, _stream()), "st_tps_mix_blend")
    return tmask, mix, mixmask, blend


# ---- mix_fn plug-ins (csrc/tps_pipeline.hip, second half) -----------------------------------------
def box_sum_cmp(plane, k, pad, out_hw, cmp):
    H, W = plane.shape[-2:]
    out = torch.empty((1, 1, out_hw[0], out_hw[1]), device=plane.device, dtype=torch.float32)
    check(lib.st_box_sum_cmp(_pc(plane), H, W, _p(out), out_hw[0], out_hw[1], k, pad, cmp, _stream()), "st_box_sum_cmp")
    return out


def dilate_thin_area_plane(mask0, dilation_kernel_size=8, thickening_kernel_size=8):
    """core/inference/utils.py:125-160 on one plane [1,1,H,W] -> (result [1,1,H,W], result >= 1 as float)."""
    H, W = mask0.shape[-2:]
    k, p = dilation_kernel_size, dilation_kernel_size // 2
    ho, wo = H + 2 * p - k + 1, W + 2 * p - k + 1
    er = box_sum_cmp(mask0, k, p, (ho, wo), 1)
    di = box_sum_cmp(er, k, p, (H, W), 2)
    thick, thin = torch.empty_like(mask0), torch.empty_like(mask0)
    check(lib.st_mix_plane_op(_pc(mask0), _pc(di), _p(thick), _p(thin), mask0.numel(), 0, 0.0, _stream()), "st_mix_plane_op")
    k2, p2 = thickening_kernel_size, thickening_kernel_size // 2
    dt = box_sum_cmp(thin, k2, p2, (H, W), 2)
    res, ge1 = torch.empty_like(mask0), torch.empty_like(mask0)
    check(lib.st_mix_plane_op(_pc(thick), _pc(dt), _p(res), _p(ge1), mask0.numel(), 1, 0.0, _stream()), "st_mix_plane_op")
    return res, ge1


def plane_threshold(a, thr):
    out = torch.empty_like(a)
    check(lib.st_mix_plane_op(_pc(a), None, _p(out), None, a.numel(), 2, float(thr), _stream()), "st_mix_plane_op")
    return out


def mix_stage_a(final_warp, occ, mask1, tps, tmask, method):
    _, _, H, W = final_warp.shape
    tfw, tfwm = torch.empty_like(final_warp), torch.empty_like(final_warp)
    iam0 = torch.empty((1, 1, H, W), device=final_warp.device, dtype=torch.float32)
    check(lib.st_mix_stage_a(_pc(final_warp), _pc(occ), _pc(mask1), _pc(tps), _pc(tmask), _p(tfw), _p(tfwm), _p(iam0), H, W, method,
                             _stream()), "st_mix_stage_a")
    return tfw, tfwm, iam0


def mix_stage_b(iam, dil, mask1, tfw, output1):
    _, _, H, W = tfw.shape
    only1 = torch.empty_like(tfw)
    other0 = torch.empty((1, 1, H, W), device=tfw.device, dtype=torch.float32)
    check(lib.st_mix_stage_b(_pc(iam), _pc(dil), _pc(mask1), _pc(tfw), _pc(output1), _p(only1), _p(other0), H, W, _stream()),
          "st_mix_stage_b")
    return only1, other0


def mix_mul_mask(img3, mask=None, invert=False, clip=False):
    _, _, H, W = img3.shape
    out = torch.empty_like(img3)
    check(lib.st_mix_mul_mask(_pc(img3), _pc(mask) if mask is not None else None, _p(out), H, W, int(invert), int(clip), _stream()),
          "st_mix_mul_mask")
    return out


def blend_pair(output1, mask1, output2, mask2):
    _, _, H, W = output1.shape
    blend = torch.empty((1, 3, H, W), device=output1.device, dtype=torch.uint8)
    check(lib.st_blend_pair(_pc(output1), _pc(mask1), _pc(output2), _pc(mask2), mask2.shape[1], _p(blend), H, W, _stream()), "st_blend_pair")
    return blend
