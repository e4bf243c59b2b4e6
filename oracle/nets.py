"""CPU oracle: the two networks of the stitching path as pure functions over a flat weight dict
(test infrastructure, torch-CPU fp32).  Keys are the reference checkpoint keys (oracle/spec.py).

  * ``homo_offsets``  -- UDIS2 homography regression (core/UDIS2/Homography/network.py:121-199)
  * ``flowformer``    -- FlowFormer++ / PerCostFormer3 (core/FlowFormer/PerCostFormer3/*.py),
                         with timm-0.4.12 Twins-SVT-L stages 1-2 restated from its published
                         definition (structural copy in the reference: twins.py:587-680,793-936).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


class W:
    """Prefix view over a flat state dict."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def __call__(self, name):
        return self.sd[self.prefix + name]

    def sub(self, name):
        return W(self.sd, self.prefix + name)

    def has(self, name):
        return (self.prefix + name) in self.sd


def linear(w, name, x):
    return F.linear(x, w(name + ".weight"), w(name + ".bias") if w.has(name + ".bias") else None)


def conv(w, name, x, stride=1, padding=0, groups=1):
    return F.conv2d(x, w(name + ".weight"), w(name + ".bias") if w.has(name + ".bias") else None,
                    stride=stride, padding=padding, groups=groups)


def lnorm(w, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w(name + ".weight"), w(name + ".bias"), eps)


def bnorm(w, name, x):
    return F.batch_norm(x, w(name + ".running_mean"), w(name + ".running_var"), w(name + ".weight"),
                        w(name + ".bias"), False, 0.0, 1e-5)


# ------------------------------------------------------------------ homography net
def _bottleneck(w, x, stride):
    o = F.relu(bnorm(w, "bn1", conv(w, "conv1", x)))
    o = F.relu(bnorm(w, "bn2", conv(w, "conv2", o, stride=stride, padding=1)))
    o = bnorm(w, "bn3", conv(w, "conv3", o))
    idt = x
    if w.has("downsample.0.weight"):
        idt = bnorm(w, "downsample.1", conv(w, "downsample.0", x, stride=stride))
    return F.relu(o + idt)


def _res_layer(w, x, blocks, stride):
    for i in range(blocks):
        x = _bottleneck(w.sub(f"{i}."), x, stride if i == 0 else 1)
    return x


def resnet_stage1(w, x):
    """conv1, bn1, relu, maxpool, layer1, layer2 (network.py:103-118)."""
    s = w.sub("feature_extractor_stage1.")
    x = F.relu(bnorm(s, "1", conv(s, "0", x, stride=2, padding=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    x = _res_layer(s.sub("4."), x, 3, 1)
    return _res_layer(s.sub("5."), x, 4, 2)


def resnet_stage2(w, x):
    return _res_layer(w.sub("feature_extractor_stage2.0."), x, 6, 2)


def ccl(f1, f2):
    """Contextual correlation layer (network.py:147-199) -> [B,2,h,w] = (flow_w, flow_h)."""
    B, C, h, w = f1.shape
    n1 = F.normalize(f1, p=2, dim=1)
    n2 = F.normalize(f2, p=2, dim=1)
    # 3x3 patches of n2 (zero padded) as 1024 filters, [h*w, C, 3, 3]
    pat = F.unfold(n2, 3, padding=1).view(B, C, 3, 3, h * w).permute(0, 4, 1, 2, 3)
    vol = torch.cat([F.conv2d(n1[i:i + 1], pat[i], padding=1) for i in range(B)], 0)
    vol = F.softmax(vol * 10, 1)
    ch = h * w
    c_one = torch.linspace(0, ch - 1, ch).view(1, ch, 1, 1)
    h_one = torch.linspace(0, h - 1, h).view(1, 1, h, 1)
    w_one = torch.linspace(0, w - 1, w).view(1, 1, 1, w)
    flow_h = (vol * (torch.div(c_one, w, rounding_mode="floor") - h_one)).sum(1, keepdim=True)
    flow_w = (vol * (torch.remainder(c_one, w) - w_one)).sum(1, keepdim=True)
    return torch.cat([flow_w, flow_h], 1)


def regress(w, x):
    p = w.sub("regressNet1_part1.")
    for i, idx in enumerate([0, 2, 5, 7, 10, 12]):
        x = F.relu(conv(p, str(idx), x, padding=1))
        if i % 2 == 1:
            x = F.max_pool2d(x, 2, 2)
    x = x.reshape(x.shape[0], -1)
    q = w.sub("regressNet1_part2.")
    x = F.relu(linear(q, "0", x))
    x = F.relu(linear(q, "2", x))
    return linear(q, "4", x)


def homo_offsets(w, img1, img2):
    """predict_homo (core/flowHomoAdpater.py:53-61): images 0..255 -> corner offsets [B,4,2]."""
    a = img1 / 127.5 - 1.0
    b = img2 / 127.5 - 1.0
    f1 = resnet_stage2(w, resnet_stage1(w, a))
    f2 = resnet_stage2(w, resnet_stage1(w, b))
    return regress(w, ccl(f1, f2)).reshape(-1, 4, 2)


# ------------------------------------------------------------------ shared pieces
def sine_pe(x, dim, norm=1 / 200):
    """LinearPositionEmbeddingSine (attention.py:156-161): literal 3.14, [sin x|cos x|sin y|cos y]."""
    fb = torch.linspace(0, dim // 4 - 1, dim // 4)
    ax, ay = x[..., -2:-1], x[..., -1:]
    return torch.cat([torch.sin(3.14 * ax * fb * norm), torch.cos(3.14 * ax * fb * norm),
                      torch.sin(3.14 * ay * fb * norm), torch.cos(3.14 * ay * fb * norm)], -1)


def coords_grid(B, H, W):
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    return torch.stack([xs, ys], 0).to(torch.get_default_dtype())[None].repeat(B, 1, 1, 1)


def mha(q, k, v, heads, scale):
    """softmax(q k^T * scale) v per head; q [B,Nq,C], k/v [B,Nk,C] (q batch may be 1 = broadcast)."""
    B, Nk, C = k.shape
    d = C // heads
    qh = q.reshape(q.shape[0], -1, heads, d).permute(0, 2, 1, 3)
    kh = k.reshape(B, Nk, heads, d).permute(0, 2, 1, 3)
    vh = v.reshape(B, Nk, heads, v.shape[-1] // heads).permute(0, 2, 1, 3)
    att = torch.softmax(torch.matmul(qh, kh.transpose(-1, -2)) * scale, -1)
    o = torch.matmul(att, vh)
    return o.permute(0, 2, 1, 3).reshape(B, -1, v.shape[-1])


def mlp(w, name, x):
    return linear(w, name + ".fc2", F.gelu(linear(w, name + ".fc1", x)))


# ------------------------------------------------------------------ Twins-SVT-L stages 1-2
def _lsa(w, x, size, heads, ws=7):
    B, N, C = x.shape
    H, Wd = size
    x = x.view(B, H, Wd, C)
    pr, pb = (ws - Wd % ws) % ws, (ws - H % ws) % ws
    x = F.pad(x, (0, 0, 0, pr, 0, pb))
    Hp, Wp = H + pb, Wd + pr
    _h, _w = Hp // ws, Wp // ws
    x = x.reshape(B, _h, ws, _w, ws, C).transpose(2, 3).reshape(B * _h * _w, ws * ws, C)
    qkv = linear(w, "qkv", x)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    o = mha(q, k, v, heads, (C // heads) ** -0.5)
    o = o.reshape(B, _h, _w, ws, ws, C).transpose(2, 3).reshape(B, Hp, Wp, C)[:, :H, :Wd].reshape(B, N, C)
    return linear(w, "proj", o)


def _gsa(w, x, size, heads, sr):
    B, N, C = x.shape
    q = linear(w, "q", x)
    xs = x.permute(0, 2, 1).reshape(B, C, *size)
    xs = conv(w, "sr", xs, stride=sr).reshape(B, C, -1).permute(0, 2, 1)
    xs = lnorm(w, "norm", xs)
    kv = linear(w, "kv", xs)
    o = mha(q, kv[..., :C], kv[..., C:], heads, (C // heads) ** -0.5)
    return linear(w, "proj", o)


def twins_svt(w, x):
    """twins_svt_large.forward (core/FlowFormer/encoders.py:21-40): image in [-1,1] -> [B,256,H/8,W/8]."""
    B = x.shape[0]
    dims, heads, srs, patch = [128, 256], [4, 8], [8, 4], [4, 2]
    for s in range(2):
        x = conv(w, f"patch_embeds.{s}.proj", x, stride=patch[s])
        size = x.shape[2:]
        x = lnorm(w, f"patch_embeds.{s}.norm", x.flatten(2).transpose(1, 2))
        b0, b1 = w.sub(f"blocks.{s}.0."), w.sub(f"blocks.{s}.1.")
        x = x + _lsa(b0.sub("attn."), lnorm(b0, "norm1", x, 1e-6), size, heads[s])
        x = x + mlp(b0, "mlp", lnorm(b0, "norm2", x, 1e-6))
        # PEG: depthwise 3x3 + identity (twins.py:793-808)
        t = x.transpose(1, 2).reshape(B, dims[s], *size)
        t = conv(w, f"pos_block.{s}.proj.0", t, padding=1, groups=dims[s]) + t
        x = t.flatten(2).transpose(1, 2)
        x = x + _gsa(b1.sub("attn."), lnorm(b1, "norm1", x, 1e-6), size, heads[s], srs[s])
        x = x + mlp(b1, "mlp", lnorm(b1, "norm2", x, 1e-6))
        x = x.reshape(B, *size, -1).permute(0, 3, 1, 2).contiguous()
    return x


# ------------------------------------------------------------------ cost-volume encoder
def corr_volume(f1, f2):
    """MemoryEncoder.corr (encoder.py:359-369): all-pairs dot product, no scaling. -> [B,N1,N2]."""
    B, C, H, Wd = f1.shape
    a = f1.reshape(B, C, H * Wd).transpose(1, 2)
    b = f2.reshape(B, C, -1).transpose(1, 2)
    return torch.matmul(a, b.transpose(1, 2))


def patch_embed(w, cost_maps):
    """PatchEmbed.forward (encoder.py:60-95): [M,1,H2,W2] -> [M, H3*W3, 128]."""
    H2, W2 = cost_maps.shape[-2:]
    x = F.pad(cost_maps, (0, (8 - W2 % 8) % 8, 0, (8 - H2 % 8) % 8))                  # :63-66
    for i, idx in enumerate([0, 2, 4]):
        x = conv(w, f"proj.{idx}", x, stride=2, padding=2)
        if i < 2:
            x = F.relu(x)
    M, _, H3, W3 = x.shape
    pc = coords_grid(1, H3, W3) * 8 + 4
    pe = sine_pe(pc.view(1, 2, -1).permute(0, 2, 1), 64).permute(0, 2, 1).view(1, 64, H3, W3)
    x = torch.cat([x, pe.expand(M, -1, -1, -1)], 1)
    x = conv(w, "ffn_with_coord.2", F.relu(conv(w, "ffn_with_coord.0", x)))
    return lnorm(w, "norm", x.flatten(2).transpose(1, 2)), (H3, W3)


def _ffn(w, x):
    return linear(w, "ffn.3", F.gelu(linear(w, "ffn.0", x)))


def latent_cross_attn(w, latents, tokens):
    """encoder CrossAttentionLayer (crossattentionlayer.py:37-56); latents [1,8,128] broadcast."""
    q = linear(w, "q", lnorm(w, "norm1", latents))
    k, v = linear(w, "k", tokens), linear(w, "v", tokens)
    x = latents + linear(w, "proj", mha(q, k, v, 8, (q.shape[-1] / 8) ** -0.5))
    return x + _ffn(w, lnorm(w, "norm2", x))


def latent_self_attn(w, x):
    """SelfAttentionLayer (encoder.py:156-172)."""
    y = lnorm(w, "norm1", x)
    o = mha(linear(w, "q", y), linear(w, "k", y), linear(w, "v", y), 8, (x.shape[-1] / 8) ** -0.5)
    x = x + linear(w, "proj", o)
    return x + _ffn(w, lnorm(w, "norm2", x))


def _ctx_tokens(w, context, B):
    c = context.repeat(B // context.shape[0], 1, 1, 1)
    c = c.view(B, c.shape[1], -1).permute(0, 2, 1)
    return linear(w, "context_proj", c)


def vert_lsa(w, x, size, context, heads=8, ws=7):
    """LocallyGroupedAttnRPEContext (twins.py:253-304)."""
    B, N, C = x.shape
    H, Wd = size
    ctx = _ctx_tokens(w, context, B).view(B, H, Wd, -1)
    x = x.view(B, H, Wd, C)
    xqk = torch.cat([x, ctx], -1)
    Cq = xqk.shape[-1]
    pr, pb = (ws - Wd % ws) % ws, (ws - H % ws) % ws
    x = F.pad(x, (0, 0, 0, pr, 0, pb))
    xqk = F.pad(xqk, (0, 0, 0, pr, 0, pb))
    Hp, Wp = H + pb, Wd + pr
    _h, _w = Hp // ws, Wp // ws
    x = x.reshape(B, _h, ws, _w, ws, C).transpose(2, 3).reshape(-1, ws * ws, C)
    xqk = xqk.reshape(B, _h, ws, _w, ws, Cq).transpose(2, 3).reshape(-1, ws * ws, Cq)
    pe = sine_pe(coords_grid(1, ws, ws).view(1, 2, -1).permute(0, 2, 1), Cq)
    xqk = xqk + pe
    o = mha(linear(w, "q", xqk), linear(w, "k", xqk), linear(w, "v", x), heads, (C // heads) ** -0.5)
    o = o.reshape(B, _h, _w, ws, ws, C).transpose(2, 3).reshape(B, Hp, Wp, C)[:, :H, :Wd].reshape(B, N, C)
    return linear(w, "proj", o)


def vert_gsa(w, x, size, context, heads=8, sr=4):
    """GlobalSubSampleAttnRPEContext (twins.py:336-392)."""
    B, N, C = x.shape
    H, Wd = size
    ctx = _ctx_tokens(w, context, B)
    xqk = torch.cat([x, ctx], -1)
    Cq = xqk.shape[-1]
    pe_q = sine_pe(coords_grid(1, H, Wd).view(1, 2, -1).permute(0, 2, 1), Cq)
    q = linear(w, "q", xqk + pe_q)
    xs = conv(w, "sr_value", x.permute(0, 2, 1).reshape(B, C, H, Wd), stride=sr).reshape(B, C, -1).permute(0, 2, 1)
    xk = conv(w, "sr_key", xqk.permute(0, 2, 1).reshape(B, Cq, H, Wd), stride=sr).reshape(B, C, -1).permute(0, 2, 1)
    xs, xk = lnorm(w, "norm", xs), lnorm(w, "norm", xk)
    pe_k = sine_pe(coords_grid(1, H // sr, Wd // sr).view(1, 2, -1).permute(0, 2, 1) * sr, C)
    o = mha(q, linear(w, "k", xk + pe_k), linear(w, "v", xs), heads, (C // heads) ** -0.5)
    return linear(w, "proj", o)


def vert_layer(w, x, size, context):
    """VerticalSelfAttentionLayer (encoder.py:121-125) = Block(LSA) -> Block(GSA) (twins.py:787-790)."""
    lb, gb = w.sub("local_block."), w.sub("global_block.")
    x = x + vert_lsa(lb.sub("attn."), lnorm(lb, "norm1", x), size, context)
    x = x + mlp(lb, "mlp", lnorm(lb, "norm2", x))
    x = x + vert_gsa(gb.sub("attn."), lnorm(gb, "norm1", x), size, context)
    return x + mlp(gb, "mlp", lnorm(gb, "norm2", x))


def cost_encoder(w, cost_maps, B, size, context):
    """CostPerceiverEncoder.forward (encoder.py:258-287): cost_maps [B*N,1,H2,W2] -> [B*N,8,128]."""
    H1, W1 = size
    tokens, _ = patch_embed(w.sub("patch_embed."), cost_maps)
    x = latent_cross_attn(w.sub("input_layer."), w("latent_tokens"), tokens)
    short = x
    L = x.shape[1]
    for i in range(3):
        x = latent_self_attn(w.sub(f"encoder_layers.{i}."), x)
        x = x.view(B, H1 * W1, L, -1).permute(0, 2, 1, 3).reshape(B * L, H1 * W1, -1)
        x = vert_layer(w.sub(f"vertical_encoder_layers.{i}."), x, size, context)
        x = x.view(B, L, H1 * W1, -1).permute(0, 2, 1, 3).reshape(B * H1 * W1, L, -1)
    return x + short


# ------------------------------------------------------------------ decoder
def cost_lookup(cost_maps, coords, r=4):
    """encode_flow_token + bilinear_sampler (decoder.py:242-260, core/utils/utils.py:62-76)."""
    B, _, H1, W1 = coords.shape
    H2, W2 = cost_maps.shape[-2:]
    c = coords.permute(0, 2, 3, 1).reshape(B * H1 * W1, 1, 1, 2)
    d = torch.linspace(-r, r, 2 * r + 1)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), -1).view(1, 2 * r + 1, 2 * r + 1, 2)
    g = c + delta
    gx = 2 * g[..., 0:1] / (W2 - 1) - 1
    gy = 2 * g[..., 1:2] / (H2 - 1) - 1
    s = F.grid_sample(cost_maps, torch.cat([gx, gy], -1), align_corners=True)
    return s.view(B, H1, W1, -1).permute(0, 3, 1, 2)


def decoder_cross_attn(w, query, k, v, coords):
    """decoder CrossAttentionLayer (decoder.py:62-109), flow_or_pe='and'."""
    B, _, H1, W1 = coords.shape
    qc = coords.reshape(B, 2, -1).permute(0, 2, 1).reshape(B * H1 * W1, 1, 2)
    pe = sine_pe(qc, 64)
    q = linear(w, "q", lnorm(w, "norm1", query) + pe)
    x = query + linear(w, "proj", mha(q, k, v, 8, (q.shape[-1] // 8) ** -0.5))
    return x + _ffn(w, lnorm(w, "norm2", x))


def gma_attention(w, inp):
    """gma.Attention (gma.py:54-76), heads=1, dim_head=128 -> [B,N,N]."""
    B, C, H, Wd = inp.shape
    qk = conv(w, "to_qk", inp)
    q = qk[:, :128].reshape(B, 128, -1).transpose(1, 2) * (128 ** -0.5)
    k = qk[:, 128:].reshape(B, 128, -1)
    return torch.softmax(torch.matmul(q, k), -1)


def gma_aggregate(w, attn, fmap):
    """gma.Aggregate (gma.py:102-115)."""
    B, C, H, Wd = fmap.shape
    v = conv(w, "to_v", fmap).reshape(B, C, -1).transpose(1, 2)
    o = torch.matmul(attn, v).transpose(1, 2).reshape(B, C, H, Wd)
    return fmap + w("gamma") * o


def motion_encoder(w, flow, corr):
    """BasicMotionEncoder (gru.py:246-254)."""
    cor = F.relu(conv(w, "convc1", corr))
    cor = F.relu(conv(w, "convc2", cor, padding=1))
    flo = F.relu(conv(w, "convf1", flow, padding=3))
    flo = F.relu(conv(w, "convf2", flo, padding=1))
    out = F.relu(conv(w, "conv", torch.cat([cor, flo], 1), padding=1))
    return torch.cat([out, flow], 1)


def sepconv_gru(w, h, x):
    """SepConvGRU (gru.py:44-59)."""
    for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(conv(w, "convz" + sfx, hx, padding=pad))
        r = torch.sigmoid(conv(w, "convr" + sfx, hx, padding=pad))
        q = torch.tanh(conv(w, "convq" + sfx, torch.cat([r * h, x], 1), padding=pad))
        h = (1 - z) * h + z * q
    return h


def update_block(w, net, inp, corr, flow, attn):
    """GMAUpdateBlock (gru.py:322-334)."""
    mf = motion_encoder(w.sub("encoder."), flow, corr)
    mfg = gma_aggregate(w.sub("aggregator."), attn, mf)
    net = sepconv_gru(w.sub("gru."), net, torch.cat([inp, mf, mfg], 1))
    dflow = conv(w, "flow_head.conv2", F.relu(conv(w, "flow_head.conv1", net, padding=1)), padding=1)
    mask = 0.25 * conv(w, "mask.2", F.relu(conv(w, "mask.0", net, padding=1)))
    return net, mask, dflow


def convex_upsample(flow, mask):
    """upsample_flow (decoder.py:214-225)."""
    N, _, H, Wd = flow.shape
    m = torch.softmax(mask.view(N, 1, 9, 8, 8, H, Wd), 2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(N, 2, 9, 1, 1, H, Wd)
    up = (m * up).sum(2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(N, 2, 8 * H, 8 * Wd)


def decoder(w, cost_memory, context, cost_maps, iters=12, trace=None):
    """MemoryDecoder.forward, eval branch (decoder.py:262-344) -> (flow_up, flow_lowres)."""
    B, _, H1, W1 = context.shape
    coords0 = coords_grid(B, H1, W1)
    coords1 = coords_grid(B, H1, W1)
    ctx = conv(w, "proj", context)
    net, inp = torch.tanh(ctx[:, :128]), F.relu(ctx[:, 128:])
    attn = gma_attention(w.sub("att."), inp)
    ca = w.sub("decoder_layer.cross_attend.")
    k, v = linear(ca, "k", cost_memory), linear(ca, "v", cost_memory)
    flow_up = None
    for it in range(iters):
        cf = cost_lookup(cost_maps, coords1)
        qy = conv(w, "flow_token_encoder.2", F.gelu(conv(w, "flow_token_encoder.0", cf)))
        qy = qy.permute(0, 2, 3, 1).reshape(B * H1 * W1, 1, -1)
        cg = decoder_cross_attn(ca, qy, k, v, coords1).view(B, H1, W1, -1).permute(0, 3, 1, 2)
        corr = torch.cat([cg, cf], 1)
        net, mask, dflow = update_block(w.sub("update_block."), net, inp, corr, coords1 - coords0, attn)
        coords1 = coords1 + dflow
        if trace is not None:
            trace.append(dict(cost_forward=cf, cost_global=cg, net=net, dflow=dflow))
        if it == iters - 1:
            flow_up = convex_upsample(coords1 - coords0, mask)
    return flow_up, coords1 - coords0


def flowformer(w, image1, image2, iters=12, trace=None):
    """FlowFormer.forward (transformer.py:47-65); images 0..255 -> full-resolution flow [B,2,H,W]."""
    a = 2 * (image1 / 255.0) - 1.0
    b = 2 * (image2 / 255.0) - 1.0
    context = twins_svt(w.sub("context_encoder.svt."), a)
    fe = w.sub("memory_encoder.feat_encoder.svt.")
    fs, ft = twins_svt(fe, a), twins_svt(fe, b)
    B, C, H1, W1 = fs.shape
    cost_maps = corr_volume(fs, ft).reshape(B * H1 * W1, 1, H1, W1)
    mem = cost_encoder(w.sub("memory_encoder.cost_perceiver_encoder."), cost_maps, B, (H1, W1), context)
    if trace is not None:
        trace.append(dict(context=context, feat_s=fs, feat_t=ft, cost_memory=mem))
    flow_up, flow_lr = decoder(w.sub("memory_decoder."), mem, context, cost_maps, iters, trace)
    return flow_up, flow_lr
