"""FlowFormer++ (PerCostFormer3) flow estimator on the MI355X kernels -- drop-in for
``build_flowformer(cfg)`` / ``FlowFormer.forward`` (reference: core/FlowFormer/__init__.py:2-9,
core/FlowFormer/PerCostFormer3/transformer.py:47-65).

Design notes (MI355X-first, not a translation):
  * activations are channels-last rows ``[B*H*W, C]``; every Linear / conv is one fp32-MFMA implicit
    GEMM (st_conv_gemm) with bias / activation / residual / GRU gating fused in the epilogue, and
    channel concatenations are column slices of one wide buffer (no cat / permute passes);
  * the 8 latent tokens stay in ``[pixel, latent, C]`` order for the whole cost encoder; the
    "vertical" Twins layers address that layout with strides instead of transposing it
    (encoder.py:277-279 permutes twice per layer);
  * context / positional terms that the reference concatenates to every token and re-projects per
    latent (twins.py:261-264,285-288,340-377) are linear, so they are folded into per-position bias
    tables computed once per layer (4096 rows instead of 8*4096) and added in the GEMM epilogue;
  * only the last of the 12 convex upsamplings is used in eval (decoder.py:341-344): the mask head
    runs on the last iteration only.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import ops
from .checkpoint import HP, ParamTree, flat_params, flow_spec
from .homography import pack_conv


# LayerNorm in front of a K = 128 Linear runs inside the row-streaming GEMM (ops.conv_gemm(ln_eps=...)): gamma / beta are
# folded into the weights at pack time.  ST_FUSE_LN=0 keeps the separate LayerNorm kernel (A/B measurements).
FUSE_LN = os.environ.get("ST_FUSE_LN", "1") != "0"
PAIR_CONVS = os.environ.get("ST_PAIR_CONVS", "1") != "0"       # convc2 + convf2 of the motion encoder as one launch
FUSE_MLP = os.environ.get("ST_FUSE_MLP", "1") != "0"            # the C = 128 Twins MLPs (LN -> fc1 + GELU -> fc2 + residual) as one st_mlp128 launch
FUSE_PROJ = os.environ.get("ST_FUSE_PROJ", "1") != "0"          # ... with the Block's attention output projection + residual in front, same launch
FUSE_CHAIN = os.environ.get("ST_FUSE_CHAIN", "1") != "0"        # the latent layers' 128-wide tails as one st_linear_chain128 launch
# ST_FORK=1: the context branch of a pass (cnet Twins, then everything of the decoder that only needs the context: proj_net / proj_inp, the
# SepConvGRU tables, the GMA attention matrix) on a second HIP stream beside the feature branch (fnet Twins, correlation volume, PatchEmbed,
# latent layers); joined by an event before the first vertical layer (context) and by a stream join before the refinement loop.  Captured
# into the forward's hipGraph as a parallel branch.  Same kernels, same operands: bit-identical.  Measured in round 5 (section 5 of DESIGN.md).
FORK = os.environ.get("ST_FORK", "0") == "1"
FORK_ENC = os.environ.get("ST_FORK_ENC", "0") == "1"           # experiment: flow_encode of an iteration on a side stream beside cost lookup + token chain (which fill half the CUs)
# The decoder's long-K contractions (SepConvGRU, motion-encoder 3x3 convs, GMA aggregate, flow / mask head conv1: ~5.3 of the 12 ms of a pair)
# on the bf16 matrix cores with EXACTLY split operands (csrc/gemm_split3.h: x = hi + mid + lo in three bf16, six products, fp32 accumulate --
# error against fp64 0.83x the fp32 MFMA chain's, 1.5-1.76x its speed, profiles/r6_split3_probe.json).  Every operand travels as blocked
# bf16 planes written by the epilogue of the kernel that produced it; ST_SPLIT3=0 = the fp32-MFMA kernels of rounds 1-5 (A/B).
SPLIT3 = os.environ.get("ST_SPLIT3", "1") != "0"
FUSE_PE = os.environ.get("ST_FUSE_PE", "1") != "0"              # PatchEmbed c0 + c2 per cost map in one launch (csrc/patchembed.hip; the library reads the same switch)
assert not (SPLIT3 and FORK_ENC), "ST_FORK_ENC is an fp32-path experiment"
S3_PAIR = os.environ.get("ST_S3_PAIR", "1") != "0"
S3_PE = os.environ.get("ST_S3_PE", "1") != "0"                 # PatchEmbed's third convolution on planes
S3_MLP = os.environ.get("ST_S3_MLP", "1") != "0"               # the C = 128 block tails (st_mlp128) on the split3 kernel (st_mlp128_split3, weights packed into its image once)
S3_PE_TAIL = os.environ.get("ST_S3_PE_TAIL", "1") != "0"       # PatchEmbed's ffn_with_coord + LayerNorm (three HBM-bound launches over M P rows) as one split3 launch
S3_LIN = os.environ.get("ST_S3_LIN", "1") != "0"               # LayerNorm -> q | k | v projection (K = 128, N = 384) on the split3 row kernel (st_rowlin128_split3)
S3_CHAIN = os.environ.get("ST_S3_CHAIN", "1") != "0"           # the latent layers' 128-wide tails (st_linear_chain128) on the same split3 kernel, hidden = 128
S3_AGG = os.environ.get("ST_S3_AGG", "0") == "1"               # GMA aggregate on planes (measured equal to the fp32 kernel in the chain: HBM-bound; default off)
S3_OFF = int(os.environ.get("ST_S3_OFF", "0"))     # bisecting aid: bit 1 mask-head conv, 2 flow-head conv, 8 GRU, 16 motion conv, 32 conv pair back on the fp32 kernels
_SIDE = {}


def _side_stream(cur):
    key = (cur.device_index, cur.cuda_stream)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=cur.device)
    return _SIDE[key]


def _new(rows, cols, dev, zero=False):
    return (torch.zeros if zero else torch.empty)((rows, cols), device=dev, dtype=torch.float32)


class FlowFormer(ParamTree):
    def __init__(self, cfg=None):
        super().__init__(flow_spec())
        self.cfg = cfg
        self._pk = None
        self._const = {}
        self._gen = 0                    # bumped whenever the packed weights are dropped: captured hipGraphs compare it before replay
        self.register_load_state_dict_post_hook(lambda m, k: m._invalidate())

    def _invalidate(self):
        self._pk = None
        self._const = {}
        self._gen += 1

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    # ================================================================== weight prepack
    def pack(self):
        p = flat_params(self)
        pk = {}

        def lin(name):
            return p[name + ".weight"].contiguous(), p[name + ".bias"].contiguous()

        def cat_lin(names):
            return (torch.cat([p[n + ".weight"] for n in names], 0).contiguous(),
                    torch.cat([p[n + ".bias"] for n in names], 0).contiguous())

        def conv(name, cin_pad=None):
            b = p.get(name + ".bias")
            return pack_conv(p[name + ".weight"], cin_pad), (b.contiguous() if b is not None else None)

        def mlp_image(fc1_ln, fc2, proj):
            """the split3 image of a block tail (projection + LayerNorm-folded fc1 + fc2), or None: CPU weights (tests of the pack layouts), switch off"""
            if not (SPLIT3 and S3_MLP and FUSE_LN and FUSE_MLP and FUSE_PROJ and fc1_ln[0].is_cuda and fc1_ln[0].shape[1] == 128
                    and (proj is None or proj[0].shape == (128, 128))):
                return None
            return ops.mlp128_split3_pack(fc1_ln[0], fc1_ln[1], fc2[0], proj=None if proj is None else (proj[0], proj[1]))

        def lin_image(w_b):
            """the split3 image of a LayerNorm-folded Linear(128 -> N), or None (CPU weights, switch off)"""
            w, b = w_b if isinstance(w_b, (tuple, list)) else (w_b, None)
            if not (SPLIT3 and S3_LIN and FUSE_LN and w.is_cuda and w.shape[1] == 128 and w.shape[0] % 32 == 0 and w.is_contiguous()):
                return None
            return ops.rowlin128_split3_pack(w, b)

        def twins(prefix):
            t = {}
            t["pe0"] = conv(prefix + "patch_embeds.0.proj", 4)
            t["pe1"] = conv(prefix + "patch_embeds.1.proj")
            for s in range(2):
                t[f"pen{s}"] = lin(prefix + f"patch_embeds.{s}.norm")
                b0, b1 = prefix + f"blocks.{s}.0.", prefix + f"blocks.{s}.1."
                C = p[b0 + "norm1.weight"].shape[0]
                qkv_w, qkv_b = lin(b0 + "attn.qkv")
                t[f"l{s}"] = dict(n1=lin(b0 + "norm1"), qkv=(qkv_w, qkv_b), proj=lin(b0 + "attn.proj"), n2=lin(b0 + "norm2"),
                                  fc1=lin(b0 + "mlp.fc1"), fc2=lin(b0 + "mlp.fc2"),
                                  # a zero-padded token's q/k/v is the bias (twins.py:606-611)
                                  pads=tuple(qkv_b[i * C:(i + 1) * C].expand(49, C).contiguous() for i in range(3)))
                if C == 128:
                    t[f"l{s}"]["qkv_ln"] = ops.fold_layernorm(*t[f"l{s}"]["n1"], qkv_w, qkv_b)
                    t[f"l{s}"]["qkv_s3"] = lin_image(t[f"l{s}"]["qkv_ln"])
                    t[f"l{s}"]["fc1_ln"] = ops.fold_layernorm(*t[f"l{s}"]["n2"], *t[f"l{s}"]["fc1"])
                    t[f"l{s}"]["mlp_s3"] = mlp_image(t[f"l{s}"]["fc1_ln"], t[f"l{s}"]["fc2"], t[f"l{s}"]["proj"])
                w9 = p[prefix + f"pos_block.{s}.proj.0.weight"].reshape(C, 9).t().contiguous()
                t[f"peg{s}"] = (w9, p[prefix + f"pos_block.{s}.proj.0.bias"].contiguous())
                t[f"g{s}"] = dict(n1=lin(b1 + "norm1"), q=lin(b1 + "attn.q"), kv=lin(b1 + "attn.kv"), sr=conv(b1 + "attn.sr"),
                                  srn=lin(b1 + "attn.norm"), proj=lin(b1 + "attn.proj"), n2=lin(b1 + "norm2"),
                                  fc1=lin(b1 + "mlp.fc1"), fc2=lin(b1 + "mlp.fc2"))
                if C == 128:
                    t[f"g{s}"]["fc1_ln"] = ops.fold_layernorm(*t[f"g{s}"]["n2"], *t[f"g{s}"]["fc1"])
                    t[f"g{s}"]["mlp_s3"] = mlp_image(t[f"g{s}"]["fc1_ln"], t[f"g{s}"]["fc2"], t[f"g{s}"]["proj"])
            return t

        pk["fnet"] = twins("memory_encoder.feat_encoder.svt.")
        pk["cnet"] = twins("context_encoder.svt.")
        c = "memory_encoder.cost_perceiver_encoder."
        pe = dict(c0=conv(c + "patch_embed.proj.0"), c2=conv(c + "patch_embed.proj.2"), c4=conv(c + "patch_embed.proj.4"),
                  f0=conv(c + "patch_embed.ffn_with_coord.0"), f2=conv(c + "patch_embed.ffn_with_coord.2"),
                  norm=lin(c + "patch_embed.norm"))
        pe["c0_direct"] = (p[c + "patch_embed.proj.0.weight"].reshape(16, 36).t().contiguous(), pe["c0"][1])
        pe["embed11"] = [pe["c0_direct"][0], pe["c0_direct"][1], pe["c2"][0], pe["c2"][1], pe["c4"][0], pe["c4"][1],
                         pe["f0"][0], pe["f2"][0], pe["f2"][1], pe["norm"][0], pe["norm"][1]]
        if SPLIT3 and pe["c4"][0].is_cuda:               # (a pack on the CPU -- layout tests -- has no planes: the module cannot run there anyway)
            pe["c4_s3"] = ops.split3_pack(pe["c4"][0])          # [64, 36 taps * 32]: the split3 form of PatchEmbed's third convolution
            if S3_PE_TAIL:
                pe["tail_s3"] = ops.pe_tail_split3_pack(pe["f0"][0], pe["f2"][0])       # ffn_with_coord (64 -> 128 -> 128) + LayerNorm as one launch
        pk["pe"] = pe
        pk["latents"] = p[c + "latent_tokens"][0].contiguous()

        def attn_layer(name, fuse_qkv):
            d = dict(n1=lin(name + ".norm1"), n2=lin(name + ".norm2"), proj=lin(name + ".proj"), f0=lin(name + ".ffn.0"),
                     f3=lin(name + ".ffn.3"), q=lin(name + ".q"))
            if fuse_qkv:
                d["qkv"] = cat_lin([name + ".q", name + ".k", name + ".v"])
                d["qkv_ln"] = ops.fold_layernorm(*d["n1"], *d["qkv"])
                d["qkv_s3"] = lin_image(d["qkv_ln"])
            d["kv"] = cat_lin([name + ".k", name + ".v"])
            d["f0_ln"] = ops.fold_layernorm(*d["n2"], *d["f0"])
            if S3_CHAIN:
                # the latent layers' tails are the same operator as the Block tails with hidden = 128 (proj + residual -> LN -> ffn.0 + GELU -> ffn.3 + residual)
                d["mlp_s3"] = mlp_image(d["f0_ln"], d["f3"], d["proj"])
                d["mlp_s3_plain"] = mlp_image(d["f0_ln"], d["f3"], None)
            return d
        pk["xin"] = attn_layer(c + "input_layer", False)
        # first layer: the queries are the (normalised, projected) latent tokens themselves -- constants of the weights.
        # Fold them through the key projection (scores against un-projected tokens; the key bias cancels in the softmax)
        # and fold the value bias through the output projection (softmax weights sum to one).
        with torch.no_grad():
            # weight-only constant folding at load time, in fp64 on the HOST (a few 8x128 / 128x128 products: no GPU library call)
            L = pk["xin"]
            dev = L["proj"][0].device
            c64 = lambda t: t.detach().double().cpu()                                      # noqa: E731
            lat = c64(p[c + "latent_tokens"][0])
            qn = torch.nn.functional.layer_norm(lat, (lat.shape[1],), c64(L["n1"][0]), c64(L["n1"][1]), 1e-5)
            qv = qn @ c64(L["q"][0]).t() + c64(L["q"][1])                                  # [nl, 128]
            Wk, Wv = c64(L["kv"][0][:128]), c64(L["kv"][0][128:])
            bv = c64(L["kv"][1][128:])
            nl_, hd = qv.shape[0], 16
            qp = torch.einsum("lhe,hec->lhc", qv.view(nl_, 8, hd), Wk.view(8, hd, 128)) * hd ** -0.5
            L["qfold"] = qp.reshape(nl_ * 8, 128).float().contiguous().to(dev)             # row = latent*8 + head
            L["wv_heads"] = Wv.float().contiguous().to(dev)                                # [8 heads x 16, 128]
            L["proj_b_fold"] = (c64(L["proj"][1]) + c64(L["proj"][0]) @ bv).float().contiguous().to(dev)
        pk["self"] = [attn_layer(c + f"encoder_layers.{i}", True) for i in range(HP["encoder_depth"])]
        vert = []
        for i in range(HP["encoder_depth"]):
            v = c + f"vertical_encoder_layers.{i}."
            lb, gb = v + "local_block.", v + "global_block."
            sk_w = p[gb + "attn.sr_key.weight"]           # [128, 192, 4, 4]: channels = [x(128) | ctx(64)]
            vert.append(dict(
                ln1=lin(lb + "norm1"), lctx=lin(lb + "attn.context_proj"), lq=lin(lb + "attn.q"), lk=lin(lb + "attn.k"),
                lv=lin(lb + "attn.v"), lproj=lin(lb + "attn.proj"), ln2=lin(lb + "norm2"), lfc1=lin(lb + "mlp.fc1"),
                lfc2=lin(lb + "mlp.fc2"),
                gn1=lin(gb + "norm1"), gctx=lin(gb + "attn.context_proj"), gq=lin(gb + "attn.q"), gk=lin(gb + "attn.k"),
                gv=lin(gb + "attn.v"), gproj=lin(gb + "attn.proj"), gn2=lin(gb + "norm2"), gfc1=lin(gb + "mlp.fc1"),
                gfc2=lin(gb + "mlp.fc2"), gsrn=lin(gb + "attn.norm"),
                gskx=pack_conv(sk_w[:, :128].contiguous()), gskc=pack_conv(sk_w[:, 128:].contiguous()),
                gskb=p[gb + "attn.sr_key.bias"].contiguous(), gsv=conv(gb + "attn.sr_value")))
            vert[-1]["lqkv"] = torch.cat([vert[-1]["lq"][0][:, :128], vert[-1]["lk"][0][:, :128], vert[-1]["lv"][0]], 0).contiguous()
            vert[-1]["gskv"] = torch.cat([vert[-1]["gskx"], vert[-1]["gsv"][0]], 0).contiguous()   # sr_key (x part) | sr_value
            Vd = vert[-1]
            # per-pixel pre-activation tables in ONE product each: [q | k | v-bias] of the local block (the v columns have zero
            # weights: they carry v's bias), [sr_key over the context channels | sr_value's bias] of the global block
            z128 = torch.zeros_like(Vd["lq"][0])
            Vd["ltab"] = (torch.cat([Vd["lq"][0], Vd["lk"][0], z128], 0).contiguous(), torch.cat([Vd["lq"][1], Vd["lk"][1], Vd["lv"][1]]).contiguous())
            Vd["gtab"] = (torch.cat([Vd["gskc"], torch.zeros_like(Vd["gskc"])], 0).contiguous(), torch.cat([Vd["gskb"], Vd["gsv"][1]]).contiguous())
            vert[-1]["lqkv_ln"] = ops.fold_layernorm(*vert[-1]["ln1"], vert[-1]["lqkv"])
            vert[-1]["lqkv_s3"] = lin_image(vert[-1]["lqkv_ln"])
            vert[-1]["lfc1_ln"] = ops.fold_layernorm(*vert[-1]["ln2"], *vert[-1]["lfc1"])
            vert[-1]["gfc1_ln"] = ops.fold_layernorm(*vert[-1]["gn2"], *vert[-1]["gfc1"])
            vert[-1]["lmlp_s3"] = mlp_image(vert[-1]["lfc1_ln"], vert[-1]["lfc2"], vert[-1]["lproj"])
            vert[-1]["gmlp_s3"] = mlp_image(vert[-1]["gfc1_ln"], vert[-1]["gfc2"], vert[-1]["gproj"])
        pk["vert"] = vert
        m = "memory_decoder."
        Q = HP["query_latent_dim"]
        # local cost buffer layout: [cost_forward 81 | 3 zero | cost_global 64] (reference order is
        # cat([cost_global, cost_forward]) decoder.py:319 -> permute convc1's input columns)
        f0w = p[m + "flow_token_encoder.0.weight"].reshape(Q, 81)
        f0 = torch.zeros((Q, 84), device=f0w.device)
        f0[:, :81] = f0w
        c1w = p[m + "update_block.encoder.convc1.weight"].reshape(256, 145)
        c1 = torch.zeros((256, 160), device=c1w.device)         # K padded to whole 32-channel steps (LDS-DMA kernel)
        c1[:, :81] = c1w[:, 64:]
        c1[:, 84:148] = c1w[:, :64]
        ub = m + "update_block."
        pw, pb = p[m + "proj.weight"].reshape(256, 256).contiguous(), p[m + "proj.bias"].contiguous()
        ca = m + "decoder_layer.cross_attend"
        dec = dict(
            fte0=(f0.contiguous(), p[m + "flow_token_encoder.0.bias"].contiguous()), fte2=conv(m + "flow_token_encoder.2"),
            proj_net=(pw[:128].contiguous(), pb[:128].contiguous()), proj_inp=(pw[128:].contiguous(), pb[128:].contiguous()),
            ca=attn_layer(ca, False), qk=pack_conv(p[m + "att.to_qk.weight"]),
            convc1=(c1.contiguous(), p[ub + "encoder.convc1.bias"].contiguous()), convc2=conv(ub + "encoder.convc2"),
            convf1=(p[ub + "encoder.convf1.weight"].permute(2, 3, 1, 0).reshape(98, 128).contiguous(),      # [tap][c][co]
                    p[ub + "encoder.convf1.bias"].contiguous()), convf2=conv(ub + "encoder.convf2"), conv=conv(ub + "encoder.conv"),
            to_v=pack_conv(p[ub + "aggregator.to_v.weight"]), gamma=p[ub + "aggregator.gamma"].contiguous(),
            fh1=conv(ub + "flow_head.conv1"), fh2=conv(ub + "flow_head.conv2"), m0=conv(ub + "mask.0"))
        cad = dec["ca"]
        dec["chain16"] = [dec["fte0"][0], dec["fte0"][1], dec["fte2"][0], dec["fte2"][1], cad["n1"][0], cad["n1"][1],
                          cad["q"][0], cad["q"][1], cad["proj"][0], cad["proj"][1], cad["n2"][0], cad["n2"][1],
                          cad["f0"][0], cad["f0"][1], cad["f3"][0], cad["f3"][1]]
        m2w, m2b = conv(ub + "mask.2")
        dec["m2"] = (m2w, (0.25 * m2b).contiguous())          # mask = .25 * conv (gru.py:333): alpha scales acc, bias pre-scaled
        # SepConvGRU (gru.py:32-59).  Input channels are [h | inp | motion | motion_global]; `inp` is constant
        # over the 12 iterations, so its contribution (+ bias) becomes a per-pass table and the recurrent GEMMs
        # contract over [h | motion | motion_global] only (K 2560 -> 1920).  z and r share their input: one GEMM.
        for sfx in ("1", "2"):
            parts = {}
            for gate in ("z", "r", "q"):
                wg = p[ub + f"gru.conv{gate}{sfx}.weight"]
                parts[gate] = (pack_conv(torch.cat([wg[:, :128], wg[:, 256:]], 1).contiguous()),
                               pack_conv(wg[:, 128:256].contiguous()), p[ub + f"gru.conv{gate}{sfx}.bias"])
            dec["zr" + sfx] = torch.cat([parts["z"][0], parts["r"][0]], 0).contiguous()
            dec["q" + sfx] = parts["q"][0]
            dec["inp" + sfx] = (torch.cat([parts[g][1] for g in ("z", "r", "q")], 0).contiguous(),
                                torch.cat([parts[g][2] for g in ("z", "r", "q")], 0).contiguous())
        if SPLIT3 and dec["zr1"].is_cuda:
            # exact three-way bf16 split of the weights of the split3 launches, once (K ordered (tap, channel) as in the fp32 matrices)
            dec["s3"] = {k: ops.split3_pack(dec[k] if torch.is_tensor(dec[k]) else dec[k][0])
                         for k in ("zr1", "q1", "zr2", "q2", "convc2", "convf2", "conv", "fh1", "m0")}
        pk["dec"] = dec
        self._pk = pk
        return pk

    # ================================================================== shared blocks
    @staticmethod
    def _mlp(x, n2, fc1, fc2, eps, out=None, fc1_ln=None, extra_res=None, proj=None, image=None):
        """x + fc2(GELU(fc1(LN(x)))) [+ extra_res] (timm Mlp inside Block, twins.py:785-790).  proj = (att, (w, b), res): x is the Block's
        attention branch x = att @ w^T + b + res (twins.py:622-623 / 676-677), computed by the same launch when the rows are 128 wide."""
        if proj is not None:
            att, (pw, pb), pres = proj
            dev = att.device
            if (fc1_ln is not None and FUSE_LN and FUSE_MLP and FUSE_PROJ and att.shape[1] == 128 and pw.is_contiguous()
                    and fc2[0].is_contiguous() and fc1_ln[0].is_contiguous() and pw.data_ptr() % 16 == 0 and pb.data_ptr() % 16 == 0):
                o = _new(att.shape[0], 128, dev) if out is None else out
                return ops.mlp128(att, o, fc1_ln[0], fc1_ln[1], fc2[0], fc2[1], ln_eps=eps, res=extra_res, proj=(pw, pb, pres), image=image)
            x = _new(att.shape[0], att.shape[1], dev)
            ops.conv_gemm(att, pw, x, bias=pb, aux0=pres)
        dev = x.device
        if fc1_ln is not None and FUSE_LN and FUSE_MLP and x.shape[1] == 128 and fc2[0].is_contiguous() and fc1_ln[0].is_contiguous():
            o = _new(x.shape[0], 128, dev) if out is None else out
            return ops.mlp128(x, o, fc1_ln[0], fc1_ln[1], fc2[0], fc2[1], ln_eps=eps, res=extra_res)
        h = _new(x.shape[0], fc1[0].shape[0], dev)
        if fc1_ln is not None and FUSE_LN:
            ops.conv_gemm(x, fc1_ln[0], h, bias=fc1_ln[1], act="gelu", ln_eps=eps)
        else:
            y = _new(x.shape[0], x.shape[1], dev)
            ops.layernorm(x, n2[0], n2[1], y, eps)
            ops.conv_gemm(y, fc1[0], h, bias=fc1[1], act="gelu")
        o = _new(x.shape[0], x.shape[1], dev) if out is None else out
        if extra_res is None:
            ops.conv_gemm(h, fc2[0], o, bias=fc2[1], aux0=x)
        else:
            ops.conv_gemm(h, fc2[0], o, bias=fc2[1], aux0=x, epi="add", aux1=extra_res)      # a second residual in the same epilogue
        return o

    # ------------------------------------------------------------------ Twins-SVT-L stages 1-2
    def _twins(self, t, x, B, H, W):
        """encoders.py:21-40; x: prepped image rows [B*H*W, 4] -> feature rows [B*(H/8)*(W/8), 256]."""
        dev = x.device
        dims, heads, srs, patch = (128, 256), (4, 8), (8, 4), (4, 2)
        for s in range(2):
            C, hd, sr, ps = dims[s], heads[s], srs[s], patch[s]
            H, W = H // ps, W // ps
            N = B * H * W
            e = _new(N, C, dev)
            ops.conv_gemm(x, t[f"pe{s}"][0], e, geom=(B, H * ps, W * ps, ps, ps, ps, ps, 0, 0), bias=t[f"pe{s}"][1])
            x = _new(N, C, dev)
            ops.layernorm(e, t[f"pen{s}"][0], t[f"pen{s}"][1], x, 1e-5)
            # ---- LSA block (twins.py:587-631)
            L = t[f"l{s}"]
            y = _new(N, C, dev)
            qkv = _new(N, 3 * C, dev)
            if L.get("qkv_s3") is not None and FUSE_LN:
                ops.rowlin128_split3(x, qkv, L["qkv_s3"], ln_eps=1e-6)
            elif "qkv_ln" in L and FUSE_LN:
                ops.conv_gemm(x, L["qkv_ln"][0], qkv, bias=L["qkv_ln"][1], ln_eps=1e-6)
            else:
                ops.layernorm(x, L["n1"][0], L["n1"][1], y, 1e-6)
                ops.conv_gemm(y, L["qkv"][0], qkv, bias=L["qkv"][1])
            att = _new(N, C, dev)
            ops.window_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], H * W * 3 * C, 3 * C, *L["pads"], att,
                                 H * W * C, C, B, H, W, hd, C // hd, 7, (C // hd) ** -0.5)
            x2 = self._mlp(None, L["n2"], L["fc1"], L["fc2"], 1e-6, fc1_ln=L.get("fc1_ln"), proj=(att, L["proj"], x), image=L.get("mlp_s3"))
            # ---- PEG (twins.py:793-808)
            x3 = _new(N, C, dev)
            ops.dwconv3x3_residual(x2, t[f"peg{s}"][0], t[f"peg{s}"][1], x3, B, H, W, C)
            # ---- GSA block (twins.py:633-680)
            Gk = t[f"g{s}"]
            ops.layernorm(x3, Gk["n1"][0], Gk["n1"][1], y, 1e-6)
            q = _new(N, C, dev)
            ops.conv_gemm(y, Gk["q"][0], q, bias=Gk["q"][1])
            Hk, Wk = H // sr, W // sr
            Nk = Hk * Wk
            xs = _new(B * Nk, C, dev)
            ops.conv_gemm(y, Gk["sr"][0], xs, geom=(B, H, W, sr, sr, sr, sr, 0, 0), bias=Gk["sr"][1])
            xsn = _new(B * Nk, C, dev)
            ops.layernorm(xs, Gk["srn"][0], Gk["srn"][1], xsn, 1e-5)
            kv = _new(B * Nk, 2 * C, dev)
            ops.conv_gemm(xsn, Gk["kv"][0], kv, bias=Gk["kv"][1])
            ops.attention_kvlds(q, (H * W * C, C), kv[:, :C], (Nk * 2 * C, 2 * C), kv[:, C:], (Nk * 2 * C, 2 * C), att,
                                (H * W * C, C), B, hd, H * W, Nk, C // hd, (C // hd) ** -0.5)
            x = self._mlp(None, Gk["n2"], Gk["fc1"], Gk["fc2"], 1e-6, fc1_ln=Gk.get("fc1_ln"), proj=(att, Gk["proj"], x3), image=Gk.get("mlp_s3"))
        return x, H, W

    # ------------------------------------------------------------------ cost-volume encoder
    def _patch_embed(self, cost_maps, M, H2, W2):
        """PatchEmbed.forward (encoder.py:60-95): cost maps [M, H2*W2] -> tokens [M*P, 128], P = H3*W3."""
        pe = self._pk["pe"]
        dev = cost_maps.device
        Hp, Wp = (H2 + 7) // 8 * 8, (W2 + 7) // 8 * 8           # zero-pad to a multiple of the patch size (:63-66)
        H1, W1, H2p, W2p, h, w = Hp // 2, Wp // 2, Hp // 4, Wp // 4, Hp // 8, Wp // 8
        P = h * w
        key = ("pe_tab", h, w)
        if key not in self._const:
            # the 64 sine channels only depend on the patch position: fold ffn_with_coord.0's PE half
            # into a per-position bias table  W[:, 64:] . pe(pos) + b   (encoder.py:77-88)
            tab_in = _new(P, 64, dev)
            ops.sine_pe(tab_in, 64, Wg=w, cscale=8.0, coff=4.0)
            tab = _new(P, 128, dev)
            ops.conv_gemm(tab_in, pe["f0"][0][:, 64:], tab, bias=pe["f0"][1])
            self._const[key] = tab
        # 64x64 maps: the first two convs run as one launch that keeps the first feature map (64 KiB per map, 537 MB per pair) on the CU
        s3, s4, f = _new(M * P, 64, dev), _new(M * P, 128, dev), _new(M * P, 128, dev)
        if SPLIT3 and S3_PE and FUSE_PE and H2 == 64 and W2 == 64 and "c4_s3" in pe:
            # Conv2d(32, 64, 6, 2, 2) -- 77 of the operator's 99 GFLOP -- on exact-split operands: the fused c0 + c2 launch emits bf16 planes
            # (no fp32 second feature map at all), in chunks of <= 16 384 maps (2 GiB buffer offsets)
            CH = 16384
            s2p = ops.Planes(min(M, CH) * 256, 32, dev)
            for m0 in range(0, M, CH):
                m1 = min(M, m0 + CH)
                ops.patch_embed_split3(cost_maps[m0:m1], pe["embed11"], pe["f0"][0].stride(0), self._const[key], s2p, pe["c4_s3"],
                                       s3[m0 * P:m1 * P], s4[m0 * P:m1 * P], f[m0 * P:m1 * P], m1 - m0, H2, W2, tail_image=pe.get("tail_s3"))
            return f, P
        s1 = None if (H2 == 64 and W2 == 64 and FUSE_PE) else _new(M * H1 * W1, 16, dev)
        s2 = _new(M * H2p * W2p, 32, dev)
        ops.patch_embed(cost_maps, pe["embed11"], pe["f0"][0].stride(0), self._const[key], s1, s2, s3, s4, f, M, H2, W2)
        return f, P

    def _latent_layer(self, L, x, M, first, tokens=None, P=0):
        """crossattentionlayer.py:37-56 (first=True: latents x patch tokens) / encoder.py:156-172."""
        dev = x.device if x is not None else tokens.device
        lat = self._pk["latents"]
        nl = lat.shape[0]
        if first:
            att = _new(M * nl, 128, dev)
            if nl == 8 and P % 2 == 0 and P <= 64:
                # scores against the raw tokens with the folded queries, per-pixel softmax + token pooling, then the value
                # projection per head on the pooled tokens: the [M*P, 256] K|V tensor is never built
                S = _new(M * P, nl * 8, dev)
                ops.conv_gemm(tokens, L["qfold"], S)
                z = _new(M * nl * 8, 128, dev)
                ops.latent_pool(S, tokens, z, M, P)
                ops.conv_gemm(z.view(M * nl, 8 * 128)[:, :128], L["wv_heads"][:16], att[:, :16], batch=8, bsa=128, bsw=16 * 128, bsc=16)
                proj_b = L["proj_b_fold"]
            else:
                qn = _new(nl, 128, dev)
                ops.layernorm(lat, L["n1"][0], L["n1"][1], qn, 1e-5)
                q = _new(nl, 128, dev)
                ops.conv_gemm(qn, L["q"][0], q, bias=L["q"][1])
                kv = _new(M * P, 256, dev)
                ops.conv_gemm(tokens, L["kv"][0], kv, bias=L["kv"][1])
                ops.attention_small(q, (0, 128), kv[:, :128], (P * 256, 256), kv[:, 128:], (P * 256, 256), att, (nl * 128, 128),
                                    M, 8, nl, P, 16, 16 ** -0.5)
                proj_b = L["proj"][1]
            x1 = _new(M * nl, 128, dev)
            ops.conv_gemm(att, L["proj"][0], x1, bias=proj_b, aux0=lat, row_mod=nl)
        else:
            qkv = _new(M * nl, 384, dev)
            if FUSE_LN and L.get("qkv_s3") is not None:
                ops.rowlin128_split3(x, qkv, L["qkv_s3"], ln_eps=1e-5)
            elif FUSE_LN:
                ops.conv_gemm(x, L["qkv_ln"][0], qkv, bias=L["qkv_ln"][1], ln_eps=1e-5)
            else:
                y = _new(M * nl, 128, dev)
                ops.layernorm(x, L["n1"][0], L["n1"][1], y, 1e-5)
                ops.conv_gemm(y, L["qkv"][0], qkv, bias=L["qkv"][1])
            att = _new(M * nl, 128, dev)
            ops.attention_small(qkv[:, :128], (nl * 384, 384), qkv[:, 128:256], (nl * 384, 384), qkv[:, 256:], (nl * 384, 384),
                                att, (nl * 128, 128), M, 8, nl, nl, 16, 16 ** -0.5)
            if FUSE_CHAIN:
                # proj + residual -> LayerNorm -> ffn.0 + GELU -> ffn.3 + residual in ONE launch: x1 and the hidden activation
                # never leave the CU (encoder.py:163-172)
                o = _new(M * nl, 128, dev)
                if L.get("mlp_s3") is not None:                 # ... on the split3 kernel (st_mlp128_split3, hidden = 128)
                    return ops.mlp128(att, o, L["f0_ln"][0], L["f0_ln"][1], L["f3"][0], L["f3"][1], ln_eps=1e-5, proj=(L["proj"][0], L["proj"][1], x), image=L["mlp_s3"])
                return ops.linear_chain128(att, o, [dict(w=L["proj"][0], bias=L["proj"][1], res=x),
                                                    dict(w=L["f0_ln"][0], bias=L["f0_ln"][1], act="gelu", ln_eps=1e-5),
                                                    dict(w=L["f3"][0], bias=L["f3"][1], res=1)])
            x1 = _new(M * nl, 128, dev)
            ops.conv_gemm(att, L["proj"][0], x1, bias=L["proj"][1], aux0=x)
        return self._mlp_plain(x1, L)

    @staticmethod
    def _mlp_plain(x, L):
        dev = x.device
        if FUSE_CHAIN:                                          # LayerNorm -> ffn.0 + GELU -> ffn.3 + residual, one launch
            if L.get("mlp_s3_plain") is not None:
                return ops.mlp128(x, _new(x.shape[0], x.shape[1], dev), L["f0_ln"][0], L["f0_ln"][1], L["f3"][0], L["f3"][1], ln_eps=1e-5, image=L["mlp_s3_plain"])
            return ops.linear_chain128(x, _new(x.shape[0], x.shape[1], dev),
                                       [dict(w=L["f0_ln"][0], bias=L["f0_ln"][1], act="gelu", ln_eps=1e-5), dict(w=L["f3"][0], bias=L["f3"][1], res=0)])
        h = _new(x.shape[0], L["f0"][0].shape[0], dev)
        if FUSE_LN:
            ops.conv_gemm(x, L["f0_ln"][0], h, bias=L["f0_ln"][1], act="gelu", ln_eps=1e-5)
        else:
            y = _new(x.shape[0], x.shape[1], dev)
            ops.layernorm(x, L["n2"][0], L["n2"][1], y, 1e-5)
            ops.conv_gemm(y, L["f0"][0], h, bias=L["f0"][1], act="gelu")
        o = _new(x.shape[0], x.shape[1], dev)
        ops.conv_gemm(h, L["f3"][0], o, bias=L["f3"][1], aux0=x)
        return o

    def _vertical(self, V, x, ctx, B, H1, W1, nl, extra_res=None):
        """VerticalSelfAttentionLayer (encoder.py:121-125): Block(LSA ws7) -> Block(GSA sr4) with context
        (twins.py:253-304, 336-392, 787-790).  x rows are (b, pixel n, latent l) -> row (b*N + n)*nl + l."""
        dev = x.device
        N = H1 * W1
        R = B * N * nl
        C, Cc = 128, 64
        Cq = C + Cc
        # ---------------- local block
        y = _new(R, C, dev)
        if not FUSE_LN:
            ops.layernorm(x, V["ln1"][0], V["ln1"][1], y, 1e-5)
        z = _new(B * N, Cq, dev)                                             # [0 + code | context projection + code]
        ops.conv_gemm(ctx, V["lctx"][0], z[:, C:], bias=V["lctx"][1])
        ops.sine_pe(z, Cq, Wg=W1, ws=7, period=N, accumulate=C)             # window-local code (twins.py:285-288)
        # q | k | v of every latent row in ONE N = 384 product: the context / position part of q and k is a per-pixel
        # table (row = m / nl), v's bias rides in the same table
        T = _new(B * N, 3 * C, dev)
        ops.conv_gemm(z, V["ltab"][0], T, bias=V["ltab"][1])
        qkv = _new(R, 3 * C, dev)
        if FUSE_LN and V.get("lqkv_s3") is not None:
            ops.rowlin128_split3(x, qkv, V["lqkv_s3"], ln_eps=1e-5, aux=T, row_div=nl)
        elif FUSE_LN:
            ops.conv_gemm(x, V["lqkv_ln"][0], qkv, bias=V["lqkv_ln"][1], aux0=T, row_div=nl, ln_eps=1e-5)
        else:
            ops.conv_gemm(y, V["lqkv"], qkv, aux0=T, row_div=nl)
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        key = ("lsa_pad", id(V))
        if key not in self._const:
            zp = _new(49, Cq, dev, zero=True)                                 # zero token + window code
            ops.sine_pe(zp, Cq, Wg=7, accumulate=True)
            qp, kp = _new(49, C, dev), _new(49, C, dev)
            ops.conv_gemm(zp, V["lq"][0], qp, bias=V["lq"][1])
            ops.conv_gemm(zp, V["lk"][0], kp, bias=V["lk"][1])
            self._const[key] = (qp, kp, V["lv"][1].expand(49, C).contiguous())
        qp, kp, vp = self._const[key]
        att = _new(R, C, dev)
        for b in range(B):
            sl = slice(b * N * nl, (b + 1) * N * nl)
            ops.window_attention(q[sl], k[sl], v[sl], 3 * C, nl * 3 * C, qp, kp, vp, att[sl], C, nl * C, nl, H1, W1, 8, 16, 7,
                                 16 ** -0.5)
        x2 = self._mlp(None, V["ln2"], V["lfc1"], V["lfc2"], 1e-5, fc1_ln=V["lfc1_ln"], proj=(att, V["lproj"], x), image=V.get("lmlp_s3"))
        # ---------------- global block
        ops.layernorm(x2, V["gn1"][0], V["gn1"][1], y, 1e-5)
        z = _new(B * N, Cq, dev)
        ops.conv_gemm(ctx, V["gctx"][0], z[:, C:], bias=V["gctx"][1])
        Hk, Wk = H1 // 4, W1 // 4
        Nk = Hk * Wk
        # pre-activation table of the fused [sr_key | sr_value] conv: key half = sr_key over the context channels + bias,
        # value half = sr_value's bias
        Tkv = _new(B * Nk, 2 * C, dev)
        ops.conv_gemm(z[:, C:], V["gtab"][0], Tkv, geom=(B, H1, W1, 4, 4, 4, 4, 0, 0), bias=V["gtab"][1])
        ops.sine_pe(z, Cq, Wg=W1, period=N, accumulate=C)                    # full-grid code on q (twins.py:358-361)
        Tq, q = _new(B * N, C, dev), _new(R, C, dev)
        ops.conv_gemm(z, V["gq"][0], Tq, bias=V["gq"][1])
        ops.conv_gemm(y, V["gq"][0][:, :C], q, aux0=Tq, row_div=nl)
        key = ("gsa_kpe", id(V), Hk, Wk)
        if key not in self._const:
            pe_k = _new(Nk, C, dev)
            ops.sine_pe(pe_k, C, Wg=Wk, cscale=4.0)                          # stride-4 code on k (twins.py:372-376)
            tk = _new(Nk, C, dev)
            ops.conv_gemm(pe_k, V["gk"][0], tk, bias=V["gk"][1])
            self._const[key] = tk
        Tkpe = self._const[key]
        # 4x4 stride-4 spatial reduction of every latent's pixel grid (twins.py:364-371): latent l is the channel
        # slice [l*C, (l+1)*C) of the [B*N, nl*C] row view, so all latents, both batches and both convs are ONE
        # batched launch; rows come out as [l][b][Nk], columns [key | value]
        xkv = _new(nl * B * Nk, 2 * C, dev)
        ops.conv_gemm(y.view(B * N, nl * C)[:, :C], V["gskv"], xkv[:B * Nk], geom=(B, H1, W1, 4, 4, 4, 4, 0, 0),
                      aux0=Tkv, batch=nl, bsa=C, bsw=0, bsc=B * Nk * 2 * C)
        xkvn = _new(nl * B * Nk, 2 * C, dev)
        # the key and the value half take the same LayerNorm (twins.py:371): rows of 2C = two rows of C in one launch
        ops.layernorm(xkv.view(-1, C), V["gsrn"][0], V["gsrn"][1], xkvn.view(-1, C), 1e-5)
        kv = _new(nl * B * Nk, 2 * C, dev)
        ops.conv_gemm(xkvn[:, :C], V["gk"][0], kv[:, :C], aux0=Tkpe, row_mod=Nk)
        ops.conv_gemm(xkvn[:, C:], V["gv"][0], kv[:, C:], bias=V["gv"][1])
        for b in range(B):
            sl = slice(b * N * nl, (b + 1) * N * nl)
            kb = kv[b * Nk:]
            ops.attention_kvlds(q[sl], (C, nl * C), kb[:, :C], (B * Nk * 2 * C, 2 * C), kb[:, C:], (B * Nk * 2 * C, 2 * C),
                                att[sl], (C, nl * C), nl, 8, N, Nk, 16, 16 ** -0.5)
        return self._mlp(None, V["gn2"], V["gfc1"], V["gfc2"], 1e-5, fc1_ln=V["gfc1_ln"], extra_res=extra_res, proj=(att, V["gproj"], x2), image=V.get("gmlp_s3"))

    def _cost_encoder(self, cost_maps, ctx, B, H1, W1, ctx_ready=None):
        """CostPerceiverEncoder.forward (encoder.py:258-287) -> cost memory rows [B*N*8, 128]."""
        pk = self._pk
        M = B * H1 * W1
        tokens, P = self._patch_embed(cost_maps, M, H1, W1)
        x = self._latent_layer(pk["xin"], None, M, True, tokens, P)
        short = x
        nl = pk["latents"].shape[0]
        for i in range(HP["encoder_depth"]):
            x = self._latent_layer(pk["self"][i], x, M, False)
            if i == 0 and ctx_ready is not None:
                torch.cuda.current_stream().wait_event(ctx_ready)            # ST_FORK: the context comes from the side stream
            # cost_encoder_res (encoder.py:281-282) adds the short-cut to the output of the last layer: a second residual operand
            # in that layer's final GEMM epilogue (no add pass, one k/v projection in the decoder)
            x = self._vertical(pk["vert"][i], x, ctx, B, H1, W1, nl, extra_res=short if i == HP["encoder_depth"] - 1 else None)
        return x, None

    # ------------------------------------------------------------------ decoder
    def _gru_tables(self, inp, B, H1, W1):
        """per-pass SepConvGRU tables: conv over the constant `inp` channels + bias, [z | r | q] for each half."""
        D = self._pk["dec"]
        gru_geom = {"1": (B, H1, W1, 1, 5, 1, 1, 0, 2), "2": (B, H1, W1, 5, 1, 1, 1, 2, 0)}
        tabs = {}
        for sfx in ("1", "2"):
            tabs[sfx] = _new(inp.shape[0], 384, inp.device)
            ops.conv_gemm(inp, D["inp" + sfx][0], tabs[sfx], geom=gru_geom[sfx], bias=D["inp" + sfx][1])
        return tabs

    def _update_state(self, R, B, N, dev):
        """work buffers of the refinement loop.  hxA = [h | motion(126)+flow(2) | motion_global], hxB = [r*h | unused] (same stride);
        corr = [cost_forward 81 | 3 zero | cost_global 64 | 12 zero]."""
        S = dict(hxA=_new(R, 384, dev), hxB=_new(R, 384, dev), corr=_new(R, 160, dev, zero=True),
                 cor1=_new(R, 256, dev), corflo=_new(R, 256, dev), flo1=_new(R, 128, dev),
                 vT=torch.empty((B, 128, N), device=dev), zbuf=_new(R, 128, dev), fh=_new(R, 256, dev))
        S["s3"] = SPLIT3 and N % 32 == 0 and "s3" in self._pk["dec"]          # (plane rows are whole 32-row tiles; other map sizes keep the fp32 kernels)
        if S["s3"]:
            # plane images (ops.Planes) of the tensors the split3 contractions read; hxB's image only ever holds r*h (columns 0..127) but
            # shares hxA's strides (second A source of the q convs).  Every channel that is read is written first in each iteration:
            # hxA 0..127 by the q convs / proj_net, 128..253 by `conv`, 254..255 by flow_encode, 256..383 by the aggregate.
            S.update(hxA_p=ops.Planes(R, 384, dev), hxB_p=ops.Planes(R, 384, dev), cor1_p=ops.Planes(R, 256, dev), flo1_p=ops.Planes(R, 128, dev),
                     corflo_p=ops.Planes(R, 256, dev), vT_p=ops.Planes(B * 128, N, dev))
        return S

    def _update_block(self, S, coords1, attn, gru_tab, B, H1, W1, enc_done=None):
        """GMAUpdateBlock.forward (gru.py:322-334) without the mask head: BasicMotionEncoder (gru.py:246-254), GMA
        aggregate (gma.py:102-115), SepConvGRU (gru.py:44-59), flow head (gru.py:5-13); coords1 += delta_flow
        (decoder.py:329).  Reads S['corr'] (cost_forward | cost_global), updates S['hxA'][:, :128] (net) and coords1."""
        D = self._pk["dec"]
        N = H1 * W1
        hxA, hxB, corr = S["hxA"], S["hxB"], S["corr"]
        g3 = (B, H1, W1, 3, 3, 1, 1, 1, 1)
        if S["s3"]:
            W3 = D["s3"]
            hxA_p = S["hxA_p"]
            if S3_AGG and torch.is_tensor(attn):      # a caller that built the attention matrix itself (tests): its planes, here
                attn = ops.split3_pack(attn.view(B * N, N))
            # convc1 (K = 160) stays on the fp32 kernel and emits cor1's planes; flow_encode emits flo1's and the flow's two channels
            ops.conv_gemm(corr, D["convc1"][0], S["cor1"], bias=D["convc1"][1], act="relu", out_planes=S["cor1_p"])
            ops.flow_encode_split3(coords1, D["convf1"][0], D["convf1"][1], S["flo1"], hxA[:, 254:256], B, H1, W1, S["flo1_p"], (hxA_p, 254))
            nf = not (S3_OFF & 16)
            if S3_OFF & 32:
                ops.conv_gemm_pair((S["cor1"], D["convc2"][0], S["corflo"][:, :192], dict(geom=g3, bias=D["convc2"][1], act="relu", out_planes=S["corflo_p"].cols(0, 192))),
                                   (S["flo1"], D["convf2"][0], S["corflo"][:, 192:], dict(geom=g3, bias=D["convf2"][1], act="relu", out_planes=S["corflo_p"].cols(192, 256))))
            elif not S3_PAIR:
                ops.conv_gemm(S["cor1_p"], W3["convc2"], S["corflo"][:, :192], geom=g3, bias=D["convc2"][1], act="relu", out_planes=S["corflo_p"].cols(0, 192), no_f32=nf)
                ops.conv_gemm(S["flo1_p"], W3["convf2"], S["corflo"][:, 192:], geom=g3, bias=D["convf2"][1], act="relu", out_planes=S["corflo_p"].cols(192, 256), no_f32=nf)
            else:
                ops.conv_gemm_pair((S["cor1_p"], W3["convc2"], S["corflo"][:, :192],
                                    dict(geom=g3, bias=D["convc2"][1], act="relu", out_planes=S["corflo_p"].cols(0, 192), no_f32=nf)),
                                   (S["flo1_p"], W3["convf2"], S["corflo"][:, 192:],
                                    dict(geom=g3, bias=D["convf2"][1], act="relu", out_planes=S["corflo_p"].cols(192, 256), no_f32=nf)))
            if S3_OFF & 16:
                ops.conv_gemm(S["corflo"], D["conv"][0], hxA[:, 128:254], geom=g3, bias=D["conv"][1], act="relu", out_planes=hxA_p.cols(128, 256))
            else:
                ops.conv_gemm(S["corflo_p"], W3["conv"], hxA[:, 128:254], geom=g3, bias=D["conv"][1], act="relu", out_planes=hxA_p.cols(128, 256))
            if S3_AGG:
                ops.gma_aggregate_split3(attn, hxA[:, 128:256], D["to_v"], D["gamma"], S["vT"], S["vT_p"], hxA[:, 256:], hxA_p.cols(256, 384), B, N)
            else:
                # the aggregate reads the whole attention matrix every iteration and is HBM-bound either way (fp32: 134 MB per launch, 64.8 us in
                # the chain; planes: 201 MB, 66.2 us): it stays on the fp32 kernel, whose epilogue emits the planes of its result
                ops.gma_aggregate(attn, hxA[:, 128:256], D["to_v"], D["gamma"], S["vT"], hxA[:, 256:], B, N, out_planes=hxA_p.cols(256, 384))
            if S3_OFF & 8:
                ops.sepconv_gru(hxA, hxB, S["zbuf"], gru_tab["1"], gru_tab["2"], D["zr1"], D["q1"], D["zr2"], D["q2"], B, H1, W1)
                ops.split3_pack(hxA[:, :128], out=hxA_p.cols(0, 128))
            else:
                ops.sepconv_gru_split3(hxA, hxA_p, S["hxB_p"], S["zbuf"], gru_tab["1"], gru_tab["2"], W3["zr1"], W3["q1"], W3["zr2"], W3["q2"], B, H1, W1)
            if S3_OFF & 2:
                ops.conv_gemm(hxA[:, :128], D["fh1"][0], S["fh"], geom=g3, bias=D["fh1"][1], act="relu")
            else:
                ops.conv_gemm(hxA_p.cols(0, 128), W3["fh1"], S["fh"], geom=g3, bias=D["fh1"][1], act="relu")
            ops.conv_gemm(S["fh"], D["fh2"][0], coords1, geom=g3, bias=D["fh2"][1], epi="add", aux1=coords1)
            return
        ops.conv_gemm(corr, D["convc1"][0], S["cor1"], bias=D["convc1"][1], act="relu")
        if enc_done is not None:
            torch.cuda.current_stream().wait_event(enc_done)         # ST_FORK_ENC: already enqueued on the side stream by _decoder
        else:
            ops.flow_encode(coords1, D["convf1"][0], D["convf1"][1], S["flo1"], hxA[:, 254:256], B, H1, W1)      # :321, gru.py:251,254
        # convc2 (384 tiles) and convf2 (128 tiles) are independent and ready together: one launch, two workgroups per CU, no
        # split-K slabs (gru.py:252-253)
        if PAIR_CONVS:
            ops.conv_gemm_pair((S["cor1"], D["convc2"][0], S["corflo"][:, :192], dict(geom=g3, bias=D["convc2"][1], act="relu")),
                               (S["flo1"], D["convf2"][0], S["corflo"][:, 192:], dict(geom=g3, bias=D["convf2"][1], act="relu")))
        else:
            ops.conv_gemm(S["cor1"], D["convc2"][0], S["corflo"][:, :192], geom=g3, bias=D["convc2"][1], act="relu", split_k=1)
            ops.conv_gemm(S["flo1"], D["convf2"][0], S["corflo"][:, 192:], geom=g3, bias=D["convf2"][1], act="relu", split_k=1)
        ops.conv_gemm(S["corflo"], D["conv"][0], hxA[:, 128:254], geom=g3, bias=D["conv"][1], act="relu")
        # GMA aggregate: v^T = Wv . mf^T, out = mf + gamma * attn @ v
        ops.gma_aggregate(attn, hxA[:, 128:256], D["to_v"], D["gamma"], S["vT"], hxA[:, 256:], B, N)
        # SepConvGRU: horizontal 1x5 then vertical 5x1
        ops.sepconv_gru(hxA, hxB, S["zbuf"], gru_tab["1"], gru_tab["2"], D["zr1"], D["q1"], D["zr2"], D["q2"], B, H1, W1)
        ops.conv_gemm(hxA[:, :128], D["fh1"][0], S["fh"], geom=g3, bias=D["fh1"][1], act="relu")
        ops.conv_gemm(S["fh"], D["fh2"][0], coords1, geom=g3, bias=D["fh2"][1], epi="add", aux1=coords1)

    def _mask_head(self, S, B, H1, W1):
        """mask = .25 * conv1x1(relu(conv3x3(net))) (gru.py:315-318,333) -> rows [R, 576]."""
        D = self._pk["dec"]
        if S["s3"] and not (S3_OFF & 1):
            ops.conv_gemm(S["hxA_p"].cols(0, 128), D["s3"]["m0"], S["fh"], geom=(B, H1, W1, 3, 3, 1, 1, 1, 1), bias=D["m0"][1], act="relu")
        else:
            ops.conv_gemm(S["hxA"][:, :128], D["m0"][0], S["fh"], geom=(B, H1, W1, 3, 3, 1, 1, 1, 1), bias=D["m0"][1], act="relu")
        mask = _new(S["hxA"].shape[0], 576, S["hxA"].device)
        ops.conv_gemm(S["fh"], D["m2"][0], mask, bias=D["m2"][1], alpha=0.25)
        return mask

    def _decoder_prologue(self, ctx, B, H1, W1):
        """what the decoder derives from the context alone, once per pass: net / inp (decoder.py:283-287), the SepConvGRU tables, the GMA
        attention matrix (gma.py:54-76)."""
        D = self._pk["dec"]
        dev = ctx.device
        N = H1 * W1
        R = B * N
        S = self._update_state(R, B, N, dev)
        inp = _new(R, 128, dev)
        ops.conv_gemm(ctx, D["proj_net"][0], S["hxA"][:, :128], bias=D["proj_net"][1], act="tanh",
                      out_planes=S["hxA_p"].cols(0, 128) if S["s3"] else None)
        ops.conv_gemm(ctx, D["proj_inp"][0], inp, bias=D["proj_inp"][1], act="relu")
        gru_tab = self._gru_tables(inp, B, H1, W1)
        qk = _new(R, 256, dev)
        attn = torch.empty((B, N, N), device=dev)
        ops.gma_attention(inp, D["qk"], qk, attn, B, N)
        if S["s3"] and S3_AGG:     # (ST_S3_AGG=1: the aggregate on planes too: the attention matrix's planes, once per pass)
            attn = ops.split3_pack(attn.view(B * N, N))
        return dict(S=S, inp=inp, gru_tab=gru_tab, attn=attn, qk=qk)

    def _decoder(self, mem, mem_short, ctx, cost_maps, B, H1, W1, iters, trace=None, pre=None):
        """MemoryDecoder.forward eval branch (decoder.py:262-344)."""
        D = self._pk["dec"]
        dev = ctx.device
        N = H1 * W1
        R = B * N
        nl = self._pk["latents"].shape[0]
        if pre is None:
            pre = self._decoder_prologue(ctx, B, H1, W1)
        S, gru_tab, attn = pre["S"], pre["gru_tab"], pre["attn"]
        # k, v of the cost-memory cross attention, once (decoder.py:68-70); memory = x + short_cut (linear -> two GEMMs)
        ca = D["ca"]
        kv = _new(R * nl, 128, dev)
        if mem_short is None:
            ops.conv_gemm(mem, ca["kv"][0], kv, bias=ca["kv"][1])
        else:                                                  # memory given as two addends: kv(x + s) = kv(x) + kv(s)
            kv0 = _new(R * nl, 128, dev)
            ops.conv_gemm(mem_short, ca["kv"][0], kv0, bias=ca["kv"][1])
            ops.conv_gemm(mem, ca["kv"][0], kv, aux0=kv0)
        coords1 = _new(R, 2, dev)
        ops.coords_grid(coords1, B, H1, W1)
        for it in range(iters):
            enc_done = None
            if FORK_ENC:
                cur = torch.cuda.current_stream()
                side = _side_stream(cur)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    ops.flow_encode(coords1, D["convf1"][0], D["convf1"][1], S["flo1"], S["hxA"][:, 254:256], B, H1, W1)
                    enc_done = torch.cuda.Event()
                    enc_done.record(side)
            ops.cost_lookup9x9(cost_maps, coords1, S["corr"], R, H1, W1)                              # decoder.py:291
            # flow_token_encoder + cost-memory cross attention + FFN: one fused launch (decoder.py:305-312)
            ops.decoder_token_chain(S["corr"], coords1, kv, D["chain16"], R, nl)
            self._update_block(S, coords1, attn, gru_tab, B, H1, W1, enc_done=enc_done)
            if trace is not None:
                trace.append(dict(coords1=coords1.clone(), net=S["hxA"][:, :128].clone(), corr=S["corr"].clone()))
        # mask head + convex upsampling, last iteration only (gru.py:315-318,333; decoder.py:214-225)
        mask = self._mask_head(S, B, H1, W1)
        flow_up = torch.empty((B, 2, 8 * H1, 8 * W1), device=dev)
        ops.convex_upsample(coords1, mask, flow_up, B, H1, W1)
        return flow_up, coords1

    # ================================================================== forward
    def flow_rows(self, image1, image2, iters=None, trace=None):
        """images NCHW 0..255 -> (flow_up [B,2,H,W], coords1 rows [B*N,2], (B,H1,W1))."""
        if not image1.is_cuda:
            raise RuntimeError("FlowFormer runs on the MI355X HIP kernels only: move the module and inputs to cuda")
        pk = self._pk or self.pack()
        iters = HP["decoder_depth"] if iters is None else iters
        B, _, H, W = image1.shape
        if H % 32 or W % 32:
            raise RuntimeError(f"input size {H}x{W} must be a multiple of 32 (reference runs both nets at 512x512)")
        dev = image1.device
        x = _new(2 * B * H * W, 4, dev)
        ops.prep_image(image1.contiguous(), x[:B * H * W], 4, 2.0, 255.0, 1.0)        # transformer.py:53-54
        ops.prep_image(image2.contiguous(), x[B * H * W:], 4, 2.0, 255.0, 1.0)
        ctx, H1, W1 = self._twins(pk["cnet"], x[:B * H * W], B, H, W)                 # context = cnet(image1)
        feats, _, _ = self._twins(pk["fnet"], x, 2 * B, H, W)                        # fnet(image1), fnet(image2)
        N = H1 * W1
        feats = feats.view(2, B, N, 256)
        cost_maps = torch.empty((B * N, N), device=dev)                              # all-pairs volume (encoder.py:359-369)
        ops.corr_volume(feats[0], feats[1], cost_maps.view(B, N, N))
        mem, short = self._cost_encoder(cost_maps, ctx, B, H1, W1)
        if trace is not None:
            trace.append(dict(context=ctx, feats=feats, cost_maps=cost_maps, mem=mem, short=short))
        flow_up, coords1 = self._decoder(mem, short, ctx, cost_maps, B, H1, W1, iters, trace)
        return flow_up, coords1, (B, H1, W1)

    def flow_rows_pair(self, image_a, image_b, iters=None):
        """Both directions at once: returns flow_up [2B,2,H,W] = [flow a->b ; flow b->a].

        The stitching path always needs the forward AND the backward flow of the same image pair
        (flowHomoAdpater.py:167,178 / :236,326) and the two FlowFormer passes are independent, so they run
        as one batch of 2B: the feature encoder sees each image once instead of twice (the reference
        recomputes fnet(a), fnet(b) in the second pass), every GEMM of the encoder/decoder has twice the
        rows (M = 8192 instead of 4096 per launch), and the launch count per pair halves."""
        if not image_a.is_cuda:
            raise RuntimeError("FlowFormer runs on the MI355X HIP kernels only: move the module and inputs to cuda")
        pk = self._pk or self.pack()
        iters = HP["decoder_depth"] if iters is None else iters
        B, _, H, W = image_a.shape
        if H % 32 or W % 32:
            raise RuntimeError(f"input size {H}x{W} must be a multiple of 32 (reference runs both nets at 512x512)")
        dev = image_a.device
        x = _new(2 * B * H * W, 4, dev)
        ops.prep_image(image_a.contiguous(), x[:B * H * W], 4, 2.0, 255.0, 1.0)
        ops.prep_image(image_b.contiguous(), x[B * H * W:], 4, 2.0, 255.0, 1.0)
        pre = ctx_ready = side = None
        if FORK:
            cur = torch.cuda.current_stream()
            side = _side_stream(cur)
            side_ws = ops.side_workspace(dev)                                # split-K slabs of the side branch (looked up on THIS stream)
            side.wait_stream(cur)
            with torch.cuda.stream(side), ops.workspace_scope(side_ws):
                ctx, H1, W1 = self._twins(pk["cnet"], x, 2 * B, H, W)
                ctx_ready = torch.cuda.Event()
                ctx_ready.record(side)
                pre = self._decoder_prologue(ctx, 2 * B, H1, W1)
        else:
            ctx, H1, W1 = self._twins(pk["cnet"], x, 2 * B, H, W)          # context of a (pass a->b) then of b (pass b->a)
        feats, _, _ = self._twins(pk["fnet"], x, 2 * B, H, W)
        N = H1 * W1
        feats = feats.view(2, B, N, 256)
        cost_maps = torch.empty((2 * B * N, N), device=dev)
        # the reverse direction's volume is the transpose of the forward one: one product, two stores
        ops.corr_volume_both(feats[0], feats[1], cost_maps[:B * N].view(B, N, N), cost_maps[B * N:].view(B, N, N))
        mem, short = self._cost_encoder(cost_maps, ctx, 2 * B, H1, W1, ctx_ready=ctx_ready)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        flow_up, coords1 = self._decoder(mem, short, ctx, cost_maps, 2 * B, H1, W1, iters, pre=pre)
        return flow_up, coords1, (2 * B, H1, W1)

    def forward(self, image1, image2, mask=None, output=None, flow_init=None):
        """Reference surface (transformer.py:47-65, eval): returns (flow_up, flow_lowres)."""
        if flow_init is not None:
            raise NotImplementedError("flow_init (warm start) is not on the stitching path")
        flow_up, coords1, (B, H1, W1) = self.flow_rows(image1, image2)
        flow4 = _new(B * H1 * W1, 4, flow_up.device)
        ops.flow_from_coords(coords1, flow4, None, B, H1, W1)
        low = flow4[:, :2].reshape(B, H1, W1, 2).permute(0, 3, 1, 2).contiguous()       # layout only (unused by the adapter)
        return flow_up, low


def build_flowformer(cfg=None):
    """reference: core/FlowFormer/__init__.py:2-9 (only 'percostformer3' exists)."""
    name = getattr(cfg, "transformer", "percostformer3") if cfg is not None else "percostformer3"
    if name != "percostformer3":
        raise ValueError(f"FlowFormer = {name} is not a valid optimizer!")
    sub = cfg[name] if cfg is not None and hasattr(cfg, "__getitem__") and name in cfg else None
    return FlowFormer(sub)
