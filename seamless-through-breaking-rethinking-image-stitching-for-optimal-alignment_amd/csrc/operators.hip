// Operator-level entry points: one C symbol per reference operator of the FlowFormer / stitching path
// (SURVEY.md 8b minimum symbol set).  Host-side composition only: each function enqueues the library's own
// kernels (gemm.hip, nn.hip, flowops.hip, geom.hip) on the caller's stream, works in caller-provided scratch,
// allocates nothing and keeps no state.
#include "common.h"
#include "../../include/stitch_gfx950.h"
#include <string.h>

namespace {

struct Gemm {
    st_gemm_desc d;
    Gemm(const float* a, int32_t lda, const float* w, int32_t ldw, float* c, int32_t ldc, int32_t M, int32_t N, int32_t Cin) {
        memset(&d, 0, sizeof(d));
        d.a = a; d.w = w; d.c = c;
        d.M = M; d.N = N; d.K = Cin;
        d.H = 1; d.W = M; d.Cin = Cin; d.ldx = lda;
        d.kh = d.kw = d.sh = d.sw = 1; d.Ho = 1; d.Wo = M;
        d.ldw = ldw; d.ldc = ldc;
        d.alpha = 1.f; d.batch = 1;
    }
    Gemm& conv(int32_t B, int32_t H, int32_t W, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t ph, int32_t pw,
               int32_t Ho = -1, int32_t Wo = -1) {
        d.H = H; d.W = W; d.kh = kh; d.kw = kw; d.sh = sh; d.sw = sw; d.ph = ph; d.pw = pw;
        d.Ho = Ho >= 0 ? Ho : (H + 2 * ph - kh) / sh + 1;
        d.Wo = Wo >= 0 ? Wo : (W + 2 * pw - kw) / sw + 1;
        d.M = B * d.Ho * d.Wo;
        d.K = kh * kw * d.Cin;
        return *this;
    }
    Gemm& bias(const float* b) { d.bias = b; return *this; }
    Gemm& act(int a) { d.act = a; return *this; }
    Gemm& alpha(float a) { d.alpha = a; return *this; }
    Gemm& aux0(const float* p, int32_t ld, int32_t row_div = 0, int32_t row_mod = 0) {
        d.aux0 = p; d.ld_aux0 = ld; d.aux0_row_div = row_div; d.aux0_row_mod = row_mod; return *this;
    }
    Gemm& epi(int e, const float* a1, int32_t ld1, const float* a2 = nullptr, int32_t ld2 = 0) {
        d.epi = e; d.aux1 = a1; d.ld_aux1 = ld1; d.aux2 = a2; d.ld_aux2 = ld2; return *this;
    }
    Gemm& scale(const float* s) { d.scale_ptr = s; return *this; }
    Gemm& out2(float* c2, int32_t ld) { d.c2 = c2; d.ldc2 = ld; return *this; }
    Gemm& a2(const float* p, int32_t channels) { d.a2 = p; d.a2_channels = channels; return *this; }
    Gemm& batched(int32_t n, int64_t sa, int64_t sw, int64_t sc) {
        d.batch = n; d.batch_stride_a = sa; d.batch_stride_w = sw; d.batch_stride_c = sc; return *this;
    }
    Gemm& work(void* ws, int64_t floats) { d.workspace = (float*)ws; d.workspace_floats = floats; return *this; }
    int run(void* stream) { return st_conv_gemm(&d, stream); }
};

#define ST_TRY(expr)            \
    do {                        \
        int rc__ = (expr);      \
        if (rc__) return rc__;  \
    } while (0)

}  // namespace

extern "C" {

// encode_flow_token (decoder.py:242-260): the 9x9 (r = 4) window of the reference configuration.
int st_cost_lookup9x9(const float* maps, const float* coords, float* out, int32_t ldo, int32_t Nq, int32_t H2, int32_t W2,
                      void* stream) {
    return st_cost_lookup(maps, coords, out, ldo, Nq, H2, W2, 4, stream);
}

// warp(x, flow) [* mask] (core/warp_utils.py:54-80 + flowHomoAdpater.py:171-172,339-341).
int st_grid_sample_blend(const float* x, const float* flow, const float* mul, float* out, int32_t B, int32_t C, int32_t H,
                         int32_t W, void* stream) {
    return st_flow_warp(x, flow, mul, out, B, C, H, W, stream);
}

// preprocess_occlusion_mask with the reference's fixed 19x19 structuring element (flowHomoAdpater.py:18-35).
int st_morph_open19(const float* mask, float* out, void* scratch_u8x2, int32_t N, int32_t H, int32_t W, void* stream) {
    return st_morph_open(mask, out, scratch_u8x2, N, H, W, 19, stream);
}

// PatchEmbed.forward (encoder.py:60-95) for patch_size 8 / 'single' / linear PE:
//   Conv2d(1,16,6,2,2)+ReLU -> Conv2d(16,32,6,2,2)+ReLU -> Conv2d(32,64,6,2,2) -> [x | sinePE] 1x1 + ReLU -> 1x1 -> LN.
// The sine half of ffn_with_coord.0 depends on the patch position only, so it arrives as the [P,128] table
// pe_bias = W0[:,64:] . pe(pos) + b0.  weights (host array of 11 device pointers):
//   0 c0_w[36,16] 1 c0_b | 2 c2_w[32,576] 3 c2_b | 4 c4_w[64,1152] 5 c4_b | 6 f0_w[128,ld_f0 (cols 0..63 used)]
//   | 7 f2_w[128,128] 8 f2_b | 9 ln_w 10 ln_b
// scratch: s1 [M*H/2*W/2,16] (may be NULL for 64x64 maps with ST_FUSE_PE != 0: st_patch_conv12 keeps it on the CU; ST_EINVAL if the unfused launches would need it), s2 [M*H/4*W/4,32],
// s3 [M*P,64], s4 [M*P,128]; tokens [M*P,128], P = (H/8)*(W/8).
int st_patch_embed(const float* cost_maps, const float* const* weights, int32_t ld_f0, const float* pe_bias, float* s1,
                   float* s2, float* s3, float* s4, float* tokens, int32_t M, int32_t H, int32_t W, void* workspace,
                   int64_t workspace_floats, void* stream) {
    if (!cost_maps || !weights || !pe_bias || !s2 || !s3 || !s4 || !tokens || M <= 0 || H <= 0 || W <= 0)
        return ST_EINVAL;
    const int Hp = (H + 7) / 8 * 8, Wp = (W + 7) / 8 * 8;      // zero pad to the patch size (encoder.py:63-66)
    const int H1 = Hp / 2, W1 = Wp / 2, H2 = Hp / 4, W2 = Wp / 4, H3 = Hp / 8, W3 = Wp / 8, P = H3 * W3;
    // the first two convs of a 64x64 map in one launch, s1 kept on the CU (csrc/patchembed.hip; bit-identical).  ST_FUSE_PE=0: unfused (A/B)
    static const bool fuse12 = [] { const char* e = getenv("ST_FUSE_PE"); return !(e && e[0] == '0'); }();
    const bool can_fuse = fuse12 && H == 64 && W == 64 && !(((uintptr_t)cost_maps | (uintptr_t)weights[2]) & 15);
    if (!s1 && !can_fuse) return ST_EINVAL;              // s1 == NULL: the caller counts on the fused launch (it saves the 64 KiB-per-map scratch)
    if (can_fuse) {
        ST_TRY(st_patch_conv12(cost_maps, weights[0], weights[1], weights[2], weights[3], s2, M, H, W, stream));
    } else {
        ST_TRY(st_patch_conv1(cost_maps, weights[0], weights[1], s1, M, H, W, H1, W1, stream));
        // 16-channel input, even width / stride / padding: pixel pairs (2x', 2x'+1) are 32 contiguous channels, so the
        // 6x6 stride-2 pad-2 conv is read as a 6x3 conv, stride (2,1), pad (2,1) over [M, H1, W1/2, 32] with the SAME
        // weight memory ((ky,kx,c) order) -- which makes every K step one whole 128-byte line for the LDS-DMA kernel.
        ST_TRY(Gemm(s1, 32, weights[2], 576, s2, 32, 0, 32, 32).conv(M, H1, W1 / 2, 6, 3, 2, 1, 2, 1, H2, W2)
                   .bias(weights[3]).act(ST_ACT_RELU).work(workspace, workspace_floats).run(stream));
    }
    ST_TRY(Gemm(s2, 32, weights[4], 1152, s3, 64, 0, 64, 32).conv(M, H2, W2, 6, 6, 2, 2, 2, 2, H3, W3)
               .bias(weights[5]).work(workspace, workspace_floats).run(stream));
    ST_TRY(Gemm(s3, 64, weights[6], ld_f0, s4, 128, M * P, 128, 64).aux0(pe_bias, 128, 0, P).act(ST_ACT_RELU)
               .work(workspace, workspace_floats).run(stream));
    ST_TRY(Gemm(s4, 128, weights[7], 128, tokens, 128, M * P, 128, 128).bias(weights[8])
               .work(workspace, workspace_floats).run(stream));
    return st_layernorm(tokens, 128, weights[9], weights[10], tokens, 128, M * P, 128, 1e-5f, stream);
}

// st_patch_embed for 64x64 cost maps with the third convolution (Conv2d(32,64,6,2,2), 77 of the operator's 99 GFLOP per pair) on exact-split
// operands (st_gemm_desc.split3): the fused c0 + c2 launch emits its result as bf16 planes (s2_planes: [3][1][M*256][32], s2_pstride elements
// apart, no fp32 copy), c4_w_planes = st_split3_pack of c4_w [64, 1152].  M * 256 * 32 * 6 bytes must stay below 2 GiB (M <= 16 384 maps per call).
// tail_image (st_pe_tail_split3_pack of weights[6] / weights[7]) != NULL: the three launches behind the third convolution (64 -> 128 + position table + ReLU,
// 128 -> 128 + bias, LayerNorm) run as ONE (st_pe_tail_split3); s4 is then not written and may be NULL.
int st_patch_embed_split3(const float* cost_maps, const float* const* weights, int32_t ld_f0, const float* pe_bias, void* s2_planes,
                          int64_t s2_pstride, const void* c4_w_planes, int64_t c4_w_pstride, float* s3, float* s4, float* tokens, int32_t M,
                          int32_t H, int32_t W, const void* tail_image, int64_t tail_image_bytes, void* workspace, int64_t workspace_floats, void* stream) {
    if (!cost_maps || !weights || !pe_bias || !s2_planes || !c4_w_planes || !s3 || (!s4 && !tail_image) || !tokens || M <= 0 || H != 64 || W != 64) return ST_EINVAL;
    const int H2 = 16, W2 = 16, H3 = 8, W3 = 8, P = 64;
    ST_TRY(st_patch_conv12_planes(cost_maps, weights[0], weights[1], weights[2], weights[3], s2_planes, s2_pstride, M, H, W, stream));
    {
        Gemm g((const float*)s2_planes, 32, (const float*)c4_w_planes, 1152, s3, 64, 0, 64, 32);
        g.conv(M, H2, W2, 6, 6, 2, 2, 2, 2, H3, W3).bias(weights[5]).work(workspace, workspace_floats);
        g.d.split3 = 1; g.d.a_plane_stride = s2_pstride; g.d.a_rows = (int64_t)M * H2 * W2; g.d.w_plane_stride = c4_w_pstride; g.d.w_rows = 64;
        ST_TRY(g.run(stream));
    }
    if (tail_image)
        return st_pe_tail_split3(s3, pe_bias, P, tail_image, tail_image_bytes, weights[8], weights[9], weights[10], 1e-5f, tokens, M * P, stream);
    if (tail_image)
        return st_pe_tail_split3(s3, pe_bias, P, tail_image, tail_image_bytes, weights[8], weights[9], weights[10], 1e-5f, tokens, M * P, stream);
    ST_TRY(Gemm(s3, 64, weights[6], ld_f0, s4, 128, M * P, 128, 64).aux0(pe_bias, 128, 0, P).act(ST_ACT_RELU)
               .work(workspace, workspace_floats).run(stream));
    ST_TRY(Gemm(s4, 128, weights[7], 128, tokens, 128, M * P, 128, 128).bias(weights[8])
               .work(workspace, workspace_floats).run(stream));
    return st_layernorm(tokens, 128, weights[9], weights[10], tokens, 128, M * P, 128, 1e-5f, stream);
}

// GMA Attention.forward (gma.py:54-76), heads = 1, dim_head = 128: attn[b] = softmax(scale * q k^T) with
// [q | k] = inp . Wqk^T.  qk: scratch [B*N, 256]; attn: [B, N, N].
int st_gma_attention(const float* inp, int32_t ld_inp, const float* w_qk, float* qk, float* attn, int32_t B, int32_t N,
                     void* workspace, int64_t workspace_floats, void* stream) {
    if (!inp || !w_qk || !qk || !attn || B <= 0 || N <= 0 || N > 4096) return ST_EINVAL;
    ST_TRY(Gemm(inp, ld_inp, w_qk, 128, qk, 256, B * N, 256, 128).work(workspace, workspace_floats).run(stream));
    ST_TRY(Gemm(qk, 256, qk + 128, 256, attn, N, N, N, 128).alpha(0.08838834764831845f /* 128^-0.5 */)
               .batched(B, (int64_t)N * 256, (int64_t)N * 256, (int64_t)N * N).run(stream));
    return st_softmax_rows(attn, N, B * N, N, stream);
}

// GMA Aggregate.forward (gma.py:102-115), heads = 1: out = mf + gamma * (attn @ (mf . Wv^T)).
// v is produced transposed (vT[b] = Wv . mf[b]^T, [128, N]) so that attn @ v is again an  A . W^T  contraction.
// mf/out: rows [B*N, ld] (column slices of the GRU input buffer); vT: scratch [B, 128, N].
static int gma_aggregate_impl(const float* attn, const float* mf, int32_t ld_mf, const float* w_v, const float* gamma, float* vT,
                              float* out, int32_t ld_out, void* out_planes, int64_t out_pstride, int64_t out_prows, int32_t out_col, int32_t B,
                              int32_t N, void* workspace, int64_t workspace_floats, void* stream);
int st_gma_aggregate(const float* attn, const float* mf, int32_t ld_mf, const float* w_v, const float* gamma, float* vT,
                     float* out, int32_t ld_out, int32_t B, int32_t N, void* workspace, int64_t workspace_floats,
                     void* stream) {
    return gma_aggregate_impl(attn, mf, ld_mf, w_v, gamma, vT, out, ld_out, nullptr, 0, 0, 0, B, N, workspace, workspace_floats, stream);
}
// the same, the result ALSO emitted as blocked bf16 planes (columns out_col .. out_col+127 of out_planes) for a split3 consumer
int st_gma_aggregate_planes(const float* attn, const float* mf, int32_t ld_mf, const float* w_v, const float* gamma, float* vT,
                            float* out, int32_t ld_out, void* out_planes, int64_t out_pstride, int64_t out_prows, int32_t out_col, int32_t B,
                            int32_t N, void* workspace, int64_t workspace_floats, void* stream) {
    if (!out_planes) return ST_EINVAL;
    return gma_aggregate_impl(attn, mf, ld_mf, w_v, gamma, vT, out, ld_out, out_planes, out_pstride, out_prows, out_col, B, N, workspace, workspace_floats, stream);
}
static int gma_aggregate_impl(const float* attn, const float* mf, int32_t ld_mf, const float* w_v, const float* gamma, float* vT,
                              float* out, int32_t ld_out, void* out_planes, int64_t out_pstride, int64_t out_prows, int32_t out_col, int32_t B,
                              int32_t N, void* workspace, int64_t workspace_floats, void* stream) {
    if (!attn || !mf || !w_v || !gamma || !vT || !out || B <= 0 || N <= 0) return ST_EINVAL;
    // both products run as batched launches over b (grid.z): vT[b] = Wv . mf[b]^T, then out[b] = mf[b] + gamma attn[b] vT[b]^T
    ST_TRY(Gemm(w_v, 128, mf, ld_mf, vT, N, 128, N, 128).batched(B, 0, (int64_t)N * ld_mf, (int64_t)128 * N).run(stream));
    {
        Gemm g(attn, N, vT, N, out, ld_out, N, 128, N);
        g.epi(ST_EPI_AXPY, mf, ld_mf).scale(gamma).batched(B, (int64_t)N * N, (int64_t)128 * N, (int64_t)N * ld_out);
        g.d.batch_stride_aux1 = (int64_t)N * ld_mf;
        if (out_planes) {
            g.d.c_planes = out_planes; g.d.c_plane_stride = out_pstride; g.d.c_plane_rows = out_prows; g.d.c_plane_col0 = out_col;
            g.d.c_plane_batch_rows = N;
        }
        if (B == 1) g.work(workspace, workspace_floats);        // a single map cannot fill the chip without split-K
        ST_TRY(g.run(stream));
    }
    return ST_OK;
}

// SepConvGRU.forward (gru.py:44-59): horizontal (1x5) then vertical (5x1) gated update of h.
//   hxA rows [B*H*W, ld] = [h(128) | x(ld-128)], hxB rows [B*H*W, ld] = [r*h scratch(128) | unused]  (x = motion features;
//   the q conv reads its first 128 input channels from hxB and the rest from hxA); the constant `inp` channels of the reference's hx are folded into the per-pass tables
//   tab1/tab2 [B*H*W, 384] = conv_inp([z|r|q]) + bias.  Weights: w_zr* [256, 5*ld], w_q* [128, 5*ld], K ordered (tap, c).
//   z = sigmoid(convz(hx)), r = sigmoid(convr(hx)), q = tanh(convq([r*h, x])), h = (1-z) h + z q.
int st_sepconv_gru(float* hxA, float* hxB, int32_t ld, float* zbuf, const float* tab1, const float* tab2, int32_t ld_tab,
                   const float* w_zr1, const float* w_q1, const float* w_zr2, const float* w_q2, int32_t B, int32_t H,
                   int32_t W, void* workspace, int64_t workspace_floats, void* stream) {
    if (!hxA || !hxB || !zbuf || !tab1 || !tab2 || !w_zr1 || !w_q1 || !w_zr2 || !w_q2 || ld < 128 || ld_tab < 384 ||
        B <= 0 || H <= 0 || W <= 0)
        return ST_EINVAL;
    const float* tabs[2] = {tab1, tab2};
    const float* wzr[2] = {w_zr1, w_zr2};
    const float* wq[2] = {w_q1, w_q2};
    for (int p = 0; p < 2; ++p) {
        const int kh = p ? 5 : 1, kw = p ? 1 : 5, ph = p ? 2 : 0, pw = p ? 0 : 2;
        ST_TRY(Gemm(hxA, ld, wzr[p], 5 * ld, zbuf, 128, 0, 256, ld).conv(B, H, W, kh, kw, 1, 1, ph, pw)
                   .aux0(tabs[p], ld_tab).act(ST_ACT_SIGMOID).epi(ST_EPI_ZR, hxA, ld).out2(hxB, ld)
                   .work(workspace, workspace_floats).run(stream));
        ST_TRY(Gemm(hxA, ld, wq[p], 5 * ld, hxA, ld, 0, 128, ld).a2(hxB, 128).conv(B, H, W, kh, kw, 1, 1, ph, pw)
                   .aux0(tabs[p] + 256, ld_tab).act(ST_ACT_TANH).epi(ST_EPI_GRU, zbuf, 128, hxA, ld)
                   .work(workspace, workspace_floats).run(stream));
    }
    return ST_OK;
}

// ---- the same two operators on exact-split operands (st_gemm_desc.split3, csrc/gemm_split3.h) ----------------------------------
// Plane tensors: three blocked bf16 planes [C / 32][rows][32], `pstride` ELEMENTS apart, `prows` = rows per chunk.

// st_gma_aggregate with attn and v^T as planes.  attn_planes: [3][N/32][B*N][32] (st_split3_pack of the attention matrix, once per pass);
// vT_planes: scratch [3][N/32][B*128][32], written by the v projection's epilogue; out (mf + gamma attn v) leaves as fp32 AND as columns
// out_col .. out_col+127 of out_planes (the GRU input's planes).  Matches gma.py:102-115.
int st_gma_aggregate_split3(const void* attn_planes, int64_t attn_pstride, const float* mf, int32_t ld_mf, const float* w_v, const float* gamma,
                            float* vT, void* vT_planes, int64_t vT_pstride, float* out, int32_t ld_out, void* out_planes, int64_t out_pstride,
                            int64_t out_prows, int32_t out_col, int32_t B, int32_t N, void* workspace, int64_t workspace_floats, void* stream) {
    if (!attn_planes || !mf || !w_v || !gamma || !vT || !vT_planes || !out || !out_planes || B <= 0 || N <= 0 || (N & 31)) return ST_EINVAL;
    {
        // vT[b] = Wv . mf[b]^T  (K = 128: the exact fp32 kernel), emitted as planes: rows b*128 .. b*128+127 of every pixel chunk
        Gemm g(w_v, 128, mf, ld_mf, vT, N, 128, N, 128);
        g.batched(B, 0, (int64_t)N * ld_mf, (int64_t)128 * N);
        g.d.c_planes = vT_planes; g.d.c_plane_stride = vT_pstride; g.d.c_plane_rows = (int64_t)B * 128; g.d.c_plane_batch_rows = 128;
        ST_TRY(g.run(stream));
    }
    {
        Gemm g((const float*)attn_planes, N, (const float*)vT_planes, N, out, ld_out, N, 128, N);
        g.epi(ST_EPI_AXPY, mf, ld_mf).scale(gamma).batched(B, (int64_t)N * 32, (int64_t)128 * 32, (int64_t)N * ld_out);
        g.d.batch_stride_aux1 = (int64_t)N * ld_mf;
        g.d.split3 = 1;
        g.d.a_plane_stride = attn_pstride; g.d.a_rows = (int64_t)B * N;
        g.d.w_plane_stride = vT_pstride; g.d.w_rows = (int64_t)B * 128;
        g.d.c_planes = out_planes; g.d.c_plane_stride = out_pstride; g.d.c_plane_rows = out_prows; g.d.c_plane_col0 = out_col;
        g.d.c_plane_batch_rows = N;
        if (B == 1) g.work(workspace, workspace_floats);
        ST_TRY(g.run(stream));
    }
    return ST_OK;
}

// st_sepconv_gru on planes.  hxA_planes / hxB_planes: [3][ld/32][prows][32] images of hxA = [h | x] and hxB = [r*h | unused] (same strides);
// the kernels read ONLY the planes as contraction operands, the fp32 hxA[:, :128] (h) stays the epilogue operand and receives the new
// state; r*h exists only as planes.  Weights: planes of w_zr* [256, 5*ld] / w_q* [128, 5*ld] (st_split3_pack at load time), all
// `w_pstride_zr` / `w_pstride_q` apart.  Matches gru.py:44-59.
int st_sepconv_gru_split3(float* hxA, int32_t ld, void* hxA_planes, void* hxB_planes, int64_t pstride, int64_t prows, float* zbuf,
                          const float* tab1, const float* tab2, int32_t ld_tab, const void* w_zr1, const void* w_q1, const void* w_zr2,
                          const void* w_q2, int64_t w_pstride_zr, int64_t w_pstride_q, int32_t B, int32_t H, int32_t W, void* workspace,
                          int64_t workspace_floats, void* stream) {
    if (!hxA || !hxA_planes || !hxB_planes || !zbuf || !tab1 || !tab2 || !w_zr1 || !w_q1 || !w_zr2 || !w_q2 || ld < 128 || (ld & 31) ||
        ld_tab < 384 || B <= 0 || H <= 0 || W <= 0 || prows < (int64_t)B * H * W)
        return ST_EINVAL;
    const float* tabs[2] = {tab1, tab2};
    const void* wzr[2] = {w_zr1, w_zr2};
    const void* wq[2] = {w_q1, w_q2};
    for (int p = 0; p < 2; ++p) {
        const int kh = p ? 5 : 1, kw = p ? 1 : 5, ph = p ? 2 : 0, pw = p ? 0 : 2;
        {
            // z | r: z -> zbuf (fp32, the blend's operand), r*h -> columns 0..127 of hxB's planes only
            Gemm g((const float*)hxA_planes, ld, (const float*)wzr[p], 5 * ld, zbuf, 128, 0, 256, ld);
            g.conv(B, H, W, kh, kw, 1, 1, ph, pw).aux0(tabs[p], ld_tab).act(ST_ACT_SIGMOID).epi(ST_EPI_ZR, hxA, ld).out2(hxA /* unused: c_no_f32 */, ld)
                .work(workspace, workspace_floats);
            g.d.split3 = 1; g.d.a_plane_stride = pstride; g.d.a_rows = prows; g.d.w_plane_stride = w_pstride_zr; g.d.w_rows = 256;
            g.d.c_planes = hxB_planes; g.d.c_plane_stride = pstride; g.d.c_plane_rows = prows; g.d.c_no_f32 = 1;
            ST_TRY(g.run(stream));
        }
        {
            // q on [r*h (hxB planes) | x (hxA planes)], GRU blend with z and the fp32 h; new h -> hxA fp32 AND columns 0..127 of hxA's planes
            Gemm g((const float*)hxA_planes, ld, (const float*)wq[p], 5 * ld, hxA, ld, 0, 128, ld);
            g.a2((const float*)hxB_planes, 128).conv(B, H, W, kh, kw, 1, 1, ph, pw).aux0(tabs[p] + 256, ld_tab).act(ST_ACT_TANH)
                .epi(ST_EPI_GRU, zbuf, 128, hxA, ld).work(workspace, workspace_floats);
            g.d.split3 = 1; g.d.a_plane_stride = pstride; g.d.a_rows = prows; g.d.w_plane_stride = w_pstride_q; g.d.w_rows = 128;
            g.d.c_planes = hxA_planes; g.d.c_plane_stride = pstride; g.d.c_plane_rows = prows;
            ST_TRY(g.run(stream));
        }
    }
    return ST_OK;
}

}  // extern "C"
