/* libstitch_gfx950.so -- C-ABI of the MI355X-native stitching hot path.
 *
 * The reference (gargatik/Seamless-Through-Breaking..., 100 % Python) has no native interface for
 * this path: every operator below replaces stock PyTorch ops called from
 * core/flowHomoAdpater.py:FlowHomoAdpater.forward (the drop-in boundary).  Each entry
 * point cites the reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'd / torch CUDA tensor .data_ptr()), fp32 unless
 *     stated, 16-byte aligned bases; no allocation, no ownership transfer, no global state
 *   - `stream` is a hipStream_t (0 = default stream); all calls are asynchronous on it
 *   - activations are channels-last: [B, H, W, C] with a row stride `ld*` (floats) >= C, so a
 *     channel slice of a wider buffer is addressable (concatenation = column offset)
 *   - return 0 on success, ST_EINVAL (1001) for a rejected argument, otherwise a hipError_t value
 */
#ifndef STITCH_GFX950_H
#define STITCH_GFX950_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- dense contractions -------------------------------------------------------------------- */
typedef struct st_gemm_desc {
    const float* a;        /* activations, NHWC [B,H,W,ldx] (plain matrix: H=1, W=M, kh=kw=1)      */
    const float* w;        /* weights [N, K] row-major, K ordered (ky, kx, c), row stride ldw       */
    float* c;              /* output [M, ldc], M = B*Ho*Wo                                          */
    const float* bias;     /* [N] or NULL                                                           */
    const float* aux0;     /* pre-activation addend [*, ld_aux0] or NULL; row = (m / aux0_row_div) % aux0_row_mod */
    const float* aux1;     /* epilogue operand [M, ld_aux1] or NULL                                 */
    const float* aux2;     /* epilogue operand [M, ld_aux2] or NULL                                 */
    const float* scale_ptr;/* device scalar for ST_EPI_AXPY or NULL                                 */
    int32_t M, N, K;       /* K = kh*kw*Cin                                                         */
    int32_t H, W, Cin, ldx;
    int32_t kh, kw, sh, sw, ph, pw, Ho, Wo;
    int32_t ldw, ldc, ld_aux0, ld_aux1, ld_aux2;
    int32_t aux0_row_div, aux0_row_mod;   /* 0/1 = identity row mapping                            */
    int32_t act;           /* 0 none, 1 relu, 2 gelu(erf), 3 sigmoid, 4 tanh                        */
    int32_t epi;           /* 0 store, 1 +aux1, 2 *aux1, 3 GRU blend, 4 aux1 + *scale_ptr * v,
                              5 z|r: cols<N/2 -> c, cols>=N/2 -> c2 = v*aux1                        */
    float alpha;           /* v = act(alpha*acc + bias + aux0)                                      */
    int32_t batch;         /* grid.z batches (0/1 = single)                                         */
    int64_t batch_stride_a, batch_stride_w, batch_stride_c;   /* in floats                          */
    int32_t tile_cfg;      /* 0 = auto; register-staged 1: 128x128, 2: 128x64, 3: 64x64, 4: 128x32;
                              LDS-DMA pipelined (Cin % 32 == 0) 12: 128x64, 13: 64x64, 14: 128x32, 15: 64x128 (never auto);
                              row-streaming (plain matrix, K = 64 / 128) 20 */
    int32_t split_k;       /* 0 = auto, 1 = off, >1 = K slices (needs workspace, batch <= 1)        */
    float* workspace;      /* split-K slabs [split_k, M, N] or NULL (then never split)              */
    int64_t workspace_floats;
    float* c2;             /* second output for epi 5 (fused GRU gates): [M, ldc2]                  */
    int32_t ldc2;
    uint32_t a_bytes, w_bytes;  /* filled by the library: extents of A / W for the buffer descriptors  */
    int32_t reserved0;     /* must be 0 (was the selector of an experimental split-bf16 kernel, removed)           */
    int64_t batch_stride_aux1;  /* floats; aux1 of batch z starts at aux1 + z * batch_stride_aux1          */
    int32_t dh, dw;        /* conv dilation (0 / 1 = dense): tap (ky,kx) reads pixel (oy*sh-ph+ky*dh, ox*sw-pw+kx*dw);
                              Ho / Wo are the caller's (PyTorch: H + 2*ph - dh*(kh-1) - 1) / sh + 1)            */
    int32_t a_ln;          /* 1: every row of A is layer-normalised WITHOUT affine, (x - mean) * rstd over its K = Cin
                              entries, before the contraction (nn.LayerNorm in front of a Linear, twins.py:787-790,
                              encoder.py:156-172; the caller folds gamma into W and beta into the bias).  Row-streaming
                              kernel only: plain matrix, K = 64 / 128, batch <= 1; anything else is rejected          */
    float a_ln_eps;
    const float* a2;       /* optional second source of A with the SAME geometry and row stride: input channels c < a2_channels
                              are read from a2, the others from a (SepConvGRU's q conv reads [r*h | x] as r*h from one
                              buffer and x from the other, gru.py:50, without x being copied).  LDS-DMA kernel only
                              (Cin % 32 == 0, a2_channels % 32 == 0, batch <= 1); anything else is rejected                 */
    int32_t a2_channels;
    int32_t reserved1;     /* must be 0 */
    float* c_t;            /* optional TRANSPOSED copy of the result: c_t[n, m] (row stride ld_ct, batch stride batch_stride_c) receives
                              what c[m, n] receives.  ST_EPI_STORE on the LDS-DMA kernels only, M % 4 == 0, ld_ct % 4 == 0, no split-K:
                              the all-pairs volume of the reverse flow direction is the transpose of the forward one
                              (encoder.py:359-369 evaluated for (f2, f1)) and costs a second store instead of a second product */
    int32_t ld_ct;
    int32_t reserved2;     /* must be 0 */
    int32_t split3;        /* 0: a / w are fp32 (everything above).  1: EXACT-SPLIT operands on the bf16 matrix cores (csrc/gemm_split3.h):
                              a, w (and a2) point to THREE bf16 planes hi, mid, lo with x == hi + mid + lo exactly (st_split3_pack or a
                              producing kernel's epilogue), each blocked by 32-channel chunks: plane = [C / 32][rows][32] bf16, the planes
                              a_plane_stride / w_plane_stride ELEMENTS apart; a_rows = rows per chunk of the A planes (all pixels B*H*W of the
                              activation, >= what the geometry addresses), w_rows = rows per chunk of the W planes (>= N; W chunks follow the
                              K order (ky, kx, c / 32)).  ldx / ldw are ignored; batch strides are bf16 elements inside each plane.  The six
                              products hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi are accumulated in fp32 (dropped terms <= 2^-23 |a||b|);
                              epilogue, split-K and outputs exactly as the fp32 kernels.  Cin % 32 == 0; tile_cfg 0 = auto, 31: 128x128,
                              32: 128x64, 33: 64x128, 34: 64x64, 37: 64x64 PERSISTENT walk over the tiles (many short tiles; the only one that
                              takes c_t); a_ln is rejected.                                                 */
    int32_t reserved3;     /* must be 0 */
    int64_t a_plane_stride, w_plane_stride, a_rows, w_rows;
    void* c_planes;        /* optional (any kernel of the family, fp32 or split3 operands): the result ALSO leaves as three blocked bf16 planes
                              (the operand format of a split3 consumer), written by the epilogue that produced it -- no separate pass, no split
                              work in any K loop.  Element (m, n) goes to plane p at
                                  c_planes[p * c_plane_stride + (((c_plane_col0 + n) / 32) * c_plane_rows + c_plane_row0 + z * c_plane_batch_rows + m) * 32
                                           + (c_plane_col0 + n) % 32]            (z = batch index)
                              In ST_EPI_ZR mode the planes receive the SECOND output (c2 = r*h, column n - N/2).  M % 32 == 0 and
                              c_plane_col0 % 32 == 0 are required; c_no_f32 = 1 skips the fp32 store of that output (its c / c2 pointer must
                              still be valid: split-K partial sums and the z half are unaffected).                                        */
    int64_t c_plane_stride, c_plane_rows, c_plane_batch_rows;
    int32_t c_plane_col0, c_plane_row0, c_no_f32, reserved4;
} st_gemm_desc;

/* fp32 MFMA implicit GEMM: nn.Linear / F.conv2d / einsum on the path, e.g.
 *   core/FlowFormer/PerCostFormer3/encoder.py:359-369 (corr), :60-95 (PatchEmbed convs),
 *   gru.py:44-59,246-254,5-13 (SepConvGRU, motion encoder, heads), gma.py:54-76,102-115,
 *   twins.py:253-392,587-680 (q/k/v/proj/sr/MLP), core/UDIS2/Homography/network.py:18-46,103-137. */
int st_conv_gemm(const st_gemm_desc* desc, void* stream);

/* fp32 rows [rows, ldx] (C columns, C % 32 == 0) -> the three blocked bf16 planes st_gemm_desc.split3 consumes:
 * planes[p * plane_stride + ((c / 32) * chunk_rows + row) * 32 + c % 32], p = 0 (hi), 1 (mid), 2 (lo); x == hi + mid + lo exactly
 * (+-inf / NaN stay in hi alone).  Used once per weight at pack time and for activations no split3-emitting kernel produced.
 * Operands of: gru.py:44-59,246-254, gma.py:102-115, encoder.py:359-369.                                                   */
/* All-pairs volume(s) from the feature maps' PLANES (st_split3_pack of f1 / f2 rows [B*N, C]; sample b = rows b*N ..): vol12[b] = f1[b] . f2[b]^T and, when vol21
 * is not NULL, its transpose from the same launch -- the split3 form of st_corr_volume / st_corr_volume_both on the persistent 64x64 kernel
 * (tile_cfg 37: a workgroup walks its tiles with one continuous DMA ring).  encoder.py:359-369.                                       */
int st_corr_volume_split3(const void* f1_planes, const void* f2_planes, int64_t pstride, int64_t prows, float* vol12, float* vol21, int32_t B,
                          int32_t N, int32_t C, void* stream);
int st_split3_pack(const float* x, void* planes, int64_t rows, int32_t C, int64_t ldx, int64_t plane_stride, int64_t chunk_rows, void* stream);

/* Two independent contractions in ONE launch: workgroups of both descriptors share the grid, so two mid-size convs that are
 * ready together (BasicMotionEncoder's convc2 and convf2, gru.py:252-253) fill the chip without split-K slabs or a second
 * launch.  Both must be LDS-DMA-kernel shapes (Cin % 32 == 0, K >= 128), batch <= 1, no split-K, no a_ln; results are
 * bit-identical to two st_conv_gemm calls.                                                                            */
int st_conv_gemm_pair(const st_gemm_desc* desc0, const st_gemm_desc* desc1, void* stream);

/* Profiling observer, off by default: `callback` is a
 *   void (*)(const st_gemm_desc*, void* stream, int32_t phase, void* user)
 * invoked on the launching thread before (phase 0) and after (phase 1) every st_conv_gemm enqueues its kernels
 * (also for the GEMMs launched by the operator-level entry points below), so a caller can bracket them with HIP
 * events on `stream`.  NULL switches it off.  Process-wide; do not change it while other threads launch.   */
int st_set_gemm_observer(void* callback, void* user);

/* Profiling aid: the launch plan the library chose for the calling thread's most recent st_conv_gemm:
 *   plan4[0] kernel family (0 skinny_gemm, 1 narrow_conv, 2 conv_gemm_kernel [register-staged], 3 conv_gemm_dma_kernel,
 *            4 rowstream_gemm_kernel, 5 rowchain128_kernel [st_linear_chain128, reported to the observer as M x 128L x 128],
 *            6 rowmlp128_kernel [st_mlp128, reported as M x 2 hidden x 128], 7 patch_c0c2_kernel [st_patch_conv12], 8 conv_gemm_split3_kernel
 *            [st_gemm_desc.split3: tile_cfg 31..37], 9 rowmlp128_split3_kernel [st_mlp128_split3], 10 rowlin128_split3_kernel [st_rowlin128_split3], 11 pe_tail_split3_kernel [st_pe_tail_split3])
 *   plan4[1] tile_cfg actually used, plan4[2] split_k actually used, plan4[3] 1 = persistent M walk, 2 / 3 = first / second
 *            member of an st_conv_gemm_pair launch (one dispatch, reported with the second member).
 * Used by tools/gemm_shapes_csv.py to label every launch of a step (profiles/r2_gemm_shapes.csv). */
int st_gemm_last_plan(int32_t* plan4);

/* sizeof(st_gemm_desc) as compiled into the library (binding self-check; returns the size). */
int st_abi_gemm_desc_size(void);

/* A chain of up to 3 Linear(128 -> 128) layers over the rows of a matrix, evaluated without the intermediate activations
 * leaving the CU (the tail of the latent layers: `proj + residual -> LayerNorm -> ffn.0 + GELU -> ffn.3 + residual`,
 * crossattentionlayer.py:46-56, encoder.py:163-172):
 *   x_0 = a;   x_{l+1} = act_l( LN_l(x_l) . w_l^T + bias_l ) + residual_l ;   out = x_nlayers
 * LN_l (ln = 1): layer norm of the row WITHOUT affine (the caller folds gamma / beta into w_l / bias_l);
 * residual_l: res = 0 none, 1 = rows of the global matrix res_ptr (row stride ld_res), 2 = x_{res_layer} (an earlier
 * layer's input before its LN).  w_l [128,128] row-major contiguous, bias_l [128], both 16-byte aligned.  Same k pairing and
 * summation order per layer as st_conv_gemm (bit-identical to the unfused chain).  The kernel keeps ONE saved layer input:
 * all res = 2 layers of a chain must name the same res_layer (ST_EINVAL otherwise).                                      */
typedef struct st_chain_layer {
    const float* w;
    const float* bias;
    const float* res_ptr;
    int32_t ld_res;
    int32_t act;           /* 0 none, 1 relu, 2 gelu                                                                      */
    int32_t ln;
    float ln_eps;
    int32_t res;
    int32_t res_layer;
} st_chain_layer;
typedef struct st_chain_desc {
    const float* a;        /* [M, lda], 128 columns used                                                                  */
    float* out;            /* [M, ldo]                                                                                    */
    int32_t lda, ldo, M, nlayers;
    st_chain_layer layer[3];
} st_chain_desc;
int st_linear_chain128(const st_chain_desc* desc, void* stream);
int st_abi_chain_desc_size(void);

/* The Twins / vertical-layer MLP of C = 128 rows (timm Mlp inside Block, twins.py:785-790; encoder.py:121-125):
 *   out = a + ( GELU( LN(a) . w1^T + b1 ) . w2^T + b2 ) [+ res]
 * as one launch (rowmlp128_kernel): the [M, hidden] activations never reach HBM.  LN (ln = 1): layer norm WITHOUT affine, the
 * caller folds gamma / beta into w1 / b1.  w1 [hidden, 128], w2 [128, hidden] row-major contiguous; b1 [hidden], b2 [128];
 * hidden % 32 == 0, 32..2048; every pointer 16-byte aligned; out must not alias a.  fc1 + GELU are bit-identical to
 * st_conv_gemm(act = gelu); fc2 sums its `hidden` products in one chain (st_conv_gemm folds every 256 k): same products, the
 * last bits of the sum differ.  plan4[0] = 6; reported to the observer as M x (2 hidden [+ 128 with wp]) x 128.                 */
typedef struct st_mlp_desc {
    const float* a;        /* [M, lda], 128 columns used: input and first residual                                         */
    float* out;            /* [M, ldo]                                                                                    */
    const float* w1;
    const float* b1;
    const float* w2;
    const float* b2;
    const float* res;      /* optional second residual [M, ld_res] (encoder.py:281 short-cut), or NULL                    */
    int32_t lda, ldo, ld_res, M, hidden, ln;
    float ln_eps;
    int32_t reserved;      /* must be 0                                                                                   */
    /* optional leading layer, the Block's attention output projection + residual (twins.py:622-623, 676-677, 790):
     *   x = a . wp^T + bp + res0   takes a's place above (a is then the attention output, x never reaches HBM as a tensor of
     * its own: it is parked in the block's rows of `out` and re-read as the MLP's residual).  wp [128,128]; NULL = absent.   */
    const float* wp;
    const float* bp;
    const float* res0;     /* [M, ld_res0] or NULL; must not alias out                                                    */
    int32_t ld_res0, reserved1;
} st_mlp_desc;
int st_mlp128(const st_mlp_desc* desc, void* stream);
int st_abi_mlp_desc_size(void);
/* st_mlp128 on the exact-split contraction (round 6, csrc/mlp_split3.h): every fp32 product of the three GEMMs as six bf16 MFMA
 * products of operand planes (hi / mid / lo), fp32 accumulation; same operator, same `desc`; accuracy against fp64 that of the fp32 MFMA
 * chain or better (tests/test_split3_gpu.py); a non-finite activation gives NaN in its row.  The weights are packed ONCE into an image
 * (one 49-KiB LDS stage per step of the kernel's walk: [wp chunk]* then [w1 chunk | w2 slice | b1 chunk]*):
 *   st_mlp128_split3_image_bytes(hidden, with_proj, &bytes)  size of the image
 *   st_mlp128_split3_pack(w1, b1, w2, wp, bp, hidden, image, image_bytes, stream)   wp / bp NULL = no projection
 *   st_mlp128_split3(desc, image, image_bytes, stream)   desc->w1 / b1 / w2 / bp are not read; desc->wp != NULL <=> the image holds a projection.
 * plan4[0] = 9; reported to the observer like st_mlp128 with split3 = 1.                                                            */
int st_mlp128_split3_image_bytes(int32_t hidden, int32_t with_proj, int64_t* bytes);
int st_mlp128_split3_pack(const float* w1, const float* b1, const float* w2, const float* wp, const float* bp, int32_t hidden, void* image,
                          int64_t image_bytes, void* stream);
int st_mlp128_split3(const st_mlp_desc* desc, const void* image, int64_t image_bytes, void* stream);
/* out[M, N] = LayerNorm(a)[M, 128] . w^T + b on the exact-split contraction (round 6, csrc/mlp_split3.h, rowlin128_split3_kernel): the q | k | v
 * projections behind norm1 (twins.py:598-600, encoder.py:156-160), i.e. st_conv_gemm with a_ln for K = 128.  ln = 1: layer norm without affine
 * (gamma / beta folded into w / b by the caller), ln = 0: plain rows.  w [N, 128] row-major, b [N] or NULL, N % 32 == 0; weights packed once:
 *   st_rowlin128_split3_image_bytes(N, &bytes), st_rowlin128_split3_pack(w, b, N, image, image_bytes, stream).  aux (or NULL): a table
 * [ceil(M / row_div), ld_aux] whose row (m / row_div) is added to output row m (st_gemm_desc.aux0 with row_div).  plan4[0] = 10.      */
int st_rowlin128_split3_image_bytes(int32_t N, int64_t* bytes);
int st_rowlin128_split3_pack(const float* w, const float* b, int32_t N, void* image, int64_t image_bytes, void* stream);
int st_rowlin128_split3(const float* a, int32_t lda, float* out, int32_t ldo, int32_t M, int32_t N, int32_t ln, float ln_eps, const void* image,
                        int64_t image_bytes, const float* aux, int32_t ld_aux, int32_t row_div, void* stream);

/* All-pairs correlation volume, MemoryEncoder.corr (encoder.py:359-369):
 *   f1, f2 [B, N, C] channels-last features -> vol [B, N1, N2] = f1 . f2^T (no scaling). */
int st_corr_volume(const float* f1, const float* f2, float* vol, int32_t B, int32_t N1, int32_t N2,
                   int32_t C, void* stream);
/* Both directions at once: vol12 [B, N, N] = f1 . f2^T and vol21 [B, N, N] = f2 . f1^T = vol12^T (bit for bit: products
 * commute, same k order), written by the same launch through a transposed second store (N % 4 == 0, C % 32 == 0, C >= 128;
 * other shapes run as two products).                                                                                  */
int st_corr_volume_both(const float* f1, const float* f2, float* vol12, float* vol21, int32_t B, int32_t N, int32_t C,
                        void* stream);
/* The dispatch decision of st_corr_volume_both, without launching (host only): returns 1 = one product + transposed second store,
 * 0 = two st_corr_volume products (shape / alignment of the LDS-DMA kernel not met, or N * N * 4 >= 2^31: the transposed copy
 * and the row-chunked path exclude each other).  aligned16: all of f1, f2, vol21 are 16-byte aligned.                     */
int st_corr_volume_both_plan(int32_t B, int32_t N, int32_t C, int32_t aligned16);

/* ---- row-wise network ops (channels-last rows) ------------------------------------------------ */
/* nn.LayerNorm over the last dim (encoder.py:58,140-141; twins.py:752-790, eps 1e-5 / 1e-6).        */
int st_layernorm(const float* x, int32_t ldx, const float* w, const float* b, float* out, int32_t ldo,
                 int32_t rows, int32_t C, float eps, void* stream);
/* in-place softmax over rows of length C <= 4096 (gma.py:72 `sim.softmax(dim=-1)`).                */
int st_softmax_rows(float* x, int32_t ld, int32_t rows, int32_t C, void* stream);
/* F.normalize(p=2, dim=C) on NHWC rows (core/UDIS2/Homography/network.py:150-151).                  */
int st_l2norm_rows(const float* x, float* out, int32_t rows, int32_t C, void* stream);
/* nn.MaxPool2d on NHWC (network.py:22,28,34; torchvision resnet maxpool 3/2/1).                     */
int st_maxpool_nhwc(const float* x, float* out, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                    int32_t s, int32_t p, void* stream);
/* PEG PosConv: depthwise 3x3 + bias + identity (twins.py:793-808); w [9, C].                        */
int st_dwconv3x3_residual(const float* x, const float* w, const float* bias, float* out, int32_t B,
                          int32_t H, int32_t W, int32_t C, void* stream);
/* LinearPositionEmbeddingSine (attention.py:156-161); coords [rows, ldc]=(x,y) or implicit grid
 * (r = row % period; r -> (r % Wg, r / Wg), optionally modulo ws), value*cscale + coff; accumulate: 0 = write,
 * 1 = add to what is there, n > 1 = write columns < n and add to columns >= n.                      */
int st_sine_pe(float* out, int32_t ld, int32_t rows, int32_t dim, const float* coords, int32_t ldc,
               int32_t Wg, int32_t ws, int32_t period, float cscale, float coff, int32_t accumulate,
               void* stream);
/* Multi-head softmax attention, element (b, t, h, e) at base + b*bs + t*ts + h*D + e (floats).
 *   _small : one thread per query, K/V from L2 (attention.py:9-68; 8 latents / 1x8 decoder query)
 *   _kvlds : K/V slab (Nk <= 256) staged in LDS (GSA: twins.py:336-392,633-680)                     */
int st_attention_small(const float* q, int64_t q_bs, int64_t q_ts, const float* k, int64_t k_bs, int64_t k_ts,
                       const float* v, int64_t v_bs, int64_t v_ts, float* out, int64_t o_bs, int64_t o_ts,
                       int32_t B, int32_t heads, int32_t Nq, int32_t Nk, int32_t D, float scale, void* stream);
int st_attention_kvlds(const float* q, int64_t q_bs, int64_t q_ts, const float* k, int64_t k_bs, int64_t k_ts,
                       const float* v, int64_t v_bs, int64_t v_ts, float* out, int64_t o_bs, int64_t o_ts,
                       int32_t B, int32_t heads, int32_t Nq, int32_t Nk, int32_t D, float scale, void* stream);
/* 7x7-window attention (LSA: twins.py:253-304,587-631); q/k/v pad tables [ws*ws, heads*D] hold the
 * projections of a zero-padded token at each window position.                                      */
int st_window_attention(const float* q, const float* k, const float* v, int64_t bs, int64_t ts,
                        const float* qpad, const float* kpad, const float* vpad, float* out, int64_t o_bs,
                        int64_t o_ts, int32_t B, int32_t H, int32_t W, int32_t heads, int32_t D, int32_t ws,
                        float scale, void* stream);
/* CCL: 3x3 patch correlation + softmax(10 x) + soft-argmax (network.py:147-199) from the all-pairs
 * product G [B, P, P] of the normalised features; out [B, P, ldo] = (flow_w, flow_h, 0...).         */
int st_ccl_softargmax(const float* G, float* out, int32_t ldo, int32_t B, int32_t h, int32_t w, void* stream);
/* PatchEmbed's first conv, Conv2d(1,16,k6,s2,p2)+ReLU on single-channel cost maps (encoder.py:36-37);
 * maps [M,H,W], w [36,16] (tap-major), out [M*Ho*Wo,16]; reads beyond H/W are zero (right/bottom pad). */
int st_patch_conv1(const float* maps, const float* w36x16, const float* bias, float* out, int32_t M, int32_t H,
                   int32_t W, int32_t Ho, int32_t Wo, void* stream);
/* PatchEmbed.proj[0..3] per cost map in ONE launch (encoder.py:36-39,68-72: Conv2d(1,16,6,2,2) + ReLU -> Conv2d(16,32,6,2,2) + ReLU):
 * cost_maps [M, 64*64] -> s2 rows [M*16*16, 32] (channels last), the first feature map never leaving the CU; bit-identical to
 * st_patch_conv1 followed by the (6x3 pixel-pair) st_conv_gemm of st_patch_embed.  H = W = 64 only (ST_EINVAL otherwise: callers take
 * the unfused launches).                                                                              */
int st_patch_conv12(const float* cost_maps, const float* c0_w36x16, const float* c0_b, const float* c2_w32x576,
                    const float* c2_b, float* s2, int32_t M, int32_t H, int32_t W, void* stream);
/* st_patch_conv12 whose result leaves as three blocked bf16 planes [1 chunk][M*256 rows][32] (s2_pstride elements apart) instead of fp32 rows:
 * the operand of the split3 form of PatchEmbed's third convolution (st_patch_embed_split3).  encoder.py:60-75.                    */
int st_patch_conv12_planes(const float* cost_maps, const float* c0_w36x16, const float* c0_b, const float* c2_w32x576, const float* c2_b,
                           void* s2_planes, int64_t s2_pstride, int32_t M, int32_t H, int32_t W, void* stream);
int st_copy2d(const float* src, int32_t lds, float* dst, int32_t ldd, int32_t rows, int32_t cols, void* stream);
/* NCHW image -> channels-last rows with v = mul*(x/div) - sub (flowHomoAdpater.py:55-56,
 * transformer.py:53-54); channels C..ldo-1 are zero.                                               */
int st_prep_image(const float* src, float* dst, int32_t B, int32_t C, int32_t H, int32_t W, int32_t ldo,
                  float mul, float div, float sub, void* stream);

/* Latent cross-attention with the k / v projections folded out (crossattentionlayer.py:37-56, attention.py:9-68; the 8
 * latent queries do not depend on the pixel): per pixel, softmax over its P tokens of the 64 (latent, head) score rows
 * and the pooling z = softmax(S)^T . T.  scores [pixels*P, ld_s>=64] (col = latent*8 + head), tokens [pixels*P, ld_t>=128],
 * z [pixels*64, 128].  P even, <= 64.                                                                        */
int st_latent_pool(const float* scores, int32_t ld_s, const float* tokens, int32_t ld_t, float* z, int32_t pixels,
                   int32_t P, void* stream);

/* ---- FlowFormer decoder gathers --------------------------------------------------------------- */
int st_coords_grid(float* out, int32_t B, int32_t H, int32_t W, void* stream);          /* decoder.py:22-29 */
int st_flow_from_coords(const float* coords1, float* flow4, int32_t ld4, float* dst2, int32_t ld2, int32_t B,
                        int32_t H, int32_t W, void* stream);
/* BasicMotionEncoder flow branch, first layer, fused with the flow computation (gru.py:251, decoder.py:321):
 * out [B*H*W, ldo] (first Co columns) = relu(Conv2d(2, Co, 7, padding=3)(coords1 - coords0)); w98 [49 taps][2][Co]
 * (tap-major: w98[(ky*7+kx)*2 + c][co] = weight[co][c][ky][kx]); flow2 (optional) receives the flow itself in
 * columns 0..1 of rows of stride ld2 (gru.py:254 cat([out, flow])).  Co % 4 == 0.                                 */
int st_flow_encode(const float* coords1, const float* w98, const float* bias, float* out, int32_t ldo, float* flow2,
                   int32_t ld2, int32_t B, int32_t H, int32_t W, int32_t Co, void* stream);                             /* decoder.py:321   */
/* The same, ALSO leaving both results as blocked bf16 planes (st_gemm_desc.split3 operand format, planes `*_pstride` elements apart,
 * `*_prows` rows per 32-channel chunk): out_planes [3][Co/32][out_prows][32]; channels flow_col, flow_col + 1 (flow_col even) of flow_planes
 * (may be NULL).  Co % 32 == 0.  Replaces gru.py:251,254 for a split3 consumer of gru.py:252-253 / 44-59.                          */
int st_flow_encode_split3(const float* coords1, const float* w98, const float* bias, float* out, int32_t ldo, float* flow2,
                          int32_t ld2, int32_t B, int32_t H, int32_t W, int32_t Co, void* out_planes, int64_t out_pstride,
                          int64_t out_prows, void* flow_planes, int64_t flow_pstride, int64_t flow_prows, int32_t flow_col,
                          void* stream);
/* encode_flow_token + bilinear_sampler (decoder.py:242-260, core/utils/utils.py:62-76).              */
int st_cost_lookup(const float* maps, const float* coords, float* out, int32_t ldo, int32_t Nq, int32_t H2,
                   int32_t W2, int32_t r, void* stream);
/* One fused launch for the row-local token chain of a refinement iteration: flow_token_encoder
 * (decoder.py:148-152,305) + decoder CrossAttentionLayer (decoder.py:62-109: LN, sine PE of coords1, q, 8-head
 * attention over the pixel's ntok<=8 memory tokens kv [rows*ntok,128]=(k|v), proj + short-cut, FFN).
 * corr [rows, ld>=148]: reads cols 0..83 (cost_forward, 81 + 3 zero), writes cols 84..147 (cost_global).
 * weights16 (host array of 16 device pointers): w0[64,84] b0 w2[64,64] b2 n1w n1b wq bq wp bp n2w n2b wf0 bf0 wf3 bf3. */
int st_decoder_token_chain(float* corr, int32_t ld_corr, const float* coords1, const float* kv,
                           const float* const* weights16, int32_t rows, int32_t ntok, void* stream);
/* upsample_flow (decoder.py:214-225): mask [B*H*W, ldm>=576] -> out NCHW [B,2,8H,8W].              */
int st_convex_upsample(const float* coords1, const float* mask, int32_t ldm, float* out, int32_t B, int32_t H,
                       int32_t W, void* stream);

/* ---- geometric stage (NCHW images; bit-exact integer sample indices) --------------------------- */
/* tensor_DLT (core/udis_utils/torch_DLT.py:17-45): src [4,2] shared corners, dst = src +
 * motion[B,4,2]*(mscale_x, mscale_y), both divided by `div` -> H [B,3,3].                           */
int st_dlt4(const float* src4x2, const float* motion, float* H, int32_t B, float mscale_x, float mscale_y,
            float div, void* stream);
/* out[b] = L @ (invert ? X[b]^-1 : X[b]) @ R (flowHomoAdpater.py:108,112,226,307).                  */
int st_mat3_sandwich(const float* L, const float* X, const float* R, float* out, int32_t B, int32_t invert,
                     void* stream);
/* torch_homo_transform.transformer (core/udis_utils/torch_homo_transform.py:5-151); the last n_ones
 * output channels are the warp of an all-ones image; idx (optional) [B,oh,ow,4] = x0,x1,y0,y1.      */
int st_homo_warp(const float* U, const float* theta, float* out, int32_t* idx, int32_t B, int32_t C,
                 int32_t n_ones, int32_t H, int32_t W, int32_t oh, int32_t ow, void* stream);
/* get_rigid_mesh + H2Mesh + min/max (core/warp_utils.py:10-34, flowHomoAdpater.py:254-266).          */
int st_mesh_bounds(const float* H, float* out4, int32_t B, float width, float height, int32_t gw, int32_t gh,
                   void* stream);
/* warp() = grid_sample(bilinear, zeros, align_corners=True) at pix+flow (core/warp_utils.py:54-80). */
int st_flow_warp(const float* x, const float* flow, const float* mul, float* out, int32_t B, int32_t C,
                 int32_t H, int32_t W, void* stream);
/* F.interpolate bilinear: resize_flow (warp_utils.py:38-46) / Resize((512,512)) (flowHomoAdpater.py:14);
 * align_corners == 2: scale_factor form of out.py:281 (half-pixel centres, source step (div0, div1) = 1/scale). */
int st_resize_bilinear(const float* x, float* out, int32_t planes, int32_t H, int32_t W, int32_t oh, int32_t ow,
                       int32_t align_corners, float div0, float div1, int32_t ndiv, void* stream);
/* compute_range_map (core/warp_utils.py:114-175), deterministic fixed-point splat.                  */
int st_range_map(const float* flow, void* scratch_u64, float* out, int32_t B, int32_t H, int32_t W, void* stream);
int st_occlusion_from_range(const float* range, float* out, int64_t n, int32_t threshold, void* stream);
/* preprocess_occlusion_mask (flowHomoAdpater.py:18-35); scratch: 2*N*H*W bytes.                     */
int st_morph_open(const float* mask, float* out, void* scratch_u8x2, int32_t N, int32_t H, int32_t W,
                  int32_t ksz, void* stream);
int st_eval_finish(float* final6, const float* occ, float* overlap, int32_t B, int32_t H, int32_t W, void* stream);
int st_blend(const float* homo1, const float* homo2, float* fin, const float* occ, float* output2, float* mask1,
             float* mask2, uint8_t* blend, int32_t h, int32_t w, void* stream);          /* :339-360 */
int st_mean_threshold(const float* x, float* out, int32_t B, int32_t C, int32_t H, int32_t W, float thr,
                      void* stream);                                                      /* :233-234 */
/* UDIS2 TPS transformer (core/udis_utils/torch_tps_transform.py:7-190): fp64 solve -> T [B,2,N+3],
 * then grid + 4-tap gather.  work_f64: B*(N+3)*(N+5) doubles.                                       */
int st_tps_solve_grid(const float* U, const float* source, const float* target, void* work_f64, float* T,
                      float* out, int32_t* idx, int32_t B, int32_t C, int32_t H, int32_t W, int32_t N,
                      int32_t oh, int32_t ow, void* stream);

/* ---- evaluation metric ------------------------------------------------------- */
/* Masked PSNR / SSIM per image exactly as evaluate.py:53-65 feeds skimage 0.19.3 (uint8 truncation, mask =
 * uint8(channel-mean mask), 7x7 uniform SSIM, K1=.01 K2=.03, sample covariance, 3-px crop, channel mean).
 * image1 [B,3,H,W]; warped: B images of 3 planes, batch stride in floats (e.g. 6*H*W for final_warp_output);
 * maskmean [B,H,W]; partial_f64: 2*B*ceil(3HW/256) doubles of scratch; out [B,2] fp64 = (psnr, ssim). */
int st_masked_psnr_ssim(const float* image1, const float* warped, int64_t warped_batch_stride, const float* maskmean,
                        void* partial_f64, double* out_psnr_ssim, int32_t B, int32_t H, int32_t W, void* stream);
/* mean over C planes of x[b] (evaluate.py:45). */
int st_channel_mean(const float* x, int64_t batch_stride, float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                    void* stream);

/* Loader tail of the harnesses (core/datasets.py:383-386 `torch.from_numpy(img).permute(2,0,1).float()`, out.py:137-143):
 * interleaved uint8 [B,H,W,3] -> planar float32 [B,3,H,W] (exact).  H*W must be a multiple of 4, src 4-byte and dst 16-byte
 * aligned (ST_EINVAL otherwise). */
int st_load_rgb8(const void* src_u8_hwc, float* dst_chw, int32_t B, int32_t H, int32_t W, void* stream);

/* ---- operator-level entry points (one per reference operator; host-side composition of the kernels
 *      above on the caller's stream, caller-provided scratch, no allocation, no state) ------------------ */
/* encode_flow_token with the reference's 9x9 window (decoder.py:242-260).                            */
int st_cost_lookup9x9(const float* maps, const float* coords, float* out, int32_t ldo, int32_t Nq, int32_t H2,
                      int32_t W2, void* stream);
/* warp(x, flow) [* mask] (core/warp_utils.py:54-80, flowHomoAdpater.py:171-172,339-341).             */
int st_grid_sample_blend(const float* x, const float* flow, const float* mul, float* out, int32_t B, int32_t C,
                         int32_t H, int32_t W, void* stream);
/* preprocess_occlusion_mask, 19x19 structuring element (flowHomoAdpater.py:18-35).                   */
int st_morph_open19(const float* mask, float* out, void* scratch_u8x2, int32_t N, int32_t H, int32_t W,
                    void* stream);
/* PatchEmbed.forward (encoder.py:60-95; patch 8, 'single', linear PE): cost maps [M,H,W] -> tokens
 * [M*P,128], P = ceil(H/8)*ceil(W/8).  weights (host array of 11 device pointers): c0_w[36,16] c0_b
 * c2_w[32,576] c2_b c4_w[64,1152] c4_b f0_w[128,ld_f0] (cols 0..63) f2_w[128,128] f2_b ln_w ln_b;
 * pe_bias [P,128] = f0_w[:,64:] . sinePE(pos) + f0_b.  scratch rows: s1 16, s2 32, s3 64, s4 128 wide.
 * 64x64 maps run the first two convs as st_patch_conv12 (s1 stays on the CU): s1 may then be NULL. */
int st_patch_embed(const float* cost_maps, const float* const* weights, int32_t ld_f0, const float* pe_bias,
                   float* s1, float* s2, float* s3, float* s4, float* tokens, int32_t M, int32_t H, int32_t W,
                   void* workspace, int64_t workspace_floats, void* stream);
/* st_patch_embed for 64 x 64 maps with Conv2d(32, 64, 6, 2, 2) -- 77 of the operator's 99 GFLOP per pair -- on exact-split operands
 * (st_gemm_desc.split3): s2_planes scratch [3][1][M*256][32] written by the fused c0 + c2 launch, c4_w_planes = st_split3_pack(c4_w [64, 1152]).
 * M <= 16 384 maps per call (2 GiB buffer offsets).  encoder.py:60-95.                                                              */
int st_patch_embed_split3(const float* cost_maps, const float* const* weights, int32_t ld_f0, const float* pe_bias, void* s2_planes,
                          int64_t s2_pstride, const void* c4_w_planes, int64_t c4_w_pstride, float* s3, float* s4, float* tokens, int32_t M,
                          int32_t H, int32_t W, const void* tail_image, int64_t tail_image_bytes, void* workspace, int64_t workspace_floats, void* stream);
/* PatchEmbed's tail (encoder.py:77-95) as one launch on the exact-split contraction (csrc/mlp_split3.h, pe_tail_split3_kernel):
 *   out[R, 128] = LayerNorm_affine( ReLU( x[R, 64] . w1^T + tab[r mod P] ) . w2^T + b2 )
 * w1 = ffn_with_coord.0's first 64 input columns [128, ld1], tab [P, 128] = its position half + bias (st_patch_embed's pe_bias), w2 [128, 128];
 * both weight matrices are packed once into a 144-KiB image that the kernel holds in LDS (st_pe_tail_split3_image_bytes / _pack).  st_patch_embed_split3
 * takes the image as tail_image (NULL = the three separate launches, s4 [M P, 128] scratch required).  plan4[0] = 11, reported as R x 192 x 128.        */
int st_pe_tail_split3_image_bytes(int64_t* bytes);
int st_pe_tail_split3_pack(const float* w1, int32_t ld1, const float* w2, void* image, int64_t image_bytes, void* stream);
int st_pe_tail_split3(const float* x, const float* tab, int32_t P, const void* image, int64_t image_bytes, const float* b2, const float* gamma,
                      const float* beta, float eps, float* out, int32_t R, void* stream);
/* GMA Attention.forward (gma.py:54-76), 1 head x 128: attn [B,N,N] = softmax(128^-0.5 q k^T),
 * [q|k] = inp . w_qk^T (w_qk [256,128]); qk scratch [B*N,256].                                        */
int st_gma_attention(const float* inp, int32_t ld_inp, const float* w_qk, float* qk, float* attn, int32_t B,
                     int32_t N, void* workspace, int64_t workspace_floats, void* stream);
/* GMA Aggregate.forward (gma.py:102-115): out = mf + gamma * attn @ (mf . w_v^T); vT scratch [B,128,N]. */
int st_gma_aggregate(const float* attn, const float* mf, int32_t ld_mf, const float* w_v, const float* gamma,
                     float* vT, float* out, int32_t ld_out, int32_t B, int32_t N, void* workspace,
                     int64_t workspace_floats, void* stream);
/* SepConvGRU.forward (gru.py:44-59).  hxA rows [h(128) | x(ld-128)], hxB rows of the same stride [r*h scratch | unused]; the constant
 * `inp` channels arrive folded into tab1/tab2 [rows, ld_tab>=384] = conv_inp([z|r|q]) + bias; w_zr* [256,5*ld],
 * w_q* [128,5*ld] with K ordered (tap, channel); zbuf scratch [rows,128].  h is updated in place in hxA.   */
int st_sepconv_gru(float* hxA, float* hxB, int32_t ld, float* zbuf, const float* tab1, const float* tab2,
                   int32_t ld_tab, const float* w_zr1, const float* w_q1, const float* w_zr2, const float* w_q2,
                   int32_t B, int32_t H, int32_t W, void* workspace, int64_t workspace_floats, void* stream);
/* The same two operators on EXACT-SPLIT operands (st_gemm_desc.split3): every contraction operand is three blocked bf16 planes
 * [C/32][prows][32] (`*_pstride` elements apart) that the producing kernels' epilogues wrote, the six partial products run on the bf16
 * matrix cores with fp32 accumulation (error <= the fp32 chain's, tools/split3_probe.py), epilogue operands and results stay fp32.
 * st_gma_aggregate_split3 (gma.py:102-115): attn_planes [3][N/32][B*N][32] = st_split3_pack of attn, once per pass; vT_planes scratch
 * [3][N/32][B*128][32]; the result also goes to columns out_col..out_col+127 of out_planes.
 * st_sepconv_gru_split3 (gru.py:44-59): hxA_planes / hxB_planes image hxA = [h | x] and hxB = [r*h | -] (ld channels, ld % 32 == 0);
 * r*h exists only as planes; the new h goes to hxA (fp32, in place) and to columns 0..127 of hxA_planes; w_* are st_split3_pack
 * images of the fp32 operator's weight matrices.                                                                                   */
/* st_gma_aggregate whose result ALSO leaves as blocked bf16 planes (columns out_col..out_col+127 of out_planes; st_gemm_desc.c_planes): the
 * aggregate reads the whole attention matrix every call and is HBM-bound on either operand format, so it stays on the fp32 kernel while
 * its consumer (st_sepconv_gru_split3) reads planes.  gma.py:102-115.                                                              */
int st_gma_aggregate_planes(const float* attn, const float* mf, int32_t ld_mf, const float* w_v, const float* gamma, float* vT,
                            float* out, int32_t ld_out, void* out_planes, int64_t out_pstride, int64_t out_prows, int32_t out_col, int32_t B,
                            int32_t N, void* workspace, int64_t workspace_floats, void* stream);
int st_gma_aggregate_split3(const void* attn_planes, int64_t attn_pstride, const float* mf, int32_t ld_mf, const float* w_v, const float* gamma,
                            float* vT, void* vT_planes, int64_t vT_pstride, float* out, int32_t ld_out, void* out_planes, int64_t out_pstride,
                            int64_t out_prows, int32_t out_col, int32_t B, int32_t N, void* workspace, int64_t workspace_floats, void* stream);
int st_sepconv_gru_split3(float* hxA, int32_t ld, void* hxA_planes, void* hxB_planes, int64_t pstride, int64_t prows, float* zbuf,
                          const float* tab1, const float* tab2, int32_t ld_tab, const void* w_zr1, const void* w_q1, const void* w_zr2,
                          const void* w_q2, int64_t w_pstride_zr, int64_t w_pstride_q, int32_t B, int32_t H, int32_t W, void* workspace,
                          int64_t workspace_floats, void* stream);

/* ---- UDIS2 composition stage (the convolutions run on st_conv_gemm with dh/dw) ------------ */
/* F.interpolate(mode='nearest') to (oh, ow), channels-last rows, C % 4 == 0 (Composition/network.py:70).   */
int st_resize_nearest_rows(const float* x, int32_t ldx, float* out, int32_t ldo, int32_t B, int32_t H, int32_t W,
                           int32_t C, int32_t oh, int32_t ow, void* stream);
/* out = a - b on [rows, C] row views (network.py:120-124).                                                 */
int st_sub_rows(const float* a, int32_t lda, const float* b, int32_t ldb, float* out, int32_t ldo, int64_t rows,
                int32_t C, void* stream);
/* build_model (network.py:8-22): NCHW [B,3,H,W] images / masks, net_out rows [B*H*W] with stride ld_net.    */
int st_compose_blend(const float* warp1, const float* warp2, const float* mask1, const float* mask2,
                     const float* net_out, int32_t ld_net, float* learned_mask1, float* learned_mask2,
                     float* stitched, int32_t B, int32_t H, int32_t W, void* stream);
/* out.py:284 normalize_fn: clip(0,255)/127.5 - 1.                                                          */
int st_compose_normalize(const float* x, float* out, int64_t n, void* stream);

/* ---- TPS post-pipeline (the core/inference package, in-tree "kornia" back-end, no inpainter) ---------------- */
/* preprocess (tps_pipline.py:213-244): zero-padded k x k mean of flow [B,C,H,W] (row-major window sum / k^2), optional
 * negation (residual_flow_use_forward = False), optional * valid [B,1,H,W].                                             */
int st_flow_boxavg(const float* flow, const float* valid, float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                   int32_t k, int32_t negate, void* stream);
/* advanced_uniform_sample_border_points (sample_point_methods.py:70-90): image [C,H,W] -> grad [H,W] =
 * mean_c |Sobel_x| + mean_c |Sobel_y| (zero padding).                                                                    */
int st_sobel_magnitude(const float* img, float* grad, int32_t C, int32_t H, int32_t W, void* stream);
/* sample_point_methods.py:93-113: for every range (x1,y1,x2,y2) the first arg-max of grad over rows [y1-2, y2+2) x
 * cols [x1-2, x2+2) -> flat index y*W + x.                                                                               */
int st_range_argmax(const float* grad, const int32_t* ranges, int32_t* out_flat_idx, int32_t n_ranges, int32_t H,
                    int32_t W, void* stream);
/* get_point_pairs flow lookup (core/inference/utils.py:61-68) / border_points_mask filter (tps_pipline.py:111-128):
 * out[i, p] = planes[p, y_i, x_i] for integer points (x, y).                                                             */
int st_gather_points(const float* planes, const int32_t* points_xy, float* out, int32_t n, int32_t P, int32_t H,
                     int32_t W, void* stream);
/* TPS fit f(sites_i) = values_i, f(v) = a0 + [ax ay].v + sum_j w_j U(|v - centers_j|) -> kernel_w [n,2], affine_w [3,2];
 * work_f64: (n+3)*(n+6) doubles.  mode 0 = kornia get_tps_transform(points_src = sites, points_dst = centers = values) as
 * called by warp_by_tps (tps_pipline.py:362-378, kornia_tps.py:47-112): normalised points, U = 0.5 d2 log(d2 + 1e-8);
 * mode 1 = pixel-unit r^2 log r^2 spline with centers = sites (OpenCV ThinPlateSplineShapeTransformer's formulation,
 * opencv_tps.py:8-18).  status (device int32, may be NULL): 0, or 1 when a pivot collapsed to rounding level (coincident or
 * collinear control points: the reference's torch.linalg.solve raises there) -- the weights are then meaningless.       */
int st_tps2_solve(const float* sites, const float* centers, const float* values, void* work_f64, float* kernel_w,
                  float* affine_w, int32_t n, int32_t mode, int32_t* status, void* stream);
/* warp_image_tps (kornia_tps.py:114-176): img [C,H,W] -> out [C,H,W]; centers [n,2]; grid_sample(bilinear, zeros).  mode 0 =
 * kornia (normalised mesh), 1 = pixel-unit spline sampled directly, 3 = 1 on uint8 data as the reference's OpenCV branch sees it
 * (opencv_tps.py + utils.py:10: taps truncated to 0..255 integers, result rounded half-to-even and saturated).            */
int st_tps2_warp(const float* img, const float* centers, const float* kernel_w, const float* affine_w, float* out,
                 int32_t C, int32_t H, int32_t W, int32_t n, float kernel_scale, float affine_scale,
                 int32_t align_corners, int32_t mode, void* stream);
/* cv2.erode / cv2.dilate with a k x k rectangle (tps_pipline.py:143-148) = two 1-D passes of this filter (axis 0: x,
 * axis 1: y), window clipped to the image; in != out.                                                                    */
int st_minmax_filter(const float* in, float* out, int32_t planes, int32_t H, int32_t W, int32_t k, int32_t is_max,
                     int32_t axis, void* stream);
/* tps_pipline.py:139-141: inv [h,w] = 1 - (mean_c(warped_mask [C,h,w]) >= 0.5).                                          */
int st_tps_mask_inv(const float* warped_mask, float* inv, int32_t C, int32_t h, int32_t w, void* stream);
/* tps_pipline.py:150-176 with inv_clean = dilate(erode(inv)): tps3 *= tmask (in place), tmask [h,w], mix3 = output2 of
 * the flow/TPS mix, mixmask [h,w], blend3 uint8 [3,h,w] = clip((output1*mask1 + mix*mixmask) / (mask1 + mixmask)).      */
int st_tps_mix_blend(float* tps3, const float* inv_clean, const float* final_warp3, const float* output1_3,
                     const float* mask1_3, float* tmask, float* mix3, float* mixmask, uint8_t* blend3, int32_t h,
                     int32_t w, void* stream);

/* ---- mix_fn plug-ins of the post-pipeline (core/inference/mix_methods/all_img1_with_inpaint.py:8-113,
 *      inpaint_all_area.py:8-73; helpers core/inference/utils.py:125-170): everything but the neural inpainter ----------- */
/* F.conv2d(plane, ones(k,k), padding=pad) on the output domain Ho x Wo (even kernels: the reference crops to [:H,:W]),
 * cmp 0: sum, 1: sum == k*k (erosion of dilate_thin_area), 2: sum >= 1 (its dilations).                                    */
int st_box_sum_cmp(const float* in, int32_t H, int32_t W, float* out, int32_t Ho, int32_t Wo, int32_t k, int32_t pad,
                   int32_t cmp, void* stream);
/* plane ops of dilate_thin_area / thresholds: op 0: o0 = clamp(a*b,0,1), o1 = a*(1-o0); 1: o0 = clamp(a+b,0,1), o1 = (o0>=1);
 * 2: o0 = a > thr.                                                                                                         */
int st_mix_plane_op(const float* a, const float* b, float* o0, float* o1, int64_t n, int32_t op, float thr, void* stream);
/* first lines of mix_fn (method 0: all_img1_with_inpaint.py:44-53, 1: inpaint_all_area.py:43-51): tps_final_warp,
 * tps_final_warp_mask (3 planes each) and channel 0 of the inpaint-area mask before dilate_thin_area.                       */
int st_mix_stage_a(const float* final_warp3, const float* occ, const float* mask1_3, const float* tps3, const float* tmask,
                   float* tfw3, float* tfwm3, float* iam0, int32_t h, int32_t w, int32_t method, void* stream);
/* all_img1_with_inpaint.py:58-77: border / image-1 fill -> inpaint_img_by_only_img1 (3 planes), channel 0 of the
 * "inpaint by other" mask before dilate_thin_area.                                                                          */
int st_mix_stage_b(const float* iam, const float* dil, const float* mask1_3, const float* tfw3, const float* output1_3,
                   float* only_img1_3, float* other0, int32_t h, int32_t w, void* stream);
/* out3 = [clip 0..255](img3) * (invert ? 1 - mask : mask); mask NULL = no mask (all_img1_with_inpaint.py:82,85,101).       */
int st_mix_mul_mask(const float* img3, const float* mask, float* out3, int32_t h, int32_t w, int32_t invert, int32_t clip,
                    void* stream);
/* new_blend_image (tps_pipline.py:186-187): clip((output1*mask1 + output2*mask2)/(mask1+mask2), 0, 255) -> uint8.            */
int st_blend_pair(const float* output1_3, const float* mask1_3, const float* output2_3, const float* mask2, int32_t mask2_planes,
                  uint8_t* blend3, int32_t h, int32_t w, void* stream);

#ifdef __cplusplus
}
#endif
#endif
