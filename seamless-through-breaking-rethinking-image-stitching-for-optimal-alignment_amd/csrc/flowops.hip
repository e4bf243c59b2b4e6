// FlowFormer++ decoder gather kernels on gfx950: 9x9 cost lookup, convex 8x upsampling, coordinate
// bookkeeping.  Pure gather / HBM-bound work; lanes of a wave walk contiguous addresses.
#include "common.h"
#include <type_traits>
#include "../../include/stitch_gfx950.h"

// coords0 = (x, y) pixel grid, channels-last [B*H*W, 2]   (decoder.py:22-29 initialize_flow)
__global__ void coords_grid_kernel(float* __restrict__ out, int B, int H, int W) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * H * W) return;
    const int r = idx % ((size_t)H * W);
    out[idx * 2] = (float)(r % W);
    out[idx * 2 + 1] = (float)(r / W);
}

extern "C" int st_coords_grid(float* out, int32_t B, int32_t H, int32_t W, void* stream) {
    if (!out) return ST_EINVAL;
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(coords_grid_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, B, H, W);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// flow = coords1 - coords0 (decoder.py:321) written to a zero-padded [N, ld4] buffer (conv input)
// and, optionally, into two columns of a wider activation buffer (gru.py:254 cat([out, flow])).
__global__ void flow_from_coords_kernel(const float* __restrict__ coords1, float* __restrict__ flow4, int ld4,
                                        float* __restrict__ dst2, int ld2, int B, int H, int W) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * H * W) return;
    const int r = idx % ((size_t)H * W);
    const float fx = coords1[idx * 2] - (float)(r % W);
    const float fy = coords1[idx * 2 + 1] - (float)(r / W);
    if (flow4) {
        flow4[idx * ld4] = fx; flow4[idx * ld4 + 1] = fy;
        for (int c = 2; c < ld4; ++c) flow4[idx * ld4 + c] = 0.f;
    }
    if (dst2) { dst2[idx * ld2] = fx; dst2[idx * ld2 + 1] = fy; }
}

extern "C" int st_flow_from_coords(const float* coords1, float* flow4, int32_t ld4, float* dst2, int32_t ld2, int32_t B,
                                   int32_t H, int32_t W, void* stream) {
    if (!coords1) return ST_EINVAL;
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(flow_from_coords_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, coords1, flow4,
                       ld4, dst2, ld2, B, H, W);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// BasicMotionEncoder's flow branch, first layer (gru.py:251 `flo = relu(convf1(flow))`, Conv2d(2, 128, 7, padding=3)) fused
// with flow = coords1 - coords0 (decoder.py:321): a 2-channel 7x7 conv is 98 taps per output -- as an implicit GEMM
// (K = 196 with the channels padded to 4) it ran on the register-staged fallback at 0.17 of the MFMA peak.  Here a lane
// is an OUTPUT CHANNEL (w [49 taps][2][Co], tap-major, staged once per workgroup into LDS with 16-byte loads; a lane
// reads the 14 weights of a kernel row into registers); a 512-thread workgroup takes an 8x4 pixel tile, builds the zero-padded 14x10 flow
// patch in LDS from coords1, and each wave produces 64 channels for one row of 8 pixels: per kernel row 14 broadcast LDS
// reads feed 112 FMAs on 8 independent accumulators; stores are whole 256-byte channel runs.  Also writes the flow
// itself where the following layers want it (flow2 [.., ld2]: gru.py:254 cat([out, flow])).  (A first version with
// lane = pixel and scalar-loaded weights was latency-bound on the scalar cache: 25 us.)
__global__ __launch_bounds__(512) void flow_encode_kernel(const float* __restrict__ coords1, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, int ldo,
                                                          float* __restrict__ flow2, int ld2, int H, int W, int Co,
                                                          __bf16* __restrict__ out_planes, long long out_pstride, long long out_prows,
                                                          __bf16* __restrict__ flow_planes, long long flow_pstride, long long flow_prows, int flow_col) {
    __shared__ float2 patch[10 * 14];
    __shared__ __attribute__((aligned(16))) float wsm[98 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (W + 7) >> 3;
    const int b = blockIdx.z, ty = (int)blockIdx.x / tiles_x, tx = (int)blockIdx.x - ty * tiles_x;
    const size_t img = (size_t)b * H * W;
    const int cbase = (int)blockIdx.y * 128;                          // this workgroup's 128 output channels
    {
        // weights of the channel block -> LDS [98][128] (Co % 4 == 0: rows are 16-byte aligned)
        float4 v[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int f = u * 512 + tid, t = f >> 5, c4 = (f & 31) * 4;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < 98 && cbase + c4 < Co) v[u] = *reinterpret_cast<const float4*>(w + (size_t)t * Co + cbase + c4);
        }
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int f = u * 512 + tid;
            if (f < 98 * 32) *reinterpret_cast<float4*>(wsm + f * 4) = v[u];
        }
    }
    if (tid < 140) {
        const int py = tid / 14, px = tid - py * 14;
        const int y = ty * 4 + py - 3, x = tx * 8 + px - 3;
        float2 f = make_float2(0.f, 0.f);
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const float2 cc = *reinterpret_cast<const float2*>(coords1 + (img + (size_t)y * W + x) * 2);
            f = make_float2(cc.x - (float)x, cc.y - (float)y);
        }
        patch[tid] = f;
    }
    __syncthreads();
    const int row = wave >> 1;                                       // pixel row of the tile (wave-uniform)
    const int y = ty * 4 + row;
    if (y >= H) return;
    const int cl = (wave & 1) * 64 + lane, c = cbase + cl;           // this lane's output channel
    const bool cok = c < Co;
    const float bv = cok ? bias[c] : 0.f;
    float acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) acc[p] = bv;
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
        float2 f[14];
        float wr[14];
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            f[i] = patch[(row + ky) * 14 + i];                      // same address in every lane: broadcast
            wr[i] = wsm[(ky * 14 + i) * 128 + cl];                  // the lane's weights of this kernel row: (kx, component)
        }
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                acc[p] = fmaf(f[p + kx].x, wr[kx * 2], acc[p]);
                acc[p] = fmaf(f[p + kx].y, wr[kx * 2 + 1], acc[p]);
            }
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int x = tx * 8 + p;
        if (x < W && cok) {
            const float o = fmaxf(acc[p], 0.f);
            out[(img + (size_t)y * W + x) * ldo + c] = o;
            if (out_planes) {        // the blocked bf16 planes a split3 consumer reads (st_gemm_desc.split3): 32 lanes = one 64-byte chunk row;
                // neighbouring lanes (channels c, c ^ 1; Co % 32 == 0, so both are inside) exchange halves and the even one stores the dword
                __bf16 h, m, l;
                st_split3(o, h, m, l);
                const unsigned ph = st_bf16_bits(h), pm = st_bf16_bits(m), pl = st_bf16_bits(l);
                const unsigned nh = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ph, 0xB1, 0xF, 0xF, false);
                const unsigned nm = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pm, 0xB1, 0xF, 0xF, false);
                const unsigned nl = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pl, 0xB1, 0xF, 0xF, false);
                if (!(c & 1)) {
                    unsigned* pp = reinterpret_cast<unsigned*>(out_planes + ((size_t)(c >> 5) * out_prows + img + (size_t)y * W + x) * 32 + (c & 31));
                    pp[0] = ph | (nh << 16); pp[out_pstride / 2] = pm | (nm << 16); pp[out_pstride] = pl | (nl << 16);
                }
            }
        }
    }
    if (flow2 && blockIdx.y == 0 && (wave & 1) == 0 && lane < 8 && tx * 8 + lane < W) {
        const float2 f = patch[(row + 3) * 14 + lane + 3];
        const size_t r = img + (size_t)y * W + tx * 8 + lane;
        flow2[r * ld2] = f.x; flow2[r * ld2 + 1] = f.y;
        if (flow_planes) {           // channels flow_col, flow_col + 1 (same chunk: flow_col is even) of the GRU input's planes
            __bf16 hx, mx, lx, hy, my, ly;
            unsigned* pp = reinterpret_cast<unsigned*>(flow_planes + ((size_t)(flow_col >> 5) * flow_prows + r) * 32 + (flow_col & 31));
            st_split3(f.x, hx, mx, lx);
            st_split3(f.y, hy, my, ly);
            pp[0] = st_bf16_bits(hx) | ((unsigned)st_bf16_bits(hy) << 16);
            pp[flow_pstride / 2] = st_bf16_bits(mx) | ((unsigned)st_bf16_bits(my) << 16);
            pp[flow_pstride] = st_bf16_bits(lx) | ((unsigned)st_bf16_bits(ly) << 16);
        }
    }
}

extern "C" int st_flow_encode(const float* coords1, const float* w98, const float* bias, float* out, int32_t ldo, float* flow2,
                              int32_t ld2, int32_t B, int32_t H, int32_t W, int32_t Co, void* stream) {
    if (!coords1 || !w98 || !bias || !out || B <= 0 || H <= 0 || W <= 0 || Co <= 0 || (Co & 3) || ((uintptr_t)w98 & 15)) return ST_EINVAL;
    dim3 grid(((H + 3) / 4) * ((W + 7) / 8), (Co + 127) / 128, B);
    hipLaunchKernelGGL(flow_encode_kernel, grid, dim3(512), 0, (hipStream_t)stream, coords1, w98, bias, out, ldo, flow2, ld2, H, W, Co,
                       (__bf16*)nullptr, 0LL, 0LL, (__bf16*)nullptr, 0LL, 0LL, 0);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// st_flow_encode that ALSO leaves both results as blocked bf16 planes (the operand format of st_gemm_desc.split3): out_planes
// [3][Co/32][out_prows][32] (planes out_pstride elements apart) and channels flow_col, flow_col + 1 of flow_planes.
extern "C" int st_flow_encode_split3(const float* coords1, const float* w98, const float* bias, float* out, int32_t ldo, float* flow2,
                                     int32_t ld2, int32_t B, int32_t H, int32_t W, int32_t Co, void* out_planes, int64_t out_pstride,
                                     int64_t out_prows, void* flow_planes, int64_t flow_pstride, int64_t flow_prows, int32_t flow_col,
                                     void* stream) {
    if (!coords1 || !w98 || !bias || !out || B <= 0 || H <= 0 || W <= 0 || Co <= 0 || (Co & 31) || ((uintptr_t)w98 & 15)) return ST_EINVAL;
    if (!out_planes || out_pstride <= 0 || out_prows < (int64_t)B * H * W) return ST_EINVAL;
    if (flow_planes && (!flow2 || flow_pstride <= 0 || flow_prows < (int64_t)B * H * W || flow_col < 0 || (flow_col & 1))) return ST_EINVAL;
    dim3 grid(((H + 3) / 4) * ((W + 7) / 8), (Co + 127) / 128, B);
    hipLaunchKernelGGL(flow_encode_kernel, grid, dim3(512), 0, (hipStream_t)stream, coords1, w98, bias, out, ldo, flow2, ld2, H, W, Co,
                       (__bf16*)out_planes, (long long)out_pstride, (long long)out_prows, (__bf16*)flow_planes, (long long)flow_pstride,
                       (long long)flow_prows, (int)flow_col);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// 9x9 bilinear cost lookup: encode_flow_token + bilinear_sampler (decoder.py:242-260,
// core/utils/utils.py:62-76).  Every query pixel n samples ITS OWN cost map (row n of the
// all-pairs volume, [H2, W2]) at coords1[n] + (i-r, j-r): the first grid axis (i) goes to x
// (RAFT quirk), channel = i*(2r+1) + j.  grid_sample(bilinear, zeros, align_corners=True).
__global__ __launch_bounds__(256) void cost_lookup_kernel(const float* __restrict__ maps, const float* __restrict__ coords,
                                                          float* __restrict__ out, int ldo, int Nq, int H2, int W2, int r) {
    // every rounding of the reference's coordinate round trip (normalise in bilinear_sampler, un-normalise in
    // grid_sample) and of the four weights is kept: with contraction the weights would come from the unrounded
    // product ix = a*(W-1) while floor() sees the rounded one, a ~2e-6 px inconsistency that the cost volume's
    // ~1e3-per-pixel gradients turn into 5e-3 errors of the looked-up costs (profiles/r2_parity_trace_*.txt)
#pragma clang fp contract(off)
    const int side = 2 * r + 1, nch = side * side;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)Nq * nch) return;
    const int ch = idx % nch;
    const size_t n = idx / nch;
    const int i = ch / side, j = ch % side;
    const float x = coords[n * 2] + (float)(i - r);
    const float y = coords[n * 2 + 1] + (float)(j - r);
    // normalise / un-normalise exactly like bilinear_sampler + grid_sample(align_corners=True)
    const float gx = 2.0f * x / (float)(W2 - 1) - 1.0f, gy = 2.0f * y / (float)(H2 - 1) - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W2 - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(H2 - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
    const float nw = (x1f - ix) * (y1f - iy), ne = (ix - x0f) * (y1f - iy);
    const float sw = (x1f - ix) * (iy - y0f), se = (ix - x0f) * (iy - y0f);
    const float* m = maps + n * (size_t)(H2 * W2);
    const bool xin0 = x0 >= 0 && x0 < W2, xin1 = x1 >= 0 && x1 < W2;
    const bool yin0 = y0 >= 0 && y0 < H2, yin1 = y1 >= 0 && y1 < H2;
    // ATen CPU grid_sample: one product, then three fused multiply-adds (out-of-range taps contribute 0)
    float v = (xin0 && yin0 ? m[y0 * W2 + x0] : 0.f) * nw;
    v = __fmaf_rn(xin1 && yin0 ? m[y0 * W2 + x1] : 0.f, ne, v);
    v = __fmaf_rn(xin0 && yin1 ? m[y1 * W2 + x0] : 0.f, sw, v);
    v = __fmaf_rn(xin1 && yin1 ? m[y1 * W2 + x1] : 0.f, se, v);
    out[n * ldo + ch] = v;
}

extern "C" int st_cost_lookup(const float* maps, const float* coords, float* out, int32_t ldo, int32_t Nq, int32_t H2,
                              int32_t W2, int32_t r, void* stream) {
    if (!maps || !coords || !out || Nq <= 0 || r < 0 || ldo < (2 * r + 1) * (2 * r + 1)) return ST_EINVAL;
    const size_t total = (size_t)Nq * (2 * r + 1) * (2 * r + 1);
    hipLaunchKernelGGL(cost_lookup_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, maps, coords, out,
                       ldo, Nq, H2, W2, r);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// Convex 8x upsampling, MemoryDecoder.upsample_flow (decoder.py:214-225).
//   coords1 [B*H*W, 2]; mask [B*H*W, ldm] with channel k*64 + i*8 + j (k = 3x3 tap, (i, j) sub-pixel)
//   out NCHW [B, 2, 8H, 8W] = sum_k softmax_k(mask) * 8*flow(tap k, zero padded)
__global__ __launch_bounds__(256) void convex_upsample_kernel(const float* __restrict__ coords1, const float* __restrict__ mask,
                                                              int ldm, float* __restrict__ out, int B, int H, int W) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t pix = gid >> 6;
    if (pix >= (size_t)B * H * W) return;
    const int sub = gid & 63, i = sub >> 3, j = sub & 7;
    const int b = pix / ((size_t)H * W), r = pix % ((size_t)H * W);
    const int y = r / W, x = r % W;
    const float* mp = mask + pix * ldm + sub;
    float mv[9], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; ++k) { mv[k] = mp[k * 64]; mx = fmaxf(mx, mv[k]); }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { mv[k] = expf(mv[k] - mx); sum += mv[k]; }
    float ox = 0.f, oy = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        // (unconditional loads from a clamped address: the nine taps are in flight together)
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const size_t p2 = ((size_t)b * H + min(max(yy, 0), H - 1)) * W + min(max(xx, 0), W - 1);
        const float2 cxy = *reinterpret_cast<const float2*>(coords1 + p2 * 2);
        const float okf = ok ? 8.0f : 0.0f;                          // (a select on the loaded value would be turned back into a branch around the load)
        const float fx = okf * (cxy.x - (float)xx), fy = okf * (cxy.y - (float)yy);
        const float wgt = mv[k] / sum;
        ox += wgt * fx; oy += wgt * fy;
    }
    const size_t HW8 = (size_t)64 * H * W;
    const size_t o = (size_t)b * 2 * HW8 + (size_t)(8 * y + i) * (8 * W) + (8 * x + j);
    out[o] = ox;
    out[o + HW8] = oy;
}

extern "C" int st_convex_upsample(const float* coords1, const float* mask, int32_t ldm, float* out, int32_t B, int32_t H,
                                  int32_t W, void* stream) {
    if (!coords1 || !mask || !out || ldm < 576) return ST_EINVAL;
    const size_t total = (size_t)B * H * W * 64;
    hipLaunchKernelGGL(convex_upsample_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, coords1, mask,
                       ldm, out, B, H, W);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// Fused per-pixel token chain of one decoder refinement iteration (decoder.py:305-312 with
// decoder.py:62-109 CrossAttentionLayer, flow_or_pe='and'):
//   query = W2 . gelu(W0 . cost_forward + b0) + b2                       (flow_token_encoder)
//   q     = Wq . (LN1(query) + sinePE(coords1)) + bq
//   x     = query + Wp . MHA(q; k, v of the pixel's 8 cost-memory tokens) + bp     (8 heads x 8)
//   out   = x + Wf3 . gelu(Wf0 . LN2(x) + bf0) + bf3                      -> cost_global
// Every step is row-local with 64 channels, so a 16-row tile goes through all six 64-wide products on
// v_mfma_f32_16x16x4_f32 without leaving the CU (two waves per tile, see the kernel): weights sit in LDS once per
// workgroup, the tile bounces through LDS slabs only to turn the MFMA C layout back into an A operand.
// Replaces 11 launches per iteration (2 GEMM + LN + PE + GEMM + attention + GEMM + LN + 2 GEMM) that were each
// latency-bound at M = 4096..8192 rows.
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TC_LDW 68      // LDS row stride of a 64-wide weight / activation row (floats): conflict-free b128 reads
#define TC_LDW0 100    // row stride for the 84(+12 zero)-wide first layer

struct TokenChainArgs {
    const float *w0, *b0, *w2, *b2, *n1w, *n1b, *wq, *bq, *wp, *bp, *n2w, *n2b, *wf0, *bf0, *wf3, *bf3;
};

__device__ __forceinline__ float sum16(float v) {     // across the 16 lanes that share lane>>4
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}

// acc[t] (tile t of THIS wave's 16x32 column half, C layout) = X[16 x K] . Wh[32 x K]^T ; Wh = the 32 weight rows of the half
template <int K, int LDX, int LDWT>
__device__ __forceinline__ void tc_gemm2(const float* __restrict__ X, const float* __restrict__ Wh, f32x4 acc[2], int lane) {
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < K / 16; ++j) {
        const float4 a = *reinterpret_cast<const float4*>(X + i * LDX + 16 * j + 4 * g);
        float4 b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) b[t] = *reinterpret_cast<const float4*>(Wh + (16 * t + i) * LDWT + 16 * j + 4 * g);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[t].w, acc[t], 0, 0, 0);
    }
}

// C layout -> the wave's 32 columns (starting at col0) of a row-major LDS slab with row stride TC_LDW
__device__ __forceinline__ void tc_store2(float* __restrict__ X, const f32x4 v[2], int col0, int lane) {
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) X[(4 * g + r) * TC_LDW + col0 + 16 * t + c] = v[t][r];
}

// LayerNorm over the 64 columns of a row whose halves live in two waves: each wave reduces its 32 columns to (mean, sum of squared
// deviations) exactly as a two-pass LayerNorm would, the two halves meet through `red` (one workgroup barrier) and are merged with the
// pairwise update  M2 = M2_a + M2_b + (mean_a - mean_b)^2 * 32*32/64  -- no E[x^2] - mean^2 cancellation anywhere.
__device__ __forceinline__ void tc_layernorm2(f32x4 v[2], const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ red_mine,
                                              const float* __restrict__ red_other, int col0, int lane) {
    const int c = lane & 15, g = lane >> 4;
    float mh[4], m2h[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        mh[r] = sum16(v[0][r] + v[1][r]) * (1.0f / 32.0f);
        const float d0 = v[0][r] - mh[r], d1 = v[1][r] - mh[r];
        m2h[r] = sum16(d0 * d0 + d1 * d1);
        if (c == 0) { red_mine[(4 * g + r) * 2] = mh[r]; red_mine[(4 * g + r) * 2 + 1] = m2h[r]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float mo = red_other[(4 * g + r) * 2], m2o = red_other[(4 * g + r) * 2 + 1];
        const float mean = 0.5f * (mh[r] + mo), d = mh[r] - mo;
        const float var = ((m2h[r] + m2o) + d * d * 16.0f) * (1.0f / 64.0f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
        for (int t = 0; t < 2; ++t) v[t][r] = (v[t][r] - mean) * rstd * w[col0 + 16 * t + c] + b[col0 + 16 * t + c];
    }
}

template <int I, int N, class F>
__device__ __forceinline__ void tc_static_for(F&& f) {       // literal indices: the staging arrays become registers before any scheduling
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); tc_static_for<I + 1, N>(f); }
}

// Work split: a 16-row tile is carried by TWO waves, each owning 32 of the 64 columns of every product (two 16x16 MFMA tiles, half the
// MFMAs, half the epilogue / GELU / sine arithmetic, 4 of the 8 attention heads): the chain is a serial sequence of short dependent
// steps, one wave per SIMD, so its duration is the instruction count of ONE wave -- halving that count matters, MFMA utilisation does
// not.  The halves meet in LDS: activations ping-pong between two slabs (S0 / S1) with one workgroup barrier per product, LayerNorm
// statistics through `red`.  A workgroup = RG row tiles x 2 halves (29 -> 19.5 us stand-alone at 8192 rows with RG = 2).
// Memory schedule: nothing hides a memory round trip here, so the global operands are requested in two batches of unconditional
// loads (a row / token out of range reads a clamped address and is masked in the arithmetic: hipcc turns `cond ? *p : 0` into
// load / s_waitcnt vmcnt(0) pairs -- an earlier form of this kernel paid 40 dependent L2 round trips, tools/isa_waits.py): first the
// cost rows, coords, the six weight matrices and the ten bias / LayerNorm vectors, all staged into LDS; then, once those registers are
// free, the k AND v slices of the wave's heads (32 float4 per lane), which land under the first three products.
template <int RG>                                    // row tiles per workgroup (2 waves each)
__global__ __launch_bounds__(128 * RG) void decoder_token_chain_kernel(float* __restrict__ corr, int ldc, const float* __restrict__ coords1,
                                                                       const float* __restrict__ kv, TokenChainArgs A, int rows, int ntok) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* W0 = sm;                                  // [64][TC_LDW0]
    float* W2 = W0 + 64 * TC_LDW0;                   // five [64][TC_LDW]
    float* Wq = W2 + 64 * TC_LDW;
    float* Wp = Wq + 64 * TC_LDW;
    float* Wf0 = Wp + 64 * TC_LDW;
    float* Wf3 = Wf0 + 64 * TC_LDW;
    float* Vec = Wf3 + 64 * TC_LDW;                  // ten 64-float vectors: b0 b2 n1w n1b bq bp n2w n2b bf0 bf3
    float* Slab = Vec + 10 * 64;                     // RG row tiles x { S0 [16][TC_LDW0], S1 [16][TC_LDW] }
    float* Red = Slab + RG * 16 * (TC_LDW0 + TC_LDW); // RG row tiles x 2 halves x 16 rows x (mean, M2)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NT = 128 * RG;
    const int rg = wave >> 1, hf = wave & 1, col0 = 32 * hf;
    float* S0 = Slab + rg * 16 * (TC_LDW0 + TC_LDW);
    float* S1 = S0 + 16 * TC_LDW0;
    float* red_mine = Red + (rg * 2 + hf) * 32;
    const float* red_other = Red + (rg * 2 + (hf ^ 1)) * 32;
    const int row0 = (blockIdx.x * RG + rg) * 16;
    const int c = lane & 15, g = lane >> 4;
    const int jmax = ntok - 1;

    // ---- batch 1: cost rows, weights, vectors, coords
    constexpr int P0 = 1536 / NT, P1 = 1024 / NT;
    float4 xv[3], w0v[P0];
    f32x4 wv[5 * P1], vecv;              // (first-class vectors: a float4 copied global -> array -> LDS stays a memcpy chain through the stack)
    float cx[4], cy[4];
    {
        const int e = min(tid, 159), q = e >> 4;
        const float* vp = q == 0 ? A.b0 : q == 1 ? A.b2 : q == 2 ? A.n1w : q == 3 ? A.n1b : q == 4 ? A.bq : q == 5 ? A.bp : q == 6 ? A.n2w
                          : q == 7 ? A.n2b : q == 8 ? A.bf0 : A.bf3;
        vecv = *reinterpret_cast<const f32x4*>(vp + 4 * (e & 15));
    }
    tc_static_for<0, 3>([&](auto ic) {               // 16 cost_forward rows (84 wide = 21 float4, zero padded to 96), half of them per wave
        constexpr int i = decltype(ic)::value;
        const int e = (hf * 64 + lane) + 128 * i, r = e / 24, k4 = e % 24;
        xv[i] = *reinterpret_cast<const float4*>(corr + (size_t)min(row0 + r, rows - 1) * ldc + 4 * min(k4, 20));
    });
    tc_static_for<0, P0>([&](auto ic) {              // first layer: 64 x 24 float4 (84 real columns = 21, zero up to 96)
        constexpr int i = decltype(ic)::value;
        const int e = tid + NT * i, n = e / 24, k4 = e % 24;
        w0v[i] = *reinterpret_cast<const float4*>(A.w0 + n * 84 + 4 * min(k4, 20));
    });
    tc_static_for<0, P1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int off = 4 * (tid + NT * i);
        wv[5 * i + 0] = *reinterpret_cast<const f32x4*>(A.w2 + off);
        wv[5 * i + 1] = *reinterpret_cast<const f32x4*>(A.wq + off);
        wv[5 * i + 2] = *reinterpret_cast<const f32x4*>(A.wp + off);
        wv[5 * i + 3] = *reinterpret_cast<const f32x4*>(A.wf0 + off);
        wv[5 * i + 4] = *reinterpret_cast<const f32x4*>(A.wf3 + off);
    });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float2 p2 = *reinterpret_cast<const float2*>(coords1 + (size_t)min(row0 + 4 * g + r, rows - 1) * 2);
        cx[r] = p2.x; cy[r] = p2.y;
    }
    // ---- -> LDS
    tc_static_for<0, 3>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int e = (hf * 64 + lane) + 128 * i, r = e / 24, k4 = e % 24;
        *reinterpret_cast<float4*>(S0 + r * TC_LDW0 + 4 * k4) = k4 < 21 ? xv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    });
    tc_static_for<0, P0>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int e = tid + NT * i, n = e / 24, k4 = e % 24;
        *reinterpret_cast<float4*>(W0 + n * TC_LDW0 + 4 * k4) = k4 < 21 ? w0v[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    });
    tc_static_for<0, P1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int e = tid + NT * i, n = e >> 4, k4 = e & 15, off = n * TC_LDW + 4 * k4;
        *reinterpret_cast<f32x4*>(W2 + off) = wv[5 * i + 0];
        *reinterpret_cast<f32x4*>(Wq + off) = wv[5 * i + 1];
        *reinterpret_cast<f32x4*>(Wp + off) = wv[5 * i + 2];
        *reinterpret_cast<f32x4*>(Wf0 + off) = wv[5 * i + 3];
        *reinterpret_cast<f32x4*>(Wf3 + off) = wv[5 * i + 4];
    });
    if (tid < 160) *reinterpret_cast<f32x4*>(Vec + 4 * tid) = vecv;
    // ---- batch 2: lane (row = lane & 15, e = lane >> 4) owns head 4 hf + e of its row (columns 8 head .. +7): that slice of the row's
    // k / v tokens, first read three products later
    asm volatile("" ::: "memory");
    const int head = 4 * hf + g;
    float4 kk[8][2], vv[8][2];
    {
        const float* kvr = kv + (size_t)min(row0 + c, rows - 1) * ntok * 128 + 8 * head;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) kk[j][e] = *reinterpret_cast<const float4*>(kvr + min(j, jmax) * 128 + 4 * e);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) vv[j][e] = *reinterpret_cast<const float4*>(kvr + min(j, jmax) * 128 + 64 + 4 * e);
    }
    asm volatile("" ::: "memory");
    __syncthreads();
    const float *b0 = Vec, *b2 = Vec + 64, *n1w = Vec + 128, *n1b = Vec + 192, *bq = Vec + 256, *bp = Vec + 320, *n2w = Vec + 384,
                *n2b = Vec + 448, *bf0 = Vec + 512, *bf3 = Vec + 576;
    f32x4 acc[2], query[2], x[2];
    // ---- flow_token_encoder
    tc_gemm2<96, TC_LDW0, TC_LDW0>(S0, W0 + col0 * TC_LDW0, acc, lane);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = st_act(acc[t][r] + b0[col0 + 16 * t + c], ST_ACT_GELU);
    tc_store2(S1, acc, col0, lane);
    __syncthreads();
    tc_gemm2<64, TC_LDW, TC_LDW>(S1, W2 + col0 * TC_LDW, query, lane);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) query[t][r] += b2[col0 + 16 * t + c];
    // ---- LN1 + sine PE of coords1 (columns: sin x | cos x | sin y | cos y, 16 each, band = c: this wave has x (hf = 0) or y)
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = query[t];
    tc_layernorm2(acc, n1w, n1b, red_mine, red_other, col0, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float a = ((3.14f * (hf ? cy[r] : cx[r])) * (float)c) * 0.005f;
        acc[0][r] += sinf(a); acc[1][r] += cosf(a);
    }
    tc_store2(S0, acc, col0, lane);                  // S0 as a [16][TC_LDW] slab from here on
    __syncthreads();
    tc_gemm2<64, TC_LDW, TC_LDW>(S0, Wq + col0 * TC_LDW, acc, lane);     // q: the wave's 32 columns = its own 4 heads
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] += bq[col0 + 16 * t + c];
    // ---- 8-head attention over the pixel's ntok memory tokens (head = column / 8), in ROW layout: the wave's q columns go to the
    // slab (its own columns: a wave barrier is enough), lane (row, e) reads head 4 hf + e of its row, k / v come from registers
    tc_store2(S1, acc, col0, lane);
    __builtin_amdgcn_wave_barrier();
    {
        const float scale = 0.35355339059327373f;                         // 8^-0.5
        float qv[8], o[8];
        {
            const float4 t0 = *reinterpret_cast<const float4*>(S1 + c * TC_LDW + 8 * head);
            const float4 t1 = *reinterpret_cast<const float4*>(S1 + c * TC_LDW + 8 * head + 4);
            qv[0] = t0.x; qv[1] = t0.y; qv[2] = t0.z; qv[3] = t0.w; qv[4] = t1.x; qv[5] = t1.y; qv[6] = t1.z; qv[7] = t1.w;
        }
        float s0[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a0 = 0.f;
            const float kf[8] = {kk[j][0].x, kk[j][0].y, kk[j][0].z, kk[j][0].w, kk[j][1].x, kk[j][1].y, kk[j][1].z, kk[j][1].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) a0 = fmaf(qv[e], kf[e], a0);
            s0[j] = j < ntok ? a0 * scale : -INFINITY;
        }
        float m0 = s0[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) m0 = fmaxf(m0, s0[j]);
        float d0 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // (a token past ntok has s = -inf: p = exp(-inf) = 0 exactly, and its clamped v adds +0)
            const float p0 = expf(s0[j] - m0);
            d0 += p0;
            const float vf[8] = {vv[j][0].x, vv[j][0].y, vv[j][0].z, vv[j][0].w, vv[j][1].x, vv[j][1].y, vv[j][1].z, vv[j][1].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(p0, vf[e], o[e]);
        }
        const float i0 = 1.0f / d0;
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<float4*>(S1 + c * TC_LDW + 8 * head) = make_float4(o[0] * i0, o[1] * i0, o[2] * i0, o[3] * i0);
        *reinterpret_cast<float4*>(S1 + c * TC_LDW + 8 * head + 4) = make_float4(o[4] * i0, o[5] * i0, o[6] * i0, o[7] * i0);
    }
    __syncthreads();
    tc_gemm2<64, TC_LDW, TC_LDW>(S1, Wp + col0 * TC_LDW, acc, lane);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[t][r] = query[t][r] + (acc[t][r] + bp[col0 + 16 * t + c]);     // short_cut + proj
    // ---- FFN
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = x[t];
    tc_layernorm2(acc, n2w, n2b, red_mine, red_other, col0, lane);
    tc_store2(S0, acc, col0, lane);
    __syncthreads();
    tc_gemm2<64, TC_LDW, TC_LDW>(S0, Wf0 + col0 * TC_LDW, acc, lane);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = st_act(acc[t][r] + bf0[col0 + 16 * t + c], ST_ACT_GELU);
    tc_store2(S1, acc, col0, lane);
    __syncthreads();
    tc_gemm2<64, TC_LDW, TC_LDW>(S1, Wf3 + col0 * TC_LDW, acc, lane);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 4 * g + r;
            if (row < rows) corr[(size_t)row * ldc + 84 + col0 + 16 * t + c] = x[t][r] + (acc[t][r] + bf3[col0 + 16 * t + c]);
        }
}

extern "C" int st_decoder_token_chain(float* corr, int32_t ld_corr, const float* coords1, const float* kv, const float* const* weights16,
                                      int32_t rows, int32_t ntok, void* stream) {
    if (!corr || !coords1 || !kv || !weights16 || rows <= 0 || ntok <= 0 || ntok > 8 || ld_corr < 148 || (ld_corr & 3)) return ST_EINVAL;
    TokenChainArgs A;
    const float** dst = reinterpret_cast<const float**>(&A);
    for (int i = 0; i < 16; ++i) { if (!weights16[i]) return ST_EINVAL; dst[i] = weights16[i]; }
    // 4 row tiles (8 waves, 64 rows) per workgroup: the kernel holds 156 KB of LDS and keeps the CUs it runs on to itself, so it should
    // hold as few CU-microseconds as possible: 128 workgroups x ~24 us (two waves per SIMD) against 256 x 19.5 us with 2 row tiles --
    // measured with three pairs in flight: 81.4 vs 81.2 pairs/s (one pair in flight: 71.8 vs 72.1)
    constexpr int RG = 4;
    const size_t lds = (size_t)(64 * TC_LDW0 + 5 * 64 * TC_LDW + 640 + RG * 16 * (TC_LDW0 + TC_LDW) + RG * 64) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)decoder_token_chain_kernel<RG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(decoder_token_chain_kernel<RG>, dim3((rows + 16 * RG - 1) / (16 * RG)), dim3(128 * RG), lds, (hipStream_t)stream, corr, ld_corr,
                       coords1, kv, A, rows, ntok);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
