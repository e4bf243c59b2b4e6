// FlowFormer++ decoder gather kernels on gfx950: 9x9 cost lookup, convex 8x upsampling, coordinate
// bookkeeping.  Pure gather / HBM-bound work; lanes of a wave walk contiguous addresses.
#include "common.h"
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

// 9x9 bilinear cost lookup: encode_flow_token + bilinear_sampler (decoder.py:242-260,
// core/utils/utils.py:62-76).  Every query pixel n samples ITS OWN cost map (row n of the
// all-pairs volume, [H2, W2]) at coords1[n] + (i-r, j-r): the first grid axis (i) goes to x
// (RAFT quirk), channel = i*(2r+1) + j.  grid_sample(bilinear, zeros, align_corners=True).
__global__ __launch_bounds__(256) void cost_lookup_kernel(const float* __restrict__ maps, const float* __restrict__ coords,
                                                          float* __restrict__ out, int ldo, int Nq, int H2, int W2, int r) {
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
    float v = 0.f;
    if (xin0 && yin0) v += m[y0 * W2 + x0] * nw;
    if (xin1 && yin0) v += m[y0 * W2 + x1] * ne;
    if (xin0 && yin1) v += m[y1 * W2 + x0] * sw;
    if (xin1 && yin1) v += m[y1 * W2 + x1] * se;
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
        float fx = 0.f, fy = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const size_t p2 = ((size_t)b * H + yy) * W + xx;
            fx = 8.0f * (coords1[p2 * 2] - (float)xx);
            fy = 8.0f * (coords1[p2 * 2 + 1] - (float)yy);
        }
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
