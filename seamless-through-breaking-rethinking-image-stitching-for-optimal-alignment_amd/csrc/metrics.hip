// Evaluation metric of the stitching path on the GPU: masked PSNR / SSIM exactly as
// evaluate.py:53-65 feeds scikit-image 0.19.3 -- uint8 truncation of both images, mask = uint8(mean mask)
// (only exactly 1.0 survives), uint8 products, PSNR over the whole 3xHxW array, SSIM with a 7x7 uniform
// window, K1=.01, K2=.03, sample covariance, (win-1)/2 border crop, mean over channels.
// Window sums of uint8 data are exact integers; the per-pixel SSIM is evaluated in fp64; block partials are
// reduced in a fixed order, so the result is deterministic.
#include "common.h"
#include "../../include/stitch_gfx950.h"

__device__ __forceinline__ int u8_trunc(float v) { return (int)fminf(fmaxf(v, 0.f), 255.f); }   // clip(0,255).to(uint8)

__global__ __launch_bounds__(256) void metrics_partial_kernel(const float* __restrict__ img, const float* __restrict__ warped, int wstride,
                                                              const float* __restrict__ maskmean, double* __restrict__ partial, int H, int W,
                                                              int nblk) {
    __shared__ double red[2][4];
    const int b = blockIdx.y;
    const size_t hw = (size_t)H * W;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // over 3*H*W elements of image b
    double se = 0.0, ss = 0.0;
    if (idx < 3 * hw) {
        const int c = idx / hw;
        const int y = (idx % hw) / W, x = idx % W;
        const float* A = img + ((size_t)b * 3 + c) * hw;
        const float* Bw = warped + (size_t)b * wstride + (size_t)c * hw;
        const float* M = maskmean + (size_t)b * hw;
        {
            const int m = (int)M[(size_t)y * W + x];                     // .to(uint8) of the channel-mean mask
            const int a = u8_trunc(A[(size_t)y * W + x]) * m, bb = u8_trunc(Bw[(size_t)y * W + x]) * m;
            se = (double)((a - bb) * (a - bb));
        }
        if (y >= 3 && y < H - 3 && x >= 3 && x < W - 3) {
            int sa = 0, sb = 0, saa = 0, sbb = 0, sab = 0;
            for (int dy = -3; dy <= 3; ++dy)
                for (int dx = -3; dx <= 3; ++dx) {
                    const size_t p = (size_t)(y + dy) * W + (x + dx);
                    const int m = (int)M[p];
                    const int a = u8_trunc(A[p]) * m, bb = u8_trunc(Bw[p]) * m;
                    sa += a; sb += bb; saa += a * a; sbb += bb * bb; sab += a * bb;
                }
            const double NP = 49.0, cov = NP / (NP - 1.0);
            const double ux = sa / NP, uy = sb / NP, uxx = saa / NP, uyy = sbb / NP, uxy = sab / NP;
            const double vx = cov * (uxx - ux * ux), vy = cov * (uyy - uy * uy), vxy = cov * (uxy - ux * uy);
            const double C1 = 6.5025, C2 = 58.5225;                      // (0.01*255)^2, (0.03*255)^2
            ss = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
        }
    }
    for (int o = 32; o > 0; o >>= 1) { se += __shfl_xor(se, o, 64); ss += __shfl_xor(ss, o, 64); }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = se; red[1][wv] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[((size_t)b * nblk + blockIdx.x) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[((size_t)b * nblk + blockIdx.x) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// fixed-order parallel reduction of the per-block partial sums (deterministic: thread t adds partials t, t+256, ... in
// order, then a fixed binary tree) -- one thread walking all 3072 partials took 0.5 ms per pair
__global__ __launch_bounds__(256) void metrics_final_kernel(const double* __restrict__ partial, double* __restrict__ out, int nblk,
                                                            int H, int W) {
    __shared__ double red[2][256];
    const int b = blockIdx.x, t = threadIdx.x;
    double se = 0.0, ss = 0.0;
    for (int i = t; i < nblk; i += 256) { se += partial[((size_t)b * nblk + i) * 2]; ss += partial[((size_t)b * nblk + i) * 2 + 1]; }
    red[0][t] = se; red[1][t] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) { red[0][t] += red[0][t + o]; red[1][t] += red[1][t + o]; }
        __syncthreads();
    }
    if (t == 0) {
        const double mse = red[0][0] / (3.0 * H * W);
        out[2 * b] = 10.0 * log10(65025.0 / mse);
        out[2 * b + 1] = red[1][0] / (3.0 * (H - 6) * (W - 6));
    }
}

extern "C" int st_masked_psnr_ssim(const float* image1, const float* warped, int64_t warped_batch_stride, const float* maskmean,
                                   void* partial_f64, double* out_psnr_ssim, int32_t B, int32_t H, int32_t W, void* stream) {
    if (!image1 || !warped || !maskmean || !partial_f64 || !out_psnr_ssim || B <= 0 || H < 7 || W < 7) return ST_EINVAL;
    const int nblk = (int)(((size_t)3 * H * W + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(metrics_partial_kernel, dim3(nblk, B), dim3(256), 0, s, image1, warped, (int)warped_batch_stride, maskmean,
                       (double*)partial_f64, H, W, nblk);
    hipLaunchKernelGGL(metrics_final_kernel, dim3(B), dim3(256), 0, s, (const double*)partial_f64, out_psnr_ssim, nblk, H, W);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// channel mean of a [B,C,H,W] slice (evaluate.py:45 `valid = final_warp_output[:,3:6].mean(dim=1)`)
__global__ void channel_mean_kernel(const float* __restrict__ x, long bstride, float* __restrict__ out, int B, int C, size_t hw) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * hw) return;
    const size_t b = idx / hw, p = idx % hw;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += x[b * bstride + c * hw + p];
    out[idx] = s / (float)C;
}

extern "C" int st_channel_mean(const float* x, int64_t batch_stride, float* out, int32_t B, int32_t C, int32_t H, int32_t W, void* stream) {
    if (!x || !out || B <= 0 || C <= 0) return ST_EINVAL;
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(channel_mean_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, (long)batch_stride, out, B, C, (size_t)H * W);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// Loader tail of the evaluation / inference harnesses on the GPU (core/datasets.py:383-386, out.py:137-143:
// `torch.from_numpy(img).permute(2, 0, 1).float()`): interleaved uint8 [B,H,W,3] (what the JPEG decoder leaves in the pinned
// staging buffer) -> planar float32 [B,3,H,W].  uint8 -> float is exact, so the planes hold the same values as the host-side
// conversion.  A thread converts 4 consecutive pixels (three aligned dword reads, three float4 stores).
__global__ __launch_bounds__(256) void load_rgb8_kernel(const uint32_t* __restrict__ src, float* __restrict__ dst, size_t hw, size_t quads) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;         // 4-pixel group inside image blockIdx.y
    if (q >= quads) return;
    const size_t b = blockIdx.y;
    const uint32_t* s = src + (b * hw * 3) / 4 + q * 3;
    const uint32_t w0 = s[0], w1 = s[1], w2 = s[2];                    // r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
    float* d = dst + b * 3 * hw + q * 4;
    *(float4*)(d) = make_float4((float)(w0 & 255u), (float)(w0 >> 24), (float)((w1 >> 16) & 255u), (float)((w2 >> 8) & 255u));
    *(float4*)(d + hw) = make_float4((float)((w0 >> 8) & 255u), (float)(w1 & 255u), (float)(w1 >> 24), (float)((w2 >> 16) & 255u));
    *(float4*)(d + 2 * hw) = make_float4((float)((w0 >> 16) & 255u), (float)((w1 >> 8) & 255u), (float)(w2 & 255u), (float)(w2 >> 24));
}

extern "C" int st_load_rgb8(const void* src_u8_hwc, float* dst_chw, int32_t B, int32_t H, int32_t W, void* stream) {
    const size_t hw = (size_t)H * W;
    if (!src_u8_hwc || !dst_chw || B <= 0 || H <= 0 || W <= 0 || (hw & 3) || ((uintptr_t)src_u8_hwc & 3) || ((uintptr_t)dst_chw & 15)) return ST_EINVAL;
    const size_t quads = hw / 4;
    hipLaunchKernelGGL(load_rgb8_kernel, dim3((unsigned)((quads + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)src_u8_hwc, dst_chw,
                       hw, quads);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
