// Element-wise / resampling kernels of the UDIS2 composition stage: everything around the
// convolutions of core/UDIS2/Composition/network.py (which run on st_conv_gemm with dilation).  HBM-bound.
#include "common.h"
#include "../../include/stitch_gfx950.h"

// F.interpolate(mode='nearest') to (oh, ow) on channels-last rows (network.py:70).  ATen's legacy 'nearest':
// src = min(floor(dst * (float)in / out), in - 1), the identity when sizes agree.
__global__ __launch_bounds__(256) void resize_nearest_rows_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int ldo,
                                                                  int B, int H, int W, int C4, int oh, int ow, float sy, float sx) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)B * oh * ow * C4;
    if (idx >= total) return;
    const int c4 = idx % C4;
    const size_t p = idx / C4;
    const int ox = p % ow, oy = (p / ow) % oh, b = p / ((size_t)ow * oh);
    const int iy = oh == H ? oy : min((int)floorf(oy * sy), H - 1);
    const int ix = ow == W ? ox : min((int)floorf(ox * sx), W - 1);
    const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)(b * H + iy) * W + ix) * ldx + 4 * c4);
    *reinterpret_cast<float4*>(out + p * ldo + 4 * c4) = v;
}

extern "C" int st_resize_nearest_rows(const float* x, int32_t ldx, float* out, int32_t ldo, int32_t B, int32_t H, int32_t W,
                                      int32_t C, int32_t oh, int32_t ow, void* stream) {
    if (!x || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldo & 3) || oh <= 0 || ow <= 0) return ST_EINVAL;
    const size_t total = (size_t)B * oh * ow * (C / 4);
    hipLaunchKernelGGL(resize_nearest_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ldo, B,
                       H, W, C / 4, oh, ow, (float)H / (float)oh, (float)W / (float)ow);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// out = a - b on [rows, C] row views (network.py:120-124: x_k - y_k of the shared encoder).
__global__ __launch_bounds__(256) void sub_rows_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                       float* __restrict__ out, int ldo, size_t rows, int C4) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C4) return;
    const size_t r = idx / C4;
    const int c = 4 * (int)(idx - r * C4);
    const float4 x = *reinterpret_cast<const float4*>(a + r * lda + c), y = *reinterpret_cast<const float4*>(b + r * ldb + c);
    *reinterpret_cast<float4*>(out + r * ldo + c) = make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
}

extern "C" int st_sub_rows(const float* a, int32_t lda, const float* b, int32_t ldb, float* out, int32_t ldo, int64_t rows, int32_t C,
                           void* stream) {
    if (!a || !b || !out || rows <= 0 || C <= 0 || (C & 3) || (lda & 3) || (ldb & 3) || (ldo & 3)) return ST_EINVAL;
    const size_t total = (size_t)rows * (C / 4);
    hipLaunchKernelGGL(sub_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, out, ldo,
                       (size_t)rows, C / 4);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// build_model (network.py:8-22): learned masks and the stitched image from the U-Net's sigmoid output.
// warp1/warp2/mask1/mask2 NCHW [B,3,H,W]; net_out rows [B*H*W] (stride ld); outputs NCHW [B,3,H,W].
__global__ __launch_bounds__(256) void compose_blend_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                            const float* __restrict__ m1, const float* __restrict__ m2,
                                                            const float* __restrict__ net, int ld, float* __restrict__ lm1,
                                                            float* __restrict__ lm2, float* __restrict__ st, int B, size_t HW) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * 3 * HW) return;
    const size_t pix = idx % HW, b = idx / (3 * HW);
    const float o = net[(b * HW + pix) * ld];
    const float a = m1[idx], c = m2[idx], ac = a * c;
    const float l1 = (a - ac) + ac * o, l2 = (c - ac) + ac * (1.f - o);
    lm1[idx] = l1; lm2[idx] = l2;
    st[idx] = (w1[idx] + 1.f) * l1 + (w2[idx] + 1.f) * l2 - 1.f;
}

extern "C" int st_compose_blend(const float* warp1, const float* warp2, const float* mask1, const float* mask2, const float* net_out,
                                int32_t ld_net, float* learned_mask1, float* learned_mask2, float* stitched, int32_t B, int32_t H,
                                int32_t W, void* stream) {
    if (!warp1 || !warp2 || !mask1 || !mask2 || !net_out || !learned_mask1 || !learned_mask2 || !stitched || B <= 0 || H <= 0 || W <= 0)
        return ST_EINVAL;
    const size_t total = (size_t)B * 3 * H * W;
    hipLaunchKernelGGL(compose_blend_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, warp1, warp2, mask1, mask2,
                       net_out, ld_net, learned_mask1, learned_mask2, stitched, B, (size_t)H * W);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// out.py:284 normalize_fn: x.clip(0, 255) / 127.5 - 1 (NCHW -> NCHW).
__global__ __launch_bounds__(256) void compose_normalize_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < n) out[idx] = fminf(fmaxf(x[idx], 0.f), 255.f) / 127.5f - 1.0f;
}

extern "C" int st_compose_normalize(const float* x, float* out, int64_t n, void* stream) {
    if (!x || !out || n <= 0) return ST_EINVAL;
    hipLaunchKernelGGL(compose_normalize_kernel, dim3(((size_t)n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, out, (size_t)n);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
