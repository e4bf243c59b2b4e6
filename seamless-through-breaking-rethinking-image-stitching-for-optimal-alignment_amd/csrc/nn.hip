// Row-wise / window / small-attention kernels of the FlowFormer++ and UDIS2 networks on gfx950.
// HBM-bound elementwise and reduction work: 16-B accesses where the layout allows, wave-shuffle
// reductions (64 lanes), K/V tiles staged in LDS for the attention variants.
#include "common.h"
#include "../../include/stitch_gfx950.h"

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim (nn.LayerNorm; reference e.g. encoder.py:58,156-172, twins.py:787)
// one wave per row, C <= 64*16
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ out, int ldo,
                                                        int rows, int C, float eps) {
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < C ? xr[c] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        const float dlt = c < C ? v[i] - mean : 0.f;
        q += dlt * dlt;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    float* orow = out + (size_t)row * ldo;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        if (c < C) orow[c] = (v[i] - mean) * rstd * w[c] + b[c];
    }
}

// C == 128 (every LayerNorm of the cost encoder / vertical layers, 65536 x 128 and up: pure HBM streaming): half a
// wave per row with one 16-B load per lane, four rows per thread issued before the first reduction so that each wave
// keeps 4 KiB in flight instead of 512 B.
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(256) void layernorm128_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ out, int ldo,
                                                           int rows, float eps) {
    constexpr int R = 4;                                   // rows per thread; a workgroup covers 8 * R rows
    const int sub = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int row0 = blockIdx.x * (8 * R) + sub;
    float4 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + 8 * r;
        v[r] = row < rows ? *reinterpret_cast<const float4*>(x + (size_t)row * ldx + 4 * l) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 wv = *reinterpret_cast<const float4*>(w + 4 * l), bv = *reinterpret_cast<const float4*>(b + 4 * l);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + 8 * r;
        const float mean = half_wave_sum((v[r].x + v[r].y) + (v[r].z + v[r].w)) * (1.0f / 128.0f);
        const float dx = v[r].x - mean, dy = v[r].y - mean, dz = v[r].z - mean, dw = v[r].w - mean;
        const float rstd = 1.0f / sqrtf(half_wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.0f / 128.0f) + eps);
        if (row < rows)
            *reinterpret_cast<float4*>(out + (size_t)row * ldo + 4 * l) =
                make_float4(dx * rstd * wv.x + bv.x, dy * rstd * wv.y + bv.y, dz * rstd * wv.z + bv.z, dw * rstd * wv.w + bv.w);
    }
}

extern "C" int st_layernorm(const float* x, int32_t ldx, const float* w, const float* b, float* out, int32_t ldo,
                            int32_t rows, int32_t C, float eps, void* stream) {
    if (!x || !w || !b || !out || rows <= 0 || C <= 0 || C > 1024) return ST_EINVAL;
    if (C == 128 && !(ldx & 3) && !(ldo & 3) && !(((uintptr_t)x | (uintptr_t)out | (uintptr_t)w | (uintptr_t)b) & 15)) {
        hipLaunchKernelGGL(layernorm128_kernel, dim3((rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, ldx, w, b, out, ldo,
                           rows, eps);
        ST_CHECK_LAUNCH();
        return ST_OK;
    }
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, w, b, out,
                       ldo, rows, C, eps);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// In-place row softmax (GMA attention, gma.py:72): one workgroup per row, row kept in registers.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, int ld, int C) {
    __shared__ float red[8];
    float* xr = x + (size_t)blockIdx.x * ld;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    float v[16];                      // C <= 4096
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = t + 256 * i;
        v[i] = c < C ? xr[c] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = t + 256 * i;
        v[i] = c < C ? expf(v[i] - mx) : 0.f;
        s += v[i];
    }
    s = wave_sum(s);
    if (lane == 0) red[4 + wv] = s;
    __syncthreads();
    s = (red[4] + red[5]) + (red[6] + red[7]);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = t + 256 * i;
        if (c < C) xr[c] = v[i] / s;
    }
}

extern "C" int st_softmax_rows(float* x, int32_t ld, int32_t rows, int32_t C, void* stream) {
    if (!x || rows <= 0 || C <= 0 || C > 4096) return ST_EINVAL;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, ld, C);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// F.normalize(p=2, dim=channel) on NHWC rows (network.py:150-151): x / max(||x||, 1e-12)
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int C) {
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c] * xr[c];
    const float nrm = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    for (int c = lane; c < C; c += 64) out[(size_t)row * C + c] = xr[c] / nrm;
}

extern "C" int st_l2norm_rows(const float* x, float* out, int32_t rows, int32_t C, void* stream) {
    if (!x || !out || rows <= 0 || C <= 0) return ST_EINVAL;
    hipLaunchKernelGGL(l2norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, out, rows, C);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d on NHWC (network.py:22-35 2x2/2, torchvision resnet maxpool 3x3/2 pad 1)
__global__ void maxpool_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H, int W, int C,
                               int k, int s, int p, int Ho, int Wo) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * Ho * Wo * C;
    if (idx >= total) return;
    const int c = idx % C;
    size_t r = idx / C;
    const int ox = r % Wo; r /= Wo;
    const int oy = r % Ho;
    const int b = r / Ho;
    float m = -INFINITY;
    for (int a = 0; a < k; ++a)
        for (int e = 0; e < k; ++e) {
            const int iy = oy * s - p + a, ix = ox * s - p + e;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) m = fmaxf(m, x[(((size_t)b * H + iy) * W + ix) * C + c]);
        }
    out[idx] = m;
}

extern "C" int st_maxpool_nhwc(const float* x, float* out, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                               int32_t s, int32_t p, void* stream) {
    if (!x || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || s <= 0) return ST_EINVAL;
    const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
    const size_t total = (size_t)B * Ho * Wo * C;
    hipLaunchKernelGGL(maxpool_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, out, B, H, W, C, k,
                       s, p, Ho, Wo);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// PEG: depthwise 3x3 (pad 1) + bias + identity on NHWC tokens (twins.py:793-808). w: [9, C].
__global__ void dwconv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                 float* __restrict__ out, int B, int H, int W, int C) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // float4 granules
    const int C4 = C >> 2;
    const size_t total = (size_t)B * H * W * C4;
    if (idx >= total) return;
    const int c = (idx % C4) * 4;
    size_t r = idx / C4;
    const int ox = r % W; r /= W;
    const int oy = r % H;
    const int b = r / H;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            // unconditional loads (a padded tap reads a clamped address and is zeroed afterwards): all 18 are in flight together --
            // behind a branch hipcc waited for each pair in turn, nine dependent round trips per thread
            const int iy = oy - 1 + a, ix = ox - 1 + e;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
            float4 xv = *reinterpret_cast<const float4*>(x + (((size_t)b * H + cy) * W + cx) * C + c);
            const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)(a * 3 + e) * C + c);
            if (!ok) xv = make_float4(0.f, 0.f, 0.f, 0.f);           // fma(0, w, acc) = acc: the same sum as skipping the tap
            acc.x = fmaf(xv.x, wv.x, acc.x); acc.y = fmaf(xv.y, wv.y, acc.y);
            acc.z = fmaf(xv.z, wv.z, acc.z); acc.w = fmaf(xv.w, wv.w, acc.w);
        }
    const float4 bv = *reinterpret_cast<const float4*>(bias + c);
    const float4 idt = *reinterpret_cast<const float4*>(x + (((size_t)b * H + oy) * W + ox) * C + c);
    float4 o;
    o.x = (acc.x + bv.x) + idt.x; o.y = (acc.y + bv.y) + idt.y;
    o.z = (acc.z + bv.z) + idt.z; o.w = (acc.w + bv.w) + idt.w;
    *reinterpret_cast<float4*>(out + (((size_t)b * H + oy) * W + ox) * C + c) = o;
}

extern "C" int st_dwconv3x3_residual(const float* x, const float* w, const float* bias, float* out, int32_t B, int32_t H,
                                     int32_t W, int32_t C, void* stream) {
    if (!x || !w || !bias || !out || C % 4) return ST_EINVAL;
    const size_t total = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, B,
                       H, W, C);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// LinearPositionEmbeddingSine (attention.py:156-161): literal 3.14, bands k/200, layout
// [sin x | cos x | sin y | cos y].  Coordinates come either from `coords` ([rows, ldc] = x, y) or
// from the row index on a Wg-wide grid (optionally window-local modulo ws), scaled + offset.
__global__ void sine_pe_kernel(float* __restrict__ out, int ld, int rows, int dim, const float* __restrict__ coords,
                               int ldc, int Wg, int ws, int period, float cscale, float coff, int accumulate) {
    const int q = dim >> 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)rows * q) return;
    const int f = idx % q, row = idx / q;
    float x, y;
    if (coords) { x = coords[(size_t)row * ldc]; y = coords[(size_t)row * ldc + 1]; }
    else {
        const int pr = period > 0 ? row % period : row;
        int gx = pr % Wg, gy = pr / Wg;
        if (ws > 0) { gx %= ws; gy %= ws; }
        x = (float)gx * cscale + coff; y = (float)gy * cscale + coff;
    }
    const float fb = (float)f;
    const float ax = ((3.14f * x) * fb) * 0.005f, ay = ((3.14f * y) * fb) * 0.005f;
    float* o = out + (size_t)row * ld;
    const float v0 = sinf(ax), v1 = cosf(ax), v2 = sinf(ay), v3 = cosf(ay);
    // accumulate: 0 = write, 1 = add, n > 1 = write the columns below n and add to the others (a buffer whose first
    // n columns carry no other term needs no zero fill)
    const float val[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * q + f;
        if (accumulate == 1 || (accumulate > 1 && c >= accumulate)) o[c] += val[i];
        else o[c] = val[i];
    }
}

extern "C" int st_sine_pe(float* out, int32_t ld, int32_t rows, int32_t dim, const float* coords, int32_t ldc, int32_t Wg,
                          int32_t ws, int32_t period, float cscale, float coff, int32_t accumulate, void* stream) {
    if (!out || rows <= 0 || dim <= 0 || dim % 4 || (!coords && Wg <= 0)) return ST_EINVAL;
    const size_t total = (size_t)rows * (dim / 4);
    hipLaunchKernelGGL(sine_pe_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, ld, rows, dim,
                       coords, ldc, Wg, ws, period, cscale, coff, accumulate);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// Small multi-head attention: one thread per (batch, head, query), keys/values read straight
// from L2 (lanes of one (batch, head) share the same K/V addresses -> broadcast).
// Used for the 8-latent cross/self attention (crossattentionlayer.py:37-56, encoder.py:156-172,
// attention.py:9-68) and the decoder's 1-query x 8-key attention (decoder.py:62-109).
// Element (b, t, h, e) of q lives at q + b*bs + t*ts + h*D + e (same for k, v, out).
template <int D>
__global__ __launch_bounds__(256) void attention_small_kernel(const float* __restrict__ q, long q_bs, long q_ts,
                                                              const float* __restrict__ k, long k_bs, long k_ts,
                                                              const float* __restrict__ v, long v_bs, long v_ts,
                                                              float* __restrict__ out, long o_bs, long o_ts, int B,
                                                              int heads, int Nq, int Nk, float scale) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * heads * Nq) return;
    const int iq = idx % Nq;
    const int h = (idx / Nq) % heads;
    const int b = idx / ((long)Nq * heads);
    float qr[D], acc[D];
    const float* qp = q + b * q_bs + iq * q_ts + h * D;
#pragma unroll
    for (int e = 0; e < D; e += 4) {
        const float4 t = *reinterpret_cast<const float4*>(qp + e);
        qr[e] = t.x; qr[e + 1] = t.y; qr[e + 2] = t.z; qr[e + 3] = t.w;
    }
    const float* kb = k + b * k_bs + h * D;
    const float* vb = v + b * v_bs + h * D;
    float mx = -INFINITY;
    for (int j = 0; j < Nk; ++j) {
        const float* kp = kb + j * k_ts;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < D; e += 4) {
            const float4 t = *reinterpret_cast<const float4*>(kp + e);
            s = fmaf(qr[e], t.x, s); s = fmaf(qr[e + 1], t.y, s); s = fmaf(qr[e + 2], t.z, s); s = fmaf(qr[e + 3], t.w, s);
        }
        mx = fmaxf(mx, s * scale);
    }
#pragma unroll
    for (int e = 0; e < D; ++e) acc[e] = 0.f;
    float sum = 0.f;
    for (int j = 0; j < Nk; ++j) {
        const float* kp = kb + j * k_ts;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < D; e += 4) {
            const float4 t = *reinterpret_cast<const float4*>(kp + e);
            s = fmaf(qr[e], t.x, s); s = fmaf(qr[e + 1], t.y, s); s = fmaf(qr[e + 2], t.z, s); s = fmaf(qr[e + 3], t.w, s);
        }
        const float p = __expf(s * scale - mx);
        sum += p;
        const float* vp = vb + j * v_ts;
#pragma unroll
        for (int e = 0; e < D; e += 4) {
            const float4 t = *reinterpret_cast<const float4*>(vp + e);
            acc[e] = fmaf(p, t.x, acc[e]); acc[e + 1] = fmaf(p, t.y, acc[e + 1]);
            acc[e + 2] = fmaf(p, t.z, acc[e + 2]); acc[e + 3] = fmaf(p, t.w, acc[e + 3]);
        }
    }
    const float inv = 1.0f / sum;
    float* op = out + b * o_bs + iq * o_ts + h * D;
#pragma unroll
    for (int e = 0; e < D; e += 4)
        *reinterpret_cast<float4*>(op + e) = make_float4(acc[e] * inv, acc[e + 1] * inv, acc[e + 2] * inv, acc[e + 3] * inv);
}

extern "C" int st_attention_small(const float* q, int64_t q_bs, int64_t q_ts, const float* k, int64_t k_bs, int64_t k_ts,
                                  const float* v, int64_t v_bs, int64_t v_ts, float* out, int64_t o_bs, int64_t o_ts,
                                  int32_t B, int32_t heads, int32_t Nq, int32_t Nk, int32_t D, float scale, void* stream) {
    if (!q || !k || !v || !out || B <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0) return ST_EINVAL;
    const long total = (long)B * heads * Nq;
    dim3 grid((total + 255) / 256), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_AS(DD) hipLaunchKernelGGL(attention_small_kernel<DD>, grid, block, 0, s, q, q_bs, q_ts, k, k_bs, k_ts, v, \
                                         v_bs, v_ts, out, o_bs, o_ts, B, heads, Nq, Nk, scale)
    if (D == 8) LAUNCH_AS(8); else if (D == 16) LAUNCH_AS(16); else if (D == 32) LAUNCH_AS(32); else return ST_EINVAL;
#undef LAUNCH_AS
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// Latent cross-attention with pixel-independent queries (crossattentionlayer.py:37-56 + attention.py:9-68, first layer of
// the cost encoder): the 8 latent queries are the same for every pixel, so  q.(Wk t + bk) = (Wk^T q).t + const  and the
// constant cancels in the softmax; likewise  sum_t p_t (Wv t + bv) = Wv (sum_t p_t t) + bv.  The host therefore takes the
// scores against the un-projected patch tokens (one N = 64 GEMM with the folded queries) and this kernel does the rest
// per pixel: softmax over the P tokens for each of the 64 (latent, head) rows and the pooling z = softmax(S)^T . T
// (64 x P x 128) on v_mfma_f32_32x32x2_f32 -- operands go straight from L2 into the fragment registers (a lane needs exactly
// S[tok][row] and T[tok][col], both coalesced), no LDS, no barrier.  The per-pixel K|V tensor of the reference (P x 256
// floats per pixel: 537 MB at 512^2) is never materialised.
//   scores [pixels*P, ld_s >= 64] (row = (pixel, token), col = latent*8 + head), tokens [pixels*P, ld_t >= 128],
//   z [pixels*64, 128] (row = (pixel, latent, head)).  P even, P <= 64.
typedef float lp_f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void latent_pool_kernel(const float* __restrict__ scores, int ld_s, const float* __restrict__ tokens,
                                                          int ld_t, float* __restrict__ z, int P) {
    const size_t pix = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int ib = wave & 1, jb0 = (wave >> 1) * 2;            // this wave: rows [32*ib, +32), column blocks jb0, jb0+1
    const float* Sp = scores + pix * P * ld_s + ib * 32 + li;  // + tok*ld_s
    const float* Tp = tokens + pix * P * ld_t + jb0 * 32 + li; // + tok*ld_t (+32 for the second block)
    const int steps = P >> 1;                                  // MFMA k = 2: tokens 2s + lh
    float a[32];
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        a[s] = s < steps ? Sp[(size_t)(2 * s + lh) * ld_s] : -INFINITY;
        mx = fmaxf(mx, a[s]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) { a[s] = __expf(a[s] - mx); sum += a[s]; }       // exp(-inf) = 0 for the unused steps
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    lp_f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int s0 = 0; s0 < 32; s0 += 8) {
        if (s0 < steps) {                                      // wave-uniform
            float b0[8], b1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int tok = min(2 * (s0 + u) + lh, P - 1);
                b0[u] = Tp[(size_t)tok * ld_t];
                b1[u] = Tp[(size_t)tok * ld_t + 32];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float av = a[s0 + u] * inv;              // 0 beyond P
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[u], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1[u], acc1, 0, 0, 0);
            }
        }
    }
    float* zp = z + (pix * 64 + ib * 32 + 4 * lh) * 128 + jb0 * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        zp[(size_t)row * 128] = acc0[r];
        zp[(size_t)row * 128 + 32] = acc1[r];
    }
}

extern "C" int st_latent_pool(const float* scores, int32_t ld_s, const float* tokens, int32_t ld_t, float* z, int32_t pixels,
                              int32_t P, void* stream) {
    if (!scores || !tokens || !z || pixels <= 0 || P <= 0 || (P & 1) || P > 64 || ld_s < 64 || ld_t < 128) return ST_EINVAL;
    hipLaunchKernelGGL(latent_pool_kernel, dim3(pixels), dim3(256), 0, (hipStream_t)stream, scores, ld_s, tokens, ld_t, z, P);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// Global sub-sampled attention (GSA, twins.py:336-392,633-680): many queries against <= 256
// pooled keys.  The (batch, head) K/V slab is staged once in LDS and every thread owns one query
// (online softmax); LDS reads are wave-uniform broadcasts.
template <int D>
__global__ __launch_bounds__(256) void attention_kvlds_kernel(const float* __restrict__ q, long q_bs, long q_ts,
                                                              const float* __restrict__ k, long k_bs, long k_ts,
                                                              const float* __restrict__ v, long v_bs, long v_ts,
                                                              float* __restrict__ out, long o_bs, long o_ts, int Nq, int Nk,
                                                              float scale) {
    extern __shared__ __attribute__((aligned(16))) float kv[];   // [Nk][D] K then [Nk][D] V
    constexpr int QPT = 2;                                        // queries per thread: each LDS broadcast feeds 2x the FMAs
    const int h = blockIdx.y, b = blockIdx.z;
    float* Ks = kv;
    float* Vs = kv + (size_t)Nk * D;
    for (int i = threadIdx.x; i < Nk * (D / 4); i += 256) {
        const int j = i / (D / 4), e = (i % (D / 4)) * 4;
        *reinterpret_cast<float4*>(Ks + j * D + e) = *reinterpret_cast<const float4*>(k + b * k_bs + j * k_ts + h * D + e);
        *reinterpret_cast<float4*>(Vs + j * D + e) = *reinterpret_cast<const float4*>(v + b * v_bs + j * v_ts + h * D + e);
    }
    __syncthreads();
    const int iq0 = blockIdx.x * (256 * QPT) + threadIdx.x;      // this thread's queries: iq0, iq0 + 256
    if (iq0 >= Nq) return;
    float qr[QPT][D], acc[QPT][D], mx[QPT], sum[QPT];
    bool live[QPT];
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        const int iq = iq0 + 256 * u;
        live[u] = iq < Nq;
        const float* qp = q + b * q_bs + (long)(live[u] ? iq : iq0) * q_ts + h * D;
#pragma unroll
        for (int e = 0; e < D; e += 4) {
            const float4 t = *reinterpret_cast<const float4*>(qp + e);
            qr[u][e] = t.x * scale; qr[u][e + 1] = t.y * scale; qr[u][e + 2] = t.z * scale; qr[u][e + 3] = t.w * scale;
        }
#pragma unroll
        for (int e = 0; e < D; ++e) acc[u][e] = 0.f;
        mx[u] = -INFINITY; sum[u] = 0.f;
    }
    // online softmax in chunks of 8 keys: one running-max update / accumulator rescale per chunk instead of per key
    for (int j0 = 0; j0 < Nk; j0 += 8) {
        float s[QPT][8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = j0 + jj;
#pragma unroll
            for (int u = 0; u < QPT; ++u) s[u][jj] = 0.f;
            if (j < Nk) {
#pragma unroll
                for (int e = 0; e < D; e += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(Ks + j * D + e);
#pragma unroll
                    for (int u = 0; u < QPT; ++u) {
                        s[u][jj] = fmaf(qr[u][e], t.x, s[u][jj]); s[u][jj] = fmaf(qr[u][e + 1], t.y, s[u][jj]);
                        s[u][jj] = fmaf(qr[u][e + 2], t.z, s[u][jj]); s[u][jj] = fmaf(qr[u][e + 3], t.w, s[u][jj]);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < QPT; ++u) s[u][jj] = -INFINITY;
            }
        }
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
            float nm = mx[u];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) nm = fmaxf(nm, s[u][jj]);
            const float corr = __expf(mx[u] - nm);
            mx[u] = nm;
            sum[u] *= corr;
#pragma unroll
            for (int e = 0; e < D; ++e) acc[u][e] *= corr;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) { s[u][jj] = __expf(s[u][jj] - nm); sum[u] += s[u][jj]; }
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = j0 + jj;
            if (j < Nk) {
#pragma unroll
                for (int e = 0; e < D; e += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(Vs + j * D + e);
#pragma unroll
                    for (int u = 0; u < QPT; ++u) {
                        acc[u][e] = fmaf(s[u][jj], t.x, acc[u][e]); acc[u][e + 1] = fmaf(s[u][jj], t.y, acc[u][e + 1]);
                        acc[u][e + 2] = fmaf(s[u][jj], t.z, acc[u][e + 2]); acc[u][e + 3] = fmaf(s[u][jj], t.w, acc[u][e + 3]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        if (!live[u]) continue;
        const float inv = 1.0f / sum[u];
        float* op = out + b * o_bs + (long)(iq0 + 256 * u) * o_ts + h * D;
#pragma unroll
        for (int e = 0; e < D; e += 4)
            *reinterpret_cast<float4*>(op + e) = make_float4(acc[u][e] * inv, acc[u][e + 1] * inv, acc[u][e + 2] * inv, acc[u][e + 3] * inv);
    }
}

// The same attention on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32) for Nk % 16 == 0, Nk <= 256.
// A wave takes 16 queries at a time:
//   S^T[key, query] = K . Q^T   one 16x16 tile per 16 keys, D/4 MFMAs each; the tile lands with the QUERY in the lane's
//                               column, so a lane holds Nk/4 scores of one query and the softmax is register-local apart from
//                               two shuffles (max, sum) across the four 16-lane groups
//   O^T[d, query]   = V^T . P^T the probabilities are already in the B-operand layout (lane = query column, register =
//                               k slot); V^T fragments are single ds_read_b32 from a [key][D+4] slab (conflict-free)
// The k slot <-> dim / key assignment inside a 4-deep MFMA step is free as long as both operands agree: dims are taken
// as g*(D/4)+s (one or two 16-B reads per lane), keys as 16*kt + 4*g + s (the accumulator layout of S^T).
// The VALU formulation above issues ~2x(16+16) FMAs + bookkeeping per (query, key) on the datapath the MFMA shares, at
// roughly 40 % of the FMA peak; here the same products cost D/2 MFMA cycles per (query, key) and the VALU only does the exp.
typedef float kv_f32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(256) void attention_kv_mfma_kernel(const float* __restrict__ q, long q_bs, long q_ts,
                                                                const float* __restrict__ k, long k_bs, long k_ts,
                                                                const float* __restrict__ v, long v_bs, long v_ts,
                                                                float* __restrict__ out, long o_bs, long o_ts, int Nq, int Nk,
                                                                float scale, int tiles_per_wave) {
    constexpr int LD = D + 4;                       // slab row stride (floats): conflict-free V^T reads, 16-B aligned K reads
    constexpr int DS = D / 4;                       // MFMA steps over the head dim; also dims per k slot
    constexpr int DB = D / 16;                      // 16-row blocks of O^T
    constexpr int MAXT = 16;                        // Nk <= 256
    extern __shared__ __attribute__((aligned(16))) float kv[];
    const int h = blockIdx.y, b = blockIdx.z;
    float* Ks = kv;
    float* Vs = kv + (size_t)Nk * LD;
    for (int i = threadIdx.x; i < Nk * (D / 4); i += 256) {
        const int j = i / (D / 4), e = (i % (D / 4)) * 4;
        *reinterpret_cast<float4*>(Ks + j * LD + e) = *reinterpret_cast<const float4*>(k + b * k_bs + j * k_ts + h * D + e);
        *reinterpret_cast<float4*>(Vs + j * LD + e) = *reinterpret_cast<const float4*>(v + b * v_bs + j * v_ts + h * D + e);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int nkt = Nk >> 4;
    const float* Kl = Ks + c * LD + g * DS;         // + 16*kt*LD: this lane's K fragment (DS consecutive dims)
    const float* Vl = Vs + (4 * g) * LD + c;        // + (16*kt + s)*LD (+16 for the second d block)
    for (int t = 0; t < tiles_per_wave; ++t) {
        const int q0 = ((blockIdx.x * 4 + wave) * tiles_per_wave + t) * 16;
        if (q0 >= Nq) break;                        // wave-uniform
        const int qi = min(q0 + c, Nq - 1);
        float qf[DS];
        {
            const float* qp = q + b * q_bs + (long)qi * q_ts + h * D + g * DS;
#pragma unroll
            for (int e = 0; e < DS; e += 4) {
                const float4 tq = *reinterpret_cast<const float4*>(qp + e);
                // scores are kept in the log2 domain (scale * log2 e folded into Q): the softmax is then a bare v_exp_f32
                const float sl = scale * 1.44269504088896340736f;
                qf[e] = tq.x * sl; qf[e + 1] = tq.y * sl; qf[e + 2] = tq.z * sl; qf[e + 3] = tq.w * sl;
            }
        }
        kv_f32x4 sc[MAXT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < MAXT; ++kt) {
            if (kt < nkt) {
                float kf[DS];
#pragma unroll
                for (int e = 0; e < DS; e += 4) {
                    const float4 tk = *reinterpret_cast<const float4*>(Kl + kt * 16 * LD + e);
                    kf[e] = tk.x; kf[e + 1] = tk.y; kf[e + 2] = tk.z; kf[e + 3] = tk.w;
                }
                kv_f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < DS; ++e) a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[e], a, 0, 0, 0);
                sc[kt] = a;
                mx = fmaxf(mx, fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])));
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
        kv_f32x4 o[DB], o2[DB];                     // two accumulation chains (keys r even / odd), folded at the end
#pragma unroll
        for (int db = 0; db < DB; ++db) { o[db] = kv_f32x4{0.f, 0.f, 0.f, 0.f}; o2[db] = kv_f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kt = 0; kt < MAXT; ++kt) {
            if (kt < nkt) {
                kv_f32x4 p;
#pragma unroll
                for (int r = 0; r < 4; ++r) { p[r] = __builtin_amdgcn_exp2f(sc[kt][r] - mx); sum += p[r]; }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int db = 0; db < DB; ++db) {
                        if (r & 1) o2[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vl[(kt * 16 + r) * LD + 16 * db], p[r], o2[db], 0, 0, 0);
                        else o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vl[(kt * 16 + r) * LD + 16 * db], p[r], o[db], 0, 0, 0);
                    }
                }
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int db = 0; db < DB; ++db) o[db] += o2[db];
        if (q0 + c < Nq) {
            float* op = out + b * o_bs + (long)(q0 + c) * o_ts + h * D + 4 * g;
#pragma unroll
            for (int db = 0; db < DB; ++db)
                *reinterpret_cast<float4*>(op + 16 * db) = make_float4(o[db][0] * inv, o[db][1] * inv, o[db][2] * inv, o[db][3] * inv);
        }
    }
}

extern "C" int st_attention_kvlds(const float* q, int64_t q_bs, int64_t q_ts, const float* k, int64_t k_bs, int64_t k_ts,
                                  const float* v, int64_t v_bs, int64_t v_ts, float* out, int64_t o_bs, int64_t o_ts,
                                  int32_t B, int32_t heads, int32_t Nq, int32_t Nk, int32_t D, float scale, void* stream) {
    if (!q || !k || !v || !out || B <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0) return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if ((D == 16 || D == 32) && Nk % 16 == 0 && Nk <= 256 && !((q_ts | k_ts | v_ts | o_ts | q_bs | k_bs | v_bs | o_bs) & 3)) {
        const int tpw = Nq >= 8192 ? 4 : 2;                       // 16-query tiles per wave (K/V staging amortised over 128..256 queries)
        const size_t ldsm = (size_t)2 * Nk * (D + 4) * sizeof(float);
        dim3 gridm((Nq + 64 * tpw - 1) / (64 * tpw), heads, B);
        if (D == 16)
            hipLaunchKernelGGL(attention_kv_mfma_kernel<16>, gridm, dim3(256), ldsm, s, q, q_bs, q_ts, k, k_bs, k_ts, v, v_bs, v_ts, out,
                               o_bs, o_ts, Nq, Nk, scale, tpw);
        else
            hipLaunchKernelGGL(attention_kv_mfma_kernel<32>, gridm, dim3(256), ldsm, s, q, q_bs, q_ts, k, k_bs, k_ts, v, v_bs, v_ts, out,
                               o_bs, o_ts, Nq, Nk, scale, tpw);
        ST_CHECK_LAUNCH();
        return ST_OK;
    }
    const size_t lds = (size_t)2 * Nk * D * sizeof(float);
    if (lds > 160 * 1024) return ST_EINVAL;
    dim3 grid((Nq + 511) / 512, heads, B), block(256);
#define LAUNCH_AK(DD)                                                                                                   \
    do {                                                                                                                \
        auto kern = attention_kvlds_kernel<DD>;                                                                         \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kern, grid, block, lds, s, q, q_bs, q_ts, k, k_bs, k_ts, v, v_bs, v_ts, out, o_bs, o_ts, Nq, \
                           Nk, scale);                                                                                  \
    } while (0)
    if (D == 16) LAUNCH_AK(16); else if (D == 32) LAUNCH_AK(32); else return ST_EINVAL;
#undef LAUNCH_AK
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// Locally-grouped (7x7 window) attention, LSA (twins.py:253-304,587-631).  Token (b, y, x) of
// q/k/v lives at base + b*bs + (y*W + x)*ts + h*D.  The grid is zero-padded to a multiple of ws
// *before* the q/k/v projections in the reference, so a padded token's q/k/v is a constant per
// window position: tables qpad/kpad/vpad [ws*ws, heads*D].  One wave per (window, head).
// On the matrix cores, like attention_kv_mfma_kernel: a wave owns one (window, head); the window's ws*ws <= 64 tokens are its keys
// AND its queries.  K / V rows are staged once into a per-wave [64][D+4] slab (lane = token; rows past ws*ws hold finite filler and
// their scores are masked to -inf, so their probabilities are exactly 0), then four 16-query tiles run S^T = K . Q^T (16x16x4 MFMA,
// the query in the lane's column), a register-local softmax in the log2 domain and O^T = V^T . P^T.  The VALU form of this kernel
// issued ~45 instructions per (query, key) pair group on the datapath the MFMA shares (26 of its 34 us at 65 536 tokens x 8 heads).
template <int D>
__global__ __launch_bounds__(256) void window_attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, long bs, long ts,
                                                               const float* __restrict__ qpad, const float* __restrict__ kpad,
                                                               const float* __restrict__ vpad, float* __restrict__ out,
                                                               long o_bs, long o_ts, int H, int W, int heads, int ws,
                                                               float scale) {
    constexpr int LD = D + 4, DS = D / 4, DB = D / 16;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nww = (W + ws - 1) / ws;
    const int win = blockIdx.x, b = blockIdx.z;
    const int h = blockIdx.y * 4 + wave;
    if (h >= heads) return;                                    // whole wave exits together
    const int wy = win / nww, wx = win % nww;
    const int T = ws * ws, C = heads * D;
    float* Ks = sm + (size_t)wave * 2 * 64 * LD;
    float* Vs = Ks + 64 * LD;
    // token t of the window -> its q / k / v row: the image token, or the pad table row of its window position (twins.py:587-600)
    auto row_of = [&](const float* base, const float* pad, int t) {
        const int tt = t < T ? t : 0;                             // (filler rows read token 0: finite, masked below)
        const int ty = tt / ws, tx = tt - ty * ws;
        const int y = wy * ws + ty, x = wx * ws + tx;
        return (y < H && x < W) ? base + b * bs + ((long)y * W + x) * ts + h * D : pad + (size_t)tt * C + h * D;
    };
    {
        // staging: D/4 consecutive lanes fetch the D/4 16-byte pieces of one token's row, so an instruction touches 64 / (D/4) rows of
        // D*4 contiguous bytes (lane = token would touch 64 rows of 16 bytes each: four times the cache-line requests)
        constexpr int PPT = D / 4, TPI = 64 / PPT;              // pieces per token, tokens per instruction
        const int piece = lane % PPT, tl = lane / PPT;
        float4 kr[PPT], vr[PPT];
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int t = i * TPI + tl;
            kr[i] = *reinterpret_cast<const float4*>(row_of(k, kpad, t) + 4 * piece);
            vr[i] = *reinterpret_cast<const float4*>(row_of(v, vpad, t) + 4 * piece);
        }
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int t = i * TPI + tl;
            *reinterpret_cast<float4*>(Ks + t * LD + 4 * piece) = kr[i];
            *reinterpret_cast<float4*>(Vs + t * LD + 4 * piece) = vr[i];
        }
    }
    const int c = lane & 15, g = lane >> 4;
    // the four query tiles' fragments are requested before the slab is read (one memory round trip for the whole wave)
    float qf[4][DS];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float* qp = row_of(q, qpad, 16 * t + c) + g * DS;
        const float sl = scale * 1.44269504088896340736f;        // scores in the log2 domain: the softmax is a bare v_exp_f32
#pragma unroll
        for (int e = 0; e < DS; e += 4) {
            const float4 tq = *reinterpret_cast<const float4*>(qp + e);
            qf[t][e] = tq.x * sl; qf[t][e + 1] = tq.y * sl; qf[t][e + 2] = tq.z * sl; qf[t][e + 3] = tq.w * sl;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS writes visible to its own lanes
    const float* Kl = Ks + c * LD + g * DS;         // + 16*kt*LD: this lane's K fragment (DS consecutive dims)
    const float* Vl = Vs + (4 * g) * LD + c;        // + (16*kt + s)*LD (+16 for the second d block)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (16 * t >= T) break;                     // wave-uniform
        kv_f32x4 sc[4];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            float kf[DS];
#pragma unroll
            for (int e = 0; e < DS; e += 4) {
                const float4 tk = *reinterpret_cast<const float4*>(Kl + kt * 16 * LD + e);
                kf[e] = tk.x; kf[e + 1] = tk.y; kf[e + 2] = tk.z; kf[e + 3] = tk.w;
            }
            kv_f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < DS; ++e) a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[t][e], a, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = (16 * kt + 4 * g + r) < T ? a[r] : -INFINITY;     // keys past the window
            sc[kt] = a;
            mx = fmaxf(mx, fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
        kv_f32x4 o[DB], o2[DB];                     // two accumulation chains (keys r even / odd), folded at the end
#pragma unroll
        for (int db = 0; db < DB; ++db) { o[db] = kv_f32x4{0.f, 0.f, 0.f, 0.f}; o2[db] = kv_f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            kv_f32x4 p;
#pragma unroll
            for (int r = 0; r < 4; ++r) { p[r] = __builtin_amdgcn_exp2f(sc[kt][r] - mx); sum += p[r]; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    if (r & 1) o2[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vl[(kt * 16 + r) * LD + 16 * db], p[r], o2[db], 0, 0, 0);
                    else o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vl[(kt * 16 + r) * LD + 16 * db], p[r], o[db], 0, 0, 0);
                }
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int db = 0; db < DB; ++db) o[db] += o2[db];
        const int tq = 16 * t + c;
        const int ty = tq / ws, tx = tq - ty * ws;
        const int y = wy * ws + ty, x = wx * ws + tx;
        if (tq < T && y < H && x < W) {
            float* op = out + b * o_bs + ((long)y * W + x) * o_ts + h * D + 4 * g;
#pragma unroll
            for (int db = 0; db < DB; ++db)
                *reinterpret_cast<float4*>(op + 16 * db) = make_float4(o[db][0] * inv, o[db][1] * inv, o[db][2] * inv, o[db][3] * inv);
        }
    }
}

extern "C" int st_window_attention(const float* q, const float* k, const float* v, int64_t bs, int64_t ts, const float* qpad,
                                   const float* kpad, const float* vpad, float* out, int64_t o_bs, int64_t o_ts, int32_t B,
                                   int32_t H, int32_t W, int32_t heads, int32_t D, int32_t ws, float scale, void* stream) {
    if (!q || !k || !v || !qpad || !kpad || !vpad || !out || ws * ws > 64 || ws <= 0) return ST_EINVAL;
    const int nwh = (H + ws - 1) / ws, nww = (W + ws - 1) / ws;
    dim3 grid(nwh * nww, (heads + 3) / 4, B), block(256);
    const size_t lds = (size_t)4 * 2 * 64 * (D + 4) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_WA(DD)                                                                                                       \
    do {                                                                                                                    \
        if (lds > 48 * 1024)                                                                                                \
            (void)hipFuncSetAttribute((const void*)window_attention_kernel<DD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(window_attention_kernel<DD>, grid, block, lds, s, q, k, v, bs, ts, qpad, kpad, vpad, out, o_bs, o_ts, H, W, \
                           heads, ws, scale);                                                                               \
    } while (0)
    if (D == 16) LAUNCH_WA(16); else if (D == 32) LAUNCH_WA(32); else return ST_EINVAL;
#undef LAUNCH_WA
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// CCL soft-argmax (network.py:147-199).  G = n1 . n2^T is the plain all-pairs product of the
// L2-normalised features ([B, P, P], P = h*w); the reference's "3x3 patches of n2 as conv filters
// over n1" is vol[p, q] = sum over the 9 offsets d of G[p+d, q+d] (both in bounds).  Softmax over
// q with temperature 10, then the expected displacement.  out [B, P, ldo] = (flow_w, flow_h, 0..).
__global__ __launch_bounds__(256) void ccl_softargmax_kernel(const float* __restrict__ G, float* __restrict__ out, int ldo,
                                                             int h, int w) {
    __shared__ float red[12];
    const int P = h * w, p = blockIdx.x, b = blockIdx.y;
    const int py = p / w, px = p % w;
    const float* Gb = G + (size_t)b * P * P;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    float vol[4];                                   // P <= 1024
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = t + 256 * i;
        float s = -INFINITY;
        if (q < P) {
            const int qy = q / w, qx = q % w;
            s = 0.f;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int ay = py + dy, ax = px + dx, by = qy + dy, bx = qx + dx;
                    const bool ok = ay >= 0 && ay < h && ax >= 0 && ax < w && by >= 0 && by < h && bx >= 0 && bx < w;
                    // unconditional load from a clamped address, then the same ordered sum (s + 0 = s): the 9 taps of the 4 columns are
                    // in flight together instead of 36 dependent round trips
                    const float gv = Gb[(size_t)(min(max(ay, 0), h - 1) * w + min(max(ax, 0), w - 1)) * P + min(max(by, 0), h - 1) * w + min(max(bx, 0), w - 1)];
                    s += ok ? gv : 0.f;
                }
            s *= 10.0f;
        }
        vol[i] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f, fh = 0.f, fw = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = t + 256 * i;
        if (q < P) {
            const float e = expf(vol[i] - mx);
            sum += e;
            fh += e * (float)(q / w - py);
            fw += e * (float)(q % w - px);
        }
    }
    sum = wave_sum(sum); fh = wave_sum(fh); fw = wave_sum(fw);
    if (lane == 0) { red[4 + wv] = sum; red[8 + wv] = fh; }
    __syncthreads();
    const float tsum = (red[4] + red[5]) + (red[6] + red[7]);
    const float tfh = (red[8] + red[9]) + (red[10] + red[11]);
    __syncthreads();
    if (lane == 0) red[wv] = fw;
    __syncthreads();
    if (t == 0) {
        const float tfw = (red[0] + red[1]) + (red[2] + red[3]);
        float* o = out + ((size_t)b * P + p) * ldo;
        o[0] = tfw / tsum; o[1] = tfh / tsum;
        for (int c = 2; c < ldo; ++c) o[c] = 0.f;
    }
}

extern "C" int st_ccl_softargmax(const float* G, float* out, int32_t ldo, int32_t B, int32_t h, int32_t w, void* stream) {
    if (!G || !out || h * w > 1024 || ldo < 2) return ST_EINVAL;
    hipLaunchKernelGGL(ccl_softargmax_kernel, dim3(h * w, B), dim3(256), 0, (hipStream_t)stream, G, out, ldo, h, w);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// strided 2-D copy (channel-slice concatenation)
__global__ void copy2d_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd, int rows, int cols) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)rows * cols) return;
    const int c = idx % cols;
    const size_t r = idx / cols;
    dst[r * ldd + c] = src[r * lds_ + c];
}

extern "C" int st_copy2d(const float* src, int32_t lds_, float* dst, int32_t ldd, int32_t rows, int32_t cols, void* stream) {
    if (!src || !dst || rows <= 0 || cols <= 0) return ST_EINVAL;
    const size_t total = (size_t)rows * cols;
    hipLaunchKernelGGL(copy2d_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, lds_, dst, ldd, rows,
                       cols);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// NCHW image -> channels-last rows [B*H*W, ldo] with the per-branch input scaling
// (flowHomoAdpater.py:55-56  x/127.5 - 1 ;  transformer.py:53-54  2*(x/255) - 1); pad channels = 0.
__global__ void prep_image_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int H, int W, int ldo,
                                  float mul, float div, float sub) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t hw = (size_t)H * W;
    if (idx >= (size_t)B * hw) return;
    const size_t b = idx / hw, r = idx % hw;
    for (int c = 0; c < ldo; ++c) {
        float v = 0.f;
        if (c < C) v = mul * (src[(b * C + c) * hw + r] / div) - sub;
        dst[idx * ldo + c] = v;
    }
}

extern "C" int st_prep_image(const float* src, float* dst, int32_t B, int32_t C, int32_t H, int32_t W, int32_t ldo,
                             float mul, float div, float sub, void* stream) {
    if (!src || !dst || ldo < C) return ST_EINVAL;
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(prep_image_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, B, C, H, W,
                       ldo, mul, div, sub);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// First PatchEmbed convolution (encoder.py:36 Conv2d(1, 16, k6, s2, p2) + ReLU) as a direct
// kernel: the input has ONE channel (each row of the all-pairs volume is a cost map), so the
// implicit-GEMM gather degenerates to scalar loads.  One thread = one output pixel x 16 channels;
// the 36x16 weights are wave-uniform (scalar loads), the 16 outputs leave as four 16-B stores.
// HBM-bound: 4 B read + 64 B written per output pixel.  maps [M, H, W]; w [36, 16]; out [M*Ho*Wo, 16].
__global__ __launch_bounds__(256) void patch_conv1_kernel(const float* __restrict__ maps, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, int M,
                                                          int H, int W, int Ho, int Wo) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)M * Ho * Wo) return;
    const int ox = idx % Wo;
    const int oy = (idx / Wo) % Ho;
    const size_t m = idx / ((size_t)Wo * Ho);
    const float* im = maps + m * (size_t)H * W;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = bias[c];
#pragma unroll
    for (int ky = 0; ky < 6; ++ky) {
        const int iy = oy * 2 - 2 + ky;
#pragma unroll
        for (int kx = 0; kx < 6; ++kx) {
            const int ix = ox * 2 - 2 + kx;
            const float x = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? im[(size_t)iy * W + ix] : 0.f;
            const float* wt = w + (ky * 6 + kx) * 16;
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] = fmaf(x, wt[c], acc[c]);
        }
    }
    float4* o = reinterpret_cast<float4*>(out + idx * 16);
#pragma unroll
    for (int c = 0; c < 4; ++c)
        o[c] = make_float4(fmaxf(acc[4 * c], 0.f), fmaxf(acc[4 * c + 1], 0.f), fmaxf(acc[4 * c + 2], 0.f), fmaxf(acc[4 * c + 3], 0.f));
}

// The same convolution on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32): out[16 px, 16 ch] += A[16 px, 4 taps] .
// W[4 taps, 16 ch], nine MFMAs for the 36 taps.  One workgroup per cost map: the map is staged once into LDS with a zero
// halo (pad 2 left/top, >= 3 right/bottom), every lane gathers its A value -- one input pixel per (output pixel, tap) --
// with a single ds_read_b32 from a running per-tap address; a wave walks output rows, 16 pixels per tile.
// The scalar kernel above spends ~25 address / predicate instructions per tap-FMA group on the VALU that
// v_mfma shares; here a tile costs 9 MFMAs + 9 LDS reads + ~12 VALU.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void patch_conv1_mfma_kernel(const float* __restrict__ maps, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ out, int H,
                                                               int W, int Ho, int Wo, int Hl, int Wl) {
    extern __shared__ __attribute__((aligned(16))) float img[];          // [Hl][Wl], image pixel (y, x) at [y+2][x+2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t m = blockIdx.x;
    const float* im = maps + m * (size_t)H * W;
    // stage the map: interior by 16-B loads (W % 4 == 0; 16 threads per row, all of a thread's loads issued before its LDS
    // writes), halo zeroed separately (disjoint addresses, so one barrier)
    {
        const int ty = tid >> 4, tx = (tid & 15) * 4;
        for (int y0 = 0; y0 < H; y0 += 64) {
            for (int x0 = 0; x0 < W; x0 += 64) {
                float4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int y = y0 + ty + 16 * k, x = x0 + tx;
                    v[k] = (y < H && x < W) ? *reinterpret_cast<const float4*>(im + (size_t)y * W + x) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int y = y0 + ty + 16 * k, x = x0 + tx;
                    if (y < H && x < W) {
                        float2* dst = reinterpret_cast<float2*>(img + (y + 2) * Wl + x + 2);       // Wl even: 8-B aligned
                        dst[0] = make_float2(v[k].x, v[k].y);
                        dst[1] = make_float2(v[k].z, v[k].w);
                    }
                }
            }
        }
        for (int yl = wave; yl < Hl; yl += 4) {
            const bool full = yl < 2 || yl >= H + 2;
            for (int xl = lane; xl < Wl; xl += 64)
                if (full || xl < 2 || xl >= W + 2) img[yl * Wl + xl] = 0.f;
        }
    }
    __syncthreads();
    const int px = lane & 15, ks = lane >> 4;                              // A: pixel px of the tile, k slot ks; B: channel px
    float wv[9];
    int toff[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int t = 4 * j + ks, ky = t / 6, kx = t - ky * 6;
        wv[j] = w[t * 16 + px];
        toff[j] = ky * Wl + kx + 2 * px;
    }
    const float bv = bias[px];
    const int ntx = (Wo + 15) >> 4;
    for (int oy = wave; oy < Ho; oy += 4) {
        for (int tx = 0; tx < ntx; ++tx) {
            const int base = 2 * oy * Wl + 32 * tx;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 9; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(img[base + toff[j]], wv[j], acc, 0, 0, 0);
            // acc[r]: pixel 4*ks + r of the tile, channel px
            float* o = out + ((m * Ho + oy) * (size_t)Wo + 16 * tx + 4 * ks) * 16 + px;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * tx + 4 * ks + r < Wo) o[r * 16] = fmaxf(acc[r] + bv, 0.f);
        }
    }
}

extern "C" int st_patch_conv1(const float* maps, const float* w36x16, const float* bias, float* out, int32_t M, int32_t H,
                              int32_t W, int32_t Ho, int32_t Wo, void* stream) {
    if (!maps || !w36x16 || !bias || !out || M <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return ST_EINVAL;
    // LDS image: rows 0 .. 2*Ho+3, columns 0 .. 2*roundup16(Wo)+3 (the last partial tile reads, but never stores)
    const int Hl = 2 * Ho + 4, Wl = 2 * ((Wo + 15) / 16 * 16) + 4;
    const size_t lds = (size_t)Hl * Wl * sizeof(float);
    if (H + 2 <= Hl && W + 2 <= Wl && lds <= 64 * 1024 && W % 4 == 0 && ((uintptr_t)maps % 16) == 0) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void*)patch_conv1_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(patch_conv1_mfma_kernel, dim3(M), dim3(256), lds, (hipStream_t)stream, maps, w36x16, bias, out, H, W, Ho, Wo,
                           Hl, Wl);
        ST_CHECK_LAUNCH();
        return ST_OK;
    }
    const size_t total = (size_t)M * Ho * Wo;
    hipLaunchKernelGGL(patch_conv1_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, maps, w36x16, bias, out,
                       M, H, W, Ho, Wo);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
