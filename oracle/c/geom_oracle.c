/* CPU oracle (plain C) for the integer / index arithmetic of the stitching hot path.
 * TEST INFRASTRUCTURE ONLY: linked by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg through ctypes; never by the product.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/build_oracle.py).  Contraction is
 * off so that every fp32 rounding below is explicit; the only fused operations are the
 * fmaf() calls, which restate what the reference's torch-CPU kernels do (measured in the build
 * container, see tests/test_oracle_pin.py):
 *   - torch.linspace (fp32, CPU)  : i <  n/2 -> fmaf(step, i, start)
 *                                   i >= n/2 -> fmaf(-step, n-1-i, end),  step = (end-start)/(n-1)
 *   - torch.matmul [3x3]@[3xN]    : acc = t0*gx; acc = fmaf(t1, gy, acc); acc = acc + t2
 *
 * Each function cites the reference file:line it follows (paths relative to /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* torch.linspace(start, end, n) in fp32 -- used by core/udis_utils/torch_homo_transform.py:96-99 */
void orc_linspace(float start, float end, int n, float *out) {
    if (n == 1) { out[0] = start; return; }
    float step = (end - start) / (float)(n - 1);
    int half = n / 2;
    for (int i = 0; i < n; ++i)
        out[i] = (i < half) ? fmaf(step, (float)i, start) : fmaf(-step, (float)(n - 1 - i), end);
}

/* x86 cvttss2si semantics of `.int()` on an out-of-range / NaN float (torch CPU): INT_MIN */
static int32_t f2i_x86(float v) {
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return INT32_MIN;
    return (int32_t)v;
}

static int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* One output pixel of the homography transformer: normalised target (gx, gy) -> source sample.
 * Follows core/udis_utils/torch_homo_transform.py:114-141 (_transform) and :29-41,86-89
 * (_interpolate).  Writes the 4 clamped integer neighbours and the 4 weights. */
static void homo_sample(const float *th, float gx, float gy, int W, int H,
                        int32_t *x0, int32_t *x1, int32_t *y0, int32_t *y1, float w[4]) {
    float xs = th[0] * gx; xs = fmaf(th[1], gy, xs); xs = xs + th[2];
    float ys = th[3] * gx; ys = fmaf(th[4], gy, ys); ys = ys + th[5];
    float ts = th[6] * gx; ts = fmaf(th[7], gy, ts); ts = ts + th[8];
    float ge = (fabsf(ts) >= 1e-7f) ? 1.0f : 0.0f;          /* :134-137 */
    float smallers = 1e-6f * (1.0f - ge);
    ts = ts + smallers;
    float x = xs / ts, y = ys / ts;                           /* :140-141 */
    x = (x + 1.0f) * (float)W / 2.0f;                         /* :29-30 */
    y = (y + 1.0f) * (float)H / 2.0f;
    int32_t ix0 = f2i_x86(floorf(x)), iy0 = f2i_x86(floorf(y));
    int32_t ix1 = (int32_t)((uint32_t)ix0 + 1u), iy1 = (int32_t)((uint32_t)iy0 + 1u);
    ix0 = clampi(ix0, 0, W - 1); ix1 = clampi(ix1, 0, W - 1); /* :38-41 */
    iy0 = clampi(iy0, 0, H - 1); iy1 = clampi(iy1, 0, H - 1);
    float x0f = (float)ix0, x1f = (float)ix1, y0f = (float)iy0, y1f = (float)iy1;
    w[0] = (x1f - x) * (y1f - y);                              /* wa : (y0,x0)  :86-89 */
    w[1] = (x1f - x) * (y - y0f);                              /* wb : (y1,x0) */
    w[2] = (x - x0f) * (y1f - y);                              /* wc : (y0,x1) */
    w[3] = (x - x0f) * (y - y0f);                              /* wd : (y1,x1) */
    *x0 = ix0; *x1 = ix1; *y0 = iy0; *y1 = iy1;
}

/* Homography spatial transformer (core/udis_utils/torch_homo_transform.py:5-151).
 *   U     [B,C,H,W] fp32 NCHW, theta [B,9], out [B,C,oh,ow];
 *   idx   optional [B,oh,ow,4] int32 = (x0,x1,y0,y1) clamped neighbours (bit-exact contract). */
void orc_homo_warp(const float *U, const float *theta, float *out, int32_t *idx,
                   int B, int C, int H, int W, int oh, int ow) {
    float *lx = (float *)malloc(sizeof(float) * ow), *ly = (float *)malloc(sizeof(float) * oh);
    orc_linspace(-1.0f, 1.0f, ow, lx);
    orc_linspace(-1.0f, 1.0f, oh, ly);
    for (int b = 0; b < B; ++b) {
        const float *th = theta + 9 * b;
        for (int i = 0; i < oh; ++i)
            for (int j = 0; j < ow; ++j) {
                int32_t x0, x1, y0, y1; float w[4];
                homo_sample(th, lx[j], ly[i], W, H, &x0, &x1, &y0, &y1, w);
                if (idx) {
                    int32_t *p = idx + (((size_t)b * oh + i) * ow + j) * 4;
                    p[0] = x0; p[1] = x1; p[2] = y0; p[3] = y1;
                }
                if (!out) continue;
                for (int c = 0; c < C; ++c) {
                    const float *im = U + ((size_t)b * C + c) * H * W;
                    float Ia = im[(size_t)y0 * W + x0], Ib = im[(size_t)y1 * W + x0];
                    float Ic = im[(size_t)y0 * W + x1], Id = im[(size_t)y1 * W + x1];
                    float v = w[0] * Ia;                       /* :90  wa*Ia+wb*Ib+wc*Ic+wd*Id */
                    v = v + w[1] * Ib; v = v + w[2] * Ic; v = v + w[3] * Id;
                    out[(((size_t)b * C + c) * oh + i) * ow + j] = v;
                }
            }
    }
    free(lx); free(ly);
}

/* Range map = bilinear forward splat of ones along `flow` (core/warp_utils.py:114-175).
 * flow [B,2,H,W]; out [B,H,W].  The reference accumulates with scatter_add_ (fp32, order =
 * tap-major then raster); the sum here is taken in double so the result is order-free. */
void orc_range_map(const float *flow, float *out, int B, int H, int W) {
    double *acc = (double *)calloc((size_t)H * W, sizeof(double));
    for (int b = 0; b < B; ++b) {
        memset(acc, 0, sizeof(double) * H * W);
        const float *fx = flow + (size_t)b * 2 * H * W, *fy = fx + (size_t)H * W;
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                float cx = (float)j + fx[(size_t)i * W + j];  /* flow_to_warp :54-69 */
                float cy = (float)i + fy[(size_t)i * W + j];
                float flx = floorf(cx), fly = floorf(cy);
                float ox = cx - flx, oy = cy - fly;            /* :124-126 */
                int32_t x0 = f2i_x86(flx), y0 = f2i_x86(fly);
                for (int di = 0; di < 2; ++di)
                    for (int dj = 0; dj < 2; ++dj) {
                        int64_t xi = (int64_t)x0 + di, yj = (int64_t)y0 + dj;
                        if (xi < 0 || xi >= W || yj < 0 || yj >= H) continue;      /* :154-157 */
                        float wi = di ? ox : (1.0f - ox);      /* :162-163 */
                        float wj = dj ? oy : (1.0f - oy);
                        acc[(size_t)yj * W + xi] += (double)(wi * wj);
                    }
            }
        for (size_t k = 0; k < (size_t)H * W; ++k) out[(size_t)b * H * W + k] = (float)acc[k];
    }
    free(acc);
}

/* preprocess_occlusion_mask (core/flowHomoAdpater.py:18-35): threshold >= 0.5, then
 * morphological open with a ksz x ksz box (zero padded): erosion = (box sum == ksz*ksz),
 * dilation = (box sum of erosion >= 1).  m, out: [N,H,W] (N = batch*channels, depthwise). */
void orc_morph_open(const float *m, float *out, int N, int H, int W, int ksz) {
    int r = ksz / 2;
    uint8_t *bin = (uint8_t *)malloc((size_t)H * W), *ero = (uint8_t *)malloc((size_t)H * W);
    for (int n = 0; n < N; ++n) {
        const float *src = m + (size_t)n * H * W;
        for (size_t k = 0; k < (size_t)H * W; ++k) bin[k] = src[k] >= 0.5f;
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                int s = 0;
                for (int a = -r; a <= r; ++a)
                    for (int c = -r; c <= r; ++c) {
                        int y = i + a, x = j + c;
                        if (y >= 0 && y < H && x >= 0 && x < W) s += bin[(size_t)y * W + x];
                    }
                ero[(size_t)i * W + j] = (s == ksz * ksz);
            }
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                int s = 0;
                for (int a = -r; a <= r && !s; ++a)
                    for (int c = -r; c <= r; ++c) {
                        int y = i + a, x = j + c;
                        if (y >= 0 && y < H && x >= 0 && x < W && ero[(size_t)y * W + x]) { s = 1; break; }
                    }
                out[(size_t)n * H * W + (size_t)i * W + j] = s ? 1.0f : 0.0f;
            }
    }
    free(bin); free(ero);
}
