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

/* ------------------------------------------------------------------------------------------------
 * Small dense linear algebra exactly as the reference's torch-CPU calls evaluate it (measured bit for
 * bit against torch 2.10 / oneMKL 2024.2 in the build container on thousands of matrices, see
 * tests/test_oracle_pin.py::test_small_linalg_*):
 *
 *   torch.inverse(A)  (core/udis_utils/torch_DLT.py:42, core/flowHomoAdpater.py:105,112, core/warp_utils.py:24)
 *     = linalg.solve(A, I): sgetrf of A^T (a C-contiguous A is handed to LAPACK as its transpose), then
 *       sgetrs('T') against the identity: U^T y = e_c, L^T z = y, rows swapped back in reverse pivot order.
 *       n = 8: right-looking LU, column scaled by the pivot's reciprocal, trailing update fmaf(-l, u, a);
 *              both triangular solves are dot products over k whose 8 products (NOT fused) sit in
 *              8 SIMD lanes (lane = k) and are reduced as ((l0+l4)+(l2+l6)) + ((l1+l5)+(l3+l7)); U's
 *              diagonal is divided by.
 *       n = 3: column 0 scaled by the reciprocal, column 1 divided; U's diagonal multiplied by its
 *              reciprocal; y_i = (b_i - sum of unfused products) * r_i; x1 = fmaf(-l21, x2, y1);
 *              x0 = y0 - fmaf(l10, x1, l20 * x2).
 *   torch.matmul of small batched matrices ([8x8]@[8x1] torch_DLT.py:43, [3x3]@[3x3]
 *   flowHomoAdpater.py:108,112,226,291,306-307): contraction*rows*cols < 400 takes ATen's plain
 *   baddbmm loop: acc = 0; acc += a[i][k]*b[k][j] for ascending k, products not fused.
 * ---------------------------------------------------------------------------------------------- */
static float lanes8_sum(const float *l) {
    float a0 = l[0] + l[4], a1 = l[1] + l[5], a2 = l[2] + l[6], a3 = l[3] + l[7];
    float b0 = a0 + a2, b1 = a1 + a3;
    return b0 + b1;
}

/* torch.inverse of a row-major 8x8 */
void orc_inv8(const float *Ain, float *out) {
    enum { n = 8 };
    float A[n][n], B[n][n];
    int piv[n];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i][j] = Ain[j * n + i];
    for (int k = 0; k < n; ++k) {
        int p = k;
        float best = fabsf(A[k][k]);
        for (int i = k + 1; i < n; ++i) if (fabsf(A[i][k]) > best) { best = fabsf(A[i][k]); p = i; }
        piv[k] = p;
        if (p != k) for (int j = 0; j < n; ++j) { float t = A[k][j]; A[k][j] = A[p][j]; A[p][j] = t; }
        float r = 1.0f / A[k][k];
        for (int i = k + 1; i < n; ++i) A[i][k] = A[i][k] * r;
        for (int i = k + 1; i < n; ++i) for (int j = k + 1; j < n; ++j) A[i][j] = fmaf(-A[i][k], A[k][j], A[i][j]);
    }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) B[i][j] = (i == j) ? 1.0f : 0.0f;
    for (int c = 0; c < n; ++c) {
        for (int i = 0; i < n; ++i) {
            float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = 0; k < i; ++k) l[k] = A[k][i] * B[k][c];
            B[i][c] = (B[i][c] - lanes8_sum(l)) / A[i][i];
        }
        for (int i = n - 1; i >= 0; --i) {
            float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = i + 1; k < n; ++k) l[k] = A[k][i] * B[k][c];
            B[i][c] = B[i][c] - lanes8_sum(l);
        }
    }
    for (int k = n - 1; k >= 0; --k)
        if (piv[k] != k) for (int j = 0; j < n; ++j) { float t = B[k][j]; B[k][j] = B[piv[k]][j]; B[piv[k]][j] = t; }
    memcpy(out, B, sizeof(B));
}

/* torch.inverse of a row-major 3x3 */
void orc_inv3(const float *Ain, float *out) {
    float A[3][3], B[3][3];
    int piv[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = Ain[j * 3 + i];
    for (int k = 0; k < 3; ++k) {
        int p = k;
        float best = fabsf(A[k][k]);
        for (int i = k + 1; i < 3; ++i) if (fabsf(A[i][k]) > best) { best = fabsf(A[i][k]); p = i; }
        piv[k] = p;
        if (p != k) for (int j = 0; j < 3; ++j) { float t = A[k][j]; A[k][j] = A[p][j]; A[p][j] = t; }
        if (k == 0) { float r = 1.0f / A[0][0]; A[1][0] = A[1][0] * r; A[2][0] = A[2][0] * r; }
        else if (k == 1) A[2][1] = A[2][1] / A[1][1];
        for (int i = k + 1; i < 3; ++i) for (int j = k + 1; j < 3; ++j) A[i][j] = fmaf(-A[i][k], A[k][j], A[i][j]);
    }
    float r0 = 1.0f / A[0][0], r1 = 1.0f / A[1][1], r2 = 1.0f / A[2][2];
    for (int c = 0; c < 3; ++c) {
        float b0 = (c == 0), b1 = (c == 1), b2 = (c == 2);
        float y0 = b0 * r0;
        float y1 = (b1 - A[0][1] * y0) * r1;
        float y2 = (b2 - (A[0][2] * y0 + A[1][2] * y1)) * r2;
        float x2 = y2;
        float x1 = fmaf(-A[2][1], x2, y1);
        float x0 = y0 - fmaf(A[1][0], x1, A[2][0] * x2);
        B[0][c] = x0; B[1][c] = x1; B[2][c] = x2;
    }
    for (int k = 2; k >= 0; --k)
        if (piv[k] != k) for (int j = 0; j < 3; ++j) { float t = B[k][j]; B[k][j] = B[piv[k]][j]; B[piv[k]][j] = t; }
    memcpy(out, B, sizeof(B));
}

/* small torch.matmul: [n x m] @ [m x p], plain ascending-k accumulation from zero */
void orc_matmul_small(const float *A, const float *Bm, float *out, int n, int m, int p) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < p; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < m; ++k) acc = acc + A[i * m + k] * Bm[k * p + j];
            out[i * p + j] = acc;
        }
}

/* tensor_DLT (core/udis_utils/torch_DLT.py:17-45): src, dst [4][2] -> H [9] (H[8] = 1).
 * Rows (x y 1 0 0 0 -u*x -u*y), (0 0 0 x y 1 -v*x -v*y); rhs = (u, v) interleaved; h = inverse(A) @ b. */
void orc_dlt4(const float *src, const float *dst, float *H) {
    float A[64], Ainv[64], b[8];
    for (int p = 0; p < 4; ++p) {
        float x = src[2 * p], y = src[2 * p + 1], u = dst[2 * p], v = dst[2 * p + 1];
        float *r0 = A + 16 * p, *r1 = r0 + 8;
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -(u * x); r0[7] = -(u * y);
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -(v * x); r1[7] = -(v * y);
        b[2 * p] = u; b[2 * p + 1] = v;
    }
    orc_inv8(A, Ainv);
    orc_matmul_small(Ainv, b, H, 8, 8, 1);
    H[8] = 1.0f;
}


/* fp64 inverse of a row-major n x n matrix by Gauss-Jordan elimination with partial pivoting: the TPS system of
 * core/udis_utils/torch_tps_transform.py:173 (`torch.inverse(W)` on an (N+3)^2 fp64 matrix).  The reference's result is
 * LAPACK dgetrf + dgetri; in fp64 the two differ by ~1e-13 relative, which the cast of T to fp32 (:185) absorbs except for
 * last-bit ties -- the test bound on T is 1e-6 relative.  Plain C on purpose: MKL's threaded batched dgetrf returned
 * inconsistent pivots on a GPU box ("Intel oneMKL ERROR: Parameter 6 was incorrect on entry to DLASWP", round 2), which
 * made the checker itself flaky.  work: n * 2n doubles.  Returns 0, or 1 if a pivot is exactly zero.               */
int orc_inv_f64(const double *A, double *out, int n, double *work) {
    const int ld = 2 * n;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) { work[i * ld + j] = A[i * n + j]; work[i * ld + n + j] = (i == j) ? 1.0 : 0.0; }
    }
    for (int c = 0; c < n; ++c) {
        int p = c;
        double best = fabs(work[c * ld + c]);
        for (int r = c + 1; r < n; ++r) {
            const double v = fabs(work[r * ld + c]);
            if (v > best) { best = v; p = r; }
        }
        if (best == 0.0) return 1;
        if (p != c)
            for (int j = 0; j < ld; ++j) { const double t = work[c * ld + j]; work[c * ld + j] = work[p * ld + j]; work[p * ld + j] = t; }
        const double inv = 1.0 / work[c * ld + c];
        for (int j = 0; j < ld; ++j) work[c * ld + j] *= inv;
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = work[r * ld + c];
            if (f == 0.0) continue;
            for (int j = 0; j < ld; ++j) work[r * ld + j] -= f * work[c * ld + j];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) out[i * n + j] = work[i * ld + n + j];
    return 0;
}
