// TPS post-pipeline kernels (reference: core/inference/tps_pipline.py, sample_point_methods.py,
// utils.py, tps_methods/kornia_tps.py).  gfx950 only.  Compiled with -ffp-contract=off: the reference evaluates these
// stages as chains of separate torch-CPU ops (one fp32 rounding each) and the kernels keep exactly those roundings,
// so everything except the TPS solve / kernel sum (an MKL LU and a vectorised reduction in the reference) is bit-exact.
//
//   flow_boxavg_kernel     preprocess (tps_pipline.py:213-244): zero-padded k x k mean, negate, * valid
//   sobel_mag_kernel       |Sobel_x|, |Sobel_y| per channel, channel means, sum (sample_point_methods.py:70-90)
//   range_argmax_kernel    first arg-max of that magnitude inside each border range window (:93-113)
//   gather_points_kernel   plane values at integer points (utils.py:61-68 flow lookup, tps_pipline.py:111-128 mask filter)
//   tps2_solve_kernel      kornia get_tps_transform system [K P; P^T 0] w = [dst; 0], fp64 Gauss-Jordan
//   tps2_warp_kernel       kornia warp_points_tps on the normalised mesh + grid_sample(bilinear, zeros, align_corners=False)
//   minmax_filter_kernel   cv2.erode / cv2.dilate with a k x k rectangle (window clipped to the image = OpenCV's default border)
//   tps_mix_blend_kernel   mask algebra, mix with the flow warp, uint8 blend (tps_pipline.py:139-176)
#include "common.h"
#include "../../include/stitch_gfx950.h"

__device__ __forceinline__ float lin_at2(float start, float end, int n, int i) {       // torch.linspace, fp32 (see geom.hip)
    if (n == 1) return start;
    const float step = (end - start) / (float)(n - 1);
    return (i < n / 2) ? __fmaf_rn(step, (float)i, start) : __fmaf_rn(-step, (float)(n - 1 - i), end);
}

// ---------------------------------------------------------------------------------------------
// F.pad(zeros) + F.avg_pool2d(k, stride 1) = row-major window sum / k^2 (ATen's CPU kernel order, measured bit for bit);
// the following F.interpolate to the same size is the identity.  Then optional negation and * valid.
__global__ __launch_bounds__(256) void flow_boxavg_kernel(const float* __restrict__ in, const float* __restrict__ valid,
                                                          float* __restrict__ out, int planes, int planes_per_b, int H, int W,
                                                          int k, int negate) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), p = blockIdx.z;
    if (x >= W || y >= H) return;
    const float* im = in + (size_t)p * H * W;
    const int r = (k - 1) / 2;
    float s = 0.f;
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            const int yy = y + dy, xx = x + dx;
            s = s + ((yy >= 0 && yy < H && xx >= 0 && xx < W) ? im[(size_t)yy * W + xx] : 0.f);
        }
    float v = s / (float)(k * k);
    if (negate) v = -v;
    if (valid) v = v * valid[(size_t)(p / planes_per_b) * H * W + (size_t)y * W + x];
    out[(size_t)p * H * W + (size_t)y * W + x] = v;
}

extern "C" int st_flow_boxavg(const float* flow, const float* valid, float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                              int32_t k, int32_t negate, void* stream) {
    if (!flow || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || k < 1 || !(k & 1)) return ST_EINVAL;
    dim3 grid((W + 63) / 64, (H + 3) / 4, B * C);
    hipLaunchKernelGGL(flow_boxavg_kernel, grid, dim3(256), 0, (hipStream_t)stream, flow, valid, out, B * C, C, H, W, k, negate);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// grad = mean_c |conv(x_c, Sx)| + mean_c |conv(x_c, Sy)|, zero padding 1; taps accumulated in row-major order (what
// torch's depthwise conv2d does on CPU: measured bit for bit), channel mean = ((c0 + c1) + c2 ...) / C.
__global__ __launch_bounds__(256) void sobel_mag_kernel(const float* __restrict__ img, float* __restrict__ grad, int C, int H, int W) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const float kx[9] = {-1.f, 0.f, 1.f, -2.f, 0.f, 2.f, -1.f, 0.f, 1.f};
    const float ky[9] = {-1.f, -2.f, -1.f, 0.f, 0.f, 0.f, 1.f, 2.f, 1.f};
    float sx = 0.f, sy = 0.f;
    for (int c = 0; c < C; ++c) {
        const float* im = img + (size_t)c * H * W;
        float gx = 0.f, gy = 0.f;
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? im[(size_t)yy * W + xx] : 0.f;
            gx = gx + kx[t] * v;
            gy = gy + ky[t] * v;
        }
        sx = c == 0 ? fabsf(gx) : sx + fabsf(gx);
        sy = c == 0 ? fabsf(gy) : sy + fabsf(gy);
    }
    grad[(size_t)y * W + x] = fabsf(sx / (float)C) + fabsf(sy / (float)C);
}

extern "C" int st_sobel_magnitude(const float* img, float* grad, int32_t C, int32_t H, int32_t W, void* stream) {
    if (!img || !grad || C <= 0 || H <= 0 || W <= 0) return ST_EINVAL;
    hipLaunchKernelGGL(sobel_mag_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, (hipStream_t)stream, img, grad, C, H, W);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// One workgroup per range (x1, y1, x2, y2): arg-max of grad over rows [y1-2, y2+2) x cols [x1-2, x2+2) (clipped to the
// image like a Python slice; starts must be >= 0, i.e. pad_num >= 2), ties -> the first element in row-major order
// (torch.argmax).  Everything outside the window is -1 in the reference, so an empty window returns flat index 0.
__global__ __launch_bounds__(256) void range_argmax_kernel(const float* __restrict__ grad, const int* __restrict__ ranges,
                                                           int* __restrict__ out, int H, int W) {
    const int r = blockIdx.x;
    const int x1 = ranges[4 * r], y1 = ranges[4 * r + 1], x2 = ranges[4 * r + 2], y2 = ranges[4 * r + 3];
    const int xa = x1 - 2, xb = min(x2 + 2, W), ya = y1 - 2, yb = min(y2 + 2, H);
    const int ww = xb - xa, hh = yb - ya;
    float best = -2.f;
    int bidx = 0x7fffffff;
    if (ww > 0 && hh > 0)
        for (int e = threadIdx.x; e < ww * hh; e += 256) {
            const int yy = ya + e / ww, xx = xa + e % ww;
            const float v = grad[(size_t)yy * W + xx];
            const int idx = yy * W + xx;
            if (v > best || (v == best && idx < bidx)) { best = v; bidx = idx; }
        }
    __shared__ float sb[256];
    __shared__ int si[256];
    sb[threadIdx.x] = best; si[threadIdx.x] = bidx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float ov = sb[threadIdx.x + s];
            const int oi = si[threadIdx.x + s];
            if (ov > sb[threadIdx.x] || (ov == sb[threadIdx.x] && oi < si[threadIdx.x])) { sb[threadIdx.x] = ov; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[r] = (si[0] == 0x7fffffff) ? 0 : si[0];
}

extern "C" int st_range_argmax(const float* grad, const int32_t* ranges, int32_t* out_flat_idx, int32_t n_ranges, int32_t H,
                               int32_t W, void* stream) {
    if (!grad || !ranges || !out_flat_idx || n_ranges <= 0 || H <= 0 || W <= 0) return ST_EINVAL;
    hipLaunchKernelGGL(range_argmax_kernel, dim3(n_ranges), dim3(256), 0, (hipStream_t)stream, grad, ranges, out_flat_idx, H, W);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// out[i, p] = planes[p, y_i, x_i]  (points are (x, y) int32; callers guarantee they are inside the image)
__global__ void gather_points_kernel(const float* __restrict__ planes, const int* __restrict__ pts, float* __restrict__ out,
                                     int n, int P, int H, int W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * P) return;
    const int pt = i / P, p = i % P;
    const int x = pts[2 * pt], y = pts[2 * pt + 1];
    out[i] = (x >= 0 && x < W && y >= 0 && y < H) ? planes[((size_t)p * H + y) * W + x] : 0.f;
}

extern "C" int st_gather_points(const float* planes, const int32_t* points_xy, float* out, int32_t n, int32_t P, int32_t H,
                                int32_t W, void* stream) {
    if (!planes || !points_xy || !out || n <= 0 || P <= 0) return ST_EINVAL;
    hipLaunchKernelGGL(gather_points_kernel, dim3((n * P + 255) / 256), dim3(256), 0, (hipStream_t)stream, planes, points_xy, out,
                       n, P, H, W);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// kornia TPS.  U(d2) = 0.5 * d2 * log(d2 + 1e-8), d2(a, b) = clamp(-2 a.b + |a|^2 + |b|^2, 0) evaluated in fp32 in that
// operation order (kornia's _pair_square_euclidean / _kernel_distance).  mode 1 ("pixel"): d2 = |a - b|^2 directly and
// U = d2 * log(d2 + 1.19e-7) -- the plain r^2 log r^2 spline in pixel units that OpenCV's ThinPlateSplineShapeTransformer
// fits (same interpolant; the 1/2 is absorbed by the weights).
__device__ __forceinline__ float tps2_u(float ax, float ay, float bx, float by, int mode) {
    if (mode == 1) {
        const float dx = ax - bx, dy = ay - by;
        const float d2 = dx * dx + dy * dy;
        return d2 * st_logf_cr(d2 + 1.1920929e-7f);
    }
    const float dot = __fmaf_rn(ay, by, ax * bx);                   // [N,2] @ [2,M]: torch's k = 2 contraction
    const float a2 = ax * ax + ay * ay, b2 = bx * bx + by * by;
    float d2 = (-2.0f * dot + a2) + b2;
    d2 = fmaxf(d2, 0.0f);
    return 0.5f * d2 * st_logf_cr(d2 + 1e-8f);
}

// f(A_i) = rhs_i for f(v) = a0 + [ax ay].v + sum_j w_j U(v, Bp_j): L = [[K, P], [P^T, 0]], K_ij = U(A_i, Bp_j), P = [1, A],
// right-hand side [rhs; 0].  kornia's get_tps_transform(points_src = A, points_dst = Bp) puts the kernel centres AND the
// values at Bp (rhs = Bp); the classical spline (OpenCV) has its centres at the sites (Bp = A).  fp64 Gauss-Jordan with partial pivoting on
// work [n+3, n+5]; weights out: kernel [n,2], affine [3,2] fp32.  (The reference solves in fp32 through MKL's blocked LU,
// whose operation order is not reproducible; the fp64 solve is the exact solution of the same fp32 system.)
__global__ __launch_bounds__(256) void tps2_solve_kernel(const float* __restrict__ A, const float* __restrict__ Bp,
                                                         const float* __restrict__ rhs, double* __restrict__ work_g,
                                                         float* __restrict__ kw, float* __restrict__ aw, int n, int mode, int use_lds,
                                                         int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double tps2_lds[];
    double* __restrict__ work = use_lds ? tps2_lds : work_g;       // n <= ~130: the augmented matrix lives in LDS (91 pivot steps
                                                                   // of global round trips cost 1.9 ms; in LDS 0.2 ms)
    const int n3 = n + 3, ld = n + 5;
    __shared__ int s_piv;
    __shared__ double s_pmin, s_pmax;             // smallest / largest pivot magnitude: singular-system detection
    __shared__ double s_best[4];
    __shared__ int s_idx[4];
    __shared__ double s_fac_lds[144];             // per-row elimination factors: LDS mode has n + 3 <= 140
    double* __restrict__ s_fac = use_lds ? s_fac_lds : work_g + (size_t)n3 * ld;   // otherwise behind the matrix ((n+3)*(n+6) scratch)
    for (int e = threadIdx.x; e < n3 * ld; e += 256) {
        const int r = e / ld, c = e % ld;
        double v = 0.0;
        if (r < n) {
            if (c < n) v = tps2_u(A[2 * r], A[2 * r + 1], Bp[2 * c], Bp[2 * c + 1], mode);
            else if (c == n) v = 1.0;
            else if (c < n3) v = A[2 * r + (c - n - 1)];
            else v = rhs[2 * r + (c - n3)];
        } else if (c < n) {
            const int k = r - n;
            v = (k == 0) ? 1.0 : A[2 * c + (k - 1)];
        }
        work[e] = v;
    }
    if (threadIdx.x == 0) { s_pmin = 1e300; s_pmax = 0.0; }
    __syncthreads();
    for (int c = 0; c < n3; ++c) {
        double best = -1.0;
        int bi = c;
        for (int r = c + threadIdx.x; r < n3; r += 256) {
            const double a = fabs(work[(size_t)r * ld + c]);
            if (a > best) { best = a; bi = r; }
        }
        // arg-max |a|, first row on ties (LAPACK's idamax): wave shuffles, then the 4 wave results
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if ((threadIdx.x & 63) == 0) { s_best[threadIdx.x >> 6] = best; s_idx[threadIdx.x >> 6] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double bb = s_best[0];
            int ii = s_idx[0];
            for (int t = 1; t < 4; ++t) if (s_best[t] > bb || (s_best[t] == bb && s_idx[t] < ii)) { bb = s_best[t]; ii = s_idx[t]; }
            s_piv = ii;
            if (!(bb >= s_pmin)) s_pmin = bb;        // (a NaN pivot lands here too)
            if (bb > s_pmax) s_pmax = bb;
        }
        __syncthreads();
        const int piv = s_piv;
        if (piv != c)
            for (int k = threadIdx.x; k < ld; k += 256) {
                const double t = work[(size_t)c * ld + k];
                work[(size_t)c * ld + k] = work[(size_t)piv * ld + k];
                work[(size_t)piv * ld + k] = t;
            }
        __syncthreads();
        // elimination of column c from every other row, parallel over all (row, column) elements of the trailing block:
        // factors first (they read column c, which the update does not touch), then one flat pass
        const double pv = work[(size_t)c * ld + c];
        const double inv = 1.0 / (pv != 0.0 ? pv : 1.0);             // a zero pivot is reported through `status`; keep the sweep finite
        for (int r = threadIdx.x; r < n3; r += 256) s_fac[r] = (r == c) ? 0.0 : work[(size_t)r * ld + c] * inv;
        __syncthreads();
        const int wcols = ld - (c + 1);
        for (int e = threadIdx.x; e < n3 * wcols; e += 256) {
            const int r = e / wcols, k = c + 1 + e % wcols;
            const double f = s_fac[r];
            if (f != 0.0) work[(size_t)r * ld + k] -= f * work[(size_t)c * ld + k];
        }
        __syncthreads();
    }
    for (int r = threadIdx.x; r < n3; r += 256) {
        const double d = work[(size_t)r * ld + r];
        const float wx = (float)(work[(size_t)r * ld + n3] / d), wy = (float)(work[(size_t)r * ld + n3 + 1] / d);
        if (r < n) { kw[2 * r] = wx; kw[2 * r + 1] = wy; }
        else { aw[2 * (r - n)] = wx; aw[2 * (r - n) + 1] = wy; }
    }
    // coincident control points (two equal rows) or fewer than three non-collinear ones: a pivot collapses to rounding level
    if (threadIdx.x == 0 && status) status[0] = (s_pmin == s_pmin && s_pmin > 1e-13 * s_pmax) ? 0 : 1;
}

extern "C" int st_tps2_solve(const float* sites, const float* centers, const float* values, void* work_f64, float* kernel_w,
                             float* affine_w, int32_t n, int32_t mode, int32_t* status, void* stream) {
    if (!sites || !centers || !values || !work_f64 || !kernel_w || !affine_w || n < 3 || n > 4096) return ST_EINVAL;
    const size_t bytes = (size_t)(n + 3) * (n + 5) * sizeof(double);
    const int use_lds = bytes <= 150 * 1024 && n + 3 <= 144;
    if (use_lds && bytes > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)tps2_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(tps2_solve_kernel, dim3(1), dim3(256), use_lds ? bytes : 0, (hipStream_t)stream, sites, centers, values,
                       (double*)work_f64, kernel_w, affine_w, n, mode & 1, use_lds, status);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// warp_image_tps (kornia_tps.py:114-176): coords = create_meshgrid(h, w) in [-1, 1]; warped = sum_i U(coord, center_i) w_i +
// coord . A[1:3] + A[0]; grid_sample(bilinear, zeros, align_corners).  mode 1: coords are pixel indices, the result is a
// source pixel position sampled directly (cv2.remap INTER_LINEAR, constant 0 border).  mode 3 = mode 1 on uint8 data as the
// reference's OpenCV branch sees it (core/inference/tps_methods/opencv_tps.py: `to_pillow_fn` = `.to(torch.uint8)` truncates
// image AND mask before cv2's warpImage, which returns uint8): every tap is truncated toward zero and clamped to 0..255, the
// result is rounded half-to-even and saturated (cv::saturate_cast<uchar>).  cv2's fixed-point interpolation itself (5-bit
// coordinate fractions, 15-bit weights) is not restated: parity vs OpenCV unpinned.
__global__ __launch_bounds__(256) void tps2_warp_kernel(const float* __restrict__ img, const float* __restrict__ centers,
                                                        const float* __restrict__ kw, const float* __restrict__ aw,
                                                        float* __restrict__ out, int C, int H, int W, int n, float kscale,
                                                        float ascale, int align_corners, int mode) {
    extern __shared__ float sh[];                 // centers [n,2], kw [n,2]
    float* s_c = sh;
    float* s_w = sh + 2 * n;
    for (int e = threadIdx.x; e < 2 * n; e += 256) { s_c[e] = centers[e]; s_w[e] = kw[e] * kscale; }
    __syncthreads();
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const bool quant = mode & 2;
    mode &= 1;
    float cx, cy;
    if (mode == 1) { cx = (float)x; cy = (float)y; }
    else {
        cx = (lin_at2(0.f, (float)(W - 1), W, x) / (float)(W - 1) - 0.5f) * 2.0f;
        cy = (lin_at2(0.f, (float)(H - 1), H, y) / (float)(H - 1) - 0.5f) * 2.0f;
    }
    float kx = 0.f, ky = 0.f;
    for (int i = 0; i < n; ++i) {
        const float u = tps2_u(cx, cy, s_c[2 * i], s_c[2 * i + 1], mode);
        kx = kx + u * s_w[2 * i];
        ky = ky + u * s_w[2 * i + 1];
    }
    const float a0x = aw[0] * ascale, a0y = aw[1] * ascale, a1x = aw[2] * ascale, a1y = aw[3] * ascale, a2x = aw[4] * ascale,
                a2y = aw[5] * ascale;
    const float gx = (kx + (cx * a1x + cy * a2x)) + a0x;
    const float gy = (ky + (cx * a1y + cy * a2y)) + a0y;
    float ix, iy;
    if (mode == 1) { ix = gx; iy = gy; }
    else if (align_corners) { ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1); iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1); }
    else { ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f; iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f; }
    const float x0f = floorf(ix), y0f = floorf(iy), x1f = x0f + 1.0f, y1f = y0f + 1.0f;
    const float nw = (x1f - ix) * (y1f - iy), ne = (ix - x0f) * (y1f - iy), sw = (x1f - ix) * (iy - y0f), se = (ix - x0f) * (iy - y0f);
    const bool fin = fabsf(ix) < 1e9f && fabsf(iy) < 1e9f;
    const int x0 = fin ? (int)x0f : -2, y0 = fin ? (int)y0f : -2, x1 = x0 + 1, y1 = y0 + 1;
    const bool xin0 = x0 >= 0 && x0 < W, xin1 = x1 >= 0 && x1 < W, yin0 = y0 >= 0 && y0 < H, yin1 = y1 >= 0 && y1 < H;
    const size_t hw = (size_t)H * W;
    for (int c = 0; c < C; ++c) {
        const float* im = img + (size_t)c * hw;
        float t00 = xin0 && yin0 ? im[(size_t)y0 * W + x0] : 0.f, t01 = xin1 && yin0 ? im[(size_t)y0 * W + x1] : 0.f;
        float t10 = xin0 && yin1 ? im[(size_t)y1 * W + x0] : 0.f, t11 = xin1 && yin1 ? im[(size_t)y1 * W + x1] : 0.f;
        if (quant) {
            t00 = fminf(fmaxf(truncf(t00), 0.f), 255.f); t01 = fminf(fmaxf(truncf(t01), 0.f), 255.f);
            t10 = fminf(fmaxf(truncf(t10), 0.f), 255.f); t11 = fminf(fmaxf(truncf(t11), 0.f), 255.f);
        }
        float v = t00 * nw;
        v = __fmaf_rn(t01, ne, v);
        v = __fmaf_rn(t10, sw, v);
        v = __fmaf_rn(t11, se, v);
        if (!fin) v = 0.f;                               // non-finite coordinate: no tap is in range
        if (quant) v = fminf(fmaxf(rintf(v), 0.f), 255.f);
        out[(size_t)c * hw + (size_t)y * W + x] = v;
    }
}

extern "C" int st_tps2_warp(const float* img, const float* centers, const float* kernel_w, const float* affine_w, float* out,
                            int32_t C, int32_t H, int32_t W, int32_t n, float kernel_scale, float affine_scale,
                            int32_t align_corners, int32_t mode, void* stream) {
    if (!img || !centers || !kernel_w || !affine_w || !out || C <= 0 || H < 2 || W < 2 || n < 1 || n > 3800) return ST_EINVAL;
    const size_t lds = (size_t)4 * n * sizeof(float);
    hipLaunchKernelGGL(tps2_warp_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), lds, (hipStream_t)stream, img, centers,
                       kernel_w, affine_w, out, C, H, W, n, kernel_scale, affine_scale, align_corners, mode);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// 1-D min / max filter of width k along x (axis 0) or y (axis 1), window clipped to the image: two passes = cv2.erode /
// cv2.dilate with a k x k rectangle and OpenCV's default (ignored) border.
__global__ __launch_bounds__(256) void minmax_filter_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                            int k, int is_max, int axis) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), p = blockIdx.z;
    if (x >= W || y >= H) return;
    const float* im = in + (size_t)p * H * W;
    const int r = k / 2;
    float v = is_max ? -INFINITY : INFINITY;
    for (int d = -r; d <= r; ++d) {
        const int xx = axis == 0 ? x + d : x, yy = axis == 0 ? y : y + d;
        if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
        const float t = im[(size_t)yy * W + xx];
        v = is_max ? fmaxf(v, t) : fminf(v, t);
    }
    out[(size_t)p * H * W + (size_t)y * W + x] = v;
}

extern "C" int st_minmax_filter(const float* in, float* out, int32_t planes, int32_t H, int32_t W, int32_t k, int32_t is_max,
                                int32_t axis, void* stream) {
    if (!in || !out || in == out || planes <= 0 || H <= 0 || W <= 0 || k < 1 || !(k & 1)) return ST_EINVAL;
    dim3 grid((W + 63) / 64, (H + 3) / 4, planes);
    hipLaunchKernelGGL(minmax_filter_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, H, W, k, is_max, axis);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// stage 0 (tps_pipline.py:139-141): inv = 1 - (mean_c(warped mask) >= 0.5)                        -> inv [h,w]
// stage 1 (:150-176) with tmask = 1 - open(inv): tps *= tmask; final-warp / image-1 masks; mix; blend uint8.
__global__ void tps_mask_inv_kernel(const float* __restrict__ wm, float* __restrict__ inv, int C, size_t hw) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    float s = wm[p];
    for (int c = 1; c < C; ++c) s = s + wm[(size_t)c * hw + p];
    const float m = (s / (float)C) >= 0.5f ? 1.0f : 0.0f;
    inv[p] = 1.0f - m;
}

__global__ void tps_mix_blend_kernel(float* __restrict__ tps, const float* __restrict__ inv_clean, const float* __restrict__ final_warp,
                                     const float* __restrict__ output1, const float* __restrict__ mask1, float* __restrict__ tmask_o,
                                     float* __restrict__ mix_o, float* __restrict__ mixmask_o, unsigned char* __restrict__ blend,
                                     size_t hw) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const float tmask = 1.0f - inv_clean[p];
    float fw[3], m1[3], t[3];
    for (int c = 0; c < 3; ++c) {
        fw[c] = final_warp[c * hw + p]; m1[c] = mask1[c * hw + p];
        t[c] = tps[c * hw + p] * tmask;                                                   // :151
        tps[c * hw + p] = t[c];
    }
    const float f3 = ((fw[0] >= 3.0f ? 1.f : 0.f) + (fw[1] >= 3.0f ? 1.f : 0.f)) + (fw[2] >= 3.0f ? 1.f : 0.f);
    const float fmask = (f3 / 3.0f) >= 0.5f ? 1.0f : 0.0f;                                // :154-155
    const float i3 = ((1.0f - m1[0]) + (1.0f - m1[1])) + (1.0f - m1[2]);
    const float inv1 = (i3 / 3.0f) >= 0.5f ? 1.0f : 0.0f;                                 // :157-158
    const float mixmask = fmask + (1.0f - fmask) * tmask * inv1;                          // :160
    tmask_o[p] = tmask; mixmask_o[p] = mixmask;
    for (int c = 0; c < 3; ++c) {
        const float mix = fw[c] * fmask + t[c] * (1.0f - fmask) * inv1;                   // :159
        const float o2 = mix * mixmask;                                                   // :165
        mix_o[c * hw + p] = o2;
        float bl = (output1[c * hw + p] * m1[c] + o2 * mixmask) / (m1[c] + mixmask);      // :171
        bl = fminf(fmaxf(bl, 0.0f), 255.0f);
        blend[c * hw + p] = (bl == bl) ? (unsigned char)bl : (unsigned char)0;            // 0/0 = NaN -> 0 like the CPU cast
    }
}

extern "C" int st_tps_mask_inv(const float* warped_mask, float* inv, int32_t C, int32_t h, int32_t w, void* stream) {
    if (!warped_mask || !inv || C <= 0) return ST_EINVAL;
    const size_t hw = (size_t)h * w;
    hipLaunchKernelGGL(tps_mask_inv_kernel, dim3((hw + 255) / 256), dim3(256), 0, (hipStream_t)stream, warped_mask, inv, C, hw);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

extern "C" int st_tps_mix_blend(float* tps3, const float* inv_clean, const float* final_warp3, const float* output1_3,
                                const float* mask1_3, float* tmask, float* mix3, float* mixmask, uint8_t* blend3, int32_t h,
                                int32_t w, void* stream) {
    if (!tps3 || !inv_clean || !final_warp3 || !output1_3 || !mask1_3 || !tmask || !mix3 || !mixmask || !blend3) return ST_EINVAL;
    const size_t hw = (size_t)h * w;
    hipLaunchKernelGGL(tps_mix_blend_kernel, dim3((hw + 255) / 256), dim3(256), 0, (hipStream_t)stream, tps3, inv_clean, final_warp3,
                       output1_3, mask1_3, tmask, mix3, mixmask, blend3, hw);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// =============================================================================================
// mix_fn plug-ins (core/inference/mix_methods/all_img1_with_inpaint.py, inpaint_all_area.py; helpers
// core/inference/utils.py:125-170).  Everything but the neural inpainter: elementwise mask algebra (one fp32 rounding per
// torch op, kept) and box sums with the reference's even-kernel geometry.

// F.conv2d(x, ones(k, k), padding=pad) on one plane, output domain Ho x Wo (the reference crops the H+1 / H+2 rows an even
// kernel produces to [:H, :W]); taps accumulated in row-major order; cmp 0: raw sum, 1: sum == k*k (erosion), 2: sum >= 1.
__global__ __launch_bounds__(256) void box_sum_cmp_kernel(const float* __restrict__ in, int H, int W, float* __restrict__ out, int Ho,
                                                          int Wo, int k, int pad, int cmp) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= Wo || y >= Ho) return;
    float s = 0.f;
    for (int dy = 0; dy < k; ++dy) {
        const int yy = y - pad + dy;
        for (int dx = 0; dx < k; ++dx) {
            const int xx = x - pad + dx;
            s = s + ((yy >= 0 && yy < H && xx >= 0 && xx < W) ? in[(size_t)yy * W + xx] : 0.f);
        }
    }
    float v = s;
    if (cmp == 1) v = (s == (float)(k * k)) ? 1.f : 0.f;
    else if (cmp == 2) v = (s >= 1.0f) ? 1.f : 0.f;
    out[(size_t)y * Wo + x] = v;
}

extern "C" int st_box_sum_cmp(const float* in, int32_t H, int32_t W, float* out, int32_t Ho, int32_t Wo, int32_t k, int32_t pad,
                              int32_t cmp, void* stream) {
    if (!in || !out || in == out || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || k < 1 || pad < 0 || cmp < 0 || cmp > 2) return ST_EINVAL;
    hipLaunchKernelGGL(box_sum_cmp_kernel, dim3((Wo + 63) / 64, (Ho + 3) / 4), dim3(256), 0, (hipStream_t)stream, in, H, W, out, Ho, Wo,
                       k, pad, cmp);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// Elementwise stages.  Planes are [h*w]; "3" = three channel planes.  op:
//  0 dilate_thin_area middle (utils.py:143-146): a = mask, b = dilation -> o0 = thick = clamp(a*b, 0, 1), o1 = thin = a*(1-thick)
//  1 dilate_thin_area end (:158-159):           a = thick, b = dilated thin -> o0 = clamp(a + b, 0, 1), o1 = (o0 >= 1) (uint8 truncation
//                                                of dilate_mask's to_pillow_fn, utils.py:165)
//  2 threshold:                                  o0 = a > thr ? 1 : 0
__global__ void mix_plane_op_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o0, float* __restrict__ o1,
                                    size_t n, int op, float thr) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (op == 0) {
        const float thick = fminf(fmaxf(a[p] * b[p], 0.f), 1.f);
        o0[p] = thick; o1[p] = a[p] * (1.0f - thick);
    } else if (op == 1) {
        const float r = fminf(fmaxf(a[p] + b[p], 0.f), 1.f);
        o0[p] = r; if (o1) o1[p] = (r >= 1.0f) ? 1.f : 0.f;
    } else {
        o0[p] = a[p] > thr ? 1.f : 0.f;
    }
}

extern "C" int st_mix_plane_op(const float* a, const float* b, float* o0, float* o1, int64_t n, int32_t op, float thr, void* stream) {
    if (!a || !o0 || n <= 0 || op < 0 || op > 2 || (op < 2 && !b) || (op == 0 && !o1)) return ST_EINVAL;
    hipLaunchKernelGGL(mix_plane_op_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, b, o0, o1, (size_t)n, op, thr);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// stage A of both plug-ins (all_img1_with_inpaint.py:44-53 / inpaint_all_area.py:43-51): method 0 = all_img1_with_inpaint,
// 1 = inpaint_all_area.  fw, m1, tps: 3 planes; occ, tmask: 1 plane -> tfw3, tfwm3, iam0 (channel 0 of the inpaint-area mask)
__global__ void mix_stage_a_kernel(const float* __restrict__ fw, const float* __restrict__ occ, const float* __restrict__ m1,
                                   const float* __restrict__ tps, const float* __restrict__ tmask, float* __restrict__ tfw,
                                   float* __restrict__ tfwm, float* __restrict__ iam0, size_t hw, int method) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const float oc = occ[p], tm = tmask[p];
    for (int c = 0; c < 3; ++c) {
        const float m = m1[c * hw + p], f = fw[c * hw + p], t = tps[c * hw + p];
        float a, am;
        if (method == 0) {
            const float inv = 1.0f - (m > 0.5f ? 1.0f : 0.0f);
            a = f * oc * m + t * inv;
            am = oc * m + tm * inv;
        } else {
            const float inv = 1.0f - m;
            a = f * oc + t * inv;
            am = oc + tm * inv;
        }
        tfw[c * hw + p] = a; tfwm[c * hw + p] = am;
        if (c == 0) iam0[p] = method == 0 ? (1.0f - am) * m : (1.0f - am) * m * tm;
    }
}

extern "C" int st_mix_stage_a(const float* final_warp3, const float* occ, const float* mask1_3, const float* tps3, const float* tmask,
                              float* tfw3, float* tfwm3, float* iam0, int32_t h, int32_t w, int32_t method, void* stream) {
    if (!final_warp3 || !occ || !mask1_3 || !tps3 || !tmask || !tfw3 || !tfwm3 || !iam0 || method < 0 || method > 1) return ST_EINVAL;
    const size_t hw = (size_t)h * w;
    hipLaunchKernelGGL(mix_stage_a_kernel, dim3((hw + 255) / 256), dim3(256), 0, (hipStream_t)stream, final_warp3, occ, mask1_3, tps3, tmask,
                       tfw3, tfwm3, iam0, hw, method);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// all_img1_with_inpaint.py:58-70,77: iam (thin-area mask), dil (its 7x7 dilation, binary) ->
//   border = |iam - dil|, by1 = (1 - border) * dil * mask1, only_img1 = tfw*(1-by1) + (output1*by1)*by1, other0 = (1 - by1_c0) * border
__global__ void mix_stage_b_kernel(const float* __restrict__ iam, const float* __restrict__ dil, const float* __restrict__ m1,
                                   const float* __restrict__ tfw, const float* __restrict__ o1, float* __restrict__ only_img1,
                                   float* __restrict__ other0, size_t hw) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const float d = dil[p], border = fabsf(iam[p] - d);
    for (int c = 0; c < 3; ++c) {
        const float by1 = (1.0f - border) * d * m1[c * hw + p];
        only_img1[c * hw + p] = tfw[c * hw + p] * (1.0f - by1) + (o1[c * hw + p] * by1) * by1;
        if (c == 0) other0[p] = (1.0f - by1) * border;
    }
}

extern "C" int st_mix_stage_b(const float* iam, const float* dil, const float* mask1_3, const float* tfw3, const float* output1_3,
                              float* only_img1_3, float* other0, int32_t h, int32_t w, void* stream) {
    if (!iam || !dil || !mask1_3 || !tfw3 || !output1_3 || !only_img1_3 || !other0) return ST_EINVAL;
    const size_t hw = (size_t)h * w;
    hipLaunchKernelGGL(mix_stage_b_kernel, dim3((hw + 255) / 256), dim3(256), 0, (hipStream_t)stream, iam, dil, mask1_3, tfw3, output1_3,
                       only_img1_3, other0, hw);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// out3 = clip?(img3, 0, 255) * (inv ? 1 - mask : mask)    (all_img1_with_inpaint.py:82,85,101; clip for the control image)
__global__ void mix_mul_mask_kernel(const float* __restrict__ img, const float* __restrict__ mask, float* __restrict__ out, size_t hw,
                                    int inv, int clip, int use_mask) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const float m = use_mask ? (inv ? 1.0f - mask[p] : mask[p]) : 1.0f;
    for (int c = 0; c < 3; ++c) {
        float v = img[c * hw + p];
        if (clip) v = fminf(fmaxf(v, 0.f), 255.f);
        out[c * hw + p] = use_mask ? v * m : v;
    }
}

extern "C" int st_mix_mul_mask(const float* img3, const float* mask, float* out3, int32_t h, int32_t w, int32_t invert, int32_t clip,
                               void* stream) {
    if (!img3 || !out3) return ST_EINVAL;
    const size_t hw = (size_t)h * w;
    hipLaunchKernelGGL(mix_mul_mask_kernel, dim3((hw + 255) / 256), dim3(256), 0, (hipStream_t)stream, img3, mask, out3, hw, invert, clip,
                       mask ? 1 : 0);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// new_blend_image = clip((output1*mask1 + output2*mask2) / (mask1 + mask2), 0, 255) -> uint8 (tps_pipline.py:186-187);
// mask2 has c2 (1 or 3) planes.
__global__ void blend_pair_kernel(const float* __restrict__ o1, const float* __restrict__ m1, const float* __restrict__ o2,
                                  const float* __restrict__ m2, int c2, unsigned char* __restrict__ blend, size_t hw) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    for (int c = 0; c < 3; ++c) {
        const float mm2 = m2[(c2 == 3 ? c : 0) * hw + p], mm1 = m1[c * hw + p];
        float bl = (o1[c * hw + p] * mm1 + o2[c * hw + p] * mm2) / (mm1 + mm2);
        bl = fminf(fmaxf(bl, 0.0f), 255.0f);
        blend[c * hw + p] = (bl == bl) ? (unsigned char)bl : (unsigned char)0;
    }
}

extern "C" int st_blend_pair(const float* output1_3, const float* mask1_3, const float* output2_3, const float* mask2, int32_t mask2_planes,
                             uint8_t* blend3, int32_t h, int32_t w, void* stream) {
    if (!output1_3 || !mask1_3 || !output2_3 || !mask2 || !blend3 || (mask2_planes != 1 && mask2_planes != 3)) return ST_EINVAL;
    const size_t hw = (size_t)h * w;
    hipLaunchKernelGGL(blend_pair_kernel, dim3((hw + 255) / 256), dim3(256), 0, (hipStream_t)stream, output1_3, mask1_3, output2_3, mask2,
                       mask2_planes, blend3, hw);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
