// Geometric stage of the stitching path on gfx950: DLT solve, homography / TPS spatial
// transformers (bit-exact integer sample indices), flow warp (grid_sample), bilinear resize,
// range-map splat, morphological open, mask algebra + blend.
//
// THIS FILE IS COMPILED WITH -ffp-contract=off: every fp32 rounding is spelled out so the integer
// sample indices match the CPU oracle (oracle/c/geom_oracle.c) bit for bit.  Fused operations appear
// only as explicit __fmaf_rn(), restating what the reference's torch-CPU kernels do:
//   torch.linspace : i < n/2 ? fma(step, i, start) : fma(-step, n-1-i, end)
//   [3x3]@[3xN]    : acc = t0*gx; acc = fma(t1, gy, acc); acc = acc + t2
#include "common.h"
#include "../../include/stitch_gfx950.h"

__device__ __forceinline__ float lin_at(float start, float end, int n, int i) {
    if (n == 1) return start;
    const float step = (end - start) / (float)(n - 1);
    return (i < n / 2) ? __fmaf_rn(step, (float)i, start) : __fmaf_rn(-step, (float)(n - 1 - i), end);
}

// x86 cvttss2si semantics of torch's float -> int32 `.int()`: out-of-range / NaN -> INT_MIN
__device__ __forceinline__ int f2i_x86(float v) {
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return (int)0x80000000;
    return (int)v;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct Tap4 { int x0, x1, y0, y1; float wa, wb, wc, wd; };

// `_interpolate` of torch_homo_transform.py:13-92 / torch_tps_transform.py:18-94 for one sample
__device__ __forceinline__ Tap4 taps_from_normalised(float xn, float yn, int W, int H) {
    Tap4 t;
    const float x = (xn + 1.0f) * (float)W / 2.0f;
    const float y = (yn + 1.0f) * (float)H / 2.0f;
    int ix0 = f2i_x86(floorf(x)), iy0 = f2i_x86(floorf(y));
    int ix1 = (int)((unsigned)ix0 + 1u), iy1 = (int)((unsigned)iy0 + 1u);
    ix0 = clampi(ix0, 0, W - 1); ix1 = clampi(ix1, 0, W - 1);
    iy0 = clampi(iy0, 0, H - 1); iy1 = clampi(iy1, 0, H - 1);
    const float x0f = (float)ix0, x1f = (float)ix1, y0f = (float)iy0, y1f = (float)iy1;
    t.wa = (x1f - x) * (y1f - y);
    t.wb = (x1f - x) * (y - y0f);
    t.wc = (x - x0f) * (y1f - y);
    t.wd = (x - x0f) * (y - y0f);
    t.x0 = ix0; t.x1 = ix1; t.y0 = iy0; t.y1 = iy1;
    return t;
}

// ---------------------------------------------------------------------------------------------
// Homography spatial transformer (core/udis_utils/torch_homo_transform.py:5-151).
//   U [B,C,H,W] NCHW, theta [B,9]; out [B, C+n_ones, oh, ow]: the last n_ones channels are the
//   warp of an all-ones image (the reference concatenates torch.ones_like() before the call);
//   idx (optional) [B,oh,ow,4] int32 = (x0,x1,y0,y1).
__global__ __launch_bounds__(256) void homo_warp_kernel(const float* __restrict__ U, const float* __restrict__ theta,
                                                        float* __restrict__ out, int* __restrict__ idx, int C, int n_ones,
                                                        int H, int W, int oh, int ow) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (i >= oh || j >= ow) return;
    const float* th = theta + 9 * b;
    const float gx = lin_at(-1.0f, 1.0f, ow, j), gy = lin_at(-1.0f, 1.0f, oh, i);
    float xs = th[0] * gx; xs = __fmaf_rn(th[1], gy, xs); xs = xs + th[2];
    float ys = th[3] * gx; ys = __fmaf_rn(th[4], gy, ys); ys = ys + th[5];
    float ts = th[6] * gx; ts = __fmaf_rn(th[7], gy, ts); ts = ts + th[8];
    const float ge = (fabsf(ts) >= 1e-7f) ? 1.0f : 0.0f;
    ts = ts + 1e-6f * (1.0f - ge);
    const Tap4 t = taps_from_normalised(xs / ts, ys / ts, W, H);
    const size_t opix = (size_t)i * ow + j, ohw = (size_t)oh * ow;
    if (idx) {
        int* p = idx + (((size_t)b * oh + i) * ow + j) * 4;
        p[0] = t.x0; p[1] = t.x1; p[2] = t.y0; p[3] = t.y1;
    }
    if (!out) return;
    const size_t ia = (size_t)t.y0 * W + t.x0, ib = (size_t)t.y1 * W + t.x0;
    const size_t ic = (size_t)t.y0 * W + t.x1, id = (size_t)t.y1 * W + t.x1;
    float* ob = out + (size_t)b * (C + n_ones) * ohw + opix;
    for (int c = 0; c < C; ++c) {
        const float* im = U + ((size_t)b * C + c) * H * W;
        float v = t.wa * im[ia];
        v = v + t.wb * im[ib]; v = v + t.wc * im[ic]; v = v + t.wd * im[id];
        ob[(size_t)c * ohw] = v;
    }
    if (n_ones > 0) {
        float v = t.wa * 1.0f;
        v = v + t.wb * 1.0f; v = v + t.wc * 1.0f; v = v + t.wd * 1.0f;
        for (int c = 0; c < n_ones; ++c) ob[(size_t)(C + c) * ohw] = v;
    }
}

extern "C" int st_homo_warp(const float* U, const float* theta, float* out, int32_t* idx, int32_t B, int32_t C,
                            int32_t n_ones, int32_t H, int32_t W, int32_t oh, int32_t ow, void* stream) {
    if (!theta || (!out && !idx) || (C > 0 && !U) || B <= 0 || H <= 0 || W <= 0 || oh <= 0 || ow <= 0) return ST_EINVAL;
    dim3 grid((ow + 63) / 64, (oh + 3) / 4, B);
    hipLaunchKernelGGL(homo_warp_kernel, grid, dim3(256), 0, (hipStream_t)stream, U, theta, out, idx, C, n_ones, H, W, oh, ow);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// 3x3 helpers (tiny; one thread per batch element).  The reference evaluates these with torch-CPU calls whose
// operation order is restated in oracle/c/geom_oracle.c (orc_inv3 / orc_inv8 / orc_matmul_small, pinned bit for
// bit to the reference golden); this file is compiled with -ffp-contract=off, so a*b+c below is two roundings and
// only __fmaf_rn is fused.
// small torch.matmul (flowHomoAdpater.py:108,112,226,291,306-307): acc = 0; acc += a*b for ascending k, unfused.
__device__ __forceinline__ void mat3_mul(const float* A, const float* Bm, float* o) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            float acc = A[r * 3] * Bm[c];
            acc = acc + A[r * 3 + 1] * Bm[3 + c];
            acc = acc + A[r * 3 + 2] * Bm[6 + c];
            o[r * 3 + c] = acc;
        }
}
// torch.inverse of a 3x3 (flowHomoAdpater.py:112, warp_utils.py:24) = sgetrf(A^T) + sgetrs('T', I) in MKL's order:
// column 0 scaled by the pivot's reciprocal, column 1 divided, fused trailing updates; U's diagonal by reciprocal.
__device__ __forceinline__ void mat3_inv(const float* Ain, float* o) {
    float A[3][3], Bm[3][3];
    int piv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) A[i][j] = Ain[j * 3 + i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int p = k;
        float best = fabsf(A[k][k]);
#pragma unroll
        for (int i = k + 1; i < 3; ++i) if (fabsf(A[i][k]) > best) { best = fabsf(A[i][k]); p = i; }
        piv[k] = p;
#pragma unroll
        for (int q = 1; q < 3; ++q)
            if (q > k && p == q) {
#pragma unroll
                for (int j = 0; j < 3; ++j) { const float t = A[k][j]; A[k][j] = A[q][j]; A[q][j] = t; }
            }
        if (k == 0) { const float r = 1.0f / A[0][0]; A[1][0] = A[1][0] * r; A[2][0] = A[2][0] * r; }
        else if (k == 1) A[2][1] = A[2][1] / A[1][1];
#pragma unroll
        for (int i = k + 1; i < 3; ++i)
#pragma unroll
            for (int j = k + 1; j < 3; ++j) A[i][j] = __fmaf_rn(-A[i][k], A[k][j], A[i][j]);
    }
    const float r0 = 1.0f / A[0][0], r1 = 1.0f / A[1][1], r2 = 1.0f / A[2][2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float b0 = (c == 0) ? 1.f : 0.f, b1 = (c == 1) ? 1.f : 0.f, b2 = (c == 2) ? 1.f : 0.f;
        const float y0 = b0 * r0;
        const float y1 = (b1 - A[0][1] * y0) * r1;
        const float y2 = (b2 - (A[0][2] * y0 + A[1][2] * y1)) * r2;
        const float x1 = __fmaf_rn(-A[2][1], y2, y1);
        const float x0 = y0 - __fmaf_rn(A[1][0], x1, A[2][0] * y2);
        Bm[0][c] = x0; Bm[1][c] = x1; Bm[2][c] = y2;
    }
#pragma unroll
    for (int k = 2; k >= 0; --k)
#pragma unroll
        for (int q = 1; q < 3; ++q)
            if (q > k && piv[k] == q) {
#pragma unroll
                for (int j = 0; j < 3; ++j) { const float t = Bm[k][j]; Bm[k][j] = Bm[q][j]; Bm[q][j] = t; }
            }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) o[i * 3 + j] = Bm[i][j];
}

// out[b] = L @ (invert ? inv(X[b]) : X[b]) @ R   with L, R shared 3x3 (flowHomoAdpater.py:108,112,226,307)
__global__ void mat3_sandwich_kernel(const float* __restrict__ L, const float* __restrict__ X, const float* __restrict__ R,
                                     float* __restrict__ out, int B, int invert) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float l[9], r[9], x[9], t[9], u[9];
    for (int k = 0; k < 9; ++k) { l[k] = L[k]; r[k] = R[k]; x[k] = X[9 * b + k]; }
    if (invert) { mat3_inv(x, t); for (int k = 0; k < 9; ++k) x[k] = t[k]; }
    mat3_mul(l, x, t);
    mat3_mul(t, r, u);
    for (int k = 0; k < 9; ++k) out[9 * b + k] = u[k];
}

extern "C" int st_mat3_sandwich(const float* L, const float* X, const float* R, float* out, int32_t B, int32_t invert,
                                void* stream) {
    if (!L || !X || !R || !out || B <= 0) return ST_EINVAL;
    hipLaunchKernelGGL(mat3_sandwich_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, L, X, R, out, B, invert);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// 4-point DLT (core/udis_utils/torch_DLT.py:17-45): src/dst [B,4,2] (+ optional motion added to
// dst and a common divisor, flowHomoAdpater.py:95-96) -> H [B,3,3].  h = inverse(A) @ b in fp32, in the operation
// order of the reference's torch-CPU evaluation (oracle/c/geom_oracle.c: orc_inv8 / orc_dlt4, bit-identical to the
// reference golden): LU of A^T with partial pivoting (reciprocal column scaling, fused trailing update),
// sgetrs('T') against the identity with 8-lane tree-reduced unfused dot products, plain ascending-k mat-vec.
__device__ __forceinline__ float lanes8_sum(const float* l) {
    const float a0 = l[0] + l[4], a1 = l[1] + l[5], a2 = l[2] + l[6], a3 = l[3] + l[7];
    const float b0 = a0 + a2, b1 = a1 + a3;
    return b0 + b1;
}
__global__ void dlt4_kernel(const float* __restrict__ src, const float* __restrict__ motion, float* __restrict__ Hout, int B,
                            float mscale_x, float mscale_y, float div) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float A[8][8], X[8][8], rhs[8];             // A holds the TRANSPOSE of the DLT matrix: A[col][row]
    int piv[8];
    for (int p = 0; p < 4; ++p) {
        const float sx = src[p * 2], sy = src[p * 2 + 1];
        float dx = sx, dy = sy;
        if (motion) { dx = sx + motion[(b * 4 + p) * 2] * mscale_x; dy = sy + motion[(b * 4 + p) * 2 + 1] * mscale_y; }
        const float x = sx / div, y = sy / div, u = dx / div, v = dy / div;
        const float r0[8] = {x, y, 1.f, 0.f, 0.f, 0.f, -(u * x), -(u * y)};
        const float r1[8] = {0.f, 0.f, 0.f, x, y, 1.f, -(v * x), -(v * y)};
        for (int j = 0; j < 8; ++j) { A[j][2 * p] = r0[j]; A[j][2 * p + 1] = r1[j]; }
        rhs[2 * p] = u; rhs[2 * p + 1] = v;
    }
    for (int k = 0; k < 8; ++k) {
        int p = k;
        float best = fabsf(A[k][k]);
        for (int i = k + 1; i < 8; ++i) if (fabsf(A[i][k]) > best) { best = fabsf(A[i][k]); p = i; }
        piv[k] = p;
        if (p != k) for (int j = 0; j < 8; ++j) { const float t = A[k][j]; A[k][j] = A[p][j]; A[p][j] = t; }
        const float r = 1.0f / A[k][k];
        for (int i = k + 1; i < 8; ++i) A[i][k] = A[i][k] * r;
        for (int i = k + 1; i < 8; ++i)
            for (int j = k + 1; j < 8; ++j) A[i][j] = __fmaf_rn(-A[i][k], A[k][j], A[i][j]);
    }
    for (int c = 0; c < 8; ++c) {
        for (int i = 0; i < 8; ++i) {           // U^T y = e_c
            float l[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < i; ++k) l[k] = A[k][i] * X[k][c];
            X[i][c] = ((i == c ? 1.f : 0.f) - lanes8_sum(l)) / A[i][i];
        }
        for (int i = 7; i >= 0; --i) {          // L^T z = y
            float l[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int k = i + 1; k < 8; ++k) l[k] = A[k][i] * X[k][c];
            X[i][c] = X[i][c] - lanes8_sum(l);
        }
    }
    for (int k = 7; k >= 0; --k)
        if (piv[k] != k) for (int j = 0; j < 8; ++j) { const float t = X[k][j]; X[k][j] = X[piv[k]][j]; X[piv[k]][j] = t; }
    for (int i = 0; i < 8; ++i) {
        float acc = 0.f;
        for (int k = 0; k < 8; ++k) acc = acc + X[i][k] * rhs[k];
        Hout[9 * b + i] = acc;
    }
    Hout[9 * b + 8] = 1.0f;
}

extern "C" int st_dlt4(const float* src4x2, const float* motion, float* H, int32_t B, float mscale_x, float mscale_y,
                       float div, void* stream) {
    if (!src4x2 || !H || B <= 0 || div == 0.f) return ST_EINVAL;
    hipLaunchKernelGGL(dlt4_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, src4x2, motion, H, B, mscale_x,
                       mscale_y, div);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// Canvas bounds: min/max of the rigid (gw+1)x(gh+1) mesh of [0,width]x[0,height] mapped through
// H^-1 with perspective divide (core/warp_utils.py:10-34, flowHomoAdpater.py:254-266).
// out[4] = (min x, max x, min y, max y) over all batches.
// min / max are order independent (exact), so the mesh is spread over many workgroups and the per-workgroup extremes meet in out[4]
// through integer atomics on the float bit patterns (non-negative floats order like ints, negative ones like reversed unsigned ints);
// a one-wave kernel sets out to (+inf, -inf, +inf, -inf) first.  One workgroup took 200 us for the reference's 513 x 513 mesh -- on the
// critical path of test_out, right before the host reads the canvas size.
__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
    v = v + 0.0f;                                   // -0.0 -> +0.0: its bit pattern is INT_MIN, which the signed compare would rank below every negative float
    if (v >= 0.f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
    v = v + 0.0f;
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__global__ void mesh_bounds_init_kernel(float* __restrict__ out) {
    if (threadIdx.x < 4) out[threadIdx.x] = (threadIdx.x & 1) ? -INFINITY : INFINITY;
}
__global__ __launch_bounds__(256) void mesh_bounds_kernel(const float* __restrict__ Hm, float* __restrict__ out, int B,
                                                          float width, float height, int gw, int gh) {
    __shared__ float red[4][4];
    float mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
    const int npt = (gw + 1) * (gh + 1);
    for (int b = 0; b < B; ++b) {
        float hi[9];
        mat3_inv(Hm + 9 * b, hi);
        for (int p = blockIdx.x * 256 + threadIdx.x; p < npt; p += gridDim.x * 256) {
            const float x = lin_at(0.0f, width, gw + 1, p % (gw + 1)), y = lin_at(0.0f, height, gh + 1, p / (gw + 1));
            float tx = hi[0] * x; tx = __fmaf_rn(hi[1], y, tx); tx = tx + hi[2];
            float ty = hi[3] * x; ty = __fmaf_rn(hi[4], y, ty); ty = ty + hi[5];
            float tz = hi[6] * x; tz = __fmaf_rn(hi[7], y, tz); tz = tz + hi[8];
            const float mx = tx / tz, my = ty / tz;
            mnx = fminf(mnx, mx); mxx = fmaxf(mxx, mx); mny = fminf(mny, my); mxy = fmaxf(mxy, my);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        mnx = fminf(mnx, __shfl_xor(mnx, o, 64)); mxx = fmaxf(mxx, __shfl_xor(mxx, o, 64));
        mny = fminf(mny, __shfl_xor(mny, o, 64)); mxy = fmaxf(mxy, __shfl_xor(mxy, o, 64));
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wv][0] = mnx; red[wv][1] = mxx; red[wv][2] = mny; red[wv][3] = mxy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomic_min_f32(out + 0, fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0])));
        atomic_max_f32(out + 1, fmaxf(fmaxf(red[0][1], red[1][1]), fmaxf(red[2][1], red[3][1])));
        atomic_min_f32(out + 2, fminf(fminf(red[0][2], red[1][2]), fminf(red[2][2], red[3][2])));
        atomic_max_f32(out + 3, fmaxf(fmaxf(red[0][3], red[1][3]), fmaxf(red[2][3], red[3][3])));
    }
}

extern "C" int st_mesh_bounds(const float* H, float* out4, int32_t B, float width, float height, int32_t gw, int32_t gh,
                              void* stream) {
    if (!H || !out4 || B <= 0 || gw < 0 || gh < 0) return ST_EINVAL;
    const long npt = (long)(gw + 1) * (gh + 1);
    int nb = (int)((npt + 1023) / 1024);                       // ~4 points per thread
    if (nb > 512) nb = 512;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(mesh_bounds_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out4);
    hipLaunchKernelGGL(mesh_bounds_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, H, out4, B, width, height, gw, gh);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// Backward flow warp = F.grid_sample(bilinear, zeros, align_corners=True) of x at pix + flow
// (core/warp_utils.py:54-80).  x [B,C,H,W], flow [B,2,H,W] NCHW -> out [B,C,H,W];
// optional per-pixel multiplier mul [B,1,H,W] (flowHomoAdpater.py:317).
__global__ __launch_bounds__(256) void flow_warp_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                        const float* __restrict__ mul, float* __restrict__ out, int C, int H,
                                                        int W) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (i >= H || j >= W) return;
    const size_t hw = (size_t)H * W, pix = (size_t)i * W + j;
    const float fx = flow[(size_t)b * 2 * hw + pix], fy = flow[(size_t)b * 2 * hw + hw + pix];
    const float wm = (float)(W - 1 > 1 ? W - 1 : 1), hm = (float)(H - 1 > 1 ? H - 1 : 1);
    const float gx = 2.0f * ((float)j + fx) / wm - 1.0f;
    const float gy = 2.0f * ((float)i + fy) / hm - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
    const float nw = (x1f - ix) * (y1f - iy), ne = (ix - x0f) * (y1f - iy);
    const float sw = (x1f - ix) * (iy - y0f), se = (ix - x0f) * (iy - y0f);
    const bool fin = fabsf(ix) < 1e9f && fabsf(iy) < 1e9f;
    const int x0 = fin ? (int)x0f : -2, y0 = fin ? (int)y0f : -2, x1 = x0 + 1, y1 = y0 + 1;
    const bool xin0 = x0 >= 0 && x0 < W, xin1 = x1 >= 0 && x1 < W, yin0 = y0 >= 0 && y0 < H, yin1 = y1 >= 0 && y1 < H;
    const float m = mul ? mul[(size_t)b * hw + pix] : 1.0f;
    for (int c = 0; c < C; ++c) {
        const float* im = x + ((size_t)b * C + c) * hw;
        // ATen's CPU grid_sample accumulates nw*v + ne*v + sw*v + se*v as one product followed by three fused
        // multiply-adds (measured bit for bit), out-of-range taps contributing 0
        float v = (xin0 && yin0 ? im[(size_t)y0 * W + x0] : 0.f) * nw;
        v = __fmaf_rn(xin1 && yin0 ? im[(size_t)y0 * W + x1] : 0.f, ne, v);
        v = __fmaf_rn(xin0 && yin1 ? im[(size_t)y1 * W + x0] : 0.f, sw, v);
        v = __fmaf_rn(xin1 && yin1 ? im[(size_t)y1 * W + x1] : 0.f, se, v);
        if (!fin) v = 0.f;                               // non-finite coordinate: ATen's bounds tests all fail -> 0 (0 * NaN weights would give NaN)
        out[((size_t)b * C + c) * hw + pix] = mul ? v * m : v;
    }
}

extern "C" int st_flow_warp(const float* x, const float* flow, const float* mul, float* out, int32_t B, int32_t C, int32_t H,
                            int32_t W, void* stream) {
    if (!x || !flow || !out || B <= 0 || C <= 0) return ST_EINVAL;
    dim3 grid((W + 63) / 64, (H + 3) / 4, B);
    hipLaunchKernelGGL(flow_warp_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, flow, mul, out, C, H, W);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// Bilinear resize of NCHW planes (F.interpolate): align_corners=1 for resize_flow
// (core/warp_utils.py:38-46, per-channel divisors), =0 for torchvision Resize((512,512)) without
// antialias (flowHomoAdpater.py:14,204-205).  div[c % ndiv] divides channel c afterwards.  align_corners == 2: the
// scale_factor form of out.py:281 (source step = (div0, div1) = 1/scale_factor, half-pixel centres, no division).
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ x, float* __restrict__ out, int planes,
                                                              int H, int W, int oh, int ow, int align, float div0, float div1,
                                                              int ndiv) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int p = blockIdx.z;
    if (i >= oh || j >= ow) return;
    float sy, sx;
    if (align == 1) {
        const float rh = oh > 1 ? (float)(H - 1) / (float)(oh - 1) : 0.f, rw = ow > 1 ? (float)(W - 1) / (float)(ow - 1) : 0.f;
        sy = rh * (float)i; sx = rw * (float)j;
    } else {
        // align == 2: F.interpolate(scale_factor=s) -- ATen steps the source by 1/s (passed in div0 / div1), not by H/oh
        const float rh = align == 2 ? div0 : (float)H / (float)oh, rw = align == 2 ? div1 : (float)W / (float)ow;
        sy = rh * ((float)i + 0.5f) - 0.5f; sx = rw * ((float)j + 0.5f) - 0.5f;
        if (sy < 0.f) sy = 0.f;
        if (sx < 0.f) sx = 0.f;
    }
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
    const float* im = x + (size_t)p * H * W;
    float v = hy * (hx * im[(size_t)y0 * W + x0] + lx * im[(size_t)y0 * W + x1]) +
              ly * (hx * im[(size_t)y1 * W + x0] + lx * im[(size_t)y1 * W + x1]);
    if (ndiv > 0 && align != 2) v = v / ((p % ndiv) == 0 ? div0 : div1);
    out[(size_t)p * oh * ow + (size_t)i * ow + j] = v;
}

extern "C" int st_resize_bilinear(const float* x, float* out, int32_t planes, int32_t H, int32_t W, int32_t oh, int32_t ow,
                                  int32_t align_corners, float div0, float div1, int32_t ndiv, void* stream) {
    if (!x || !out || planes <= 0 || H <= 0 || W <= 0 || oh <= 0 || ow <= 0) return ST_EINVAL;
    dim3 grid((ow + 63) / 64, (oh + 3) / 4, planes);
    hipLaunchKernelGGL(resize_bilinear_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, out, planes, H, W, oh, ow,
                       align_corners, div0, div1, ndiv);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// Range map: bilinear forward splat of ones along `flow` (core/warp_utils.py:114-175).  The
// reference's scatter_add_ is order-dependent in fp32; here the weights are summed in 2^-32
// fixed point with 64-bit integer atomics, so the result is deterministic and order-free.
// flow [B,2,H,W]; acc: caller-provided u64 scratch [B*H*W] (zeroed here); out [B,H,W] fp32.
__global__ __launch_bounds__(256) void range_splat_kernel(const float* __restrict__ flow, unsigned long long* __restrict__ acc,
                                                          int B, int H, int W) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t hw = (size_t)H * W;
    if (idx >= (size_t)B * hw) return;
    const size_t b = idx / hw, pix = idx % hw;
    const int i = pix / W, j = pix % W;
    const float cx = (float)j + flow[b * 2 * hw + pix], cy = (float)i + flow[b * 2 * hw + hw + pix];
    const float flx = floorf(cx), fly = floorf(cy);
    const float ox = cx - flx, oy = cy - fly;
    const int x0 = f2i_x86(flx), y0 = f2i_x86(fly);
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj) {
            const long xi = (long)x0 + di, yj = (long)y0 + dj;
            if (xi < 0 || xi >= W || yj < 0 || yj >= H) continue;
            const float wi = di ? ox : (1.0f - ox), wj = dj ? oy : (1.0f - oy);
            const float wgt = wi * wj;
            if (!(wgt > 0.f)) continue;
            const unsigned long long q = (unsigned long long)((double)wgt * 4294967296.0 + 0.5);
            atomicAdd(acc + b * hw + (size_t)yj * W + xi, q);
        }
}
__global__ void zero_u64_kernel(unsigned long long* __restrict__ acc, size_t n) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) acc[idx] = 0ull;
}
__global__ void range_finish_kernel(const unsigned long long* __restrict__ acc, float* __restrict__ out, size_t n) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) out[idx] = (float)((double)acc[idx] * (1.0 / 4294967296.0));
}

extern "C" int st_range_map(const float* flow, void* scratch_u64, float* out, int32_t B, int32_t H, int32_t W, void* stream) {
    if (!flow || !scratch_u64 || !out || B <= 0) return ST_EINVAL;
    const size_t n = (size_t)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    // zeroed by a kernel, not hipMemsetAsync: captured into a hipGraph, the memset node did not clear the accumulator on replays
    // (ROCm 7.2: the second replay of a captured forward summed onto the first one's range map -- tools/harness_determinism6.py)
    hipLaunchKernelGGL(zero_u64_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (unsigned long long*)scratch_u64, n);
    hipLaunchKernelGGL(range_splat_kernel, dim3((n + 255) / 256), dim3(256), 0, s, flow, (unsigned long long*)scratch_u64, B, H, W);
    hipLaunchKernelGGL(range_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const unsigned long long*)scratch_u64, out, n);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// occlusion mask of compute_occlusion('wang', occlusion_are_zeros=True) (warp_utils.py:212-220):
// 1 - (1 - clamp(range, 0, 1)); optional hard threshold >= 0.5 (flowHomoAdpater.py:181).
__global__ void occlusion_kernel(const float* __restrict__ range, float* __restrict__ out, size_t n, int threshold) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float c = fminf(fmaxf(range[idx], 0.0f), 1.0f);
    float o = 1.0f - (1.0f - c);
    if (threshold) o = o >= 0.5f ? 1.0f : 0.0f;
    out[idx] = o;
}

extern "C" int st_occlusion_from_range(const float* range, float* out, int64_t n, int32_t threshold, void* stream) {
    if (!range || !out || n <= 0) return ST_EINVAL;
    hipLaunchKernelGGL(occlusion_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, range, out, (size_t)n, threshold);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// preprocess_occlusion_mask (flowHomoAdpater.py:18-35): threshold >= 0.5, erosion with a ksz x ksz
// zero-padded box (all ones <=> conv == ksz^2), dilation (conv >= 1).  Separable, exact on bytes.
// mode 0: erode, 1: dilate; horiz 1: along x, 0: along y; thr: input is fp32 to threshold.
__global__ void morph_pass_kernel(const float* __restrict__ srcf, const unsigned char* __restrict__ srcb,
                                  unsigned char* __restrict__ dstb, float* __restrict__ dstf, int N, int H, int W, int r,
                                  int horiz, int mode) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t hw = (size_t)H * W;
    if (idx >= (size_t)N * hw) return;
    const size_t n = idx / hw, pix = idx % hw;
    const int y = pix / W, x = pix % W;
    int all = 1, any = 0;
    for (int a = -r; a <= r; ++a) {
        const int yy = horiz ? y : y + a, xx = horiz ? x + a : x;
        int v = 0;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const size_t k = n * hw + (size_t)yy * W + xx;
            v = srcf ? (srcf[k] >= 0.5f) : srcb[k];
        }
        all &= v; any |= v;
    }
    const int o = mode ? any : all;
    if (dstb) dstb[idx] = (unsigned char)o;
    if (dstf) dstf[idx] = o ? 1.0f : 0.0f;
}

extern "C" int st_morph_open(const float* mask, float* out, void* scratch_u8x2, int32_t N, int32_t H, int32_t W, int32_t ksz,
                             void* stream) {
    if (!mask || !out || !scratch_u8x2 || N <= 0 || ksz <= 0 || !(ksz & 1)) return ST_EINVAL;
    const size_t n = (size_t)N * H * W;
    unsigned char* t0 = (unsigned char*)scratch_u8x2;
    unsigned char* t1 = t0 + n;
    hipStream_t s = (hipStream_t)stream;
    dim3 g((n + 255) / 256), b(256);
    const int r = ksz / 2;
    hipLaunchKernelGGL(morph_pass_kernel, g, b, 0, s, mask, (const unsigned char*)nullptr, t0, (float*)nullptr, N, H, W, r, 1, 0);
    hipLaunchKernelGGL(morph_pass_kernel, g, b, 0, s, (const float*)nullptr, (const unsigned char*)t0, t1, (float*)nullptr, N, H, W, r, 0, 0);
    hipLaunchKernelGGL(morph_pass_kernel, g, b, 0, s, (const float*)nullptr, (const unsigned char*)t1, t0, (float*)nullptr, N, H, W, r, 1, 1);
    hipLaunchKernelGGL(morph_pass_kernel, g, b, 0, s, (const float*)nullptr, (const unsigned char*)t0, (unsigned char*)nullptr, out, N, H, W, r, 0, 1);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// eval-mode tail (flowHomoAdpater.py:171-183): overlap = (mean(final[:,3:6]) < 0.9),
// final *= occ (occ already thresholded).  final [B,6,H,W] in place; overlap [B,H,W].
__global__ void eval_finish_kernel(float* __restrict__ fin, const float* __restrict__ occ, float* __restrict__ overlap, int B,
                                   size_t hw) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * hw) return;
    const size_t b = idx / hw, pix = idx % hw;
    float* f = fin + b * 6 * hw + pix;
    const float m = ((f[3 * hw] + f[4 * hw]) + f[5 * hw]) / 3.0f;
    overlap[idx] = m < 0.9f ? 1.0f : 0.0f;
    const float o = occ[idx];
    for (int c = 0; c < 6; ++c) f[c * hw] = f[c * hw] * o;
}

extern "C" int st_eval_finish(float* final6, const float* occ, float* overlap, int32_t B, int32_t H, int32_t W, void* stream) {
    if (!final6 || !occ || !overlap) return ST_EINVAL;
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(eval_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, final6, occ, overlap, B, (size_t)H * W);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// test_out mask algebra + blend (flowHomoAdpater.py:339-360), one pixel per thread.
//   homo1, homo2, fin: [6, h, w] (rgb + mask); fin already multiplied by flow mask; occ [h, w]
//   -> final_out (fin*occ, 6ch), output1/output2/mask1/mask2 [3,h,w] fp32, blend [3,h,w] u8
__global__ void blend_kernel(const float* __restrict__ homo1, const float* __restrict__ homo2, float* __restrict__ fin,
                             const float* __restrict__ occ, float* __restrict__ output2, float* __restrict__ mask1o,
                             float* __restrict__ mask2o, unsigned char* __restrict__ blend, size_t hw) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const float oc = occ[p];
    float o2[3], m2[3], m1[3];
    for (int c = 0; c < 6; ++c) fin[c * hw + p] = fin[c * hw + p] * oc;                 // :339
    for (int c = 0; c < 3; ++c) {
        const float mm1 = homo1[(3 + c) * hw + p], mm2 = fin[(3 + c) * hw + p];
        const float nov = 1.0f - mm1;                                                     // :343
        o2[c] = homo2[c * hw + p] * (1.0f - mm2) * nov + fin[c * hw + p] * mm2;           // :345
        m2[c] = homo2[(3 + c) * hw + p] * (1.0f - mm2) * nov + mm2 * mm2;                 // :346
        m1[c] = mm1;
        float bl = (homo1[c * hw + p] * mm1 + o2[c] * m2[c]) / (mm1 + m2[c]);             // :355
        bl = fminf(fmaxf(bl, 0.0f), 255.0f);                                              // NaN -> 0 like the CPU cast
        blend[c * hw + p] = (bl == bl) ? (unsigned char)bl : (unsigned char)0;
        output2[c * hw + p] = o2[c];
    }
    const float a1 = fminf(fmaxf(((m1[0] + m1[1]) + m1[2]) / 3.0f, 0.0f), 1.0f);         // :359-360
    const float a2 = fminf(fmaxf(((m2[0] + m2[1]) + m2[2]) / 3.0f, 0.0f), 1.0f);
    for (int c = 0; c < 3; ++c) { mask1o[c * hw + p] = a1; mask2o[c * hw + p] = a2; }
}

extern "C" int st_blend(const float* homo1, const float* homo2, float* fin, const float* occ, float* output2, float* mask1,
                        float* mask2, uint8_t* blend, int32_t h, int32_t w, void* stream) {
    if (!homo1 || !homo2 || !fin || !occ || !output2 || !mask1 || !mask2 || !blend) return ST_EINVAL;
    const size_t hw = (size_t)h * w;
    hipLaunchKernelGGL(blend_kernel, dim3((hw + 255) / 256), dim3(256), 0, (hipStream_t)stream, homo1, homo2, fin, occ, output2,
                       mask1, mask2, blend, hw);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// (mean over channels > thr) ? 1 : 0   (flowHomoAdpater.py:233-234 warp_input2_mask)
__global__ void mean_threshold_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int C, size_t hw, float thr) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * hw) return;
    const size_t b = idx / hw, p = idx % hw;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = s + x[(b * C + c) * hw + p];
    out[idx] = (s / (float)C) > thr ? 1.0f : 0.0f;
}

extern "C" int st_mean_threshold(const float* x, float* out, int32_t B, int32_t C, int32_t H, int32_t W, float thr, void* stream) {
    if (!x || !out) return ST_EINVAL;
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(mean_threshold_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, out, B, C, (size_t)H * W, thr);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// UDIS2 TPS transformer (core/udis_utils/torch_tps_transform.py:7-190).
// (1) solve: W = [[P, R], [0, P^T]] (N+3)^2 with R = d2*log(d2 + 1e-6) (fp32 entries), fp64
//     Gauss-Jordan with partial pivoting on [W | target;0] -> T [B, 2, N+3] fp32.
//     work: fp64 scratch [B, (N+3), (N+5)].
__global__ __launch_bounds__(256) void tps_solve_kernel(const float* __restrict__ source, const float* __restrict__ target,
                                                        double* __restrict__ work, float* __restrict__ T, int N) {
    const int b = blockIdx.x, n3 = N + 3, ld = N + 5;
    double* A = work + (size_t)b * n3 * ld;
    const float* sp = source + (size_t)b * N * 2;
    const float* tp = target + (size_t)b * N * 2;
    __shared__ int s_piv;
    __shared__ double s_best[256];
    __shared__ int s_idx[256];
    for (int e = threadIdx.x; e < n3 * ld; e += 256) {
        const int r = e / ld, c = e % ld;
        double v = 0.0;
        if (r < N) {
            const float px = sp[r * 2], py = sp[r * 2 + 1];
            if (c == 0) v = 1.0;
            else if (c == 1) v = px;
            else if (c == 2) v = py;
            else if (c < n3) {
                const float qx = sp[(c - 3) * 2], qy = sp[(c - 3) * 2 + 1];
                const float d0 = 1.0f - 1.0f, dx = px - qx, dy = py - qy;
                const float d2 = (d0 * d0 + dx * dx) + dy * dy;                 // sum over (1,x,y) components (:157)
                v = d2 * st_logf_cr(d2 + 1e-6f);
            } else v = tp[r * 2 + (c - n3)];
        } else {
            const int k = r - N;                                                // rows [0 | P^T]
            if (c >= 3 && c < n3) v = (k == 0) ? 1.0 : sp[(c - 3) * 2 + (k - 1)];
        }
        A[e] = v;
    }
    __syncthreads();
    for (int c = 0; c < n3; ++c) {
        double best = -1.0; int bi = c;
        for (int r = c + threadIdx.x; r < n3; r += 256) { const double a = fabs(A[(size_t)r * ld + c]); if (a > best) { best = a; bi = r; } }
        s_best[threadIdx.x] = best; s_idx[threadIdx.x] = bi;
        __syncthreads();
        if (threadIdx.x == 0) {
            double bb = -1.0; int ii = c;
            for (int t = 0; t < 256; ++t) if (s_best[t] > bb) { bb = s_best[t]; ii = s_idx[t]; }
            s_piv = ii;
        }
        __syncthreads();
        const int piv = s_piv;
        if (piv != c) for (int k = threadIdx.x; k < ld; k += 256) { const double t = A[(size_t)c * ld + k]; A[(size_t)c * ld + k] = A[(size_t)piv * ld + k]; A[(size_t)piv * ld + k] = t; }
        __syncthreads();
        const double inv = 1.0 / A[(size_t)c * ld + c];
        for (int r = threadIdx.x; r < n3; r += 256) {
            if (r == c) continue;
            const double f = A[(size_t)r * ld + c] * inv;
            if (f != 0.0) for (int k = c + 1; k < ld; ++k) A[(size_t)r * ld + k] -= f * A[(size_t)c * ld + k];
            A[(size_t)r * ld + c] = 0.0;
        }
        __syncthreads();
    }
    for (int r = threadIdx.x; r < n3; r += 256) {
        const double d = A[(size_t)r * ld + r];
        T[((size_t)b * 2 + 0) * n3 + r] = (float)(A[(size_t)r * ld + n3] / d);
        T[((size_t)b * 2 + 1) * n3 + r] = (float)(A[(size_t)r * ld + n3 + 1] / d);
    }
}

// (2) warp: grid row = [1, x, y, r_1..r_N] per output pixel, (x_s, y_s) = T @ grid as a sequential
//     fma chain over k, then the same 4-tap gather as the homography transformer.
__global__ __launch_bounds__(256) void tps_warp_kernel(const float* __restrict__ U, const float* __restrict__ source,
                                                       const float* __restrict__ T, float* __restrict__ out, int* __restrict__ idx,
                                                       int C, int H, int W, int oh, int ow, int N) {
    extern __shared__ float sh[];           // source [N,2], T [2, N+3]
    const int b = blockIdx.z, n3 = N + 3;
    float* s_src = sh;
    float* s_T = sh + 2 * N;
    for (int e = threadIdx.x; e < 2 * N; e += 256) s_src[e] = source[(size_t)b * 2 * N + e];
    for (int e = threadIdx.x; e < 2 * n3; e += 256) s_T[e] = T[(size_t)b * 2 * n3 + e];
    __syncthreads();
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= oh || j >= ow) return;
    const float gx = lin_at(-1.0f, 1.0f, ow, j), gy = lin_at(-1.0f, 1.0f, oh, i);
    float xs = s_T[0] * 1.0f, ys = s_T[n3] * 1.0f;
    xs = __fmaf_rn(s_T[1], gx, xs); ys = __fmaf_rn(s_T[n3 + 1], gx, ys);
    xs = __fmaf_rn(s_T[2], gy, xs); ys = __fmaf_rn(s_T[n3 + 2], gy, ys);
    for (int k = 0; k < N; ++k) {
        const float dx = gx - s_src[2 * k], dy = gy - s_src[2 * k + 1];
        const float d2 = dx * dx + dy * dy;
        const float r = d2 * st_logf_cr(d2 + 1e-6f);
        xs = __fmaf_rn(s_T[3 + k], r, xs); ys = __fmaf_rn(s_T[n3 + 3 + k], r, ys);
    }
    const Tap4 t = taps_from_normalised(xs, ys, W, H);
    if (idx) {
        int* p = idx + (((size_t)b * oh + i) * ow + j) * 4;
        p[0] = t.x0; p[1] = t.x1; p[2] = t.y0; p[3] = t.y1;
    }
    if (!out) return;
    const size_t ohw = (size_t)oh * ow;
    for (int c = 0; c < C; ++c) {
        const float* im = U + ((size_t)b * C + c) * H * W;
        float v = t.wa * im[(size_t)t.y0 * W + t.x0];
        v = v + t.wb * im[(size_t)t.y1 * W + t.x0];
        v = v + t.wc * im[(size_t)t.y0 * W + t.x1];
        v = v + t.wd * im[(size_t)t.y1 * W + t.x1];
        out[((size_t)b * C + c) * ohw + (size_t)i * ow + j] = v;
    }
}

extern "C" int st_tps_solve_grid(const float* U, const float* source, const float* target, void* work_f64, float* T, float* out,
                                 int32_t* idx, int32_t B, int32_t C, int32_t H, int32_t W, int32_t N, int32_t oh, int32_t ow,
                                 void* stream) {
    if (!source || !target || !work_f64 || !T || B <= 0 || N <= 0 || N > 4000) return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tps_solve_kernel, dim3(B), dim3(256), 0, s, source, target, (double*)work_f64, T, N);
    if (out || idx) {
        if (!U && out) return ST_EINVAL;
        dim3 grid((ow + 63) / 64, (oh + 3) / 4, B);
        const size_t lds = (size_t)(2 * N + 2 * (N + 3)) * sizeof(float);
        hipLaunchKernelGGL(tps_warp_kernel, grid, dim3(256), lds, s, U, source, T, out, idx, C, H, W, oh, ow, N);
    }
    ST_CHECK_LAUNCH();
    return ST_OK;
}
