// fp32 implicit-GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chain).
//
//   C[m, n] = epilogue( alpha * sum_k A(m, k) * Wt[n, k] + bias[n] )
//
// A is an NHWC activation read through convolution addressing (kh x kw window, stride, zero
// padding); a plain row-major matrix is the 1x1 case.  Wt is [N, K] row-major with
// K = kh*kw*Cin ordered (ky, kx, c) -- the layout nn.Linear already has and the layout the host
// prepacks conv weights into.  Both operand tiles are K-contiguous in LDS; each lane fetches 4 consecutive k
// of its row with one ds_read_b128 and feeds them to 4 consecutive MFMAs, lane half h taking k = 8j+4h+t for
// both operands so the contraction pairs up without any transposition.
//
// Kernels in this file (dispatch: conv_gemm_launch):
//   conv_gemm_dma_kernel   Cin % 32 == 0 (the default): global -> LDS DMA ring, VALU-free K loop, optional persistent walk
//                          over M tiles; tiles 64x64 / 128x64 / 128x32
//   conv_gemm_kernel       any Cin % 4 == 0 (or scalar gather otherwise): register-staged, [rows][BK+4] padded LDS tiles
//   skinny_gemm_kernel (M <= 8), narrow_conv_kernel (N <= 4), splitk_reduce_kernel
// and one epilogue (bias, alpha, row-mapped addend, activation, residual / gate / GRU / axpy / fused z|r) shared by all.
//
// Replaces (reference, /root/reference): every F.conv2d / nn.Linear / einsum contraction on the
// FlowHomoAdpater path, e.g. core/FlowFormer/PerCostFormer3/encoder.py:359-369 (all-pairs corr),
// gru.py:44-59 (SepConvGRU), core/UDIS2/Homography/network.py:103-137 (ResNet-50 + regressor).
#include "common.h"
#include <string.h>
#include "../../include/stitch_gfx950.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// optional profiling observer (st_set_gemm_observer) and the launch plan of the calling thread's last launch (st_gemm_last_plan)
typedef void (*st_gemm_observer_fn)(const st_gemm_desc*, void* stream, int32_t phase, void* user);
static st_gemm_observer_fn g_observer = nullptr;
static void* g_observer_user = nullptr;
static thread_local int32_t g_last_plan[4] = {-1, 0, 0, 0};

#define BK 32
#define LDS_LD (BK + 4)

// Operand tiles are fetched with raw buffer loads: a lane whose tap / row / k is out of range gets an
// offset past the descriptor's num_records and the hardware returns zeros -- no branch and, crucially, no
// select on the loaded data (a select makes the compiler wait for the load right after issuing it, which
// serialises L2 latency with the MFMA block; measured: 3000 instead of ~1300 cycles per K step).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define ST_OOB 0x80000000u

__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// v = act(alpha*acc + bias[n] + aux0) followed by the combine mode; shared by the GEMM epilogue and
// the split-K reducer.
__device__ __forceinline__ float gemm_epilogue(const st_gemm_desc& d, int m, int n, float acc, float sc) {
    float v = fmaf(acc, d.alpha, d.bias ? d.bias[n] : 0.f);      // explicit fma everywhere: every kernel's epilogue rounds alike
    if (d.aux0) {
        int ar = m;
        if (d.aux0_row_div > 1) ar = m / d.aux0_row_div;
        if (d.aux0_row_mod > 0) ar = ar % d.aux0_row_mod;
        v += d.aux0[(size_t)ar * d.ld_aux0 + n];
    }
    v = st_act(v, d.act);
    switch (d.epi) {
        case ST_EPI_ADD: v += d.aux1[(size_t)m * d.ld_aux1 + n]; break;
        case ST_EPI_MUL: v *= d.aux1[(size_t)m * d.ld_aux1 + n]; break;
        case ST_EPI_GRU: {
            const float z = d.aux1[(size_t)m * d.ld_aux1 + n], h = d.aux2[(size_t)m * d.ld_aux2 + n];
            v = (1.0f - z) * h + z * v;
        } break;
        case ST_EPI_AXPY: v = fmaf(sc, v, d.aux1[(size_t)m * d.ld_aux1 + n]); break;
        default: break;
    }
    return v;
}

// st_gemm_desc.c_planes: element (m, col) of the plane-carrying output into the three blocked bf16 planes (scalar form: split-K reducer)
// (consecutive threads of the reducer own consecutive columns of a row -- N is even -- so lane ^ 1 holds the neighbouring column: dword stores)
__device__ __forceinline__ void gemm_store_planes(const st_gemm_desc& d, int m, int col, float v) {
    __bf16 h, mi, lo;
    st_split3(v, h, mi, lo);
    const unsigned ph = st_bf16_bits(h), pm = st_bf16_bits(mi), pl = st_bf16_bits(lo);
    const unsigned nh = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ph, 0xB1, 0xF, 0xF, false);
    const unsigned nm = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pm, 0xB1, 0xF, 0xF, false);
    const unsigned nl = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pl, 0xB1, 0xF, 0xF, false);
    if (col & 1) return;
    const int cc = d.c_plane_col0 + col;
    unsigned* p = reinterpret_cast<unsigned*>(reinterpret_cast<__bf16*>(d.c_planes) + ((size_t)(cc >> 5) * d.c_plane_rows + d.c_plane_row0 + m) * 32 + (cc & 31));
    p[0] = ph | (nh << 16); p[d.c_plane_stride / 2] = pm | (nm << 16); p[d.c_plane_stride] = pl | (nl << 16);
}

// epilogue + store.  ST_EPI_ZR (fused GRU gates, gru.py:47-49): columns [0, N/2) are z -> C,
// columns [N/2, N) are r and leave as r*h -> c2 (aux1 = h).
__device__ __forceinline__ void gemm_store(const st_gemm_desc& d, float* __restrict__ C, int m, int n, float acc, float sc) {
    if (d.epi == ST_EPI_ZR) {
        const int half = d.N >> 1;
        float v = fmaf(acc, d.alpha, d.bias ? d.bias[n] : 0.f);
        if (d.aux0) v += d.aux0[(size_t)m * d.ld_aux0 + n];
        v = st_act(v, d.act);
        if (n < half) C[(size_t)m * d.ldc + n] = v;
        else {
            const float rh = v * d.aux1[(size_t)m * d.ld_aux1 + (n - half)];
            if (!d.c_no_f32) d.c2[(size_t)m * d.ldc2 + (n - half)] = rh;
            if (d.c_planes) gemm_store_planes(d, m, n - half, rh);
        }
        return;
    }
    const float o = gemm_epilogue(d, m, n, acc, sc);
    if (!d.c_no_f32) C[(size_t)m * d.ldc + n] = o;
    if (d.c_planes) gemm_store_planes(d, m, n, o);
}

// Epilogue shared by the fp32 and the split-bf16 kernels, in two halves so that the operand loads (bias, the
// pre-activation addend, residual / gate operands) can be issued BEFORE the K loop and land under its MFMAs:
// issued after it, they are two or three dependent L2 round trips that a short-K tile (K = 128: 64 MFMAs) cannot hide.
// acc[r] is C[row = (r&3) + 8*(r>>2) + 4*lh][col = li] of the 32x32 tile.
// Everything goes through raw buffer instructions: the per-lane offset (first row of the lane, its column) is
// computed once per 32x32 sub-tile, the 16 row steps are SGPR offsets, and the hardware range check (which
// includes the SGPR offset on gfx950 -- probed) drops rows >= M; absent operands get a zero-record descriptor, so the
// loads are unconditional, return 0 and touch no memory.  Result: no per-element address arithmetic or predication
// on the VALU, which v_mfma_f32_32x32x2_f32 shares its datapath with.
template <int TM, int TN>
struct EpiOperands {
    float sc;
    float bv[TN];
    float a0[TM][TN][16];      // aux0 (pre-activation addend)
    float x1[TM][TN][16];      // aux1
    float x2[TM][TN][16];      // aux2 (GRU state)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t epi_rsrc(const float* p, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_st(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)voff, (int)soff, 0);
}
// st_gemm_desc.c_planes: the lane's value of accumulator row r into the three blocked bf16 planes.  Lanes li = 0..31 of a sub-tile are the 32
// channels of one chunk row; neighbouring lanes exchange their halves (one DPP move per plane) and the EVEN lane stores the pair as a dword
// (vp of the odd lanes is the out-of-range sentinel): 16 dword lanes = one 64-byte chunk row per wave half, no sub-dword store anywhere.
__device__ __forceinline__ void buf_st_planes(float v, __amdgpu_buffer_rsrc_t r, unsigned vp_even, unsigned soff, unsigned plane_b) {
    __bf16 h, mi, lo;
    st_split3(v, h, mi, lo);
    const unsigned ph = st_bf16_bits(h), pm = st_bf16_bits(mi), pl = st_bf16_bits(lo);
    const unsigned nh = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ph, 0xB1, 0xF, 0xF, false);     // quad_perm [1, 0, 3, 2]: lane ^ 1
    const unsigned nm = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pm, 0xB1, 0xF, 0xF, false);
    const unsigned nl = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pl, 0xB1, 0xF, 0xF, false);
    __builtin_amdgcn_raw_buffer_store_b32(ph | (nh << 16), r, (int)vp_even, (int)soff, 0);
    __builtin_amdgcn_raw_buffer_store_b32(pm | (nm << 16), r, (int)vp_even, (int)(soff + plane_b), 0);
    __builtin_amdgcn_raw_buffer_store_b32(pl | (nl << 16), r, (int)vp_even, (int)(soff + 2u * plane_b), 0);
}
// row r of a lane's 16 accumulator registers, relative to the lane's first row
#define ST_EPI_ROW(r) (((r) & 3) + 8 * ((r) >> 2))

// per-workgroup constants of the epilogue: bias of the lane's columns, the device scalar of ST_EPI_AXPY
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_consts(const st_gemm_desc& d, EpiOperands<TM, TN>& e, int n0, int wn, int li, int split) {
    const bool raw = split > 1;
    e.sc = (d.scale_ptr && !raw) ? *d.scale_ptr : 1.0f;
    const __amdgpu_buffer_rsrc_t rb = epi_rsrc(raw ? nullptr : d.bias, (long long)d.N * 4);
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int n = n0 + wn * TN * 32 + jn * 32 + li;
        e.bv[jn] = buf_ld(rb, (unsigned)(n < d.N ? n : d.N - 1) * 4u, 0);
    }
}

// LITE (row-streaming kernel; the host checks the descriptor): no per-element row mapping, no GRU / z|r modes -- their
// operand registers and code are not instantiated.
template <int TM, int TN, bool LITE = false>
__device__ __forceinline__ void gemm_epilogue_load(const st_gemm_desc& d, EpiOperands<TM, TN>& e, int m0, int n0, int wm, int wn,
                                                   int li, int lh, int split) {
    const bool raw = split > 1;                                // raw partial sums: nothing to fetch
    const int half = d.N >> 1;
    const bool zr = !LITE && d.epi == ST_EPI_ZR;
    const long long M = d.M;
    // aux0 row = (m / div) % mod.  div == 8 without mod (one table row per pixel, 8 latent rows each -- the vertical
    // layers' q / k tables) keeps the SGPR-step form: a lane's rows m0' + (r&3) + 8*(r>>2), m0' % 4 == 0, map to
    // table rows m0'/8 + (r>>2), i.e. four loads.  Other mappings are computed per element (small GEMMs only).
    // mod % 32 == 0 without div (a table of `mod` rows repeated down the matrix -- PatchEmbed's per-patch position table):
    // a 32-row sub-tile never wraps, so it is the identity form started at row (sub-tile start) % mod.
    const bool div8 = d.aux0_row_div == 8 && d.aux0_row_mod <= 0;
    const bool mod32 = d.aux0_row_div <= 1 && d.aux0_row_mod > 0 && (d.aux0_row_mod & 31) == 0;
    const bool mapped = !LITE && !div8 && !mod32 && (d.aux0_row_div > 1 || d.aux0_row_mod > 0);
    const __amdgpu_buffer_rsrc_t r0 = epi_rsrc(raw ? nullptr : d.aux0, mapped ? 0x7fffffffLL
                                               : div8 ? (((M + 7) / 8 - 1) * d.ld_aux0 + d.N) * 4
                                               : mod32 ? ((long long)(d.aux0_row_mod - 1) * d.ld_aux0 + d.N) * 4 : ((M - 1) * d.ld_aux0 + d.N) * 4);
    const float* aux1 = d.aux1 ? d.aux1 + (size_t)(d.batch > 1 ? blockIdx.z : 0) * d.batch_stride_aux1 : nullptr;
    const __amdgpu_buffer_rsrc_t r1 = epi_rsrc((raw || d.epi == ST_EPI_STORE) ? nullptr : aux1, ((M - 1) * d.ld_aux1 + (zr ? half : d.N)) * 4);
    const __amdgpu_buffer_rsrc_t r2 = epi_rsrc((raw || d.epi != ST_EPI_GRU) ? nullptr : d.aux2, ((M - 1) * d.ld_aux2 + d.N) * 4);
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int n = n0 + wn * TN * 32 + jn * 32 + li;
        const int nc = n < d.N ? n : d.N - 1;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row0 = m0 + wm * TM * 32 + i * 32 + 4 * lh;              // this lane's first row
            // aux0: identity rows, or the (row / div) % mod table mapping (per element; small tables).  Operands the
            // mode does not use are not fetched (wave-uniform branches; their registers stay undefined and unread).
            if (!raw && d.aux0) {
                if (mapped) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int ar = min(row0 + ST_EPI_ROW(r), d.M - 1);
                        if (d.aux0_row_div > 1) ar = ar / d.aux0_row_div;
                        if (d.aux0_row_mod > 0) ar = ar % d.aux0_row_mod;
                        e.a0[i][jn][r] = buf_ld(r0, (unsigned)(ar * d.ld_aux0 + nc) * 4u, 0);
                    }
                } else if (div8) {
                    const unsigned v0 = (unsigned)((row0 >> 3) * d.ld_aux0 + nc) * 4u;
                    float t4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) t4[q] = buf_ld(r0, v0, (unsigned)(q * d.ld_aux0) * 4u);
#pragma unroll
                    for (int r = 0; r < 16; ++r) e.a0[i][jn][r] = t4[r >> 2];
                } else {
                    const unsigned v0 = (unsigned)((mod32 ? row0 % d.aux0_row_mod : row0) * d.ld_aux0 + nc) * 4u;
#pragma unroll
                    for (int r = 0; r < 16; ++r) e.a0[i][jn][r] = buf_ld(r0, v0, (unsigned)(ST_EPI_ROW(r) * d.ld_aux0) * 4u);
                }
            }
            if (!raw && d.epi != ST_EPI_STORE) {
                const int c1 = zr ? (nc >= half ? nc - half : 0) : nc;
                const unsigned v1 = (unsigned)(row0 * d.ld_aux1 + c1) * 4u;
#pragma unroll
                for (int r = 0; r < 16; ++r) e.x1[i][jn][r] = buf_ld(r1, v1, (unsigned)(ST_EPI_ROW(r) * d.ld_aux1) * 4u);
                if (!LITE && d.epi == ST_EPI_GRU) {
                    const unsigned v2 = (unsigned)(row0 * d.ld_aux2 + nc) * 4u;
#pragma unroll
                    for (int r = 0; r < 16; ++r) e.x2[i][jn][r] = buf_ld(r2, v2, (unsigned)(ST_EPI_ROW(r) * d.ld_aux2) * 4u);
                }
            }
        }
    }
}

template <int TM, int TN, bool LITE = false, bool CT = false>
__device__ __forceinline__ void gemm_epilogue_store(const st_gemm_desc& d, float* __restrict__ C, f32x16 (&acc)[TM][TN],
                                                    const EpiOperands<TM, TN>& e, int m0, int n0, int wm, int wn, int li, int lh,
                                                    int split, int kz) {
    const int half = d.N >> 1;
    const long long M = d.M;
    if (split > 1) {                                           // raw partial sums -> slab kz of the workspace
        const __amdgpu_buffer_rsrc_t rw = epi_rsrc(d.workspace + (size_t)kz * d.M * d.N, M * d.N * 4);
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int n = n0 + wn * TN * 32 + jn * 32 + li;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row0 = m0 + wm * TM * 32 + i * 32 + 4 * lh;
                const unsigned vo = n < d.N ? (unsigned)(row0 * d.N + n) * 4u : ST_OOB;
#pragma unroll
                for (int r = 0; r < 16; ++r) buf_st(acc[i][jn][r], rw, vo, (unsigned)(ST_EPI_ROW(r) * d.N) * 4u);
            }
        }
        return;
    }
    const bool zr = !LITE && d.epi == ST_EPI_ZR;
    // c_no_f32: the plane-carrying output (c, or c2 in z|r mode) is not stored as fp32 (zero-record descriptor: the stores are dropped)
    const __amdgpu_buffer_rsrc_t rc = epi_rsrc((d.c_no_f32 && !zr) ? nullptr : C, ((M - 1) * d.ldc + (zr ? half : d.N)) * 4);
    const __amdgpu_buffer_rsrc_t rc2 = epi_rsrc((zr && !d.c_no_f32) ? d.c2 : nullptr, ((M - 1) * d.ldc2 + half) * 4);
    // optional plane copy of the result (st_gemm_desc.c_planes): host-checked M % 32 == 0, c_plane_col0 % 32 == 0, extents < 2 GiB
    const bool planes = !LITE && d.c_planes != nullptr;
    const unsigned plane_b = (unsigned)(d.c_plane_stride * 2);
    const __amdgpu_buffer_rsrc_t rp = epi_rsrc(planes ? reinterpret_cast<const float*>(d.c_planes) : nullptr, 0x7fffffffLL);
    const long long prow0 = d.c_plane_row0 + (long long)(d.batch > 1 ? blockIdx.z : 0) * d.c_plane_batch_rows;
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int n = n0 + wn * TN * 32 + jn * 32 + li;
        const bool ncol = n < d.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row0 = m0 + wm * TM * 32 + i * 32 + 4 * lh;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaf(acc[i][jn][r], d.alpha, e.bv[jn]);
            if (d.aux0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += e.a0[i][jn][r];
            }
            switch (d.act) {                                   // wave-uniform, outside the register loop
                case ST_ACT_RELU:
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
                    break;
                case ST_ACT_GELU:
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = st_act(v[r], ST_ACT_GELU);
                    break;
                case ST_ACT_SIGMOID:
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = st_act(v[r], ST_ACT_SIGMOID);
                    break;
                case ST_ACT_TANH:
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = st_act(v[r], ST_ACT_TANH);
                    break;
                default: break;
            }
            // plane copy: byte offset of (this lane's first row, its channel) inside plane 0; a 32-row sub-tile is wholly inside M
            unsigned vp = ST_OOB;
            if (planes) {
                const int pcol = d.c_plane_col0 + (zr ? n - half : n);
                // (even lane: its column and the next are both inside -- the plane-carrying output has an even number of columns, host-checked)
                if (!(li & 1) && ncol && (!zr || n >= half) && row0 < d.M)
                    vp = (unsigned)((((long long)(pcol >> 5) * d.c_plane_rows + prow0 + row0) * 32 + (pcol & 31)) * 2);
            }
            if (zr) {
                const unsigned vc = (ncol && n < half) ? (unsigned)(row0 * d.ldc + n) * 4u : ST_OOB;
                const unsigned vc2 = (ncol && n >= half) ? (unsigned)(row0 * d.ldc2 + n - half) * 4u : ST_OOB;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    buf_st(v[r], rc, vc, (unsigned)(ST_EPI_ROW(r) * d.ldc) * 4u);
                    const float rh = v[r] * e.x1[i][jn][r];
                    buf_st(rh, rc2, vc2, (unsigned)(ST_EPI_ROW(r) * d.ldc2) * 4u);
                    if (planes) buf_st_planes(rh, rp, vp, (unsigned)(ST_EPI_ROW(r) * 64), plane_b);
                }
            } else {
                const unsigned vc = ncol ? (unsigned)(row0 * d.ldc + n) * 4u : ST_OOB;
                if (d.epi == ST_EPI_STORE) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) buf_st(v[r], rc, vc, (unsigned)(ST_EPI_ROW(r) * d.ldc) * 4u);
                    if (planes) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) buf_st_planes(v[r], rp, vp, (unsigned)(ST_EPI_ROW(r) * 64), plane_b);
                    }
                    if (CT && d.c_t) {                           // (only the CT instantiations carry this code)
                        // transposed copy: the lane's 16 values are 4 runs of 4 consecutive rows of column n -> 4 x 16-byte stores
                        // into row n of c_t (M % 4 == 0: a run is inside the matrix or wholly outside)
                        float* ctb = d.c_t + (size_t)(d.batch > 1 ? blockIdx.z : 0) * d.batch_stride_c;
                        const __amdgpu_buffer_rsrc_t rt = epi_rsrc(ctb, ((long long)(d.N - 1) * d.ld_ct + M) * 4);
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const unsigned vt = (ncol && row0 + 8 * q4 < d.M) ? (unsigned)(n * d.ld_ct + row0 + 8 * q4) * 4u : ST_OOB;
                            const u32x4 pk = {__float_as_uint(v[4 * q4]), __float_as_uint(v[4 * q4 + 1]), __float_as_uint(v[4 * q4 + 2]),
                                              __float_as_uint(v[4 * q4 + 3])};
                            __builtin_amdgcn_raw_buffer_store_b128(pk, rt, (int)vt, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float x1 = e.x1[i][jn][r];
                        float o = v[r] + x1;                                   // ST_EPI_ADD
                        if (d.epi == ST_EPI_MUL) o = v[r] * x1;
                        else if (!LITE && d.epi == ST_EPI_GRU) o = (1.0f - x1) * e.x2[i][jn][r] + x1 * v[r];
                        else if (d.epi == ST_EPI_AXPY) o = fmaf(e.sc, v[r], x1);
                        buf_st(o, rc, vc, (unsigned)(ST_EPI_ROW(r) * d.ldc) * 4u);
                        if (planes) buf_st_planes(o, rp, vp, (unsigned)(ST_EPI_ROW(r) * 64), plane_b);
                    }
                }
            }
        }
    }
}

template <int TM, int TN>
__device__ __forceinline__ void gemm_tile_epilogue(const st_gemm_desc& d, float* __restrict__ C, f32x16 (&acc)[TM][TN], int m0, int n0,
                                                   int wm, int wn, int li, int lh, int split, int kz) {
    EpiOperands<TM, TN> e;
    gemm_epilogue_consts<TM, TN>(d, e, n0, wn, li, split);
    gemm_epilogue_load<TM, TN>(d, e, m0, n0, wm, wn, li, lh, split);
    gemm_epilogue_store<TM, TN>(d, C, acc, e, m0, n0, wm, wn, li, lh, split, kz);
}

template <int WARPS_M, int WARPS_N, int TM, int TN, bool VEC>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const st_gemm_desc d) {
    constexpr int BM = WARPS_M * TM * 32;
    constexpr int BN = WARPS_N * TN * 32;
    static_assert(WARPS_M * WARPS_N == 4, "4 waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                      // [2][BM][LDS_LD]
    float* Bs = smem + 2 * BM * LDS_LD;    // [2][BN][LDS_LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;

    // grid.z is either the batch index or, for split-K launches (batch == 1), the K slice
    const int split = d.split_k > 1 ? d.split_k : 1;
    const int bz = split > 1 ? 0 : blockIdx.z;
    const int kz = split > 1 ? blockIdx.z : 0;
    const float* __restrict__ X = d.a + (size_t)bz * d.batch_stride_a;
    const float* __restrict__ Wt = d.w + (size_t)bz * d.batch_stride_w;
    float* __restrict__ C = d.c + (size_t)bz * d.batch_stride_c;

    // XCD-aware tile order: consecutive block ids are dealt round-robin over the 8 XCDs, so give
    // each XCD a contiguous run of tiles (neighbouring tiles share operand panels in its L2).
    const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % ntn, tile_m = bid / ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int K = d.K;
    const int dil_h = d.dh > 1 ? d.dh : 1, dil_w = d.dw > 1 ? d.dw : 1;      // conv dilation (0 / 1 = dense)
    const int kcol = (tid & 7) * 4;   // this thread's float4 column inside a BK chunk
    const int rrow = tid >> 3;        // 0..31

    // per-thread A rows: decode m -> (b, oy, ox) once
    constexpr int AP = BM / 32, BP = BN / 32;
    int a_pix[AP];      // pixel index of (b, iy0, ix0) top-left, or <0 row invalid
    int a_iy0[AP], a_ix0[AP], a_b[AP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        const int m = m0 + rrow + 32 * p;
        if (m < d.M) {
            const int hw = d.Ho * d.Wo;
            const int b = m / hw, r = m - b * hw;
            const int oy = r / d.Wo, ox = r - oy * d.Wo;
            a_b[p] = b; a_iy0[p] = oy * d.sh - d.ph; a_ix0[p] = ox * d.sw - d.pw;
            a_pix[p] = 0;
        } else { a_pix[p] = -1; a_b[p] = 0; a_iy0[p] = 0; a_ix0[p] = 0; }
    }

    // descriptors over the whole A / W extents of this batch slice (wave-uniform: kernel args + blockIdx)
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (int)d.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wt), 0, (int)d.w_bytes, 0x00020000);
    float4 ra[AP], rb[BP];
    // VEC path state: this thread's (ky, kx, c) for the tile being loaded, advanced incrementally
    // (one integer division per kernel instead of two per K step); loads are branch-free -- an
    // out-of-range tap / row / k reads a safe address and is zeroed by a select, so the compiler
    // can issue all of a step's global loads back to back.
    int t_ky = 0, t_kx = 0, t_c = 0, t_kt = -2;   // -2: the first tile always decodes by division
    auto load_tile = [&](int kt) {
        const int k = kt * BK + kcol;
        if constexpr (VEC) {
            if (d.kh * d.kw > 1) {
                if (t_kt + 1 == kt && d.Cin >= BK) {
                    t_c += BK;
                    if (t_c >= d.Cin) { t_c -= d.Cin; if (++t_kx == d.kw) { t_kx = 0; ++t_ky; } }
                } else {
                    const int kyx = k / d.Cin;
                    t_c = k - kyx * d.Cin; t_ky = kyx / d.kw; t_kx = kyx - t_ky * d.kw;
                }
                t_kt = kt;
            } else { t_c = k; }
            const bool kin = k < K;
#pragma unroll
            for (int p = 0; p < AP; ++p) {
                const int iy = a_iy0[p] + t_ky * dil_h, ix = a_ix0[p] + t_kx * dil_w;
                const bool ok = kin && a_pix[p] >= 0 && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
                const unsigned off = ok ? (unsigned)(((a_b[p] * d.H + iy) * d.W + ix) * d.ldx + t_c) * 4u : ST_OOB;
                ra[p] = buf_load16(rsrcA, off);
            }
#pragma unroll
            for (int p = 0; p < BP; ++p) {
                const int n = n0 + rrow + 32 * p;
                const bool ok = kin && n < d.N;
                const unsigned off = ok ? (unsigned)(n * d.ldw + k) * 4u : ST_OOB;
                rb[p] = buf_load16(rsrcW, off);
            }
        } else {
#pragma unroll
            for (int p = 0; p < AP; ++p) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ke = k + e;
                    float x = 0.f;
                    if (a_pix[p] >= 0 && ke < K) {
                        const int kyx = ke / d.Cin, c = ke - kyx * d.Cin;
                        const int ky = kyx / d.kw, kx = kyx - ky * d.kw;
                        const int iy = a_iy0[p] + ky * dil_h, ix = a_ix0[p] + kx * dil_w;
                        if (iy >= 0 && iy < d.H && ix >= 0 && ix < d.W)
                            x = X[((size_t)(a_b[p] * d.H + iy) * d.W + ix) * d.ldx + c];
                    }
                    v[e] = x;
                }
                ra[p] = make_float4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int p = 0; p < BP; ++p) {
                const int n = n0 + rrow + 32 * p;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (n < d.N && k + e < K) ? Wt[(size_t)n * d.ldw + k + e] : 0.f;
                rb[p] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int p = 0; p < AP; ++p)
            *reinterpret_cast<float4*>(As + ((size_t)buf * BM + rrow + 32 * p) * LDS_LD + kcol) = ra[p];
#pragma unroll
        for (int p = 0; p < BP; ++p)
            *reinterpret_cast<float4*>(Bs + ((size_t)buf * BN + rrow + 32 * p) * LDS_LD + kcol) = rb[p];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x16 tot[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
    const int nkt_all = (K + BK - 1) / BK;
    const int per = (nkt_all + split - 1) / split;
    const int kt0 = kz * per;
    const int nkt = min(nkt_all, kt0 + per);
    const int li = lane & 31, lh = lane >> 5;
    if (kt0 < nkt) {
        load_tile(kt0);
        store_tile(kt0 & 1);
    }
    __syncthreads();
    for (int kt = kt0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);   // global loads in flight under the MFMAs
        const float* Ab = As + ((size_t)buf * BM + wm * TM * 32 + li) * LDS_LD + 4 * lh;
        const float* Bb = Bs + ((size_t)buf * BN + wn * TN * 32 + li) * LDS_LD + 4 * lh;
#pragma unroll
        for (int j = 0; j < BK / 8; ++j) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_LD + 8 * j);
#pragma unroll
            for (int i = 0; i < TN; ++i) bf[i] = *reinterpret_cast<const float4*>(Bb + i * 32 * LDS_LD + 8 * j);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[jn].x, acc[i][jn], 0, 0, 0);
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[jn].y, acc[i][jn], 0, 0, 0);
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[jn].z, acc[i][jn], 0, 0, 0);
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[jn].w, acc[i][jn], 0, 0, 0);
                }
        }
        if (kt + 1 < nkt) store_tile(buf ^ 1);
        // K-blocked accumulation, same block boundaries as conv_gemm_dma_kernel (every 8 K steps = 256 k): see there
        if (((kt - kt0 + 1) & 7) == 0 && kt + 1 < nkt) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    tot[i][jn] = tot[i][jn] + acc[i][jn];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;
                }
        }
        __syncthreads();
    }
    if (nkt - kt0 > 8) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) acc[i][jn] = acc[i][jn] + tot[i][jn];
    }

    gemm_tile_epilogue<TM, TN>(d, C, acc, m0, n0, wm, wn, li, lh, split, kz);
}

// ---------------------------------------------------------------------------------------------
// Deep-pipelined variant of the same contraction: operand tiles go global -> LDS directly
// (buffer_load_dwordx4 ... lds, no VGPR staging, no ds_write), STAGES tiles deep, one raw s_barrier per
// K step with a counted s_waitcnt vmcnt so that STAGES-2 tiles stay in flight across every barrier.
// This is what keeps a workgroup's MFMAs fed when it is alone on its CU (the M = 8192 decoder GEMMs launch
// 256..512 workgroups): the register-staged kernel above can hold one tile in flight, whose L2 latency is
// longer than the 16 MFMAs of a K step.
//   LDS image of a stage: (BM + BN) rows x 128 B (32 k), unpadded -- one DMA wave-instruction writes 1 KiB
//   = 8 whole rows, lane l -> row l/8, 16-B slot l%8.  Bank conflicts are avoided by an XOR swizzle applied
//   on the SOURCE side: slot s of row r holds k-chunk s ^ ((r >> 1) & 7); the fragment reads
//   (32 rows x one chunk per ds_read_b128) then hit 16 distinct bank quads in each hardware lane group.
//   Out-of-image taps use an offset past the descriptor's num_records: the DMA writes zeros (probed on gfx950).
//   The DMA is issued from inline asm: hipcc treats the builtin form as a pending LDS write and puts
//   s_waitcnt vmcnt(0) in front of the next ds_read, which would drain the whole pipeline every K step.
// Preconditions (checked by the host): 16-B aligned operands, Cin % 32 == 0 (a K step never straddles taps).
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct st_true { static constexpr bool value = true; };
struct st_false { static constexpr bool value = false; };

// M0 is not used by anything else in these kernels (gfx9 DS ops do not need it), so it is simply overwritten.
__device__ __forceinline__ void lds_dma16(i32x4 rsrc, unsigned lds_byte_addr, unsigned voff, unsigned soff) {
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(lds_byte_addr), "v"(voff), "s"(rsrc), "s"(soff)
        : "memory");
}

__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));   // stride 0, no swizzle
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

// The K loop is written around one measured fact (tools/probes/mfma_gap.hip: v_mfma_f32_32x32x2_f32 shares the fp32 FMA
// datapath with the VALU): every VALU instruction issued between two MFMAs costs 4-8 cycles of matrix time
// (2 VALU per MFMA: 67 -> 89 cycles per MFMA at one wave per SIMD), while SALU (<= 4 per MFMA), ds_read and
// buffer_load...lds issue for free.  So the loop body contains NO per-tile VALU work:
//   - fragment reads use loop-invariant address VGPRs + immediate offsets (the loop is unrolled by STAGES so the
//     stage offset is a literal);
//   - DMA source addresses are loop-invariant VGPR offsets + an SGPR offset advanced on the SALU; the only VALU
//     left is the padding test, redone when the (ky, kx) tap changes (every Cin/32 tiles), never for 1x1;
//   - waits, barrier, loop control are scalar.
// PERSIST (plain matrices, K % (32*STAGES) == 0, no split-K): a workgroup walks M tiles g, g+G, g+2G, ... of its
// column tile with ONE continuous DMA ring, so the operand tiles of the next M tile stream in under the epilogue
// of the current one (short-K GEMMs -- K = 128 has four K steps per tile -- are otherwise all load latency).
template <int WM, int WN, int TM, int TN, int STAGES, bool PERSIST, bool CT = false>
__device__ __forceinline__ void conv_gemm_dma_body(const st_gemm_desc& d, const int block_id) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, ROWS = BM + BN;
    constexpr int PA = BM / 32, PB = BN / 32;      // 1-KiB pieces (8 rows x 128 B) per wave and K step
    constexpr int PPW = PA + PB;
    constexpr int STAGE_FLOATS = ROWS * 32;
    extern __shared__ __attribute__((aligned(1024))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int split = d.split_k > 1 ? d.split_k : 1;
    const int bz = split > 1 ? 0 : blockIdx.z;
    const int kz = split > 1 ? blockIdx.z : 0;
    const float* __restrict__ X = d.a + (size_t)bz * d.batch_stride_a;
    const float* __restrict__ Wt = d.w + (size_t)bz * d.batch_stride_w;
    float* __restrict__ C = d.c + (size_t)bz * d.batch_stride_c;

    const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
    const int nwg = PERSIST ? (int)gridDim.x : ntm * ntn;
    int bid = block_id;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % ntn, tile_m = bid / ntn;        // PERSIST: tile_m = first M tile, stride G
    const int G = PERSIST ? nwg / ntn : 1;
    const int nmt = PERSIST ? (ntm - tile_m + G - 1) / G : 1;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const i32x4 rsrcA = make_rsrc(X, d.a_bytes), rsrcW = make_rsrc(Wt, d.w_bytes);
    // optional second A source (same geometry): channels below a2_channels come from it -- a wave-uniform descriptor
    // choice per issued piece, on the SALU
    const i32x4 rsrcA2 = make_rsrc(d.a2 ? d.a2 : X, d.a_bytes);
    const int a2c = (!PERSIST && d.a2) ? d.a2_channels : 0;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;   // LDS byte address

    // staging slots of this lane: wave w stages A rows [8*PA*w, 8*PA*(w+1)) and B rows [8*PB*w, 8*PB*(w+1)),
    // 8 rows per piece, lane l -> row l/8 of the piece, 16-B slot l%8 holding k-chunk (l%8) ^ ((row>>1)&7)
    int a_base[PA], a_iy0[PA], a_ix0[PA];
    unsigned voffA[PA], voffB[PB];
    auto set_mtile = [&](int mbase) {               // per-lane decomposition (image, oy, ox) of the A rows of the M tile starting at mbase
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int r = 8 * (wave * PA + i) + (lane >> 3);
            const int chunk4 = (((lane & 7) ^ ((r >> 1) & 7)) << 2);
            const int m = min(mbase + r, d.M - 1);                              // rows past M: any valid row (never stored)
            const int hw = d.Ho * d.Wo;
            const int b = m / hw, rr = m - b * hw;
            const int oy = rr / d.Wo, ox = rr - oy * d.Wo;
            a_iy0[i] = oy * d.sh - d.ph; a_ix0[i] = ox * d.sw - d.pw;
            a_base[i] = ((b * d.H + a_iy0[i]) * d.W + a_ix0[i]) * d.ldx + chunk4;
        }
    };
    set_mtile(m0);
    // PERSIST walks plain matrices only (rows are contiguous: set_rows).  (Round 5 also let convolutions walk that way, re-deriving the tap
    // addresses at every M-tile change: measured -1.9 % in the product, DESIGN.md section 5, and removed again in round 6 -- ADVICE r5.)
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int r = 8 * (wave * PB + i) + (lane >> 3);                    // row inside the B part; BM % 16 == 0
        const int chunk4 = (((lane & 7) ^ ((r >> 1) & 7)) << 2);
        voffB[i] = (unsigned)(min(n0 + r, d.N - 1) * d.ldw + chunk4) * 4u;
    }

    const int nkt_all = d.K / 32;
    const int per = (nkt_all + split - 1) / split;
    const int kt0 = kz * per;
    const int nkt = min(nkt_all, kt0 + per) - kt0;          // K steps per output tile
    const int ntiles = PERSIST ? nmt * nkt : nkt;           // flat (M tile, K step) sequence of this workgroup

    // wave-uniform state of the next tile to issue: tap (ky, kx), channel offset c0; byte offsets for the SGPR operand
    int i_c0, i_ky, i_kx, i_kt = 0, i_m0 = m0;
    {
        const int k = kt0 * 32, tap = k / d.Cin;
        i_c0 = k - tap * d.Cin; i_ky = tap / d.kw; i_kx = tap - i_ky * d.kw;
    }
    unsigned soffA = (unsigned)i_c0 * 4u, soffB = (unsigned)kt0 * 128u;
    auto set_tap = [&]() {                          // per-lane source offsets of the A pieces for tap (i_ky, i_kx)
        const int ty = i_ky * (d.dh > 1 ? d.dh : 1), tx = i_kx * (d.dw > 1 ? d.dw : 1);      // dilated tap offset
        const int tapbase = (ty * d.W + tx) * d.ldx;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const bool ok = (unsigned)(a_iy0[i] + ty) < (unsigned)d.H && (unsigned)(a_ix0[i] + tx) < (unsigned)d.W;
            voffA[i] = ok ? (unsigned)(a_base[i] + tapbase) * 4u : ST_OOB;
        }
    };
    auto set_rows = [&]() {                         // PERSIST: A rows of the M tile starting at i_m0 (plain matrix)
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int r = 8 * (wave * PA + i) + (lane >> 3);
            voffA[i] = (unsigned)(min(i_m0 + r, d.M - 1) * d.ldx + (((lane & 7) ^ ((r >> 1) & 7)) << 2)) * 4u;
        }
    };
    if (PERSIST) set_rows(); else set_tap();
    auto issue_a = [&](int stage, int i) {
        lds_dma16(i_c0 < a2c ? rsrcA2 : rsrcA, lds0 + (unsigned)(stage * STAGE_FLOATS + (wave * PA + i) * 256) * 4u, voffA[i], soffA);
    };
    auto issue_b = [&](int stage, int i) {
        lds_dma16(rsrcW, lds0 + (unsigned)(stage * STAGE_FLOATS + BM * 32 + (wave * PB + i) * 256) * 4u, voffB[i], soffB);
    };
    auto advance = [&]() {                          // scalar, except the address refresh at a tap / M-tile change
        soffB += 128u; soffA += 128u;
        if (PERSIST) {
            if (++i_kt == nkt) {
                i_kt = 0; soffA = 0; soffB = 0; i_m0 += G * BM;
                set_rows();
            }
        } else {
            i_c0 += 32;
            if (i_c0 >= d.Cin) {
                i_c0 = 0; soffA = 0;
                if (++i_kx == d.kw) { i_kx = 0; ++i_ky; }
                set_tap();
            }
        }
    };
    // wait until at most `tiles` whole tiles of this wave's DMA pieces are still in flight (wave-uniform)
    auto wait_tiles = [&](int tiles) {
        if (tiles >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
        else if (tiles == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else if (tiles == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // K-blocked accumulation: one fmaf chain over all of K has a rounding error that grows like sqrt(K); the reference's
    // CPU GEMM (MKL) accumulates in K blocks of a few hundred, and at K >= 512 a single chain is measurably worse than it
    // (rms 5.7e-7 vs 3.4e-7 at K = 1024, 8.6e-7 vs 3.1e-7 at K = 2304; profiles/r2_parity_trace_*.txt).  So the chain is
    // cut every KBLK K steps (256 k): the running tile is folded into `tot` and restarted from zero.  16 v_add per
    // 32x32 sub-tile every 128 MFMAs; K <= 256 never folds.
    constexpr int KBLK = 8;                         // K steps per block (256 k), whatever the ring depth
    f32x16 tot[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
    auto fold = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                tot[i][j] = tot[i][j] + acc[i][j];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto unfold = [&]() {                           // result of a blocked tile back into acc (epilogue operand)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = acc[i][j] + tot[i][j];
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
            }
        __builtin_amdgcn_sched_barrier(0);
    };

    const int li = lane & 31, lh = lane >> 5;
    const int swz = (li >> 1) & 7;
    // loop-invariant fragment pointers (one per 8-k step: the XOR swizzle is lane dependent); stage / sub-tile
    // offsets are literals folded into the ds_read offset field
    const float* pa[4];
    const float* pb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int foff = ((2 * j + lh) ^ swz) << 2;
        pa[j] = smem + (wm * TM * 32 + li) * 32 + foff;
        pb[j] = smem + (BM + wn * TN * 32 + li) * 32 + foff;
    }

    float4 fa[2][TM], fb[2][TN];                   // fragment ping-pong: step j+1 is read under the MFMAs of step j
    auto read_frags = [&](int buf, int stage, int j) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[buf][i] = *reinterpret_cast<const float4*>(pa[j] + stage * STAGE_FLOATS + i * 1024);
#pragma unroll
        for (int i = 0; i < TN; ++i) fb[buf][i] = *reinterpret_cast<const float4*>(pb[j] + stage * STAGE_FLOATS + i * 1024);
    };
    auto comp = [](const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; };
    auto mfma_k = [&](int buf, int e) {            // the TM*TN MFMAs of one k (k = 8j + 4*lane_half + e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(fa[buf][i], e), comp(fb[buf][jn], e), acc[i][jn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
#define ST_GAP(stmt) do { stmt; __builtin_amdgcn_sched_barrier(0); } while (0)

    // epilogue operands are fetched ahead of the K loop (ordinary loads, older than every DMA of the ring: the
    // counted vmcnt waits below only ever over-wait because of them)
    EpiOperands<TM, TN> eop;
    gemm_epilogue_consts<TM, TN>(d, eop, n0, wn, li, split);                 // once per workgroup (also for every M tile of PERSIST)
    if (!PERSIST) gemm_epilogue_load<TM, TN>(d, eop, m0, n0, wm, wn, li, lh, split);
    __builtin_amdgcn_sched_barrier(0);

    if (ntiles > 0) {
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t)
            if (t < ntiles) {
#pragma unroll
                for (int i = 0; i < PA; ++i) issue_a(t, i);
#pragma unroll
                for (int i = 0; i < PB; ++i) issue_b(t, i);
                advance();
            }
        wait_tiles(min(STAGES - 2, ntiles - 1));    // tile 0 landed (this wave's pieces) ...
        asm volatile("s_barrier" ::: "memory");     // ... and everybody else's
        read_frags(0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // one tile: stage s (literal), tile index t.  FULL = steady state (tile t+STAGES-1 exists): no branches at all.
        auto tile_body = [&](auto full_c, int s, int t) {
            constexpr bool FULL = decltype(full_c)::value;
            const bool more = FULL || t + 1 < ntiles, feed = FULL || t + STAGES - 1 < ntiles;
            const int fstage = (s + STAGES - 1) % STAGES, nstage = (s + 1) % STAGES;
            // ---- k-step 0 (fragments in buffer 0)
            mfma_k(0, 0); ST_GAP(read_frags(1, s, 1));
            mfma_k(0, 1);
            // tile t+1 has landed everywhere and every wave is past tile t-1, whose stage is therefore free
            if (FULL) ST_GAP(wait_tiles(STAGES - 3); asm volatile("s_barrier" ::: "memory"));
            else ST_GAP(if (more) { wait_tiles(min(STAGES - 3, ntiles - 2 - t)); asm volatile("s_barrier" ::: "memory"); });
            mfma_k(0, 2); ST_GAP(if (feed) issue_a(fstage, 0));
            mfma_k(0, 3); ST_GAP(if (feed) { _Pragma("unroll") for (int i = 1; i < PA; ++i) issue_a(fstage, i); });
            // ---- k-step 1 (buffer 1)
            mfma_k(1, 0); ST_GAP(read_frags(0, s, 2));
            mfma_k(1, 1); ST_GAP(if (feed) issue_b(fstage, 0));
            mfma_k(1, 2); ST_GAP(if (feed) { _Pragma("unroll") for (int i = 1; i < PB; ++i) issue_b(fstage, i); });
            mfma_k(1, 3); ST_GAP(if (feed) advance());
            // ---- k-step 2 (buffer 0)
            mfma_k(0, 0); ST_GAP(read_frags(1, s, 3));
            mfma_k(0, 1); mfma_k(0, 2); mfma_k(0, 3);
            // ---- k-step 3 (buffer 1); step 0 of the next tile is read under it (a stale read after the last tile)
            mfma_k(1, 0); ST_GAP(read_frags(0, nstage, 0));
            mfma_k(1, 1); mfma_k(1, 2); mfma_k(1, 3);
            if (STAGES != 4 && (t & (KBLK - 1)) == KBLK - 1 && t + 1 < ntiles) fold();    // (4-deep ring: folded by the callers, at block ends)
        };
        // (round 4 measured ONE barrier per two K steps -- the ring used in stage pairs, an even tile issuing the whole next pair's DMA, the odd
        // tile closing the pair with vmcnt(0) + s_barrier: bit-identical, 81.85 -> 81.65 pairs/s and 72.0 -> 71.2 with one pair in flight: what
        // this loop waits for is the DMA's distance ahead (3 tiles here, 2 in the pair scheme), not the barrier itself)
        if (PERSIST) {
            for (int mt = 0; mt < nmt; ++mt) {
                gemm_epilogue_load<TM, TN>(d, eop, (tile_m + mt * G) * BM, n0, wm, wn, li, lh, 1);   // lands under the MFMAs
                __builtin_amdgcn_sched_barrier(0);
                for (int tb = mt * nkt, kb = STAGES; tb < (mt + 1) * nkt; tb += STAGES, kb += STAGES) {    // nkt % STAGES == 0: stage == s
                    if (tb + 2 * STAGES - 1 <= ntiles) {
#pragma unroll
                        for (int s = 0; s < STAGES; ++s) tile_body(st_true{}, s, tb + s);
                    } else {
#pragma unroll
                        for (int s = 0; s < STAGES; ++s) tile_body(st_false{}, s, tb + s);
                    }
                    if (kb % KBLK == 0 && kb < nkt) fold();
                }
                if (nkt > KBLK) unfold();
                gemm_epilogue_store<TM, TN, false, CT>(d, C, acc, eop, (tile_m + mt * G) * BM, n0, wm, wn, li, lh, 1, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
        } else {
            int tb = 0;
            for (; tb + 2 * STAGES - 1 <= ntiles; tb += STAGES) {       // every tile of the block still feeds a new one
#pragma unroll
                for (int s = 0; s < STAGES; ++s) tile_body(st_true{}, s, tb + s);
                if (STAGES == 4 && (tb + STAGES) % KBLK == 0) fold();
            }
            for (; tb < ntiles; tb += STAGES) {
#pragma unroll
                for (int s = 0; s < STAGES; ++s)
                    if (tb + s < ntiles) tile_body(st_false{}, s, tb + s);
                if (STAGES == 4 && (tb + STAGES) % KBLK == 0 && tb + STAGES < ntiles) fold();
            }
            if (ntiles > KBLK) unfold();
        }
    }
#undef ST_GAP

    if (!PERSIST) gemm_epilogue_store<TM, TN, false, CT>(d, C, acc, eop, m0, n0, wm, wn, li, lh, split, kz);
}

template <int WM, int WN, int TM, int TN, int STAGES, bool PERSIST, bool CT = false>
__global__ __launch_bounds__(256) void conv_gemm_dma_kernel(const st_gemm_desc d) {
    conv_gemm_dma_body<WM, WN, TM, TN, STAGES, PERSIST, CT>(d, (int)blockIdx.x);
}

// Two independent contractions in ONE launch (st_conv_gemm_pair): workgroups [0, tiles0) run d[0], the rest d[1].  For pairs of
// mid-size convs that are ready at the same time and each fill only part of the chip (BasicMotionEncoder's convc2: 384 tiles, and
// convf2: 128 tiles, gru.py:252-253): together they give every CU two workgroups without split-K slabs or a second launch.
struct st_gemm_pair_args {
    st_gemm_desc d[2];
    int32_t tiles0;
};
template <int WM, int WN, int TM, int TN, int STAGES>
__global__ __launch_bounds__(256) void conv_gemm_dma_pair_kernel(const st_gemm_pair_args g) {
    const bool second = (int)blockIdx.x >= g.tiles0;             // workgroup-uniform
    conv_gemm_dma_body<WM, WN, TM, TN, STAGES, false>(second ? g.d[1] : g.d[0], second ? (int)blockIdx.x - g.tiles0 : (int)blockIdx.x);
}

#include "gemm_split3.h"

// ---------------------------------------------------------------------------------------------
// Row-streaming GEMM for the short-K linears (K = 64 / 128: the Twins / latent / vertical-layer linears at
// M = 32768 ... 524288: twins.py:253-304,336-392,785, encoder.py:156-172, crossattentionlayer.py:37-56).  Those shapes are
// MFMA-bound and HBM-bound at once (64 MFMAs per 64x64 tile against 48 KB of traffic), so the tiled kernels pay for
// every barrier and for fetching A once per column tile.  Here:
//   - the weight slice of the workgroup (<= 8 chunks of 32 columns x K) is staged into LDS ONCE and stays there
//     ([n][K + 4] padded rows: the 16 lanes of a ds_read_b128 group hit 16 distinct bank quads), with its bias;
//   - a wave owns whole 32-row blocks of A and keeps a block in REGISTERS (K / 2 floats per lane, loaded straight from
//     global memory with 16-byte buffer loads; lane (li, lh) holds k = 8j + 4lh + t of row li -- the same k pairing and
//     the same summation order as the tiled kernels, so results are bit-identical to theirs); for K <= 128 the next
//     block is prefetched under the MFMAs of the current one (in two halves, each issued AFTER the epilogue operand
//     loads of its chunk: vmcnt retires in order, so an epilogue must never have to wait for the prefetch);
//   - so the main loop has no barrier, no LDS traffic for A, and each wave runs independently: for every 32-column
//     chunk, K/2 MFMAs fed by K/8 ds_read_b128, then the shared epilogue (a half-chunk start delay for waves 4-7, so
//     that the two waves of a SIMD do not run their epilogues side by side, was measured neutral to slightly worse);
//   - because a lane pair holds a complete row, LayerNorm of the A rows (d.a_ln) is two in-register reductions and one
//     cross-half exchange: the LayerNorm kernel in front of these linears, its write and its re-read disappear.
// Column slices (more chunks than a workgroup keeps, or too few row blocks to give every wave one) are separate
// workgroups that walk the same rows; the slices of a row group get block ids that land on the same XCD (shared L2).
template <int KC, bool PREFETCH>
__global__ __launch_bounds__(512) void rowstream_gemm_kernel(const st_gemm_desc d, const int chunks_per_slice, const int nslices) {
    constexpr int K = 32 * KC, LDW = K + 4, NJ = 4 * KC, NW = 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int S = nslices, G = (int)gridDim.x / S;
    const int q = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
    const int sl = q % S, g = (q / S) * 8 + xcd;                 // slice, workgroup index inside the slice (0 .. G-1)
    const int n_base = sl * chunks_per_slice * 32;
    const int ncols = min(d.N - n_base, chunks_per_slice * 32);
    const int nch = (ncols + 31) >> 5;
    const int nblk = (d.M + 31) >> 5;
    const int stride = G * NW;
    float* bias_lds = smem + chunks_per_slice * 32 * LDW;

    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.a), 0, (int)d.a_bytes, 0x00020000);
    auto row_off = [&](int blk) {
        const int row = blk * 32 + li;
        return (blk < nblk && row < d.M) ? (unsigned)(row * d.ldx + 4 * lh) * 4u : ST_OOB;
    };
    int blk = g * NW + wave;
    float4 a[NJ], an[PREFETCH ? NJ : 1];
    {
        const unsigned off = row_off(blk);                       // in flight while the weight slice is staged
#pragma unroll
        for (int j = 0; j < NJ; ++j) a[j] = buf_load16(rsrcA, off + 32u * j);
    }
    {
        // weight slice -> LDS, 8 float4 per thread in flight at a time
        const int total = nch * 32 * (K / 4);
        for (int f0 = 0; f0 < total; f0 += 512 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int f = f0 + u * 512 + tid, r = f / (K / 4), c4 = f - r * (K / 4);
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (f < total && r < ncols) v[u] = *reinterpret_cast<const float4*>(d.w + (size_t)(n_base + r) * d.ldw + c4 * 4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int f = f0 + u * 512 + tid, r = f / (K / 4), c4 = f - r * (K / 4);
                if (f < total) *reinterpret_cast<float4*>(smem + r * LDW + c4 * 4) = v[u];
            }
        }
        if (tid < nch * 32) bias_lds[tid] = (d.bias && tid < ncols) ? d.bias[n_base + tid] : 0.f;
    }
    const float sc = d.scale_ptr ? *d.scale_ptr : 1.0f;
    __syncthreads();

    const float* wl = smem + li * LDW + 4 * lh;
    EpiOperands<1, 1> e;
    e.sc = sc;
    for (; blk < nblk; blk += stride) {
        const int m0 = blk * 32;
        const unsigned noff = PREFETCH ? row_off(blk + stride) : 0u;
        if (d.a_ln) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) s += (a[j].x + a[j].y) + (a[j].z + a[j].w);
            s += __shfl_xor(s, 32, 64);
            const float mean = s * (1.0f / K);
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                a[j].x -= mean; a[j].y -= mean; a[j].z -= mean; a[j].w -= mean;
                v += (a[j].x * a[j].x + a[j].y * a[j].y) + (a[j].z * a[j].z + a[j].w * a[j].w);
            }
            v += __shfl_xor(v, 32, 64);
            const float rstd = 1.0f / sqrtf(v * (1.0f / K) + d.a_ln_eps);
#pragma unroll
            for (int j = 0; j < NJ; ++j) { a[j].x *= rstd; a[j].y *= rstd; a[j].z *= rstd; a[j].w *= rstd; }
        }
        for (int c = 0; c < nch; ++c) {
            const int n0 = n_base + c * 32;
            e.bv[0] = bias_lds[c * 32 + li];
            gemm_epilogue_load<1, 1, true>(d, e, m0, n0, 0, 0, li, lh, 1);
            if (PREFETCH) {
                if (c == 0) {
#pragma unroll
                    for (int j = 0; j < NJ / 2; ++j) an[j] = buf_load16(rsrcA, noff + 32u * j);
                }
                if (c == 1 || nch == 1) {
#pragma unroll
                    for (int j = NJ / 2; j < NJ; ++j) an[PREFETCH ? j : 0] = buf_load16(rsrcA, noff + 32u * j);
                }
            }
            f32x16 acc[1][1];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
            const float* wb = wl + c * 32 * LDW;
            float4 b = *reinterpret_cast<const float4*>(wb);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // the fragment of step j+1 is read under the MFMAs of step j
                const float4 bn = *reinterpret_cast<const float4*>(wb + 8 * (j + 1 < NJ ? j + 1 : j));
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].x, b.x, acc[0][0], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].y, b.y, acc[0][0], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].z, b.z, acc[0][0], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].w, b.w, acc[0][0], 0, 0, 0);
                b = bn;
                __builtin_amdgcn_sched_barrier(0);
            }
            gemm_epilogue_store<1, 1, true>(d, d.c, acc, e, m0, n0, 0, 0, li, lh, 1, 0);
        }
        if (PREFETCH) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) a[j] = an[j];
        } else if (blk + stride < nblk) {
            const unsigned off = row_off(blk + stride);
#pragma unroll
            for (int j = 0; j < NJ; ++j) a[j] = buf_load16(rsrcA, off + 32u * j);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row chain: up to three Linear(128 -> 128) layers applied to 32-row blocks that never leave the CU (st_linear_chain128).
// Built on the row-streaming kernel: a wave keeps its block in registers in the A-operand layout (lane (li, lh) holds
// k = 8j + 4lh + t of row li); a layer is four 32-column chunks of 64 MFMAs computed as the TRANSPOSED product (weights first
// operand, activations second): a chunk's accumulators are then row li's output features 8 jj + 4 lh + t, i.e. four more float4 of
// the NEXT layer's A operand -- bias, activation, LayerNorm and the residual adds happen in that layout, in registers, and nothing
// crosses LDS between layers.  Weights stream through a 3-stage LDS ring of 32-row chunks shared by the waves of the workgroup
// (LDS DMA, XOR-swizzled 128-B-row image as in conv_gemm_dma_kernel; one barrier per chunk; the DMA of chunk q + 2 is issued at the
// start of step q, but every step opens with s_waitcnt vmcnt(0), so a chunk has ONE step -- 64 MFMAs per wave -- to land, not two).
// Per layer the k pairing and summation order are those of the other kernels: bit-identical.
#define RC_NW 4                                                  // waves per workgroup, two workgroups per CU (starting half of them half a
                                                                 // step late so that co-resident waves run out of phase: measured neutral)
__global__ __launch_bounds__(256, 2) void rowchain128_kernel(const st_chain_desc d) {
    constexpr int NJ = 16, NW = RC_NW;
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    float* ring = smem;                                        // [3][32 rows][128 k] unpadded, 16-B slots XOR-swizzled by (row & 15)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int nblk = (d.M + 31) >> 5;
    const int G = (int)gridDim.x;
    const int blk0 = (int)blockIdx.x * NW;
    const int rounds = blk0 < nblk ? (nblk - blk0 + G * NW - 1) / (G * NW) : 0;
    const int L = d.nlayers, steps = 4 * L, total = rounds * steps;
    if (total == 0) return;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;

    // weight chunk q of the (round-periodic) sequence = rows [32c, 32c + 32) of layer (q % steps) / 4 -> ring stage q % 3, by LDS DMA:
    // 16 pieces of 1 KiB (two rows each), 16 / NW per wave; lane l of a piece writes slot l & 31 of row 2p + (l >> 5), which holds
    // k-chunk slot ^ (row & 15) (the swizzle is applied on the source side)
    auto dma_chunk = [&](int q) {
        const int qq = q % steps, l = qq >> 2, c = qq & 3;
        const i32x4 rs = make_rsrc(d.layer[l].w, 128 * 128 * 4);
#pragma unroll
        for (int u = 0; u < 16 / NW; ++u) {
            const int p = wave * (16 / NW) + u, r = 2 * p + (lane >> 5);
            const unsigned voff = (unsigned)(((c * 32 + r) * 128 + (((lane & 31) ^ (r & 15)) << 2)) * 4);
            lds_dma16(rs, lds0 + (unsigned)(((q % 3) * 32 * 128 + p * 256) * 4), voff, 0u);
        }
    };
    dma_chunk(0);
    if (total > 1) dma_chunk(1);
    // fragment slot offsets (floats) of this lane: 16-B slot (2j + lh) ^ (li & 15) -- the XOR touches the low four bits only
    int foff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) foff[j] = (((2 * j + lh) ^ (li & 15)) << 2);

    float4 a[NJ], an[NJ], sv[NJ];
    int q = 0;
    for (int rd = 0; rd < rounds; ++rd) {
        const int blk = blk0 + wave + rd * G * NW;
        const bool active = blk < nblk;                         // wave-uniform; idle waves still load weights and meet the barriers
        const int row = blk * 32 + li;
        const bool rok = active && row < d.M;
        if (active) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                a[j] = rok ? *reinterpret_cast<const float4*>(d.a + (size_t)row * d.lda + 8 * j + 4 * lh) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int l = 0; l < L; ++l) {
            const st_chain_layer& Ly = d.layer[l];
            if (active) {
                // a later layer adds THIS layer's input (before its LN) as residual: keep a copy (parking it in the block's rows of `out`
                // instead frees 64 registers and removes the ~30 spilled ones, but its 67 MB of extra traffic cost 5 us per launch: measured)
                bool keep = false;
                for (int m = l; m < L; ++m) keep = keep || (d.layer[m].res == 2 && d.layer[m].res_layer == l);
                if (keep) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) sv[j] = a[j];
                }
                if (Ly.ln) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) s += (a[j].x + a[j].y) + (a[j].z + a[j].w);
                    s += __shfl_xor(s, 32, 64);
                    const float mean = s * (1.0f / 128.0f);
                    float v = 0.f;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        a[j].x -= mean; a[j].y -= mean; a[j].z -= mean; a[j].w -= mean;
                        v += (a[j].x * a[j].x + a[j].y * a[j].y) + (a[j].z * a[j].z + a[j].w * a[j].w);
                    }
                    v += __shfl_xor(v, 32, 64);
                    const float rstd = 1.0f / sqrtf(v * (1.0f / 128.0f) + Ly.ln_eps);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) { a[j].x *= rstd; a[j].y *= rstd; a[j].z *= rstd; a[j].w *= rstd; }
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c, ++q) {
                // chunk q (DMA issued two steps ago; the vmcnt(0) below also drains chunk q + 1, issued one step ago, so the ring's
                // effective lead is one step) is in the ring once every wave's pieces have landed; everyone is past chunk q - 1,
                // whose stage chunk q + 2 may now overwrite
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (q + 2 < total) dma_chunk(q + 2);
                if (active) {
                    const float* wb = ring + (q % 3) * 32 * 128 + li * 128;
                    // bias of the 16 output features this lane ends up holding (see below): four runs of four
                    float4 bv[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        bv[jj] = Ly.bias ? *reinterpret_cast<const float4*>(Ly.bias + c * 32 + 8 * jj + 4 * lh) : make_float4(0.f, 0.f, 0.f, 0.f);
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    float4 b = *reinterpret_cast<const float4*>(wb + foff[0]);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int jn = j + 1 < NJ ? j + 1 : j;
                        const float4 bn = *reinterpret_cast<const float4*>(wb + foff[jn & 7] + (jn >> 3) * 64);
                        // TRANSPOSED product D[n][m] = sum_k W[n][k] X[m][k]: the weight fragment is the first MFMA operand, the
                        // activations the second (same products, same k order: the same bits as X . W^T).  Lane (li, lh) then holds
                        // row m = li and output features n = (r & 3) + 8 (r >> 2) + 4 lh, r = 0..15 -- which IS the A-operand layout
                        // of the next layer (k = 8 j + 4 lh + t with j = r >> 2, t = r & 3): no trip through LDS between layers.
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a[j].x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a[j].y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a[j].z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a[j].w, acc, 0, 0, 0);
                        b = bn;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    float v[16];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        v[4 * jj] = acc[4 * jj] + bv[jj].x; v[4 * jj + 1] = acc[4 * jj + 1] + bv[jj].y;
                        v[4 * jj + 2] = acc[4 * jj + 2] + bv[jj].z; v[4 * jj + 3] = acc[4 * jj + 3] + bv[jj].w;
                    }
                    if (Ly.act == ST_ACT_GELU) {                 // wave-uniform, outside the register loop (none / relu / gelu only)
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] = st_gelu(v[r]);
                    } else if (Ly.act == ST_ACT_RELU) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) an[4 * c + jj] = make_float4(v[4 * jj], v[4 * jj + 1], v[4 * jj + 2], v[4 * jj + 3]);
                }
            }
            if (active) {
                if (Ly.res == 1) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const float4 x = rok ? *reinterpret_cast<const float4*>(Ly.res_ptr + (size_t)row * Ly.ld_res + 8 * j + 4 * lh) : make_float4(0.f, 0.f, 0.f, 0.f);
                        an[j].x += x.x; an[j].y += x.y; an[j].z += x.z; an[j].w += x.w;
                    }
                } else if (Ly.res == 2) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) { an[j].x += sv[j].x; an[j].y += sv[j].y; an[j].z += sv[j].z; an[j].w += sv[j].w; }
                }
                if (l == L - 1) {
                    if (rok) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j) *reinterpret_cast<float4*>(d.out + (size_t)row * d.ldo + 8 * j + 4 * lh) = an[j];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) a[j] = an[j];
                }
            }
        }
    }
}

extern "C" int st_abi_chain_desc_size(void) { return (int)sizeof(st_chain_desc); }

extern "C" int st_linear_chain128(const st_chain_desc* desc, void* stream) {
    if (!desc) return ST_EINVAL;
    const st_chain_desc& d = *desc;
    if (!d.a || !d.out || d.M <= 0 || d.nlayers < 1 || d.nlayers > 3 || d.lda < 128 || d.ldo < 128 || (d.lda & 3) || (d.ldo & 3) ||
        ((uintptr_t)d.a & 15) || ((uintptr_t)d.out & 15) || (int64_t)d.M * (d.lda > d.ldo ? d.lda : d.ldo) >= ((int64_t)1 << 40))
        return ST_EINVAL;
    for (int l = 0; l < d.nlayers; ++l) {
        const st_chain_layer& y = d.layer[l];
        if (!y.w || ((uintptr_t)y.w & 15) || y.act < 0 || y.act > ST_ACT_GELU || y.res < 0 || y.res > 2) return ST_EINVAL;
        if (y.res == 1 && (!y.res_ptr || y.ld_res < 128 || (y.ld_res & 3) || ((uintptr_t)y.res_ptr & 15))) return ST_EINVAL;
        if (y.res == 2 && (y.res_layer < 0 || y.res_layer > l)) return ST_EINVAL;
        if (y.bias && ((uintptr_t)y.bias & 15)) return ST_EINVAL;        // read with 16-byte loads
    }
    {
        // the kernel keeps ONE saved layer input (`sv`): every res == 2 layer must name the same res_layer -- a second one would
        // overwrite the copy a later layer still needs and silently add the wrong tensor
        int saved = -1;
        for (int l = 0; l < d.nlayers; ++l)
            if (d.layer[l].res == 2) {
                if (saved >= 0 && d.layer[l].res_layer != saved) return ST_EINVAL;
                saved = d.layer[l].res_layer;
            }
    }
    const int nblk = (d.M + 31) / 32;
    int G = (nblk + RC_NW - 1) / RC_NW;
    if (G > 512) G = 512;                                       // two workgroups per CU
    const size_t lds = (size_t)(3 * 32 * 128) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)rowchain128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // the profiling observer sees the chain as one launch of the family: M x (128 * nlayers) x 128 (its FLOPs; the A + W + C byte
    // formula of the tools then counts the intermediate activations that this kernel does NOT move)
    st_gemm_observer_fn obs = g_observer;
    st_gemm_desc od;
    if (obs) {
        memset(&od, 0, sizeof(od));
        od.a = d.a; od.c = d.out; od.w = d.layer[0].w;
        od.M = d.M; od.N = 128 * d.nlayers; od.K = 128; od.H = 1; od.W = d.M; od.Cin = 128; od.ldx = d.lda; od.ldc = d.ldo; od.ldw = 128;
        od.kh = od.kw = od.sh = od.sw = 1; od.Ho = 1; od.Wo = d.M; od.batch = 1; od.alpha = 1.f;
        obs(&od, stream, 0, g_observer_user);
    }
    g_last_plan[0] = 5; g_last_plan[1] = 30; g_last_plan[2] = 1; g_last_plan[3] = 1;
    hipLaunchKernelGGL(rowchain128_kernel, dim3(G), dim3(64 * RC_NW), lds, (hipStream_t)stream, d);
    if (obs) obs(&od, stream, 1, g_observer_user);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// The Twins MLP (timm Mlp, twins.py:785-790: x + fc2(GELU(fc1(LN(x)))), C = 128, hidden = 512) as ONE launch: the hidden
// activations never leave the CU.  Unfused, fc1's [M, 512] tensor is written and read back: 268 MB per MLP at M = 65536, and fc1
// and fc2 are two K = 128 / N = 128 launches at 0.53 / 0.66 of the fp32-MFMA peak (profiles/r3_gemm_shapes.csv).
// Structure = rowchain128_kernel's: a wave owns a 32-row block whose LayerNorm'ed rows sit in registers in the MFMA operand
// layout; the hidden dimension is walked in chunks of 32 features:
//   stage A   h = GELU(W1[32 hc .. +32, :] . x^T + b1)     64 MFMAs, TRANSPOSED product (weights first): lane (li, lh) then holds
//             row li's hidden features 8 j + 4 lh + t of the chunk -- the operand layout of the k slice [32 hc, 32 hc + 32) of fc2
//   stage B   o[oc] += W2[32 oc .. +32, 32 hc .. +32] . h^T, oc = 0..3     64 MFMAs, transposed again: four accumulator tiles whose
//             layout is the block's own row layout, so bias, the residual x (re-read: it is L2-warm) and the optional second
//             residual are added in registers and stored as 16-byte runs.
// Per step the workgroup's four waves share W1's chunk (32 x 128) and W2's slice (128 x 32) through a 2-stage LDS ring filled by
// LDS-DMA one step ahead (32 KB per stage, 64 KB per workgroup, two workgroups per CU), one barrier per step.  k pairing and order
// inside both products are those of the other kernels (k = 8 j + 4 lane_half + t); fc1 is bit-identical to the unfused launch,
// fc2 accumulates its 512 k in ONE chain (the unfused kernels fold at k = 256; holding that fold would need 64 more registers
// per lane than two waves per SIMD have) -- same products, the sum differs in the last bits.
#define MLP_NW 4
template <bool PROJ>
__global__ __launch_bounds__(256, 2) void rowmlp128_kernel(const st_mlp_desc d) {
    constexpr int NJ = 16, NW = MLP_NW, STAGE = 2 * 32 * 128;   // floats per ring stage: [W1 chunk 32 x 128 | W2 slice 128 x 32]
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int nblk = (d.M + 31) >> 5;
    const int G = (int)gridDim.x;
    const int blk0 = (int)blockIdx.x * NW;
    const int rounds = blk0 < nblk ? (nblk - blk0 + G * NW - 1) / (G * NW) : 0;
    // optional leading layer (the attention output projection of the Block, twins.py:622-623 / 676-677): x = a . wp^T + bp + res0,
    // four more steps of 32 output features each in front of the hidden chunks; x then takes a's place
    constexpr bool proj = PROJ;
    const int npre = proj ? 4 : 0;
    const int nhc = d.hidden >> 5, spr = npre + nhc, total = rounds * spr;
    if (total == 0) return;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    const i32x4 rs1 = make_rsrc(d.w1, (unsigned)d.hidden * 128u * 4u), rs2 = make_rsrc(d.w2, 128u * (unsigned)d.hidden * 4u);
    const i32x4 rsp = make_rsrc(proj ? d.wp : d.w1, 128u * 128u * 4u);

    // step q = step s = q % spr of a round -> ring stage q & 1.  s < npre: chunk s of wp (image of a W1 chunk); otherwise hidden chunk
    // hc = s - npre.  W1 chunk: 16 pieces of 1 KiB = two 512-B rows, slot t of row r holds k-chunk t ^ (r & 15) (rowchain128's image).
    // W2 slice: 16 pieces of 1 KiB = eight 128-B rows, slot t of row r holds k-chunk t ^ ((r >> 1) & 7) (conv_gemm_dma's image).
    // The swizzles are applied on the source side; 8 (4 for a wp step) pieces per wave and step.
    auto dma_step = [&](int q) {
        const int s = __builtin_amdgcn_readfirstlane(q % spr);
        const bool pre = PROJ && s < npre;
        const int hc = pre ? s : s - npre;
        const unsigned st = lds0 + (unsigned)((q & 1) * STAGE * 4);
        if (pre) {                                              // (a scalar branch: the descriptor operand of the DMA must be an SGPR quad)
#pragma unroll
            for (int u = 0; u < 16 / NW; ++u) {
                const int p = wave * (16 / NW) + u, r = 2 * p + (lane >> 5);
                lds_dma16(rsp, st + (unsigned)(p * 1024), (unsigned)((((hc << 5) + r) * 128 + (((lane & 31) ^ (r & 15)) << 2)) * 4), 0u);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16 / NW; ++u) {
                const int p = wave * (16 / NW) + u, r = 2 * p + (lane >> 5);
                lds_dma16(rs1, st + (unsigned)(p * 1024), (unsigned)((((hc << 5) + r) * 128 + (((lane & 31) ^ (r & 15)) << 2)) * 4), 0u);
            }
        }
        if (!pre) {
#pragma unroll
            for (int u = 0; u < 16 / NW; ++u) {
                const int p = wave * (16 / NW) + u, r = 8 * p + (lane >> 3);
                const unsigned voff = (unsigned)((r * d.hidden + (((lane & 7) ^ ((r >> 1) & 7)) << 2)) * 4);
                lds_dma16(rs2, st + (unsigned)(32 * 128 * 4 + p * 1024), voff, (unsigned)(hc << 7));
            }
        }
    };
    dma_step(0);
    if (total > 1) dma_step(1);                                 // both stages are free at the start
    int foff[8], goff[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) foff[j] = (((2 * j + lh) ^ (li & 15)) << 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) goff[j] = li * 32 + (((2 * j + lh) ^ ((li >> 1) & 7)) << 2);

    // one K = 128 product of the block with the 32-row weight chunk in ring stage (q & 1): TRANSPOSED (weights first), so lane (li, lh)
    // ends up with row li's output features 8 jj + 4 lh + t of the chunk
    float4 a[NJ];
    auto chunk128 = [&](const float* ws) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        float4 b = *reinterpret_cast<const float4*>(ws + foff[0]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int jn = j + 1 < NJ ? j + 1 : j;
            const float4 bn = *reinterpret_cast<const float4*>(ws + foff[jn & 7] + (jn >> 3) * 64);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a[j].w, acc, 0, 0, 0);
            b = bn;
            __builtin_amdgcn_sched_barrier(0);
        }
        return acc;
    };
    auto step_sync = [&](int q) {
        // step q's weights (DMA issued one step ago; steps 0 and 1 before the loop) are in the ring once every wave's pieces have
        // landed; everyone is past step q - 1, whose stage step q + 1 may now overwrite
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (q > 0 && q + 1 < total) dma_step(q + 1);
    };

    const float* xres = proj ? d.out : d.a;                      // where the rows the MLP adds back live (x is parked in `out` when computed here)
    const int ld_xres = proj ? d.ldo : d.lda;
    int q = 0;
    for (int rd = 0; rd < rounds; ++rd) {
        const int blk = blk0 + wave + rd * G * NW;
        const bool active = blk < nblk;                         // wave-uniform; idle waves still load weights and meet the barriers
        const int row = blk * 32 + li;
        const bool rok = active && row < d.M;
        const size_t rowc = (size_t)(row < d.M ? row : d.M - 1);   // rows past M (last block only) read a valid row and are never stored:
        if (active) {                                           // unconditional loads, no per-lane branches around them
#pragma unroll
            for (int j = 0; j < NJ; ++j) a[j] = *reinterpret_cast<const float4*>(d.a + rowc * d.lda + 8 * j + 4 * lh);
        }
        if (proj) {
            // x = a . wp^T + bp + res0, 32 features per step, written straight to the block's rows of `out` and read back below: x is
            // needed twice (as this MLP's input and as its residual) and holding both a and x in registers next to the accumulators
            // does not fit two waves per SIMD (hipcc spilled 55 registers); the rows are L2-warm when they come back
#pragma unroll 1
            for (int c = 0; c < 4; ++c, ++q) {
                step_sync(q);
                if (active) {
                    // bias and residual of the 16 features this lane ends up holding: requested before the MFMAs, used after them
                    float4 bv[4], ev[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int col = (c << 5) + 8 * jj + 4 * lh;
                        // (wave-uniform conditions: scalar branches, no exec masking)
                        bv[jj] = d.bp ? *reinterpret_cast<const float4*>(d.bp + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                        ev[jj] = d.res0 ? *reinterpret_cast<const float4*>(d.res0 + rowc * d.ld_res0 + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    const f32x16 acc = chunk128(smem + (q & 1) * STAGE + li * 128);
                    if (rok) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            // (acc + bias) + residual: the unfused epilogue's order
                            *reinterpret_cast<float4*>(d.out + rowc * d.ldo + (c << 5) + 8 * jj + 4 * lh) =
                                make_float4((acc[4 * jj] + bv[jj].x) + ev[jj].x, (acc[4 * jj + 1] + bv[jj].y) + ev[jj].y,
                                            (acc[4 * jj + 2] + bv[jj].z) + ev[jj].z, (acc[4 * jj + 3] + bv[jj].w) + ev[jj].w);
                        }
                    }
                }
            }
            if (active) {
                // the stores above are complete (written through to L2; the vector L1 does not allocate on a store, and these rows were
                // never read by this CU before) -> read x back in the operand layout
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < NJ; ++j) a[j] = *reinterpret_cast<const float4*>(d.out + rowc * d.ldo + 8 * j + 4 * lh);
            }
        }
        if (active && d.ln) {                                   // LayerNorm without affine (gamma / beta are folded into w1 / b1): as rowchain128
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) s += (a[j].x + a[j].y) + (a[j].z + a[j].w);
            s += __shfl_xor(s, 32, 64);
            const float mean = s * (1.0f / 128.0f);
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                a[j].x -= mean; a[j].y -= mean; a[j].z -= mean; a[j].w -= mean;
                v += (a[j].x * a[j].x + a[j].y * a[j].y) + (a[j].z * a[j].z + a[j].w * a[j].w);
            }
            v += __shfl_xor(v, 32, 64);
            const float rstd = 1.0f / sqrtf(v * (1.0f / 128.0f) + d.ln_eps);
#pragma unroll
            for (int j = 0; j < NJ; ++j) { a[j].x *= rstd; a[j].y *= rstd; a[j].z *= rstd; a[j].w *= rstd; }
        }
        __builtin_amdgcn_sched_barrier(0);                      // (keeps the 64 accumulator zeros below from being scheduled above the projection)
        f32x16 o[4];
#pragma unroll
        for (int oc = 0; oc < 4; ++oc)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[oc][r] = 0.f;
        for (int hc = 0; hc < nhc; ++hc, ++q) {
            step_sync(q);
            if (active) {
                const float* w2s = smem + (q & 1) * STAGE + 32 * 128;
                float4 bv[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) bv[jj] = *reinterpret_cast<const float4*>(d.b1 + (hc << 5) + 8 * jj + 4 * lh);
                // ---- stage A: hidden chunk, K = 128
                const f32x16 acc = chunk128(smem + (q & 1) * STAGE + li * 128);
                // first W2 fragments of stage B are requested before the GELU arithmetic
                float4 g[4];
#pragma unroll
                for (int oc = 0; oc < 4; ++oc) g[oc] = *reinterpret_cast<const float4*>(w2s + oc * 1024 + goff[0]);
                float4 hq[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    hq[jj].x = st_gelu(acc[4 * jj] + bv[jj].x); hq[jj].y = st_gelu(acc[4 * jj + 1] + bv[jj].y);
                    hq[jj].z = st_gelu(acc[4 * jj + 2] + bv[jj].z); hq[jj].w = st_gelu(acc[4 * jj + 3] + bv[jj].w);
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- stage B: the chunk is the k slice [32 hc, 32 hc + 32) of fc2; four independent accumulator tiles
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 gn[4];
                    const int jn = j + 1 < 4 ? j + 1 : j;
#pragma unroll
                    for (int oc = 0; oc < 4; ++oc) gn[oc] = *reinterpret_cast<const float4*>(w2s + oc * 1024 + goff[jn]);
#pragma unroll
                    for (int oc = 0; oc < 4; ++oc) o[oc] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[oc].x, hq[j].x, o[oc], 0, 0, 0);
#pragma unroll
                    for (int oc = 0; oc < 4; ++oc) o[oc] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[oc].y, hq[j].y, o[oc], 0, 0, 0);
#pragma unroll
                    for (int oc = 0; oc < 4; ++oc) o[oc] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[oc].z, hq[j].z, o[oc], 0, 0, 0);
#pragma unroll
                    for (int oc = 0; oc < 4; ++oc) o[oc] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[oc].w, hq[j].w, o[oc], 0, 0, 0);
#pragma unroll
                    for (int oc = 0; oc < 4; ++oc) g[oc] = gn[oc];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (rok) {
            // out = (fc2 + b2) + x [+ res]: the unfused epilogue's order (fma(acc, 1, bias), + aux0, + aux1).  The pointer is laundered
            // per round: b2's 16 loads are invariant across the rounds loop and hipcc otherwise hoists them to the top of the kernel,
            // where they hold 64 registers for its whole length (the projection variant then spilled 55)
            const float* b2p = d.b2;
            asm volatile("" : "+s"(b2p));
#pragma unroll
            for (int oc = 0; oc < 4; ++oc)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int col = oc * 32 + 8 * jj + 4 * lh;
                    const float4 bb = *reinterpret_cast<const float4*>(b2p + col);
                    const float4 x = *reinterpret_cast<const float4*>(xres + rowc * ld_xres + col);
                    float4 v = make_float4((o[oc][4 * jj] + bb.x) + x.x, (o[oc][4 * jj + 1] + bb.y) + x.y, (o[oc][4 * jj + 2] + bb.z) + x.z,
                                           (o[oc][4 * jj + 3] + bb.w) + x.w);
                    if (d.res) {
                        const float4 e = *reinterpret_cast<const float4*>(d.res + rowc * d.ld_res + col);
                        v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
                    }
                    *reinterpret_cast<float4*>(d.out + rowc * d.ldo + col) = v;
                }
        }
    }
}

extern "C" int st_abi_mlp_desc_size(void) { return (int)sizeof(st_mlp_desc); }

extern "C" int st_mlp128(const st_mlp_desc* desc, void* stream) {
    if (!desc) return ST_EINVAL;
    const st_mlp_desc& d = *desc;
    if (!d.a || !d.out || !d.w1 || !d.b1 || !d.w2 || !d.b2 || d.M <= 0 || d.hidden < 32 || d.hidden > 2048 || (d.hidden & 31) || d.lda < 128 ||
        d.ldo < 128 || (d.lda & 3) || (d.ldo & 3) || d.reserved != 0 || (int64_t)d.M * (d.lda > d.ldo ? d.lda : d.ldo) >= ((int64_t)1 << 40))
        return ST_EINVAL;
    if ((((uintptr_t)d.a | (uintptr_t)d.out | (uintptr_t)d.w1 | (uintptr_t)d.b1 | (uintptr_t)d.w2 | (uintptr_t)d.b2) & 15)) return ST_EINVAL;
    if (d.res && (d.ld_res < 128 || (d.ld_res & 3) || ((uintptr_t)d.res & 15))) return ST_EINVAL;
    if (d.a == d.out) return ST_EINVAL;                        // the residual x is re-read at the end of a block: not in place
    if (d.wp && (((uintptr_t)d.wp & 15) || (d.bp && ((uintptr_t)d.bp & 15)))) return ST_EINVAL;
    if (!d.wp && (d.bp || d.res0)) return ST_EINVAL;           // bias / residual of a projection that is not there
    if (d.wp && d.res == d.out) return ST_EINVAL;              // with a projection the block's rows of `out` hold the parked x until the end
    if (d.res0 && (d.ld_res0 < 128 || (d.ld_res0 & 3) || ((uintptr_t)d.res0 & 15) || d.res0 == d.out)) return ST_EINVAL;
    const int nblk = (d.M + 31) / 32;
    int G = (nblk + MLP_NW - 1) / MLP_NW;
    if (G > 512) G = 512;                                       // two workgroups per CU
    const size_t lds = (size_t)(2 * 2 * 32 * 128) * sizeof(float);
    auto kern = d.wp ? rowmlp128_kernel<true> : rowmlp128_kernel<false>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // the profiling observer sees the MLP as one launch of the family: M x (2 * hidden) x 128 = its FLOPs (2 M 128 hidden per product)
    st_gemm_observer_fn obs = g_observer;
    st_gemm_desc od;
    if (obs) {
        memset(&od, 0, sizeof(od));
        od.a = d.a; od.c = d.out; od.w = d.w1;
        od.M = d.M; od.N = 2 * d.hidden + (d.wp ? 128 : 0); od.K = 128; od.H = 1; od.W = d.M; od.Cin = 128; od.ldx = d.lda; od.ldc = d.ldo; od.ldw = 128;
        od.kh = od.kw = od.sh = od.sw = 1; od.Ho = 1; od.Wo = d.M; od.batch = 1; od.alpha = 1.f;
        obs(&od, stream, 0, g_observer_user);
    }
    g_last_plan[0] = 6; g_last_plan[1] = 31; g_last_plan[2] = 1; g_last_plan[3] = 1;
    hipLaunchKernelGGL(kern, dim3(G), dim3(64 * MLP_NW), lds, (hipStream_t)stream, d);
    if (obs) obs(&od, stream, 1, g_observer_user);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

#include "mlp_split3.h"

// bytes of the weight image of st_mlp128_split3 (one 49-KiB LDS stage image per step of the walk)
static int64_t mlp_split3_image_bytes(int32_t hidden, bool with_proj) {
    if (hidden < 32 || hidden > 2048 || (hidden & 31)) return 0;
    return (int64_t)((with_proj ? 4 : 0) + hidden / 32) * MS3_STAGE_B;
}
extern "C" int st_mlp128_split3_image_bytes(int32_t hidden, int32_t with_proj, int64_t* bytes) {
    if (!bytes) return ST_EINVAL;
    *bytes = mlp_split3_image_bytes(hidden, with_proj != 0);
    return *bytes ? ST_OK : ST_EINVAL;
}

extern "C" int st_mlp128_split3_pack(const float* w1, const float* b1, const float* w2, const float* wp, const float* bp, int32_t hidden, void* image,
                                     int64_t image_bytes, void* stream) {
    if (!w1 || !b1 || !w2 || !image || hidden < 32 || hidden > 2048 || (hidden & 31) || ((uintptr_t)image & 15) || (!wp && bp)) return ST_EINVAL;
    if (image_bytes < mlp_split3_image_bytes(hidden, wp != nullptr)) return ST_EINVAL;
    const int steps = (wp ? 4 : 0) + hidden / 32;
    hipLaunchKernelGGL(mlp_split3_pack_kernel, dim3(5, steps), dim3(256), 0, (hipStream_t)stream, w1, b1, w2, wp, bp, (int)hidden, (unsigned char*)image);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// st_mlp128 on the exact-split contraction (csrc/mlp_split3.h).  `desc` as for st_mlp128 -- w1 / b1 / w2 / bp are not read (the image holds them),
// wp != NULL says that the image was packed WITH the projection; b2 is read.
extern "C" int st_mlp128_split3(const st_mlp_desc* desc, const void* image, int64_t image_bytes, void* stream) {
    if (!desc || !image) return ST_EINVAL;
    const st_mlp_desc& d = *desc;
    if (!d.a || !d.out || !d.b2 || d.M <= 0 || d.hidden < 32 || d.hidden > 2048 || (d.hidden & 31) || d.lda < 128 || d.ldo < 128 || (d.lda & 3) ||
        (d.ldo & 3) || d.reserved != 0 || (int64_t)d.M * (d.lda > d.ldo ? d.lda : d.ldo) >= ((int64_t)1 << 40))
        return ST_EINVAL;
    if ((((uintptr_t)d.a | (uintptr_t)d.out | (uintptr_t)d.b2 | (uintptr_t)image) & 15)) return ST_EINVAL;
    if (d.res && (d.ld_res < 128 || (d.ld_res & 3) || ((uintptr_t)d.res & 15))) return ST_EINVAL;
    if (d.a == d.out) return ST_EINVAL;
    if (!d.wp && (d.bp || d.res0)) return ST_EINVAL;
    if (d.res0 && (d.ld_res0 < 128 || (d.ld_res0 & 3) || ((uintptr_t)d.res0 & 15) || d.res0 == d.out)) return ST_EINVAL;
    const int64_t need = mlp_split3_image_bytes(d.hidden, d.wp != nullptr);
    if (!need) return ST_EINVAL;
    if (image_bytes < need || need >= ((int64_t)1 << 31)) return ST_EINVAL;
    const int nblk = (d.M + 31) / 32;
    int G = (nblk + MS3_NWAVES - 1) / MS3_NWAVES;
    if (G > 256) G = 256;                                       // 147 KB of LDS: one workgroup per CU
    const size_t lds = (size_t)3 * MS3_STAGE_B;
    // timing experiment (tools/mlp_split3_probe.py --diag): ST_MLP3_DIAG=1 runs the instrumented instance, waits for it and prints per-phase cycle sums of wave 0
    static const int diag = [] { const char* e = getenv("ST_MLP3_DIAG"); return e ? atoi(e) : 0; }();
    if (diag) {
        static unsigned long long* dbuf = nullptr;
        if (!dbuf && hipHostMalloc((void**)&dbuf, 256 * 8 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) return ST_EINVAL;      // pinned: the kernel writes it, the host reads it after the sync
        auto kd = d.wp ? rowmlp128_split3_kernel<true, true> : rowmlp128_split3_kernel<false, true>;
        (void)hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kd, dim3(G), dim3(64 * MS3_NWAVES), lds, (hipStream_t)stream, d, (const unsigned char*)image, (unsigned)need, dbuf);
        ST_CHECK_LAUNCH();
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return ST_EINVAL;
        const unsigned long long* h = dbuf;
        double s8[8] = {0}, sum = 0;
        for (int b = 0; b < G; ++b) for (int i = 0; i < 8; ++i) s8[i] += (double)h[8 * b + i] / G;
        for (int i = 0; i < 8; ++i) sum += s8[i];
        fprintf(stderr, "mlp_split3 diag (mean s_memtime cycles of wave 0 over %d workgroups): syncs %.0f  load+split %.0f  projection %.0f  LN+split %.0f  first fc1 %.0f  phase1 %.0f  phase2 %.0f  last+epilogue %.0f  sum %.0f\n",
                G, s8[0], s8[1], s8[2], s8[3], s8[4], s8[5], s8[6], s8[7], sum);
        return ST_OK;
    }
    auto kern = d.wp ? rowmlp128_split3_kernel<true> : rowmlp128_split3_kernel<false>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    st_gemm_observer_fn obs = g_observer;
    st_gemm_desc od;
    if (obs) {
        memset(&od, 0, sizeof(od));
        od.a = d.a; od.c = d.out; od.w = (const float*)image;
        od.M = d.M; od.N = 2 * d.hidden + (d.wp ? 128 : 0); od.K = 128; od.H = 1; od.W = d.M; od.Cin = 128; od.ldx = d.lda; od.ldc = d.ldo; od.ldw = 128;
        od.kh = od.kw = od.sh = od.sw = 1; od.Ho = 1; od.Wo = d.M; od.batch = 1; od.alpha = 1.f; od.split3 = 1;
        obs(&od, stream, 0, g_observer_user);
    }
    g_last_plan[0] = 9; g_last_plan[1] = 38; g_last_plan[2] = 1; g_last_plan[3] = 1;
    hipLaunchKernelGGL(kern, dim3(G), dim3(64 * MS3_NWAVES), lds, (hipStream_t)stream, d, (const unsigned char*)image, (unsigned)need, (unsigned long long*)nullptr);
    if (obs) obs(&od, stream, 1, g_observer_user);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// LayerNorm -> Linear(128 -> N) + bias on the exact-split contraction (csrc/mlp_split3.h, rowlin128_split3_kernel): the weights packed once into
// N / 32 stage images of 25 KiB
extern "C" int st_rowlin128_split3_image_bytes(int32_t N, int64_t* bytes) {
    if (!bytes || N < 32 || N > 4096 || (N & 31)) return ST_EINVAL;
    *bytes = (int64_t)(N / 32) * LS3_STAGE_B;
    return ST_OK;
}
extern "C" int st_rowlin128_split3_pack(const float* w, const float* b, int32_t N, void* image, int64_t image_bytes, void* stream) {
    if (!w || !image || N < 32 || N > 4096 || (N & 31) || ((uintptr_t)image & 15) || image_bytes < (int64_t)(N / 32) * LS3_STAGE_B) return ST_EINVAL;
    hipLaunchKernelGGL(rowlin_split3_pack_kernel, dim3(3, N / 32), dim3(256), 0, (hipStream_t)stream, w, b, (unsigned char*)image);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
extern "C" int st_rowlin128_split3(const float* a, int32_t lda, float* out, int32_t ldo, int32_t M, int32_t N, int32_t ln, float ln_eps, const void* image,
                                   int64_t image_bytes, const float* aux, int32_t ld_aux, int32_t row_div, void* stream) {
    if (aux && (ld_aux < N || (ld_aux & 3) || row_div < 1 || ((uintptr_t)aux & 15) || aux == out)) return ST_EINVAL;
    if (!a || !out || !image || M <= 0 || N < 32 || N > 4096 || (N & 31) || lda < 128 || ldo < N || (lda & 3) || (ldo & 3) || a == out) return ST_EINVAL;
    if ((((uintptr_t)a | (uintptr_t)out | (uintptr_t)image) & 15) || (int64_t)M * (lda > ldo ? lda : ldo) >= ((int64_t)1 << 40)) return ST_EINVAL;
    const int64_t need = (int64_t)(N / 32) * LS3_STAGE_B;
    if (image_bytes < need) return ST_EINVAL;
    const int nblk = (M + 31) / 32;
    int G = (nblk + 3) / 4;
    if (G > 512) G = 512;                                       // 75 KB of LDS: two workgroups per CU
    const size_t lds = (size_t)3 * LS3_STAGE_B;
    auto kern = aux ? rowlin128_split3_kernel<true> : rowlin128_split3_kernel<false>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    st_gemm_observer_fn obs = g_observer;
    st_gemm_desc od;
    if (obs) {
        memset(&od, 0, sizeof(od));
        od.a = a; od.c = out; od.w = (const float*)image;
        od.M = M; od.N = N; od.K = 128; od.H = 1; od.W = M; od.Cin = 128; od.ldx = lda; od.ldc = ldo; od.ldw = 128;
        od.kh = od.kw = od.sh = od.sw = 1; od.Ho = 1; od.Wo = M; od.batch = 1; od.alpha = 1.f; od.split3 = 1;
        obs(&od, stream, 0, g_observer_user);
    }
    g_last_plan[0] = 10; g_last_plan[1] = 40; g_last_plan[2] = 1; g_last_plan[3] = 1;
    hipLaunchKernelGGL(kern, dim3(G), dim3(256), lds, (hipStream_t)stream, a, (int)lda, out, (int)ldo, (int)M, (int)N, (int)ln, ln_eps,
                       (const unsigned char*)image, (unsigned)need, aux, (int)ld_aux, (int)(row_div > 0 ? row_div : 1));
    if (obs) obs(&od, stream, 1, g_observer_user);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// PatchEmbed's tail (csrc/mlp_split3.h, pe_tail_split3_kernel): tokens[R, 128] = LayerNorm(ReLU(x[R, 64] . w1^T + tab[r % P]) . w2^T + b2) in one launch
extern "C" int st_pe_tail_split3_image_bytes(int64_t* bytes) {
    if (!bytes) return ST_EINVAL;
    *bytes = PT3_IMAGE_B;
    return ST_OK;
}
extern "C" int st_pe_tail_split3_pack(const float* w1, int32_t ld1, const float* w2, void* image, int64_t image_bytes, void* stream) {
    if (!w1 || !w2 || !image || ld1 < 64 || ((uintptr_t)image & 15) || image_bytes < PT3_IMAGE_B) return ST_EINVAL;
    hipLaunchKernelGGL(pe_tail_split3_pack_kernel, dim3(3, 4), dim3(256), 0, (hipStream_t)stream, w1, (int)ld1, w2, (unsigned char*)image);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
extern "C" int st_pe_tail_split3(const float* x, const float* tab, int32_t P, const void* image, int64_t image_bytes, const float* b2, const float* gamma,
                                 const float* beta, float eps, float* out, int32_t R, void* stream) {
    if (!x || !tab || !image || !b2 || !gamma || !beta || !out || R <= 0 || P <= 0 || image_bytes < PT3_IMAGE_B || x == out) return ST_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)tab | (uintptr_t)image | (uintptr_t)out) & 15) || (int64_t)R * 128 >= ((int64_t)1 << 40)) return ST_EINVAL;
    const int nblk = (R + 31) / 32;
    int G = (nblk + 3) / 4;
    if (G > 256) G = 256;                                       // 146 KB of LDS: one workgroup per CU
    // a grid whose row stride (G * 128) is a multiple of the table period keeps a wave's table rows the same for all its blocks: the largest such G <= 256
    bool tabinv = false;
    for (int g = G; g >= (G > 8 ? G - G / 8 : 1); --g)
        if (((long)g * 128) % P == 0) { G = g; tabinv = true; break; }
    const size_t lds = (size_t)PT3_IMAGE_B + PT3_VEC_B;
    auto kern = tabinv ? pe_tail_split3_kernel<true> : pe_tail_split3_kernel<false>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    st_gemm_observer_fn obs = g_observer;
    st_gemm_desc od;
    if (obs) {                                                  // reported as R x 192 x 128: its FLOPs (2 R (128 . 64 + 128 . 128))
        memset(&od, 0, sizeof(od));
        od.a = x; od.c = out; od.w = (const float*)image;
        od.M = R; od.N = 192; od.K = 128; od.H = 1; od.W = R; od.Cin = 128; od.ldx = 64; od.ldc = 128; od.ldw = 128;
        od.kh = od.kw = od.sh = od.sw = 1; od.Ho = 1; od.Wo = R; od.batch = 1; od.alpha = 1.f; od.split3 = 1;
        obs(&od, stream, 0, g_observer_user);
    }
    g_last_plan[0] = 11; g_last_plan[1] = 41; g_last_plan[2] = 1; g_last_plan[3] = 1;
    hipLaunchKernelGGL(kern, dim3(G), dim3(256), lds, (hipStream_t)stream, x, tab, (const unsigned char*)image, b2, gamma, beta, eps, out, (int)R, (int)P);
    if (obs) obs(&od, stream, 1, g_observer_user);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// split-K tail: sum the K-slice slabs [split][M][N] in slice order (deterministic) + epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const st_gemm_desc d) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)d.M * d.N) return;
    const int m = idx / d.N, n = idx % d.N;
    float acc = 0.f;
    for (int z = 0; z < d.split_k; ++z) acc += d.workspace[((size_t)z * d.M + m) * d.N + n];
    const float sc = d.scale_ptr ? *d.scale_ptr : 1.0f;
    gemm_store(d, d.c, m, n, acc, sc);
}

// Skinny GEMM (M <= 8 rows, e.g. the batch-1 regression head): weight-read bound, one wave per
// output column, K split across the lanes with 16-B loads, wave-shuffle reduction.
template <int MR>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const st_gemm_desc d) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= d.N) return;
    const float* w = d.w + (size_t)wave * d.ldw;
    float acc[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = 0.f;
    for (int k = lane * 4; k < d.K; k += 256) {
        const float4 wv = *reinterpret_cast<const float4*>(w + k);
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m < d.M) {
                const float4 av = *reinterpret_cast<const float4*>(d.a + (size_t)m * d.ldx + k);
                acc[m] = fmaf(av.x, wv.x, acc[m]); acc[m] = fmaf(av.y, wv.y, acc[m]);
                acc[m] = fmaf(av.z, wv.z, acc[m]); acc[m] = fmaf(av.w, wv.w, acc[m]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = wave_sum(acc[m]);
    if (lane == 0) {
        const float bv = d.bias ? d.bias[wave] : 0.f;
        for (int m = 0; m < d.M && m < MR; ++m) d.c[(size_t)m * d.ldc + wave] = st_act(acc[m] * d.alpha + bv, d.act);
    }
}

// Very narrow outputs (N <= 4: the flow head's 2-channel 3x3 conv, gru.py:5-13): an MFMA tile would be > 90 % padding.
// One wave per output pixel; lanes stride the channels of each tap with float4 loads, N running dot products,
// one wave reduction at the end.  Pure L2 bandwidth (every input row is re-read kh*kw times).
template <int NMAX>
__global__ __launch_bounds__(256) void narrow_conv_kernel(const st_gemm_desc d) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= d.M) return;
    const int hw = d.Ho * d.Wo;
    const int b = m / hw, rr = m - b * hw;
    const int oy = rr / d.Wo, ox = rr - oy * d.Wo;
    float acc[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) acc[n] = 0.f;
    for (int ky = 0; ky < d.kh; ++ky) {
        const int iy = oy * d.sh - d.ph + ky * (d.dh > 1 ? d.dh : 1);
        if (iy < 0 || iy >= d.H) continue;
        for (int kx = 0; kx < d.kw; ++kx) {
            const int ix = ox * d.sw - d.pw + kx * (d.dw > 1 ? d.dw : 1);
            if (ix < 0 || ix >= d.W) continue;
            const float* xr = d.a + ((size_t)(b * d.H + iy) * d.W + ix) * d.ldx;
            const float* wr = d.w + (size_t)(ky * d.kw + kx) * d.Cin;
            for (int c = lane * 4; c < d.Cin; c += 256) {
                const float4 xv = *reinterpret_cast<const float4*>(xr + c);
#pragma unroll
                for (int n = 0; n < NMAX; ++n) {
                    if (n < d.N) {
                        const float4 wv = *reinterpret_cast<const float4*>(wr + (size_t)n * d.ldw + c);
                        acc[n] = fmaf(xv.x, wv.x, acc[n]); acc[n] = fmaf(xv.y, wv.y, acc[n]);
                        acc[n] = fmaf(xv.z, wv.z, acc[n]); acc[n] = fmaf(xv.w, wv.w, acc[n]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NMAX; ++n) acc[n] = wave_sum(acc[n]);
    if (lane == 0) {
        const float sc = d.scale_ptr ? *d.scale_ptr : 1.0f;
        for (int n = 0; n < d.N && n < NMAX; ++n) d.c[(size_t)m * d.ldc + n] = gemm_epilogue(d, m, n, acc[n], sc);
    }
}

// The flow head's second conv (gru.py:5-13: Conv2d(256, 2, 3, padding=1)) and its like: 3x3, stride 1, pad 1, Cin = 256,
// N <= 2.  A lane owns 4 input channels and keeps its 9 x N weight quads in registers; a wave produces 8 consecutive
// output pixels of one image row: per kernel row it loads the 10 input pixels it needs once (whole 1-KiB channel
// rows), 9 x 8 x N dot-4 updates, then ONE transposing butterfly reduces the 8 x N partial sums across the wave
// (17 shuffles instead of 6 per value).  Input pixels are fetched 30 times per wave instead of 72, weights once
// instead of once per pixel (narrow_conv_kernel: 17 us at M = 8192; this: ~5 us).
template <int NOUT>
__global__ __launch_bounds__(256) void narrow_conv3x3_kernel(const st_gemm_desc d) {
    constexpr int NV = 8 * NOUT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int segs = (d.W + 7) >> 3;
    const int wid = blockIdx.x * 4 + wave;
    const int nimg = d.M / (d.H * d.W);
    if (wid >= nimg * d.H * segs) return;
    const int b = wid / (d.H * segs), r = wid - b * d.H * segs;
    const int y = r / segs, x0 = (r - y * segs) * 8;
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.w), 0, (int)d.w_bytes, 0x00020000);
    float4 wv[NOUT][9];
#pragma unroll
    for (int n = 0; n < NOUT; ++n)
#pragma unroll
        for (int t = 0; t < 9; ++t)
            wv[n][t] = buf_load16(rsrcW, n < d.N ? (unsigned)(n * d.ldw + t * 256 + 4 * lane) * 4u : ST_OOB);   // (a conditional global load made
    //                                                       hipcc wait for each of the 18 loads in turn: 18 dependent L2 round trips per wave)
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    // all 30 input rows are requested before the first is used (raw buffer loads: a tap outside the image gets an offset past
    // num_records and reads zeros, so there is no branch between the loads): one memory round trip per wave instead of three
    // dependent ones -- with one wave per SIMD nothing else hides them
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.a), 0, (int)d.a_bytes, 0x00020000);
    float4 in[3][10];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = y + ky - 1;
        const bool yok = iy >= 0 && iy < d.H;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int ix = x0 - 1 + i;
            const unsigned off = (yok && ix >= 0 && ix < d.W) ? (unsigned)(((b * d.H + iy) * d.W + ix) * d.ldx + 4 * lane) * 4u : ST_OOB;
            in[ky][i] = buf_load16(rsrcA, off);
        }
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int n = 0; n < NOUT; ++n) {
                    const float4 xv = in[ky][p + kx], w4 = wv[n][ky * 3 + kx];
                    float a = acc[p * NOUT + n];
                    a = fmaf(xv.x, w4.x, a); a = fmaf(xv.y, w4.y, a); a = fmaf(xv.z, w4.z, a); a = fmaf(xv.w, w4.w, a);
                    acc[p * NOUT + n] = a;
                }
    // transposing butterfly: each step halves the number of values a lane carries; after log2(NV) steps lane l holds the
    // 64 / NV-lane partial sum of value v(l), v's bits taken from the lane bits used so far (high bits first)
    int cnt = NV / 2;
#pragma unroll
    for (int s = 32; cnt >= 1; s >>= 1, cnt >>= 1) {
        const bool hi = lane & s;
#pragma unroll
        for (int i = 0; i < NV / 2; ++i)
            if (i < cnt) {
                const float send = hi ? acc[i] : acc[i + cnt], keep = hi ? acc[i + cnt] : acc[i];
                acc[i] = keep + __shfl_xor(send, s, 64);
            }
    }
    constexpr int STEPS = NOUT == 1 ? 3 : NOUT == 2 ? 4 : 5;          // log2(NV)
    float v = acc[0];
#pragma unroll
    for (int s = 32 >> STEPS; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
    if ((lane & ((64 >> STEPS) - 1)) == 0) {
        int idx = 0, c = NV / 2;
#pragma unroll
        for (int s = 32; c >= 1; s >>= 1, c >>= 1) idx += (lane & s) ? c : 0;
        const int p = idx / NOUT, n = idx - p * NOUT;
        const int x = x0 + p;
        if (x < d.W && n < d.N) {
            const int m = (b * d.H + y) * d.W + x;
            const float sc = d.scale_ptr ? *d.scale_ptr : 1.0f;
            d.c[(size_t)m * d.ldc + n] = gemm_epilogue(d, m, n, v, sc);
        }
    }
}


template <int WARPS_M, int WARPS_N, int TM, int TN>
static int launch_cfg(const st_gemm_desc& d, bool vec, hipStream_t s) {
    constexpr int BM = WARPS_M * TM * 32, BN = WARPS_N * TN * 32;
    const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
    dim3 grid(ntm * ntn, 1, d.split_k > 1 ? d.split_k : (d.batch > 0 ? d.batch : 1));
    const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
    if (vec) {
        auto k = conv_gemm_kernel<WARPS_M, WARPS_N, TM, TN, true>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, grid, dim3(256), lds, s, d);
    } else {
        auto k = conv_gemm_kernel<WARPS_M, WARPS_N, TM, TN, false>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, grid, dim3(256), lds, s, d);
    }
    if (d.split_k > 1)
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(((size_t)d.M * d.N + 255) / 256), dim3(256), 0, s, d);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

template <int WM, int WN, int TM, int TN, int STAGES>
static int launch_dma(const st_gemm_desc& d, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
    const int batch = d.batch > 0 ? d.batch : 1;
    const size_t lds = (size_t)STAGES * (BM + BN) * 32 * sizeof(float);
    // persistent walk over M tiles: plain matrices whose K is a whole number of ring turns and whose grid would
    // otherwise be many short workgroups (two resident per CU with this LDS footprint)
    const bool plain = d.kh == 1 && d.kw == 1 && d.sh == 1 && d.sw == 1 && d.ph == 0 && d.pw == 0 && d.H * d.W == d.M &&
                       d.Ho * d.Wo == d.M;
    // Round-5 experiment, measured and NOT adopted (both switches default to the round-4 behaviour): ST_PERSIST_SLOTS=256 gives a launch ONE
    // workgroup slot per CU (a CU holds two of this LDS footprint; a workgroup then walks >= 2 M tiles with one continuous DMA ring) so that the
    // second slot is left to the kernel of another stream (ST_PERSIST_CONV=1 also let convolutions walk that way; removed in round 6).  On uniform GEMMs the idea pays (tools/persist_probe.py, 5 decoder shapes, one hipGraph
    // per stream: 3 streams 120.9 -> 127.8 TFLOP/s, 2 streams 119.2 -> 124.1, 1 stream 104.2 -> 102.4); in the product it does not (bench.py,
    // 3 pairs in flight, same box, A/B/A: slots 512 conv 0: 82.66 / 82.58 pairs/s; 256 / 1: 81.09 / 80.67; 256 / 0: 82.60; 512 / 1: 82.22;
    // one pair in flight 72.4 -> 69.0): between the GEMMs of a forward run ~150 short kernels of other kinds, and a GEMM that holds one slot
    // runs at one wave per SIMD whenever the neighbouring stream is not in a GEMM itself.
    static const int slots_env = [] { const char* e = getenv("ST_PERSIST_SLOTS"); return e ? atoi(e) : 512; }();
    const int slots = slots_env / batch;
    if (STAGES == 4 && plain && !d.a2 && d.split_k <= 1 && (d.K / 32) % STAGES == 0 && (long)ntm * ntn > slots && ntn <= slots && slots > 0) {
        int G = slots / ntn;
        if (G > ntm) G = ntm;
        auto k = d.c_t ? conv_gemm_dma_kernel<WM, WN, TM, TN, STAGES, true, true> : conv_gemm_dma_kernel<WM, WN, TM, TN, STAGES, true>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        g_last_plan[3] = 1;
        hipLaunchKernelGGL(k, dim3(G * ntn, 1, batch), dim3(256), lds, s, d);
        ST_CHECK_LAUNCH();
        return ST_OK;
    }
    dim3 grid(ntm * ntn, 1, d.split_k > 1 ? d.split_k : batch);
    auto k = d.c_t ? conv_gemm_dma_kernel<WM, WN, TM, TN, STAGES, false, true> : conv_gemm_dma_kernel<WM, WN, TM, TN, STAGES, false>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, d);
    if (d.split_k > 1)
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(((size_t)d.M * d.N + 255) / 256), dim3(256), 0, s, d);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// row-streaming kernel: S column slices of `cps` 32-column chunks, G workgroups (512 threads, one per CU) per slice.
// S is the smallest power of two that (a) keeps a slice within `max_cps` chunks and (b) yields at least one (row block,
// slice) item per wave of the chip; G * S = 256, G a multiple of 8 (one run of workgroups per XCD).
template <int KC>
static int launch_rowstream(const st_gemm_desc& d, int max_cps, hipStream_t s) {
    const int nchunks = (d.N + 31) / 32;
    const int nblk = (d.M + 31) / 32;
    const int lds_cap = (160 * 1024 - 1024) / (32 * (32 * KC + 4) * 4);
    if (max_cps > lds_cap) max_cps = lds_cap;
    int S = 1;
    while (S < 32 && ((nchunks + S - 1) / S > max_cps || ((long)nblk * S < 2048 && 2 * S <= nchunks))) S *= 2;
    const int cps = (nchunks + S - 1) / S;
    if (cps > lds_cap) return ST_EINVAL;
    S = (nchunks + cps - 1) / cps;                 // drop empty slices ...
    int G = (256 / S) / 8 * 8;                      // ... (G * S <= 256)
    if (G < 8) G = 8;
    const int need = ((nblk + 7) / 8 + 7) / 8 * 8;
    if (G > need) G = need;
    const size_t lds = ((size_t)cps * 32 * (32 * KC + 4) + cps * 32) * sizeof(float);
    auto k = rowstream_gemm_kernel<KC, true>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(G * S), dim3(512), lds, s, d, cps, S);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// st_gemm_desc.c_planes: what every kernel that can emit planes needs checked
static bool c_planes_ok(const st_gemm_desc& d) {
    if (d.reserved4 != 0) return false;
    if (!d.c_planes) return d.c_no_f32 == 0;
    const int batch = d.batch > 0 ? d.batch : 1;
    const int ncols = d.epi == ST_EPI_ZR ? d.N / 2 : d.N;
    if ((d.M & 31) || (ncols & 1) || (d.c_plane_col0 & 31) || d.c_plane_col0 < 0 || d.c_plane_row0 < 0 || d.c_plane_stride <= 0 || (d.c_plane_stride & 7) || d.c_plane_rows <= 0 ||
        ((uintptr_t)d.c_planes & 15) || d.c_t)
        return false;
    const int64_t last_row = d.c_plane_row0 + (int64_t)(batch - 1) * d.c_plane_batch_rows + d.M;
    if (last_row > d.c_plane_rows) return false;
    const int64_t chunks = (d.c_plane_col0 + ncols + 31) / 32;
    return 2 * (2 * d.c_plane_stride + chunks * d.c_plane_rows * 32) < ((int64_t)1 << 31);
}

static int conv_gemm_split3_launch(const st_gemm_desc* desc, void* stream);
static int conv_gemm_launch(const st_gemm_desc* desc, void* stream) {
    if (!c_planes_ok(*desc)) return ST_EINVAL;
    if (desc->split3) return conv_gemm_split3_launch(desc, stream);
    st_gemm_desc d = *desc;
    if (!d.a || !d.w || !d.c || d.M <= 0 || d.N <= 0 || d.K <= 0) return ST_EINVAL;
    if (d.kh <= 0 || d.kw <= 0 || d.K != d.kh * d.kw * d.Cin) return ST_EINVAL;
    if (d.Ho <= 0 || d.Wo <= 0 || d.M % (d.Ho * d.Wo)) return ST_EINVAL;
    if (d.epi != ST_EPI_STORE && !d.aux1) return ST_EINVAL;
    if (d.epi == ST_EPI_GRU && !d.aux2) return ST_EINVAL;
    if (d.epi == ST_EPI_ZR && (!d.c2 || (d.N & 1))) return ST_EINVAL;
    if (d.reserved0 != 0 || d.reserved1 != 0 || d.reserved2 != 0 || d.reserved3 != 0) return ST_EINVAL;
    if (d.c_t && (d.epi != ST_EPI_STORE || (d.M & 3) || (d.ld_ct & 3) || d.ld_ct < d.M || ((uintptr_t)d.c_t & 15) || d.split_k > 1 || d.a_ln ||
                  (int64_t)d.N * d.ld_ct * 4 >= ((int64_t)1 << 31)))
        return ST_EINVAL;
    if (d.a2 && (d.a2_channels <= 0 || d.a2_channels % 32 || d.a2_channels > d.Cin || d.batch > 1 || ((uintptr_t)d.a2 & 15))) return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // byte extents of one batch slice of A and W for the buffer descriptors (must stay below 2 GiB so the
    // out-of-range sentinel offset is always past num_records)
    {
        const bool plain_mat = d.kh == 1 && d.kw == 1 && d.sh == 1 && d.sw == 1 && d.ph == 0 && d.pw == 0 && (int64_t)d.H * d.W == d.M &&
                               (int64_t)d.Ho * d.Wo == d.M;          // a matrix: every row is its own "image"
        const int64_t hw = plain_mat ? 1 : (int64_t)d.Ho * d.Wo, nimg = d.M / hw, img_rows = plain_mat ? 1 : (int64_t)d.H * d.W;
        const int64_t ab = ((nimg * img_rows - 1) * d.ldx + d.Cin) * 4, wb = ((int64_t)(d.N - 1) * d.ldw + d.K) * 4;
        if (wb >= (int64_t)ST_OOB) return ST_EINVAL;
        // the epilogue addresses C / aux operands with 32-bit buffer offsets (first row of a lane + SGPR row step)
        const int64_t lim = (int64_t)1 << 31;
        int64_t ldmax = d.ldc > d.N ? d.ldc : d.N;
        if (d.aux0 && d.ld_aux0 > ldmax) ldmax = d.ld_aux0;
        if (d.aux1 && d.ld_aux1 > ldmax) ldmax = d.ld_aux1;
        if (d.aux2 && d.ld_aux2 > ldmax) ldmax = d.ld_aux2;
        if (d.c2 && d.ldc2 > ldmax) ldmax = d.ldc2;
        if (ab >= (int64_t)ST_OOB || ((int64_t)d.M + 256) * ldmax * 4 >= lim) {
            // Larger than the 32-bit offsets reach (whole-batch PatchEmbed maps at B >= 4, ...): run the rows in chunks.
            // A chunk is a whole number of images (convs) and of aux0 mapping periods, so every operand just shifts its base.
            if ((d.batch > 1) || d.split_k > 1 || d.c_t || d.c_planes) return ST_EINVAL;
            const int64_t div = d.aux0_row_div > 1 ? d.aux0_row_div : 1, mod = d.aux0_row_mod > 0 ? d.aux0_row_mod : 1;
            int64_t unit = hw;                                           // rows per indivisible unit
            {
                const int64_t period = div * mod;                       // lcm(unit, period) via gcd
                int64_t a = unit, b = period;
                while (b) { const int64_t t = a % b; a = b; b = t; }
                unit = unit / a * period;
            }
            const int64_t in_rows_per_unit = unit / hw * img_rows;
            int64_t per_out = (lim / 4 / ldmax - 256) / unit, per_in = ((int64_t)ST_OOB / 4 / d.ldx) / in_rows_per_unit;
            int64_t units = per_out < per_in ? per_out : per_in;
            // strictly inside both limits (the same tests as above, on the chunk)
            while (units > 0 && (((units * in_rows_per_unit - 1) * d.ldx + d.Cin) * 4 >= (int64_t)ST_OOB ||
                                 (units * unit + 256) * ldmax * 4 >= lim))
                --units;
            if (units < 1 || d.M % unit || units * unit >= d.M) return ST_EINVAL;      // (a chunk must be a strict part)
            {
                const int64_t total = d.M / unit, nchunks = (total + units - 1) / units;  // equal-sized chunks
                units = (total + nchunks - 1) / nchunks;
            }
            const int64_t chunk = units * unit;
            for (int64_t m0 = 0; m0 < d.M; m0 += chunk) {
                st_gemm_desc c = *desc;
                c.M = (int32_t)((d.M - m0) < chunk ? (d.M - m0) : chunk);
                c.a = d.a + m0 / hw * img_rows * d.ldx;
                if (d.a2) c.a2 = d.a2 + m0 / hw * img_rows * d.ldx;
                c.c = d.c + m0 * d.ldc;
                if (d.c2) c.c2 = d.c2 + m0 * d.ldc2;
                if (d.aux1) c.aux1 = d.aux1 + m0 * d.ld_aux1;
                if (d.aux2) c.aux2 = d.aux2 + m0 * d.ld_aux2;
                if (d.aux0 && d.aux0_row_mod <= 0) c.aux0 = d.aux0 + m0 / div * d.ld_aux0;   // with a modulus the chunk starts a period
                if (plain_mat) { c.H = 1; c.W = c.M; c.Ho = 1; c.Wo = c.M; }
                const int rc = conv_gemm_launch(&c, stream);
                if (rc) return rc;
            }
            return ST_OK;
        }
        d.a_bytes = (uint32_t)ab; d.w_bytes = (uint32_t)wb;
    }
    const bool aligned = ((uintptr_t)d.a % 16 == 0) && ((uintptr_t)d.w % 16 == 0) && (d.ldx % 4 == 0) &&
                         (d.ldw % 4 == 0) && (d.Cin % 4 == 0) &&
                         (d.batch_stride_a % 4 == 0) && (d.batch_stride_w % 4 == 0);
    const int batch = d.batch > 0 ? d.batch : 1;
    if (d.M <= 8 && d.kh == 1 && d.kw == 1 && aligned && batch == 1 && d.epi == ST_EPI_STORE && !d.aux0 && d.H * d.W == d.M && !d.a2 && !d.c_t && !d.c_planes) {
        g_last_plan[0] = 0; g_last_plan[1] = 0; g_last_plan[2] = 1; g_last_plan[3] = 0;
        hipLaunchKernelGGL(skinny_gemm_kernel<8>, dim3((d.N * 64 + 255) / 256), dim3(256), 0, s, d);
        ST_CHECK_LAUNCH();
        return ST_OK;
    }
    if (d.N <= 4 && aligned && batch == 1 && d.epi != ST_EPI_ZR && d.M >= 1024 && d.tile_cfg == 0 && d.split_k <= 1 && !d.a2 && !d.c_t && !d.c_planes) {
        g_last_plan[0] = 1; g_last_plan[1] = 0; g_last_plan[2] = 1; g_last_plan[3] = 0;
        if (d.N <= 2 && d.Cin == 256 && d.kh == 3 && d.kw == 3 && d.sh == 1 && d.sw == 1 && d.ph == 1 && d.pw == 1 && d.dh <= 1 && d.dw <= 1 &&
            d.Ho == d.H && d.Wo == d.W) {
            const int waves = (d.M / (d.H * d.W)) * d.H * ((d.W + 7) / 8);
            hipLaunchKernelGGL(narrow_conv3x3_kernel<2>, dim3((waves + 3) / 4), dim3(256), 0, s, d);
            ST_CHECK_LAUNCH();
            return ST_OK;
        }
        hipLaunchKernelGGL(narrow_conv_kernel<4>, dim3((d.M + 3) / 4), dim3(256), 0, s, d);
        ST_CHECK_LAUNCH();
        return ST_OK;
    }
    {
        // row-streaming kernel (weights resident in LDS, A rows in registers): plain matrices with K = 64 / 128 and enough
        // 32-row blocks to give every wave of the chip at least one
        const bool plain = d.kh == 1 && d.kw == 1 && d.sh == 1 && d.sw == 1 && d.ph == 0 && d.pw == 0 && (int64_t)d.H * d.W == d.M;
        const bool map_ok = (d.aux0_row_div <= 1 && (d.aux0_row_mod <= 0 || d.aux0_row_mod % 32 == 0)) ||
                            (d.aux0_row_div == 8 && d.aux0_row_mod <= 0) || !d.aux0;
        const bool rs_ok = plain && aligned && batch == 1 && !d.a2 && !d.c_t && !d.c_planes && d.split_k <= 1 && (d.K == 64 || d.K == 128) &&
                           d.epi != ST_EPI_ZR && d.epi != ST_EPI_GRU && map_ok;
        // measured on MI355X, in the pipeline and stand-alone (tools/rowstream_bench.py): ahead of the LDS-DMA kernel for K = 64,
        // for M >= 262144 and for N >= 384; behind it by ~2 us per launch at M <= 65536, N = 128 (one block per wave: all
        // start-up); K = 256 (no room for the prefetch) was 10-30 % slower everywhere and is not instantiated
        const bool rs_want = d.tile_cfg == 20 || (d.tile_cfg == 0 && d.M >= 16384 &&
                                                  (d.K == 64 || d.M >= 262144 || (d.N >= 384 && d.M >= 32768)));
        if (d.a_ln && !(rs_ok && (d.tile_cfg == 0 || d.tile_cfg == 20))) return ST_EINVAL;
        if (rs_ok && (rs_want || d.a_ln)) {
            g_last_plan[0] = 4; g_last_plan[1] = 20; g_last_plan[2] = 1; g_last_plan[3] = 1;
            return d.K == 64 ? launch_rowstream<2>(d, 8, s) : launch_rowstream<4>(d, 8, s);      // 8-chunk slices (4-chunk ones, which would let a
            //                                                       64-KiB GEMM workgroup co-reside, measured 0.3 % slower with 3 pairs in flight)
        }
        if (d.tile_cfg >= 20) return ST_EINVAL;
    }
    // tile choice: largest tile that still yields >= ~1.5 waves of workgroups over the 256 CUs
    auto nwg = [&](int bm, int bn) { return (long)((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn) * batch; };
    int cfg = d.tile_cfg;
    // the LDS-DMA pipelined kernel needs whole 32-channel K steps inside one tap
    const bool dma_ok = aligned && d.Cin % 32 == 0;
    if (cfg == 0) {
        // measured on MI355X (tools/tile_sweep.py, tools/dma_sweep.py): the 64x64 tile wins for every short-K /
        // mid-size shape of this path (more resident workgroups); 128-wide tiles stay selectable through tile_cfg.
        const bool plain = d.kh == 1 && d.kw == 1 && d.sh == 1 && d.sw == 1 && d.ph == 0 && d.pw == 0 && d.H * d.W == d.M;
        // K >= 128 (four K steps): the DMA ring already beats the register-staged kernel by 25-35 % per launch
        // (graph replay, tools/shortk_bench.py: 8192x256x160 14.4 -> 10.6 us, 8192x128x192 10.9 -> 7.6, 4096x128x128 7.8 -> 5.9)
        const bool dma = dma_ok && d.K >= 128;
        if (d.N <= 32) cfg = dma ? 14 : 4;
        else cfg = dma ? 13 : 3;
        // (64x128, tile_cfg 15, one workgroup per CU with two accumulator tiles per wave, stays selectable but is not chosen: round 4
        // measured it equal on the real z|r launches (67.0 / 65.3 vs 66.6 / 64.7 us, tools/tile15_bench.py, bit-identical) and 0.8 % slower
        // in the pipeline, 82.7 -> 82.0 pairs/s; it wins 2-5 % only on plain N = 256 long-K convs, which the path does not have)
    }
    if (cfg > 10 && !dma_ok) return ST_EINVAL;
    if ((d.a2 || d.c_t) && cfg <= 10) return ST_EINVAL;        // second A source / transposed copy: LDS-DMA kernels only
    // split-K: a launch that cannot fill the 256 CUs (M = 4096-pixel maps x 64..256 channels) is cut
    // along K into slabs reduced by a second tiny kernel (deterministic order; no atomics).
    static const int bms[6] = {0, 128, 128, 64, 128, 64}, bns[6] = {0, 128, 64, 64, 32, 128};
    const long tiles = cfg > 10 ? nwg(bms[cfg - 10], bns[cfg - 10]) : nwg(bms[cfg], bns[cfg]);
    int split = d.split_k;
    if (split == 0) {
        split = 1;
        if (batch == 1 && d.workspace && d.K >= 512 && !d.c_t) {
            if (cfg > 10) {
                // the pipelined kernel runs near its steady-state rate with ONE workgroup per CU, so it only needs
                // every CU covered; 257..511 tiles leave half the chip with twice the work of the other half
                if (tiles < 256) split = (int)((256 + tiles - 1) / tiles);
                else if (tiles < 512 && tiles % 256 && d.K >= 1024) split = 2;
            } else if (tiles < 256) {
                split = (int)((512 + tiles - 1) / tiles);
            }
            if (split > d.K / 256) split = d.K / 256;
            if (split > 16) split = 16;
            if (split < 1) split = 1;
            while (split > 1 && (int64_t)split * d.M * d.N > d.workspace_floats) --split;
        }
    }
    if (split > 1 && (batch != 1 || !d.workspace || (int64_t)split * d.M * d.N > d.workspace_floats)) return ST_EINVAL;
    d.split_k = split;
    g_last_plan[0] = cfg > 10 ? 3 : 2; g_last_plan[1] = cfg; g_last_plan[2] = split; g_last_plan[3] = 0;
    if (cfg == 12) return launch_dma<2, 2, 2, 1, 4>(d, s);
    // (a 3-deep ring -- 48 KB, three workgroups per CU -- is correct with this body (the K-block fold is independent of the ring depth) and
    // was measured: kernels of different pairs overlap better (3-in-flight / 1-in-flight 1.18 instead of 1.135) but a tile then has ONE K
    // step to land and every kernel slows down: 71.9 -> 67.9 pairs/s with one pair in flight, 81.6 -> 80.2 with three; a 5-deep ring
    // -- 80 KB, four tiles ahead -- for the convs and K >= 1024, round 4: 82.65 -> 81.5 / 72.5 -> 71.6: two workgroups no longer share a CU)
    if (cfg == 13) return launch_dma<2, 2, 1, 1, 4>(d, s);
    if (cfg == 15) return launch_dma<2, 2, 1, 2, 4>(d, s);      // 64x128: selectable only (see the tile choice above)
    if (cfg == 14) return launch_dma<4, 1, 1, 1, 4>(d, s);      // (80 KB of LDS; a 3-deep ring, 60 KB, measured neutral: PatchEmbed's 6x3 conv is not occupancy-bound)
    switch (cfg) {
        case 1: return launch_cfg<2, 2, 2, 2>(d, aligned, s);
        case 2: return launch_cfg<2, 2, 2, 1>(d, aligned, s);
        case 3: return launch_cfg<2, 2, 1, 1>(d, aligned, s);
        case 4: return launch_cfg<4, 1, 1, 1>(d, aligned, s);
        default: return ST_EINVAL;
    }
}

// ---- split3 (csrc/gemm_split3.h): a / w are three blocked bf16 planes -------------------------------------------------------
template <int WM, int WN, int TM, int TN, int STAGES, int DIAG = 0>
static int launch_split3(const st_gemm_desc& d, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
    const int batch = d.batch > 0 ? d.batch : 1;
    const size_t lds = (size_t)STAGES * 3 * (BM + BN) * 64;
    void (*k)(const st_gemm_desc) = conv_gemm_split3_kernel<WM, WN, TM, TN, STAGES, DIAG>;
    if constexpr (TM * TN == 1 && WM == 2 && WN == 2 && STAGES == 3) k = conv_gemm_split3_kernel64<STAGES, DIAG>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(ntm * ntn, 1, d.split_k > 1 ? d.split_k : batch), dim3(512), lds, s, d);
    if (d.split_k > 1)
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(((size_t)d.M * d.N + 255) / 256), dim3(256), 0, s, d);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// two consumer groups per workgroup (KPAR = 2, csrc/gemm_split3.h): 768 threads, same tiles, same ring
template <int WM, int WN, int TM, int TN, int STAGES>
static int launch_split3_kpar(const st_gemm_desc& d, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
    const int batch = d.batch > 0 ? d.batch : 1;
    const size_t lds = (size_t)STAGES * 3 * (BM + BN) * 64;
    void (*k)(const st_gemm_desc) = conv_gemm_split3_kpar_kernel<WM, WN, TM, TN, STAGES>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(ntm * ntn, 1, batch), dim3(768), lds, s, d);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// checks + buffer extents of a split3 descriptor (shared by st_conv_gemm and st_conv_gemm_pair)
static int split3_prepare(const st_gemm_desc* desc, st_gemm_desc& d) {
    d = *desc;
    if (!d.a || !d.w || !d.c || d.M <= 0 || d.N <= 0 || d.K <= 0) return ST_EINVAL;
    if (d.kh <= 0 || d.kw <= 0 || d.K != d.kh * d.kw * d.Cin || d.Cin % 32) return ST_EINVAL;
    if (d.Ho <= 0 || d.Wo <= 0 || d.M % (d.Ho * d.Wo)) return ST_EINVAL;
    if (d.epi != ST_EPI_STORE && !d.aux1) return ST_EINVAL;
    if (d.epi == ST_EPI_GRU && !d.aux2) return ST_EINVAL;
    if (d.epi == ST_EPI_ZR && (!d.c2 || (d.N & 1))) return ST_EINVAL;
    if (d.reserved0 != 0 || d.reserved1 != 0 || d.reserved2 != 0 || d.reserved3 != 0 || d.a_ln) return ST_EINVAL;
    if (d.c_t && (d.epi != ST_EPI_STORE || (d.M & 3) || (d.ld_ct & 3) || d.ld_ct < d.M || ((uintptr_t)d.c_t & 15) || d.split_k > 1 || d.a2 || d.c_planes ||
                  (int64_t)d.N * d.ld_ct * 4 >= ((int64_t)1 << 31)))
        return ST_EINVAL;                              // (transposed second store: the persistent 64x64 kernel only)
    if (!c_planes_ok(d)) return ST_EINVAL;
    if (d.split3 != 1) return ST_EINVAL;
    if (d.a_plane_stride <= 0 || d.w_plane_stride <= 0 || d.a_rows <= 0 || d.w_rows < d.N) return ST_EINVAL;
    if (((uintptr_t)d.a & 15) || ((uintptr_t)d.w & 15) || (d.a_plane_stride & 7) || (d.w_plane_stride & 7) || (d.batch_stride_a & 7) ||
        (d.batch_stride_w & 7))
        return ST_EINVAL;
    if (d.a2 && (d.a2_channels <= 0 || d.a2_channels % 32 || d.a2_channels > d.Cin || d.batch > 1 || ((uintptr_t)d.a2 & 15))) return ST_EINVAL;
    const bool plain_mat = d.kh == 1 && d.kw == 1 && d.sh == 1 && d.sw == 1 && d.ph == 0 && d.pw == 0 && (int64_t)d.H * d.W == d.M &&
                           (int64_t)d.Ho * d.Wo == d.M;
    const int64_t nimg = d.M / ((int64_t)d.Ho * d.Wo), in_rows = plain_mat ? d.M : nimg * d.H * d.W;
    if (in_rows > d.a_rows) return ST_EINVAL;
    // extents (bytes) from the plane-0 base to the end of plane 2; the 32-bit buffer offsets and the out-of-range sentinel need < 2 GiB
    const int64_t ab = 2 * (2 * d.a_plane_stride + (int64_t)(d.Cin / 32) * d.a_rows * 32);
    const int64_t wb = 2 * (2 * d.w_plane_stride + (int64_t)(d.K / 32) * d.w_rows * 32);
    if (ab >= (int64_t)ST_OOB || wb >= (int64_t)ST_OOB) return ST_EINVAL;
    int64_t ldmax = d.ldc > d.N ? d.ldc : d.N;
    if (d.aux0 && d.ld_aux0 > ldmax) ldmax = d.ld_aux0;
    if (d.aux1 && d.ld_aux1 > ldmax) ldmax = d.ld_aux1;
    if (d.aux2 && d.ld_aux2 > ldmax) ldmax = d.ld_aux2;
    if (d.c2 && d.ldc2 > ldmax) ldmax = d.ldc2;
    if (((int64_t)d.M + 256) * ldmax * 4 >= ((int64_t)1 << 31)) return ST_EINVAL;
    d.a_bytes = (uint32_t)ab; d.w_bytes = (uint32_t)wb;
    return ST_OK;
}

static int conv_gemm_split3_launch(const st_gemm_desc* desc, void* stream) {
    st_gemm_desc d;
    {
        const int rc = split3_prepare(desc, d);
        if (rc) return rc;
    }
    const int batch = d.batch > 0 ? d.batch : 1;
    int cfg = d.tile_cfg;
    // measured (tools/split3_probe.py, profiles/r6_split3_probe.json): 128x64 tiles on a 4-stage ring win when they still give every CU a
    // workgroup (N = 256 at M = 8 192: 41.8 vs 44.2 us), 64x64 tiles (two workgroups per CU) otherwise; 128x128 never
    if (cfg == 0) cfg = (long)((d.M + 127) / 128) * ((d.N + 63) / 64) * batch >= 256 ? 32 : 34;
    // many short tiles (>= 4 per workgroup slot, K <= 2 048): the persistent 64x64 walk (tile_cfg 37) -- the all-pairs volume, PatchEmbed's third conv
    const long ntl64 = (long)((d.M + 63) / 64) * ((d.N + 63) / 64);
    static const int persist_env = [] { const char* e = getenv("ST_SPLIT3_PERSIST"); return e ? atoi(e) : 1; }();      // ST_SPLIT3_PERSIST=0: A/B switch
    if (persist_env && d.tile_cfg == 0 && !d.a2 && d.split_k <= 1 && d.K <= 2048 && ntl64 * batch >= 2048) cfg = 37;
    if (d.c_t) cfg = 37;
    static const int tile_env = [] { const char* e = getenv("ST_SPLIT3_TILE"); return e ? atoi(e) : 0; }();      // experiments: force one tile configuration
    if (tile_env && d.tile_cfg == 0 && !d.c_t) cfg = tile_env;
    // two consumer groups per workgroup (csrc/gemm_split3.h KPAR; tile_cfg 39: 64x64 tiles, 38: 128x64) for a launch of exactly one 64x64 tile per CU and
    // a long K -- the N = 128 shapes at M = 8 192 (SepConvGRU's q convolutions, the motion encoder's 126-channel conv).  Measured (tools/split3_probe.py,
    // 8192 x 128 x 1920): 23.9 us against 29.1 for the four-consumer tile and 33.9 + a 7.4 us reducer launch for the split-K-2 form it replaces.  The
    // 128x64 form (bit 1 of ST_SPLIT3_KPAR) is SLOWER than its four-consumer twin on the N = 256 shapes (47.7 against 42.7 us): three waves per SIMD and
    // twelve waves per barrier cost more than the second MFMA issuer returns there.  ST_SPLIT3_KPAR=0: off.
    static const int kpar_env = [] { const char* e = getenv("ST_SPLIT3_KPAR"); return e ? atoi(e) : 2; }();
    if (kpar_env && d.tile_cfg == 0 && !tile_env && batch == 1 && d.split_k <= 1 && d.K >= 1024) {
        if ((kpar_env & 1) && cfg == 32 && (long)((d.M + 127) / 128) * ((d.N + 63) / 64) <= 512) cfg = 38;
        else if ((kpar_env & 2) && cfg == 34 && ntl64 == 256) cfg = 39;
    }
    static const int bms[10] = {0, 128, 128, 64, 64, 128, 64, 64, 128, 64}, bns[10] = {0, 128, 64, 128, 64, 64, 64, 64, 64, 64};
    if (cfg < 31 || cfg > 39) return ST_EINVAL;
    if (cfg >= 38 && (d.split_k > 1 || batch != 1)) return ST_EINVAL;
    if (cfg == 37 && (d.a2 || d.split_k > 1)) return ST_EINVAL;
    const long tiles = (long)((d.M + bms[cfg - 30] - 1) / bms[cfg - 30]) * ((d.N + bns[cfg - 30] - 1) / bns[cfg - 30]) * batch;
    int split = d.split_k;
    if (cfg >= 37) split = 1;
    if (split == 0) {
        split = 1;
        if (batch == 1 && d.workspace && d.K >= 512 && tiles < 256) split = (int)((256 + tiles - 1) / tiles);
        // exactly one workgroup per CU (N = 128 at M = 8 192) leaves every SIMD with ONE consumer wave -- two K halves give it two
        static const int sk2 = [] { const char* e = getenv("ST_SPLIT3_SK2"); return e ? atoi(e) : 1; }();      // measured: decoder chain 5.36 -> 5.23 ms (tools/decoder_bench.py); ST_SPLIT3_SK2=0 = off
        if (sk2 && batch == 1 && d.workspace && d.K >= 1024 && tiles == 256 && cfg == 34) split = 2;
        if (split > d.K / 256) split = d.K / 256;
        if (split > 16) split = 16;
        if (split < 1) split = 1;
        while (split > 1 && (int64_t)split * d.M * d.N > d.workspace_floats) --split;
    }
    if (split > 1) {
        if (batch != 1 || !d.workspace || (int64_t)split * d.M * d.N > d.workspace_floats) return ST_EINVAL;
        const int nkt = d.K / 32, per = (nkt + split - 1) / split;
        split = (nkt + per - 1) / per;                     // no empty slice
    }
    d.split_k = split;
    g_last_plan[0] = 8; g_last_plan[1] = cfg; g_last_plan[2] = split; g_last_plan[3] = 0;
    hipStream_t s = (hipStream_t)stream;
    // timing experiments (tools/split3_probe.py --diag): ST_SPLIT3_DIAG=1 consumers skip the MFMAs, 2 loaders skip the DMA; results are garbage
    static const int diag = [] { const char* e = getenv("ST_SPLIT3_DIAG"); return e ? atoi(e) : 0; }();
    if (diag == 1) return cfg == 34 ? launch_split3<2, 2, 1, 1, 3, 1>(d, s) : cfg == 32 ? launch_split3<2, 2, 2, 1, 4, 1>(d, s) : launch_split3<2, 2, 2, 2, 3, 1>(d, s);
    if (diag == 3) return launch_split3<2, 2, 1, 1, 3, 3>(d, s);
    if (diag == 4) return launch_split3<2, 2, 1, 1, 3, 4>(d, s);      // in-kernel clock stamps -> workspace (tools/split3_clock.py)
    if (diag == 2) return cfg == 34 ? launch_split3<2, 2, 1, 1, 3, 2>(d, s) : cfg == 32 ? launch_split3<2, 2, 2, 1, 4, 2>(d, s) : launch_split3<2, 2, 2, 2, 3, 2>(d, s);
    if (cfg == 37) {
        int G = 512 / batch;
        if (G < 1) G = 1;
        if (G > ntl64) G = (int)ntl64;
        const size_t lds = (size_t)3 * 3 * 128 * 64;
        void (*k)(const st_gemm_desc) = d.c_t ? conv_gemm_split3_persist_kernel<true> : conv_gemm_split3_persist_kernel<false>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        g_last_plan[3] = 1;
        hipLaunchKernelGGL(k, dim3(G, 1, batch), dim3(512), lds, s, d);
        ST_CHECK_LAUNCH();
        return ST_OK;
    }
    switch (cfg) {
        case 38: return launch_split3_kpar<2, 2, 2, 1, 4>(d, s);
        case 39: return launch_split3_kpar<2, 2, 1, 1, 4>(d, s);
        case 31: return launch_split3<2, 2, 2, 2, 3>(d, s);
        case 32: return launch_split3<2, 2, 2, 1, 4>(d, s);
        case 33: return launch_split3<2, 2, 1, 2, 4>(d, s);
        case 35: return launch_split3<2, 2, 2, 1, 3>(d, s);
        case 36: return launch_split3<2, 2, 1, 1, 4>(d, s);
        default: return launch_split3<2, 2, 1, 1, 3>(d, s);
    }
}

// fp32 [rows, ldx] (C columns) -> three blocked bf16 planes [C/32][chunk_rows][32], plane_stride elements apart (csrc/gemm_split3.h)
extern "C" int st_split3_pack(const float* x, void* planes, int64_t rows, int32_t C, int64_t ldx, int64_t plane_stride, int64_t chunk_rows,
                              void* stream) {
    if (!x || !planes || rows <= 0 || C <= 0 || C % 32 || ldx < C || (ldx & 3) || chunk_rows < rows || ((uintptr_t)x & 15) ||
        ((uintptr_t)planes & 15) || (plane_stride & 7) || plane_stride < (int64_t)(C / 32) * chunk_rows * 32)
        return ST_EINVAL;
    const long long n = (long long)rows * (C / 8);
    hipLaunchKernelGGL(split3_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)planes, (long long)rows, (int)C,
                       (long long)ldx, (long long)plane_stride, (long long)chunk_rows);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// Checks + buffer extents of a descriptor that must run on the plain (non-persistent, unsplit) 64x64 LDS-DMA kernel.
static int pair_member_prepare(const st_gemm_desc* desc, st_gemm_desc& d) {
    d = *desc;
    if (!c_planes_ok(d)) return ST_EINVAL;
    if (!d.a || !d.w || !d.c || d.M <= 0 || d.N <= 0 || d.K <= 0) return ST_EINVAL;
    if (d.kh <= 0 || d.kw <= 0 || d.K != d.kh * d.kw * d.Cin || d.K < 128 || d.Cin % 32) return ST_EINVAL;
    if (d.Ho <= 0 || d.Wo <= 0 || d.M % (d.Ho * d.Wo)) return ST_EINVAL;
    if (d.epi != ST_EPI_STORE && !d.aux1) return ST_EINVAL;
    if (d.epi == ST_EPI_GRU && !d.aux2) return ST_EINVAL;
    if (d.epi == ST_EPI_ZR && (!d.c2 || (d.N & 1))) return ST_EINVAL;
    if (d.reserved0 != 0 || d.reserved1 != 0 || d.reserved2 != 0 || d.reserved3 != 0 || d.split3 || d.c_t || d.a_ln || d.batch > 1 || d.split_k > 1 || (d.tile_cfg != 0 && d.tile_cfg != 13)) return ST_EINVAL;
    if (d.a2 && (d.a2_channels <= 0 || d.a2_channels % 32 || d.a2_channels > d.Cin || ((uintptr_t)d.a2 & 15))) return ST_EINVAL;
    if (((uintptr_t)d.a & 15) || ((uintptr_t)d.w & 15) || (d.ldx & 3) || (d.ldw & 3)) return ST_EINVAL;
    const bool plain_mat = d.kh == 1 && d.kw == 1 && d.sh == 1 && d.sw == 1 && d.ph == 0 && d.pw == 0 && (int64_t)d.H * d.W == d.M &&
                           (int64_t)d.Ho * d.Wo == d.M;
    const int64_t hw = plain_mat ? 1 : (int64_t)d.Ho * d.Wo, nimg = d.M / hw, img_rows = plain_mat ? 1 : (int64_t)d.H * d.W;
    const int64_t ab = ((nimg * img_rows - 1) * d.ldx + d.Cin) * 4, wb = ((int64_t)(d.N - 1) * d.ldw + d.K) * 4;
    int64_t ldmax = d.ldc > d.N ? d.ldc : d.N;
    if (d.aux0 && d.ld_aux0 > ldmax) ldmax = d.ld_aux0;
    if (d.aux1 && d.ld_aux1 > ldmax) ldmax = d.ld_aux1;
    if (d.aux2 && d.ld_aux2 > ldmax) ldmax = d.ld_aux2;
    if (d.c2 && d.ldc2 > ldmax) ldmax = d.ldc2;
    if (wb >= (int64_t)ST_OOB || ab >= (int64_t)ST_OOB || ((int64_t)d.M + 256) * ldmax * 4 >= ((int64_t)1 << 31)) return ST_EINVAL;   // (no row chunking here)
    d.a_bytes = (uint32_t)ab; d.w_bytes = (uint32_t)wb;
    d.split_k = 1; d.batch = 1;
    return ST_OK;
}

extern "C" int st_conv_gemm_pair(const st_gemm_desc* desc0, const st_gemm_desc* desc1, void* stream) {
    if (!desc0 || !desc1) return ST_EINVAL;
    st_gemm_pair_args g;
    if (desc0->split3 || desc1->split3) {            // both split3: 64x64 tiles, no split-K, one launch of 8-wave workgroups
        if (!desc0->split3 || !desc1->split3) return ST_EINVAL;
        for (int i = 0; i < 2; ++i) {
            const st_gemm_desc* di = i ? desc1 : desc0;
            if (di->batch > 1 || di->split_k > 1 || (di->tile_cfg != 0 && di->tile_cfg != 34)) return ST_EINVAL;
            const int rc = split3_prepare(di, g.d[i]);
            if (rc) return rc;
            g.d[i].split_k = 1; g.d[i].batch = 1;
        }
        auto tiles3 = [](const st_gemm_desc& d) { return ((d.M + 63) / 64) * ((d.N + 63) / 64); };
        g.tiles0 = tiles3(g.d[0]);
        const int total3 = g.tiles0 + tiles3(g.d[1]);
        st_gemm_observer_fn obs3 = g_observer;
        g_last_plan[0] = 8; g_last_plan[1] = 34; g_last_plan[2] = 1; g_last_plan[3] = 2;
        if (obs3) { obs3(desc0, stream, 0, g_observer_user); obs3(desc0, stream, 1, g_observer_user); obs3(desc1, stream, 0, g_observer_user); }
        g_last_plan[3] = 3;
        auto k3 = conv_gemm_split3_pair_kernel<2, 2, 1, 1, 3>;
        const size_t lds3 = (size_t)3 * 3 * 128 * 64;
        (void)hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        hipLaunchKernelGGL(k3, dim3(total3), dim3(512), lds3, (hipStream_t)stream, g);
        if (obs3) obs3(desc1, stream, 1, g_observer_user);
        ST_CHECK_LAUNCH();
        return ST_OK;
    }
    int rc = pair_member_prepare(desc0, g.d[0]);
    if (rc) return rc;
    rc = pair_member_prepare(desc1, g.d[1]);
    if (rc) return rc;
    auto tiles = [](const st_gemm_desc& d) { return ((d.M + 63) / 64) * ((d.N + 63) / 64); };
    g.tiles0 = tiles(g.d[0]);
    const int total = g.tiles0 + tiles(g.d[1]);
    st_gemm_observer_fn obs = g_observer;
    // plan[3]: 2 = first member of a pair (no dispatch of its own: the observer sees an empty bracket), 3 = second member (the
    // pair's single dispatch and all of its time)
    g_last_plan[0] = 3; g_last_plan[1] = 13; g_last_plan[2] = 1; g_last_plan[3] = 2;
    if (obs) { obs(desc0, stream, 0, g_observer_user); obs(desc0, stream, 1, g_observer_user); obs(desc1, stream, 0, g_observer_user); }
    g_last_plan[3] = 3;
    auto k = conv_gemm_dma_pair_kernel<2, 2, 1, 1, 4>;
    const size_t lds = (size_t)4 * (64 + 64) * 32 * sizeof(float);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(total), dim3(256), lds, (hipStream_t)stream, g);
    if (obs) obs(desc1, stream, 1, g_observer_user);           // (the pair's time is attributed to its second member)
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// All-pairs correlation volume = batched A . B^T on the same MFMA core (K = C = 256 at 512^2).
extern "C" int st_corr_volume(const float* f1, const float* f2, float* vol, int32_t B, int32_t N1, int32_t N2,
                              int32_t C, void* stream) {
    if (!f1 || !f2 || !vol || B <= 0 || N1 <= 0 || N2 <= 0 || C <= 0) return ST_EINVAL;
    st_gemm_desc d = {};
    d.a = f1; d.w = f2; d.c = vol;
    d.M = N1; d.N = N2; d.K = C;
    d.H = 1; d.W = N1; d.Cin = C; d.ldx = C;
    d.kh = d.kw = 1; d.sh = d.sw = 1; d.ph = d.pw = 0; d.Ho = 1; d.Wo = N1;
    d.ldw = C; d.ldc = N2; d.alpha = 1.0f;
    d.batch = B; d.batch_stride_a = (int64_t)N1 * C; d.batch_stride_w = (int64_t)N2 * C;
    d.batch_stride_c = (int64_t)N1 * N2;
    return st_conv_gemm(&d, stream);
}

// 1 = one product with a transposed second store, 0 = two products.  The transposed store needs a shape of the LDS-DMA kernel
// (N % 4 == 0, C % 32 == 0, C >= 128, 16-byte aligned operands) AND a volume that the 32-bit buffer offsets reach in one piece:
// N * N * 4 < 2^31 for the transposed copy (conv_gemm_launch rejects c_t otherwise) and (N + 256) * N * 4 < 2^31, f1 / f2 extents
// < 2^31 for the row-chunked path, which cannot carry c_t (N > ~23 000, i.e. flow grids beyond ~1 200 x 1 200 pixels / 8).
extern "C" int st_corr_volume_both_plan(int32_t B, int32_t N, int32_t C, int32_t aligned16) {
    if (B <= 0 || N <= 0 || C <= 0) return 0;
    if ((N & 3) || C % 32 || C < 128 || !aligned16) return 0;
    const int64_t lim = (int64_t)1 << 31;
    if ((int64_t)N * N * 4 >= lim || ((int64_t)N + 256) * N * 4 >= lim || ((int64_t)(N - 1) * C + C) * 4 >= (int64_t)ST_OOB) return 0;
    return 1;
}

extern "C" int st_corr_volume_both(const float* f1, const float* f2, float* vol12, float* vol21, int32_t B, int32_t N, int32_t C,
                                   void* stream) {
    if (!f1 || !f2 || !vol12 || !vol21 || B <= 0 || N <= 0 || C <= 0) return ST_EINVAL;
    if (!st_corr_volume_both_plan(B, N, C, (int32_t)((((uintptr_t)f1 | (uintptr_t)f2 | (uintptr_t)vol21) & 15) == 0))) {
        // not a shape of the LDS-DMA kernel's transposed store: two products
        const int rc = st_corr_volume(f1, f2, vol12, B, N, N, C, stream);
        return rc ? rc : st_corr_volume(f2, f1, vol21, B, N, N, C, stream);
    }
    st_gemm_desc d = {};
    d.a = f1; d.w = f2; d.c = vol12; d.c_t = vol21; d.ld_ct = N;
    d.M = N; d.N = N; d.K = C;
    d.H = 1; d.W = N; d.Cin = C; d.ldx = C;
    d.kh = d.kw = 1; d.sh = d.sw = 1; d.ph = d.pw = 0; d.Ho = 1; d.Wo = N;
    d.ldw = C; d.ldc = N; d.alpha = 1.0f;
    d.batch = B; d.batch_stride_a = (int64_t)N * C; d.batch_stride_w = (int64_t)N * C;
    d.batch_stride_c = (int64_t)N * N;
    return st_conv_gemm(&d, stream);
}

// All-pairs volume(s) from the feature maps' planes (st_gemm_desc.split3; planes [3][C/32][rows][32], sample b = rows b*N ..): vol12[b] = f1[b] . f2[b]^T and,
// when vol21 is given, vol21[b] = its transpose from the same launch (encoder.py:359-369 for both flow directions).
extern "C" int st_corr_volume_split3(const void* f1_planes, const void* f2_planes, int64_t pstride, int64_t prows, float* vol12, float* vol21, int32_t B,
                                     int32_t N, int32_t C, void* stream) {
    if (!f1_planes || !f2_planes || !vol12 || B <= 0 || N <= 0 || C <= 0 || (C & 31) || prows < (int64_t)B * N) return ST_EINVAL;
    st_gemm_desc d = {};
    d.a = (const float*)f1_planes; d.w = (const float*)f2_planes; d.c = vol12; d.c_t = vol21; d.ld_ct = N;
    d.M = N; d.N = N; d.K = C;
    d.H = 1; d.W = N; d.Cin = C; d.ldx = C;
    d.kh = d.kw = 1; d.sh = d.sw = 1; d.ph = d.pw = 0; d.Ho = 1; d.Wo = N;
    d.ldw = C; d.ldc = N; d.alpha = 1.0f;
    d.batch = B; d.batch_stride_a = (int64_t)N * 32; d.batch_stride_w = (int64_t)N * 32; d.batch_stride_c = (int64_t)N * N;
    d.split3 = 1; d.a_plane_stride = pstride; d.w_plane_stride = pstride; d.a_rows = prows; d.w_rows = prows;
    d.tile_cfg = 37;
    return st_conv_gemm(&d, stream);
}

extern "C" int st_gemm_last_plan(int32_t* plan4) {
    if (!plan4) return ST_EINVAL;
    for (int i = 0; i < 4; ++i) plan4[i] = g_last_plan[i];
    return ST_OK;
}

// ABI self-check for bindings: size of st_gemm_desc as this library was compiled.
extern "C" int st_abi_gemm_desc_size(void) { return (int)sizeof(st_gemm_desc); }


// Optional profiling observer (bench.py's live roofline): called on the launching thread before (phase 0)
// and after (phase 1) the kernels of every st_conv_gemm are enqueued -- including the launches made by the
// operator-level entry points -- so the caller can record HIP events on `stream`.  NULL (default) = off.

// launches of the family that live in other translation units (csrc/patchembed.hip) report themselves through this: phase 0 before the
// launch, 1 after it (also records the plan of the calling thread: kernel id, tile, split-K, persistent)
bool st_internal_observe(const st_gemm_desc* od, void* stream, int phase, int plan_kernel) {
    if (phase == 1) { g_last_plan[0] = plan_kernel; g_last_plan[1] = 0; g_last_plan[2] = 1; g_last_plan[3] = 1; }
    st_gemm_observer_fn obs = g_observer;
    if (!obs) return false;
    obs(od, stream, phase, g_observer_user);
    return true;
}

extern "C" int st_set_gemm_observer(void* callback, void* user) {
    g_observer = (st_gemm_observer_fn)callback;
    g_observer_user = user;
    return ST_OK;
}

extern "C" int st_conv_gemm(const st_gemm_desc* desc, void* stream) {
    if (!desc) return ST_EINVAL;
    st_gemm_observer_fn obs = g_observer;
    if (!obs) return conv_gemm_launch(desc, stream);
    obs(desc, stream, 0, g_observer_user);
    const int rc = conv_gemm_launch(desc, stream);
    obs(desc, stream, 1, g_observer_user);
    return rc;
}
