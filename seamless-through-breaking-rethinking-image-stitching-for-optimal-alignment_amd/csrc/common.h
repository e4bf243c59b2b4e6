// Shared helpers for the gfx950 stitching kernels (MI355X / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ST_OK 0
#define ST_EINVAL 1001

#define ST_CHECK_LAUNCH()                         \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

// epilogue activations
enum { ST_ACT_NONE = 0, ST_ACT_RELU = 1, ST_ACT_GELU = 2, ST_ACT_SIGMOID = 3, ST_ACT_TANH = 4 };
// epilogue combine modes (v = act(alpha*acc + bias))
enum {
    ST_EPI_STORE = 0,     // out = v
    ST_EPI_ADD = 1,       // out = v + aux1                       (residual)
    ST_EPI_MUL = 2,       // out = v * aux1                       (GRU r*h)
    ST_EPI_GRU = 3,       // out = (1-aux1)*aux2 + aux1*v         (GRU state update, aux1=z, aux2=h)
    ST_EPI_AXPY = 4,      // out = aux1 + (*scale_ptr)*v          (GMA aggregate: fmap + gamma*out)
    ST_EPI_ZR = 5         // cols < N/2: out = v ; cols >= N/2: c2 = v*aux1   (fused GRU z | r*h)
};

// Exact (erf) GELU of nn.GELU(), x * Phi(x), branch-free in ~13 VALU instructions: with s = |x| / sqrt(2),
// erfc(s) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-s^2), t = 1 / (1 + p s)  (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7),
// Phi(x) = 1 - erfc(s) / 2 for x >= 0 and erfc(s) / 2 for x < 0 -- the negative side has no 1 + erf cancellation.
// Measured against the fp64 value on 4 M points of [-10, 10]: max abs error 4.7e-7, max relative error 1.7e-5 where
// |gelu| > 1e-2; torch's own fp32 CPU GELU (the reference's arithmetic): 1.2e-6 and 6.4e-5.
// ST_EXACT_TRANSCENDENTALS (diagnostic build only: `ST_EXACT_TRANSCENDENTALS=1 python <pkg>/build.py --force`): ocml's erff / expf / tanhf with IEEE
// division instead of the three fast forms below -- used once per round to attribute parity drift to the epilogue arithmetic
// (profiles/r4_exact_transcendentals.txt); never shipped: ~2x the VALU instructions on the datapath the fp32 MFMA shares.
#ifdef ST_EXACT_TRANSCENDENTALS
__device__ __forceinline__ float st_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float st_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ float st_tanh(float v) { return tanhf(v); }
#else
__device__ __forceinline__ float st_gelu(float x) {
    const float s = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, s, 1.0f));
    float y = fmaf(1.061405429f, t, -1.453152027f);
    y = fmaf(y, t, 1.421413741f);
    y = fmaf(y, t, -0.284496736f);
    y = fmaf(y, t, 0.254829592f);
    y = y * t * __builtin_amdgcn_exp2f(s * s * -1.44269504088896340736f) * 0.5f;     // erfc(s) / 2
    return x * (x >= 0.f ? 1.0f - y : y);
}

// sigmoid / tanh of the GRU gates (gru.py:47-51) on the hardware exp2 / reciprocal (2 and 7 VALU instructions instead of ~25 and ~40 for
// expf + IEEE division and ocml's tanhf -- in the epilogue of the four SepConvGRU GEMMs of every refinement iteration, on the datapath the
// fp32 MFMA shares).  tanh(x) = sign(x) (1 - e) / (1 + e), e = exp(-2|x|) in (0, 1]: no cancellation for large |x|, an exact subtraction for
// small.  Against fp64 on 5 M points of [-20, 20] (emulated roundings): max abs error 9.1e-8 (sigmoid), 1.3e-7 (tanh); torch's own fp32
// CPU functions: 8.9e-8 and 3.2e-8.
__device__ __forceinline__ float st_sigmoid(float v) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}
__device__ __forceinline__ float st_tanh(float v) {
    const float e = __builtin_amdgcn_exp2f(fabsf(v) * -2.88539008177792681472f);
    return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), v);
}
#endif

__device__ __forceinline__ float st_act(float v, int act) {
    switch (act) {
        case ST_ACT_RELU: return v > 0.f ? v : 0.f;
        case ST_ACT_GELU: return st_gelu(v);
        case ST_ACT_SIGMOID: return st_sigmoid(v);
        case ST_ACT_TANH: return st_tanh(v);
        default: return v;
    }
}

// The fp32 natural log of the TPS kernel term r^2 log r^2 (torch_tps_transform.py:113-114,161, kornia_tps.py:45).  torch-CPU's
// float `torch.log` is MKL VML's vsLn in HA mode, not SLEEF (the bits do not change with ATEN_CPU_CAPABILITY=default/avx2/avx512,
// and `torch.logit`, which calls the same VML entry, agrees on 10^7 inputs): a closed algorithm, measured here to return the
// CORRECTLY ROUNDED fp32 log on 99.97 % of 5e7 inputs (denormals, [1e-8, 8], random bit patterns) and the neighbouring float
// otherwise.  ocml's fp64 log (< 1 ulp of fp64) rounded once to fp32 is the correctly rounded value except within 2^-29 (relative)
// of a rounding midpoint: the closest statable function.  It matters because the (N+3)^2 solve amplifies a 1-ulp change of a kernel
// entry by the system's condition number (ocml's fp32 logf: T off by 4.4e-5 relative; this: 1.8e-7, tests/test_ops_gpu.py::test_tps).
__device__ __forceinline__ float st_logf_cr(float x) { return (float)log((double)x); }

// ---- exact three-way bf16 split of an fp32 value (operand format of csrc/gemm_split3.h) --------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// x -> (hi, mid, lo), x == hi + mid + lo exactly for every finite x with |x| >= 2^-110 (below that lo leaves bf16's
// subnormal range).  +-inf / NaN stay in hi alone (mid = lo = 0): inf - inf would otherwise make the residual NaN and turn
// inf . finite into NaN; an x that only bf16 rounding pushes to inf keeps the largest finite bf16 as hi.
__device__ __forceinline__ void st_split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)x;
    float hf = (float)hi;
    if (__builtin_isinf(hf) || hf != hf) {
        if (__builtin_isinf(x) || x != x) { mid = (__bf16)0.f; lo = (__bf16)0.f; return; }
        hf = copysignf(3.3895313892515355e38f, x);          // 0x7F7F0000: largest finite bf16
        hi = (__bf16)hf;
    }
    const float r1 = x - hf;
    mid = (__bf16)r1;
    lo = (__bf16)(r1 - (float)mid);
}

// one value into the three planes of a blocked plane tensor (byte offset `off` inside plane 0, planes `plane_bytes` apart)
__device__ __forceinline__ unsigned short st_bf16_bits(__bf16 v) { return __builtin_bit_cast(unsigned short, v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
