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

__device__ __forceinline__ float st_act(float v, int act) {
    switch (act) {
        case ST_ACT_RELU: return v > 0.f ? v : 0.f;
        case ST_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        case ST_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        case ST_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

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
