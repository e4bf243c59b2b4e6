#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}
template <int NACC> void run(const char* name, int wgs, int threads, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<NACC><<<wgs, threads>>>(out, 100, 1.f, 1.f);
    hipEventRecord(e0);
    mfma_loop<NACC><<<wgs, threads>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)wgs * threads / 64, fl = waves * iters * 16.0 * 32 * 32 * 2 * 2;
    printf("%s: wgs %d x %d thr  %.3f ms  %.1f TFLOP/s  (cycles/MFMA/SIMD at 2.4GHz: %.1f)\n", name, wgs, threads, ms, fl / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (iters * 16.0 * (waves / 1024.0)));
}
int main() {
    float* out; hipMalloc(&out, 4);
    run<1>("1 acc chain, 1 wave/SIMD", 256, 256, out);
    run<2>("2 acc, 1 wave/SIMD", 256, 256, out);
    run<4>("4 acc, 1 wave/SIMD", 256, 256, out);
    run<1>("1 acc chain, 2 waves/SIMD", 512, 256, out);
    run<1>("1 acc chain, 4 waves/SIMD", 1024, 256, out);
    run<4>("4 acc, 2 waves/SIMD", 512, 256, out);
    run<1>("1 acc, 1 wave per CU only", 256, 64, out);
    return 0;
}
