// Which fp32 MFMA shape sustains more FLOP/s under load?  v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32 issue the same FLOPs per
// cycle (64 FLOP/clk/SIMD), but MI355X_MICROARCH.md (DVFS give-back, item 7) reports that for bf16 the 16x16 shape held a ~15 % higher
// clock on random data.  Bare loops, operands in registers (random, non-trivial), long launches; wall TFLOP/s and the in-kernel clock
// (s_memtime / s_memrealtime) are printed.   hipcc --offload-arch=gfx950 -O3 mfma_shape_clock.hip -o /tmp/msc && /tmp/msc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* out, unsigned long long* clk, int iters) {
    const int lane = threadIdx.x & 63;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(blockIdx.x * 256 + threadIdx.x + 977 * i) & 65535]; b[i] = in[(threadIdx.x * 7 + 131 * i + blockIdx.x) & 65535]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[2];
        for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {      // 16 MFMAs of 4096 FLOP
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[7 - u], acc[1], 0, 0, 0);
            }
        }
        for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    } else {
        f32x4 acc[8];
        for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {      // 32 MFMAs of 2048 FLOP
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + j) & 7], b[(2 * u + j) & 7], acc[j], 0, 0, 0);
            }
        }
        for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && (threadIdx.x >> 6) == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    if (s == 12345.678f) out[0] = s;
}

template <int SHAPE> void run(const char* name, int wgs, const float* in, float* out, unsigned long long* clk, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<SHAPE><<<wgs, 256>>>(in, out, clk, iters);      // ~1 s of back-to-back load before the timed launch
    hipEventRecord(e0);
    k<SHAPE><<<wgs, 256>>>(in, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2048];
    hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * wgs, hipMemcpyDeviceToHost);
    double ratio = 0; for (int i = 0; i < wgs; ++i) ratio += (double)h[2 * i] / (double)h[2 * i + 1];
    const double flop = (double)wgs * 4 * iters * 16.0 * 4096.0;
    printf("%-28s wgs %4d: %8.2f ms  %7.1f TFLOP/s wall   in-kernel clock %.3f GHz\n", name, wgs, ms, flop / ms / 1e9, ratio / wgs * 0.1);
}
int main() {
    float *in, *out; unsigned long long* clk;
    hipMalloc(&in, 65536 * 4); hipMalloc(&out, 4); hipMalloc(&clk, 2048 * 16);
    float* h = (float*)malloc(65536 * 4);
    srand(1); for (int i = 0; i < 65536; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep)
        for (int wgs : {256, 512}) {
            run<32>("v_mfma_f32_32x32x2_f32", wgs, in, out, clk, 400000 / (wgs / 256));
            run<16>("v_mfma_f32_16x16x4_f32", wgs, in, out, clk, 400000 / (wgs / 256));
        }
    return 0;
}
