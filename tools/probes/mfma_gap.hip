#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// MODE: 0 pure, 1: NV valu ops per MFMA, 2: NV salu ops per mfma, 3: ds_read_b128 pair per 4 MFMAs (+NV valu)
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, int seed) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 36 * 4];
    for (int i = threadIdx.x; i < 64 * 36 * 4; i += 256) lds[i] = 0.f;
    __syncthreads();
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = a0, b = a0;
    int v0 = threadIdx.x, v1 = seed, v2 = 3, v3 = 5;
    int s0 = seed;
    typedef float f4 __attribute__((ext_vector_type(4))); f4 fa = {a0, a0, a0, a0}, fb = fa;
    const float* p = lds + (threadIdx.x & 63) * 36;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 3) {
                if ((u & 3) == 0) {
                    asm volatile("ds_read_b128 %0, %1" : "=v"(fa) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) float*)(p + (u & 4))));
                    asm volatile("ds_read_b128 %0, %1 offset:9216" : "=v"(fb) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) float*)(p + (u & 4))));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa), "+v"(fb));
                }
                a = fa.x; b = fb.x;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 1 || MODE == 3) {
#pragma unroll
                for (int q = 0; q < NV; ++q) asm volatile("v_add_u32 %0, %1, %2" : "=v"(v0) : "v"(v0), "v"(v1));
            }
            if (MODE == 2) {
#pragma unroll
                for (int q = 0; q < NV; ++q) asm volatile("s_add_i32 %0, %1, %2" : "=s"(s0) : "s"(s0), "s"(seed) : "scc");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = v0 + v2 + v3 + s0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[0] = s;
}
template <int MODE, int NV> void run(const char* name, int wgs, float* out) {
    const int iters = 5000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NV><<<wgs, 256>>>(out, 50, 0.f, 1);
    hipEventRecord(e0);
    k<MODE, NV><<<wgs, 256>>>(out, iters, 0.f, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)wgs * 4;
    printf("%-44s wgs %4d: %.1f cycles/MFMA/SIMD\n", name, wgs, ms * 1e-3 * 2.4e9 / (iters * 16.0 * (waves / 1024.0)));
}
int main() {
    float* out; hipMalloc(&out, 4);
    for (int wgs : {256, 512}) {
        run<0, 0>("pure", wgs, out);
        run<1, 2>("2 VALU per MFMA", wgs, out);
        run<1, 4>("4 VALU per MFMA", wgs, out);
        run<1, 8>("8 VALU per MFMA", wgs, out);
        run<1, 12>("12 VALU per MFMA", wgs, out);
        run<2, 4>("4 SALU per MFMA", wgs, out);
        run<2, 12>("12 SALU per MFMA", wgs, out);
        run<3, 0>("2 ds_read_b128 per 4 MFMA", wgs, out);
        run<3, 4>("2 ds_read_b128 per 4 MFMA + 4 VALU/MFMA", wgs, out);
    }
    return 0;
}
