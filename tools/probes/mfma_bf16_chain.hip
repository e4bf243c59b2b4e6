// How fast does ONE wave per SIMD issue v_mfma_f32_32x32x16_bf16 when consecutive MFMAs share their accumulator (a dependent chain) and when
// they rotate over 2 / 4 accumulators -- alone and with independent VALU work between them (8 fp32 FMAs, or 6 FMAs + v_exp_f32 + v_rcp_f32)?
// The question behind csrc/mlp_split3.h (one wave per SIMD, six products per 16 k on one accumulator tile).
//   hipcc --offload-arch=gfx950 -O3 mfma_bf16_chain.hip -o /tmp/mbc && /tmp/mbc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int VALU>
__global__ __launch_bounds__(512, 1) void k(const float* __restrict__ in, float* out, unsigned long long* clk, int iters) {
    extern __shared__ float pad[];
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)in[(threadIdx.x * 8 + e + 977 * i) & 65535]; b[i][e] = (__bf16)in[(threadIdx.x * 5 + e * 3 + 131 * i) & 65535]; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[(threadIdx.x + 31 * i) & 65535];
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 3], b[(u >> 2) & 3], acc[u % NACC], 0, 0, 0);
            if (VALU == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.25f);
            } else if (VALU == 2) {
#pragma unroll
                for (int i = 0; i < 6; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.25f);
                v[6] = __builtin_amdgcn_exp2f(v[6]); v[7] = __builtin_amdgcn_rcpf(v[7]);
            }
            if (VALU) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 8, 0); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC, int VALU>
void run(const float* in, float* out, unsigned long long* clk, const char* what, int threads = 256) {
    const int iters = 400, G = 256;
    hipFuncSetAttribute((const void*)k<NACC, VALU>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NACC, VALU>), dim3(G), dim3(threads), 100 * 1024, 0, in, out, clk, iters);
    hipDeviceSynchronize();
    unsigned long long h[2 * 256];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, ticks = 0;
    for (int i = 0; i < G; ++i) { cyc += h[2 * i]; ticks += h[2 * i + 1]; }
    cyc /= G; ticks /= G;
    printf("%-58s %d wave(s) per SIMD: %6.1f cycles per MFMA of a wave = %5.1f per MFMA of the SIMD (s_memtime), in-kernel clock %.2f GHz\n", what, threads / 256, cyc / (iters * 24.0), cyc / (iters * 24.0) / (threads / 256), cyc / (ticks * 10.0));
}

int main() {
    float *in, *out; unsigned long long* clk;
    hipMalloc(&in, 65536 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 2 * 256 * 8);
    float* h = (float*)malloc(65536 * 4);
    for (int i = 0; i < 65536; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    run<1, 0>(in, out, clk, "one accumulator (dependent chain), MFMAs only");
    run<2, 0>(in, out, clk, "two accumulators, MFMAs only");
    run<4, 0>(in, out, clk, "four accumulators, MFMAs only");
    run<1, 1>(in, out, clk, "one accumulator + 8 FMAs per MFMA");
    run<2, 1>(in, out, clk, "two accumulators + 8 FMAs per MFMA");
    run<1, 2>(in, out, clk, "one accumulator + 6 FMAs + exp + rcp per MFMA");
    run<2, 2>(in, out, clk, "two accumulators + 6 FMAs + exp + rcp per MFMA");
    run<1, 0>(in, out, clk, "one accumulator (dependent chain), MFMAs only", 512);
    run<1, 1>(in, out, clk, "one accumulator + 8 FMAs per MFMA", 512);
    run<1, 2>(in, out, clk, "one accumulator + 6 FMAs + exp + rcp per MFMA", 512);
    return 0;
}
