#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// probe: buffer_load_dwordx4 ... lds with an out-of-range offset: does it write zeros into LDS or leave it?
__global__ void probe(const float* src, int bytes, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4];
    const int lane = threadIdx.x;
    for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = 777.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
    const unsigned off = (lane & 1) ? 0x80000000u : (unsigned)lane * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}
int main() {
    float *src, *out; hipMalloc(&src, 64 * 16); hipMalloc(&out, 64 * 16);
    float h[256]; for (int i = 0; i < 256; ++i) h[i] = (float)i + 1;
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(src, 64 * 16, out);
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 6; ++l) printf("lane %d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
