#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* src, int bytes, float* out, float* dst, int dbytes) {
    const int lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
    // voffset in range, soffset pushes past num_records
    float a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, 0, 0));
    float b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, 512, 0));   // 512 + lane*4 >= 256 for all lanes
    float c = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, 128, 0));   // in range for lanes < 32
    out[lane] = a; out[64 + lane] = b; out[128 + lane] = c;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, dbytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 5.0f), rd, lane * 4, 128, 0);    // lanes >= 32 out of range
}
int main() {
    float *src, *out, *dst; hipMalloc(&src, 4096); hipMalloc(&out, 4096); hipMalloc(&dst, 4096);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)i + 1;
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    hipMemset(dst, 0, 4096);
    probe<<<1, 64>>>(src, 256, out, dst, 256);
    hipMemcpy(h, out, 192 * 4, hipMemcpyDeviceToHost);
    printf("a[0]=%g a[63]=%g | b[0]=%g b[63]=%g (expect 0 if soffset is range-checked) | c[0]=%g c[31]=%g c[32]=%g c[63]=%g\n", h[0], h[63], h[64], h[127], h[128], h[159], h[160], h[191]);
    hipMemcpy(h, dst, 512, hipMemcpyDeviceToHost);
    printf("store: dst[32+0]=%g dst[32+31]=%g dst[64]=%g dst[95]=%g (expect 5 5 0 0 if soffset range-checked)\n", h[32], h[63], h[64], h[95]);
    return 0;
}
