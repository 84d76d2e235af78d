#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* in, float* o32, float* o16) {
    float v = in[threadIdx.x];
    unsigned u = __builtin_bit_cast(unsigned, v);
    unsigned u2 = u;
    asm volatile("" : "+v"(u2));   // distinct register, opaque to CSE
    auto r = __builtin_amdgcn_permlane32_swap(u, u2, false, false);
    o32[threadIdx.x] = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
    unsigned w = u, w2 = u;
    asm volatile("" : "+v"(w2));
    auto q = __builtin_amdgcn_permlane16_swap(w, w2, false, false);
    o16[threadIdx.x] = __builtin_bit_cast(float, q[0]) + __builtin_bit_cast(float, q[1]);
}
int main() {
    float h[64], a[64], b[64];
    for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i % 16)) + i * 100000.f;
    float *d, *d32, *d16;
    hipMalloc(&d, 256); hipMalloc(&d32, 256); hipMalloc(&d16, 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, d32, d16);
    hipMemcpy(a, d32, 256, hipMemcpyDeviceToHost); hipMemcpy(b, d16, 256, hipMemcpyDeviceToHost);
    int ok32 = 1, ok16 = 1;
    for (int i = 0; i < 64; ++i) {
        if (a[i] != h[i] + h[i ^ 32]) ok32 = 0;
        if (b[i] != h[i] + h[i ^ 16]) ok16 = 0;
    }
    printf("permlane32_swap xor32-sum ok=%d  permlane16_swap xor16-sum ok=%d\n", ok32, ok16);
    for (int i = 0; i < 64; i += 9) printf("lane %d: in %.0f o32 %.0f (want %.0f) o16 %.0f (want %.0f)\n", i, h[i], a[i], h[i] + h[i ^ 32], b[i], h[i] + h[i ^ 16]);
    return 0;
}
