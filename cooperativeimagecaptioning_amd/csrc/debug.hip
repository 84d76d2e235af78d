// Diagnostics of the development build (-DCIC_DEVTOOLS, libcic_hip_dev.so; declared in include/cic_dev.h): shader-clock
// probe and an empty launch, used by tools/ to interpret timings.  The product library does not contain them.
#include "cic_common.h"
#ifdef CIC_DEVTOOLS
#include "cic_dev.h"

namespace {
__global__ void clock_probe_kernel(float* out, int spin) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    float x = 1.0f;
    for (int i = 0; i < spin; ++i) x = x * 1.0000001f + 1e-9f;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[0] = (float)(t1 - t0) / (float)(r1 - r0) * 100.0f;   // MHz (s_memrealtime ticks at 100 MHz)
    out[1] = x;
}
__global__ void empty_kernel(int* p) {
    if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 1;
}
}  // namespace

extern "C" int cic_debug_clock_mhz(float* out2, int spin, cic_stream_t s) {
    CIC_REQUIRE(out2 && spin > 0);
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, cic_s(s), out2, spin);
    CIC_LAUNCH_CHECK();
    return 0;
}
extern "C" int cic_debug_empty(int grid, int block, cic_stream_t s) {
    CIC_REQUIRE(grid > 0 && block > 0 && block <= 1024);
    hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(block), 0, cic_s(s), (int*)nullptr);
    CIC_LAUNCH_CHECK();
    return 0;
}
#endif  // CIC_DEVTOOLS
