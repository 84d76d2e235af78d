// Diagnostics of the development build (-DCIC_DEVTOOLS, libcic_hip_dev.so; declared in include/cic_dev.h): shader-clock
// probe and an empty launch, used by tools/ to interpret timings.  The product library does not contain them.
#include "cic_common.h"
#ifdef CIC_DEVTOOLS
#include "cic_dev.h"

unsigned long long g_spin_ticks = 1000ull * 100000ull;
int g_fault_loop = 0, g_fault_wg = -1;
extern "C" int cic_debug_spin_ticks(unsigned long long ticks) { g_spin_ticks = ticks ? ticks : 1000ull * 100000ull; return 0; }
extern "C" int cic_debug_handoff_fault(int loop_bits, int wg) { g_fault_loop = loop_bits; g_fault_wg = loop_bits ? wg : -1; return 0; }

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
// Fill-rate probe: `wgs` workgroups of 512 threads, each streaming `f4_per_wg` float4 (in 8 loads per lane at a time) out
// of a region of `region_f4` float4 that `share` consecutive workgroups (same XCD: ids b, b+8, ...) read together.
__global__ __launch_bounds__(512) void stream_probe_kernel(const f32x4* __restrict__ src, int64_t region_f4, int f4_per_wg,
                                                           int share, int regions, float* __restrict__ sink) {
    // workgroups b and b + 8 land on the same XCD: group g = the `share` workgroups that read one region
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int group = (slot / share) * 8 + xcd;
    const f32x4* base = src + (size_t)(group % regions) * region_f4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < f4_per_wg; i += 512 * 8) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = i + 512 * j;
            v[j] = base[(q < f4_per_wg ? q : i) % region_f4];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[blockIdx.x] = acc[0];
}
// Stand-in for a collective's kernel: `wgs` workgroups that each HOLD a CU's worth of LDS (so that no 100 KB workgroup of a
// one-launch recurrence fits beside them) for `ticks` of the 100 MHz clock, doing nothing else.
__global__ __launch_bounds__(256) void hold_cus_kernel(unsigned long long ticks, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) float hold_lds[];
    hold_lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    if (hold_lds[threadIdx.x] < 0.f && sink) sink[0] = 1u;
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
extern "C" int cic_debug_stream_probe(const float* src, int64_t region_floats, int regions, int wgs, int floats_per_wg,
                                      int share, float* sink, int iters, double* avg_us, cic_stream_t s) {
    CIC_REQUIRE(src && sink && avg_us && wgs > 0 && (wgs % 8) == 0 && share >= 1 && regions >= 1 && iters > 0 &&
                region_floats >= 4 && floats_per_wg >= 4);
    hipEvent_t e0, e1;
    CIC_HIP(hipEventCreate(&e0));
    CIC_HIP(hipEventCreate(&e1));
    auto go = [&]() {
        hipLaunchKernelGGL(stream_probe_kernel, dim3(wgs), dim3(512), 0, cic_s(s), reinterpret_cast<const f32x4*>(src),
                           region_floats / 4, floats_per_wg / 4, share, regions, sink);
    };
    for (int i = 0; i < 3; ++i) go();
    CIC_HIP(hipEventRecord(e0, cic_s(s)));
    for (int i = 0; i < iters; ++i) go();
    CIC_HIP(hipEventRecord(e1, cic_s(s)));
    CIC_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    CIC_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_us = (double)ms * 1e3 / iters;
    CIC_LAUNCH_CHECK();
    return 0;
}
extern "C" int cic_debug_hold_cus(int wgs, unsigned long long ticks, int lds_bytes, cic_stream_t s) {
    CIC_REQUIRE(wgs > 0 && wgs <= 256 && lds_bytes >= 1024 && lds_bytes <= 160 * 1024 && ticks <= 100000000ull);
    static DeviceOnce attr_set;
    if (attr_set.first())
        CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hold_cus_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(hold_cus_kernel, dim3(wgs), dim3(256), (size_t)lds_bytes, cic_s(s), ticks, (unsigned*)nullptr);
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
