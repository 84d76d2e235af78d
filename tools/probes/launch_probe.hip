// Dependent-launch cost probe: how long does a chain of N trivial kernels take per launch on this stack, by launch form?
//   hipcc --offload-arch=gfx950 -O3 -o launch_probe tools/probes/launch_probe.hip && ./launch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Big { float* p[48]; int n[16]; };
__global__ __launch_bounds__(512) void k_empty(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
__global__ __launch_bounds__(512) void k_big(Big b) { if (b.p[0] && threadIdx.x == 9999) b.p[0][0] = 1.f; }
__global__ __launch_bounds__(1024) void k_lds(float* p) { extern __shared__ float sm[]; if (threadIdx.x == 9999) { sm[0] = 1.f; p[0] = sm[0]; } }
__global__ __launch_bounds__(512) void k_touch(float* p, int n) {   // every workgroup writes 2 KB: dirty lines at the boundary
    const int i = blockIdx.x * 512 + threadIdx.x; if (i < n) p[i] = p[i] + 1.f; }

template <class F> static int timeit(const char* name, hipStream_t st, int N, F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 50; ++i) f();
    CK(hipStreamSynchronize(st));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a, st));
        for (int i = 0; i < N; ++i) f();
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("%-58s %7.2f us / launch\n", name, best * 1e3f / N);
    return 0;
}

int main() {
    float* d; CK(hipMalloc(&d, 64 << 20));
    CK(hipMemset(d, 0, 64 << 20));
    hipStream_t st, stnb; CK(hipStreamCreate(&st)); CK(hipStreamCreateWithFlags(&stnb, hipStreamNonBlocking));
    const int N = 2000;
    Big big{}; big.p[0] = d;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    timeit("empty 256x512, created stream", st, N, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, st, d); });
    timeit("empty 256x512, non-blocking stream", stnb, N, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, stnb, d); });
    timeit("empty 256x512, null stream", 0, N, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, 0, d); });
    timeit("empty 32x64, created stream", st, N, [&] { hipLaunchKernelGGL(k_empty, dim3(32), dim3(64), 0, st, d); });
    timeit("448-byte kernarg 256x512", st, N, [&] { hipLaunchKernelGGL(k_big, dim3(256), dim3(512), 0, st, big); });
    timeit("128 KB dynamic LDS 256x1024", st, N, [&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(1024), 128 * 1024, st, d); });
    timeit("touch 512 KB (r+w) 256x512", st, N, [&] { hipLaunchKernelGGL(k_touch, dim3(256), dim3(512), 0, st, d, 256 * 512); });
    timeit("touch 16 MB (r+w) 8192x512", st, N, [&] { hipLaunchKernelGGL(k_touch, dim3(8192), dim3(512), 0, st, d, 8192 * 512); });
    timeit("empty + hipGetLastError after each", st, N, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, st, d); (void)hipGetLastError(); });
    // graph of 100 empty kernels
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, st, d);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(a, st));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-58s %7.2f us / launch\n", "graph of 100 empty 256x512 kernels, 20 replays", ms * 1e3f / 2000);
    }
    return 0;
}
