// In-launch hand-off probe: what does ONE dependent hand-off between the workgroups of a strip cost, by protocol?
//   A  counter protocol (what the one-launch recurrences use): write-through stores, drain (s_waitcnt vmcnt(0)), barrier, one
//      atomic add per workgroup; the readers: one poller, barrier, sc1 loads of the strip's rows.
//   B  flag-in-data ("LL") protocol: every handed-over float travels as an 8-byte word (value, step stamp), write-through; the
//      readers load their operands directly (sc1) and repeat the loads until every stamp is the expected one.  No drain, no
//      atomic, no barrier on the waiting side; twice the bytes.
// Geometry of spk_teacher_seq_kernel / spk_bptt_seq_kernel: 256 workgroups x 512 threads, 8 strips of 16 rows x 512 columns,
// 32 workgroups per strip (each owns 16 columns and reads all 16 x 512 values of the strip every step, K split over 8 waves,
// cross-wave sum through LDS).  `local` = a strip's workgroups on one XCD (blockIdx & 7) or spread over all eight.
//   hipcc --offload-arch=gfx950 -O3 -o handoff_probe tools/probes/handoff_probe.hip && ./handoff_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int H = 512, ROWS = 16, TJ = 32, STRIPS = 8;
constexpr unsigned long long BOUND = 50000000ull;      // 0.5 s of 100 MHz ticks

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* ptr, size_t bytes) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, (int)bytes, 0x00020000);
}

template <bool LL>
__global__ __launch_bounds__(512) void k_handoff(float* slab, unsigned* cnt, int steps, int local, float* sink, unsigned* err) {
    __shared__ float red[8 * 64];
    __shared__ int ok_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
    const int strip = local ? (blockIdx.x & 7) : (blockIdx.x / TJ), jt = local ? (blockIdx.x >> 3) : (blockIdx.x % TJ);
    const bool owner = w < 4;
    const int orow = 4 * lq + (w & 3), ocol = jt * 16 + li;
    const size_t buf_elems = (size_t)STRIPS * ROWS * H;
    const size_t esz = LL ? 8 : 4;
    const auto r = rsrc(slab, 2 * buf_elems * esz);
    if (tid == 0) ok_s = 1;
    __syncthreads();
    float carry = (float)blockIdx.x;
    bool dead = false;                                  // a wave that has given up once does not wait again
    for (int t = 0; t < steps; ++t) {
        const size_t in0 = ((size_t)(t & 1) * STRIPS + strip) * ROWS * H, out0 = ((size_t)((t + 1) & 1) * STRIPS + strip) * ROWS * H;
        float v = 0.f;
        if (t > 0) {
            if (!LL) {
                if (tid == 0 && ok_s) {
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    while (__hip_atomic_load(cnt + strip * steps + (t - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)TJ) {
                        __builtin_amdgcn_s_sleep(1);
                        if (__builtin_amdgcn_s_memrealtime() - t0 > BOUND) { ok_s = 0; *err = 1; break; }
                    }
                }
                __syncthreads();
                // row li, columns 64 w + 16 i + 4 lq .. +3
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        r, (int)((in0 + (size_t)li * H + 64 * w + 16 * i + 4 * lq) * 4), 0, 16));
                    v += (a[0] + a[1]) + (a[2] + a[3]);
                }
            } else {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                bool good = dead;
                while (!good) {
                    u32x4 q[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh)
                            q[2 * i + hh] = __builtin_amdgcn_raw_buffer_load_b128(
                                r, (int)((in0 + (size_t)li * H + 64 * w + 16 * i + 4 * lq + 2 * hh) * 8), 0, 16);
                    bool mine = true;
                    v = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        mine = mine && q[i][1] == (unsigned)t && q[i][3] == (unsigned)t;
                        v += __uint_as_float(q[i][0]) + __uint_as_float(q[i][2]);
                    }
                    good = __builtin_amdgcn_ballot_w64(mine) == ~0ull;
                    if (!good && __builtin_amdgcn_s_memrealtime() - t0 > BOUND) { *err = 1; dead = true; break; }
                }
            }
        }
        red[w * 64 + lane] = v;
        __syncthreads();
        if (owner) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += red[q * 64 + lane];
            carry = carry * 0.5f + s * 1e-3f;
            if (LL) {
                const u32x2 pk = {__float_as_uint(carry), (unsigned)(t + 1)};
                __builtin_amdgcn_raw_buffer_store_b64(pk, r, (int)((out0 + (size_t)orow * H + ocol) * 8), 0, 16);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(carry), r, (int)((out0 + (size_t)orow * H + ocol) * 4), 0, 16);
            }
        }
        if (!LL) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(cnt + strip * steps + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __syncthreads();      // red[] is reused by the next step
        }
    }
    if (owner && carry == 123.456f) sink[0] = carry;
}

int main() {
    const int steps = 2000;
    float* slab; unsigned* cnt; float* sink; unsigned* err;
    const size_t slab_bytes = 2ull * STRIPS * ROWS * H * 8;
    CK(hipMalloc(&slab, slab_bytes)); CK(hipMalloc(&cnt, sizeof(unsigned) * STRIPS * steps)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&err, 4));
    CK(hipMemset(err, 0, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int local = 1; local >= 0; --local)
        for (int ll = 0; ll < 2; ++ll) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipMemsetAsync(slab, 0, slab_bytes, st));
                CK(hipMemsetAsync(cnt, 0, sizeof(unsigned) * STRIPS * steps, st));
                CK(hipEventRecord(a, st));
                if (ll) hipLaunchKernelGGL(k_handoff<true>, dim3(256), dim3(512), 0, st, slab, cnt, steps, local, sink, err);
                else hipLaunchKernelGGL(k_handoff<false>, dim3(256), dim3(512), 0, st, slab, cnt, steps, local, sink, err);
                CK(hipEventRecord(b, st));
                CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            unsigned e = 0; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
            printf("%-22s %-28s %6.2f us per dependent step%s\n", local ? "strip on one XCD" : "strip over 8 XCDs",
                   ll ? "flag-in-data (8 B / float)" : "counter + drain + poll", best * 1e3f / steps, e ? "  (TIMED OUT)" : "");
        }
    return 0;
}
