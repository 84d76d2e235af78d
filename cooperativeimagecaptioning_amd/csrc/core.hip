// Library plumbing (version, error text) and the counter-based RNG kernels.
#include <stdarg.h>
#include <string.h>
#include "cic_common.h"

static thread_local char g_err[512] = "";

void cic_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int cic_version(void) { return 100; }
extern "C" const char* cic_last_error(void) { return g_err; }

// ---- in-situ kernel timing: a caller-owned object (no library state) ----------------------------------------
// The engines bracket the launches of a few kernels with a pair of HIP events on their own stream when the caller
// hands them a cic_timer (cic_decode_io.timer); the events live in that object.
#include <vector>
struct cic_timer {
    struct Pair { hipEvent_t a, b; int id; };
    std::vector<Pair> pairs;          // recorded since the last reset
    std::vector<hipEvent_t> pool;     // recycled events
    hipEvent_t event() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};

void* cic_timer_begin(cic_timer* t, int id, hipStream_t st) {
    if (!t) return nullptr;
    cic_timer::Pair p{t->event(), t->event(), id};
    if (!p.a || !p.b) return nullptr;
    (void)hipEventRecord(p.a, st);
    t->pairs.push_back(p);
    return p.b;
}
void cic_timer_end(void* h, hipStream_t st) {
    if (h) (void)hipEventRecord(static_cast<hipEvent_t>(h), st);
}
extern "C" cic_timer* cic_timer_create(void) { return new cic_timer(); }
extern "C" void cic_timer_destroy(cic_timer* t) {
    if (!t) return;
    for (auto& p : t->pairs) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : t->pool) (void)hipEventDestroy(e);
    delete t;
}
extern "C" int cic_timer_reset(cic_timer* t) {
    CIC_REQUIRE(t);
    for (auto& p : t->pairs) { t->pool.push_back(p.a); t->pool.push_back(p.b); }
    t->pairs.clear();
    return 0;
}
// What a bracket costs by itself: the average elapsed time of `pairs` event pairs recorded back to back on `s` with
// nothing between them (the command processor's own time for the two event packets).  A bracketed launch's duration is
// its elapsed time minus this.
extern "C" int cic_timer_bracket_overhead(cic_timer* t, int pairs, double* avg_us, cic_stream_t s) {
    CIC_REQUIRE(t && pairs > 0 && avg_us);
    hipStream_t st = cic_s(s);
    std::vector<hipEvent_t> ev(2 * pairs);
    for (auto& e : ev) { e = t->event(); CIC_REQUIRE(e != nullptr); }
    CIC_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < pairs; ++i) {
        CIC_HIP(hipEventRecord(ev[2 * i], st));
        CIC_HIP(hipEventRecord(ev[2 * i + 1], st));
    }
    CIC_HIP(hipEventSynchronize(ev.back()));
    double tot = 0.0;
    for (int i = 0; i < pairs; ++i) {
        float ms = 0.f;
        CIC_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
        tot += ms;
    }
    for (auto e : ev) t->pool.push_back(e);
    *avg_us = tot * 1e3 / pairs;
    return 0;
}
extern "C" int cic_timer_collect(cic_timer* t, int id, double* total_ms, int* launches) {
    CIC_REQUIRE(t && total_ms && launches && id >= 0 && id < CIC_TIMED_COUNT);
    double tot = 0.0;
    int n = 0;
    for (auto& p : t->pairs) {
        if (p.id != id) continue;
        CIC_HIP(hipEventSynchronize(p.b));
        float ms = 0.f;
        CIC_HIP(hipEventElapsedTime(&ms, p.a, p.b));
        tot += ms;
        ++n;
    }
    *total_ms = tot;
    *launches = n;
    return 0;
}

namespace {

__global__ __launch_bounds__(256) void uniform_kernel(float* __restrict__ out, int64_t n, uint64_t seed,
                                                      uint64_t offset) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one Philox call = 4 outputs
    const int64_t i = q * 4;
    if (i >= n) return;
    Philox4 r = philox4x32_10(offset + (uint64_t)q, seed);
    if (i + 3 < n && ((reinterpret_cast<uintptr_t>(out + i) & 15) == 0)) {
        f32x4 v = {u32_to_unit(r.v[0]), u32_to_unit(r.v[1]), u32_to_unit(r.v[2]), u32_to_unit(r.v[3])};
        *reinterpret_cast<f32x4*>(out + i) = v;
    } else {
        for (int j = 0; j < 4 && i + j < n; ++j) out[i + j] = u32_to_unit(r.v[j]);
    }
}

__global__ __launch_bounds__(256) void keep_kernel(uint8_t* __restrict__ keep, int64_t n, float p, uint64_t seed,
                                                   uint64_t offset) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = q * 4;
    if (i >= n) return;
    Philox4 r = philox4x32_10(offset + (uint64_t)q, seed);
    for (int j = 0; j < 4 && i + j < n; ++j) keep[i + j] = u32_to_unit(r.v[j]) >= p ? 1 : 0;
}

struct KeepSegs {
    uint8_t* keep[CIC_KEEP_MAX_SEGMENTS];
    int64_t n[CIC_KEEP_MAX_SEGMENTS];
    uint64_t offset[CIC_KEEP_MAX_SEGMENTS];
    int first_block[CIC_KEEP_MAX_SEGMENTS + 1];     // blocks of segment i: [first_block[i], first_block[i + 1])
    int count;
};

__global__ __launch_bounds__(256) void keep_multi_kernel(KeepSegs sg, float p, uint64_t seed) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < CIC_KEEP_MAX_SEGMENTS; ++k)
        if (k < sg.count && (int)blockIdx.x >= sg.first_block[k]) i = k;
    const int64_t q = (int64_t)(blockIdx.x - sg.first_block[i]) * blockDim.x + threadIdx.x;
    const int64_t e = q * 4, n = sg.n[i];
    if (e >= n) return;
    uint8_t* keep = sg.keep[i];
    Philox4 r = philox4x32_10(sg.offset[i] + (uint64_t)q, seed);
    for (int j = 0; j < 4 && e + j < n; ++j) keep[e + j] = u32_to_unit(r.v[j]) >= p ? 1 : 0;
}

}  // namespace

extern "C" int cic_uniform_f32(float* out, int64_t n, uint64_t seed, uint64_t offset, cic_stream_t s) {
    CIC_REQUIRE(out && n > 0);
    const int64_t q = (n + 3) / 4;
    hipLaunchKernelGGL(uniform_kernel, dim3(cic_cdiv(q, 256)), dim3(256), 0, cic_s(s), out, n, seed, offset);
    CIC_LAUNCH_CHECK();
    return 0;
}

extern "C" int cic_dropout_keep_u8_multi(uint8_t* const* keep, const int64_t* n, const uint64_t* offset, int count,
                                         float p, uint64_t seed, cic_stream_t s) {
    CIC_REQUIRE(keep && n && offset && count >= 1 && count <= CIC_KEEP_MAX_SEGMENTS && p >= 0.f && p < 1.f);
    KeepSegs sg = {};
    sg.count = count;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        CIC_REQUIRE(keep[i] && n[i] > 0);
        sg.keep[i] = keep[i]; sg.n[i] = n[i]; sg.offset[i] = offset[i]; sg.first_block[i] = blocks;
        blocks += (int)cic_cdiv((n[i] + 3) / 4, 256);
    }
    sg.first_block[count] = blocks;
    hipLaunchKernelGGL(keep_multi_kernel, dim3(blocks), dim3(256), 0, cic_s(s), sg, p, seed);
    CIC_LAUNCH_CHECK();
    return 0;
}

extern "C" int cic_dropout_keep_u8(uint8_t* keep, int64_t n, float p, uint64_t seed, uint64_t offset,
                                   cic_stream_t s) {
    CIC_REQUIRE(keep && n > 0 && p >= 0.f && p < 1.f);
    const int64_t q = (n + 3) / 4;
    hipLaunchKernelGGL(keep_kernel, dim3(cic_cdiv(q, 256)), dim3(256), 0, cic_s(s), keep, n, p, seed, offset);
    CIC_LAUNCH_CHECK();
    return 0;
}
