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

// ---- in-situ kernel timing -------------------------------------------------------------------
#include <vector>
namespace {
struct ProfPair { hipEvent_t a, b; int id; };
bool g_prof_on = false;
std::vector<ProfPair> g_prof_pairs;      // recorded since the last reset
std::vector<hipEvent_t> g_prof_pool;     // recycled events
hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

bool cic_prof_on() { return g_prof_on; }
void* cic_prof_begin(int id, hipStream_t st) {
    if (!g_prof_on) return nullptr;
    ProfPair p{prof_event(), prof_event(), id};
    if (!p.a || !p.b) return nullptr;
    (void)hipEventRecord(p.a, st);
    g_prof_pairs.push_back(p);
    return p.b;
}
void cic_prof_end(void* h, hipStream_t st) {
    if (h) (void)hipEventRecord(static_cast<hipEvent_t>(h), st);
}
extern "C" int cic_prof_enable(int on) { g_prof_on = on != 0; return 0; }
extern "C" int cic_prof_reset(void) {
    for (auto& p : g_prof_pairs) { g_prof_pool.push_back(p.a); g_prof_pool.push_back(p.b); }
    g_prof_pairs.clear();
    return 0;
}
extern "C" int cic_prof_collect(int id, double* total_ms, int* launches) {
    CIC_REQUIRE(total_ms && launches && id >= 0 && id < CIC_PROF_COUNT);
    double tot = 0.0;
    int n = 0;
    for (auto& p : g_prof_pairs) {
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

// ---- HIP graph cache ---------------------------------------------------------------------------------
#include <unordered_map>
namespace {
bool g_graph_on = false;
std::unordered_map<uint64_t, hipGraphExec_t> g_graphs;
int64_t g_graph_stats[3] = {0, 0, 0};   // captures, replays, fallbacks
}  // namespace

uint64_t cic_hash_bytes(const void* p, size_t n, uint64_t h) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }   // FNV-1a
    return h;
}

CicGraphScope::CicGraphScope(hipStream_t s, uint64_t k) : st(s), key(k) {
    if (!g_graph_on || g_prof_on) return;
    auto it = g_graphs.find(key);
    if (it != g_graphs.end()) {
        if (hipGraphLaunch(it->second, st) == hipSuccess) {
            replayed = true;
            ++g_graph_stats[1];
            return;
        }
        (void)hipGetLastError();
    }
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        capturing = true;
    } else {
        (void)hipGetLastError();   // e.g. the legacy default stream: run the launches directly
        ++g_graph_stats[2];
    }
}

int CicGraphScope::finish(int rc) {
    if (!capturing) return rc;
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(st, &graph);
    capturing = false;
    if (e != hipSuccess || graph == nullptr) {
        (void)hipGetLastError();
        cic_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
        return rc ? rc : 2;
    }
    if (rc != 0) {   // an engine error during capture: nothing was launched
        (void)hipGraphDestroy(graph);
        return rc;
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        cic_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return 2;
    }
    if (g_graphs.size() >= 64) {   // bounded cache: drop everything rather than track recency
        for (auto& kv : g_graphs) (void)hipGraphExecDestroy(kv.second);
        g_graphs.clear();
    }
    g_graphs[key] = exec;
    ++g_graph_stats[0];
    CIC_HIP(hipGraphLaunch(exec, st));
    return 0;
}

// ---- side stream: fork / join inside one engine call ----------------------------------------------------
// A latency-bound launch chain (the BPTT loop: ~35 us per step on a fraction of the CUs) leaves most of the chip
// idle; a product that does not depend on the chain (the logit layer's weight gradient) runs beside it on a second,
// non-blocking HIP stream.  fork: side waits for everything `main` has queued so far; join: main waits for the side
// work.  Both are event waits on the device - the host never blocks.
static hipStream_t g_side = nullptr;
static hipEvent_t g_side_fork = nullptr, g_side_join = nullptr;
static int g_side_on = 0;   // measured: ON 6.08 ms/step vs OFF 5.81 (the GEMM's workgroups hold the CUs the chain's short
                            // kernels need; their dispatch then waits for whole 25-100 us tiles): off by default
extern "C" int cic_debug_side_stream(int on) { g_side_on = on; return 0; }

int cic_side_fork(hipStream_t main, hipStream_t* side) {
    *side = main;
    if (!g_side_on) return 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return 0;   // graphs: one stream
    if (!g_side) {
        CIC_HIP(hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking));
        CIC_HIP(hipEventCreateWithFlags(&g_side_fork, hipEventDisableTiming));
        CIC_HIP(hipEventCreateWithFlags(&g_side_join, hipEventDisableTiming));
    }
    CIC_HIP(hipEventRecord(g_side_fork, main));
    CIC_HIP(hipStreamWaitEvent(g_side, g_side_fork, 0));
    *side = g_side;
    return 0;
}

int cic_side_join(hipStream_t main, hipStream_t side) {
    if (side == main) return 0;
    CIC_HIP(hipEventRecord(g_side_join, side));
    CIC_HIP(hipStreamWaitEvent(main, g_side_join, 0));
    return 0;
}

extern "C" int cic_graph_enable(int on) { g_graph_on = on != 0; return 0; }
extern "C" int cic_graph_clear(void) {
    for (auto& kv : g_graphs) (void)hipGraphExecDestroy(kv.second);
    g_graphs.clear();
    return 0;
}
extern "C" int cic_graph_stats(int64_t* out3) {
    CIC_REQUIRE(out3);
    for (int i = 0; i < 3; ++i) out3[i] = g_graph_stats[i];
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

}  // namespace

extern "C" int cic_uniform_f32(float* out, int64_t n, uint64_t seed, uint64_t offset, cic_stream_t s) {
    CIC_REQUIRE(out && n > 0);
    const int64_t q = (n + 3) / 4;
    hipLaunchKernelGGL(uniform_kernel, dim3(cic_cdiv(q, 256)), dim3(256), 0, cic_s(s), out, n, seed, offset);
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
