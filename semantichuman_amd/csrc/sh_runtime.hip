// Library-level plumbing of libsh_kernels.so: version and the thread-local error message.
#include <stdarg.h>
#include <stdlib.h>
#include "sh_common.h"

namespace {
thread_local char g_err[512] = "";
}

void sh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int sh_env_int(const char* name, int dflt, int lo, int hi) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    int x = atoi(v);
    return x < lo ? lo : (x > hi ? hi : x);
}

// ---- kernel timing -----------------------------------------------------------------------
#include <mutex>
#include <string>
#include <vector>
namespace {
struct ProfRec { std::string name; hipEvent_t a, b; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
bool g_prof_on = false;
}
bool sh_profile_on() { return g_prof_on; }
void sh_profile_push(const char* name, hipEvent_t a, hipEvent_t b) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back({name, a, b});
}
ShProfScope::ShProfScope(hipStream_t s, const char* fmt, ...) : st(s), a(nullptr), b(nullptr), on(g_prof_on), ext(false) {
    if (!on) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof(name), fmt, ap);
    va_end(ap);
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, st);
}
ShProfScope::~ShProfScope() {
    if (!on) return;
    if (!ext) (void)hipEventRecord(b, st);        // scopes whose launches carry the events themselves need no record
    sh_profile_push(name, a, b);
}

extern "C" {
int sh_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (on) {
        for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        g_prof.clear();
    }
    g_prof_on = on != 0;
    return SH_OK;
}
int sh_profile_count(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    return (int)g_prof.size();
}
int sh_profile_get(int i, char* name, int name_len, float* ms) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (i < 0 || i >= (int)g_prof.size() || !name || !ms || name_len <= 0) return SH_ERR_INVALID_ARG;
    snprintf(name, name_len, "%s", g_prof[i].name.c_str());
    if (hipEventSynchronize(g_prof[i].b) != hipSuccess) return SH_ERR_LAUNCH;
    if (hipEventElapsedTime(ms, g_prof[i].a, g_prof[i].b) != hipSuccess) return SH_ERR_LAUNCH;
    return SH_OK;
}
// ---- shader-clock probe (diagnostic) ------------------------------------------------------
// One wavefront per workgroup stamps the shader-cycle counter and the 100 MHz wall counter around a fixed loop of
// dependent fp32 MFMAs; out[2*wg] = shader cycles, out[2*wg+1] = 100 MHz ticks.  clock = cycles / ticks * 100 MHz.
// Launched between the steps of a running workload it reads the clock the chip currently grants that workload.
}  // extern "C"
namespace {
typedef float sh_probe_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, int iters) {
    sh_probe_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    if (acc[0] == 12345.678f) out[0] = 0;                        // keep the loop
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}
}  // namespace
extern "C" {
int sh_clock_probe(unsigned long long* out, int n_workgroups, int iters, sh_stream_t stream) {
    SH_REQUIRE(out && n_workgroups > 0 && iters > 0, SH_ERR_INVALID_ARG, "sh_clock_probe: bad argument");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(n_workgroups), dim3(64), 0, static_cast<hipStream_t>(stream), out, iters);
    SH_CHECK_LAUNCH("sh_clock_probe");
    return SH_OK;
}
// ---- arithmetic form of the fp32 path's matrix products: per call, pinned on the calling thread by the entry point
}
namespace {
thread_local int tl_mma_mode = SH_MMA_EXACT;
}
int sh_f32_mma_mode() { return tl_mma_mode; }
bool sh_mma_mode_valid(int mode) { return mode == SH_MMA_EXACT || mode == SH_MMA_SPLIT3 || mode == SH_MMA_PLANES3; }
ShMmaScope::ShMmaScope(int mode) : was(tl_mma_mode) { tl_mma_mode = mode; }
ShMmaScope::~ShMmaScope() { tl_mma_mode = was; }
extern "C" {
#ifndef SH_BUILD_ID
#define SH_BUILD_ID "unknown"
#endif
const char* sh_build_id(void) { return SH_BUILD_ID; }
int sh_version(void) { return 100; }   // major*10000 + minor*100 + patch
const char* sh_last_error(void) { return g_err; }
}
