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
int sh_version(void) { return 100; }   // major*10000 + minor*100 + patch
const char* sh_last_error(void) { return g_err; }
}
