// error string + device probe for libwise_hip.so
#include "common.h"
#include <string.h>

namespace wise {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct ProfSlot { hipEvent_t a, b; int cls; double work; };
static ProfSlot* g_prof = nullptr;
static int g_prof_cap = 0, g_prof_n = 0;

int prof_open(int cls, double work, hipStream_t st) {
    if (!g_prof || g_prof_n >= g_prof_cap) return -1;
    const int s = g_prof_n++;
    g_prof[s].cls = cls;
    g_prof[s].work = work;
    (void)hipEventRecord(g_prof[s].a, st);
    return s;
}
void prof_close(int slot, hipStream_t st) { (void)hipEventRecord(g_prof[slot].b, st); }
}  // namespace wise

extern "C" int wise_prof_begin(int capacity) {
    using namespace wise;
    if (g_prof) { set_error("prof_begin: already profiling"); return WISE_E_INVALID; }
    if (capacity < 1) { set_error("prof_begin: capacity=%d", capacity); return WISE_E_INVALID; }
    g_prof = new ProfSlot[capacity];
    for (int i = 0; i < capacity; ++i) {
        hipError_t e = hipEventCreate(&g_prof[i].a);
        if (e == hipSuccess) e = hipEventCreate(&g_prof[i].b);
        if (e != hipSuccess) { set_error("prof_begin: %s", hipGetErrorString(e)); return (int)e; }
    }
    g_prof_cap = capacity;
    g_prof_n = 0;
    return WISE_OK;
}

extern "C" int wise_prof_end(double* ms_sum, int64_t* launches, double* work_sum) {
    using namespace wise;
    if (!g_prof) { set_error("prof_end: not profiling"); return WISE_E_INVALID; }
    for (int c = 0; c < PROF_CLASSES; ++c) { ms_sum[c] = 0; launches[c] = 0; work_sum[c] = 0; }
    int rc = WISE_OK;
    for (int i = 0; i < g_prof_n; ++i) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(g_prof[i].b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, g_prof[i].a, g_prof[i].b);
        if (e != hipSuccess) { set_error("prof_end: %s", hipGetErrorString(e)); rc = (int)e; break; }
        ms_sum[g_prof[i].cls] += ms;
        launches[g_prof[i].cls] += 1;
        work_sum[g_prof[i].cls] += g_prof[i].work;
    }
    for (int i = 0; i < g_prof_cap; ++i) { (void)hipEventDestroy(g_prof[i].a); (void)hipEventDestroy(g_prof[i].b); }
    delete[] g_prof;
    g_prof = nullptr;
    g_prof_cap = g_prof_n = 0;
    return rc;
}

extern "C" const char* wise_last_error(void) { return wise::g_err; }
extern "C" int wise_abi_version(void) { return 5; }
#ifndef WISE_BUILD_FLAGS
#define WISE_BUILD_FLAGS "unknown (not built by wise_amd/build.py)"
#endif
extern "C" const char* wise_build_flags(void) { return WISE_BUILD_FLAGS; }
extern "C" int wise_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
