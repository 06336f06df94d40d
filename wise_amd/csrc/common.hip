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
}  // namespace wise

extern "C" const char* wise_last_error(void) { return wise::g_err; }
extern "C" int wise_abi_version(void) { return 1; }
extern "C" int wise_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
