// Error reporting and version for the C ABI (include/nlc_hip.h).
#include "common.h"

static thread_local char g_err[512] = "";

void nlc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* nlc_last_error(void) { return g_err; }
extern "C" int nlc_version(void) { return NLC_ABI_VERSION; }
