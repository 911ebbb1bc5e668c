// Library-wide pieces of the C ABI: version and per-thread error text.
#include <cstdarg>
#include <cstdio>

#include "vo_common.h"

namespace {
thread_local char g_last_error[512] = "";
}

int vo_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int vo_version(void) { return VO_HIP_VERSION; }

extern "C" const char* vo_last_error(void) { return g_last_error; }
