// Error plumbing shared by every entry point of libgsseg_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "gsseg.h"

static thread_local char g_err[512] = "";

void gs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gs_last_error(void) { return g_err; }
extern "C" int gs_abi_version(void) { return GS_ABI_VERSION; }
