// Error plumbing and version of libllada_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/llada_hip.h"

static thread_local char g_err[512] = "";

void llada_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* llada_last_error(void) { return g_err; }
extern "C" int llada_abi_version(void) { return LLADA_ABI_VERSION; }
