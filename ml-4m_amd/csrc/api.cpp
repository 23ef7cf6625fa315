// Error string + version for the C ABI.
#include <stdarg.h>
#include <stdio.h>
#include "fourm_hip.h"

static thread_local char g_err[512] = "";

void fm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* fm_last_error(void) { return g_err; }
extern "C" int fm_abi_version(void) { return FM_ABI_VERSION; }
