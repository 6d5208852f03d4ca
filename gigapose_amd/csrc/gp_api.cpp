// Error reporting + version for the C-ABI (include/gigapose_hip.h).
#include <cstdarg>
#include <cstdio>

#include "gp_common.h"

static thread_local char g_err[512] = "";

void gp_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
const char* gp_last_error(void) { return g_err; }
int gp_abi_version(void) { return 1; }
}
