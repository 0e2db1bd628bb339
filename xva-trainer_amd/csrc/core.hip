// core.hip — error reporting and library identification for libxvahip.so.
#include "xva_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void xva_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* xva_last_error(void) { return g_err; }
extern "C" int xva_abi_version(void) { return 1; }
extern "C" const char* xva_target_arch(void) { return "gfx950"; }
