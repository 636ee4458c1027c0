#include "tt_common.h"

static thread_local char g_err[512] = "";

int tt_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

TT_EXPORT const char *tt_version(void) { return "tt 0.1.0 (gfx950)"; }
TT_EXPORT const char *tt_last_error(void) { return g_err; }
