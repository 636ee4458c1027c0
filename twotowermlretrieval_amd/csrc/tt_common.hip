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

// Thin hipEvent wrappers so a host without HIP headers (the Python shim, bench.py) can time a single
// kernel inside a multi-launch entry point (prof_events arguments).
TT_EXPORT int tt_event_create(void **ev)
{
    hipEvent_t e;
    TT_HIP_CHECK(hipEventCreate(&e));
    *ev = (void *)e;
    return TT_OK;
}
TT_EXPORT int tt_event_destroy(void *ev)
{
    TT_HIP_CHECK(hipEventDestroy((hipEvent_t)ev));
    return TT_OK;
}
TT_EXPORT int tt_event_elapsed_ms(void *start, void *stop, float *ms) /* blocks until `stop` has happened */
{
    TT_HIP_CHECK(hipEventSynchronize((hipEvent_t)stop));
    TT_HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return TT_OK;
}
