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

TT_EXPORT const char *tt_version(void) { return "tt 0.2.0 (gfx950)"; }
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

namespace {
__global__ __launch_bounds__(256) void zero_kernel(unsigned *__restrict__ p, size_t n_words)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if ((((uintptr_t)p) & 15) == 0) { // 16-byte stores over the aligned body
        uint4 *q = (uint4 *)p;
        const size_t n4 = n_words / 4;
        for (size_t k = i; k < n4; k += stride)
            q[k] = make_uint4(0u, 0u, 0u, 0u);
        for (size_t k = n4 * 4 + i; k < n_words; k += stride)
            p[k] = 0u;
    } else {
        for (; i < n_words; i += stride)
            p[i] = 0u;
    }
}
struct ZeroRegions {
    unsigned *p[3];
    size_t n_words[3];
};
__global__ __launch_bounds__(256) void zero3_kernel(ZeroRegions z)
{
    const size_t stride = (size_t)gridDim.x * 256;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        unsigned *p = z.p[r];
        const size_t n_words = z.n_words[r];
        if ((((uintptr_t)p) & 15) == 0) {
            uint4 *q = (uint4 *)p;
            const size_t n4 = n_words / 4;
            for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += stride)
                q[k] = make_uint4(0u, 0u, 0u, 0u);
            for (size_t k = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; k < n_words; k += stride)
                p[k] = 0u;
        } else {
            for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n_words; k += stride)
                p[k] = 0u;
        }
    }
}
} // namespace

// up to three regions in ONE launch (a region with 0 bytes is skipped): a ~5 us launch each otherwise
int tt_zero3_async(void *p0, size_t b0, void *p1, size_t b1, void *p2, size_t b2, hipStream_t st)
{
    void *ps[3] = {p0, p1, p2};
    const size_t bs[3] = {b0, b1, b2};
    ZeroRegions z;
    size_t most = 0;
    for (int r = 0; r < 3; ++r) {
        if (bs[r] && (!ps[r] || (bs[r] & 3) || ((uintptr_t)ps[r] & 3)))
            return tt_fail(TT_ERR_BAD_SHAPE, "tt_zero3_async: %zu bytes at %p (need 4-byte alignment)", bs[r], ps[r]);
        z.p[r] = (unsigned *)ps[r];
        z.n_words[r] = bs[r] / 4;
        most = bs[r] / 4 > most ? bs[r] / 4 : most;
    }
    if (most == 0)
        return TT_OK;
    const size_t want = (most / 4 + 255) / 256 + 1;
    hipLaunchKernelGGL(zero3_kernel, dim3((unsigned)(want > 4096 ? 4096 : want)), dim3(256), 0, st, z);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

int tt_zero_async(void *p, size_t bytes, hipStream_t st)
{
    if (bytes == 0)
        return TT_OK;
    if (!p || (bytes & 3) || ((uintptr_t)p & 3))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_zero_async: %zu bytes at %p (need 4-byte alignment)", bytes, p);
    const size_t words = bytes / 4, want = (words / 4 + 255) / 256 + 1;
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)(want > 4096 ? 4096 : want)), dim3(256), 0, st, (unsigned *)p, words);
    TT_LAUNCH_CHECK();
    return TT_OK;
}
