// Shared host/device helpers for libtt.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tt.h"

#define TT_EXPORT extern "C" __attribute__((visibility("default")))

int tt_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define TT_HIP_CHECK(expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return tt_fail(TT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                              \
    } while (0)

#define TT_LAUNCH_CHECK() TT_HIP_CHECK(hipGetLastError())

#define TT_RC_CHECK(expr)       \
    do {                        \
        const int rc_ = (expr); \
        if (rc_ != TT_OK)       \
            return rc_;         \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline size_t tt_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Zero `bytes` bytes (multiple of 4, 4-byte aligned) at p with a KERNEL.  Never hipMemsetAsync on a path a caller may
// capture into a HIP graph: on ROCm 7.2 a captured memset node of a larger graph fills with garbage from the second
// replay on (a repeating 16-byte pattern that looks like two kernel-argument pointers: the node's pattern staging is
// recycled) -- observed in torch.cuda.graph captures of the encoder forward and of a screened search
// (tools/experiments/encoder_graph_flags.py; small stand-alone graphs do not show it: memset_graph.hip).
int tt_zero_async(void *p, size_t bytes, hipStream_t st);
int tt_zero3_async(void *p0, size_t b0, void *p1, size_t b1, void *p2, size_t b2, hipStream_t st);

// MUTATION SWITCH, never set in the product build (tools/mutation_guard.py builds the variants): a bit mask of f16-split
// kernels whose `lo` products (hi*lo and lo*hi) are compiled out, which turns "fp32-grade" into plain fp16 (2^-11 per
// product).  The parity tests' tolerances must be tight enough to FAIL on every one of them.
//   1 = K2 gru_seq16 (forward recurrence)   2 = K7 gru_bwd16 (backward recurrence)
//   4 = K1 gemm_rows16 (input projection)   8 = sgemm16 (input gradients, tiled K1, tiled weight gradients)
//  16 = wgrad16 (the weight gradients dW_ih / dW_hh of the training step)
#ifndef TT_MUTATE_DROP_LO
#define TT_MUTATE_DROP_LO 0
#endif

// A/B SWITCHES.  The product library reads NO environment variable on any call path: every switch below is the compile-time
// constant `dflt` there.  A comparison build (-DTT_AB: tools/build_variant.py ab -> ab/libtt_ab.so, loaded by the tests that
// pin a product kernel against the kernel it replaced, and by tools/experiments) reads the variable of the same name at EVERY
// call, so that one process can run both forms; the superseded kernels themselves are compiled only into that build.
#ifdef TT_AB
#include <stdlib.h>
static inline int tt_ab_env(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
#define TT_AB_SWITCH(name, dflt) tt_ab_env(#name, (dflt))
#else
#define TT_AB_SWITCH(name, dflt) (dflt)
#endif

#define TT_WAVE 64
