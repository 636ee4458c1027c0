/*
 * tt_debug.h -- TEST-ONLY exports of libtt.so.  Not part of the drop-in boundary (include/tt.h), not bound by the
 * Python package; tests/ bind them directly through ctypes.  Same conventions as tt.h (device pointers, caller-owned
 * workspace, asynchronous on `stream`, int status).
 */
#ifndef TT_DEBUG_H
#define TT_DEBUG_H

#include "tt.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Observe the approximate scores of the screened search (tt_score_topk_screened_f32, replacing
 * backend/evaluators.py:185-186) as the REAL screen kernels compute them, so that the error bound the filter rests on
 *     |s16 - s| <= eps_q = 1.10e-3 |q| Dmax + 1e-6 (|q| + Dmax)          (csrc/screen.hip: screen_eps)
 * can be checked on the hardware (tests/test_screen_bound_gpu.py).  Runs q_image_kernel and the MAXONLY form of
 *   form 0: screen_stream_kernel with two query sets (B <= 32 in the product; 33..64 run it with four),  form 1 / 2 / 4: screen_kernel<., NSET = form>
 * over the whole fp16 corpus D16 [N,256] and writes out_t [B][ceil(N/32)]: per (query, 32-document tile) the maximum
 * over the tile of
 *   thr == NULL : s16 = sum_i fp16(q_i) fp16(d_i), accumulators starting at +0 (the sample pass)
 *   thr != NULL : t = fl(s16 - thr[query]), accumulators starting at -thr[query] (the main pass; the filter keeps a
 *                 document iff t >= +0 and reasons about v = t + thr)
 * A corpus whose tiles each hold 32 copies of one document yields that document's value.  thr: device float [B].
 */
size_t tt_debug_screen_s16_workspace_bytes(int B, int64_t N, int form);
int tt_debug_screen_s16(const float *Q, int B, const void *D16, int64_t N, float dmax_norm, const float *thr, int form,
                        float *out_t, void *workspace, size_t workspace_bytes, tt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TT_DEBUG_H */
