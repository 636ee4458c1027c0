// K6 triplet loss (forward + gradient) and K8 fused clip_grad_norm_ + Adam (gfx950).
#include "tt_common.h"

namespace {

// ------------------------------------------------------------------ K6
// triplet_loss_cosine, backend/model.py:109-114:
//   loss = mean_b max(0, cos(q,n) - cos(q,p) + margin), F.cosine_similarity eps 1e-8 (each norm
//   clamped separately, as ATen does).  One wave per row; gradients are written already scaled by
//   1/B so autograd only multiplies by the incoming scalar.  clamp(min=0) passes gradient where its
//   argument is >= 0 (ATen's mask).
__global__ __launch_bounds__(256) void triplet_rows_kernel(const float *__restrict__ q, const float *__restrict__ p,
                                                           const float *__restrict__ n, int B, int H, float margin,
                                                           float *__restrict__ row_loss, float *__restrict__ dq,
                                                           float *__restrict__ dp, float *__restrict__ dn)
{
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B)
        return;
    const float *qb = q + (size_t)b * H, *pb = p + (size_t)b * H, *nb = n + (size_t)b * H;
    float sq = 0, sp = 0, sn = 0;
    for (int u = lane; u < H; u += 64) {
        sq += qb[u] * qb[u];
        sp += pb[u] * pb[u];
        sn += nb[u] * nb[u];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        sq += __shfl_xor(sq, off);
        sp += __shfl_xor(sp, off);
        sn += __shfl_xor(sn, off);
    }
    const float nq = fmaxf(sqrtf(sq), 1e-8f), np_ = fmaxf(sqrtf(sp), 1e-8f), nn = fmaxf(sqrtf(sn), 1e-8f);
    float cp = 0, cn = 0;
    for (int u = lane; u < H; u += 64) {
        const float qh = qb[u] / nq;
        cp += qh * (pb[u] / np_);
        cn += qh * (nb[u] / nn);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        cp += __shfl_xor(cp, off);
        cn += __shfl_xor(cn, off);
    }
    const float v = cn - cp + margin;
    if (lane == 0)
        row_loss[b] = v > 0.0f ? v : 0.0f;
    if (dq) {
        const float g = (v >= 0.0f ? 1.0f : 0.0f) / (float)B;
        for (int u = lane; u < H; u += 64) {
            const float qh = qb[u] / nq, ph = pb[u] / np_, nh = nb[u] / nn;
            dq[(size_t)b * H + u] = g * ((nh - cn * qh) / nq - (ph - cp * qh) / nq);
            dp[(size_t)b * H + u] = -g * (qh - cp * ph) / np_;
            dn[(size_t)b * H + u] = g * (qh - cn * nh) / nn;
        }
    }
}

// Fixed-order sum of x[0..n) by one block -> out[0] * scale (deterministic).
__global__ __launch_bounds__(1024) void sum_fixed_kernel(const float *__restrict__ x, int n, float scale,
                                                         float *__restrict__ out)
{
    __shared__ float red[16];
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 1024)
        s += x[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int w = 0; w < 16; ++w)
            t += red[w];
        out[0] = t * scale;
    }
}

// ------------------------------------------------------------------ K8
constexpr int ADAM_BLOCKS = 512;

// The step is applied iff every word of the (all-reduced) gate is zero; a NaN closes it too.
__device__ __forceinline__ bool gate_open(const float *gate)
{
    return !gate || (gate[0] == 0.0f && gate[1] == 0.0f && gate[2] == 0.0f && gate[3] == 0.0f);
}

// pass 1: per-block partial sum of (g*grad_scale)^2 in fp64 -> partial[ADAM_BLOCKS]; the gated form also counts the step
// (one thread; pass 2 of this step starts after every block of this launch, pass 2 of the previous step ended before it)
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float *__restrict__ g, int64_t n, float grad_scale,
                                                             double *__restrict__ partial, int64_t *step_counter,
                                                             const float *gate)
{
    __shared__ double red[4];
    if (step_counter && blockIdx.x == 0 && threadIdx.x == 0 && gate_open(gate))
        *step_counter += 1;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = (double)(g[i] * grad_scale);
        s += v * v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// pass 2: every block re-derives the total norm from the partials in the same fixed order, then
// applies clip + Adam to its slice.  clip_grad_norm_: coef = min(1, max_norm / (total + 1e-6)).
// Adam (torch defaults wd=0, amsgrad=False): m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2;
// p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps).   backend/main.py:222,257,259
__global__ __launch_bounds__(256) void clip_adam_kernel(float *__restrict__ p, float *__restrict__ g,
                                                        float *__restrict__ m, float *__restrict__ v, int64_t n,
                                                        float grad_scale, float max_norm, float lr, float b1,
                                                        float b2, float eps, float bc1, float bc2_sqrt,
                                                        const double *__restrict__ partial, int npartial,
                                                        float *__restrict__ total_norm_out,
                                                        const int64_t *__restrict__ step_counter,
                                                        const float *__restrict__ gate)
{
    __shared__ float s_coef, s_bc1, s_bc2_sqrt;
    __shared__ int s_open;
    __shared__ double s_red[4];
    {
        // every block sums the partials in the SAME fixed order (thread t: t, t + 256, ...; xor tree; the four waves in
        // order), so all of them clip by bit-identical coefficients; one thread walking all of them took 30 us per step
        double s = 0.0;
        for (int i = threadIdx.x; i < npartial; i += 256)
            s += partial[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
            s += __shfl_xor(s, off);
        if ((threadIdx.x & 63) == 0)
            s_red[threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ss = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
        const float total = (float)sqrt(ss);
        float coef = 1.0f;
        if (max_norm > 0.0f) {
            coef = max_norm / (total + 1e-6f);
            coef = coef > 1.0f ? 1.0f : coef;
        }
        s_coef = coef * grad_scale;
        if (blockIdx.x == 0 && total_norm_out)
            *total_norm_out = total;
        s_open = gate_open(gate) ? 1 : 0;
        if (step_counter) { // the gated form: step number from the device counter (already counted by pass 1)
            const double t = (double)*step_counter;
            bc1 = (float)(1.0 - pow((double)b1, t));
            bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
        }
        s_bc1 = bc1;
        s_bc2_sqrt = bc2_sqrt;
    }
    __syncthreads();
    if (!s_open) // some rank's batch was bad (or its recurrence gave up): no rank applies this step
        return;
    bc1 = s_bc1;
    bc2_sqrt = s_bc2_sqrt;
    const float coef = s_coef;
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * coef;
        g[i] = gi; // the reference leaves clipped gradients in .grad
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}

} // namespace

TT_EXPORT int tt_triplet_loss_f32(const float *q, const float *p, const float *n, int B, int H, float margin,
                                  float *loss, float *dq, float *dp, float *dn, float *scratch_rows,
                                  tt_stream_t stream)
{
    if (B <= 0 || H <= 0)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_triplet_loss_f32: B=%d H=%d", B, H);
    if (!q || !p || !n || !loss || !scratch_rows || ((dq || dp || dn) && !(dq && dp && dn)))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_triplet_loss_f32: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(triplet_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, st, q, p, n, B, H, margin, scratch_rows,
                       dq, dp, dn);
    hipLaunchKernelGGL(sum_fixed_kernel, dim3(1), dim3(1024), 0, st, (const float *)scratch_rows, B, 1.0f / (float)B,
                       loss);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

TT_EXPORT size_t tt_clip_adam_scratch_bytes(void) { return sizeof(double) * ADAM_BLOCKS + 256; }

TT_EXPORT int tt_clip_adam_step_f32(float *params, float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                                    int64_t step, float lr, float beta1, float beta2, float eps, float max_norm,
                                    float grad_scale, float *total_norm_out, void *scratch, tt_stream_t stream)
{
    if (n < 0 || step < 1)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_clip_adam_step_f32: n=%lld step=%lld", (long long)n, (long long)step);
    if (n == 0)
        return TT_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !scratch)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_clip_adam_step_f32: null pointer");
    hipStream_t st = (hipStream_t)stream;
    int blocks = (int)((n + 255) / 256);
    blocks = blocks > ADAM_BLOCKS ? ADAM_BLOCKS : blocks;
    double *partial = (double *)scratch;
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(blocks), dim3(256), 0, st, (const float *)grads, n, grad_scale,
                       partial, (int64_t *)nullptr, (const float *)nullptr);
    // (in double, as torch.optim.Adam computes them on the host; the gated form does the same on the device)
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(clip_adam_kernel, dim3(blocks), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n,
                       grad_scale, max_norm, lr, beta1, beta2, eps, bc1, bc2_sqrt, (const double *)partial, blocks,
                       total_norm_out, (const int64_t *)nullptr, (const float *)nullptr);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

TT_EXPORT int tt_clip_adam_step_gated_f32(float *params, float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                                          int64_t *step_counter, float lr, float beta1, float beta2, float eps,
                                          float max_norm, float grad_scale, float *total_norm_out, const float *gate,
                                          void *scratch, tt_stream_t stream)
{
    if (n < 0)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_clip_adam_step_gated_f32: n=%lld", (long long)n);
    if (n == 0)
        return TT_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !scratch || !step_counter)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_clip_adam_step_gated_f32: null pointer");
    hipStream_t st = (hipStream_t)stream;
    int blocks = (int)((n + 255) / 256);
    blocks = blocks > ADAM_BLOCKS ? ADAM_BLOCKS : blocks;
    double *partial = (double *)scratch;
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(blocks), dim3(256), 0, st, (const float *)grads, n, grad_scale,
                       partial, step_counter, gate);
    hipLaunchKernelGGL(clip_adam_kernel, dim3(blocks), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n,
                       grad_scale, max_norm, lr, beta1, beta2, eps, 1.0f, 1.0f, (const double *)partial, blocks,
                       total_norm_out, (const int64_t *)step_counter, gate);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

namespace {
struct GateWords {
    const int32_t *w[8];
};
__global__ void step_gate_kernel(GateWords g, int n, float *gate, int accumulate)
{
    const int b = threadIdx.x; // 0 .. TT_STEP_GATE_WORDS - 1
    int count = 0;
    for (int i = 0; i < n; ++i)
        if (g.w[i] && b < 3 && ((*g.w[i] >> b) & 1))
            ++count;
    gate[b] = (accumulate ? gate[b] : 0.0f) + (float)count;
}
} // namespace

TT_EXPORT int tt_step_gate_f32(const int32_t *const *status_words, int n_status, float *gate, tt_stream_t stream)
{
    if (n_status < 0 || !gate || (n_status && !status_words))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_step_gate_f32: n_status=%d, gate=%p", n_status, (void *)gate);
    // any number of words (gradient accumulation over many tower calls), eight per launch (they travel as kernel arguments)
    for (int base = 0; base == 0 || base < n_status; base += 8) {
        GateWords g = {};
        const int m = n_status - base < 8 ? n_status - base : 8;
        for (int i = 0; i < m; ++i)
            g.w[i] = status_words[base + i];
        hipLaunchKernelGGL(step_gate_kernel, dim3(1), dim3(TT_STEP_GATE_WORDS), 0, (hipStream_t)stream, g, m, gate, base > 0);
    }
    TT_LAUNCH_CHECK();
    return TT_OK;
}
