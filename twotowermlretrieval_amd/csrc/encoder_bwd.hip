// K7: gradient of RNNEncoder.forward w.r.t. every trainable tensor (what loss.backward() computes
// through backend/model.py:48-75; backend/main.py:254).  The embedding table is frozen
// (model.py:25-27) and receives no gradient.
//
//   head_bwd   F.normalize backward, then (bidirectional) the Linear(2H,H) backward as GEMMs
//   gru_bwd    reverse-time recurrence: one workgroup = the same 16 batch rows as the forward; the
//              gate derivatives are lane-local (they reuse the r,z,n,W_hn h stash the forward wrote),
//              dh_{t-1} = dh_t z + dGh W_hh runs on v_mfma_f32_16x16x4_f32 with dGh staged in LDS and
//              W_hh streamed from L2 in a pre-packed (transposed) B-operand order
//   weights    dW_ih = dGi^T X, dW_hh = dGh^T H_prev over ALL tokens at once: split-K fp32 MFMA GEMMs
//              with a deterministic slab reduction; X rows are gathered from the embedding table and
//              H_prev rows are addressed through a "previous token" index map (no copies)
#include "encoder.h"
#include "sgemm.h"

#include <stdlib.h>

int enc_check_shape(const char *who, int B, int T, int E, int H, int L, int64_t V);

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fast_tanh_b(float x) { return tt_fast_tanh(x); } // the forward's tanh: tanh(c_t) is the value h_t used

// d_hid = backward of y = hid / max(|hid|, 1e-12) (or identity)
// (block 0 also clears `zero_words`: the 16 scale words of this call's f16-split weight-gradient products, which the kernels
//  behind this one raise with atomicMax -- a separate launch before)
__global__ __launch_bounds__(256) void head_bwd_kernel(const float *__restrict__ hid, const float *__restrict__ d_out,
                                                       int H, int normalize, float *__restrict__ d_hid, unsigned *zero_words)
{
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b == 0 && tid < 16)
        zero_words[tid] = 0u;
    const float *x = hid + (size_t)b * H, *dy = d_out + (size_t)b * H;
    if (!normalize) {
        for (int u = tid; u < H; u += 256)
            d_hid[(size_t)b * H + u] = dy[u];
        return;
    }
    float ss = 0.0f, dt = 0.0f;
    for (int u = tid; u < H; u += 256) {
        ss += x[u] * x[u];
        dt += x[u] * dy[u];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        ss += __shfl_xor(ss, off);
        dt += __shfl_xor(dt, off);
    }
    if ((tid & 63) == 0) {
        red[tid >> 6] = ss;
        red[4 + (tid >> 6)] = dt;
    }
    __syncthreads();
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    const float dot = red[4] + red[5] + red[6] + red[7]; // x . dy
    for (int u = tid; u < H; u += 256) {
        float g;
        if (nrm < 1e-12f)
            g = dy[u] / 1e-12f;
        else
            g = (dy[u] - (x[u] / nrm) * (dot / nrm)) / nrm;
        d_hid[(size_t)b * H + u] = g;
    }
}

// column sums of X[m][n], m < *m_dyn, split over gridDim.y row slices -> slabs[y][n]; optionally the bit pattern of
// max |X| over the same elements -> *absmax (atomicMax; zeroed by the caller): the scale of the f16-split GEMMs
// that consume X next, for the price of this pass
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ X, int64_t ld, int N, int M,
                                                     const int *__restrict__ m_dyn, float *__restrict__ slabs,
                                                     unsigned *__restrict__ absmax)
{
    const int Me = m_dyn ? min(M, *m_dyn) : M;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int per = (Me + gridDim.y - 1) / gridDim.y;
    const int m0 = blockIdx.y * per, m1 = min(m0 + per, Me);
    float s = 0.0f, mx = 0.0f;
    if (n < N) {
        int m = m0;
        for (; m + 4 <= m1; m += 4) { // four loads in flight; the sum keeps its row order
            const float a = X[(size_t)m * ld + n], b = X[(size_t)(m + 1) * ld + n];
            const float c = X[(size_t)(m + 2) * ld + n], d = X[(size_t)(m + 3) * ld + n];
            s += a;
            s += b;
            s += c;
            s += d;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(a), fabsf(b))), fmaxf(fabsf(c), fabsf(d)));
        }
        for (; m < m1; ++m) {
            const float a = X[(size_t)m * ld + n];
            s += a;
            mx = fmaxf(mx, fabsf(a));
        }
        slabs[(size_t)blockIdx.y * N + n] = s;
    }
    if (absmax) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
            mx = fmaxf(mx, __shfl_xor(mx, off));
        if ((threadIdx.x & 63) == 0 && mx > 0.0f)
            atomicMax(absmax, __float_as_uint(mx));
    }
}

// wtp[((w*2 + ct)*(GH/16) + c)*256 + lane*4 + e] = W_hh[16c + 4(lane>>4) + e][32w + 16ct + (lane&15)],  GH = ng*H rows
__global__ __launch_bounds__(256) void pack_whh_t_kernel(const float *__restrict__ W, int H, int ng, float *__restrict__ wtp)
{
    const int n = ng * H * H / 4;
    const int nc = ng * H / 16;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int lane = i & 63;
        int r = i >> 6;
        const int c = r % nc;
        r /= nc;
        const int ct = r & 1, w = r >> 1;
        const int col = 32 * w + 16 * ct + (lane & 15);
        const int row = 16 * c + 4 * (lane >> 4);
        f32x4v v;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            v[e] = W[(size_t)(row + e) * H + col];
        *(f32x4v *)(wtp + (size_t)i * 4) = v;
    }
}

// (GruBwdDir / GruBwdParams: encoder.h)

// CELL_GRU: dGh = [dr_pre, dz_pre, dn_pre r], dh_{t-1} = dh z + dGh W_hh.   CELL_LSTM: dG = [di_pre, df_pre, dg_pre,
// do_pre] (the same for the input and the hidden side), dh_{t-1} = dG W_hh, dc carried in registers.   CELL_RNN:
// dG = dh (1 - h'^2), dh_{t-1} = dG W_hh.
template <int MAXW, int CELL>
__global__ __launch_bounds__(MAXW * 64) void gru_bwd_seq_kernel(GruBwdParams p)
{
    const uint64_t drop_seed_v = (p.drop_p > 0.0f && p.drop_seed_ptr) ? *p.drop_seed_ptr : p.drop_seed;
    constexpr int NG = CELL == CELL_LSTM ? 4 : (CELL == CELL_RNN ? 1 : 3);
    extern __shared__ __attribute__((aligned(16))) float gt[]; // [16][NG*H+4]
    const GruBwdDir d = p.dir[blockIdx.y];
    const int H = p.H, HG = NG * H, LDG = NG * H + 4, nc = HG / 16;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * ENC_RB;

    int len_e[4], off_e[4], rid_e[4], unit[2];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int br = row0 + kq * 4 + e;
        rid_e[e] = br < p.B ? p.perm[br] : -1;
        len_e[e] = rid_e[e] >= 0 ? p.len[rid_e[e]] : 0;
        off_e[e] = rid_e[e] >= 0 ? p.tok_off[rid_e[e]] : 0;
    }
    int steps = max(max(len_e[0], len_e[1]), max(len_e[2], len_e[3]));
    steps = max(steps, __shfl_xor(steps, 16));
    steps = max(steps, __shfl_xor(steps, 32));
    float dh[2][4];
    float dc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}; // LSTM: gradient w.r.t. the cell state
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        unit[ct] = 32 * w + 16 * ct + j;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            dh[ct][e] = (d.d_hfin && rid_e[e] >= 0) ? d.d_hfin[(size_t)rid_e[e] * H + unit[ct]] : 0.0f;
    }
    const float *wbase = d.wtp + (size_t)w * 2 * nc * 256 + lane * 4;

    // The stash of a step is loaded one step AHEAD, so the global latency hides under the previous step's MFMA loop
    // instead of sitting on the serial path of every step.  GRU: r,z,n,ghn,h_{t-1}.  LSTM: i,f,g,o in r,z,n,ghn,
    // c_{t-1} in hp, c_t in cn.  RNN: h_t in r.
    struct Stash {
        float r[2][4], z[2][4], n[2][4], ghn[2][4], hp[2][4], cn[2][4], dsv[2][4];
    };
    auto load_stash = [&](int s, Stash &st) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool a = s >= 0 && s < len_e[e];
            const int t = d.reverse ? len_e[e] - 1 - s : s;
            const size_t tok = (size_t)(off_e[e] + (a ? t : 0));
            const size_t ptok = d.reverse ? tok + 1 : tok - 1;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int u = unit[ct];
                st.r[ct][e] = st.z[ct][e] = st.n[ct][e] = st.ghn[ct][e] = st.hp[ct][e] = st.cn[ct][e] = st.dsv[ct][e] = 0.0f;
                if (a) {
                    if constexpr (CELL != CELL_RNN) {
                        const float *gs = d.gates + tok * 4 * H + u;
                        st.r[ct][e] = gs[0];
                        st.z[ct][e] = gs[H];
                        st.n[ct][e] = gs[2 * H];
                        st.ghn[ct][e] = gs[3 * H];
                    } else {
                        st.r[ct][e] = d.hseq[tok * p.ld + d.col0 + u];
                    }
                    if constexpr (CELL == CELL_GRU) {
                        if (s > 0)
                            st.hp[ct][e] = d.hseq[ptok * p.ld + d.col0 + u];
                    }
                    if constexpr (CELL == CELL_LSTM) {
                        st.cn[ct][e] = d.cseq[tok * H + u];
                        if (s > 0)
                            st.hp[ct][e] = d.cseq[ptok * H + u];
                    }
                    if (d.d_seq)
                        st.dsv[ct][e] = d.d_seq[tok * p.ld + d.col0 + u];
                }
            }
        }
    };
    Stash cur_st, next_st;
    load_stash(steps - 1, cur_st);

    for (int s = steps - 1; s >= 0; --s) {
        bool act[4];
        float direct[2][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            act[e] = s < len_e[e];
            const int t = d.reverse ? len_e[e] - 1 - s : s;
            const size_t tok = (size_t)(off_e[e] + (act[e] ? t : 0));
            float *grow = gt + (kq * 4 + e) * LDG;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int u = unit[ct];
                float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, g3 = 0.0f; // this lane's elements of dGh
                direct[ct][e] = 0.0f;
                if (act[e]) {
                    float dsv = cur_st.dsv[ct][e];
                    if (d.d_seq && p.drop_p > 0.0f)
                        dsv *= tt_dropout_scale(drop_seed_v, p.drop_layer,
                                                ((uint64_t)rid_e[e] * p.T + t) * p.ld + d.col0 + u, p.drop_p);
                    const float dhv = dh[ct][e] + dsv;
                    float *go = d.dgi + tok * HG + u;
                    if constexpr (CELL == CELL_GRU) {
                        const float r = cur_st.r[ct][e], z = cur_st.z[ct][e], n = cur_st.n[ct][e], ghn = cur_st.ghn[ct][e];
                        const float hp = cur_st.hp[ct][e];
                        const float dn_pre = dhv * (1.0f - z) * (1.0f - n * n);
                        g1 = dhv * (hp - n) * z * (1.0f - z);
                        g0 = dn_pre * ghn * r * (1.0f - r);
                        g2 = dn_pre * r;
                        direct[ct][e] = dhv * z;
                        go[0] = g0;
                        go[H] = g1;
                        go[2 * H] = dn_pre;
                        float *gh = d.dghn + tok * HG + u; // dGh differs from dGi in the n column only; kept whole so
                        gh[0] = g0;                        // that dW_hh and db_hh are ONE product / column sum each
                        gh[H] = g1;
                        gh[2 * H] = g2;
                    } else if constexpr (CELL == CELL_LSTM) {
                        const float ig = cur_st.r[ct][e], fg = cur_st.z[ct][e], gg = cur_st.n[ct][e], og = cur_st.ghn[ct][e];
                        const float cp = cur_st.hp[ct][e];
                        const float tc = fast_tanh_b(cur_st.cn[ct][e]);
                        const float dct = dc[ct][e] + dhv * og * (1.0f - tc * tc);
                        g0 = dct * gg * ig * (1.0f - ig);
                        g1 = dct * cp * fg * (1.0f - fg);
                        g2 = dct * ig * (1.0f - gg * gg);
                        g3 = dhv * tc * og * (1.0f - og);
                        dc[ct][e] = dct * fg;
                        go[0] = g0;
                        go[H] = g1;
                        go[2 * H] = g2;
                        go[3 * H] = g3;
                    } else {
                        const float hn = cur_st.r[ct][e];
                        g0 = dhv * (1.0f - hn * hn);
                        go[0] = g0;
                    }
                }
                grow[u] = g0;
                if constexpr (NG > 1)
                    grow[H + u] = g1;
                if constexpr (NG > 2)
                    grow[2 * H + u] = g2;
                if constexpr (NG > 3)
                    grow[3 * H + u] = g3;
            }
        }
        load_stash(s - 1, next_st); // in flight during the MFMA loop below
        __syncthreads();
        f32x4v acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        const float *gA = gt + j * LDG + 4 * kq;
        for (int c = 0; c < nc; ++c) {
            const f32x4v a = *(const f32x4v *)(gA + 16 * c);
            const f32x4v b0 = *(const f32x4v *)(wbase + (size_t)c * 256);
            const f32x4v b1 = *(const f32x4v *)(wbase + ((size_t)nc + c) * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b0[e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b1[e], acc[1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (act[e])
                    dh[ct][e] = direct[ct][e] + acc[ct][e];
        cur_st = next_st;
        __syncthreads();
    }
}

template <int CELL>
int launch_bwd_seq(const GruBwdParams &bp, int B, int H, int ndir, size_t lds, hipStream_t st)
{
    if (lds > 48 * 1024) {
        TT_HIP_CHECK(hipFuncSetAttribute((const void *)gru_bwd_seq_kernel<8, CELL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        TT_HIP_CHECK(hipFuncSetAttribute((const void *)gru_bwd_seq_kernel<16, CELL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (H <= 256)
        hipLaunchKernelGGL((gru_bwd_seq_kernel<8, CELL>), dim3((B + ENC_RB - 1) / ENC_RB, ndir), dim3(H / 32 * 64), lds, st, bp);
    else
        hipLaunchKernelGGL((gru_bwd_seq_kernel<16, CELL>), dim3((B + ENC_RB - 1) / ENC_RB, ndir), dim3(H / 32 * 64), lds, st, bp);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

// g_table[ids[m]][c] += dx[m][c] for the valid tokens m with ids[m] != 0 (float atomics at the memory side)
__global__ __launch_bounds__(256) void table_scatter_kernel(const float *__restrict__ dx, const int32_t *__restrict__ ids,
                                                            const int *__restrict__ m_dyn, int M, int E,
                                                            float *__restrict__ g_table)
{
    const int Me = min(M, *m_dyn);
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63, nw = gridDim.x * 4;
    for (int m = wave; m < Me; m += nw) {
        const int id = ids[m];
        if (id == 0)
            continue;
        for (int c = lane; c < E; c += 64)
            atomicAdd(g_table + (size_t)id * E + c, dx[(size_t)m * E + c]);
    }
}

// out_ih[c] = sum over row groups of slab[g][c], out_hh[c] = ... of slab[g][n + c].  A block takes 32 columns x 8 parts:
// part p adds its eighth of the row groups in order, the 8 partial sums are combined in part order (fixed order, so the
// result does not depend on the launch; one thread walking all groups of a column was 64 dependent loads, 12-19 us).
__global__ __launch_bounds__(256) void bias_reduce_kernel(const float *__restrict__ slab, int ngroups, int n,
                                                          float *__restrict__ out_ih, float *__restrict__ out_hh)
{
    __shared__ float part[8][32];
    const int col = threadIdx.x & 31, p = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + col;
    const int per = (ngroups + 7) / 8;
    float s = 0.0f;
    if (c < 2 * n)
        for (int g = p * per; g < min((p + 1) * per, ngroups); ++g)
            s += slab[(size_t)g * 2 * n + c];
    part[p][col] = s;
    __syncthreads();
    if (p == 0 && c < 2 * n) {
        float t = part[0][col];
#pragma unroll
        for (int q = 1; q < 8; ++q)
            t += part[q][col];
        if (c < n)
            out_ih[c] = t;
        else
            out_hh[c - n] = t;
    }
}

constexpr int COLSUM_SLICES = 256; // N <= 3H <= 1536 columns: 256 x N floats fit the split-K slab buffer

int colsum(const float *X, int64_t ld, int N, int M, const int *m_dyn, float *slabs, float *out, hipStream_t st,
           unsigned *absmax = nullptr)
{
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, COLSUM_SLICES), dim3(256), 0, st, X, ld, N, M, m_dyn, slabs,
                       absmax);
    TT_LAUNCH_CHECK();
    return tt_slab_reduce(slabs, COLSUM_SLICES, N, out, 0, st);
}

// C[Mo][No] = A[:, a0:a0+Mo]^T * Bsrc (both summed over tokens), split-K + deterministic reduce.
// a_absmax != nullptr: on the f16 pipes (fp16 hi/lo split of both operands, sgemm.h): A = gradients, scaled by the
// power of two that *a_absmax (max |A|, from the colsum pass over the same matrix) implies; B scaled by 2^b_exp.
// b_rows: an upper bound of Bsrc's row count (the 256-row-tile kernel addresses B with 32-bit offsets and checks it).
int gemm_tn(const float *A, int64_t lda, int Mo, const float *Bsrc, int64_t ldb, const int32_t *b_map, int64_t b_rows, int No,
            int Ktok, const int *k_dyn, float *slabs, float *out, hipStream_t st, const unsigned *a_absmax = nullptr,
            int b_exp = 0, const unsigned *b_absmax = nullptr)
{
    SgemmParams g;
    g.A = A;
    g.B = Bsrc;
    g.C = slabs;
    g.bias = nullptr;
    g.a_map = nullptr;
    g.b_map = b_map;
    g.m_dyn = nullptr;
    g.k_dyn = k_dyn;
    g.M = Mo;
    g.N = No;
    g.K = Ktok;
    g.lda = lda;
    g.ldb = ldb;
    g.ldc = No;
    g.slab_stride = (int64_t)Mo * No;
    g.accumulate = 0;
    g.a_absmax = a_absmax;
    g.b_absmax = a_absmax ? b_absmax : nullptr;
    g.a_exp = 0;
    g.b_exp = b_exp;
    g.b_hi16 = g.b_lo16 = nullptr;
    g.ldb16 = 0;
    if (a_absmax && tt_wgrad16_supported(Mo, No, lda, ldb, b_rows) && !((uintptr_t)A & 15) && !((uintptr_t)Bsrc & 15)) {
        // 256-row output tiles with the whole K slab in one workgroup per CU (wgrad16.hip)
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            cus <= 0) {
            (void)hipGetLastError();
            cus = 256;
        }
        const int ns = tt_wgrad16_slabs(Mo, No, cus, ENC_SPLITK);
        TT_RC_CHECK(tt_wgrad16(g, ns, st));
        return tt_slab_reduce(slabs, ns, (int64_t)Mo * No, out, 0, st);
    }
    int rc = a_absmax ? tt_sgemm16(g, true, true, ENC_SPLITK, st) : tt_sgemm(g, true, true, ENC_SPLITK, st);
    if (rc != TT_OK)
        return rc;
    return tt_slab_reduce(slabs, ENC_SPLITK, (int64_t)Mo * No, out, 0, st);
}

} // namespace

TT_EXPORT int tt_encoder_backward_f32(const int64_t *ids, int B, int T, const float *table, int64_t V, int E, int H,
                                      int num_layers, int bidirectional, int rnn_type, const float *const *weights,
                                      const float *proj_w, const float *proj_b, int normalize, float dropout_p,
                                      uint64_t dropout_seed, const float *d_out, float *const *grads,
                                      float *g_proj_w, float *g_proj_b, float *g_table, void *workspace,
                                      size_t workspace_bytes, int opts, int32_t *status, const tt_enc_sync_t *sync,
                                      tt_stream_t stream)
{
    (void)ids;
    (void)proj_b;
    hipStream_t st = (hipStream_t)stream;
    int rc = enc_check_shape("tt_encoder_backward_f32", B, T, E, H, num_layers, V);
    if (rc != TT_OK)
        return rc;
    if (opts & ~(TT_ENC_ONE_WORKGROUP | TT_ENC_SEED_ON_DEVICE | TT_ENC_F32))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_encoder_backward_f32: opts=0x%x (0, TT_ENC_ONE_WORKGROUP, TT_ENC_SEED_ON_DEVICE, TT_ENC_F32)", opts);
    const bool one_wg = (opts & TT_ENC_ONE_WORKGROUP) != 0;
    if (!table || !weights || !d_out || !grads || (bidirectional && (!proj_w || !g_proj_w || !g_proj_b)))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_encoder_backward_f32: null pointer");
    if (rnn_type < CELL_GRU || rnn_type > CELL_RNN)
        return tt_fail(TT_ERR_UNSUPPORTED, "tt_encoder_backward_f32: rnn_type=%d (0 GRU, 1 LSTM, 2 RNN)", rnn_type);
    const bool drop = dropout_p > 0.0f && num_layers > 1;
    const EncLayout lo = enc_layout(B, T, E, H, num_layers, bidirectional, g_table ? 2 : 1, drop, rnn_type);
    if (!workspace || workspace_bytes < lo.total || ((uintptr_t)workspace & 255))
        return tt_fail(TT_ERR_WORKSPACE, "tt_encoder_backward_f32: workspace %zu < %zu bytes: pass the buffer the "
                                         "forward call (train=%d) filled", workspace_bytes, lo.total, g_table ? 2 : 1);
    char *ws = (char *)workspace;
    const int ndir = lo.ndir, NG = lo.ng, H3 = NG * H; // H3: gate rows (3H for the GRU)
    const int32_t *len = (const int32_t *)(ws + lo.len), *tok_off = (const int32_t *)(ws + lo.tok_off);
    const int32_t *perm = (const int32_t *)(ws + lo.perm), *idsp = (const int32_t *)(ws + lo.ids);
    const int *m_valid = tok_off + B;
    float *slabs = (float *)(ws + lo.slabs);
    float *d_hfin = (float *)(ws + lo.d_hfin), *d_hid = (float *)(ws + lo.d_hid);
    const float *hfin = (const float *)(ws + lo.hfin);
    const int MT = (int)lo.MT;

    // scale words of the f16-split weight-gradient products (max |dGi|, max |dGh_n| per layer and direction), written
    // by the colsum passes: words 48..63 of the forward's status block

    // ---- head ------------------------------------------------------------------
    hipLaunchKernelGGL(head_bwd_kernel, dim3(B), dim3(256), 0, st, (const float *)(ws + lo.hid), d_out, H, normalize,
                       bidirectional ? d_hid : d_hfin, (unsigned *)(ws + lo.flag) + 48);
    TT_LAUNCH_CHECK();
    if (bidirectional) {
        rc = colsum(d_hid, H, H, B, nullptr, slabs, g_proj_b, st);
        if (rc != TT_OK)
            return rc;
        for (int d = 0; d < 2; ++d) {
            // g_proj_w[:, dH:(d+1)H] = d_hid^T * hfin[d]
            SgemmParams g;
            g.A = d_hid;
            g.B = hfin + (size_t)d * B * H;
            g.C = g_proj_w + (size_t)d * H;
            g.bias = nullptr;
            g.a_map = g.b_map = nullptr;
            g.m_dyn = g.k_dyn = nullptr;
            g.M = H;
            g.N = H;
            g.K = B;
            g.lda = H;
            g.ldb = H;
            g.ldc = 2 * H;
            g.slab_stride = 0;
            g.accumulate = 0;
            // The product sums over the B batch rows: with one k-range a 256 x 256 output is 4 workgroups walking B / 32
            // tiles each (160 us at B = 1024).  Split it over 16 k-ranges into contiguous slabs, reduce in fixed order,
            // then place the H x H block into its half of g_proj_w (row pitch 2H).
            const int psplit = B >= 256 ? 16 : 1;
            if (psplit > 1) {
                float *ptmp = slabs + (size_t)psplit * H * H; // behind the slabs, inside the split-K scratch
                g.C = slabs;
                g.ldc = H;
                g.slab_stride = (int64_t)H * H;
                rc = tt_sgemm(g, true, true, psplit, st);
                if (rc != TT_OK)
                    return rc;
                rc = tt_slab_reduce(slabs, psplit, (int64_t)H * H, ptmp, 0, st);
                if (rc != TT_OK)
                    return rc;
                TT_HIP_CHECK(hipMemcpy2DAsync(g_proj_w + (size_t)d * H, sizeof(float) * 2 * H, ptmp, sizeof(float) * H,
                                              sizeof(float) * H, H, hipMemcpyDeviceToDevice, st));
                g.slab_stride = 0;
            } else {
                rc = tt_sgemm(g, true, true, 1, st);
                if (rc != TT_OK)
                    return rc;
            }
            // d_hfin[d] = d_hid * proj_w[:, dH:(d+1)H]
            g.A = d_hid;
            g.B = proj_w + (size_t)d * H;
            g.C = d_hfin + (size_t)d * B * H;
            g.M = B;
            g.N = H;
            g.K = H;
            g.lda = H;
            g.ldb = 2 * H;
            g.ldc = H;
            rc = tt_sgemm(g, false, true, 1, st);
            if (rc != TT_OK)
                return rc;
        }
    }

    // (the "previous token" maps were made by the training forward's prep: csrc/encoder.hip)

    const size_t lds = sizeof(float) * ENC_RB * (H3 + 4);
    const bool force_f32 = (opts & TT_ENC_F32) != 0 || TT_AB_SWITCH(TT_GRU_F32, 0) != 0; // as the forward that filled the workspace
    const bool use16 = rnn_type == CELL_GRU && gru16_supported(H) && !force_f32; // (gru16.hip)
    const int n_rowgroups = (B + ENC_RB - 1) / ENC_RB;
    // the recurrence kernel's bias partial sums live in the split-K scratch (free while it runs): they must fit
    const size_t slab_floats = (size_t)ENC_SPLITK * (NG < 3 ? 3 : NG) * H * (size_t)((E > ndir * H ? E : ndir * H) > H ? (E > ndir * H ? E : ndir * H) : H);
    const bool fused_bias = use16 && (size_t)n_rowgroups * 2 * H3 * ndir <= slab_floats;
    for (int l = num_layers - 1; l >= 0; --l) {
        const int I = l == 0 ? E : ndir * H;
        const bool top = l == num_layers - 1;
        const float *hseq = (const float *)(ws + lo.x[l + 1]);
        const float *d_seq = top ? nullptr : (const float *)(ws + lo.dx[(l + 1) & 1]);
        GruBwdParams bp;
        bp.len = len;
        bp.tok_off = tok_off;
        bp.perm = perm;
        bp.B = B;
        bp.H = H;
        bp.ld = ndir * H;
        bp.drop_p = (drop && !top) ? dropout_p : 0.0f;
        bp.drop_seed = (opts & TT_ENC_SEED_ON_DEVICE) ? 0ull : dropout_seed;
        bp.drop_seed_ptr = (opts & TT_ENC_SEED_ON_DEVICE) ? (const uint64_t *)(uintptr_t)dropout_seed : nullptr;
        bp.drop_layer = l;
        bp.T = T;
        for (int d = 0; d < ndir; ++d) {
            const float *const *w = weights + ((size_t)l * ndir + d) * 4;
            // max |W_hh| of this layer and direction: the training forward left it in its status block (encoder.hip)
            const unsigned *wmax = (const unsigned *)(ws + lo.flag) + 16 + 2 * l + d;
            if (use16) {
                rc = gru16_pack_t(w[1], H, wmax, ws + lo.wtp[d], st);
                if (rc != TT_OK)
                    return rc;
            } else {
                hipLaunchKernelGGL(pack_whh_t_kernel, dim3(96), dim3(256), 0, st, w[1], H, NG, (float *)(ws + lo.wtp[d]));
            }
            bp.dir[d].wmax = wmax;
            bp.dir[d].gates = (const float *)(ws + lo.gates[l][d]);
            bp.dir[d].cseq = rnn_type == CELL_LSTM ? (const float *)(ws + lo.cseq[l][d]) : nullptr;
            bp.dir[d].hseq = hseq;
            bp.dir[d].d_seq = d_seq;
            bp.dir[d].d_hfin = top ? d_hfin + (size_t)d * B * H : nullptr;
            bp.dir[d].wtp = (const float *)(ws + lo.wtp[d]);
            bp.dir[d].dgi = (float *)(ws + lo.dgi[d]);
            bp.dir[d].dghn = (float *)(ws + lo.dghn[d]);
            bp.dir[d].col0 = d * H;
            bp.dir[d].reverse = d;
            // f16-split recurrence: bias sums per row group and the operand maxima come out of the kernel itself
            bp.dir[d].bias_slab = fused_bias ? slabs + (size_t)d * n_rowgroups * 2 * H3 : nullptr;
            bp.dir[d].mx_dgi = (unsigned *)(ws + lo.flag) + 48 + 2 * (2 * l + d);
            bp.dir[d].mx_dghn = bp.dir[d].mx_dgi + 1;
        }
        if (ndir == 1)
            bp.dir[1] = bp.dir[0];
        if (l == num_layers - 1 && sync && sync->wait_before_recurrence) // (the first recurrence launch of the call)
            TT_HIP_CHECK(hipStreamWaitEvent(st, (hipEvent_t)sync->wait_before_recurrence, 0));
        if (use16 && fused_bias && lo.xchb && !one_wg && gru16x4_bwd_usable(B, H, ndir)) {
            // a row group's reduction over the gate columns on four CUs (gru16x4.hip); a time-out ORs bit 2 into `status`
            rc = gru16x4_bwd_launch(bp, ndir, ws + lo.xchb, status, st);
        } else if (use16) {
            rc = gru16_bwd_launch(bp, ndir, st);
        } else {
            rc = rnn_type == CELL_LSTM ? launch_bwd_seq<CELL_LSTM>(bp, B, H, ndir, lds, st)
                                       : (rnn_type == CELL_RNN ? launch_bwd_seq<CELL_RNN>(bp, B, H, ndir, lds, st)
                                                               : launch_bwd_seq<CELL_GRU>(bp, B, H, ndir, lds, st));
        }
        if (rc != TT_OK)
            return rc;
        if (l == 0 && sync && sync->record_after_recurrence) // (the last recurrence launch of the call)
            TT_HIP_CHECK(hipEventRecord((hipEvent_t)sync->record_after_recurrence, st));
        if (fused_bias) {
            // the recurrence kernel left one partial sum per row group in the split-K scratch: reduce them (row-group order:
            // deterministic) for BOTH directions before the first weight-gradient product reuses that scratch
            for (int d = 0; d < ndir; ++d) {
                float *const *g = grads + ((size_t)l * ndir + d) * 4;
                hipLaunchKernelGGL(bias_reduce_kernel, dim3((2 * H3 + 31) / 32), dim3(256), 0, st,
                                   (const float *)(slabs + (size_t)d * n_rowgroups * 2 * H3), n_rowgroups, H3, g[2], g[3]);
            }
            TT_LAUNCH_CHECK();
        }

        for (int d = 0; d < ndir; ++d) {
            const float *const *w = weights + ((size_t)l * ndir + d) * 4;
            float *const *g = grads + ((size_t)l * ndir + d) * 4;
            const float *dgi = (const float *)(ws + lo.dgi[d]);
            const float *dghn = (const float *)(ws + lo.dghn[d]);
            // biases: b_ih <- colsum(dGi); b_hh <- colsum(dGh) (GRU: the n gate's hidden-side pre-activation is scaled by r)
            // or the same sums (LSTM / RNN: one pre-activation per gate)
            unsigned *mx_dgi = force_f32 ? nullptr : (unsigned *)(ws + lo.flag) + 48 + 2 * (2 * l + d);
            unsigned *mx_dghn = force_f32 ? nullptr : mx_dgi + 1;
            if (!fused_bias) {
                rc = colsum(dgi, H3, H3, MT, m_valid, slabs, g[2], st, mx_dgi);
                if (rc != TT_OK)
                    return rc;
            }
            if (fused_bias) {
            } else if (rnn_type == CELL_GRU) {
                rc = colsum(dghn, H3, H3, MT, m_valid, slabs, g[3], st, mx_dghn);
                if (rc != TT_OK)
                    return rc;
            } else {
                TT_HIP_CHECK(hipMemcpyAsync(g[3], g[2], sizeof(float) * H3, hipMemcpyDeviceToDevice, st));
            }
            // W_ih <- dGi^T X   (X = gathered table rows for layer 0, the layer below's output above)
            // (f16-split products: embedding rows scaled by the power of two of the batch's largest |x|, which the training
            //  forward left in the status block; hidden states (|h| < 1, times 1/(1-p) when dropped) 2^6)
            if (l == 0)
                rc = gemm_tn(dgi, H3, H3, table, E, idsp, V, E, MT, m_valid, slabs, g[0], st, mx_dgi, 0,
                             (const unsigned *)(ws + lo.flag) + ENC_FLAG_XMAX);
            else
                rc = gemm_tn(dgi, H3, H3, (const float *)(ws + (drop ? lo.xd[l] : lo.x[l])), I, nullptr, MT, I, MT, m_valid,
                             slabs, g[0], st, mx_dgi, 6);
            if (rc != TT_OK)
                return rc;
            // W_hh <- dGh^T H_prev, H_prev rows through the previous-token map into this layer's own output
            const int32_t *pm = (const int32_t *)(ws + lo.prevmap[d]);
            // (GRU: the hidden-side gradients dGh = [dr_pre, dz_pre, dn_pre r]; LSTM / RNN: the same matrix as dGi)
            rc = gemm_tn(rnn_type == CELL_GRU ? dghn : dgi, H3, H3, hseq + (size_t)d * H, ndir * H, pm, (int64_t)MT + 1, H, MT, m_valid, slabs, g[1],
                         st, rnn_type == CELL_GRU ? mx_dghn : mx_dgi, 10);
            if (rc != TT_OK)
                return rc;
            // gradient w.r.t. this layer's input sequence (below layer 0 only when the table is trained)
            if (l > 0 || g_table) {
                SgemmParams x;
                x.A = dgi;
                x.B = w[0];
                x.C = l > 0 ? (float *)(ws + lo.dx[l & 1]) : (float *)(ws + lo.dx0);
                x.bias = nullptr;
                x.a_map = x.b_map = nullptr;
                x.m_dyn = m_valid;
                x.k_dyn = nullptr;
                x.M = MT;
                x.N = I;
                x.K = H3;
                x.lda = H3;
                x.ldb = I;
                x.ldc = I;
                x.slab_stride = 0;
                x.accumulate = d;
                if (!force_f32) {
                    // on the f16 pipes like every other product of the step: dGi scaled by its maximum (the column-sum
                    // pass found it), W_ih by the power of two the forward pass derived (status word 40 + 2 l + d)
                    x.a_absmax = mx_dgi;
                    x.b_absmax = (const unsigned *)(ws + lo.flag) + 40 + 2 * l + d;
                    x.a_exp = x.b_exp = 0;
                    x.b_hi16 = x.b_lo16 = nullptr;
                    x.ldb16 = 0;
                    rc = tt_sgemm16(x, false, true, 1, st);
                } else
                    rc = tt_sgemm(x, false, true, 1, st);
                if (rc != TT_OK)
                    return rc;
            }
        }
    }
    if (g_table) {
        // nn.Embedding backward (model.py:23-27 without GloVe vectors): row id accumulates the input gradients of the
        // positions that hold it; padding_idx = 0 gets none.  Dense [V,E] gradient, as torch's.
        TT_RC_CHECK(tt_zero_async(g_table, sizeof(float) * (size_t)V * E, st));
        hipLaunchKernelGGL(table_scatter_kernel, dim3(2048), dim3(256), 0, st, (const float *)(ws + lo.dx0), idsp, m_valid, MT,
                           E, g_table);
        TT_LAUNCH_CHECK();
    }
    return TT_OK;
}
