// K1-K3: RNNEncoder.forward (GRU) -- backend/model.py:48-75 -- on gfx950.
//
//   prep      lengths = count of non-zero ids (model.py:52, computed ON DEVICE: the reference's
//             .cpu() sync is not reproduced), id range check, exclusive scan -> packed token
//             offsets, counting sort of rows by length (so a recurrence block's 16 rows have
//             similar lengths), packed id list.
//   K1        input projections for ALL valid tokens at once: Gi = X W_ih^T + b_ih as one fp32
//             MFMA GEMM whose A rows are gathered straight from the embedding table
//             (model.py:49 + the x-side half of nn.GRU, model.py:59-62); padded positions are
//             never computed (packed token order).
//   K2        the recurrence: one workgroup = 16 batch rows x all H hidden units, persistent over
//             the T steps of its rows; h lives in LDS (double-buffered, fp32) and in registers;
//             per step Gh = h W_hh^T + b_hh on v_mfma_f32_16x16x4_f32 with W_hh streamed from L2 in
//             a pre-packed, fully coalesced B-operand order; gates r,z,n and the blend are lane-local
//             (the three gates of one (row, unit) sit in the same lane).
//   K3        head: (bidirectional) cat + Linear(2H,H) (model.py:65-69), F.normalize eps 1e-12
//             (model.py:73-74).
#include "encoder.h"
#include "pack16.h"
#include "sgemm.h"

#include <limits.h>
#include <stdlib.h>

namespace {

// ------------------------------------------------------------------ prep
__global__ __launch_bounds__(256) void prep_len_kernel(const int64_t *__restrict__ ids, int B, int T, int64_t V,
                                                       int32_t *__restrict__ len, int32_t *__restrict__ flag)
{
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B)
        return;
    int c = 0, bad = 0;
    for (int t = lane; t < T; t += 64) {
        const int64_t id = ids[(size_t)b * T + t];
        c += id != 0;
        bad |= (id < 0 || id >= V);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        c += __shfl_xor(c, off);
        bad |= __shfl_xor(bad, off);
    }
    if (lane == 0) {
        len[b] = c;
        if (c == 0)
            atomicOr(flag, 1); // zero-length row: pack_padded_sequence raises (model.py:55-57)
        if (bad)
            atomicOr(flag, 2); // id outside [0,V): nn.Embedding raises IndexError
    }
}

constexpr int SORT_BINS = 256; // length classes of the row sort (8 ballots per row; [16][256] per-wave counts in LDS)

int enc_cus()
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        cus = 256;
    }
    return cus;
}

// Single block of 1024 threads: tok_off = exclusive scan of len; perm = rows sorted by length, longest first (a counting
// sort over SORT_BINS length classes), rows of one class in ASCENDING ROW ORDER.  The order inside a class matters: perm
// decides which rows share a 16-row recurrence workgroup, and the backward kernels sum the bias gradients per workgroup, so
// a scatter that hands out a class's slots by LDS atomics -- in whatever order the waves arrive -- made those sums (not the
// weight gradients, which are token-parallel) differ in their last bits from one run to the next.  Here the scatter is
// STABLE: rows are taken 1024 at a time; inside a wave a row's rank among the rows of its class is a popcount over the
// lanes below it (the class's lane mask: one ballot per class bit), the waves' counts per class are laid out in wave order
// by a 16-step scan, and a row lands at class start + rows of earlier chunks and waves + its rank.  No atomics, no second pass.
__device__ __forceinline__ void scan_sort_block(const int32_t *len, int B, int T, int32_t *tok_off, int32_t *perm)
{
    __shared__ int hist[SORT_BINS];      // class sizes, then the running start of each class
    __shared__ int whist[16][SORT_BINS]; // per wave and class: count in the current chunk, then that wave's base slot
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < SORT_BINS; i += 1024)
        hist[i] = 0;
    if (tid == 0)
        carry = 0;
    __syncthreads();
    const int nb = min(T + 1, SORT_BINS);
    auto bin_of = [&](int l) { return nb - 1 - (int)((int64_t)l * nb / (T + 1)); }; // long rows -> low classes
    for (int base = 0; base < B; base += 1024) {
        const int b = base + tid;
        const int v = b < B ? len[b] : 0;
        if (b < B)
            atomicAdd(&hist[bin_of(v)], 1); // (a count: the order of the adds does not matter)
        int x = v; // inclusive wave scan
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(x, off);
            if (lane >= off)
                x += y;
        }
        if (lane == 63)
            wsum[wave] = x;
        __syncthreads();
        int pre = carry;
        for (int w = 0; w < wave; ++w)
            pre += wsum[w];
        if (b < B)
            tok_off[b] = pre + x - v;
        __syncthreads();
        if (tid == 1023)
            carry = pre + x;
        __syncthreads();
    }
    if (tid == 0)
        tok_off[B] = carry;
    // exclusive scan of the class sizes (SORT_BINS = 256: one class per thread of the first four waves)
    {
        const int s = tid < SORT_BINS ? hist[tid] : 0;
        int x = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(x, off);
            if (lane >= off)
                x += y;
        }
        if (lane == 63)
            wsum[wave] = x;
        __syncthreads();
        int pre = 0;
        for (int w = 0; w < wave; ++w)
            pre += wsum[w];
        if (tid < SORT_BINS)
            hist[tid] = pre + x - s;
        __syncthreads();
    }
    for (int base = 0; base < B; base += 1024) {
        for (int i = tid; i < 16 * SORT_BINS; i += 1024)
            (&whist[0][0])[i] = 0;
        __syncthreads();
        const int b = base + tid;
        const bool valid = b < B;
        const int bin = valid ? bin_of(len[b]) : 0;
        unsigned long long mask = __ballot(valid); // lanes of this wave in the same class as this one
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const unsigned long long bb = __ballot(valid && ((bin >> bit) & 1));
            mask &= ((bin >> bit) & 1) ? bb : ~bb;
        }
        const int rank = __popcll(mask & ((1ull << lane) - 1ull));
        if (valid && rank == 0)
            whist[wave][bin] = __popcll(mask);
        __syncthreads();
        if (tid < SORT_BINS) {
            int run = hist[tid];
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                const int c = whist[w][tid];
                whist[w][tid] = run;
                run += c;
            }
            hist[tid] = run;
        }
        __syncthreads();
        if (valid)
            perm[whist[wave][bin] + rank] = b;
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void prep_scan_sort_kernel(const int32_t *__restrict__ len, int B, int T,
                                                              int32_t *__restrict__ tok_off, int32_t *__restrict__ perm)
{
    scan_sort_block(len, B, T, tok_off, perm);
}

// The whole prep in ONE workgroup for small batches (B <= 1024 rows, B*T <= 8192 ids: every query-tower call): clears the
// 64-word status block, lengths + input flags, scan + sort, packed ids and the caller's status word -- five dependent
// launches of ~5 us each otherwise.  Thread = id (8 per thread, all loads in flight at once; a row per wave with its loads
// one after the other was slower than the five launches); a row's non-zero count is the popcount of the wave's ballot
// over the row's lanes, added to an LDS counter by the row's first lane in the wave.
constexpr int PREP_FUSED_PER = 8;
constexpr int64_t PREP_FUSED_MAX_IDS = 1024 * PREP_FUSED_PER;
// pm_fwd / pm_rev (training; nullable): packed index of the token one step earlier / later in its row, `none` where there is
// none (the backward's "previous token" maps: they only depend on the lengths, so the prep makes them).
__global__ __launch_bounds__(1024) void prep_fused_kernel(const int64_t *__restrict__ ids, int B, int T, int64_t V,
                                                          int32_t *len, int32_t *flag, int32_t *tok_off, int32_t *perm,
                                                          int32_t *__restrict__ packed, int32_t *__restrict__ status,
                                                          int32_t *__restrict__ pm_fwd, int32_t *__restrict__ pm_rev, int none)
{
    __shared__ int cnt[1024];
    __shared__ int st_bits;
    const int tid = threadIdx.x, lane = tid & 63;
    const int n = B * T;
    cnt[tid] = 0;
    if (tid == 0)
        st_bits = 0;
    __syncthreads();
    int64_t id[PREP_FUSED_PER];
#pragma unroll
    for (int i = 0; i < PREP_FUSED_PER; ++i) {
        const int idx = i * 1024 + tid;
        id[i] = idx < n ? ids[idx] : 0;
    }
    int bad = 0;
#pragma unroll
    for (int i = 0; i < PREP_FUSED_PER; ++i) {
        const int idx = i * 1024 + tid;
        const bool live = idx < n;
        bad |= live && (id[i] < 0 || id[i] >= V);
        const unsigned long long nz = __ballot(live && id[i] != 0);
        if (live) {
            const int row = idx / T, first = row * T, base = idx - lane; // the wave covers ids [base, base + 64)
            const int lo = max(first - base, 0), hi = min(first + T - base, 64);
            if (lane == lo) { // the row's first lane in this wave
                const unsigned long long m = (hi >= 64 ? ~0ull : ((1ull << hi) - 1)) & ~((1ull << lo) - 1);
                const int c = __popcll(nz & m);
                if (c)
                    atomicAdd(&cnt[row], c);
            }
        }
    }
    if (__ballot(bad) != 0ull && lane == 0)
        atomicOr(&st_bits, 2); // id outside [0,V): nn.Embedding raises IndexError
    __syncthreads();
    if (tid < B) {
        len[tid] = cnt[tid];
        if (cnt[tid] == 0)
            atomicOr(&st_bits, 1); // zero-length row: pack_padded_sequence raises (model.py:55-57)
    }
    __syncthreads(); // (workgroup-scope fence: the lengths are visible to every thread of the block)
    if (tid < 64)
        flag[tid] = tid == 0 ? st_bits : 0;
    if (tid == 0 && status)
        status[0] = st_bits;
    __syncthreads();
    scan_sort_block(len, B, T, tok_off, perm);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PREP_FUSED_PER; ++i) {
        const int idx = i * 1024 + tid;
        if (idx < n) {
            const int row = idx / T, t = idx - row * T;
            if (t < cnt[row]) {
                const int o = tok_off[row] + t;
                packed[o] = (int32_t)(id[i] < 0 || id[i] >= V ? 0 : id[i]); // flagged, never dereferenced
                if (pm_fwd)
                    pm_fwd[o] = t > 0 ? o - 1 : none;
                if (pm_rev)
                    pm_rev[o] = t < cnt[row] - 1 ? o + 1 : none;
            }
        }
    }
}

// (last kernel of the four-kernel prep: block 0 also hands the status word to the caller -- a device-to-device copy before)
__global__ __launch_bounds__(256) void prep_pack_ids_kernel(const int64_t *__restrict__ ids, int B, int T,
                                                            const int32_t *__restrict__ len,
                                                            const int32_t *__restrict__ tok_off, int64_t V,
                                                            int32_t *__restrict__ packed, int32_t *flag, int32_t *status,
                                                            int32_t *__restrict__ pm_fwd, int32_t *__restrict__ pm_rev, int none)
{
    const int b = blockIdx.x;
    const int L = len[b], o = tok_off[b];
    if (b == 0 && threadIdx.x == 0 && status)
        status[0] = flag[0];
    for (int t = threadIdx.x; t < L; t += 256) {
        int64_t id = ids[(size_t)b * T + t];
        packed[o + t] = (int32_t)(id < 0 || id >= V ? 0 : id); // out-of-range ids are flagged, never dereferenced
        if (pm_fwd)
            pm_fwd[o + t] = t > 0 ? o + t - 1 : none;
        if (pm_rev)
            pm_rev[o + t] = t < L - 1 ? o + t + 1 : none;
    }
}

// ------------------------------------------------------------------ W_hh -> MFMA B-operand order
// wp[(((w*ng + g)*2 + ct)*(H/16) + c)*256 + lane*4 + e] = W_hh[g*H + 32w + 16ct + (lane&15)][16c + 4(lane>>4) + e]
__global__ __launch_bounds__(256) void pack_whh_kernel(const float *__restrict__ W, int H, int ng, float *__restrict__ wp)
{
    const int n = ng * H * H / 4;
    const int nc = H / 16;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int lane = i & 63;
        int r = i >> 6;
        const int c = r % nc;
        r /= nc;
        const int ct = r & 1;
        r >>= 1;
        const int g = r % ng, w = r / ng;
        const int row = g * H + 32 * w + 16 * ct + (lane & 15);
        const int col = 16 * c + 4 * (lane >> 4);
        *(f32x4 *)(wp + (size_t)i * 4) = *(const f32x4 *)(W + (size_t)row * H + col);
    }
}

// ------------------------------------------------------------------ K2 recurrence (GruDir / GruParams: encoder.h)
__device__ __forceinline__ float fast_sigmoid(float x) { return tt_fast_sigmoid(x); }
__device__ __forceinline__ float fast_tanh(float x) { return tt_fast_tanh(x); }

typedef float f32x4v __attribute__((ext_vector_type(4)));

// CELL: CELL_GRU (gates r,z,n), CELL_LSTM (i,f,g,o; the cell state lives in registers next to h), CELL_RNN (tanh).
template <int MAXW, int CELL>
__global__ __launch_bounds__(MAXW * 64) void gru_seq_kernel(GruParams p)
{
    constexpr int NG = CELL == CELL_LSTM ? 4 : (CELL == CELL_RNN ? 1 : 3);
    extern __shared__ __attribute__((aligned(16))) float hbuf[]; // [2][16][H+4]
    const GruDir d = p.dir[blockIdx.y];
    const int H = p.H, LDH = H + 4, nc = H / 16;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kq = lane >> 4;
    // Rows are sorted longest first and two workgroups share a CU: with the plain order CU c would run groups c and
    // c + slots/2 (long + medium) while the last CUs idle early.  Inside each round of `slots` workgroups the second
    // half is taken in reverse, so the CU that got the longest group also gets the shortest (LPT pairing).
    int grp = blockIdx.x;
    {
        const int base = grp / p.slots * p.slots, r = grp - base;
        const int nr = min(p.slots, (int)gridDim.x - base), half = (nr + 1) / 2;
        if (r >= half)
            grp = base + nr - 1 - (r - half);
    }
    const int row0 = grp * ENC_RB;

    int len_e[4], off_e[4], rid_e[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int br = row0 + kq * 4 + e;
        rid_e[e] = br < p.B ? p.perm[br] : -1;
        len_e[e] = rid_e[e] >= 0 ? p.len[rid_e[e]] : 0;
        off_e[e] = rid_e[e] >= 0 ? p.tok_off[rid_e[e]] : 0;
    }
    int steps = max(max(len_e[0], len_e[1]), max(len_e[2], len_e[3])); // block-wide max length
    steps = max(steps, __shfl_xor(steps, 16));
    steps = max(steps, __shfl_xor(steps, 32));
    int unit[2];
    float bias[NG][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        unit[ct] = 32 * w + 16 * ct + j;
#pragma unroll
        for (int g = 0; g < NG; ++g)
            bias[g][ct] = d.b_hh[g * H + unit[ct]];
    }
    for (int i = threadIdx.x; i < 2 * ENC_RB * LDH; i += blockDim.x)
        hbuf[i] = 0.0f;
    float hreg[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    float creg[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}; // LSTM cell state
    __syncthreads();

    const float *wbase = d.wp + (size_t)w * 2 * NG * nc * 256 + lane * 4;
    const int HG = NG * H;
    int cur = 0;
    for (int s = 0; s < steps; ++s) {
        bool act[4];
        size_t tok[4];
        float giv[NG][2][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            act[e] = s < len_e[e];
            const int t = d.reverse ? len_e[e] - 1 - s : s;
            tok[e] = (size_t)(off_e[e] + (act[e] ? t : 0));
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    giv[g][ct][e] = act[e] ? d.gi[tok[e] * HG + g * H + unit[ct]] : 0.0f;
        }
        f32x4v acc[NG][2];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                acc[g][ct] = (f32x4v){bias[g][ct], bias[g][ct], bias[g][ct], bias[g][ct]};

        const float *hA = hbuf + cur * ENC_RB * LDH + j * LDH + 4 * kq;
        for (int c = 0; c < nc; ++c) {
            const f32x4v a = *(const f32x4v *)(hA + 16 * c);
            f32x4v b[NG][2];
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    b[g][ct] = *(const f32x4v *)(wbase + ((size_t)(g * 2 + ct) * nc + c) * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[g][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[g][ct][e], acc[g][ct], 0, 0, 0);
        }

        float *hN = hbuf + (cur ^ 1) * ENC_RB * LDH;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float hn;
                if constexpr (CELL == CELL_GRU) {
                    const float r = fast_sigmoid(giv[0][ct][e] + acc[0][ct][e]);
                    const float z = fast_sigmoid(giv[1][ct][e] + acc[1][ct][e]);
                    const float ghn = acc[2][ct][e];
                    const float n = fast_tanh(giv[2][ct][e] + r * ghn);
                    hn = (hreg[ct][e] - n) * z + n;
                    if (act[e] && d.gates) {
                        float *gs = d.gates + tok[e] * 4 * H + unit[ct];
                        gs[0] = r;
                        gs[H] = z;
                        gs[2 * H] = n;
                        gs[3 * H] = ghn;
                    }
                } else if constexpr (CELL == CELL_LSTM) {
                    const float ig = fast_sigmoid(giv[0][ct][e] + acc[0][ct][e]);
                    const float fg = fast_sigmoid(giv[1][ct][e] + acc[1][ct][e]);
                    const float gg = fast_tanh(giv[2][ct][e] + acc[2][ct][e]);
                    const float og = fast_sigmoid(giv[3][ct][e] + acc[3][ct][e]);
                    const float cn = fg * creg[ct][e] + ig * gg;
                    hn = og * fast_tanh(cn);
                    if (act[e]) {
                        creg[ct][e] = cn;
                        if (d.gates) {
                            float *gs = d.gates + tok[e] * 4 * H + unit[ct];
                            gs[0] = ig;
                            gs[H] = fg;
                            gs[2 * H] = gg;
                            gs[3 * H] = og;
                        }
                        if (d.cseq)
                            d.cseq[tok[e] * H + unit[ct]] = cn;
                    }
                } else {
                    hn = fast_tanh(giv[0][ct][e] + acc[0][ct][e]);
                }
                if (act[e]) {
                    hreg[ct][e] = hn;
                    if (d.out_seq)
                        d.out_seq[tok[e] * p.out_ld + d.out_col0 + unit[ct]] = hn;
                }
                hN[(kq * 4 + e) * LDH + unit[ct]] = hreg[ct][e];
            }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (rid_e[e] >= 0)
                d.h_final[(size_t)rid_e[e] * H + unit[ct]] = hreg[ct][e];
}

template <int CELL>
int launch_seq(const GruParams &gp, int B, int H, int ndir, size_t lds, hipStream_t st)
{
    if (H <= 256)
        hipLaunchKernelGGL((gru_seq_kernel<8, CELL>), dim3((B + ENC_RB - 1) / ENC_RB, ndir), dim3(H / 32 * 64), lds, st, gp);
    else
        hipLaunchKernelGGL((gru_seq_kernel<16, CELL>), dim3((B + ENC_RB - 1) / ENC_RB, ndir), dim3(H / 32 * 64), lds, st, gp);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

// ------------------------------------------------------------------ inter-layer dropout (train mode)
// xd = x * mask/(1-p) over the valid tokens of every row; mask = tt_dropout_scale(seed, layer, padded index)
__global__ __launch_bounds__(256) void dropout_apply_kernel(const float *__restrict__ x, float *__restrict__ xd,
                                                            const int32_t *__restrict__ len,
                                                            const int32_t *__restrict__ tok_off, int T, int ld,
                                                            int layer, float p, uint64_t seed_value,
                                                            const uint64_t *__restrict__ seed_ptr)
{
    const uint64_t seed = seed_ptr ? *seed_ptr : seed_value; // (TT_ENC_SEED_ON_DEVICE: read when the kernel runs)
    const int b = blockIdx.x;
    const int L = len[b], o = tok_off[b];
    // grid.y blocks share a row (a small batch is few rows: one block per row left most of the chip idle); a thread takes whole
    // tokens' columns c, c + 256, ... so that no index is divided
    for (int t = blockIdx.y; t < L; t += gridDim.y) {
        const size_t tok = (size_t)(o + t);
        const uint64_t base = ((uint64_t)b * T + t) * ld;
        for (int c = threadIdx.x; c < ld; c += 256)
            xd[tok * ld + c] = x[tok * ld + c] * tt_dropout_scale(seed, layer, base + c, p);
    }
}

// ------------------------------------------------------------------ K3 head
__global__ __launch_bounds__(256) void head_kernel(const float *__restrict__ hfin, int B, int H, int ndir,
                                                   const float *__restrict__ proj_w,
                                                   const float *__restrict__ proj_b, int normalize,
                                                   float *__restrict__ hid_out, float *__restrict__ out)
{
    __shared__ float cat[1024];
    __shared__ float hid[512];
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < ndir * H; i += 256)
        cat[i] = hfin[(size_t)(i / H) * B * H + (size_t)b * H + (i % H)];
    __syncthreads();
    float ss = 0.0f;
    for (int u = tid; u < H; u += 256) {
        float v;
        if (ndir == 2) {
            v = proj_b[u];
            const float *wr = proj_w + (size_t)u * 2 * H;
            for (int k = 0; k < 2 * H; ++k)
                v += cat[k] * wr[k];
        } else {
            v = cat[u];
        }
        hid[u] = v;
        if (hid_out)
            hid_out[(size_t)b * H + u] = v;
        ss += v * v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        ss += __shfl_xor(ss, off);
    if ((tid & 63) == 0)
        red[tid >> 6] = ss;
    __syncthreads();
    const float nrm = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);
    for (int u = tid; u < H; u += 256)
        out[(size_t)b * H + u] = normalize ? hid[u] / nrm : hid[u];
}

} // namespace

int enc_check_shape(const char *who, int B, int T, int E, int H, int L, int64_t V)
{
    if (B <= 0 || T <= 0)
        return tt_fail(TT_ERR_BAD_SHAPE, "%s: Cannot pack empty tensors (B=%d T=%d)", who, B, T);
    if (H < 32 || H > 512 || (H & 31))
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: HIDDEN_DIM=%d (supported: multiples of 32 in [32,512])", who, H);
    if (E < 4 || (E & 3))
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: EMBED_DIM=%d must be a multiple of 4", who, E);
    if (L < 1 || L > ENC_MAX_LAYERS)
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: NUM_LAYERS=%d (supported: 1..%d)", who, L, ENC_MAX_LAYERS);
    if ((int64_t)B * T >= INT_MAX / 4 || V <= 0 || V >= INT_MAX)
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: B*T or V too large for 32-bit token indices", who);
    return TT_OK;
}

TT_EXPORT size_t tt_encoder_workspace_bytes(int B, int T, int E, int H, int num_layers, int bidirectional, int rnn_type,
                                            int train, int dropout)
{
    if (B <= 0 || T <= 0 || num_layers < 1 || num_layers > ENC_MAX_LAYERS || rnn_type < 0 || rnn_type > 2)
        return 0;
    const bool projected = train > 0 && (train & TT_ENC_PROJECTED) != 0;
    train = train < 0 ? 0 : (train & TT_ENC_TRAIN_MASK); // (the other option bits above the mask do not change the layout)
    return enc_layout(B, T, E, H, num_layers, bidirectional, train > 2 ? 2 : train, dropout, rnn_type, projected && train == 0).total;
}

// H = 128 / 256: the recurrence runs on the f16 matrix pipes with both operands split into fp16 hi + lo parts
// (gru16.hip; fp32-grade accuracy at 3/16 of the fp32 MFMA time).  The option bit TT_ENC_F32 of a call (include/tt.h) keeps
// every product of that call on the fp32-MFMA kernels instead: the reference's own arithmetic (nn.GRU in fp32, model.py:31-37).
static bool enc_force_f32(int flags = 0)
{
    return (flags & TT_ENC_F32) != 0 || TT_AB_SWITCH(TT_GRU_F32, 0) != 0;
}
static bool enc_tiled_k1() // A/B switch: the tiled f16 GEMM instead of the token-stationary one
{
    return TT_AB_SWITCH(TT_K1_TILED, 0) != 0;
}
static bool enc_rows16(int NGH, int I) { return !enc_tiled_k1() && tt_gemm_rows16_supported(NGH, I, I, NGH); }

// One layer and direction's weights into the forms the forward kernels read.  wih_max / wmax: zeroed words that receive
// the bit patterns of max |W_ih| / max |W_hh|; w16: W_ih split into fp16 hi/lo parts (scaled by the power of two its
// maximum implies) as gemm_rows16's fragment stream or the tiled GEMM's [rows][Kp] images; wp: W_hh in packed order.
namespace {
// both conversions of a GRU layer in one launch (pack16.h): blockIdx.y = 0: W_ih -> fragment stream, 1: W_hh -> packed order
__global__ __launch_bounds__(256) void pack2_kernel(const float *__restrict__ Wih, int N, int K, int nks, int waves, int ctn,
                                                    const unsigned *__restrict__ wih_max, _Float16 *__restrict__ w16,
                                                    const float *__restrict__ Whh, int H, const unsigned *__restrict__ wmax,
                                                    _Float16 *__restrict__ wp16)
{
    if (blockIdx.y == 0)
        pack_frag16_body(Wih, N, K, nks, waves, ctn, wih_max, w16, (int)blockIdx.x, (int)gridDim.x);
    else
        pack_whh16_body(Whh, H, wmax, wp16, (int)blockIdx.x, (int)gridDim.x);
}
} // namespace

static int enc_pack_weights(int I, int H, int rnn_type, const float *const *w, unsigned *wih_max, unsigned *wmax, char *w16,
                            char *wp, hipStream_t st, bool f32 = false)
{
    const int NGH = enc_gates(rnn_type) * H;
    f32 = f32 || enc_force_f32();
    if (!f32 && rnn_type == CELL_GRU && gru16_supported(H) && enc_rows16(NGH, I) && (I + 15) / 16 <= 19) {
        // the north-star shape: two launches instead of four (both maxima, then both conversions)
        TT_RC_CHECK(tt_absmax2(w[0], (int64_t)NGH * I, wih_max, w[1], (int64_t)3 * H * H, wmax, st));
        const int nks = (I + 15) / 16;
        hipLaunchKernelGGL(pack2_kernel, dim3(96, 2), dim3(256), 0, st, w[0], NGH, I, nks, 4, 2, (const unsigned *)wih_max,
                           (_Float16 *)w16, w[1], H, (const unsigned *)wmax, (_Float16 *)wp);
        TT_LAUNCH_CHECK();
        return TT_OK;
    }
    if (!f32) {
        TT_RC_CHECK(tt_absmax(w[0], (int64_t)NGH * I, wih_max, st));
        if (enc_rows16(NGH, I)) {
            TT_RC_CHECK(tt_pack_frag16(w[0], NGH, I, wih_max, w16, st));
        } else {
            // (every 128-token workgroup would otherwise convert the tiles of W_ih it touches)
            const int Kp = (I + 31) / 32 * 32;
            TT_RC_CHECK(tt_pack_rows16(w[0], NGH, I, wih_max, w16, w16 + (size_t)NGH * Kp * sizeof(uint16_t), st));
        }
    }
    if (rnn_type == CELL_GRU && gru16_supported(H) && !f32) {
        TT_RC_CHECK(gru16_pack(w[1], H, wmax, wp, st));
    } else {
        hipLaunchKernelGGL(pack_whh_kernel, dim3(96), dim3(256), 0, st, w[1], H, enc_gates(rnn_type), (float *)wp);
        TT_LAUNCH_CHECK();
    }
    return TT_OK;
}

namespace {
__global__ __launch_bounds__(256) void concat_ids_kernel(const int64_t *__restrict__ a, int Ba, int Ta, const int64_t *__restrict__ b,
                                                         int Bb, int Tb, int64_t *__restrict__ out, int T)
{
    const int64_t n = (int64_t)(Ba + Bb) * T;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / T), t = (int)(i % T);
        int64_t v = 0;
        if (r < Ba) {
            if (t < Ta)
                v = a[(int64_t)r * Ta + t];
        } else if (t < Tb) {
            v = b[(int64_t)(r - Ba) * Tb + t];
        }
        out[i] = v;
    }
}
} // namespace

TT_EXPORT int tt_concat_ids_i64(const int64_t *a, int Ba, int Ta, const int64_t *b, int Bb, int Tb, int64_t *out, int T,
                                tt_stream_t stream)
{
    if (Ba < 0 || Bb < 0 || Ta < 0 || Tb < 0 || T < Ta || T < Tb || (Ba > 0 && Ta > 0 && !a) || (Bb > 0 && Tb > 0 && !b) || !out)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_concat_ids_i64: Ba=%d Ta=%d Bb=%d Tb=%d T=%d (or null pointer)", Ba, Ta, Bb, Tb, T);
    const int64_t n = (int64_t)(Ba + Bb) * T;
    if (n == 0)
        return TT_OK;
    const int64_t want = (n + 255) / 256;
    hipLaunchKernelGGL(concat_ids_kernel, dim3((unsigned)(want > 2048 ? 2048 : want)), dim3(256), 0, (hipStream_t)stream, a, Ba, Ta, b, Bb,
                       Tb, out, T);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

TT_EXPORT size_t tt_encoder_prepared_bytes(int E, int H, int num_layers, int bidirectional, int rnn_type)
{
    if (num_layers < 1 || num_layers > ENC_MAX_LAYERS || rnn_type < 0 || rnn_type > 2 || E <= 0 || H <= 0)
        return 0;
    return enc_prepared_layout(E, H, num_layers, bidirectional, rnn_type).total;
}

TT_EXPORT int tt_encoder_prepare_f32(int E, int H, int num_layers, int bidirectional, int rnn_type,
                                     const float *const *weights, void *prepared, size_t prepared_bytes, tt_stream_t stream)
{
    hipStream_t st = (hipStream_t)stream;
    int rc = enc_check_shape("tt_encoder_prepare_f32", 1, 1, E, H, num_layers, 1);
    if (rc != TT_OK)
        return rc;
    if (rnn_type < CELL_GRU || rnn_type > CELL_RNN)
        return tt_fail(TT_ERR_UNSUPPORTED, "tt_encoder_prepare_f32: rnn_type=%d (0 GRU, 1 LSTM, 2 RNN)", rnn_type);
    const EncPrepared pl = enc_prepared_layout(E, H, num_layers, bidirectional, rnn_type);
    if (!weights || !prepared || prepared_bytes < pl.total || ((uintptr_t)prepared & 255))
        return tt_fail(TT_ERR_WORKSPACE, "tt_encoder_prepare_f32: buffer %zu < %zu bytes (or null / not 256-B aligned)",
                       prepared_bytes, pl.total);
    char *pb = (char *)prepared;
    const int ndir = bidirectional ? 2 : 1;
    TT_RC_CHECK(tt_zero_async(pb, 256, st));
    for (int l = 0; l < num_layers; ++l)
        for (int d = 0; d < ndir; ++d) {
            unsigned *words = (unsigned *)pb + 2 * (2 * l + d);
            TT_RC_CHECK(enc_pack_weights(l == 0 ? E : ndir * H, H, rnn_type, weights + ((size_t)l * ndir + d) * 4, words,
                                         words + 1, pb + pl.wih[l][d], pb + pl.wp[l][d], st));
        }
    return TT_OK;
}


// ------------------------------------------------------------------ the projected table (inference)
// With GloVe loaded the table is frozen (backend/model.py:25-27) and at inference W_ih is fixed too, so layer 0's input
// projection of a token depends on its id alone: P[v] = table[v] W_ih^T + b_ih for every vocabulary row v, V x 3H floats per
// direction (400 003 x 768 x 4 B = 1.23 GB for the north-star tower).  It is built by the launch the forward itself would make
// (same kernel, same weight images and scale words, rows taken in order instead of through the packed ids; the token-stationary
// kernel scales every row by its own power of two), so row v holds the bits K1 writes for a token with id v, and the
// recurrence kernels gather their step's rows from it (GruParams::gi_ids): K1 -- half the index build's GPU time, and a
// [tokens][3H] buffer written and read back -- is gone from inference.
static bool enc_projected_supported(int H, int rnn_type)
{
    return rnn_type == CELL_GRU && gru16_supported(H) && !enc_force_f32();
}
static size_t enc_projected_dir_bytes(int64_t V, int H, int rnn_type)
{
    return tt_align_up(sizeof(float) * (size_t)V * enc_gates(rnn_type) * H, 256);
}

static int encoder_forward(const char *who, const int64_t *ids, int B, int T, const float *table, int64_t V, int E, int H,
                           int num_layers, int bidirectional, int rnn_type, const float *const *weights,
                           const void *prepared, const float *proj_w, const float *proj_b, int normalize, int train,
                           float dropout_p, uint64_t dropout_seed, float *out, void *workspace, size_t workspace_bytes,
                           int32_t *status, hipStream_t st, const tt_enc_sync_t *sync = nullptr, const void *projected = nullptr)
{
    int rc = enc_check_shape(who, B, T, E, H, num_layers, V);
    if (rc != TT_OK)
        return rc;
    if (train < 0 || (train & ~(TT_ENC_TRAIN_MASK | TT_ENC_ONE_WORKGROUP | TT_ENC_PHASE_BEGIN | TT_ENC_PHASE_FINISH | TT_ENC_SEED_ON_DEVICE | TT_ENC_F32)) ||
        (train & TT_ENC_TRAIN_MASK) > 2 || ((train & TT_ENC_PHASE_BEGIN) && (train & TT_ENC_PHASE_FINISH)))
        return tt_fail(TT_ERR_BAD_SHAPE, "%s: train=0x%x (0, 1 or 2, optionally | TT_ENC_ONE_WORKGROUP | one TT_ENC_PHASE_* | TT_ENC_F32)", who, train);
    const bool force_f32 = enc_force_f32(train); // every product of this call on the fp32-MFMA kernels
    if (force_f32)
        prepared = nullptr; // (the prepared images are the f16-split kernels'; this path derives its own in the workspace)
    const bool one_wg = (train & TT_ENC_ONE_WORKGROUP) != 0; // the caller keeps the recurrences off the column-split kernels
    // the call in two halves (include/tt.h): BEGIN = everything in front of the first recurrence launch, FINISH = the rest
    const bool ph_begin = (train & TT_ENC_PHASE_BEGIN) != 0, ph_finish = (train & TT_ENC_PHASE_FINISH) != 0;
    // the dropout seed as the address of a device word the kernels read when they run (a captured step replays with new masks)
    const uint64_t *seed_dev = (train & TT_ENC_SEED_ON_DEVICE) ? (const uint64_t *)(uintptr_t)dropout_seed : nullptr;
    train &= TT_ENC_TRAIN_MASK;
    if (!ids || (!table && !projected) || !weights || !out || (bidirectional && (!proj_w || !proj_b)))
        return tt_fail(TT_ERR_BAD_SHAPE, "%s: null pointer", who);
    if (projected && (train || ph_begin || ph_finish || force_f32 || !enc_projected_supported(H, rnn_type)))
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: a projected table serves inference calls of the f16-split GRU (H = 128, 256) only", who);
    if (!(dropout_p >= 0.0f && dropout_p < 1.0f))
        return tt_fail(TT_ERR_BAD_SHAPE, "%s: dropout_p=%g", who, dropout_p);
    if (rnn_type < CELL_GRU || rnn_type > CELL_RNN)
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: rnn_type=%d (0 GRU, 1 LSTM, 2 RNN)", who, rnn_type);
    const int NGH = enc_gates(rnn_type) * H;
    const bool drop = train && dropout_p > 0.0f && num_layers > 1;
    const EncLayout lo = enc_layout(B, T, E, H, num_layers, bidirectional, train, drop, rnn_type, projected != nullptr);
    if (!workspace || workspace_bytes < lo.total || ((uintptr_t)workspace & 255))
        return tt_fail(TT_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes (or not 256-B aligned)", who, workspace_bytes,
                       lo.total);
    char *ws = (char *)workspace;
    int32_t *len = (int32_t *)(ws + lo.len), *tok_off = (int32_t *)(ws + lo.tok_off);
    int32_t *perm = (int32_t *)(ws + lo.perm), *idsp = (int32_t *)(ws + lo.ids), *flag = (int32_t *)(ws + lo.flag);
    const int ndir = lo.ndir;

    // (training: the backward's "previous token" maps are made here -- they depend on the lengths only)
    int32_t *pm_fwd = train ? (int32_t *)(ws + lo.prevmap[0]) : nullptr;
    int32_t *pm_rev = (train && ndir == 2) ? (int32_t *)(ws + lo.prevmap[1]) : nullptr;
    // Layer 0's zero fills ride on ONE launch in front of the prep kernels (each was its own ~5 us launch on the call's chain):
    // the status flags (when the prep kernels do not clear them themselves), the all-zero row behind layer 0's output
    // (training) and the split recurrence's hand-off slots.
    const bool use16_early = rnn_type == CELL_GRU && gru16_supported(H) && !force_f32;
    const bool split0 = use16_early && lo.xch && !one_wg && gru16x4_usable(B, H, ndir);
    const bool fused_prep = B <= 1024 && (int64_t)B * T <= PREP_FUSED_MAX_IDS;
    if (!ph_finish) {
        float *x1 = (num_layers == 1 && !train) ? nullptr : (float *)(ws + lo.x[1]);
        TT_RC_CHECK(tt_zero3_async(fused_prep ? nullptr : (void *)flag, fused_prep ? 0 : 256,
                                   train ? (void *)(x1 + (size_t)lo.MT * ndir * H) : nullptr, train ? sizeof(float) * ndir * H : 0,
                                   split0 ? (void *)(ws + lo.xch) : nullptr, split0 ? gru16x4_xch_bytes(B, H, ndir) : 0, st));
    }
    if (ph_finish) {
        // (the BEGIN half made lengths, packed ids, maps and layer 0's projections in this workspace)
    } else if (fused_prep) {
        hipLaunchKernelGGL(prep_fused_kernel, dim3(1), dim3(1024), 0, st, ids, B, T, V, len, flag, tok_off, perm, idsp, status,
                           pm_fwd, pm_rev, (int)lo.MT);
        TT_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(prep_len_kernel, dim3((B + 3) / 4), dim3(256), 0, st, ids, B, T, V, len, flag);
        hipLaunchKernelGGL(prep_scan_sort_kernel, dim3(1), dim3(1024), 0, st, len, B, T, tok_off, perm);
        hipLaunchKernelGGL(prep_pack_ids_kernel, dim3(B), dim3(256), 0, st, ids, B, T, len, tok_off, V, idsp, flag, status,
                           pm_fwd, pm_rev, (int)lo.MT);
        TT_LAUNCH_CHECK();
    }

    const size_t lds = sizeof(float) * 2 * ENC_RB * (H + 4);
    const bool use16 = rnn_type == CELL_GRU && gru16_supported(H) && !force_f32;
    const EncPrepared pl = enc_prepared_layout(E, H, num_layers, bidirectional, rnn_type);
    const char *pb = (const char *)prepared;
    for (int l = 0; l < num_layers; ++l) {
        const int I = l == 0 ? E : ndir * H;
        GruParams gp;
        gp.len = len;
        gp.tok_off = tok_off;
        gp.perm = perm;
        gp.B = B;
        gp.H = H;
        gp.out_ld = ndir * H;
        gp.slots = 2 * enc_cus();
        const bool last = l == num_layers - 1;
        float *xout = (last && !train) ? nullptr : (float *)(ws + lo.x[l + 1]);
        if (train && l > 0) // the all-zero row that stands for "h before the first step" in the backward GEMMs (layer 0: above)
            TT_RC_CHECK(tt_zero_async(xout + (size_t)lo.MT * ndir * H, sizeof(float) * ndir * H, st));
        for (int d = 0; d < ndir; ++d) {
            const float *const *w = weights + ((size_t)l * ndir + d) * 4;
            SgemmParams g;
            g.A = l == 0 ? table : (const float *)(ws + ((drop && l > 0) ? lo.xd[l] : lo.x[l]));
            g.a_map = l == 0 ? idsp : nullptr;
            g.B = w[0];
            g.b_map = nullptr;
            g.C = (float *)(ws + lo.gi[d]);
            g.bias = w[2];
            g.m_dyn = tok_off + B;
            g.k_dyn = nullptr;
            g.M = (int)lo.MT;
            g.N = NGH;
            g.K = I;
            g.lda = I;
            g.ldb = I;
            g.ldc = NGH;
            g.slab_stride = 0;
            g.accumulate = 0;
            // the weights in kernel form: from the caller's prepared buffer, or derived here into the workspace
            unsigned *wih_max, *wmax;
            char *w16, *wp;
            if (pb) {
                wih_max = (unsigned *)pb + 2 * (2 * l + d);
                wmax = wih_max + 1;
                w16 = (char *)pb + pl.wih[l][d];
                wp = (char *)pb + pl.wp[l][d];
            } else {
                wih_max = (unsigned *)flag + 40 + 2 * l + d; // in the status block cleared above
                wmax = (unsigned *)flag + 16 + 2 * l + d;
                w16 = ws + lo.wih16[d];
                wp = ws + lo.wp[d];
                if (!(ph_finish && l == 0)) {
                    rc = enc_pack_weights(I, H, rnn_type, w, wih_max, wmax, w16, wp, st, force_f32);
                    if (rc != TT_OK)
                        return rc;
                }
            }
            if (ph_finish && l == 0) {
                rc = TT_OK; // (layer 0's projection was launched by the BEGIN half)
            } else if (projected && l == 0) {
                rc = TT_OK; // (every vocabulary row's projection is in the table: the recurrence gathers them by token id)
            } else if (!force_f32) {
                // K1 on the f16 pipes (fp16 hi/lo split, fp32-grade; sgemm.h): W_ih is scaled by the power of two
                // that puts its largest element in [2^13, 2^14); deeper layers' A rows are hidden states in (-1, 1) (times
                // 1/(1-p) under dropout): 2^6; layer 0's are embedding vectors of ANY magnitude: the token-stationary
                // kernel scales every row by its own power of two (the tiled kernel, for the shapes that one does not take,
                // takes them as they are: fp32-grade for |x|max in ~[0.1, 6e4])
                g.a_absmax = nullptr;
                g.a_exp = l == 0 ? 0 : 6;
                g.b_absmax = wih_max;
                g.b_exp = 0;
                g.b_hi16 = w16;
                unsigned *xmax = (train && l == 0) ? (unsigned *)flag + ENC_FLAG_XMAX : nullptr; // for the backward's dW_ih
                if (enc_rows16(NGH, I)) { // token-stationary form (gemm_rows16.hip): W_ih as a fragment stream
                    g.b_lo16 = nullptr;
                    g.ldb16 = 0;
                    g.a_row_scale = l == 0;
                    g.a_absmax_out = d == 0 ? xmax : nullptr;
                    rc = tt_gemm_rows16(g, st);
                } else {
                    if (xmax && d == 0)
                        TT_RC_CHECK(tt_absmax_rows(table, E, E, idsp, (int)lo.MT, tok_off + B, xmax, st));
                    const int Kp = (I + 31) / 32 * 32;
                    g.b_lo16 = w16 + (size_t)NGH * Kp * sizeof(uint16_t);
                    g.ldb16 = Kp;
                    rc = tt_sgemm16(g, false, false, 1, st);
                }
            } else {
                rc = tt_sgemm(g, false, false, 1, st);
            }
            if (rc != TT_OK)
                return rc;
            gp.dir[d].gi = (const float *)(ws + lo.gi[d]);
            if (projected && l == 0) {
                gp.dir[d].gi = (const float *)((const char *)projected + (size_t)d * enc_projected_dir_bytes(V, H, rnn_type));
                gp.gi_ids = idsp;
                gp.gi_rows = (unsigned)V;
            }
            gp.dir[d].wp = (const float *)wp;
            gp.dir[d].wmax = wmax;
            gp.dir[d].b_hh = w[3];
            gp.dir[d].out_seq = xout;
            gp.dir[d].gates = (train && rnn_type != CELL_RNN) ? (float *)(ws + lo.gates[l][d]) : nullptr;
            gp.dir[d].cseq = nullptr;
            if (train && rnn_type == CELL_LSTM) { // c before the first step: the all-zero row at index MT
                gp.dir[d].cseq = (float *)(ws + lo.cseq[l][d]);
                TT_RC_CHECK(tt_zero_async(gp.dir[d].cseq + (size_t)lo.MT * H, sizeof(float) * H, st));
            }
            // only the last layer's final hidden state is used (model.py:65-71)
            gp.dir[d].h_final = (float *)(ws + lo.hfin) + (size_t)d * B * H;
            gp.dir[d].out_col0 = d * H;
            gp.dir[d].reverse = d;
        }
        if (ndir == 1)
            gp.dir[1] = gp.dir[0];
        if (ph_begin) { // everything in front of the first recurrence launch is in the queue
            TT_LAUNCH_CHECK();
            return TT_OK;
        }
        if (l == 0 && sync && sync->wait_before_recurrence) // (include/tt.h: the host orders this call's recurrences behind another call's)
            TT_HIP_CHECK(hipStreamWaitEvent(st, (hipEvent_t)sync->wait_before_recurrence, 0));
        if (use16 && lo.xch && !one_wg && gru16x4_usable(B, H, ndir)) {
            // a row group's gate columns on four CUs (gru16x4.hip): same bits out, ~half the time per step
            rc = gru16x4_launch(gp, ndir, ws + lo.xch, status, st, /*xch_zeroed=*/l == 0 && split0);
            if (rc != TT_OK)
                return rc;
        } else if (use16) {
            rc = gru16_launch(gp, ndir, st);
            if (rc != TT_OK)
                return rc;
        } else {
            rc = rnn_type == CELL_LSTM ? launch_seq<CELL_LSTM>(gp, B, H, ndir, lds, st)
                                       : (rnn_type == CELL_RNN ? launch_seq<CELL_RNN>(gp, B, H, ndir, lds, st)
                                                               : launch_seq<CELL_GRU>(gp, B, H, ndir, lds, st));
            if (rc != TT_OK)
                return rc;
        }
        TT_LAUNCH_CHECK();
        if (last && sync && sync->record_after_recurrence)
            TT_HIP_CHECK(hipEventRecord((hipEvent_t)sync->record_after_recurrence, st));
        if (drop && !last) { // nn.GRU's dropout sits on the outputs of every layer but the last
            hipLaunchKernelGGL(dropout_apply_kernel, dim3(B, B >= 2048 ? 1 : (B >= 512 ? 4 : 16)), dim3(256), 0, st, (const float *)xout,
                               (float *)(ws + lo.xd[l + 1]), len, tok_off, T, ndir * H, l, dropout_p, seed_dev ? 0ull : dropout_seed,
                               seed_dev);
            TT_RC_CHECK(tt_zero_async((float *)(ws + lo.xd[l + 1]) + (size_t)lo.MT * ndir * H, sizeof(float) * ndir * H, st));
            TT_LAUNCH_CHECK();
        }
    }
    if (ndir == 2 && B >= 2048) { // (below that the two launches cost what the per-row loop does: ~50 us at B = 512 .. 1024)
        // Linear(2H, H) on cat(h_fwd, h_bwd) (model.py:65-69) as two accumulating GEMMs over the two halves of the weight
        // (the head kernel's per-row scalar loop read all of proj_w once per batch row: 0.6 ms at B = 8192), then the head
        // kernel only normalises
        float *hid = (float *)(ws + lo.hid);
        for (int d = 0; d < 2; ++d) {
            SgemmParams g;
            g.A = (const float *)(ws + lo.hfin) + (size_t)d * B * H;
            g.B = proj_w + (size_t)d * H;
            g.C = hid;
            g.bias = d == 0 ? proj_b : nullptr;
            g.a_map = g.b_map = nullptr;
            g.m_dyn = g.k_dyn = nullptr;
            g.M = B;
            g.N = H;
            g.K = H;
            g.lda = H;
            g.ldb = 2 * H;
            g.ldc = H;
            g.slab_stride = 0;
            g.accumulate = d;
            rc = tt_sgemm(g, false, false, 1, st);
            if (rc != TT_OK)
                return rc;
        }
        hipLaunchKernelGGL(head_kernel, dim3(B), dim3(256), 0, st, (const float *)hid, B, H, 1, proj_w, proj_b, normalize,
                           (float *)nullptr, out);
    } else {
        hipLaunchKernelGGL(head_kernel, dim3(B), dim3(256), 0, st, (const float *)(ws + lo.hfin), B, H, ndir, proj_w,
                           proj_b, normalize, train ? (float *)(ws + lo.hid) : (float *)nullptr, out);
    }
    TT_LAUNCH_CHECK();
    return TT_OK;
}

// Workgroups (= CUs: one each) the column-split recurrence of a call of this shape occupies, 0 when the call runs the
// one-workgroup kernels anyway.  For hosts that keep several encoder calls in flight (include/tt.h).
TT_EXPORT int tt_encoder_split_workgroups(int B, int H, int bidirectional, int rnn_type)
{
    const int ndir = bidirectional ? 2 : 1;
    if (rnn_type != CELL_GRU || !gru16_supported(H) || enc_force_f32() || gru16x4_xch_bytes(B, H, ndir) == 0 ||
        !gru16x4_usable(B, H, ndir))
        return 0;
    return ((B + ENC_RB - 1) / ENC_RB + 7) / 8 * 32 * (gru16x4_launches(B, H, ndir) == 2 ? 1 : ndir); // (per launch)
}

TT_EXPORT int tt_encoder_forward_f32(const int64_t *ids, int B, int T, const float *table, int64_t V, int E, int H,
                                     int num_layers, int bidirectional, int rnn_type, const float *const *weights,
                                     const float *proj_w, const float *proj_b, int normalize, int train,
                                     float dropout_p, uint64_t dropout_seed, float *out, void *workspace,
                                     size_t workspace_bytes, int32_t *status, const tt_enc_sync_t *sync, tt_stream_t stream)
{
    return encoder_forward("tt_encoder_forward_f32", ids, B, T, table, V, E, H, num_layers, bidirectional, rnn_type, weights,
                           nullptr, proj_w, proj_b, normalize, train, dropout_p, dropout_seed, out, workspace,
                           workspace_bytes, status, (hipStream_t)stream, sync);
}

TT_EXPORT int tt_encoder_forward_prepared_f32(const int64_t *ids, int B, int T, const float *table, int64_t V, int E, int H,
                                              int num_layers, int bidirectional, int rnn_type,
                                              const float *const *weights, const void *prepared, const float *proj_w,
                                              const float *proj_b, int normalize, int opts, float *out, void *workspace,
                                              size_t workspace_bytes, int32_t *status, tt_stream_t stream)
{
    if (!prepared || ((uintptr_t)prepared & 255))
        return tt_fail(TT_ERR_WORKSPACE, "tt_encoder_forward_prepared_f32: prepared buffer null or not 256-B aligned");
    if (opts & ~(TT_ENC_ONE_WORKGROUP | TT_ENC_F32))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_encoder_forward_prepared_f32: opts=0x%x (0, TT_ENC_ONE_WORKGROUP, TT_ENC_F32)", opts);
    return encoder_forward("tt_encoder_forward_prepared_f32", ids, B, T, table, V, E, H, num_layers, bidirectional, rnn_type,
                           weights, prepared, proj_w, proj_b, normalize, /*train=*/0 | opts, 0.0f, 0, out, workspace,
                           workspace_bytes, status, (hipStream_t)stream);
}

TT_EXPORT size_t tt_encoder_projected_bytes(int64_t V, int E, int H, int bidirectional, int rnn_type)
{
    if (V <= 0 || V >= INT_MAX || E < 4 || (E & 3) || rnn_type < 0 || rnn_type > 2 || !enc_projected_supported(H, rnn_type))
        return 0;
    return (bidirectional ? 2 : 1) * enc_projected_dir_bytes(V, H, rnn_type);
}

TT_EXPORT int tt_encoder_project_table_f32(const float *table, int64_t V, int E, int H, int num_layers, int bidirectional,
                                           int rnn_type, const float *const *weights, const void *prepared, void *projected,
                                           size_t projected_bytes, tt_stream_t stream)
{
    hipStream_t st = (hipStream_t)stream;
    int rc = enc_check_shape("tt_encoder_project_table_f32", 1, 1, E, H, num_layers, V);
    if (rc != TT_OK)
        return rc;
    const size_t need = tt_encoder_projected_bytes(V, E, H, bidirectional, rnn_type);
    if (need == 0)
        return tt_fail(TT_ERR_UNSUPPORTED, "tt_encoder_project_table_f32: rnn_type=%d H=%d has no projected table (f16-split GRU: H = 128, 256)", rnn_type, H);
    if (!table || !weights || !prepared || ((uintptr_t)prepared & 255))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_encoder_project_table_f32: null pointer (or prepared not 256-B aligned)");
    if (!projected || projected_bytes < need || ((uintptr_t)projected & 255))
        return tt_fail(TT_ERR_WORKSPACE, "tt_encoder_project_table_f32: buffer %zu < %zu bytes (or null / not 256-B aligned)",
                       projected_bytes, need);
    const int ndir = bidirectional ? 2 : 1, NGH = enc_gates(rnn_type) * H;
    const EncPrepared pl = enc_prepared_layout(E, H, num_layers, bidirectional, rnn_type);
    const char *pb = (const char *)prepared;
    for (int d = 0; d < ndir; ++d) {
        const float *const *w = weights + (size_t)d * 4; // layer 0
        SgemmParams g;
        g.A = table;
        g.a_map = nullptr; // rows in order: row v of the result is what a token with id v gets
        g.B = w[0];
        g.b_map = nullptr;
        g.C = (float *)((char *)projected + (size_t)d * enc_projected_dir_bytes(V, H, rnn_type));
        g.bias = w[2];
        g.m_dyn = nullptr;
        g.k_dyn = nullptr;
        g.M = (int)V;
        g.N = NGH;
        g.K = E;
        g.lda = E;
        g.ldb = E;
        g.ldc = NGH;
        g.slab_stride = 0;
        g.accumulate = 0;
        g.a_absmax = nullptr;
        g.a_exp = 0;
        g.b_absmax = (const unsigned *)pb + 2 * d; // layer 0, direction d: max |W_ih| (tt_encoder_prepare_f32)
        g.b_exp = 0;
        g.b_hi16 = pb + pl.wih[0][d];
        if (enc_rows16(NGH, E)) {
            g.b_lo16 = nullptr;
            g.ldb16 = 0;
            g.a_row_scale = 1;
            g.a_absmax_out = nullptr;
            rc = tt_gemm_rows16(g, st);
        } else {
            const int Kp = (E + 31) / 32 * 32;
            g.b_lo16 = pb + pl.wih[0][d] + (size_t)NGH * Kp * sizeof(uint16_t);
            g.ldb16 = Kp;
            rc = tt_sgemm16(g, false, false, 1, st);
        }
        if (rc != TT_OK)
            return rc;
    }
    return TT_OK;
}

TT_EXPORT int tt_encoder_forward_projected_f32(const int64_t *ids, int B, int T, const void *projected, int64_t V, int E, int H,
                                               int num_layers, int bidirectional, int rnn_type,
                                               const float *const *weights, const void *prepared, const float *proj_w,
                                               const float *proj_b, int normalize, int opts, float *out, void *workspace,
                                               size_t workspace_bytes, int32_t *status, tt_stream_t stream)
{
    if (!prepared || ((uintptr_t)prepared & 255) || !projected || ((uintptr_t)projected & 255))
        return tt_fail(TT_ERR_WORKSPACE, "tt_encoder_forward_projected_f32: prepared / projected buffer null or not 256-B aligned");
    if (opts & ~TT_ENC_ONE_WORKGROUP)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_encoder_forward_projected_f32: opts=0x%x (0 or TT_ENC_ONE_WORKGROUP)", opts);
    return encoder_forward("tt_encoder_forward_projected_f32", ids, B, T, nullptr, V, E, H, num_layers, bidirectional, rnn_type,
                           weights, prepared, proj_w, proj_b, normalize, /*train=*/0 | opts, 0.0f, 0, out, workspace,
                           workspace_bytes, status, (hipStream_t)stream, nullptr, projected);
}
