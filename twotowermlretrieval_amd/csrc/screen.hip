// Screened exact top-k (gfx950): the path BruteForceIndex(screen=True) takes at every batch size.
//
// Exact fp32 scoring is bound by v_mfma_f32_32x32x2_f32 (157 TFLOP/s, 1/16 of the f16 MFMA rate).
// This path gets the SAME bit-exact result (oracle/tt_oracle.c:o_score_topk order and scores)
// through a rigorous filter:
//
//   1. screen   approximate scores s16 = <fp16(q), fp16(d)> with fp32 accumulation on
//               v_mfma_f32_16x16x32_f16 against an fp16 shadow copy of the corpus.  For every pair
//                   |s16 - s| <= eps_q  with  eps_q = 1.10e-3 |q| Dmax + 1e-6 (|q| + Dmax)
//               (fp16 rounding 2^-11 per operand, exact products, fp32 summation of 256 terms on
//               both sides, fp16 underflow; Dmax = largest document L2 norm; derivation in
//               DESIGN.md).  A document can be in the exact top-k only if
//               s16 >= A_k - 2 eps_q, where A_k is the k-th largest approximate score seen by
//               ANY subset of the corpus, so each workgroup keeps, per query, every candidate
//               within 2 eps_q of its running k-th best approximate score.
//   2. finish   per query: pool all candidates, A_k over the whole corpus, survivors
//               {s16 >= A_k - 2 eps_q}, EXACT fp32 FMA-chain rescoring of the survivors from the
//               fp32 corpus, exact top-k with (score desc, index asc).
//
// Anything that would break the guarantee (candidate buffer or survivor list overflow -- e.g.
// hundreds of near-duplicate documents around the k-th score --, |q| too large for fp16) raises a
// device flag; the caller then runs the plain exact kernel, predicated on that flag, so no host
// synchronisation is needed and the result is exact in every case.
//
// Three kernels share that scheme and the finish kernel:
//   q_image_kernel        once per search: queries -> f16 MFMA B-operand image, norms, initial flags
//   screen_kernel<.,NSET> B > 64: one workgroup (8 waves, one per CU) = 128 NSET queries x a contiguous chunk of
//                         documents; each wave keeps 16 NSET queries in registers; document tiles (32 docs x 256
//                         features f16 = 16 KiB) are DMA'd once per workgroup into an 8-deep LDS ring
//                         (global_load_lds, XOR-swizzled source) and read by all 8 waves
//   screen_stream_kernel  B <= 64: every wave an independent streaming engine (32 or 64 queries) with a private 4-slab ring; bound by
//                         HBM streaming of the fp16 copy (N * 512 B)
// The accumulators start at minus the query's threshold, so "any candidate in this wave-tile?" is one integer max.
#include "tt_common.h"
#include <cmath>

#include <hip/hip_fp16.h>
#include <limits.h>
#include <math.h>

int tt_score_topk_f32_pred(const float *Q, int B, int d, const float *D, int64_t N, int k, int64_t idx_offset,
                           float *out_val, int64_t *out_idx, void *workspace, size_t workspace_bytes,
                           const int *run_if, hipStream_t st);

int tt_kth_largest(const float *vals, int B, int M, int k, float *out, hipStream_t st);
int tt_k_largest_list(const float *vals, int B, int M, int k, float *list, hipStream_t st);

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

#ifndef TT_SCREEN_STAGGER
#define TT_SCREEN_STAGGER 1
#endif
constexpr int SW = 8;                 // waves per workgroup
constexpr int STILE_BYTES = 32 * 512; // 32 docs x 256 f16
constexpr int SRING = 8;              // ring depth in tiles
constexpr int STPB = 2;               // tiles per barrier interval (waves drift freely inside one)
constexpr int SCAP = 128;             // candidate entries per (workgroup, query)
constexpr int SURV_MAX = 1024;        // survivors per query the finish kernel can rescore (256 until round 4: a cluster of ~200
                                      // near-duplicates plus a group of exact duplicates overflowed it, and on a clustered
                                      // corpus the predicated exact kernel then cost more than the screen saved: DESIGN K4s)
constexpr int POOL_MAX = 8192;        // candidates per query the finish kernel can pool
constexpr int FIN_MAX_CHUNKS = 2048;  // document chunks per query the finish kernel can pool

struct SCand {
    float v;
    int x;
};

// The derivation gives (2^-10 + 2^-22 + 2 * 1.53e-5 + threshold-in-accumulator terms) = 1.023e-3 (DESIGN 4 "K4s"); the worst
// input family the hardware test can build reaches 0.892 of 1.05e-3 (tests/test_screen_bound_gpu.py).  1.10e-3 leaves 7.5 % over
// the derivation instead of 2.6 %: a violation has no fallback (a true top-k document would be lost silently), and the
// wider slack costs about one more survivor per query.
__device__ __forceinline__ float screen_eps(float qnorm, float dmax)
{
    return 1.10e-3f * qnorm * dmax + 1e-6f * (qnorm + dmax);
}

__device__ __forceinline__ SCand scand_load_l2(const SCand *p)
{
    unsigned long long u = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    SCand c;
    c.v = __uint_as_float((unsigned)u);
    c.x = (int)(u >> 32);
    return c;
}

struct ScreenParams {
    const float *Q;
    const _Float16 *D16;
    int B, N, k;
    int n_chunks, tiles_per_chunk, n_tiles;
    // Shared-tile main pass: tiles [0, static_tiles) are split evenly over the chunks; tiles [static_tiles, n_tiles) are a
    // POOL of tail_blocks blocks of tail_g tiles per query group that the workgroups draw from (tail_ctr[qgroup],
    // atomicAdd) once their own range is done.  static_tiles == n_tiles, tail_blocks == 0: everything static.
    int static_tiles, tail_g, tail_blocks;
    int *tail_ctr; // [n_qgroups], zeroed by q_image_kernel
    float dmax;
    SCand *cand;   // [n_blocks][512][SCAP]
    int *pcnt;     // [rows_pad][n_chunks]
    int *flag;     // overflow / unsupported -> exact fallback
    // sample pass (MAXONLY): per-(tile, query) maximum approximate score
    float *max_val;      // [rows_pad][n_tiles]
    const float *thr0;   // main pass: k-th largest sample maximum per query, stride thr0_stride (or null)
    int thr0_stride;
    // queries as MFMA B operands, prepared once per search by q_image_kernel:
    // qimg[((S * 8 + s) * 64 + lane)] = 8 f16 of query 16 S + (lane & 15), features 32 s + 8 (lane >> 4) .. +7
    const h8 *qimg;
    const float *qnorm;  // |q| per (padded) query row
    // test-only (tt_debug_screen_s16): MAXONLY pass whose accumulators start at -dbg_thr[query] like the main
    // pass's, so the raw value t = fl(sum - thr) the filter compares with +0 can be observed; null in the product
    const float *dbg_thr;
};

// store (v, x) at wave-uniform base + per-lane 32-bit byte offset (SGPR-base addressing: no 64-bit VALU math)
__device__ __forceinline__ void scand_store_async(const SCand *base, unsigned byte_off, float v, int x)
{
    const unsigned long long bits = ((unsigned long long)(unsigned)x << 32) | (unsigned long long)__float_as_uint(v);
    asm volatile("global_store_dwordx2 %0, %1, %2\n\ts_nop 1" ::"v"(byte_off), "v"(bits), "s"(base) : "memory");
}

__device__ __forceinline__ void f32_store_async(float *dst, float v)
{
    asm volatile("global_store_dword %0, %1, off\n\ts_nop 0" ::"v"(dst), "v"(v) : "memory");
}

// A query's 128-entry buffer is four 32-entry quarters, one per lane that holds scores of that query
// (16x16x32 MFMA: query n of a 16-query set sits in lanes n, n+16, n+32, n+48): lane (n,g) appends to
// quarter g with its OWN counter, so the append pass needs no cross-lane traffic at all.
// Compaction: keep every entry within `slack` of the k-th best of the union; rank r goes to quarter r&3,
// slot r>>2 (keeps the quarters balanced).  Lane t owns entries t and 64+t of the buffer.
constexpr int SQUART = SCAP / 4;                 // 32 entries per quarter
constexpr int SQ_TRIGGER = SQUART - 10;          // a tile adds at most 8 entries per lane: compact above this
constexpr int SQ_KEEP_MAX = 4 * SQ_TRIGGER - 4;  // more kept entries than this could not take another tile

__device__ __forceinline__ void screen_compact(SCand *base, const int (&n)[4], int k, float slack, int lane, int &n_new,
                                               float &thr_new, bool &have, bool &overflow)
{
    float v[2];
    int x[2], rank[2];
    bool live[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        v[i] = -INFINITY;
        x[i] = INT_MAX;
        rank[i] = 0;
        live[i] = (lane & 31) < ((lane >> 5) ? n[2 * i + 1] : n[2 * i]);
        if (live[i]) {
            const SCand c = scand_load_l2(base + 64 * i + lane);
            v[i] = c.v;
            x[i] = c.x;
        }
    }
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int lim = n[2 * i2 + sub];
            for (int l2 = 0; l2 < lim; ++l2) {
                const float sv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i2]), 32 * sub + l2));
                const int sx = __builtin_amdgcn_readlane(x[i2], 32 * sub + l2);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    rank[i] += (sv > v[i] || (sv == v[i] && sx < x[i])) ? 1 : 0;
            }
        }
    }
    const int total = n[0] + n[1] + n[2] + n[3];
    have = total >= k;
    overflow = false;
    n_new = total;
    thr_new = -INFINITY;
    if (have) {
        float kth = -INFINITY;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned long long bk = __ballot(live[i] && rank[i] == k - 1);
            if (bk)
                kth = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i]), __ffsll((long long)bk) - 1));
        }
        thr_new = kth - slack;
        n_new = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            n_new += __popcll(__ballot(live[i] && v[i] >= thr_new));
        if (n_new > SQ_KEEP_MAX) { // a quarter could not take one more tile: exact fallback (this pass's result is discarded)
            overflow = true;
            n_new = k;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
        if (live[i] && rank[i] < n_new) {
            SCand c;
            c.v = v[i];
            c.x = x[i];
            base[SQUART * (rank[i] & 3) + (rank[i] >> 2)] = c;
        }
}

// Once per search: the queries in the register image the screen kernels keep as MFMA B operands (one coalesced
// 1-KiB load per k-step and set instead of 64 scattered 16-byte row reads in every workgroup's prologue: the
// per-workgroup set-up was ~40 us of a 0.7 ms shard step, paid in the sample pass and again in the main pass),
// their norms, and the initial fallback flags (2 = fp16 cannot hold a query of this 32-query tile, else 0).
// One workgroup of two waves per 32-query tile, one wave per 16-query set; rows >= B read as zeros.
__global__ __launch_bounds__(128) void q_image_kernel(const float *__restrict__ Q, int B, h8 *__restrict__ img,
                                                      float *__restrict__ qnorm, int *__restrict__ flag, int n_flags,
                                                      int *__restrict__ tail_ctr, int n_ctr)
{
    __shared__ int any_bad[2];
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < n_ctr; i += 128)
            tail_ctr[i] = 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int S = blockIdx.x * 2 + wv;
    const int g = lane >> 4, n = lane & 15;
    const int qrow = 16 * S + n;
    const bool live = qrow < B;
    const float *qp = Q + (size_t)min(qrow, B - 1) * 256 + 8 * g;
    float ss = 0.0f;
    bool bad = false;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const f32x4 a = *(const f32x4 *)(qp + 32 * s);
        const f32x4 b = *(const f32x4 *)(qp + 32 * s + 4);
        h8 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x0 = live ? a[e] : 0.0f, x1 = live ? b[e] : 0.0f;
            ss += x0 * x0 + x1 * x1;
            bad |= !(fabsf(x0) <= 60000.0f) || !(fabsf(x1) <= 60000.0f);
            hv[e] = (_Float16)x0;
            hv[4 + e] = (_Float16)x1;
        }
        img[((size_t)S * 8 + s) * 64 + lane] = hv;
    }
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    if (g == 0)
        qnorm[qrow] = sqrtf(ss);
    const bool wave_bad = __ballot(bad && live) != 0ull;
    if (lane == 0)
        any_bad[wv] = wave_bad ? 1 : 0;
    __syncthreads();
    if (threadIdx.x == 0 && (int)blockIdx.x < n_flags)
        flag[blockIdx.x] = (any_bad[0] | any_bad[1]) ? 2 : 0;
}

// NSET = 16-query sets per wave: 4 (512 queries per workgroup) for large batches; 2 / 1 (256 / 128 queries per
// workgroup) spread a mid-size batch over all eight waves instead of leaving most of them without queries; 3 (384 per
// workgroup, round 4) for the batches a 512-query group would leave a quarter or more empty: B = 257 .. 384 (one group) and
// 513 .. 768 (two groups of 384 instead of a full one and a nearly empty one -- a group costs a pass whatever it holds: B = 513
// took 3.6 ms where 512 took 2.1, profiles/r04_p_batch_sweep.log).
template <bool MAXONLY, int NSET>
__global__ __launch_bounds__(SW * 64, 2) void screen_kernel(ScreenParams p)
{
    extern __shared__ __attribute__((aligned(16))) char ring[]; // [SRING][STILE_BYTES]
    // NSET < 4 (B <= 256): one query group, so every tile is read once by one workgroup: nt policy like the streaming
    // form (TSTREAM_AUX; here -2.5 % at B = 33 .. 128, -1.5 % at 256, A/B on one box); NSET == 4 may have two groups per
    // chunk that share the tile through L2: default policy
    constexpr int DMA_AUX = NSET < 3 ? 2 : 0;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = blockIdx.x % p.n_chunks;
    const int qgroup = blockIdx.x / p.n_chunks;
    int t0 = chunk * p.tiles_per_chunk;                   // this workgroup's own range first, then blocks of the pool
    int t1 = min(t0 + p.tiles_per_chunk, p.static_tiles);
    __shared__ int next_block;
    const int k = p.k;
    const int g = lane >> 4, n = lane & 15;
    constexpr int QW = 16 * NSET, QB = SW * QW; // queries per wave / per workgroup
    const int qbase = qgroup * QB + w * QW;
    const bool wave_live = qbase < p.B;

    // ---- query operands (B of v_mfma_f32_16x16x32_f16): set c holds queries qbase + 16c + n;
    //      lane (n,g) keeps features 32s + 8g .. +7 of k-step s ----
    // Main pass: the accumulators start at -thr (the MFMA's C operand of the first k-step is negthr[c]), so a
    // score passes its query's threshold iff its sign bit is clear and ONE integer max over a wave-tile's 32
    // accumulator registers decides whether anything in the tile needs a second look.  thr is always finite:
    // without a seed it is a lower bound of every possible score, -(1.01 |q| Dmax).
    h8 qreg[NSET][8];
    float eps2[NSET];
    f32x4 negthr[NSET];
    int cnt[NSET];
#pragma unroll
    for (int c = 0; c < NSET; ++c)
        cnt[c] = 0;
#pragma unroll
    for (int c = 0; c < NSET; ++c) {
        const int qrow = qbase + 16 * c + n;
        const bool live = qrow < p.B;
        float qn = 0.0f;
        if (wave_live) { // (a wave without queries only helps with the DMA)
            const h8 *src = p.qimg + ((size_t)(qbase / 16 + c) * 8) * 64 + lane;
#pragma unroll
            for (int s = 0; s < 8; ++s)
                qreg[c][s] = src[s * 64];
            qn = p.qnorm[qrow];
        }
        eps2[c] = 2.0f * screen_eps(qn, p.dmax);
        // A_k over any subset of the corpus, minus 2 eps, never exceeds the approximate score of a
        // true top-k document: the sample pass's k-th largest maximum seeds the threshold.
        const float floor_thr = -(1.01f * qn * p.dmax + 1e-30f);
        const float t_init = live ? ((!MAXONLY && p.thr0) ? fmaxf(p.thr0[(size_t)qrow * p.thr0_stride + p.thr0_stride - 1] - eps2[c], floor_thr)
                                                    : floor_thr)
                            : INFINITY; // dead query rows: accumulators stay at -inf, never a candidate
        float c_init = -t_init;
        if (MAXONLY) // sample pass: plain scores (C = +0); the debug export may plant a threshold instead
            c_init = (p.dbg_thr && live) ? -p.dbg_thr[qrow] : 0.0f;
        negthr[c] = f32x4{c_init, c_init, c_init, c_init};
    }

    SCand *const cwave = p.cand + ((size_t)blockIdx.x * QB + w * QW) * SCAP;

    auto compact_where = [&](int c, unsigned qmask) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        while (qmask) {
            const int q = __ffs((int)qmask) - 1;
            qmask &= qmask - 1;
            const int nq[4] = {__builtin_amdgcn_readlane(cnt[c], q), __builtin_amdgcn_readlane(cnt[c], q + 16),
                               __builtin_amdgcn_readlane(cnt[c], q + 32), __builtin_amdgcn_readlane(cnt[c], q + 48)};
            const float slack = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, eps2[c]), q));
            int n_new;
            float tn;
            bool have, ovf;
            screen_compact(cwave + (size_t)(16 * c + q) * SCAP, nq, k, slack, lane, n_new, tn, have, ovf);
            if (ovf && lane == 0)
                atomicOr(p.flag + ((qbase + 16 * c) >> 5), 1);
            if (n == q) {
                cnt[c] = (n_new - g + 3) >> 2; // ranks r < n_new with r & 3 == g
                if (have)
                    negthr[c] = f32x4{-tn, -tn, -tn, -tn};
            }
        }
    };

    // ---- DMA: tile = 16 wave-instructions of 1 KiB (2 docs x 512 B); wave w issues 2w, 2w+1 ----
    const char *D = (const char *)p.D16;
    const char *rowp[2];
    auto set_rows = [&](int tile) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 2 * (2 * w + i) + (lane >> 5);  // doc within the tile
            const int doc = min(tile * 32 + row, p.N - 1);
            const int chunk16 = (lane & 31) ^ (row & 15);   // source swizzle: logical = physical ^ (row & 15)
            rowp[i] = D + (size_t)doc * 512 + chunk16 * 16;
        }
    };
    auto dma_issue = [&](int tile, int stage) {
        set_rows(min(tile, t1 - 1));
        char *dst = ring + stage * STILE_BYTES + (2 * w) * 1024;
        __builtin_amdgcn_global_load_lds((gbl_void *)rowp[0], (lds_void *)dst, 16, 0, DMA_AUX);
        __builtin_amdgcn_global_load_lds((gbl_void *)rowp[1], (lds_void *)(dst + 1024), 16, 0, DMA_AUX);
    };
    // one of a tile's two pieces (the addresses were prepared by set_rows)
    auto dma_piece = [&](int i, int stage) {
        char *dst = ring + stage * STILE_BYTES + (2 * w + i) * 1024;
        __builtin_amdgcn_global_load_lds((gbl_void *)rowp[i], (lds_void *)dst, 16, 0, DMA_AUX);
    };

    for (;;) { // segments: the own range, then pool blocks
    if (t0 < t1) {
        // DMA runs SRING - STPB tiles ahead; one barrier per STPB tiles.
#pragma unroll
        for (int gi = 0; gi < SRING - STPB; ++gi)
            dma_issue(t0 + gi, gi);
        int stage = 0;
        const int rd_base = n * 512; // A row (document) n of sub-tile 0; sub-tile 1 is 16 rows = 8 KiB further
        // acc[u][c][r] = s16(doc tile*32 + 16u + 4g + r, query qbase + 16c + n) (main pass: minus the query's threshold)
        f32x4 acc[2][NSET];
        // Stagger: all eight waves run the same program between the same barriers, so left alone the two waves of a SIMD
        // reach their MFMA block, their LDS reads and their selection epilogue together.  Waves 4-7 (the second wave of
        // each SIMD) therefore DEFER a tile's selection until after the next tile iteration has started -- their
        // accumulators stay in registers across the barrier -- so that one half selects (VALU, stores) while the other
        // multiplies (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).  Same work, same results, shifted by one phase.
        const bool late = TT_SCREEN_STAGGER && w >= 4;
        bool pending = false;
        int ptile = 0;
        auto epilogue = [&](int tile) {
                const int tile_base = tile * 32;
            const bool partial = tile_base + 32 > p.N;
            if (!MAXONLY) {
                if (partial) { // rows past the corpus never pass: -inf has its sign bit set
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (tile_base + 16 * u + 4 * g + r >= p.N) {
#pragma unroll
                                for (int c = 0; c < NSET; ++c)
                                    acc[u][c][r] = -INFINITY;
                            }
                }
                // any score at or above its threshold <=> some accumulator has a clear sign bit
                // <=> the signed-integer maximum of the raw registers is >= 0
                int mall = INT_MIN;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int c = 0; c < NSET; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            mall = max(mall, __float_as_int(acc[u][c][r]));
                if (__ballot(mall >= 0) != 0ull) {
#pragma unroll
                    for (int c = 0; c < NSET; ++c) {
                        int mu[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            mu[u] = max(max(__float_as_int(acc[u][c][0]), __float_as_int(acc[u][c][1])),
                                        max(__float_as_int(acc[u][c][2]), __float_as_int(acc[u][c][3])));
                        if (__ballot(max(mu[0], mu[1]) >= 0) == 0ull)
                            continue;
                        // append pass: every lane appends to its own quarter of the query's buffer with its
                        // own counter (no ballots), through inline-asm stores (a compiler-visible VMEM op here
                        // would put s_waitcnt vmcnt(0) on the hot path and drain the DMA ring)
                        const unsigned mine = (unsigned)(((16 * c + n) * SCAP + SQUART * g) * sizeof(SCand));
                        const float thr_c = -negthr[c][0];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            if (__ballot(mu[u] >= 0) == 0ull)
                                continue;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if (__float_as_int(acc[u][c][r]) >= 0) {
                                    scand_store_async(cwave, mine + (unsigned)cnt[c] * (unsigned)sizeof(SCand),
                                                      acc[u][c][r] + thr_c, tile_base + 16 * u + 4 * g + r);
                                    ++cnt[c];
                                }
                            }
                        }
                        unsigned long long full = __ballot(cnt[c] > SQ_TRIGGER);
                        full = (full | (full >> 32));
                        full = (full | (full >> 16)) & 0xffffull;
                        if (full)
                            compact_where(c, (unsigned)full);
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < NSET; ++c) { // one maximum per (tile, query): the k-th largest of them seeds the thresholds
                    float m = -INFINITY;
                    if (!partial) {
                        // signed-integer max of the raw bits = the float max when any value is >= 0, else the
                        // smallest one: still the score of a real document of this tile, which is all the
                        // threshold argument needs (v_max3_i32: no NaN canonicalisation, 4 instructions)
                        int mi = INT_MIN;
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                mi = max(mi, __float_as_int(acc[u][c][r]));
                        m = __int_as_float(mi);
                    } else {
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                m = fmaxf(m, tile_base + 16 * u + 4 * g + r < p.N ? acc[u][c][r] : -INFINITY);
                    }
                    m = fmaxf(m, __shfl_xor(m, 16));
                    m = fmaxf(m, __shfl_xor(m, 32));
                    const int qrow = qbase + 16 * c + n;
                    if (g == 0 && qrow < p.B)
                        f32_store_async(p.max_val + (size_t)qrow * p.n_tiles + tile, m);
                }
            }
                };
        for (int tile = t0; tile < t1; ++tile) {
            if ((tile - t0) % STPB == 0) {
                // own DMAs of this interval's tiles have landed; the barrier extends that to every wave's
                // and guarantees every wave is done reading the tiles of the previous interval
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (SRING - 2 * STPB)) : "memory");
                __builtin_amdgcn_s_barrier();
            }
            // Every tile iteration refills the ring slot SRING - STPB tiles ahead (free since the last barrier);
            // the two LDS-DMA pieces go out between k-steps of the MFMA loop rather than in one burst behind
            // the barrier, where all eight waves would queue them at the same moment.
            const int fill_stage = (stage + SRING - STPB) % SRING;
            set_rows(min(tile + SRING - STPB, t1 - 1));
            if (!wave_live) {
                dma_piece(0, fill_stage);
                dma_piece(1, fill_stage);
            }
            if (wave_live) {
                if (late && pending)
                    epilogue(ptile);
                // acc[u][c][r] = s16(doc tile*32 + 16u + 4g + r, query qbase + 16c + n)
                // (main pass: minus the query's threshold, see negthr)
                const char *buf = ring + stage * STILE_BYTES + rd_base;
                // A fragments run two k-steps ahead of the MFMAs that consume them (three register sets);
                // the scheduling fences keep hipcc from sinking the reads back next to their use
                h8 a0[3], a1[3];
                auto a_read = [&](int s) {
                    const int off = ((4 * s + g) ^ n) << 4;
                    a0[s % 3] = *(const h8 *)(buf + off);
                    a1[s % 3] = *(const h8 *)(buf + 16 * 512 + off);
                };
                a_read(0);
                a_read(1);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    if (s + 2 < 8)
                        a_read(s + 2);
                    if (s == 2)
                        dma_piece(0, fill_stage);
                    if (s == 6)
                        dma_piece(1, fill_stage);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < NSET; ++c) {
                        const f32x4 c0 = s == 0 ? negthr[c] : acc[0][c];
                        const f32x4 c1 = s == 0 ? negthr[c] : acc[1][c];
                        acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[s % 3], qreg[c][s], c0, 0, 0, 0);
                        acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[s % 3], qreg[c][s], c1, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (late) { // this tile's selection runs after the next barrier, beside the other half's MFMAs
                    pending = true;
                    ptile = tile;
                } else {
                    epilogue(tile);
                }
            }
            stage = (stage + 1) % SRING;
        }
        if (wave_live && late && pending)
            epilogue(ptile);
    }
    if (MAXONLY || p.tail_blocks == 0)
        break;
    // The chip's eight XCDs do not run this loop at one speed (per-workgroup clocks: 3.61 .. 3.97 ms for identical
    // 10M-document shares, the medians of the XCDs 3.62 .. 3.96): with equal static shares the launch ends with its
    // slowest workgroup while the fastest idle for 9 % of it.  The last part of the corpus is therefore handed out in
    // blocks: whoever is done draws the next one (a fresh pipeline per block: ring primed again, ~3 us per ~30-50 us).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the clamped over-prefetch of the segment's end
    __builtin_amdgcn_s_barrier();                    // ... has landed for every wave; nobody reads the ring any more
    if (threadIdx.x == 0)
        next_block = atomicAdd(p.tail_ctr + qgroup, 1);
    __syncthreads();
    const int blk = next_block;
    if (blk >= p.tail_blocks)
        break;
    t0 = p.static_tiles + blk * p.tail_g;
    t1 = min(t0 + p.tail_g, p.n_tiles);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier(); // no wave may leave while a sibling's LDS-DMA could still be consumed

    if (MAXONLY)
        return;
    // final compaction (bounds the pool the finish kernel sees) and counts out: quarter g's count in byte g
#pragma unroll
    for (int c = 0; c < NSET; ++c) {
        int tot = cnt[c] + __shfl_xor(cnt[c], 16);
        tot += __shfl_xor(tot, 32);
        const unsigned long long over = __ballot(tot > k) & 0xffffull;
        if (over)
            compact_where(c, (unsigned)over);
        int packed = cnt[c] << (8 * g);
        packed |= __shfl_xor(packed, 16);
        packed |= __shfl_xor(packed, 32);
        const int qrow = qbase + 16 * c + n;
        if (g == 0 && qrow < p.B)
            p.pcnt[(size_t)qrow * p.n_chunks + chunk] = packed;
    }
}

// ------------------------------------------------------------------ small batches: streaming form
// For B <= 64 the shared-tile kernel above would leave most of a workgroup's eight waves without queries.  Here every
// WAVE is an independent streaming engine (the organisation of the exact kernel, score_topk.hip): it keeps one
// 32-query tile as B operands (64 VGPRs), walks its own range of documents through a private 4-slab LDS ring
// (slab = 32 documents x 64 features f16 = 4 KiB, 4 global_load_lds per slab, 3 slabs in flight, no barriers)
// and selects exactly like screen_kernel.  MFMA work is 1/16 of the exact kernel's, so the launch is bound by
// HBM streaming of the fp16 shadow corpus (N x 512 B).
constexpr int TW = 4;                    // waves per workgroup (2 workgroups per CU)
constexpr int TSLAB_BYTES = 32 * 128;    // 32 docs x 64 f16
constexpr int TSTAGE = 4;                // ring depth in slabs
constexpr int TDMA = 4;                  // DMA instructions per slab
// Cache policy of the document stream: nt (aux = 2).  Every byte is read once by one wave; with the default policy the
// same kernel reached 6.1-6.2 TB/s, with nt 6.8-6.9 (0.910 -> 0.816-0.824 ms per B = 32 search over 10M documents, A/B on
// one box; MI355X_MICROARCH.md 'nt-weights').  NOT for the exact kernel at large batches, whose 32 query tiles re-read a
// chunk from L2 (44.2 -> 52.9 ms at B = 1024 with nt), nor for the shared-tile screen (two workgroups per chunk).
constexpr int TSTREAM_AUX = 2;

// NQS = 16-query sets per wave: 2 (B <= 32), or 4 (33 <= B <= 64: ONE pass of the stream for 64 queries instead of the
// shared-tile form's 1.03 ms; twice the MFMAs per tile, still a fifth of what the stream allows)
template <bool MAXONLY, int NQS = 2>
__global__ __launch_bounds__(TW * 64, 2) void screen_stream_kernel(ScreenParams p)
{
    constexpr int QPT = 16 * NQS; // queries per task
    extern __shared__ __attribute__((aligned(16))) char ring_all[]; // [TW][TSTAGE][TSLAB_BYTES]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *const ring = ring_all + w * (TSTAGE * TSLAB_BYTES);
    const int task = blockIdx.x * TW + w;                 // = qtile * n_chunks + chunk
    const int n_qtiles = (p.B + QPT - 1) / QPT;
    if (task >= n_qtiles * p.n_chunks)
        return; // wave-uniform; this kernel has no workgroup barrier
    const int qtile = task / p.n_chunks, chunk = task % p.n_chunks;
    const int t0 = chunk * p.tiles_per_chunk;
    const int t1 = min(t0 + p.tiles_per_chunk, p.n_tiles);
    const int k = p.k;
    const int g = lane >> 4, n = lane & 15;
    const int qbase = qtile * QPT;

    // ---- query operands: set c holds queries qbase + 16c + n; lane (n,g) keeps features 32s + 8g .. +7 ----
    h8 qreg[NQS][8];
    float eps2[NQS];
    f32x4 negthr[NQS];
    int cnt[NQS];
#pragma unroll
    for (int c = 0; c < NQS; ++c) {
        cnt[c] = 0;
        const int qrow = qbase + 16 * c + n;
        const bool live = qrow < p.B;
        const h8 *src = p.qimg + ((size_t)(qbase / 16 + c) * 8) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 8; ++s)
            qreg[c][s] = src[s * 64];
        const float qn = p.qnorm[qrow];
        eps2[c] = 2.0f * screen_eps(qn, p.dmax);
        const float floor_thr = -(1.01f * qn * p.dmax + 1e-30f);
        const float t_init = live ? ((!MAXONLY && p.thr0) ? fmaxf(p.thr0[(size_t)qrow * p.thr0_stride + p.thr0_stride - 1] - eps2[c], floor_thr)
                                                    : floor_thr)
                            : INFINITY;
        float c_init = -t_init;
        if (MAXONLY)
            c_init = (p.dbg_thr && live) ? -p.dbg_thr[qrow] : 0.0f;
        negthr[c] = f32x4{c_init, c_init, c_init, c_init};
    }

    SCand *const cwave = p.cand + (size_t)task * QPT * SCAP;

    auto compact_where = [&](int c, unsigned qmask) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        while (qmask) {
            const int q = __ffs((int)qmask) - 1;
            qmask &= qmask - 1;
            const int nq[4] = {__builtin_amdgcn_readlane(cnt[c], q), __builtin_amdgcn_readlane(cnt[c], q + 16),
                               __builtin_amdgcn_readlane(cnt[c], q + 32), __builtin_amdgcn_readlane(cnt[c], q + 48)};
            const float slack = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, eps2[c]), q));
            int n_new;
            float tn;
            bool have, ovf;
            screen_compact(cwave + (size_t)(16 * c + q) * SCAP, nq, k, slack, lane, n_new, tn, have, ovf);
            if (ovf && lane == 0)
                atomicOr(p.flag + ((qbase + 16 * c) >> 5), 1);
            if (n == q) {
                cnt[c] = (n_new - g + 3) >> 2;
                if (have)
                    negthr[c] = f32x4{-tn, -tn, -tn, -tn};
            }
        }
    };

    // ---- DMA: slab (tile, kq) = features 64kq..64kq+63 of the tile's 32 documents.  Instruction jj moves rows
    //      8jj..8jj+7: lane -> (row 8jj + lane/8, physical 16-B chunk lane%8); logical chunk = physical ^ (row & 7).
    const char *D = (const char *)p.D16;
    int dma_tile = t0, dma_kq = 0;
    const char *rowp[TDMA];
    auto set_rows = [&](int tile) {
#pragma unroll
        for (int jj = 0; jj < TDMA; ++jj) {
            const int row = 8 * jj + (lane >> 3);
            const int doc = min(tile * 32 + row, p.N - 1);
            rowp[jj] = D + (size_t)doc * 512 + (((lane & 7) ^ (row & 7)) << 4);
        }
    };
    auto dma_issue = [&](int stage) {
        char *dst = ring + stage * TSLAB_BYTES;
#pragma unroll
        for (int jj = 0; jj < TDMA; ++jj)
            __builtin_amdgcn_global_load_lds((gbl_void *)(rowp[jj] + dma_kq * 128), (lds_void *)(dst + jj * 1024), 16, 0, TSTREAM_AUX);
        if (++dma_kq == 4) {
            dma_kq = 0;
            dma_tile = min(dma_tile + 1, t1 - 1); // past the end: harmless re-read
            set_rows(dma_tile);
        }
    };

    if (t0 < t1) {
        set_rows(t0);
#pragma unroll
        for (int i = 0; i < TSTAGE - 1; ++i)
            dma_issue(i);
        int stage = 0;
        // A row (document) n of sub-tile 0 / 16 + n of sub-tile 1; both have row & 7 == n & 7
        const char *rd_row = ring + n * 128;
        const int rsw = n & 7;
        for (int tile = t0; tile < t1; ++tile) {
            f32x4 acc[2][NQS]; // acc[u][c][r] = s16(doc tile*32 + 16u + 4g + r, query qbase + 16c + n) - thr
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                // slab (tile,kq) has landed once at most TSTAGE-2 younger slabs are pending (candidate stores
                // also count in vmcnt: they only make this wait stricter)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TDMA * (TSTAGE - 2)) : "memory");
                const char *buf = rd_row + stage * TSLAB_BYTES;
                h8 a[2][2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        a[u][s2] = *(const h8 *)(buf + u * 2048 + (((4 * s2 + g) ^ rsw) << 4));
                // the ring slot consumed one step ago is free once its reads have returned
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dma_issue((stage + TSTAGE - 1) % TSTAGE);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int c = 0; c < NQS; ++c)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const f32x4 cin = (kq == 0 && s2 == 0) ? negthr[c] : acc[u][c];
                            acc[u][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u][s2], qreg[c][2 * kq + s2], cin, 0, 0, 0);
                        }
                stage = (stage + 1) % TSTAGE;
            }
            const int tile_base = tile * 32;
            const bool partial = tile_base + 32 > p.N;
            if (!MAXONLY) {
                if (partial) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (tile_base + 16 * u + 4 * g + r >= p.N) {
#pragma unroll
                                for (int c = 0; c < NQS; ++c)
                                    acc[u][c][r] = -INFINITY;
                            }
                }
                int mall = INT_MIN;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int c = 0; c < NQS; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            mall = max(mall, __float_as_int(acc[u][c][r]));
                if (__ballot(mall >= 0) != 0ull) {
#pragma unroll
                    for (int c = 0; c < NQS; ++c) {
                        const unsigned mine = (unsigned)(((16 * c + n) * SCAP + SQUART * g) * sizeof(SCand));
                        const float thr_c = -negthr[c][0];
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if (__float_as_int(acc[u][c][r]) >= 0) {
                                    scand_store_async(cwave, mine + (unsigned)cnt[c] * (unsigned)sizeof(SCand),
                                                      acc[u][c][r] + thr_c, tile_base + 16 * u + 4 * g + r);
                                    ++cnt[c];
                                }
                            }
                        unsigned long long full = __ballot(cnt[c] > SQ_TRIGGER);
                        full = (full | (full >> 32));
                        full = (full | (full >> 16)) & 0xffffull;
                        if (full)
                            compact_where(c, (unsigned)full);
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < NQS; ++c) {
                    float m = -INFINITY;
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            m = fmaxf(m, (!partial || tile_base + 16 * u + 4 * g + r < p.N) ? acc[u][c][r] : -INFINITY);
                    m = fmaxf(m, __shfl_xor(m, 16));
                    m = fmaxf(m, __shfl_xor(m, 32));
                    const int qrow = qbase + 16 * c + n;
                    if (g == 0 && qrow < p.B)
                        f32_store_async(p.max_val + (size_t)qrow * p.n_tiles + tile, m);
                }
            }
        }
    }
    // LDS-DMA still in flight would land after the wave has ended: drain it (and the stores)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MAXONLY)
        return;
#pragma unroll
    for (int c = 0; c < NQS; ++c) {
        int tot = cnt[c] + __shfl_xor(cnt[c], 16);
        tot += __shfl_xor(tot, 32);
        const unsigned long long over = __ballot(tot > k) & 0xffffull;
        if (over)
            compact_where(c, (unsigned)over);
        int packed = cnt[c] << (8 * g);
        packed |= __shfl_xor(packed, 16);
        packed |= __shfl_xor(packed, 32);
        const int qrow = qbase + 16 * c + n;
        if (g == 0 && qrow < p.B)
            p.pcnt[(size_t)qrow * p.n_chunks + chunk] = packed;
    }
}

// ------------------------------------------------------------------ finish
struct FinishParams {
    const float *Q;
    const float *D32;
    int B, N, k, n_chunks;
    float dmax;
    const SCand *cand;
    const int *pcnt;
    int *flag;
    int64_t idx_offset;
    float *out_val;
    int64_t *out_idx;
    int q_per_block; // candidate layout [query group][chunk][q_per_block][SCAP]: 512 (shared tiles) or 32 (streaming)
    int *stats;      // [B][2]: pooled candidates, survivors (>= A_k - 2 eps) of each query -- what the filter let through
};

__device__ __forceinline__ bool before_f(float sa, int ia, float sb, int ib) { return sa > sb || (sa == sb && ia < ib); }

// float <-> unsigned with the same ordering (for the radix select)
__device__ __forceinline__ unsigned f32_order_key(float f)
{
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_from_order_key(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(256) void screen_finish_kernel(FinishParams p)
{
    __shared__ float pool_v[POOL_MAX];
    __shared__ float qs[256];
    __shared__ float sv_v[SURV_MAX];
    __shared__ int sv_x[SURV_MAX];
    __shared__ float red_v[4];
    __shared__ int hist[256];
    // per-chunk tables sized at launch (2 n_chunks + 1 ints): the shared-tile form has <= 255 chunks, which keeps
    // the workgroup under 40 KB of LDS and four of them on a CU; the streaming form needs up to FIN_MAX_CHUNKS
    extern __shared__ int fin_dyn[];
    int *const ccnt = fin_dyn;                  // packed quarter counts per chunk
    int *const pre = fin_dyn + p.n_chunks;      // exclusive prefix of the per-chunk totals (n_chunks + 1)
    __shared__ int sel[2];
    __shared__ int n_pool, n_surv;
    const int row = blockIdx.x, tid = threadIdx.x;
    if (tid == 0)
        n_pool = n_surv = 0;
    qs[tid] = p.Q[(size_t)row * 256 + tid];
    __syncthreads();
    // |q| (fixed-order enough: any fp32 rounding is covered by the safety factor in screen_eps)
    float ss = qs[tid] * qs[tid];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        ss += __shfl_xor(ss, off);
    if ((tid & 63) == 0)
        red_v[tid >> 6] = ss;
    __syncthreads();
    const float eps2 = 2.0f * screen_eps(sqrtf(red_v[0] + red_v[1] + red_v[2] + red_v[3]), p.dmax);
    __syncthreads();

    // ---- pool every workgroup's candidates for this query (counts -> prefix -> parallel copy) ----
    const int qgroup = row / p.q_per_block, qin = row % p.q_per_block;
    bool too_many = false;
    // per-chunk totals: thread t owns chunks [t*per, t*per+per); its sum -> hist[t]; thread 0 scans the 256 sums
    const int per = (p.n_chunks + 255) / 256;
    {
        int sum = 0;
        for (int c = tid * per; c < min(tid * per + per, p.n_chunks); ++c) {
            const int v = p.pcnt[(size_t)row * p.n_chunks + c]; // quarter counts, one per byte
            ccnt[c] = v;
            sum += (v & 0xff) + ((v >> 8) & 0xff) + ((v >> 16) & 0xff) + ((v >> 24) & 0xff);
        }
        // exclusive scan of the 256 per-thread sums: shuffles inside each wave, then the 4 wave totals
        int inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(inc, off);
            if ((tid & 63) >= off)
                inc += v;
        }
        if ((tid & 63) == 63)
            hist[tid >> 6] = inc; // wave totals in hist[0..3] (hist is free until the select)
        __syncthreads();
        int base = 0;
        for (int wv = 0; wv < (tid >> 6); ++wv)
            base += hist[wv];
        if (tid == 255)
            n_pool = base + inc;
        __syncthreads();
        hist[tid] = base + inc - sum;
    }
    __syncthreads();
    {
        int run = hist[tid];
        for (int c = tid * per; c < min(tid * per + per, p.n_chunks); ++c) {
            pre[c] = run;
            const int v = ccnt[c];
            run += (v & 0xff) + ((v >> 8) & 0xff) + ((v >> 16) & 0xff) + ((v >> 24) & 0xff);
        }
        if (tid == 255)
            pre[p.n_chunks] = n_pool;
    }
    __syncthreads();
    const int total = n_pool;
    if (total > POOL_MAX)
        too_many = true;
    // pooled position m -> its entry in the candidate buffers (chunk by binary search, then quarter and slot).
    // Only the scores are pooled in LDS (52 KB per workgroup -> three workgroups per CU); the few survivors
    // fetch their document index through the same mapping.
    auto locate = [&](int m) -> const SCand * {
        int lo = 0, hi = p.n_chunks; // largest c with pre[c] <= m
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= m)
                lo = mid;
            else
                hi = mid;
        }
        int o = m - pre[lo], slot = 0; // quarter g of the buffer starts at entry 32 g
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int ng = (ccnt[lo] >> (8 * gq)) & 0xff;
            if (o >= 0 && o < ng)
                slot = SQUART * gq + o;
            o -= ng;
            if (o < 0)
                o = INT_MIN / 2;
        }
        return p.cand + ((size_t)(qgroup * p.n_chunks + lo) * p.q_per_block + qin) * SCAP + slot;
    };
    for (int m = tid; m < total && m < POOL_MAX; m += 256)
        pool_v[m] = locate(m)->v;
    __syncthreads();
    const int np = min(n_pool, POOL_MAX);
    // ---- A_k = k-th best approximate score over the whole corpus: 4-pass radix select on the
    //      order-preserving integer image of the scores (cost independent of k) ----
    float kth = -INFINITY;
    const int found = np >= p.k ? p.k : np;
    if (np >= p.k) {
        unsigned prefix = 0u, mask = 0u;
        int k_rem = p.k;
#pragma unroll 1
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            hist[tid] = 0;
            __syncthreads();
            for (int m = tid; m < np; m += 256) {
                const unsigned key = f32_order_key(pool_v[m]);
                if ((key & mask) == prefix)
                    atomicAdd(&hist[(key >> shift) & 255u], 1);
            }
            __syncthreads();
            if (tid < 64) { // one wave: suffix sums over the 256 bins (4 per lane, high bins first)
                const int ln = tid;
                int c[4], sm = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    c[i] = hist[255 - (ln * 4 + i)];
                    sm += c[i];
                }
                int incl = sm;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int y = __shfl_up(incl, off);
                    if (ln >= off)
                        incl += y;
                }
                int above = incl - sm;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (above < k_rem && above + c[i] >= k_rem) {
                        sel[0] = 255 - (ln * 4 + i);
                        sel[1] = above;
                    }
                    above += c[i];
                }
            }
            __syncthreads();
            prefix |= (unsigned)sel[0] << shift;
            mask |= 0xffu << shift;
            k_rem -= sel[1];
            __syncthreads();
        }
        kth = f32_from_order_key(prefix);
    }
    const float cut = found == p.k ? kth - eps2 : -INFINITY;
    // ---- survivors, exact fp32 FMA-chain rescoring (the oracle's order: features ascending) ----
    for (int m = tid; m < np; m += 256) {
        if (pool_v[m] >= cut) {
            const int slot = atomicAdd(&n_surv, 1);
            if (slot < SURV_MAX)
                sv_x[slot] = locate(m)->x;
            else
                too_many = true;
        }
    }
    __syncthreads();
    const int ns = min(n_surv, SURV_MAX);
    for (int sidx = tid; sidx < ns; sidx += 256) {
        // one thread per survivor (a second, third, fourth round only beyond 256 of them): the chain is sequential by
        // definition, but the row's loads are not -- 16 of them (256 B) are issued back to back before the 64 fmaf that
        // consume them, four batches per row (left to hipcc the loop waited for one 16-byte load per iteration: ~30 us of a
        // 46 us kernel at k = 50)
        const float *drow = p.D32 + (size_t)sv_x[sidx] * 256;
        float acc = 0.0f;
#pragma unroll 1
        for (int x0 = 0; x0 < 256; x0 += 64) {
            f32x4 dv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                dv[i] = __builtin_nontemporal_load((const f32x4 *)(drow + x0 + 4 * i));
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc = fmaf(qs[x0 + 4 * i], dv[i].x, acc);
                acc = fmaf(qs[x0 + 4 * i + 1], dv[i].y, acc);
                acc = fmaf(qs[x0 + 4 * i + 2], dv[i].z, acc);
                acc = fmaf(qs[x0 + 4 * i + 3], dv[i].w, acc);
            }
        }
        sv_v[sidx] = acc;
    }
    if (tid == 0) {
        p.stats[2 * row] = n_pool;
        p.stats[2 * row + 1] = n_surv;
    }
    if (__syncthreads_or(too_many) && tid == 0)
        atomicOr(p.flag + (row >> 5), 4);
    // ---- exact top-k of the survivors: every thread ranks its own survivor against all others ----
    __syncthreads();
    for (int sidx = tid; sidx < ns; sidx += 256) {
        const float mv = sv_v[sidx];
        const int mx = sv_x[sidx];
        int rank = 0;
        for (int u = 0; u < ns; ++u)
            rank += before_f(sv_v[u], sv_x[u], mv, mx) ? 1 : 0;
        if (rank < p.k) {
            p.out_val[(size_t)row * p.k + rank] = mv;
            p.out_idx[(size_t)row * p.k + rank] = p.idx_offset + mx;
        }
    }
    for (int r = ns + tid; r < p.k; r += 256) { // fewer survivors than k (tiny corpora)
        p.out_val[(size_t)row * p.k + r] = -INFINITY;
        p.out_idx[(size_t)row * p.k + r] = -1;
    }
}

// ------------------------------------------------------------------ fp16 shadow copy + corpus stats
__global__ __launch_bounds__(256) void build_f16_kernel(const float *__restrict__ D, int64_t N, int d,
                                                        _Float16 *__restrict__ out, unsigned *__restrict__ stats)
{
    __shared__ float red[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float nmax = 0.0f, amax = 0.0f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < N; row += (int64_t)gridDim.x * 4) {
        float ss = 0.0f;
        for (int x = lane * 4; x < d; x += 256) {
            const f32x4 v = *(const f32x4 *)(D + row * d + x);
            h4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ss += v[e] * v[e];
                amax = fmaxf(amax, fabsf(v[e]));
                if (!(fabsf(v[e]) <= 3.0e38f))
                    amax = INFINITY; // NaN / inf anywhere disables the screen
                hv[e] = (_Float16)v[e];
            }
            *(h4 *)(out + row * d + x) = hv;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
            ss += __shfl_xor(ss, off);
        nmax = fmaxf(nmax, sqrtf(ss));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        amax = fmaxf(amax, __shfl_xor(amax, off));
    if (lane == 0) {
        red[wave] = nmax;
        red[4 + wave] = amax;
    }
    __syncthreads();
    if (threadIdx.x == 0) { // non-negative floats order like their bit patterns; inf is the largest
        atomicMax(stats, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
        atomicMax(stats + 1, __float_as_uint(fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]))));
    }
}

// bf16 rows (streamed corpus blocks) -> fp32 rows (exact: bf16 is a truncated fp32) + fp16 shadow + stats
__global__ __launch_bounds__(256) void build_from_bf16_kernel(const unsigned short *__restrict__ S, int64_t N, int d,
                                                              float *__restrict__ out32, _Float16 *__restrict__ out16,
                                                              unsigned *__restrict__ stats)
{
    __shared__ float red[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float nmax = 0.0f, amax = 0.0f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < N; row += (int64_t)gridDim.x * 4) {
        float ss = 0.0f;
        for (int x = lane * 4; x < d; x += 256) {
            const uint2 raw = *(const uint2 *)(S + row * d + x);
            f32x4 v;
            v[0] = __uint_as_float(raw.x << 16);
            v[1] = __uint_as_float(raw.x & 0xffff0000u);
            v[2] = __uint_as_float(raw.y << 16);
            v[3] = __uint_as_float(raw.y & 0xffff0000u);
            h4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ss += v[e] * v[e];
                amax = fmaxf(amax, fabsf(v[e]));
                if (!(fabsf(v[e]) <= 3.0e38f))
                    amax = INFINITY;
                hv[e] = (_Float16)v[e];
            }
            *(f32x4 *)(out32 + row * d + x) = v;
            if (out16)
                *(h4 *)(out16 + row * d + x) = hv;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
            ss += __shfl_xor(ss, off);
        nmax = fmaxf(nmax, sqrtf(ss));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        amax = fmaxf(amax, __shfl_xor(amax, off));
    if (lane == 0) {
        red[wave] = nmax;
        red[4 + wave] = amax;
    }
    __syncthreads();
    if (threadIdx.x == 0 && stats) {
        atomicMax(stats, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
        atomicMax(stats + 1, __float_as_uint(fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]))));
    }
}

struct SPlan {
    bool stream;      // B <= STREAM_MAX_B: one independent streaming wave per (32-query tile, document chunk)
    int nset;         // shared-tile form: 16-query sets per wave (4, 2 or 1)
    int n_qgroups;    // query groups of q_per_block rows (workgroup rows of 128 nset queries, or 32-query tiles when streaming)
    int q_per_block;
    int n_tiles, n_chunks, tiles_per_chunk, n_blocks;
    int static_tiles, tail_g, tail_blocks; // shared-tile main pass: see ScreenParams
    size_t tailctr_off;
    // sample pass
    bool sample;
    int s_tiles, s_chunks, s_tiles_per_chunk, s_blocks;
    int64_t s_docs;
    size_t cand_off, pcnt_off, smax_val_off, sthr_val_off, qimg_off, qnorm_off, stats_off, ws_bytes, lds;
    int rows_pad; // n_qgroups * q_per_block
};

constexpr int64_t SAMPLE_MIN_N = 65536;
constexpr int STREAM_MAX_B = 64; // one streaming pass: 32 queries per wave (2 sets) up to B = 32, 64 (4 sets) up to B = 64;
                                 // beyond that the shared-tile kernel wins (B = 65 .. 128: 1.20 ms against two passes)

int screen_cus()
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        cus = 256;
    }
    return cus;
}

SPlan make_splan(int B, int64_t N, int k)
{
    SPlan pl;
    pl.stream = B <= STREAM_MAX_B;
    if (B <= SW * 32) {
        pl.nset = B <= SW * 16 ? 1 : 2;
    } else { // as few groups as 512-query groups would need, as evenly filled as whole 16-query sets per wave allow
        const int groups = (B + SW * 64 - 1) / (SW * 64);
        pl.nset = (B + groups - 1) / groups <= SW * 48 ? 3 : 4;
    }
    pl.q_per_block = pl.stream ? (B <= 32 ? 32 : 64) : SW * 16 * pl.nset;
    pl.n_qgroups = (B + pl.q_per_block - 1) / pl.q_per_block;
    pl.n_tiles = (int)((N + 31) / 32);
    int want, max_chunks;
    if (pl.stream) {
        // 8 resident waves per CU (2 workgroups of 4), one round
        want = (screen_cus() * 2 * TW + pl.n_qgroups - 1) / pl.n_qgroups;
        // with seeded thresholds a chunk rarely keeps anything; without them every chunk ends with >= k entries
        // and the finish kernel's pool bounds the number of chunks
        max_chunks = N >= SAMPLE_MIN_N ? FIN_MAX_CHUNKS : POOL_MAX / (k + 16);
    } else {
        // one workgroup per CU is resident (8 waves x 256 VGPRs): aim at exactly one round, the per-workgroup
        // set-up (query load + conversion, final compaction) is ~0.1 ms and would be paid once per round
        want = (screen_cus() + pl.n_qgroups - 1) / pl.n_qgroups;
        // room in the finish kernel's pool for k + slack entries per chunk
        max_chunks = POOL_MAX / (k + 16);
        max_chunks = max_chunks > 255 ? 255 : max_chunks;
    }
    want = want > max_chunks ? max_chunks : want;
    want = want > pl.n_tiles ? pl.n_tiles : want;
    want = want < 1 ? 1 : want;
    pl.tiles_per_chunk = (pl.n_tiles + want - 1) / want;
    pl.n_chunks = (pl.n_tiles + pl.tiles_per_chunk - 1) / pl.tiles_per_chunk;
    pl.static_tiles = pl.n_tiles;
    pl.tail_g = 1;
    pl.tail_blocks = 0;
    {
        // pool = the last 1/TT_SCREEN_TAIL_DIV of every chunk's share (0: everything static), in blocks of a quarter of
        // that share, 8..32 tiles; only worth it when a share is long enough for several blocks
        const int tail_div = TT_AB_SWITCH(TT_SCREEN_TAIL_DIV, 8);
        const int share = tail_div > 0 ? pl.tiles_per_chunk / tail_div : 0;
        if (!pl.stream && share >= 16) {
            const int own = pl.tiles_per_chunk - share;
            int g = share / 4;
            g = g < 8 ? 8 : (g > 32 ? 32 : g);
            pl.tiles_per_chunk = own;
            pl.static_tiles = own * pl.n_chunks < pl.n_tiles ? own * pl.n_chunks : pl.n_tiles;
            pl.tail_g = g;
            pl.tail_blocks = (pl.n_tiles - pl.static_tiles + g - 1) / g;
        }
    }
    const int n_tasks = pl.n_qgroups * pl.n_chunks;
    pl.n_blocks = pl.stream ? (n_tasks + TW - 1) / TW : n_tasks;
    const size_t rows = (size_t)pl.n_qgroups * pl.q_per_block;
    size_t off = 0;
    pl.cand_off = off;
    off = tt_align_up(off + (size_t)n_tasks * pl.q_per_block * SCAP * sizeof(SCand), 256);
    pl.pcnt_off = off;
    off = tt_align_up(off + rows * pl.n_chunks * sizeof(int), 256);
    // sample pass: one maximum per 32-document tile of the sample; the k-th largest seeds the thresholds
    pl.sample = N >= SAMPLE_MIN_N;
    // Sample size: 1/64 of the corpus, or enough documents that the k-th sample maximum lets through about
    // one candidate per 32x32 score tile or fewer (k / s_docs per score): matters for small shards, large k.
    int64_t s_docs = N / 64;
    // measured on a 1.25M-document shard (B = 1024, A/B on one box): k = 10: 2048 -> 0.697, 4096 -> 0.664, 8192 -> 0.677 ms;
    // k = 50: 2048 -> 0.735, 4096 -> 0.767, 8192 -> 0.827 ms (the sample pass itself grows with k * per_k)
    const int per_k = k <= 16 ? 4096 : 2048;
    const int64_t s_min = (int64_t)k * per_k < N / 4 ? (int64_t)k * per_k : N / 4;
    s_docs = s_docs < s_min ? s_min : s_docs;
    s_docs = s_docs < 32 ? 32 : s_docs;
    pl.s_docs = (s_docs + 31) / 32 * 32;
    pl.s_tiles = (int)(pl.s_docs / 32);
    int s_want = pl.stream ? (screen_cus() * 2 * TW + pl.n_qgroups - 1) / pl.n_qgroups
                           : (screen_cus() + pl.n_qgroups - 1) / pl.n_qgroups; // one round, one maximum per TILE
    s_want = s_want > pl.s_tiles ? pl.s_tiles : s_want;
    s_want = s_want < 1 ? 1 : s_want;
    pl.s_tiles_per_chunk = (pl.s_tiles + s_want - 1) / s_want;
    pl.s_chunks = pl.sample ? (pl.s_tiles + pl.s_tiles_per_chunk - 1) / pl.s_tiles_per_chunk : 0;
    pl.s_blocks = pl.stream ? (pl.n_qgroups * pl.s_chunks + TW - 1) / TW : pl.n_qgroups * pl.s_chunks;
    pl.smax_val_off = off;
    off = tt_align_up(off + rows * (size_t)(pl.sample ? pl.s_tiles : 1) * sizeof(float), 256);
    pl.sthr_val_off = off;
    off = tt_align_up(off + rows * sizeof(float), 256);
    pl.rows_pad = (int)rows;
    pl.qimg_off = off;
    off = tt_align_up(off + rows * 256 * sizeof(_Float16), 256);
    pl.qnorm_off = off;
    off = tt_align_up(off + rows * sizeof(float), 256);
    pl.tailctr_off = off;
    off = tt_align_up(off + (size_t)pl.n_qgroups * sizeof(int), 256);
    pl.stats_off = off;
    off = tt_align_up(off + rows * 2 * sizeof(int), 256);
    pl.ws_bytes = off;
    pl.lds = pl.stream ? (size_t)TW * TSTAGE * TSLAB_BYTES : (size_t)SRING * STILE_BYTES;
    return pl;
}

} // namespace

TT_EXPORT int tt_index_build_f16(const float *D, int64_t N, int d, void *D16, float *stats, tt_stream_t stream)
{
    if (N < 0 || d <= 0 || (d & 3))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_index_build_f16: N=%lld d=%d", (long long)N, d);
    if (!stats || (N > 0 && (!D || !D16)))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_index_build_f16: null pointer");
    hipStream_t st = (hipStream_t)stream;
    TT_RC_CHECK(tt_zero_async(stats, 2 * sizeof(float), st));
    if (N == 0)
        return TT_OK;
    const int64_t want_blocks = (N + 3) / 4;
    hipLaunchKernelGGL(build_f16_kernel, dim3((unsigned)(want_blocks > 8192 ? 8192 : want_blocks)), dim3(256), 0, st, D,
                       N, d, (_Float16 *)D16, (unsigned *)stats);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

TT_EXPORT int tt_index_build_from_bf16(const void *D_bf16, int64_t N, int d, float *D32, void *D16, float *stats,
                                       int reset_stats, tt_stream_t stream)
{
    if (N < 0 || d <= 0 || (d & 3))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_index_build_from_bf16: N=%lld d=%d", (long long)N, d);
    if (N > 0 && (!D_bf16 || !D32))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_index_build_from_bf16: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (stats && reset_stats)
        TT_RC_CHECK(tt_zero_async(stats, 2 * sizeof(float), st));
    if (N == 0)
        return TT_OK;
    const int64_t want_blocks = (N + 3) / 4;
    hipLaunchKernelGGL(build_from_bf16_kernel, dim3((unsigned)(want_blocks > 8192 ? 8192 : want_blocks)), dim3(256), 0,
                       st, (const unsigned short *)D_bf16, N, d, D32, (_Float16 *)D16, (unsigned *)stats);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

TT_EXPORT size_t tt_score_topk_screened_workspace_bytes(int B, int64_t N, int d, int k)
{
    if (B <= 0 || N <= 0)
        return 0;
    return make_splan(B, N, k).ws_bytes + tt_score_topk_workspace_bytes(B, N, d, k);
}

// Where a finished screened search left its per-query statistics in the caller's workspace: int32 [B][2] = (pooled
// candidates, survivors within 2 eps of the k-th best approximate score) -- how much the filter let through on THIS data
// (bench.py reports it for encoder-produced corpora; queries recomputed by the exact fallback keep the screen's counts).
TT_EXPORT size_t tt_score_topk_screened_stats_offset(int B, int64_t N, int d, int k)
{
    (void)d;
    if (B <= 0 || N <= 0)
        return 0;
    return make_splan(B, N, k).stats_off;
}

namespace {
__global__ void seed_fill_kernel(float *seed, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        seed[i] = -3.0e38f; // no sample pass for this corpus size: no information (the floor threshold applies)
}

// phase 0: the whole search.  phase 1: query image + sample pass; seed[q] <- the k_seed-th largest sample maximum (nothing
// else); phase 3: the same, but seed[q][0..k_seed) <- the k_seed LARGEST sample maxima, unordered (a shard's share of the
// union seed: tt_seed_union_f32).  phase 2: the screen with the caller's seed[] as thresholds (the workspace still holds phase 1's query image and
// flags), finish, predicated exact kernels.
int screened_impl(const char *who, int phase, const float *Q, int B, int d, const float *D32, const void *D16, int64_t N, int k,
                  int k_seed, float dmax_norm, int64_t idx_offset, float *out_val, int64_t *out_idx, int32_t *fallback_flag,
                  float *seed, void *workspace, size_t workspace_bytes, void *const *prof_events, hipStream_t st)
{
    if (B <= 0 || N <= 0 || k <= 0)
        return tt_fail(TT_ERR_BAD_SHAPE, "%s: B=%d N=%lld k=%d", who, B, (long long)N, k);
    if (d != 256)
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: d=%d (supported: 256)", who, d);
    if (k > 64)
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: k=%d > 64", who, k);
    if (N >= (int64_t)INT_MAX - 64)
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: N too large; shard the corpus", who);
    if (!(dmax_norm >= 0.0f) || !(dmax_norm < 60000.0f))
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: corpus norm %g outside the fp16 range", who, dmax_norm);
    const bool seed_only = phase == 1 || phase == 3;
    if (!Q || !D16 || !fallback_flag || (!seed_only && (!D32 || !out_val || !out_idx)) || (phase != 0 && !seed))
        return tt_fail(TT_ERR_BAD_SHAPE, "%s: null pointer", who);
    if (seed_only && (k_seed < 1 || k_seed > k))
        return tt_fail(TT_ERR_BAD_SHAPE, "%s: k_seed=%d outside [1, k=%d]", who, k_seed, k);
    const SPlan pl = make_splan(B, N, k);
    const size_t need = pl.ws_bytes + tt_score_topk_workspace_bytes(B, N, d, k);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 255))
        return tt_fail(TT_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", who, workspace_bytes, need);
    char *ws = (char *)workspace;
    // (the fallback flags are initialised by q_image_kernel below -- a kernel, not hipMemsetAsync: a 16-byte-multiple
    //  memset node captured in a HIP graph came back with garbage from the second replay on; ROCm 7.2, found with
    //  GraphedSearch at B=128)

    ScreenParams sp;
    sp.Q = Q;
    sp.D16 = (const _Float16 *)D16;
    sp.B = B;
    sp.N = (int)N;
    sp.k = k;
    sp.n_chunks = pl.n_chunks;
    sp.tiles_per_chunk = pl.tiles_per_chunk;
    sp.n_tiles = pl.n_tiles;
    sp.static_tiles = pl.static_tiles;
    sp.tail_g = pl.tail_g;
    sp.tail_blocks = pl.tail_blocks;
    sp.tail_ctr = (int *)(ws + pl.tailctr_off);
    sp.dmax = dmax_norm;
    sp.cand = (SCand *)(ws + pl.cand_off);
    sp.pcnt = (int *)(ws + pl.pcnt_off);
    sp.flag = fallback_flag;
    sp.max_val = nullptr;
    sp.thr0 = nullptr;
    sp.thr0_stride = k;
    sp.qimg = (const h8 *)(ws + pl.qimg_off);
    sp.qnorm = (const float *)(ws + pl.qnorm_off);
    sp.dbg_thr = nullptr;
    if (phase != 2) {
        hipLaunchKernelGGL(q_image_kernel, dim3(pl.rows_pad / 32), dim3(128), 0, st, Q, B, (h8 *)(ws + pl.qimg_off),
                           (float *)(ws + pl.qnorm_off), fallback_flag, (B + 31) / 32, (int *)(ws + pl.tailctr_off),
                           pl.n_qgroups);
        TT_LAUNCH_CHECK();
    }
    auto launch = [&](const ScreenParams &a, int blocks, bool maxonly) -> int {
        if (pl.stream && pl.q_per_block == 64) {
            if (maxonly)
                hipLaunchKernelGGL((screen_stream_kernel<true, 4>), dim3(blocks), dim3(TW * 64), pl.lds, st, a);
            else
                hipLaunchKernelGGL((screen_stream_kernel<false, 4>), dim3(blocks), dim3(TW * 64), pl.lds, st, a);
        } else if (pl.stream) {
            if (maxonly)
                hipLaunchKernelGGL(screen_stream_kernel<true>, dim3(blocks), dim3(TW * 64), pl.lds, st, a);
            else
                hipLaunchKernelGGL(screen_stream_kernel<false>, dim3(blocks), dim3(TW * 64), pl.lds, st, a);
        } else if (pl.nset == 4) {
            if (maxonly)
                hipLaunchKernelGGL((screen_kernel<true, 4>), dim3(blocks), dim3(SW * 64), pl.lds, st, a);
            else
                hipLaunchKernelGGL((screen_kernel<false, 4>), dim3(blocks), dim3(SW * 64), pl.lds, st, a);
        } else if (pl.nset == 3) {
            if (maxonly)
                hipLaunchKernelGGL((screen_kernel<true, 3>), dim3(blocks), dim3(SW * 64), pl.lds, st, a);
            else
                hipLaunchKernelGGL((screen_kernel<false, 3>), dim3(blocks), dim3(SW * 64), pl.lds, st, a);
        } else if (pl.nset == 2) {
            if (maxonly)
                hipLaunchKernelGGL((screen_kernel<true, 2>), dim3(blocks), dim3(SW * 64), pl.lds, st, a);
            else
                hipLaunchKernelGGL((screen_kernel<false, 2>), dim3(blocks), dim3(SW * 64), pl.lds, st, a);
        } else {
            if (maxonly)
                hipLaunchKernelGGL((screen_kernel<true, 1>), dim3(blocks), dim3(SW * 64), pl.lds, st, a);
            else
                hipLaunchKernelGGL((screen_kernel<false, 1>), dim3(blocks), dim3(SW * 64), pl.lds, st, a);
        }
        TT_LAUNCH_CHECK();
        return TT_OK;
    };
    if (pl.stream && pl.q_per_block == 64) {
        TT_HIP_CHECK(hipFuncSetAttribute((const void *)screen_stream_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
        TT_HIP_CHECK(hipFuncSetAttribute((const void *)screen_stream_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
    } else if (pl.stream) {
        TT_HIP_CHECK(hipFuncSetAttribute((const void *)screen_stream_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
        TT_HIP_CHECK(hipFuncSetAttribute((const void *)screen_stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
    } else {
        const void *fns[8] = {(const void *)screen_kernel<false, 4>, (const void *)screen_kernel<true, 4>,
                              (const void *)screen_kernel<false, 2>, (const void *)screen_kernel<true, 2>,
                              (const void *)screen_kernel<false, 1>, (const void *)screen_kernel<true, 1>,
                              (const void *)screen_kernel<false, 3>, (const void *)screen_kernel<true, 3>};
        const int f0 = pl.nset == 4 ? 0 : (pl.nset == 2 ? 2 : (pl.nset == 3 ? 6 : 4));
        TT_HIP_CHECK(hipFuncSetAttribute(fns[f0], hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
        TT_HIP_CHECK(hipFuncSetAttribute(fns[f0 + 1], hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
    }
    if (phase == 2) {
        sp.thr0 = seed;
        sp.thr0_stride = 1;
    } else if (!pl.sample && seed_only) {
        const int n = phase == 3 ? B * k_seed : B;
        hipLaunchKernelGGL(seed_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, st, seed, n);
        TT_LAUNCH_CHECK();
    } else if (pl.sample) {
        ScreenParams ss = sp;
        ss.N = (int)pl.s_docs;
        ss.n_tiles = pl.s_tiles;
        ss.n_chunks = pl.s_chunks;
        ss.tiles_per_chunk = pl.s_tiles_per_chunk;
        ss.static_tiles = pl.s_tiles;
        ss.tail_blocks = 0;
        ss.max_val = (float *)(ws + pl.smax_val_off);
        int rc = launch(ss, pl.s_blocks, true);
        if (rc != TT_OK)
            return rc;
        if (phase == 3)
            rc = tt_k_largest_list(ss.max_val, B, pl.s_tiles, k_seed, seed, st);
        else
            rc = tt_kth_largest(ss.max_val, B, pl.s_tiles, phase == 1 ? k_seed : k, phase == 1 ? seed : (float *)(ws + pl.sthr_val_off), st);
        if (rc != TT_OK)
            return rc;
        sp.thr0 = (const float *)(ws + pl.sthr_val_off);
        sp.thr0_stride = 1;
    }
    if (seed_only)
        return TT_OK;
    if (prof_events)
        TT_HIP_CHECK(hipEventRecord((hipEvent_t)prof_events[0], st));
    {
        const int rc = launch(sp, pl.n_blocks, false);
        if (rc != TT_OK)
            return rc;
    }
    if (prof_events)
        TT_HIP_CHECK(hipEventRecord((hipEvent_t)prof_events[1], st));

    FinishParams fp;
    fp.Q = Q;
    fp.D32 = D32;
    fp.B = B;
    fp.N = (int)N;
    fp.k = k;
    fp.n_chunks = pl.n_chunks;
    fp.q_per_block = pl.q_per_block;
    fp.dmax = dmax_norm;
    fp.cand = sp.cand;
    fp.pcnt = sp.pcnt;
    fp.flag = fallback_flag;
    fp.idx_offset = idx_offset;
    fp.out_val = out_val;
    fp.out_idx = out_idx;
    fp.stats = (int *)(ws + pl.stats_off);
    hipLaunchKernelGGL(screen_finish_kernel, dim3(B), dim3(256), (size_t)(2 * pl.n_chunks + 1) * sizeof(int), st, fp);
    TT_LAUNCH_CHECK();
    // exact kernel, a no-op unless a workgroup raised the flag; then it rewrites every output row
    return tt_score_topk_f32_pred(Q, B, d, D32, N, k, idx_offset, out_val, out_idx, ws + pl.ws_bytes,
                                  workspace_bytes - pl.ws_bytes, fallback_flag, st);
}
} // namespace

TT_EXPORT int tt_score_topk_screened_f32(const float *Q, int B, int d, const float *D32, const void *D16, int64_t N,
                                         int k, float dmax_norm, int64_t idx_offset, float *out_val, int64_t *out_idx,
                                         int32_t *fallback_flag, void *workspace, size_t workspace_bytes,
                                         void *const *prof_events, tt_stream_t stream)
{
    return screened_impl("tt_score_topk_screened_f32", 0, Q, B, d, D32, D16, N, k, k, dmax_norm, idx_offset, out_val, out_idx,
                         fallback_flag, nullptr, workspace, workspace_bytes, prof_events, (hipStream_t)stream);
}

TT_EXPORT int tt_score_topk_screened_seed_f32(const float *Q, int B, int d, const void *D16, int64_t N, int k, int k_seed,
                                              float dmax_norm, int32_t *fallback_flag, float *seed, void *workspace,
                                              size_t workspace_bytes, tt_stream_t stream)
{
    return screened_impl("tt_score_topk_screened_seed_f32", 1, Q, B, d, nullptr, D16, N, k, k_seed, dmax_norm, 0, nullptr, nullptr,
                         fallback_flag, seed, workspace, workspace_bytes, nullptr, (hipStream_t)stream);
}

TT_EXPORT int tt_score_topk_screened_seed_list_f32(const float *Q, int B, int d, const void *D16, int64_t N, int k, int k_seed,
                                                   float dmax_norm, int32_t *fallback_flag, float *seed_list, void *workspace,
                                                   size_t workspace_bytes, tt_stream_t stream)
{
    return screened_impl("tt_score_topk_screened_seed_list_f32", 3, Q, B, d, nullptr, D16, N, k, k_seed, dmax_norm, 0, nullptr,
                         nullptr, fallback_flag, seed_list, workspace, workspace_bytes, nullptr, (hipStream_t)stream);
}

TT_EXPORT int tt_score_topk_screened_seeded_f32(const float *Q, int B, int d, const float *D32, const void *D16, int64_t N,
                                                int k, float dmax_norm, int64_t idx_offset, float *out_val, int64_t *out_idx,
                                                int32_t *fallback_flag, const float *seed, void *workspace,
                                                size_t workspace_bytes, void *const *prof_events, tt_stream_t stream)
{
    return screened_impl("tt_score_topk_screened_seeded_f32", 2, Q, B, d, D32, D16, N, k, k, dmax_norm, idx_offset, out_val,
                         out_idx, fallback_flag, (float *)seed, workspace, workspace_bytes, prof_events, (hipStream_t)stream);
}


// ------------------------------------------------------------------ test-only: observe the screen's raw scores
// include/tt_debug.h.  Runs the REAL screen kernels (q_image_kernel + the MAXONLY form of screen_stream_kernel /
// screen_kernel<.,NSET>) over the whole corpus and returns one value per (query, 32-document tile): the tile
// maximum of s16 (thr == NULL: accumulators start at +0, the sample pass's arithmetic) or of
// t = fl(sum - thr[query]) (accumulators start at -thr, the main pass's arithmetic).  A test that fills every tile
// with 32 copies of one document reads that document's value.  Not bound by the Python package.
namespace {
struct DbgPlan {
    int q_per_block, n_qgroups, rows_pad, n_tiles, n_chunks, tiles_per_chunk, n_blocks;
    size_t qimg_off, qnorm_off, flag_off, total;
};
bool make_dbg_plan(int B, int64_t N, int form, DbgPlan &pl)
{
    if (form != 0 && form != 1 && form != 2 && form != 4)
        return false;
    pl.q_per_block = form == 0 ? 32 : SW * 16 * form;
    pl.n_qgroups = (B + pl.q_per_block - 1) / pl.q_per_block;
    pl.rows_pad = pl.n_qgroups * pl.q_per_block;
    pl.n_tiles = (int)((N + 31) / 32);
    int want = form == 0 ? (screen_cus() * 2 * TW + pl.n_qgroups - 1) / pl.n_qgroups
                         : (screen_cus() + pl.n_qgroups - 1) / pl.n_qgroups;
    want = want > pl.n_tiles ? pl.n_tiles : want;
    want = want < 1 ? 1 : want;
    pl.tiles_per_chunk = (pl.n_tiles + want - 1) / want;
    pl.n_chunks = (pl.n_tiles + pl.tiles_per_chunk - 1) / pl.tiles_per_chunk;
    const int n_tasks = pl.n_qgroups * pl.n_chunks;
    pl.n_blocks = form == 0 ? (n_tasks + TW - 1) / TW : n_tasks;
    size_t off = 0;
    pl.qimg_off = off;
    off = tt_align_up(off + (size_t)pl.rows_pad * 256 * sizeof(_Float16), 256);
    pl.qnorm_off = off;
    off = tt_align_up(off + (size_t)pl.rows_pad * sizeof(float), 256);
    pl.flag_off = off;
    off = tt_align_up(off + (size_t)(pl.rows_pad / 32 + 1) * sizeof(int), 256);
    pl.total = off;
    return true;
}
} // namespace

TT_EXPORT size_t tt_debug_screen_s16_workspace_bytes(int B, int64_t N, int form)
{
    DbgPlan pl;
    if (B <= 0 || N <= 0 || !make_dbg_plan(B, N, form, pl))
        return 0;
    return pl.total;
}

TT_EXPORT int tt_debug_screen_s16(const float *Q, int B, const void *D16, int64_t N, float dmax_norm, const float *thr,
                                  int form, float *out_t, void *workspace, size_t workspace_bytes, tt_stream_t stream)
{
    hipStream_t st = (hipStream_t)stream;
    DbgPlan pl;
    if (B <= 0 || N <= 0 || N >= (int64_t)INT_MAX - 64 || !make_dbg_plan(B, N, form, pl))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_debug_screen_s16: B=%d N=%lld form=%d", B, (long long)N, form);
    if (!Q || !D16 || !out_t)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_debug_screen_s16: null pointer");
    if (!workspace || workspace_bytes < pl.total || ((uintptr_t)workspace & 255))
        return tt_fail(TT_ERR_WORKSPACE, "tt_debug_screen_s16: workspace %zu < %zu bytes", workspace_bytes, pl.total);
    char *ws = (char *)workspace;
    ScreenParams sp;
    sp.Q = Q;
    sp.D16 = (const _Float16 *)D16;
    sp.B = B;
    sp.N = (int)N;
    sp.k = 1;
    sp.n_chunks = pl.n_chunks;
    sp.tiles_per_chunk = pl.tiles_per_chunk;
    sp.n_tiles = pl.n_tiles;
    sp.static_tiles = pl.n_tiles;
    sp.tail_g = 1;
    sp.tail_blocks = 0;
    sp.tail_ctr = nullptr;
    sp.dmax = dmax_norm;
    sp.cand = nullptr;
    sp.pcnt = nullptr;
    sp.flag = (int *)(ws + pl.flag_off);
    sp.max_val = out_t; // [B][n_tiles]
    sp.thr0 = nullptr;
    sp.thr0_stride = 1;
    sp.qimg = (const h8 *)(ws + pl.qimg_off);
    sp.qnorm = (const float *)(ws + pl.qnorm_off);
    sp.dbg_thr = thr;
    hipLaunchKernelGGL(q_image_kernel, dim3(pl.rows_pad / 32), dim3(128), 0, st, Q, B, (h8 *)(ws + pl.qimg_off),
                       (float *)(ws + pl.qnorm_off), sp.flag, pl.rows_pad / 32, (int *)nullptr, 0);
    TT_LAUNCH_CHECK();
    if (form == 0) {
        const size_t lds = (size_t)TW * TSTAGE * TSLAB_BYTES;
        TT_HIP_CHECK(hipFuncSetAttribute((const void *)screen_stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(screen_stream_kernel<true>, dim3(pl.n_blocks), dim3(TW * 64), lds, st, sp);
    } else {
        const size_t lds = (size_t)SRING * STILE_BYTES;
        const void *fn = form == 4 ? (const void *)screen_kernel<true, 4>
                                   : (form == 2 ? (const void *)screen_kernel<true, 2> : (const void *)screen_kernel<true, 1>);
        TT_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (form == 4)
            hipLaunchKernelGGL((screen_kernel<true, 4>), dim3(pl.n_blocks), dim3(SW * 64), lds, st, sp);
        else if (form == 2)
            hipLaunchKernelGGL((screen_kernel<true, 2>), dim3(pl.n_blocks), dim3(SW * 64), lds, st, sp);
        else
            hipLaunchKernelGGL((screen_kernel<true, 1>), dim3(pl.n_blocks), dim3(SW * 64), lds, st, sp);
    }
    TT_LAUNCH_CHECK();
    return TT_OK;
}
