// K4/K5: fused brute-force scoring + top-k, exact fp32 (gfx950).
//
// Replaces torch.matmul(q, D.t()) + torch.topk (backend/evaluators.py:185-186,
// :269-272; backend/trainer.py:62-65) without ever forming the [B,N] matrix.
//
// Design (DESIGN.md section "K4"):
//  * one WAVE is an independent streaming engine for one task =
//    (32-query tile, contiguous range of 32-document tiles).  No block barrier
//    anywhere: the wave that issues an LDS-DMA is the wave that waits for it
//    (counted s_waitcnt vmcnt) and reads it.
//  * the 32 queries live in VGPRs as the B operand of v_mfma_f32_32x32x2_f32
//    (lane l holds Q[l&31][2s + (l>>5)], s = 0..d/2-1); documents are the A
//    operand, so a lane's 16 accumulator registers are 16 documents of ONE
//    query and the running top-k threshold is a lane-local register.
//  * documents stream HBM -> LDS with global_load_lds_dwordx4 in 4 KiB slabs
//    (32 docs x 32 features = one 128-B line per doc), a 4-deep private ring
//    per wave; the 16-B chunks of a row are XOR-swizzled on the SOURCE address
//    so the ds_read_b128 operand reads are bank-conflict free.
//  * scores are bit-for-bit the ascending-index fp32 FMA chain of
//    oracle/tt_oracle.c:o_score_topk (MFMA f32 = k-ordered fmaf chain; lane
//    half h supplies feature 2s+h, so the chain order is 0,1,2,...,d-1).
//  * selection: a score reaches the slow path only if it is >= its query's
//    current k-th best; the slow path inserts into a sorted per-(wave,query)
//    list in LDS, cooperatively (lane t owns slot t).  Ties: score desc,
//    index asc.  Partial lists go to the workspace; topk_merge_kernel reduces
//    them (also used for the cross-shard merge after the RCCL all-gather).
#include "tt_common.h"

#include <limits.h>
#include <math.h>

namespace {

constexpr int TILE_DOCS = 32;
constexpr int SLAB_BYTES = 32 * 128; // 32 docs x 32 f32
constexpr int NSTAGE = 4;            // ring depth (slabs); NSTAGE-1 in flight
constexpr int WPB = 4;               // waves per block
constexpr int DMA_PER_SLAB = 4;      // global_load_lds_dwordx4 per slab per wave

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct ScoreParams {
    const float *Q;
    const float *D;
    int B;
    int N;
    int k;
    int n_qtiles;
    int n_chunks;
    int tiles_per_chunk;
    int n_tiles;
    int n_tasks;
    float *pval;   // [n_qtiles*32][n_chunks][k]
    int64_t *pidx; // same shape, global indices (idx_offset applied), -1 = empty
    int64_t idx_offset;
};

__device__ __forceinline__ int xcd_remap(int b, int nblk)
{
    // consecutive logical blocks -> blocks that share an XCD (b % 8 equal), bijective.
    if (nblk < 16)
        return b;
    int x = b & 7, q = nblk >> 3, r = nblk & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// Cooperative sorted insert of (s, doc) into one query's list (lane t owns slot t).
// Returns the list's new k-th value.  Order: score desc, index asc.
template <int KPAD>
__device__ __forceinline__ float list_insert(float *lv, int *li, int k, int lane, float s, int doc)
{
    float v = -INFINITY;
    int ix = INT_MAX;
    if (lane < KPAD) {
        v = lv[lane];
        ix = li[lane];
    }
    bool before = (lane < k) && (v > s || (v == s && ix < doc));
    int pos = __popcll(__ballot(before));
    float vprev = __shfl_up(v, 1);
    int iprev = __shfl_up(ix, 1);
    if (pos < k) {
        if (lane == pos) {
            lv[lane] = s;
            li[lane] = doc;
            v = s;
        } else if (lane > pos && lane < k) {
            lv[lane] = vprev;
            li[lane] = iprev;
            v = vprev;
        }
    }
    return __shfl(v, k - 1);
}

template <int NS, int KPAD>
__global__ __launch_bounds__(WPB * 64, 2) void score_topk_kernel(ScoreParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WAVE_LDS = NSTAGE * SLAB_BYTES + 32 * KPAD * 8;
    constexpr int ROW_BYTES = NS * 128;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *ring = smem + wid * WAVE_LDS;
    float *lval = (float *)(ring + NSTAGE * SLAB_BYTES);
    int *lidx = (int *)(lval + 32 * KPAD);

    const int task = xcd_remap(blockIdx.x, gridDim.x) * WPB + wid;
    if (task >= p.n_tasks)
        return; // wave-uniform; the kernel has no block-level barrier
    const int qtile = task % p.n_qtiles;
    const int chunk = task / p.n_qtiles;
    const int t0 = chunk * p.tiles_per_chunk;
    const int t1 = min(t0 + p.tiles_per_chunk, p.n_tiles);
    const int k = p.k;
    const int h = lane >> 5;
    const int j = lane & 31;

    // ---- list init -------------------------------------------------------
    for (int e = lane; e < 32 * KPAD; e += 64) {
        lval[e] = -INFINITY;
        lidx[e] = INT_MAX;
    }

    // ---- query operand: lane (j,h) keeps Q[qrow][2s+h] --------------------
    const int qrow = qtile * 32 + j;
    float qreg[NS * 16];
    {
        const float *qp = p.Q + (size_t)min(qrow, p.B - 1) * (NS * 32) + h;
        const bool live = qrow < p.B;
#pragma unroll
        for (int s = 0; s < NS * 16; ++s)
            qreg[s] = live ? qp[2 * s] : 0.0f;
    }
    float thr = qrow < p.B ? -INFINITY : INFINITY; // padded queries never qualify

    // ---- DMA state -------------------------------------------------------
    // DMA instruction jj moves docs 8jj..8jj+7 of the tile: lane -> (doc 8jj + lane>>3,
    // physical 16-B chunk lane&7).  Logical chunk = physical ^ ((doc>>1)&7) (source swizzle).
    const char *Dbytes = (const char *)p.D;
    int dma_tile = t0, dma_s = 0;
    const char *rowp[DMA_PER_SLAB];
    auto set_rows = [&](int tile) {
#pragma unroll
        for (int jj = 0; jj < DMA_PER_SLAB; ++jj) {
            int di = 8 * jj + (lane >> 3);
            int doc = min(tile * TILE_DOCS + di, p.N - 1);
            int chunk16 = (lane & 7) ^ ((di >> 1) & 7);
            rowp[jj] = Dbytes + (size_t)doc * ROW_BYTES + chunk16 * 16;
        }
    };
    auto dma_issue = [&](int stage) {
        char *dst = ring + stage * SLAB_BYTES;
#pragma unroll
        for (int jj = 0; jj < DMA_PER_SLAB; ++jj)
            __builtin_amdgcn_global_load_lds((gbl_void *)(rowp[jj] + dma_s * 128),
                                             (lds_void *)(dst + jj * 1024), 16, 0, 0);
        if (++dma_s == NS) {
            dma_s = 0;
            dma_tile = min(dma_tile + 1, t1 - 1); // past the end: harmless re-read
            set_rows(dma_tile);
        }
    };

    if (t0 < t1) {
        set_rows(t0);
#pragma unroll
        for (int g = 0; g < NSTAGE - 1; ++g)
            dma_issue(g);

        // read address: lane (i=j, h) wants logical chunk c of row i
        const int rd_swz = (j >> 1) & 7;
        const char *rd_row = ring + j * 128;
        int stage = 0;

        for (int tile = t0; tile < t1; ++tile) {
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                // slab (tile,s) has landed once at most (NSTAGE-2) younger slabs are pending
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_SLAB * (NSTAGE - 2)) : "memory");
                const char *buf = rd_row + stage * SLAB_BYTES;
                f32x4 frag[8];
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    frag[c] = *(const f32x4 *)(buf + ((c ^ rd_swz) << 4));
                // the ring slot consumed one step ago is free once its reads have returned
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dma_issue((stage + NSTAGE - 1) % NSTAGE);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float a0 = h ? frag[c].y : frag[c].x;
                    float a1 = h ? frag[c].w : frag[c].z;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, qreg[16 * s + 2 * c], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, qreg[16 * s + 2 * c + 1], acc, 0, 0, 0);
                }
                stage = (stage + 1) % NSTAGE;
            }

            // ---- epilogue: acc[r] = score(doc tile*32 + (r&3)+8(r>>2)+4h, query j) ----
            const int tile_base = tile * TILE_DOCS;
            const bool partial = tile_base + TILE_DOCS > p.N;
            float m = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r)
                m = fmaxf(m, acc[r]);
            if (__ballot(m >= thr) != 0ull) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int roff = (r & 3) + 8 * (r >> 2);
                    const bool ok = !partial || (tile_base + roff + 4 * h < p.N);
                    unsigned long long mask = __ballot(ok && acc[r] >= thr);
                    while (mask) {
                        const int src = __ffsll((long long)mask) - 1;
                        const float s = __shfl(acc[r], src);
                        const int q = src & 31;
                        const int doc = tile_base + roff + 4 * (src >> 5);
                        const float nt = list_insert<KPAD>(lval + q * KPAD, lidx + q * KPAD, k, lane, s, doc);
                        if (j == q)
                            thr = nt;
                        mask &= ~((2ull << src) - 1ull);
                        mask &= __ballot(ok && acc[r] >= thr);
                    }
                }
            }
        }
    }
    // LDS-DMA still in flight would land after the wave has ended: drain it.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- partial lists out --------------------------------------------------
    for (int q = 0; q < 32; ++q) {
        const int row = qtile * 32 + q;
        if (row >= p.B)
            break;
        if (lane < k) {
            const size_t o = ((size_t)row * p.n_chunks + chunk) * k + lane;
            const int ix = lidx[q * KPAD + lane];
            p.pval[o] = lval[q * KPAD + lane];
            p.pidx[o] = ix == INT_MAX ? -1 : p.idx_offset + ix;
        }
    }
}

// ---------------------------------------------------------------------------
// K5: top-k of M unordered candidates per query, (score desc, index asc).
// k rounds; round r picks the best candidate that ranks strictly after the
// previous pick, so the input is never modified.
// ---------------------------------------------------------------------------
constexpr int MERGE_THREADS = 256;

__device__ __forceinline__ bool ranks_before(float sa, int64_t ia, float sb, int64_t ib)
{
    return sa > sb || (sa == sb && ia < ib);
}

__global__ __launch_bounds__(MERGE_THREADS) void topk_merge_kernel(const float *__restrict__ in_val,
                                                                   const int64_t *__restrict__ in_idx,
                                                                   int M, int k, float *out_val,
                                                                   int64_t *out_idx)
{
    __shared__ float sv[MERGE_THREADS / 64];
    __shared__ int64_t si[MERGE_THREADS / 64];
    const int b = blockIdx.x;
    const float *v = in_val + (size_t)b * M;
    const int64_t *ix = in_idx + (size_t)b * M;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float pv = INFINITY;
    int64_t pi = -1;
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        int64_t bi = INT64_MAX;
        for (int m = tid; m < M; m += MERGE_THREADS) {
            const float cv = v[m];
            const int64_t ci = ix[m];
            if (ci < 0 || !ranks_before(pv, pi, cv, ci))
                continue;
            if (bi == INT64_MAX || ranks_before(cv, ci, bv, bi)) {
                bv = cv;
                bi = ci;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int64_t oi = __shfl_xor(bi, off);
            if (oi != INT64_MAX && (bi == INT64_MAX || ranks_before(ov, oi, bv, bi))) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            sv[wave] = bv;
            si[wave] = bi;
        }
        __syncthreads();
        bv = sv[0];
        bi = si[0];
#pragma unroll
        for (int w = 1; w < MERGE_THREADS / 64; ++w) {
            const float ov = sv[w];
            const int64_t oi = si[w];
            if (oi != INT64_MAX && (bi == INT64_MAX || ranks_before(ov, oi, bv, bi))) {
                bv = ov;
                bi = oi;
            }
        }
        __syncthreads();
        if (bi == INT64_MAX) { // nothing left: (-inf,-1) tail, and nothing ranks after it
            if (tid == 0) {
                out_val[(size_t)b * k + r] = -INFINITY;
                out_idx[(size_t)b * k + r] = -1;
            }
            pv = -INFINITY;
            pi = INT64_MAX;
        } else {
            if (tid == 0) {
                out_val[(size_t)b * k + r] = bv;
                out_idx[(size_t)b * k + r] = bi;
            }
            pv = bv;
            pi = bi;
        }
    }
}

// ---------------------------------------------------------------------------
// Rank of a designated document (BatchEvaluator, evaluators.py:58-65).
// One block per query; scores are the same ascending-index FMA chain.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void score_rank_kernel(const float *__restrict__ Q,
                                                         const float *__restrict__ D, int N, int d,
                                                         const int64_t *__restrict__ target,
                                                         int64_t *rank)
{
    __shared__ int cnt[4];
    const int b = blockIdx.x;
    const float *q = Q + (size_t)b * d;
    const int tg = (int)target[b];
    float st = 0.0f;
    {
        const float *row = D + (size_t)tg * d;
        for (int x = 0; x < d; ++x)
            st = fmaf(q[x], row[x], st);
    }
    int c = 0;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float *row = D + (size_t)n * d;
        float acc = 0.0f;
        for (int x = 0; x < d; x += 4) {
            const f32x4 rv = *(const f32x4 *)(row + x);
            acc = fmaf(q[x], rv.x, acc);
            acc = fmaf(q[x + 1], rv.y, acc);
            acc = fmaf(q[x + 2], rv.z, acc);
            acc = fmaf(q[x + 3], rv.w, acc);
        }
        c += (n != tg) && (acc > st || (acc == st && n < tg));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0)
        cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
        rank[b] = 1 + (int64_t)cnt[0] + cnt[1] + cnt[2] + cnt[3];
}

struct Plan {
    int kpad, n_qtiles, n_tiles, n_chunks, tiles_per_chunk, n_tasks;
    size_t smem, ws_bytes, pidx_off;
};

int device_cus()
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        cus = 256; // MI355X; keeps the workspace query usable without a device
    }
    return cus;
}

Plan make_plan(int B, int64_t N, int k)
{
    Plan pl;
    pl.kpad = k <= 16 ? 16 : 64;
    pl.smem = (size_t)WPB * (NSTAGE * SLAB_BYTES + 32 * pl.kpad * 8);
    const int waves_per_cu = pl.kpad == 16 ? 8 : 4;
    const int slots = device_cus() * waves_per_cu;
    pl.n_qtiles = (B + 31) / 32;
    pl.n_tiles = (int)((N + TILE_DOCS - 1) / TILE_DOCS);
    int want = (slots + pl.n_qtiles - 1) / pl.n_qtiles;
    if (want < 1)
        want = 1;
    if (want > pl.n_tiles)
        want = pl.n_tiles;
    if (want < 1)
        want = 1;
    pl.tiles_per_chunk = pl.n_tiles > 0 ? (pl.n_tiles + want - 1) / want : 1;
    pl.n_chunks = pl.n_tiles > 0 ? (pl.n_tiles + pl.tiles_per_chunk - 1) / pl.tiles_per_chunk : 1;
    pl.n_tasks = pl.n_qtiles * pl.n_chunks;
    const size_t cand = (size_t)pl.n_qtiles * 32 * pl.n_chunks * k;
    pl.pidx_off = tt_align_up(cand * sizeof(float), 256);
    pl.ws_bytes = pl.pidx_off + cand * sizeof(int64_t);
    return pl;
}

template <int NS, int KPAD>
int launch_score(const ScoreParams &sp, const Plan &pl, hipStream_t st)
{
    auto kern = score_topk_kernel<NS, KPAD>;
    TT_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem));
    const int grid = (pl.n_tasks + WPB - 1) / WPB;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WPB * 64), pl.smem, st, sp);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

} // namespace

TT_EXPORT size_t tt_score_topk_workspace_bytes(int B, int64_t N, int d, int k)
{
    (void)d;
    if (B <= 0 || N < 0 || k <= 0)
        return 0;
    return make_plan(B, N, k).ws_bytes;
}

static int score_partials(const float *Q, int B, int d, const float *D, int64_t N, int k, int64_t idx_offset,
                          void *workspace, size_t workspace_bytes, hipStream_t st, Plan *plan_out, const char *who)
{
    if (B <= 0 || N <= 0 || k <= 0)
        return tt_fail(TT_ERR_BAD_SHAPE, "%s: B=%d N=%lld k=%d", who, B, (long long)N, k);
    if (d != 64 && d != 128 && d != 256)
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: d=%d (supported: 64, 128, 256)", who, d);
    if (k > 64)
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: k=%d > 64", who, k);
    if (N >= (int64_t)INT_MAX - 64)
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: N=%lld >= 2^31-64; shard the corpus", who, (long long)N);
    if (!Q || !D)
        return tt_fail(TT_ERR_BAD_SHAPE, "%s: null pointer", who);
    const Plan pl = make_plan(B, N, k);
    if (!workspace || workspace_bytes < pl.ws_bytes)
        return tt_fail(TT_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", who, workspace_bytes, pl.ws_bytes);
    if (((uintptr_t)D & 15) || ((uintptr_t)Q & 3))
        return tt_fail(TT_ERR_BAD_SHAPE, "%s: D must be 16-byte aligned", who);

    ScoreParams sp;
    sp.Q = Q;
    sp.D = D;
    sp.B = B;
    sp.N = (int)N;
    sp.k = k;
    sp.n_qtiles = pl.n_qtiles;
    sp.n_chunks = pl.n_chunks;
    sp.tiles_per_chunk = pl.tiles_per_chunk;
    sp.n_tiles = pl.n_tiles;
    sp.n_tasks = pl.n_tasks;
    sp.pval = (float *)workspace;
    sp.pidx = (int64_t *)((char *)workspace + pl.pidx_off);
    sp.idx_offset = idx_offset;
    *plan_out = pl;
    if (pl.kpad == 16)
        return d == 256 ? launch_score<8, 16>(sp, pl, st) : d == 128 ? launch_score<4, 16>(sp, pl, st) : launch_score<2, 16>(sp, pl, st);
    return d == 256 ? launch_score<8, 64>(sp, pl, st) : d == 128 ? launch_score<4, 64>(sp, pl, st) : launch_score<2, 64>(sp, pl, st);
}

TT_EXPORT int tt_score_topk_partials_f32(const float *Q, int B, int d, const float *D, int64_t N, int k,
                                         int64_t idx_offset, void *workspace, size_t workspace_bytes,
                                         const float **part_val, const int64_t **part_idx, int *part_m,
                                         tt_stream_t stream)
{
    Plan pl;
    int rc = score_partials(Q, B, d, D, N, k, idx_offset, workspace, workspace_bytes, (hipStream_t)stream, &pl,
                            "tt_score_topk_partials_f32");
    if (rc != TT_OK)
        return rc;
    if (part_val)
        *part_val = (const float *)workspace;
    if (part_idx)
        *part_idx = (const int64_t *)((const char *)workspace + pl.pidx_off);
    if (part_m)
        *part_m = pl.n_chunks * k;
    return TT_OK;
}

TT_EXPORT int tt_score_topk_f32(const float *Q, int B, int d, const float *D, int64_t N, int k,
                                int64_t idx_offset, float *out_val, int64_t *out_idx, void *workspace,
                                size_t workspace_bytes, tt_stream_t stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (B < 0 || N < 0 || k <= 0)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_score_topk_f32: B=%d N=%lld k=%d", B, (long long)N, k);
    if (B == 0)
        return TT_OK;
    if (d != 64 && d != 128 && d != 256)
        return tt_fail(TT_ERR_UNSUPPORTED, "tt_score_topk_f32: d=%d (supported: 64, 128, 256)", d);
    if (k > 64)
        return tt_fail(TT_ERR_UNSUPPORTED, "tt_score_topk_f32: k=%d > 64", k);
    if (!out_val || !out_idx)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_score_topk_f32: null output pointer");
    if (N == 0) { // merge over zero candidates writes the (-inf,-1) tail
        hipLaunchKernelGGL(topk_merge_kernel, dim3(B), dim3(MERGE_THREADS), 0, st, (const float *)nullptr,
                           (const int64_t *)nullptr, 0, k, out_val, out_idx);
        TT_LAUNCH_CHECK();
        return TT_OK;
    }
    Plan pl;
    int rc = score_partials(Q, B, d, D, N, k, idx_offset, workspace, workspace_bytes, st, &pl, "tt_score_topk_f32");
    if (rc != TT_OK)
        return rc;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(B), dim3(MERGE_THREADS), 0, st, (const float *)workspace,
                       (const int64_t *)((const char *)workspace + pl.pidx_off), pl.n_chunks * k, k, out_val, out_idx);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

TT_EXPORT int tt_topk_merge(const float *in_val, const int64_t *in_idx, int B, int M, int k, float *out_val,
                            int64_t *out_idx, tt_stream_t stream)
{
    if (B < 0 || M < 0 || k <= 0)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_topk_merge: B=%d M=%d k=%d", B, M, k);
    if (B == 0)
        return TT_OK;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(B), dim3(MERGE_THREADS), 0, (hipStream_t)stream, in_val, in_idx, M,
                       k, out_val, out_idx);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

TT_EXPORT int tt_score_rank_f32(const float *Q, int B, int d, const float *D, int64_t N, const int64_t *target,
                                int64_t *rank, tt_stream_t stream)
{
    if (B < 0 || N <= 0 || d <= 0 || (d & 3))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_score_rank_f32: B=%d N=%lld d=%d", B, (long long)N, d);
    if (N >= INT_MAX)
        return tt_fail(TT_ERR_UNSUPPORTED, "tt_score_rank_f32: N too large");
    if (B == 0)
        return TT_OK;
    hipLaunchKernelGGL(score_rank_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, Q, D, (int)N, d, target, rank);
    TT_LAUNCH_CHECK();
    return TT_OK;
}
