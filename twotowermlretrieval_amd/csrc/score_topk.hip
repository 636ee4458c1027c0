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
//    current k-th best (a lane-local register); the slow path appends (score, doc)
//    to the (wave, query) candidate buffer in the caller's workspace with a
//    fire-and-forget 8-byte store; a nearly full buffer -- and every buffer once
//    at the end -- is rank-compacted by the wave (each lane counts the entries
//    ranking before its own, entries of rank < k are written back sorted, the
//    k-th score becomes the new bound).  Ties: score desc, index asc.  Partial
//    lists go to the workspace; topk_merge_kernel reduces them (also used for the
//    sample-pass maxima and for the cross-shard merge after the RCCL all-gather).
#include "tt_common.h"

#include <limits.h>
#include <math.h>

namespace {

constexpr int TILE_DOCS = 32;
constexpr int SLAB_BYTES = 32 * 128; // 32 docs x 32 f32
constexpr int NSTAGE = 4;            // ring depth (slabs); NSTAGE-1 in flight
#ifndef TT_K4_NT_NSTAGE
#define TT_K4_NT_NSTAGE 4
#endif
constexpr int NSTAGE_NT = TT_K4_NT_NSTAGE; // the one-query-tile build's ring (its waves re-read nothing: more bytes in flight per CU)
constexpr int WPB = 4;               // waves per block
constexpr int DMA_PER_SLAB = 4;      // global_load_lds_dwordx4 per slab per wave
constexpr int PACE_R = 16;            // pacing counter slots per chunk (> pace_lag + 1)
constexpr int PACE_POLLS = 400;       // bound of one wait (~0.3 us per poll)
constexpr int DRAW_POLLS = 1 << 16;   // bound of the wait for a chunk-mate's pool draw (s_sleep 2 = 128 clocks per poll: ~4 ms)
constexpr int DRAW_GAVE_UP = 0x7fffffff;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct Cand;
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
    // tiles [0, static_tiles) are split evenly over the chunks (tiles_per_chunk each); tiles [static_tiles, n_tiles) are a
    // pool of tail_blocks blocks of tail_g tiles per query tile that the waves draw from (tail_ctr[qtile], atomicAdd) once
    // their own range is done: the chip's XCDs do not stream / multiply at one speed (wave end times spread over 10-13 %
    // of the launch with equal shares).  tail_ctr == nullptr: everything static.
    int static_tiles, tail_g, tail_blocks;
    int *tail_ctr;
    // Several query tiles (K4 only): the n_qtiles waves of a chunk sit on one XCD and read the same documents, but nothing
    // keeps them within an L2's reach of each other (32 waves over 160 MB each: 10x the corpus fetched at B = 1024).
    // pace[chunk][PACE_R]: a wave adds 1 to slot b % PACE_R when it has finished its b-th block of pace_g tiles and does
    // not start block b + pace_lag before all n_qtiles waves have (a HINT: the wait is bounded, and a wave that ran into
    // the bound stops waiting for good, so nothing depends on the other waves being resident).  grp_blk[chunk][grp_maxseg]:
    // the pool blocks the chunk's waves take, drawn by whichever of them gets there first (0 = not drawn, -1 = being
    // drawn, else block + 1), so that the pool is walked in step as well.  Both nullptr: every wave for itself.
    int *pace;
    int pace_g, pace_lag;
    int *pace_timeouts; // one word behind the pacing slots: how many waves gave up a pacing wait (the counters are only
                        // coherent among waves that share an XCD's L2: a chunk that straddles XCDs times out once per wave
                        // -- correct results, ~0.15 ms late; tt_score_topk_pace_timeouts_offset lets a bench report it)
    int *grp_blk;
    int grp_maxseg;
    float *pval;   // [n_qtiles*32][n_chunks][k]
    int64_t *pidx; // same shape, global indices (idx_offset applied), -1 = empty
    int64_t idx_offset;
    struct Cand *cand;  // [n_tasks][32][CAP] candidate buffers (workspace)
    const float *thr0;  // optional per-query lower bound of the final k-th score
    int thr0_stride, thr0_off;
    const int *run_if;  // optional device predicate per 32-query tile: tile t is skipped while run_if[t] == 0
    int draw_polls;     // bound of the wait for a chunk-mate's pool draw (DRAW_POLLS; the comparison build can force a give-up)
};

__device__ __forceinline__ int xcd_remap(int b, int nblk)
{
    // consecutive logical blocks -> blocks that share an XCD (b % 8 equal), bijective.
    if (nblk < 16)
        return b;
    int x = b & 7, q = nblk >> 3, r = nblk & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// One candidate: (score, document index relative to D).  8 bytes, one store.
struct __attribute__((aligned(8))) Cand {
    float v;
    int x;
};

__device__ __forceinline__ Cand cand_load_l2(const Cand *p)
{
    // sc1 load: served by L2, never by a stale line in this CU's L1 (the wave re-reads what it wrote)
    unsigned long long u = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Cand c;
    c.v = __uint_as_float((unsigned)u);
    c.x = (int)(u >> 32);
    return c;
}

// Fire-and-forget 8-byte candidate store the compiler's waitcnt pass does not see (it still counts in
// vmcnt: the counted waits of the DMA ring only get stricter, and every reader drains vmcnt(0) first).
__device__ __forceinline__ void cand_store_async(Cand *dst, float v, int x)
{
    const unsigned long long bits = ((unsigned long long)(unsigned)x << 32) | (unsigned long long)__float_as_uint(v);
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(bits) : "memory");
}

// Wave-cooperative compaction of one query's candidate buffer: keeps the k best of its n
// entries (score desc, index asc), SORTED, at the front.  Lane t owns entries t, t+64, ...
// rank(e) = #{entries ranking before e}; ranks are a permutation because (score,index) pairs
// are distinct.  Returns through the references the new count and, when n >= k, the k-th score.
template <int CAP>
__device__ __forceinline__ void compact_query(Cand *base, int n, int k, int lane, int &n_new, float &kth,
                                              bool &have_kth)
{
    constexpr int E = CAP / 64;
    float v[E];
    int x[E], rank[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int e = lane + 64 * i;
        Cand c;
        c.v = -INFINITY;
        c.x = INT_MAX;
        if (e < n)
            c = cand_load_l2(base + e);
        v[i] = c.v;
        x[i] = c.x;
        rank[i] = 0;
    }
#pragma unroll
    for (int i2 = 0; i2 < E; ++i2) {
        const int lim = min(n - 64 * i2, 64);
        for (int l2 = 0; l2 < lim; ++l2) {
            const float sv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i2]), l2));
            const int sx = __builtin_amdgcn_readlane(x[i2], l2);
#pragma unroll
            for (int i = 0; i < E; ++i)
                rank[i] += (sv > v[i] || (sv == v[i] && sx < x[i])) ? 1 : 0;
        }
    }
    have_kth = n >= k;
    kth = -INFINITY;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const bool live = lane + 64 * i < n;
        if (live && rank[i] < k) {
            Cand c;
            c.v = v[i];
            c.x = x[i];
            base[rank[i]] = c;
        }
        const unsigned long long bk = __ballot(live && rank[i] == k - 1);
        if (bk)
            kth = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i]), __ffsll((long long)bk) - 1));
    }
    n_new = min(n, k);
}

// MAXONLY = true is the sample pass: no candidate buffers, each (wave, query) only tracks the
// maximum score over its documents (one partial entry, index = chunk id).
// NT: the document stream with the nt cache policy (aux = 2): for ONE query tile, when every byte is read once by one wave
// (B = 32 over 10M x 256: 1.884 -> 1.84 ms; with 32 query tiles re-reading each chunk from L2 it costs 20 %: 44.2 -> 52.9 ms)
template <int NS, int CAP, bool MAXONLY, bool NT = false>
__global__ __launch_bounds__(WPB * 64, 2) void score_topk_kernel(ScoreParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NST = NT ? NSTAGE_NT : NSTAGE;
    constexpr int WAVE_LDS = NST * SLAB_BYTES;
    constexpr int ROW_BYTES = NS * 128;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *ring = smem + wid * WAVE_LDS;

    const int task = xcd_remap(blockIdx.x, gridDim.x) * WPB + wid;
    if (task >= p.n_tasks)
        return; // wave-uniform; the kernel has no block-level barrier
    const int qtile = task % p.n_qtiles;
    if (p.run_if && p.run_if[qtile] == 0)
        return;
    const int chunk = task / p.n_qtiles;
    int t0 = chunk * p.tiles_per_chunk; // the wave's own range first, then blocks of the pool
    int t1 = min(t0 + p.tiles_per_chunk, p.static_tiles);
    const int k = p.k;
    const int h = lane >> 5;
    const int j = lane & 31;

    // ---- query operand: lane (j,h) keeps Q[qrow][2s+h] --------------------
    // Staged through the (still idle) ring so the global reads are whole 512-B rows instead of
    // 4-byte strided accesses: coalesced global_load_dwordx4 -> swizzled ds_write_b128 -> ds_read_b128.
    const int qrow = qtile * 32 + j;
    float qreg[NS * 16];
    {
        constexpr int F = NS % 4 == 0 ? 128 : (NS % 2 == 0 ? 64 : 32); // features per pass (32 x F floats <= 16 KiB)
        constexpr int NPASS = NS * 32 / F;
        constexpr int CPR = F / 4;                 // 16-byte chunks per row
        constexpr int SWZ = CPR >= 16 ? 15 : CPR - 1;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
            for (int i = 0; i < F / 8; ++i) {
                const int idx = i * 64 + lane;
                const int row = idx / CPR, ch = idx % CPR;
                const int gr = qtile * 32 + row;
                f32x4 v = {0, 0, 0, 0};
                if (gr < p.B)
                    v = *(const f32x4 *)(p.Q + (size_t)gr * (NS * 32) + pass * F + ch * 4);
                *(f32x4 *)(ring + row * (F * 4) + ((ch ^ (row & SWZ)) << 4)) = v;
            }
#pragma unroll
            for (int t = 0; t < CPR; ++t) {
                const f32x4 v = *(const f32x4 *)(ring + j * (F * 4) + ((t ^ (j & SWZ)) << 4));
                qreg[pass * (F / 2) + 2 * t] = h ? v.y : v.x;
                qreg[pass * (F / 2) + 2 * t + 1] = h ? v.w : v.z;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // reads done before the region is rewritten
        }
    }
    // Selection state of query j, replicated in lanes j and j+32:
    //   thr = a proven lower bound of the query's final k-th score (documents below it are dropped),
    //   cnt = entries in this wave's candidate buffer for the query.
    // thr starts from the caller's bound (k-th score over a sample of the corpus, or -inf).
    float thr = INFINITY; // padded queries never qualify
    if (qrow < p.B)
        thr = (!MAXONLY && p.thr0) ? p.thr0[(size_t)qrow * p.thr0_stride + p.thr0_off] : -INFINITY;
    int cnt = 0;
    float runmax = -INFINITY;
    Cand *const cbase = p.cand + ((size_t)task * 32 + j) * CAP; // this lane's query buffer

    auto compact_where = [&](unsigned long long qmask) { // qmask: bit q set -> compact query q
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // candidate stores have reached L2
        while (qmask) {
            const int q = __ffsll((long long)qmask) - 1;
            qmask &= qmask - 1;
            const int n = __builtin_amdgcn_readlane(cnt, q);
            int n_new;
            float kth;
            bool have;
            compact_query<CAP>(p.cand + ((size_t)task * 32 + q) * CAP, n, k, lane, n_new, kth, have);
            if (j == q) {
                cnt = n_new;
                if (have)
                    thr = kth; // >= old thr: every entry was admitted under a bound <= it
            }
        }
    };

    // ---- pacing (see ScoreParams::pace) ------------------------------------
    constexpr bool PACED = !MAXONLY && !NT; // (NT is the one-query-tile build)
    int *const pace = (PACED && p.pace) ? p.pace + (size_t)chunk * PACE_R : nullptr;
    bool waits = pace != nullptr; // false once a wait ran into its bound
    int gb = 0;                   // blocks of pace_g tiles this wave has finished
    int in_blk = 0;               // tiles of the current block done
    int seg = 0;                  // pool blocks this wave has taken
    bool gave_up = false;         // a shared pool draw timed out (wave-uniform)
    auto pace_gate = [&](int b) { // all n_qtiles waves of the chunk have finished their block b?
        const int need = p.n_qtiles * (b / PACE_R + 1);
        const int *slot = pace + (b % PACE_R);
        for (int polls = 0;; ++polls) {
            int v;
            // scalar load past the scalar cache: its own counter (lgkmcnt), so the LDS-DMA ring stays in flight
            asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(slot) : "memory");
            if (v >= need)
                return;
            if (polls == PACE_POLLS) {
                waits = false;
                if (lane == 0)
                    atomicAdd(p.pace_timeouts, 1);
                return;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    };

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
                                             (lds_void *)(dst + jj * 1024), 16, 0, NT ? 2 : 0);
        if (++dma_s == NS) {
            dma_s = 0;
            dma_tile = min(dma_tile + 1, t1 - 1); // past the end: harmless re-read
            set_rows(dma_tile);
        }
    };

    for (;;) { // segments: the own range, then pool blocks
    if (t0 < t1) {
        set_rows(t0);
#pragma unroll
        for (int g = 0; g < NST - 1; ++g)
            dma_issue(g);

        // read address: lane (i=j, h) wants logical chunk c of row i
        const int rd_swz = (j >> 1) & 7;
        const char *rd_row = ring + j * 128;
        int stage = 0;

        for (int tile = t0; tile < t1; ++tile) {
            if (PACED && waits && in_blk == 0 && gb >= p.pace_lag)
                pace_gate(gb - p.pace_lag);
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                // slab (tile,s) has landed once at most (NST-2) younger slabs are pending
                // (candidate stores also count in vmcnt: they only make this wait stricter)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_SLAB * (NST - 2)) : "memory");
                const char *buf = rd_row + stage * SLAB_BYTES;
                f32x4 frag[8];
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    frag[c] = *(const f32x4 *)(buf + ((c ^ rd_swz) << 4));
                // the ring slot consumed one step ago is free once its reads have returned
#ifndef TT_K4_NO_LGKM0
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                dma_issue((stage + NST - 1) % NST);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float a0 = h ? frag[c].y : frag[c].x;
                    float a1 = h ? frag[c].w : frag[c].z;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, qreg[16 * s + 2 * c], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, qreg[16 * s + 2 * c + 1], acc, 0, 0, 0);
                }
                stage = (stage + 1) % NST;
            }

            // ---- epilogue: acc[r] = score(doc tile*32 + (r&3)+8(r>>2)+4h, query j) ----
            const int tile_base = tile * TILE_DOCS;
            const bool partial = tile_base + TILE_DOCS > p.N;
            if (MAXONLY) {
                float m = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int doc = tile_base + (r & 3) + 8 * (r >> 2) + 4 * h;
                    m = fmaxf(m, (!partial || doc < p.N) ? acc[r] : -INFINITY);
                }
                runmax = fmaxf(runmax, m);
                continue;
            }
            float m = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r)
                m = fmaxf(m, acc[r]);
            if (__ballot(m >= thr) != 0ull) {
                // Append pass.  The store is inline asm on purpose: a compiler-visible global store
                // (or the compaction's loads) inside this loop makes hipcc emit s_waitcnt vmcnt(0) at
                // every join, which drains the LDS-DMA ring on each tile that has a candidate.
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int doc = tile_base + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const bool c = (!partial || doc < p.N) && acc[r] >= thr;
                    const unsigned long long mask = __ballot(c);
                    if (mask == 0ull)
                        continue;
                    const int c_lo = (int)((mask >> j) & 1ull), c_hi = (int)((mask >> (j + 32)) & 1ull);
                    if (c)
                        cand_store_async(cbase + cnt + (h ? c_lo : 0), acc[r], doc);
                    cnt += c_lo + c_hi;
                }
                // one tile adds at most 32 entries per query: compact while there is still room for that
                const unsigned long long full = __ballot(cnt > CAP - 34) & 0xffffffffull;
                if (full)
                    compact_where(full);
            }
            if (PACED && pace && (++in_blk == p.pace_g || tile == t1 - 1)) {
                // block done (counted whether or not this wave still waits for the others: they may wait for it)
                if (lane == 0)
                    asm volatile("global_atomic_add %0, %1, off" ::"v"(pace + (gb % PACE_R)), "v"(1) : "memory");
                ++gb;
                in_blk = 0;
            }
        }
    }
    if (MAXONLY || !p.tail_ctr)
        break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the over-prefetch of the segment's end has landed: the ring is free
    int blk = 0;
    if (PACED && p.grp_blk) {
        // the chunk's waves take the same pool blocks in the same order: the first to get here draws for all of them
        if (seg >= p.grp_maxseg)
            break;
        if (lane == 0) {
            int *slot = p.grp_blk + (size_t)chunk * p.grp_maxseg + seg;
            int v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool drew = false;
            if (v == 0) {
                int expected = 0;
                if (__hip_atomic_compare_exchange_strong(slot, &expected, -1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT)) {
                    v = atomicAdd(p.tail_ctr, 1) + 1;
                    __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    drew = true;
                } else {
                    v = expected;
                }
            }
            // the drawing wave is between its two atomics (it waits for nobody: ~2 us).  Bounded like every other cross-wave
            // wait of the library: DRAW_POLLS sleeps (~4 ms) and the wave gives up -- its partial lists then say so (below), and
            // tt_score_topk_f32 redoes the wave's query tile on the static split, which waits for nobody
            if (p.draw_polls < 0 && !drew) // (comparison build, TT_DRAW_POLLS=-1: every wave that did not draw itself gives up)
                v = -1;
            for (int polls = 0; v < 0 && polls < p.draw_polls; ++polls) {
                __builtin_amdgcn_s_sleep(2);
                v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            blk = v < 0 ? DRAW_GAVE_UP : v - 1;
        }
        ++seg;
    } else if (lane == 0) {
        blk = atomicAdd(p.tail_ctr + qtile, 1);
    }
    blk = __builtin_amdgcn_readfirstlane(blk);
    if (blk == DRAW_GAVE_UP)
        gave_up = true;
    if (blk >= p.tail_blocks)
        break;
    t0 = p.static_tiles + blk * p.tail_g;
    t1 = min(t0 + p.tail_g, p.n_tiles);
    dma_tile = t0;
    dma_s = 0;
    }
    // LDS-DMA still in flight would land after the wave has ended: drain it (and the stores).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (MAXONLY) {
        runmax = fmaxf(runmax, __shfl_xor(runmax, 32));
        if (h == 0 && qrow < p.B) {
            const size_t o = (size_t)qrow * p.n_chunks + chunk;
            p.pval[o] = runmax;
            p.pidx[o] = t0 < t1 ? (int64_t)chunk : -1;
        }
        return;
    }

    // ---- final compaction + partial lists out ---------------------------------
    {
        const unsigned long long over = __ballot(cnt > k) & 0xffffffffull;
        if (over)
            compact_where(over);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // 32*k output slots, lanes take them round-robin so the candidate loads overlap
    const int rows_live = min(32, p.B - qtile * 32);
    for (int it = lane; it < rows_live * k; it += 64) {
        const int q = it / k, t = it - q * k;
        const int n = __shfl(cnt, q);
        const size_t o = ((size_t)(qtile * 32 + q) * p.n_chunks + chunk) * k + t;
        Cand c;
        c.v = -INFINITY;
        c.x = -1;
        if (t < n)
            c = cand_load_l2(p.cand + ((size_t)task * 32 + q) * CAP + t);
        // a wave that gave up a pool draw has not seen every document: its lists carry (+inf, TT_TOPK_INVALID_INDEX + t),
        // which the merge ranks first -- tt_score_topk_f32 looks for them in the merged lists and redoes those query tiles
        // (redo_flag_kernel below); a caller of the partials entry point sees them as they are
        p.pval[o] = gave_up ? INFINITY : c.v;
        p.pidx[o] = gave_up ? (int64_t)TT_TOPK_INVALID_INDEX + t : (t < n ? p.idx_offset + c.x : -1);
    }
}

// ---------------------------------------------------------------------------
// K4w: the same kernel for wide embeddings, 256 < d <= 512 (HIDDEN_DIM up to 512 is what the encoder supports).
// 32 queries x 512 features do not fit a wave's registers as 32x32x2 B operands, so a task is a 16-QUERY tile on
// v_mfma_f32_16x16x4_f32: lane (n = l&15, kq = l>>4) keeps Q[n][4s+kq], s < d/4 (128 VGPRs at d = 512) and its
// 8 accumulator registers are 8 documents of query n (two 16-document sub-tiles x rows 4kq..4kq+3).  A chain of
// 16x16x4 MFMAs with k ascending is bit-identical to the sequential fmaf chain, like the 32x32x2 form
// (tools/experiments/mfma16_order.hip), so the oracle parity carries over.  Ring, DMA, thresholds, candidate
// buffers, compaction, partial lists and merge are K4's; the launch is bound by HBM streaming (d * 4 bytes per doc).
// ---------------------------------------------------------------------------
template <int NS, int CAP, bool MAXONLY, bool NT = false>
__global__ __launch_bounds__(WPB * 64, 2) void score_topk16_kernel(ScoreParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WAVE_LDS = NSTAGE * SLAB_BYTES;
    constexpr int ROW_BYTES = NS * 128;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *ring = smem + wid * WAVE_LDS;

    const int task = xcd_remap(blockIdx.x, gridDim.x) * WPB + wid;
    if (task >= p.n_tasks)
        return; // wave-uniform; the kernel has no block-level barrier
    const int qtile = task % p.n_qtiles; // 16 queries
    if (p.run_if && p.run_if[qtile >> 1] == 0)
        return;
    const int chunk = task / p.n_qtiles;
    int t0 = chunk * p.tiles_per_chunk; // the wave's own range first, then blocks of the pool
    int t1 = min(t0 + p.tiles_per_chunk, p.static_tiles);
    const int k = p.k;
    const int n = lane & 15, kq = lane >> 4;

    const int qrow = qtile * 16 + n;
    float qreg[NS * 8];
    {
        const float *qp = p.Q + (size_t)min(qrow, p.B - 1) * (NS * 32) + kq;
#pragma unroll
        for (int s = 0; s < NS * 8; ++s)
            qreg[s] = qrow < p.B ? qp[4 * s] : 0.0f;
    }
    float thr = INFINITY; // padded queries never qualify
    if (qrow < p.B)
        thr = (!MAXONLY && p.thr0) ? p.thr0[(size_t)qrow * p.thr0_stride + p.thr0_off] : -INFINITY;
    int cnt = 0; // the same value in the four lanes of a query
    float runmax = -INFINITY;
    Cand *const cbase = p.cand + ((size_t)task * 16 + n) * CAP;

    auto compact_where = [&](unsigned qmask) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        while (qmask) {
            const int q = __ffs((int)qmask) - 1;
            qmask &= qmask - 1;
            const int nn = __builtin_amdgcn_readlane(cnt, q);
            int n_new;
            float kth;
            bool have;
            compact_query<CAP>(p.cand + ((size_t)task * 16 + q) * CAP, nn, k, lane, n_new, kth, have);
            if (n == q) {
                cnt = n_new;
                if (have)
                    thr = kth;
            }
        }
    };

    // ---- DMA: K4's (slab = 32 docs x 32 features; instruction jj moves docs 8jj..8jj+7; lane -> (doc 8jj +
    //      lane>>3, physical 16-B chunk lane&7); logical chunk = physical ^ ((doc>>1)&7)) ----
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
                                             (lds_void *)(dst + jj * 1024), 16, 0, NT ? 2 : 0);
        if (++dma_s == NS) {
            dma_s = 0;
            dma_tile = min(dma_tile + 1, t1 - 1); // past the end: harmless re-read
            set_rows(dma_tile);
        }
    };

    for (;;) { // segments: the own range, then pool blocks
    if (t0 < t1) {
        set_rows(t0);
#pragma unroll
        for (int gi = 0; gi < NSTAGE - 1; ++gi)
            dma_issue(gi);
        // A element (sub-tile u, k-step t of the slab): row 16u + n, logical chunk t, float kq of the chunk
        const char *rd[2];
        int rsw[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = 16 * u + n;
            rd[u] = ring + r * 128 + 4 * kq;
            rsw[u] = (r >> 1) & 7;
        }
        int stage = 0;
        for (int tile = t0; tile < t1; ++tile) {
            f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_SLAB * (NSTAGE - 2)) : "memory");
                float a[2][8];
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        a[u][t] = *(const float *)(rd[u] + stage * SLAB_BYTES + ((t ^ rsw[u]) << 4));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dma_issue((stage + NSTAGE - 1) % NSTAGE);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][t], qreg[8 * s + t], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][t], qreg[8 * s + t], acc[1], 0, 0, 0);
                }
                stage = (stage + 1) % NSTAGE;
            }
            // ---- epilogue: acc[u][r] = score(doc tile*32 + 16u + 4kq + r, query n) ----
            const int tile_base = tile * TILE_DOCS;
            const bool partial = tile_base + TILE_DOCS > p.N;
            float m = -INFINITY;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    m = fmaxf(m, (!partial || tile_base + 16 * u + 4 * kq + r < p.N) ? acc[u][r] : -INFINITY);
            if (MAXONLY) {
                runmax = fmaxf(runmax, m);
                continue;
            }
            if (__ballot(m >= thr) != 0ull) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int doc = tile_base + 16 * u + 4 * kq + r;
                        const bool c = (!partial || doc < p.N) && acc[u][r] >= thr;
                        const unsigned long long mask = __ballot(c);
                        if (mask == 0ull)
                            continue;
                        // the four lanes of query n sit at n, n+16, n+32, n+48: slots in kq order
                        const int b0 = (int)((mask >> n) & 1ull), b1 = (int)((mask >> (n + 16)) & 1ull);
                        const int b2 = (int)((mask >> (n + 32)) & 1ull), b3 = (int)((mask >> (n + 48)) & 1ull);
                        const int before = (kq > 0 ? b0 : 0) + (kq > 1 ? b1 : 0) + (kq > 2 ? b2 : 0);
                        if (c)
                            cand_store_async(cbase + cnt + before, acc[u][r], doc);
                        cnt += b0 + b1 + b2 + b3;
                    }
                // one tile adds at most 32 entries per query: compact while there is still room for that
                const unsigned long long full = __ballot(cnt > CAP - 34) & 0xffffull;
                if (full)
                    compact_where((unsigned)full);
            }
        }
    }
    if (MAXONLY || !p.tail_ctr)
        break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the over-prefetch of the segment's end has landed: the ring is free
    int blk = 0;
    if (lane == 0)
        blk = atomicAdd(p.tail_ctr + qtile, 1);
    blk = __builtin_amdgcn_readfirstlane(blk);
    if (blk >= p.tail_blocks)
        break;
    t0 = p.static_tiles + blk * p.tail_g;
    t1 = min(t0 + p.tail_g, p.n_tiles);
    dma_tile = t0;
    dma_s = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (MAXONLY) {
        runmax = fmaxf(runmax, __shfl_xor(runmax, 16));
        runmax = fmaxf(runmax, __shfl_xor(runmax, 32));
        if (kq == 0 && qrow < p.B) {
            const size_t o = (size_t)qrow * p.n_chunks + chunk;
            p.pval[o] = runmax;
            p.pidx[o] = t0 < t1 ? (int64_t)chunk : -1;
        }
        return;
    }
    {
        const unsigned long long over = __ballot(cnt > k) & 0xffffull;
        if (over)
            compact_where((unsigned)over);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int rows_live = min(16, p.B - qtile * 16);
    for (int it = lane; it < rows_live * k; it += 64) {
        const int q = it / k, t = it - q * k;
        const int nn = __shfl(cnt, q);
        const size_t o = ((size_t)(qtile * 16 + q) * p.n_chunks + chunk) * k + t;
        Cand c;
        c.v = -INFINITY;
        c.x = -1;
        if (t < nn)
            c = cand_load_l2(p.cand + ((size_t)task * 16 + q) * CAP + t);
        p.pval[o] = c.v;
        p.pidx[o] = t < nn ? p.idx_offset + c.x : -1;
    }
}

// ---------------------------------------------------------------------------
// K5: top-k of M unordered candidates per query, (score desc, index asc).
// One block per query.  The candidates are scanned ONCE: valid entries (idx >= 0; most
// partial-list slots are padding) are packed into an LDS pool; whenever the pool could overflow
// it is reduced to its k best.  The final reduction emits the sorted top-k.  A reduction is k
// rounds of block-wide arg-best over the pool, each round picking the best entry that ranks
// strictly after the previous pick.
// ---------------------------------------------------------------------------
constexpr int MERGE_THREADS = 256;
constexpr int MERGE_POOL = 6144;                 // LDS pool entries (12 B each)
constexpr int MERGE_SEG = MERGE_THREADS * 16;    // candidates scanned between overflow checks
constexpr int MERGE_KMAX = 64;

__device__ __forceinline__ bool ranks_before(float sa, int64_t ia, float sb, int64_t ib)
{
    return sa > sb || (sa == sb && ia < ib);
}

// Block-wide: the best pool entry ranking strictly after (pv,pi).  Returns bi == INT64_MAX if none.
__device__ __forceinline__ void pool_next_best(const float *pool_v, const int64_t *pool_i, int n, float pv,
                                               int64_t pi, float *red_v, int64_t *red_i, float &bv, int64_t &bi)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bv = -INFINITY;
    bi = INT64_MAX;
    for (int m = tid; m < n; m += MERGE_THREADS) {
        const float cv = pool_v[m];
        const int64_t ci = pool_i[m];
        if (!ranks_before(pv, pi, cv, ci))
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
        red_v[wave] = bv;
        red_i[wave] = bi;
    }
    __syncthreads();
    bv = red_v[0];
    bi = red_i[0];
#pragma unroll
    for (int w = 1; w < MERGE_THREADS / 64; ++w) {
        const float ov = red_v[w];
        const int64_t oi = red_i[w];
        if (oi != INT64_MAX && (bi == INT64_MAX || ranks_before(ov, oi, bv, bi))) {
            bv = ov;
            bi = oi;
        }
    }
    __syncthreads();
}

// Candidate m of row b lives in segment m / seg_len at [b][m % seg_len]; segment s starts seg_stride BYTES
// after segment s-1 (one segment of length M = a plain [B,M] matrix; `world` segments of length k' = the
// all-gathered per-shard lists of ShardedIndex, read in place).
__global__ __launch_bounds__(MERGE_THREADS) void topk_merge_kernel(const float *__restrict__ in_val,
                                                                   const int64_t *__restrict__ in_idx,
                                                                   int M, int k, float *out_val,
                                                                   int64_t *out_idx, const int *run_if,
                                                                   int seg_len, size_t seg_stride)
{
    if (run_if && run_if[blockIdx.x >> 5] == 0)
        return;
    __shared__ float pool_v[MERGE_POOL];
    __shared__ int64_t pool_i[MERGE_POOL];
    __shared__ float top_v[MERGE_KMAX];
    __shared__ int64_t top_i[MERGE_KMAX];
    __shared__ float red_v[MERGE_THREADS / 64];
    __shared__ int64_t red_i[MERGE_THREADS / 64];
    __shared__ int pool_n;
    const int b = blockIdx.x, tid = threadIdx.x;
    auto v_at = [&](int m) -> float {
        const int sg = m / seg_len, wi = m - sg * seg_len;
        return ((const float *)((const char *)in_val + (size_t)sg * seg_stride))[(size_t)b * seg_len + wi];
    };
    auto ix_at = [&](int m) -> int64_t {
        const int sg = m / seg_len, wi = m - sg * seg_len;
        return ((const int64_t *)((const char *)in_idx + (size_t)sg * seg_stride))[(size_t)b * seg_len + wi];
    };
    if (tid == 0)
        pool_n = 0;
    __syncthreads();

    // reduce the pool to its k best (sorted) in place; returns the new size
    auto reduce_pool = [&]() {
        const int n = pool_n;
        float pv = INFINITY;
        int64_t pi = -1;
        int kept = 0;
        for (int r = 0; r < k; ++r) {
            float bv;
            int64_t bi;
            pool_next_best(pool_v, pool_i, n, pv, pi, red_v, red_i, bv, bi);
            if (bi == INT64_MAX)
                break;
            if (tid == 0) {
                top_v[r] = bv;
                top_i[r] = bi;
            }
            pv = bv;
            pi = bi;
            kept = r + 1;
        }
        __syncthreads();
        if (tid < kept) {
            pool_v[tid] = top_v[tid];
            pool_i[tid] = top_i[tid];
        }
        if (tid == 0)
            pool_n = kept;
        __syncthreads();
        return kept;
    };

    for (int base = 0; base < M; base += MERGE_SEG) {
        if (pool_n + MERGE_SEG > MERGE_POOL) // block-uniform (pool_n read after a barrier)
            reduce_pool();
        const int end = min(base + MERGE_SEG, M);
        int64_t ci[MERGE_SEG / MERGE_THREADS];
#pragma unroll
        for (int u = 0; u < MERGE_SEG / MERGE_THREADS; ++u) { // all index loads in flight together
            const int m = base + tid + u * MERGE_THREADS;
            ci[u] = m < end ? ix_at(m) : -1;
        }
#pragma unroll
        for (int u = 0; u < MERGE_SEG / MERGE_THREADS; ++u) {
            if (ci[u] >= 0) {
                const int slot = atomicAdd(&pool_n, 1);
                pool_v[slot] = v_at(base + tid + u * MERGE_THREADS);
                pool_i[slot] = ci[u];
            }
        }
        __syncthreads();
    }
    const int kept = reduce_pool();
    if (tid < k) {
        out_val[(size_t)b * k + tid] = tid < kept ? pool_v[tid] : -INFINITY;
        out_idx[(size_t)b * k + tid] = tid < kept ? pool_i[tid] : -1;
    }
}

// ---------------------------------------------------------------------------
// k-th largest of M values per row (threshold seeding): 4-pass radix select on the order-preserving
// integer image of the floats; one block per row, cost independent of k.  -inf when M < k.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned order_key(float f)
{
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ float order_key_to_float(unsigned key)
{
    return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}

// top_out (nullable): [rows][k], receives the row's k largest values, UNORDERED (the values above the k-th in arrival
// order, then copies of the k-th; -inf padding when M < k) -- what a shard contributes to the union seed of a sharded
// search (tt_score_topk_screened_seed_list_f32).
__global__ __launch_bounds__(256) void kth_largest_kernel(const float *__restrict__ vals, int M, int k,
                                                          float *__restrict__ out, const int *run_if,
                                                          float *__restrict__ top_out)
{
    if (run_if && run_if[blockIdx.x >> 5] == 0)
        return;
    __shared__ int hist[256];
    __shared__ int sel[2];
    __shared__ int wpos;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const float *v = vals + (size_t)row * M;
    if (M < k) {
        if (tid == 0 && out)
            out[row] = -INFINITY;
        if (top_out)
            for (int i = tid; i < k; i += 256)
                top_out[(size_t)row * k + i] = i < M ? v[i] : -INFINITY;
        return;
    }
    unsigned prefix = 0u, mask = 0u;
    int k_rem = k;
    // up to 256 x KTH_REG values (every seeding pass of the searches) are read ONCE into registers; the four passes then
    // run on those (four passes of dependent global loads were 16 us per search, three quarters of this kernel)
    constexpr int KTH_REG = 16;
    const bool in_regs = M <= 256 * KTH_REG;
    unsigned keys[KTH_REG];
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < KTH_REG; ++i) {
            const int m = tid + 256 * i;
            keys[i] = m < M ? order_key(v[m]) : 0u; // (key 0 = below every float: never selected while M >= k)
        }
    }
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
        if (in_regs) {
#pragma unroll
            for (int i = 0; i < KTH_REG; ++i)
                if (tid + 256 * i < M && (keys[i] & mask) == prefix)
                    atomicAdd(&hist[(keys[i] >> shift) & 255u], 1);
        } else {
            for (int m = tid; m < M; m += 256) {
                const unsigned key = order_key(v[m]);
                if ((key & mask) == prefix)
                    atomicAdd(&hist[(key >> shift) & 255u], 1);
            }
        }
        __syncthreads();
        if (tid < 64) { // one wave: suffix sums over the 256 bins (4 bins per lane, high bins first)
            int c[4], s = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c[i] = hist[255 - (lane * 4 + i)];
                s += c[i];
            }
            int incl = s; // inclusive scan over lanes
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(incl, off);
                if (lane >= off)
                    incl += y;
            }
            int above = incl - s; // entries in strictly higher bins than this lane's first
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (above < k_rem && above + c[i] >= k_rem) {
                    sel[0] = 255 - (lane * 4 + i);
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
    const float kth = order_key_to_float(prefix);
    if (tid == 0 && out)
        out[row] = kth;
    if (top_out) {
        // k - k_rem values are strictly above the k-th (k_rem = the k-th's rank among its equals): they go first, in
        // arrival order; the remaining k_rem slots are copies of the k-th
        if (tid == 0)
            wpos = 0;
        __syncthreads();
        float *dst = top_out + (size_t)row * k;
        if (in_regs) {
#pragma unroll
            for (int i = 0; i < KTH_REG; ++i)
                if (tid + 256 * i < M && keys[i] > prefix)
                    dst[atomicAdd(&wpos, 1)] = order_key_to_float(keys[i]);
        } else {
            for (int m = tid; m < M; m += 256)
                if (order_key(v[m]) > prefix)
                    dst[atomicAdd(&wpos, 1)] = v[m];
        }
        for (int i = k - k_rem + tid; i < k; i += 256)
            dst[i] = kth;
    }
}

// ---------------------------------------------------------------------------
// Rank of a designated document (BatchEvaluator, evaluators.py:58-65).
// One block per query; scores are the same ascending-index FMA chain, one lane per document (the chain is
// sequential by definition).  A wave takes 64 documents at a time and stages them 32 features at a time through LDS:
// the global reads are whole 128-byte row segments (8 lanes x 16 B per row), the chain reads its own row from a
// 33-float-stride image (conflict-free).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void score_rank_kernel(const float *__restrict__ Q,
                                                         const float *__restrict__ D, int N, int d,
                                                         const int64_t *__restrict__ target,
                                                         int64_t *rank)
{
    __shared__ float qs[512];
    __shared__ float stage[4][64 * 33];
    __shared__ int cnt[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tg = (int)target[b];
    for (int x = threadIdx.x; x < d; x += 256)
        qs[x] = Q[(size_t)b * d + x];
    __syncthreads();
    float st = 0.0f;
    {
        const float *row = D + (size_t)tg * d;
        for (int x = 0; x < d; ++x)
            st = fmaf(qs[x], row[x], st);
    }
    float *img = stage[wv];
    int c = 0;
    for (int n0 = wv * 64; n0 < N; n0 += 256) { // wave-uniform trip count
        float acc = 0.0f;
        for (int x0 = 0; x0 < d; x0 += 32) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { // rows 8 i + lane / 8, 16-byte chunk lane % 8
                const int r = 8 * i + (lane >> 3), n = min(n0 + r, N - 1);
                f32x4 v = {0, 0, 0, 0};
                if (x0 + 4 * (lane & 7) < d) // d is a multiple of 4, not necessarily of 32
                    v = *(const f32x4 *)(D + (size_t)n * d + x0 + 4 * (lane & 7));
                float *dst = img + r * 33 + 4 * (lane & 7);
                dst[0] = v.x;
                dst[1] = v.y;
                dst[2] = v.z;
                dst[3] = v.w;
            }
            __builtin_amdgcn_wave_barrier();
            const float *mine = img + lane * 33;
            const int xe = min(32, d - x0);
            for (int x = 0; x < xe; ++x)
                acc = fmaf(qs[x0 + x], mine[x], acc);
            __builtin_amdgcn_wave_barrier();
        }
        const int n = n0 + lane;
        c += (n < N) && (n != tg) && (acc > st || (acc == st && n < tg));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        c += __shfl_xor(c, off);
    if (lane == 0)
        cnt[wv] = c;
    __syncthreads();
    if (threadIdx.x == 0)
        rank[b] = 1 + (int64_t)cnt[0] + cnt[1] + cnt[2] + cnt[3];
}

// All scores of a query: S[b][n] = the ascending-index FMA chain of <Q[b], D[n]> (the scores score_topk selects from),
// for the callers that blend the dense score of EVERY document with another signal (backend/simple_hybrid.py:53-56).
// grid (ceil(N / 256), B); a wave takes 64 documents, staged 32 features at a time through LDS like score_rank_kernel.
__global__ __launch_bounds__(256) void score_all_kernel(const float *__restrict__ Q, const float *__restrict__ D, int N,
                                                        int d, float *__restrict__ S)
{
    __shared__ float qs[512];
    __shared__ float stage[4][64 * 33];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int x = threadIdx.x; x < d; x += 256)
        qs[x] = Q[(size_t)b * d + x];
    __syncthreads();
    float *img = stage[wv];
    const int n0 = blockIdx.x * 256 + wv * 64;
    if (n0 >= N)
        return; // wave-uniform; no block barrier below
    float acc = 0.0f;
    for (int x0 = 0; x0 < d; x0 += 32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 8 * i + (lane >> 3), n = min(n0 + r, N - 1);
            f32x4 v = {0, 0, 0, 0};
            if (x0 + 4 * (lane & 7) < d)
                v = *(const f32x4 *)(D + (size_t)n * d + x0 + 4 * (lane & 7));
            float *dst = img + r * 33 + 4 * (lane & 7);
            dst[0] = v.x;
            dst[1] = v.y;
            dst[2] = v.z;
            dst[3] = v.w;
        }
        __builtin_amdgcn_wave_barrier();
        const float *mine = img + lane * 33;
        const int xe = min(32, d - x0);
        for (int x = 0; x < xe; ++x)
            acc = fmaf(qs[x0 + x], mine[x], acc);
        __builtin_amdgcn_wave_barrier();
    }
    if (n0 + lane < N)
        S[(size_t)b * N + n0 + lane] = acc;
}

// One launch of score_topk_kernel over docs [0,N): how the work is cut and where its
// partial lists live inside the workspace.
struct Pass {
    int n_qtiles, n_tiles, n_chunks, tiles_per_chunk, n_tasks;
    int tail_own, static_tiles, tail_g, tail_blocks; // with the pool on: tail_own tiles per chunk are static (see ScoreParams)
    int64_t N;
};

struct Plan {
    int cap;       // candidate-buffer entries per (wave, query): 64 (k <= 16) or 128
    size_t smem;   // dynamic LDS per block
    Pass main, pre;
    bool prepass;  // sample pass first: its k-th scores seed the main pass's thresholds
    // workspace layout (byte offsets)
    size_t cand_off, pval_off, pidx_off, pre_val_off, pre_idx_off, tailctr_off, ws_bytes;
    // three or more query tiles on the 32-query kernel: pacing counters and the chunks' pool draws behind the pool counter(s)
    bool paced;
    int pace_g, pace_lag;
    int grp_maxseg;
    size_t pace_off, grp_off, ctr_bytes; // ctr_bytes: pool counter(s) + pacing + draws, zeroed together before the main pass
    size_t redo_off; // one int per query tile: a wave of the tile gave up a pool draw -> the static-split pass redoes the tile
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

bool tt_score_pacing() // TT_SCORE_PACE=0: every wave for itself (measurement)
{
    return TT_AB_SWITCH(TT_SCORE_PACE, 1) != 0;
}

Pass make_pass(int B, int64_t N, int slots, int qt, bool tail = false)
{
    Pass ps;
    ps.N = N;
    ps.n_qtiles = (B + qt - 1) / qt; // qt queries per task: 32, or 16 for wide embeddings (d > 256)
    ps.n_tiles = (int)((N + TILE_DOCS - 1) / TILE_DOCS);
    int want = (slots + ps.n_qtiles - 1) / ps.n_qtiles;
    want = want < 1 ? 1 : want;
    want = want > ps.n_tiles ? ps.n_tiles : want;
    want = want < 1 ? 1 : want;
    ps.tiles_per_chunk = ps.n_tiles > 0 ? (ps.n_tiles + want - 1) / want : 1;
    ps.n_chunks = ps.n_tiles > 0 ? (ps.n_tiles + ps.tiles_per_chunk - 1) / ps.tiles_per_chunk : 1;
    ps.n_tasks = ps.n_qtiles * ps.n_chunks;
    ps.tail_own = ps.tiles_per_chunk;
    ps.static_tiles = ps.n_tiles;
    ps.tail_g = 1;
    ps.tail_blocks = 0;
    if (tail) {
        // pool = the last 1/TT_SCORE_TAIL_DIV of every chunk's share (0: everything static) in blocks of a quarter of it,
        // 4..64 tiles (a block costs one refill of the wave's ring, ~2 us)
        const int tail_div = TT_AB_SWITCH(TT_SCORE_TAIL_DIV, 8);
        const int share = tail_div > 0 ? ps.tiles_per_chunk / tail_div : 0;
        if (share >= 8) {
            const int own = ps.tiles_per_chunk - share;
            int g = share / 4;
            g = g < 4 ? 4 : (g > 64 ? 64 : g);
            ps.tail_own = own;
            ps.static_tiles = (int64_t)own * ps.n_chunks < ps.n_tiles ? own * ps.n_chunks : ps.n_tiles;
            ps.tail_g = g;
            ps.tail_blocks = (ps.n_tiles - ps.static_tiles + g - 1) / g;
        }
    }
    return ps;
}

// Prepass policy.  A wave's private list costs ~k(1+ln(n/k)) candidate appends per query for the
// n documents it sees, so cutting the corpus over ~2048 waves multiplies the selection work.  When
// chunks are short, first take the exact top-k of a 1/256 sample: its k-th score is a valid lower
// bound of the final k-th score, and with it the main pass admits only ~256k candidates per query
// in total.  Long chunks (large B) do not need it.
constexpr int64_t PREPASS_MIN_N = 262144;
constexpr int PREPASS_MAX_CHUNK_DOCS = 65536;
constexpr int64_t PREPASS_MIN_SAMPLE = 16384;

Plan make_plan(int B, int64_t N, int k, int d)
{
    Plan pl;
    const int qt = (d > 256 || B <= 16) ? 16 : 32; // 16-query tiles: wide embeddings, and batches that fit one such tile
    pl.cap = k <= 16 ? 64 : 128;
    pl.smem = (size_t)WPB * NSTAGE * SLAB_BYTES;
    const int slots = device_cus() * 8;
    pl.main = make_pass(B, N, slots, qt, true);
    pl.prepass = N >= PREPASS_MIN_N && (int64_t)pl.main.tiles_per_chunk * TILE_DOCS < PREPASS_MAX_CHUNK_DOCS;
    int max_tasks = pl.main.n_tasks;
    if (pl.prepass) {
        int64_t ns = N / 256;
        ns = ns < PREPASS_MIN_SAMPLE ? PREPASS_MIN_SAMPLE : ns;
        ns = (ns + TILE_DOCS - 1) / TILE_DOCS * TILE_DOCS;
        pl.pre = make_pass(B, ns, slots, qt);
        max_tasks = pl.pre.n_tasks > max_tasks ? pl.pre.n_tasks : max_tasks;
    } else {
        pl.pre = make_pass(B, 0, slots, qt);
    }
    const size_t rows = (size_t)pl.main.n_qtiles * qt;
    size_t max_chunks = pl.main.n_chunks;
    if (pl.prepass && (size_t)pl.pre.n_chunks > max_chunks)
        max_chunks = pl.pre.n_chunks;
    size_t off = 0;
    pl.cand_off = off;
    off = tt_align_up(off + (size_t)max_tasks * qt * pl.cap * 8, 256);
    pl.pval_off = off;
    off = tt_align_up(off + rows * max_chunks * k * sizeof(float), 256);
    pl.pidx_off = off;
    off = tt_align_up(off + rows * max_chunks * k * sizeof(int64_t), 256);
    pl.pre_val_off = off;
    off = tt_align_up(off + rows * k * sizeof(float), 256);
    pl.pre_idx_off = off;
    off = tt_align_up(off + rows * k * sizeof(int64_t), 256);
    pl.tailctr_off = off;
    off = tt_align_up(off + (size_t)pl.main.n_qtiles * sizeof(int), 256);
    pl.paced = qt == 32 && pl.main.n_qtiles >= 3; // (two query tiles: +0.7 % with it, 1.7x fetched either way)
    {
        // the chunks of one XCD (an eighth of them) share its 4 MiB L2: a chunk's waves must stay within its part of
        // ~3 MiB of each other, i.e. (pace_lag + 1) * pace_g tiles
        const int env_g = TT_AB_SWITCH(TT_SCORE_PACE_G, 0);
        const int env_lag = TT_AB_SWITCH(TT_SCORE_PACE_LAG, 0);
        const int64_t per_xcd = (pl.main.n_chunks + 7) / 8;
        const int64_t window = (3 << 20) / (per_xcd * TILE_DOCS * d * 4); // tiles
        pl.pace_lag = env_lag > 0 ? env_lag : 2;
        int g = (int)(window / (pl.pace_lag + 1));
        pl.pace_g = env_g > 0 ? env_g : (g < 1 ? 1 : (g > 8 ? 8 : g));
        if (pl.pace_lag > PACE_R - 2)
            pl.pace_lag = PACE_R - 2;
    }
    pl.grp_maxseg = 0;
    pl.pace_off = pl.grp_off = off;
    if (pl.paced) {
        off = tt_align_up(off + ((size_t)pl.main.n_chunks * PACE_R + 1) * sizeof(int), 256); // + the time-out count
        pl.grp_off = off;
        if (pl.main.tail_blocks > 0) {
            // a chunk's waves may take up to 4x their even share of the pool (>= 16 blocks); n_chunks * grp_maxseg >=
            // tail_blocks, so every block is drawn by a chunk that still may
            const int even = (pl.main.tail_blocks + pl.main.n_chunks - 1) / pl.main.n_chunks;
            pl.grp_maxseg = 4 * even > 16 ? 4 * even : 16;
            off = tt_align_up(off + (size_t)pl.main.n_chunks * pl.grp_maxseg * sizeof(int), 256);
        }
    }
    pl.ctr_bytes = off - pl.tailctr_off;
    pl.redo_off = off;
    off = tt_align_up(off + (size_t)pl.main.n_qtiles * sizeof(int), 256);
    pl.ws_bytes = off;
    return pl;
}

template <int NS, int CAP, bool MAXONLY, bool NT = false>
int launch_score_t(const ScoreParams &sp, const Plan &pl, hipStream_t st)
{
    auto kern = score_topk_kernel<NS, CAP, MAXONLY, NT>;
    const size_t smem = NT ? (size_t)WPB * NSTAGE_NT * SLAB_BYTES : pl.smem;
    TT_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int grid = (sp.n_tasks + WPB - 1) / WPB;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WPB * 64), smem, st, sp);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

template <int NS>
int launch_score_ns(const ScoreParams &sp, const Plan &pl, hipStream_t st, bool maxonly)
{
    if (maxonly)
        return launch_score_t<NS, 64, true>(sp, pl, st);
    if constexpr (NS == 8) // (d = 256 only: the instantiations are not free)
        if (sp.n_qtiles == 1)
            return pl.cap == 64 ? launch_score_t<NS, 64, false, true>(sp, pl, st) : launch_score_t<NS, 128, false, true>(sp, pl, st);
    return pl.cap == 64 ? launch_score_t<NS, 64, false>(sp, pl, st) : launch_score_t<NS, 128, false>(sp, pl, st);
}

constexpr bool score_dim_ok(int d)
{
    return d == 32 || d == 64 || d == 96 || d == 128 || d == 192 || d == 256 || d == 320 || d == 384 || d == 448 || d == 512;
}

template <int NS, int CAP, bool MAXONLY, bool NT = false>
int launch_score16_t(const ScoreParams &sp, const Plan &pl, hipStream_t st)
{
    auto kern = score_topk16_kernel<NS, CAP, MAXONLY, NT>;
    TT_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem));
    const int grid = (sp.n_tasks + WPB - 1) / WPB;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WPB * 64), pl.smem, st, sp);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

template <int NS>
int launch_score16_ns(const ScoreParams &sp, const Plan &pl, hipStream_t st, bool maxonly)
{
    if (maxonly)
        return launch_score16_t<NS, 64, true>(sp, pl, st);
    if constexpr (NS == 8)
        if (sp.n_qtiles == 1)
            return pl.cap == 64 ? launch_score16_t<NS, 64, false, true>(sp, pl, st) : launch_score16_t<NS, 128, false, true>(sp, pl, st);
    return pl.cap == 64 ? launch_score16_t<NS, 64, false>(sp, pl, st) : launch_score16_t<NS, 128, false>(sp, pl, st);
}

int launch_score(int d, const ScoreParams &sp, const Plan &pl, hipStream_t st, bool maxonly)
{
    if (d <= 256 && sp.B <= 16) { // half the MFMA work of a 32-query tile: the launch stays on the HBM roofline
        switch (d) {
        case 32: return launch_score16_ns<1>(sp, pl, st, maxonly);
        case 64: return launch_score16_ns<2>(sp, pl, st, maxonly);
        case 96: return launch_score16_ns<3>(sp, pl, st, maxonly);
        case 128: return launch_score16_ns<4>(sp, pl, st, maxonly);
        case 192: return launch_score16_ns<6>(sp, pl, st, maxonly);
        default: return launch_score16_ns<8>(sp, pl, st, maxonly);
        }
    }
    switch (d) {
    case 320: return launch_score16_ns<10>(sp, pl, st, maxonly);
    case 384: return launch_score16_ns<12>(sp, pl, st, maxonly);
    case 448: return launch_score16_ns<14>(sp, pl, st, maxonly);
    case 512: return launch_score16_ns<16>(sp, pl, st, maxonly);
    case 32: return launch_score_ns<1>(sp, pl, st, maxonly);
    case 64: return launch_score_ns<2>(sp, pl, st, maxonly);
    case 96: return launch_score_ns<3>(sp, pl, st, maxonly);
    case 128: return launch_score_ns<4>(sp, pl, st, maxonly);
    case 192: return launch_score_ns<6>(sp, pl, st, maxonly);
    default: return launch_score_ns<8>(sp, pl, st, maxonly);
    }
}

ScoreParams pass_params(const Pass &ps, const float *Q, int B, const float *D, int k, int64_t idx_offset, char *ws,
                        const Plan &pl)
{
    ScoreParams sp;
    sp.Q = Q;
    sp.D = D;
    sp.B = B;
    sp.N = (int)ps.N;
    sp.k = k;
    sp.n_qtiles = ps.n_qtiles;
    sp.n_chunks = ps.n_chunks;
    sp.tiles_per_chunk = ps.tiles_per_chunk;
    sp.n_tiles = ps.n_tiles;
    sp.n_tasks = ps.n_tasks;
    sp.static_tiles = ps.n_tiles; // (the caller turns the pool on for the main pass)
    sp.tail_g = 1;
    sp.tail_blocks = 0;
    sp.tail_ctr = nullptr;
    sp.pace = nullptr;
    sp.pace_timeouts = nullptr;
    sp.pace_g = sp.pace_lag = 1;
    sp.grp_blk = nullptr;
    sp.grp_maxseg = 0;
    sp.pval = (float *)(ws + pl.pval_off);
    sp.pidx = (int64_t *)(ws + pl.pidx_off);
    sp.idx_offset = idx_offset;
    sp.cand = (Cand *)(ws + pl.cand_off);
    sp.thr0 = nullptr;
    sp.thr0_stride = 0;
    sp.thr0_off = 0;
    sp.run_if = nullptr;
    sp.draw_polls = TT_AB_SWITCH(TT_DRAW_POLLS, DRAW_POLLS);
    return sp;
}

int score_partials(const float *Q, int B, int d, const float *D, int64_t N, int k, int64_t idx_offset,
                   void *workspace, size_t workspace_bytes, hipStream_t st, Plan *plan_out, const char *who,
                   const int *run_if = nullptr, void *const *prof_events = nullptr)
{
    if (B <= 0 || N <= 0 || k <= 0)
        return tt_fail(TT_ERR_BAD_SHAPE, "%s: B=%d N=%lld k=%d", who, B, (long long)N, k);
    if (!score_dim_ok(d))
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: d=%d (supported: 32, 64, 96, 128, 192, 256, 320, 384, 448, 512)", who, d);
    if (k > 64)
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: k=%d > 64", who, k);
    if (N >= (int64_t)INT_MAX - 64)
        return tt_fail(TT_ERR_UNSUPPORTED, "%s: N=%lld >= 2^31-64; shard the corpus", who, (long long)N);
    if (!Q || !D)
        return tt_fail(TT_ERR_BAD_SHAPE, "%s: null pointer", who);
    const Plan pl = make_plan(B, N, k, d);
    if (!workspace || workspace_bytes < pl.ws_bytes)
        return tt_fail(TT_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", who, workspace_bytes, pl.ws_bytes);
    if (((uintptr_t)D & 15) || ((uintptr_t)Q & 3) || ((uintptr_t)workspace & 255))
        return tt_fail(TT_ERR_BAD_SHAPE, "%s: D must be 16-byte and the workspace 256-byte aligned", who);
    char *ws = (char *)workspace;
    *plan_out = pl;
    const float *thr0 = nullptr;
    // (the predicated form -- the screened search's on-device fallback, a no-op unless a flag is raised -- skips
    //  the sample pass: it only seeds thresholds, and two fewer empty launches sit behind every screened search)
    if (pl.prepass && !run_if) {
        // sample pass over D[0:ns): per-(wave,query) maxima, then their k-th largest per query.
        // k distinct documents score at least that much, so it bounds the final k-th score from below.
        ScoreParams pp = pass_params(pl.pre, Q, B, D, k, 0, ws, pl);
        pp.run_if = run_if;
        int rc = launch_score(d, pp, pl, st, true);
        if (rc != TT_OK)
            return rc;
        hipLaunchKernelGGL(kth_largest_kernel, dim3(B), dim3(256), 0, st, (const float *)pp.pval, pl.pre.n_chunks, k,
                           (float *)(ws + pl.pre_val_off), run_if, (float *)nullptr);
        TT_LAUNCH_CHECK();
        thr0 = (const float *)(ws + pl.pre_val_off);
    }
    ScoreParams sp = pass_params(pl.main, Q, B, D, k, idx_offset, ws, pl);
    sp.thr0 = thr0;
    sp.thr0_stride = 1;
    sp.thr0_off = 0;
    sp.run_if = run_if;
    if (pl.main.tail_blocks > 0 && !run_if) {
        // (the predicated form keeps the static split: its launches sit behind every screened search as no-ops, and a
        //  counter reset would be one more)
        sp.tiles_per_chunk = pl.main.tail_own;
        sp.static_tiles = pl.main.static_tiles;
        sp.tail_g = pl.main.tail_g;
        sp.tail_blocks = pl.main.tail_blocks;
        sp.tail_ctr = (int *)(ws + pl.tailctr_off);
    }
    if (pl.paced && !run_if && tt_score_pacing()) {
        sp.pace = (int *)(ws + pl.pace_off);
        sp.pace_timeouts = sp.pace + (size_t)pl.main.n_chunks * PACE_R;
        sp.pace_g = pl.pace_g;
        sp.pace_lag = pl.pace_lag;
        if (sp.tail_ctr) {
            sp.grp_blk = (int *)(ws + pl.grp_off);
            sp.grp_maxseg = pl.grp_maxseg;
        }
    }
    if (sp.tail_ctr || sp.pace)
        TT_RC_CHECK(tt_zero_async(ws + pl.tailctr_off, pl.ctr_bytes, st));
    if (prof_events)
        TT_HIP_CHECK(hipEventRecord((hipEvent_t)prof_events[0], st));
    const int rc = launch_score(d, sp, pl, st, false);
    if (prof_events && rc == TT_OK)
        TT_HIP_CHECK(hipEventRecord((hipEvent_t)prof_events[1], st));
    return rc;
}

} // namespace

// Byte offset, in the workspace of a finished tt_score_topk(_partials)_f32 call of this shape, of an int32 that counts the waves
// whose pacing wait ran into its bound (0 when the launch was not paced): (size_t)-1 if the shape is never paced.
TT_EXPORT size_t tt_score_topk_pace_timeouts_offset(int B, int64_t N, int d, int k)
{
    if (B <= 0 || N <= 0 || k <= 0)
        return (size_t)-1;
    const Plan pl = make_plan(B, N, k, d);
    return pl.paced ? pl.pace_off + (size_t)pl.main.n_chunks * PACE_R * sizeof(int) : (size_t)-1;
}

// Diagnostic: byte offset of the per-query-tile flags (int32 each) that say which tiles the last tt_score_topk_f32 call of this
// shape did again because a wave had given up a pool draw; (size_t)-1 = the shape never draws from a shared pool.
TT_EXPORT size_t tt_score_topk_redo_flags_offset(int B, int64_t N, int d, int k)
{
    if (B <= 0 || N <= 0 || k <= 0)
        return (size_t)-1;
    const Plan pl = make_plan(B, N, k, d);
    return (pl.paced && pl.main.tail_blocks > 0) ? pl.redo_off : (size_t)-1;
}

TT_EXPORT size_t tt_score_topk_workspace_bytes(int B, int64_t N, int d, int k)
{
    if (B <= 0 || N < 0 || k <= 0)
        return 0;
    return make_plan(B, N, k, d).ws_bytes;
}

TT_EXPORT int tt_score_topk_partials_f32(const float *Q, int B, int d, const float *D, int64_t N, int k,
                                         int64_t idx_offset, void *workspace, size_t workspace_bytes,
                                         const float **part_val, const int64_t **part_idx, int *part_m,
                                         void *const *prof_events, tt_stream_t stream)
{
    Plan pl;
    int rc = score_partials(Q, B, d, D, N, k, idx_offset, workspace, workspace_bytes, (hipStream_t)stream, &pl,
                            "tt_score_topk_partials_f32", nullptr, prof_events);
    if (rc != TT_OK)
        return rc;
    if (part_val)
        *part_val = (const float *)((const char *)workspace + pl.pval_off);
    if (part_idx)
        *part_idx = (const int64_t *)((const char *)workspace + pl.pidx_off);
    if (part_m)
        *part_m = pl.main.n_chunks * k;
    return TT_OK;
}

namespace {
// flags[t] = 1 when a query of 32-query tile t came out of the merge with the give-up marker in its first place
__global__ __launch_bounds__(64) void redo_flag_kernel(const int64_t *__restrict__ out_idx, int B, int k, int qt, int *__restrict__ flags)
{
    const int t = blockIdx.x, q = t * qt + threadIdx.x;
    const bool bad = threadIdx.x < qt && q < B && out_idx[(size_t)q * k] >= (int64_t)TT_TOPK_INVALID_INDEX;
    const unsigned long long any = __ballot(bad);
    if (threadIdx.x == 0)
        flags[t] = any != 0ull;
}
} // namespace

// Exact path, optionally predicated on a device flag (the screened path's fallback).
int tt_score_topk_f32_pred(const float *Q, int B, int d, const float *D, int64_t N, int k, int64_t idx_offset,
                           float *out_val, int64_t *out_idx, void *workspace, size_t workspace_bytes,
                           const int *run_if, hipStream_t st);

TT_EXPORT int tt_score_topk_f32(const float *Q, int B, int d, const float *D, int64_t N, int k,
                                int64_t idx_offset, float *out_val, int64_t *out_idx, void *workspace,
                                size_t workspace_bytes, tt_stream_t stream)
{
    return tt_score_topk_f32_pred(Q, B, d, D, N, k, idx_offset, out_val, out_idx, workspace, workspace_bytes, nullptr,
                                  (hipStream_t)stream);
}

int tt_score_topk_f32_pred(const float *Q, int B, int d, const float *D, int64_t N, int k, int64_t idx_offset,
                           float *out_val, int64_t *out_idx, void *workspace, size_t workspace_bytes,
                           const int *run_if, hipStream_t st)
{
    if (B < 0 || N < 0 || k <= 0)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_score_topk_f32: B=%d N=%lld k=%d", B, (long long)N, k);
    if (B == 0)
        return TT_OK;
    if (!score_dim_ok(d))
        return tt_fail(TT_ERR_UNSUPPORTED, "tt_score_topk_f32: d=%d (supported: 32, 64, 96, 128, 192, 256, 320, 384, 448, 512)", d);
    if (k > 64)
        return tt_fail(TT_ERR_UNSUPPORTED, "tt_score_topk_f32: k=%d > 64", k);
    if (!out_val || !out_idx)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_score_topk_f32: null output pointer");
    if (N == 0) { // merge over zero candidates writes the (-inf,-1) tail
        hipLaunchKernelGGL(topk_merge_kernel, dim3(B), dim3(MERGE_THREADS), 0, st, (const float *)nullptr,
                           (const int64_t *)nullptr, 0, k, out_val, out_idx, (const int *)nullptr, 1, (size_t)0);
        TT_LAUNCH_CHECK();
        return TT_OK;
    }
    Plan pl;
    int rc = score_partials(Q, B, d, D, N, k, idx_offset, workspace, workspace_bytes, st, &pl, "tt_score_topk_f32", run_if);
    if (rc != TT_OK)
        return rc;
    const char *ws = (const char *)workspace;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(B), dim3(MERGE_THREADS), 0, st, (const float *)(ws + pl.pval_off),
                       (const int64_t *)(ws + pl.pidx_off), pl.main.n_chunks * k, k, out_val, out_idx, run_if,
                       pl.main.n_chunks * k, (size_t)0);
    TT_LAUNCH_CHECK();
    if (!run_if && pl.paced && pl.main.tail_blocks > 0 && tt_score_pacing()) {
        // The one wait of this path that cannot be skipped without losing documents is a wave's wait for a chunk-mate's pool
        // draw; a wave whose budget ran out marked its lists (+inf, TT_TOPK_INVALID_INDEX + t).  Nothing downstream reads that
        // marker, so it is dealt with HERE, on the device: the query tiles whose merged list starts with it are done again on
        // the static split (the predicated form: no pool, no pacing, nobody to wait for) -- three small launches that find
        // nothing to do in every run observed so far (~10 us behind a search of >= 4 ms).
        int *redo = (int *)((char *)workspace + pl.redo_off);
        const int qt = d > 256 ? 16 : 32;
        hipLaunchKernelGGL(redo_flag_kernel, dim3(pl.main.n_qtiles), dim3(64), 0, st, (const int64_t *)out_idx, B, k, qt, redo);
        TT_LAUNCH_CHECK();
        return tt_score_topk_f32_pred(Q, B, d, D, N, k, idx_offset, out_val, out_idx, workspace, workspace_bytes, redo, st);
    }
    return TT_OK;
}

// k-th largest of each row of vals [B][M] -> out [B] (internal: threshold seeding of both search paths)
int tt_kth_largest(const float *vals, int B, int M, int k, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(kth_largest_kernel, dim3(B), dim3(256), 0, st, vals, M, k, out, (const int *)nullptr, (float *)nullptr);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

// the k largest of each row of vals [B][M] -> list [B][k], unordered (-inf padding when M < k)
int tt_k_largest_list(const float *vals, int B, int M, int k, float *list, hipStream_t st)
{
    hipLaunchKernelGGL(kth_largest_kernel, dim3(B), dim3(256), 0, st, vals, M, k, (float *)nullptr, (const int *)nullptr, list);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

namespace {
// Union seed of a row-sharded search: lists [world][B][ks] (every shard's ks largest sample maxima per query, as the
// all-gather left them) -> seed[q] = the kth-th largest of the world * ks values of query q.  Those are approximate scores
// of world * ks DISTINCT documents (one per 32-document sample tile, the shards' rows are disjoint), so kth documents of the
// whole corpus score at least seed[q]: a valid lower bound of the global kth-th best approximate score, and a much
// tighter one than any single shard's kth-th sample maximum.  (ks < kth on wide jobs: 16 ranks x k = 64 would be 1024 values;
// every rank then lists its 32 largest and the 64th of the 512 is taken -- still 64 distinct documents above the seed.)
// One wave per query, rank by counting (world * ks <= 512).
__global__ __launch_bounds__(64) void seed_union_kernel(const float *__restrict__ lists, int world, int B, int ks, int kth,
                                                        float *__restrict__ seed)
{
    const int q = blockIdx.x, lane = threadIdx.x;
    const int M = world * ks;
    __shared__ float v[512];
    for (int i = lane; i < M; i += 64)
        v[i] = lists[((size_t)(i / ks) * B + q) * ks + (i % ks)];
    __syncthreads();
    for (int i = lane; i < M; i += 64) {
        const float x = v[i];
        int before = 0; // entries ranking before entry i in (value desc, position asc) order
        for (int j = 0; j < M; ++j)
            before += (v[j] > x) || (v[j] == x && j < i);
        if (before == kth - 1)
            seed[q] = x; // exactly one entry has this rank (NaNs do not occur: the lists hold MFMA sums of finite inputs)
    }
}
} // namespace

TT_EXPORT int tt_seed_union_f32(const float *lists, int world, int B, int list_len, int kth, float *seed, tt_stream_t stream)
{
    if (!lists || !seed || world < 1 || B <= 0 || list_len < 1 || (int64_t)world * list_len > 512 || kth < 1 ||
        kth > world * list_len)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_seed_union_f32: world=%d B=%d list_len=%d kth=%d (kth <= world * list_len <= 512)",
                       world, B, list_len, kth);
    hipLaunchKernelGGL(seed_union_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, lists, world, B, list_len, kth, seed);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

TT_EXPORT int tt_topk_merge(const float *in_val, const int64_t *in_idx, int B, int M, int k, float *out_val,
                            int64_t *out_idx, tt_stream_t stream)
{
    if (B < 0 || M < 0 || k <= 0)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_topk_merge: B=%d M=%d k=%d", B, M, k);
    if (k > MERGE_KMAX)
        return tt_fail(TT_ERR_UNSUPPORTED, "tt_topk_merge: k=%d > %d", k, MERGE_KMAX);
    if (B == 0)
        return TT_OK;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(B), dim3(MERGE_THREADS), 0, (hipStream_t)stream, in_val, in_idx, M,
                       k, out_val, out_idx, (const int *)nullptr, M > 0 ? M : 1, (size_t)0);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

TT_EXPORT int tt_topk_merge_shards(const void *gathered, int world, size_t rank_stride, size_t idx_byte_offset, int B,
                                   int kp, int k, float *out_val, int64_t *out_idx, tt_stream_t stream)
{
    if (B < 0 || world <= 0 || kp <= 0 || k <= 0)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_topk_merge_shards: world=%d B=%d kp=%d k=%d", world, B, kp, k);
    if (k > MERGE_KMAX)
        return tt_fail(TT_ERR_UNSUPPORTED, "tt_topk_merge_shards: k=%d > %d", k, MERGE_KMAX);
    if ((int64_t)world * kp > INT_MAX)
        return tt_fail(TT_ERR_UNSUPPORTED, "tt_topk_merge_shards: world*kp too large");
    if (!gathered || !out_val || !out_idx || ((uintptr_t)gathered & 7) || (rank_stride & 7) || (idx_byte_offset & 7) ||
        idx_byte_offset < (size_t)B * kp * sizeof(float) || rank_stride < idx_byte_offset + (size_t)B * kp * sizeof(int64_t))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_topk_merge_shards: layout (stride %zu, idx offset %zu) does not hold [B,kp] f32 + i64, 8-byte aligned",
                       rank_stride, idx_byte_offset);
    if (B == 0)
        return TT_OK;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(B), dim3(MERGE_THREADS), 0, (hipStream_t)stream, (const float *)gathered,
                       (const int64_t *)((const char *)gathered + idx_byte_offset), world * kp, k, out_val, out_idx,
                       (const int *)nullptr, kp, rank_stride);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

TT_EXPORT int tt_score_all_f32(const float *Q, int B, int d, const float *D, int64_t N, float *S, tt_stream_t stream)
{
    if (B < 0 || N <= 0 || d <= 0 || (d & 3))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_score_all_f32: B=%d N=%lld d=%d", B, (long long)N, d);
    if (N >= INT_MAX || d > 512 || B > 65535)
        return tt_fail(TT_ERR_UNSUPPORTED, "tt_score_all_f32: N, d (<= 512) or B (<= 65535) too large");
    if (B == 0)
        return TT_OK;
    if (!Q || !D || !S)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_score_all_f32: null pointer");
    hipLaunchKernelGGL(score_all_kernel, dim3((unsigned)((N + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, Q, D, (int)N, d, S);
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
    if (d > 512)
        return tt_fail(TT_ERR_UNSUPPORTED, "tt_score_rank_f32: d=%d > 512", d);
    if (B == 0)
        return TT_OK;
    hipLaunchKernelGGL(score_rank_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, Q, D, (int)N, d, target, rank);
    TT_LAUNCH_CHECK();
    return TT_OK;
}
