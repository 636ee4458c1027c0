// K7w: the weight-gradient products  dW[M][N] = sum over tokens of A[tok][M]^T B[tok][N]  (dW_ih = dGi^T X with X rows gathered
// from the embedding table, dW_hh = dGh^T H_prev with H_prev rows through the previous-token map; backend/main.py:254's
// loss.backward() for nn.GRU's weight_ih / weight_hh) on the f16 matrix pipes, fp16 hi/lo split of both operands as in sgemm.hip.
//
// The tiled kernel (sgemm16_kernel<true,true>: 128 x 128 output tiles, 64 split-K slabs) reads dGi three times and X six times
// (1.2 GB per product for the bench's 75 k tokens), stages both operands TRANSPOSED through registers (2-way conflicted
// ds_write_b32 pairs) and runs each workgroup as a chain of barrier-separated tiles with one tile of loads in flight: 263 +
// 153 us for the two products of the document tower, 27 % of the train step once the recurrences were split over four CUs.  Here
//   * a workgroup (8 waves, one per CU) owns a 256 x NW output tile (NW = 160 or 128: the whole N = 300 / 256 in two column
//     tiles) for one K slab: 80 or 64 accumulator registers per lane (4 x 5 / 4 x 4 tiles of 16 x 16 per wave), the grid is
//     3 x 2 tiles x ~42 slabs = one workgroup per CU; dGi is read twice and X three times (0.5 GB);
//   * both operands are token-major in memory -- the reduction index is the SLOW axis -- so they go into LDS exactly as they
//     arrive (a thread converts one float4 into 4 hi + 4 lo halves and writes two 8-byte pieces: no transposing writes) and are
//     read as MFMA operands with ds_read_b64_tr_b16, gfx950's transposing LDS read: a 16-lane group fetches a 4 (k) x 16 (m or n)
//     block and every lane gets ITS column's four k values, two reads per 16 x 16 x 32 operand;
//   * rows are 544 bytes apart (136 dwords = 8 banks mod 64: the four rows of a block fall into four different 8-bank windows)
//     and the 16-byte chunks of rows 8 .. 15, 24 .. 31 are XORed with 8 (128 bytes = 32 banks): the two 16-lane groups of a
//     32-lane half read rows 8 apart and land on different halves of the bank array;
//   * two LDS stages (2 x 4 images x 32 rows x 544 B = 136 KB) and two register sets of operand rows: tile t is multiplied
//     from LDS while tile t + 1 is committed and the loads of tile t + 2 are in flight; one barrier per tile.
// Measured (75 k tokens, one box): dW_ih 263 -> 180 us, dW_hh 153 -> 140 us, train step 1.52 -> 1.40 ms.  Ablation builds
// (phases switched off at run time, tools/experiments/wgrad_ablation.sh at the commit that carried the switches; profiler
// timings): loop + LDS reads + barriers alone 35 us; + MFMAs 63; + commit (of stale registers) 59; + loads (never waited for)
// 63; MFMAs + commit 84; MFMAs + loads 86; commit + LOADED data 184; everything 214: what costs is WAITING for the operand rows
// -- 2.3 us per 52 KB tile and CU whether one or two tiles are in flight -- although HBM delivers only the unique 320 MB
// (TCC misses x 128 B; L2 hit rate 64 %: the XCD placement works) = 1.8 TB/s.  Open: where those loads queue (the gathered
// table rows are the suspects: the dW_hh launch, whose B rows are the packed hidden sequence, halves without its MFMAs, the
// dW_ih launch does not).
// Products and their order per element: (A hi)(B hi) + (A lo)(B hi) + (A hi)(B lo), token tiles ascending inside a slab, slabs
// reduced in slab order by tt_slab_reduce: deterministic, fp32-grade; NOT bit-identical to the tiled kernel (another slab
// partition).  TT_WGRAD_TILED=1 keeps the tiled kernel (A/B).
#include "sgemm.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));

constexpr int WG_KT = 32;            // tokens per tile = one MFMA K
constexpr int WG_MT = 256;           // output rows per workgroup
constexpr int WG_ROW = 544;          // bytes per image row
constexpr int WG_IMG = WG_KT * WG_ROW;  // one image (hi or lo of one operand)
constexpr int WG_STAGE = 4 * WG_IMG;    // A hi, A lo, B hi, B lo
constexpr int WG_LDS = 2 * WG_STAGE;    // 139 264 B

// byte offset of 16-byte chunk `ch` (+ sub = 0 / 8) of row `row` in an image
__device__ __forceinline__ int img_off(int row, int ch, int sub) { return row * WG_ROW + ((ch ^ (((row >> 3) & 1) << 3)) << 4) + sub; }

__device__ __forceinline__ h8 tr_frag(const char *p0, const char *p1)
{
    struct Pair {
        s4v a, b;
    } t;
    t.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v *)p0);
    t.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v *)p1);
    return __builtin_bit_cast(h8, t);
}

// NT: 16-column tiles per wave along N (the workgroup's tile is 2 NT x 16 columns wide)
template <int NT>
__global__ __launch_bounds__(512, 1) void wgrad16_kernel(SgemmParams p, int n_ntiles, int kchunk, int nslabs)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int NW = 2 * NT * 16;     // columns per workgroup
    constexpr int BQ = NW / 4;          // float4 per B row
    constexpr int BJ = (WG_KT * BQ + 511) / 512; // B float4 per thread and tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w & 3, wn = w >> 2;
    // The tiles of one K slab read the same rows of both operands (dGi: once per column tile, X: once per row tile): they are
    // put on ONE XCD so that its L2 serves the re-reads -- blocks are dealt round-robin over the 8 XCDs, so block L lands on XCD
    // L % 8 (a speed bonus, never relied on): slab z = 8 (L / (8 tiles)) + L % 8, tile = (L / 8) % tiles.  Spread over the XCDs
    // in launch order the same kernel moved 730 MB instead of 320 and ran at the speed of the fabric (4.1 TB/s).
    const int n_tiles = (p.M / WG_MT) * n_ntiles;
    const int L = blockIdx.x, tile = (L >> 3) % n_tiles, slab = 8 * ((L >> 3) / n_tiles) + (L & 7);
    if (slab >= nslabs)
        return;
    const int mtile = tile / n_ntiles, ntile = tile % n_ntiles;
    const int m0 = mtile * WG_MT, n0 = ntile * NW;
    const int K = p.k_dyn ? min(p.K, *p.k_dyn) : p.K;
    const int kb = slab * kchunk, ke = min(kb + kchunk, K);
    const int ea = p.a_absmax ? tt_pow2_exponent(*p.a_absmax) : p.a_exp;
    const int eb = p.b_absmax ? tt_pow2_exponent(*p.b_absmax) : p.b_exp;
    const float sa = ldexpf(1.0f, ea), sb = ldexpf(1.0f, eb), down = ldexpf(1.0f, -(ea + eb));

    f32x4v acc[4][NT];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
            acc[a][b] = (f32x4v){0, 0, 0, 0};

    // ---- fill mapping.  A: thread -> rows (tid >> 6) + 8 j, float4 column tid & 63 (a wave reads 1 KB of one token row).
    //      B: float4 index tid + 512 j of the 32 x BQ tile, row = idx / BQ (source row through b_map), column idx % BQ.
    //      Every load is UNCONDITIONAL from a clamped, valid address (what lies past the slab's end or past column N is
    //      zeroed by a select when it is committed) and the source-row indices of a tile are fetched one tile before its
    //      data: a row index loaded right in front of its row would put a wait between any two data loads -- seven
    //      serialised round trips per tile, which is what the first build of this kernel spent its time on ----
    const int klast = K > 0 ? K - 1 : 0;
    int ia[4], ib[BJ];
    f32x4v ra0[4], rb0[BJ], ra1[4], rb1[BJ]; // two tiles of operand rows in flight (registers), two tiles staged (LDS)
    auto fetch_idx = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tok = min(k0 + (tid >> 6) + 8 * j, klast);
            ia[j] = tok;
        }
        if (p.a_map) { // (uniform)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                ia[j] = p.a_map[ia[j]];
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int idx = min(tid + 512 * j, WG_KT * BQ - 1);
            ib[j] = min(k0 + idx / BQ, klast);
        }
        if (p.b_map) {
#pragma unroll
            for (int j = 0; j < BJ; ++j)
                ib[j] = p.b_map[ib[j]];
        }
    };
    const int acol = m0 + 4 * (tid & 63);
    auto fetch = [&](f32x4v (&ra)[4], f32x4v (&rb)[BJ]) { // the tile whose indices are in ia / ib
#pragma unroll
        for (int j = 0; j < 4; ++j)
            ra[j] = *(const f32x4v *)(p.A + (int64_t)ia[j] * p.lda + acol);
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int idx = min(tid + 512 * j, WG_KT * BQ - 1);
            const int col = min(n0 + 4 * (idx % BQ), p.N - 4);
            rb[j] = *(const f32x4v *)(p.B + (int64_t)ib[j] * p.ldb + col);
        }
    };
    // one float4 -> 4 hi + 4 lo halves; rows past the slab's end / columns past N are zeroed through the scale (the clamped
    // addresses hold finite numbers: valid tokens, valid columns)
    auto split4 = [&](f32x4v v, float s, h4 &hi, h4 &lo) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = v[e] * s;
            const _Float16 hv = (_Float16)x;
            hi[e] = hv;
            lo[e] = (_Float16)(x - (float)hv);
        }
    };
    auto commit_a = [&](char *stage, int k0, int j, f32x4v v) { // A piece j of the tile that starts at token k0
        h4 hi, lo;
        const int r = (tid >> 6) + 8 * j;
        split4(v, k0 + r < ke ? sa : 0.0f, hi, lo);
        const int off = img_off(r, (tid & 63) >> 1, (tid & 1) * 8);
        *(h4 *)(stage + off) = hi;
        *(h4 *)(stage + WG_IMG + off) = lo;
    };
    auto commit_b = [&](char *stage, int k0, int j, f32x4v v) {
        const int idx = tid + 512 * j, r = idx / BQ, c = idx % BQ;
        if (idx < WG_KT * BQ) {
            h4 hi, lo;
            split4(v, (k0 + r < ke && n0 + 4 * c < p.N) ? sb : 0.0f, hi, lo);
            const int off = img_off(r, c >> 1, (c & 1) * 8);
            *(h4 *)(stage + 2 * WG_IMG + off) = hi;
            *(h4 *)(stage + 3 * WG_IMG + off) = lo;
        }
    };

    // ---- operand read addresses: lane (g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3) of a 16-lane group supplies row
    //      8 g + q (+ 4 for the second read), halves 4 pp .. + 3 of the tile's 16 columns ----
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int rrow = 8 * g + q; // (rrow >> 3) & 1 == g & 1 for both reads
    int a_off[4], b_off[NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
        a_off[mt] = img_off(rrow, (64 * wm + 16 * mt) / 8 + (pp >> 1), (pp & 1) * 8);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        b_off[nt] = img_off(rrow, (NT * 16 * wn + 16 * nt) / 8 + (pp >> 1), (pp & 1) * 8);

    // Tile t is multiplied from LDS stage t & 1 while the rows of tile t + 2 are in flight to one register set and tile
    // t + 1 is committed from the other (loaded during the previous iteration).  The commit is spread over the four
    // MFMA groups of the iteration and pinned there (sched_group_barrier: one MFMA, then a few conversion instructions):
    // with the commit BEHIND the MFMAs both waves of a SIMD converted while the matrix pipe idled and multiplied while the
    // VALU idled -- 3.1 us per tile for 0.9 us of MFMAs.  The body is branch-free: loads, index loads and commits past the
    // slab's end go to clamped addresses / the unused stage.
    int cur = 0;
    auto iter = [&](int k0, f32x4v (&la)[4], f32x4v (&lb)[BJ], const f32x4v (&ca)[4], const f32x4v (&cb)[BJ]) {
        fetch(la, lb);              // tile k0 + 64 (its indices arrived during the previous iteration)
        fetch_idx(k0 + 3 * WG_KT);
        const char *st = lds + cur * WG_STAGE;
        char *nx = lds + (cur ^ 1) * WG_STAGE;
        h8 bh[NT], bl[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bh[nt] = tr_frag(st + 2 * WG_IMG + b_off[nt], st + 2 * WG_IMG + b_off[nt] + 4 * WG_ROW);
            bl[nt] = tr_frag(st + 3 * WG_IMG + b_off[nt], st + 3 * WG_IMG + b_off[nt] + 4 * WG_ROW);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const h8 ah = tr_frag(st + a_off[mt], st + a_off[mt] + 4 * WG_ROW);
            const h8 al = tr_frag(st + WG_IMG + a_off[mt], st + WG_IMG + a_off[mt] + 4 * WG_ROW);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[nt], acc[mt][nt], 0, 0, 0);
#if !(TT_MUTATE_DROP_LO & 8)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[nt], acc[mt][nt], 0, 0, 0);
#endif
            commit_a(nx, k0 + WG_KT, mt, ca[mt]);
            if (mt < BJ)
                commit_b(nx, k0 + WG_KT, mt, cb[mt]);
#pragma unroll
            for (int i = 0; i < 3 * NT; ++i) { // this group's 3 NT MFMAs, each followed by a slice of the conversion work
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); // VALU
            }
        }
        __syncthreads();
        cur ^= 1;
    };
    if (kb < ke) {
        fetch_idx(kb);
        fetch(ra0, rb0);
        fetch_idx(kb + WG_KT);
        fetch(ra1, rb1);
        fetch_idx(kb + 2 * WG_KT);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            commit_a(lds, kb, j, ra0[j]);
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            commit_b(lds, kb, j, rb0[j]);
    }
    __syncthreads();
    for (int k0 = kb; k0 < ke; k0 += 2 * WG_KT) {
        iter(k0, ra0, rb0, ra1, rb1);
        if (k0 + WG_KT < ke)
            iter(k0 + WG_KT, ra1, rb1, ra0, rb0);
    }
    // ---- this slab's tile: rows m0 + 64 wm + 16 mt + 4 g + e, columns n0 + NT 16 wn + 16 nt + (lane & 15) ----
    float *C = p.C + (size_t)slab * p.slab_stride;
    const int jn = lane & 15;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + NT * 16 * wn + 16 * nt + jn;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int mrow = m0 + 64 * wm + 16 * mt + 4 * g + e;
                if (n < p.N && mrow < p.M)
                    C[(size_t)mrow * p.ldc + n] = acc[mt][nt][e] * down;
            }
        }
}

bool wgrad_tiled()
{
    static const bool v = [] { const char *e = getenv("TT_WGRAD_TILED"); return e && atoi(e) != 0; }();
    return v;
}

} // namespace

// shapes this kernel takes: output rows a multiple of 256, N a multiple of 4, operand rows 16-byte aligned
bool tt_wgrad16_supported(int M, int N, int64_t lda, int64_t ldb)
{
    return !wgrad_tiled() && M > 0 && M % WG_MT == 0 && N >= 64 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0;
}

// how many K slabs the launch will write (the caller reduces that many): one workgroup per CU, at most max_slabs
int tt_wgrad16_slabs(int M, int N, int cus, int max_slabs)
{
    const int NW = (N > 256 && N <= 320) ? 160 : 128;
    const int tiles = (M / WG_MT) * ((N + NW - 1) / NW);
    int s = cus / (tiles > 0 ? tiles : 1);
    s = s < 1 ? 1 : s;
    return s > max_slabs ? max_slabs : s;
}

// p as for tt_sgemm16's A^T B^T form: A [K][M] (lda), B [K][N] rows through b_map (ldb), K tokens (k_dyn), C = slabs [z][M][N]
// (ldc = N, slab_stride = M N), a_absmax / a_exp, b_absmax / b_exp.
int tt_wgrad16(const SgemmParams &p, int nslabs, hipStream_t st)
{
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || nslabs < 1)
        return TT_OK;
    if (!tt_wgrad16_supported(p.M, p.N, p.lda, p.ldb) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15))
        return TT_ERR_UNSUPPORTED;
    const bool wide = p.N > 256 && p.N <= 320;
    const int NW = wide ? 160 : 128;
    const int n_ntiles = (p.N + NW - 1) / NW;
    int kchunk = (p.K + nslabs - 1) / nslabs;
    kchunk = (kchunk + WG_KT - 1) / WG_KT * WG_KT;
    const dim3 grid((unsigned)((p.M / WG_MT) * n_ntiles * ((nslabs + 7) / 8 * 8)));
    static bool attr_done = false;
    if (!attr_done) {
        TT_HIP_CHECK(hipFuncSetAttribute((const void *)wgrad16_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS));
        TT_HIP_CHECK(hipFuncSetAttribute((const void *)wgrad16_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS));
        attr_done = true;
    }
    if (wide)
        hipLaunchKernelGGL(wgrad16_kernel<5>, grid, dim3(512), WG_LDS, st, p, n_ntiles, kchunk, nslabs);
    else
        hipLaunchKernelGGL(wgrad16_kernel<4>, grid, dim3(512), WG_LDS, st, p, n_ntiles, kchunk, nslabs);
    TT_LAUNCH_CHECK();
    return TT_OK;
}
