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
//   * two LDS stages (2 x 4 images x 32 rows x 544 B = 136 KB) and ONE register set of operand rows: tile t is multiplied
//     from LDS while tile t + 1 is converted piece by piece and every piece's registers are requested again at once for tile
//     t + 2 (a request has a whole iteration to arrive); one barrier per tile.
// What bounds it (round 3, measured in this order -- the first explanation, "waiting for the operand rows", was wrong):
//   * s_memtime stamps inside the loop: the wait for loaded rows is 10 - 50 clocks per tile; the time went into ISSUING -- 1370
//     clocks per tile for 14 load instructions per wave when all eight waves queue at the CU's one texture-address path
//     together -- and into the conversion: 1900 - 2300 clocks per tile for ~120 vector instructions;
//   * tools/experiments/valu_rate.hip: a lone wave issues a vector instruction every 8.7 clocks (v_fma_mix: 12.5), and next to
//     a wave that keeps the matrix pipe busy the SIMD's other wave issues ONE vector instruction per MFMA (16.3 clocks; the
//     packed fp32 forms 25).  A tile's 120 MFMAs per SIMD therefore hide 120 vector instructions and every further one costs
//     4 - 8 clocks of an idle matrix pipe; the first build issued 420 per SIMD and tile (64-bit row-address arithmetic with
//     quarter-rate multiplies, a bound check per float4, 17 instructions of conversion per float4);
//   * so: scalar row addresses for A (rows are wave-uniform), 24-bit multiplies and 32-bit offsets for B, the past-the-slab zero
//     on A only (scalar), the conversion as eight v_fma_mix per float4 (inline asm: hi = f16(s v), lo = f16(s v - hi), both one
//     instruction) -- 80 - 90 per wave and tile -- and the fill cut into 5 + BJ pieces spread over the twelve MFMA runs of a
//     tile (sched_group_barrier inside a run, sched_barrier between runs; one basic block: no map / tail branches);
//   * tried and dropped: the two waves of a SIMD taking "fill" and "multiply" in opposite order (the filling wave then runs
//     at one instruction per 16 clocks for the whole multiply of its partner: 3840 clocks per tile against 3400 interleaved),
//     two register sets of rows (no gain once the requests are spread), XCD-unaware block order (730 MB instead of 320).
// Measured (75 k tokens, one box, profiler): dW_ih 263 (tiled) -> 180 -> 147 us, dW_hh 153 -> 140 -> 125 us; a tile takes
// ~3400 clocks against 1920 of MFMAs.  What is left: LDS traffic is 83 % of the MFMA time (147 KB of operand reads + 57 KB of
// writes per tile at 128 B / clock), ~200 vector instructions per SIMD and tile against 120 free slots, ~600 clocks at the
// barrier.  HBM delivers only the unique 320 MB (TCC misses x 128 B; L2 hit rate 64 %: the XCD placement works).
// Products and their order per element: (A hi)(B hi) + (A lo)(B hi) + (A hi)(B lo), token tiles ascending inside a slab, slabs
// reduced in slab order by tt_slab_reduce: deterministic, fp32-grade; NOT bit-identical to the tiled kernel (another slab
// partition).  The comparison build (-DTT_AB) keeps the tiled kernel with TT_WGRAD_TILED=1 (A/B).
#include "sgemm.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));

#ifndef WG_OTHER_PER_MFMA
#define WG_OTHER_PER_MFMA 3 // vector ALU / memory / LDS-write instructions scheduled behind every MFMA of a run (A/B: 2, 4, 6 no faster)
#endif
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

#ifdef TT_WG_DBG // a measuring build (tools/experiments/wgdbg.sh), never the shipped one: s_memtime clocks per tile, printed every 16th launch
__device__ unsigned long long wg_dbg[8];
#define WG_T(i) do { __builtin_amdgcn_sched_barrier(0); tm[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WG_T(i) do { } while (0)
#endif
// NT: 16-column tiles per wave along N (the workgroup's tile is 2 NT x 16 columns wide)
template <int NT, bool BMAP>
__global__ __launch_bounds__(512, 1) void wgrad16_kernel(SgemmParams p, int n_ntiles, int kchunk, int nslabs)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int NW = 2 * NT * 16;     // columns per workgroup
    constexpr int BQ = NW / 4;          // float4 per B row
    constexpr int BJ = (WG_KT * BQ + 511) / 512; // B float4 per thread and tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w & 3, wn = w >> 2;
#ifdef TT_WG_DBG
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
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

    // ---- fill mapping and its instruction budget.  The matrix pipe takes one MFMA per 16 clocks, and next to a wave that
    //      keeps it busy a vector instruction of the SIMD's other wave issues once per MFMA (tools/experiments/valu_rate.hip:
    //      16.3 clocks per v_fma / v_cvt / v_mul next to MFMAs, 25 for the packed fp32 forms, 8.7 alone): a tile's 60 MFMAs
    //      per wave hide 60 vector instructions of its partner, and every further one costs 4 - 8 clocks of an idle matrix
    //      pipe.  The first build spent 210 per wave and tile (64-bit row-address arithmetic with quarter-rate multiplies,
    //      per-float4 bound checks, a conversion the compiler spread over 17 instructions per float4); this one 80 - 90:
    //        A: thread -> token rows w + 8 j (wave-uniform: row base and the past-the-slab test live in scalar registers),
    //           float4 column tid & 63 (a wave reads 1 KB of one row);
    //        B: float4 index (tid + 512 j) mod (32 BQ) of the 32 x BQ tile (threads past the last float4 redo the first
    //           ones: no divergent tail), row idx / BQ through b_map, 32-bit byte offset row * ldb + column (24-bit multiply;
    //           tt_wgrad16_supported bounds the operand sizes);
    //        conversion: hi = f16(s v), lo = f16(s v - hi) as ONE v_fma_mix each (s a power of two: exact products), 8 per
    //           float4; only A carries the past-the-slab zero (a zero A row makes the token's contribution zero whatever
    //           its B row holds), B only the constant past-column-N zero.
    //      Every load is UNCONDITIONAL from a clamped, valid address and the source-row indices of a tile are fetched one
    //      tile before its data (an index loaded right in front of its row would put a wait between any two data loads) ----
    const int klast = K > 0 ? K - 1 : 0;
    f32x4v ra[4], rb[BJ];    // one tile of operand rows in flight (registers), two tiles staged (LDS)
    int ib[BJ];              // B source rows of the next tile to request
    int brow[BJ];            // this thread's row of the tile per piece
    unsigned bcol[BJ];       // ... and byte offset of its float4 inside a B row (clamped to the last whole float4 of the row)
    float bscale[BJ];        // 2^eb, or 0 for a float4 past column N
    int b_lds[BJ];           // where its halves go in a B image
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int idx = (tid + 512 * j) % (WG_KT * BQ), c = idx % BQ;
        brow[j] = idx / BQ;
        bcol[j] = 4u * (unsigned)min(n0 + 4 * c, p.N - 4);
        bscale[j] = n0 + 4 * c < p.N ? sb : 0.0f;
        b_lds[j] = 2 * WG_IMG + img_off(brow[j], c >> 1, (c & 1) * 8);
    }
    const unsigned ldb4 = 4u * (unsigned)p.ldb;
    const unsigned a_lane = 4u * (unsigned)(m0 + 4 * lane); // byte offset inside an A row
    int a_lds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        a_lds[j] = img_off(w + 8 * j, lane >> 1, (lane & 1) * 8);
    auto fetch_idx = [&](int k0) {
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int tok = min(k0 + brow[j], klast);
            ib[j] = BMAP ? p.b_map[tok] : tok;
        }
    };
    auto fetch_a = [&](int k0, int j) { // (scalar row address + one lane offset)
        const char *row = (const char *)(p.A + (size_t)min(k0 + w + 8 * j, klast) * (size_t)p.lda);
        return *(const f32x4v *)(row + a_lane);
    };
    auto fetch_b = [&](int j) { return *(const f32x4v *)((const char *)p.B + (size_t)(__umul24((unsigned)ib[j], ldb4) + bcol[j])); };
    // one float4 -> 4 hi + 4 lo halves (two packed registers each)
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    auto split4 = [&](f32x4v v, float s, u2v &hi, u2v &lo) {
        unsigned h0, h1, l0, l1;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h0) : "v"(s), "v"(v[0]));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h0) : "v"(s), "v"(v[1]));
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h1) : "v"(s), "v"(v[2]));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h1) : "v"(s), "v"(v[3]));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(s), "v"(v[0]), "v"(h0));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l0) : "v"(s), "v"(v[1]), "v"(h0));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(s), "v"(v[2]), "v"(h1));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l1) : "v"(s), "v"(v[3]), "v"(h1));
        hi = (u2v){h0, h1};
        lo = (u2v){l0, l1};
    };
    auto commit_a = [&](char *stage, int k0, int j, f32x4v v) { // A piece j of the tile that starts at token k0
        u2v hi, lo;
        split4(v, k0 + w + 8 * j < ke ? sa : 0.0f, hi, lo);
        *(u2v *)(stage + a_lds[j]) = hi;
        *(u2v *)(stage + WG_IMG + a_lds[j]) = lo;
    };
    auto commit_b = [&](char *stage, int j, f32x4v v) {
        u2v hi, lo;
#ifdef TT_WG_EXP_B_PRESPLIT // TIMING EXPERIMENT ONLY (wrong numbers): the 16 loaded bytes taken as 4 hi + 4 lo halves, as if K1 and the
                            // forward recurrence had left fp16 hi / lo images of X and h -- the same bytes per element, no
                            // conversion instructions for B: the most VERDICT r03 item 4 could gain in this kernel
        hi = (u2v){__float_as_uint(v[0]), __float_as_uint(v[1])};
        lo = (u2v){__float_as_uint(v[2]), __float_as_uint(v[3])};
#else
        split4(v, bscale[j], hi, lo);
#endif
        *(u2v *)(stage + b_lds[j]) = hi;
        *(u2v *)(stage + WG_IMG + b_lds[j]) = lo;
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

    // the fill of one tile cut into 12 chunks, one behind each run of NT MFMAs: a piece is converted out of its registers and
    // the same registers are requested again for the tile two ahead (so a request has a whole iteration to arrive)
    auto chunk = [&](char *nx, int k0, int c) {
        if (c < 4) {
            commit_a(nx, k0 + WG_KT, c, ra[c]);
            ra[c] = fetch_a(k0 + 2 * WG_KT, c);
        } else if (c < 4 + BJ) {
            commit_b(nx, c - 4, rb[c - 4]);
            rb[c - 4] = fetch_b(c - 4);
        } else if (c == 4 + BJ) {
            fetch_idx(k0 + 3 * WG_KT);
        }
    };
    auto fused = [&](const char *st, char *nx, int k0) {
        h8 bh[NT], bl[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bh[nt] = tr_frag(st + 2 * WG_IMG + b_off[nt], st + 2 * WG_IMG + b_off[nt] + 4 * WG_ROW);
            bl[nt] = tr_frag(st + 3 * WG_IMG + b_off[nt], st + 3 * WG_IMG + b_off[nt] + 4 * WG_ROW);
        }
        h8 ah = tr_frag(st + a_off[0], st + a_off[0] + 4 * WG_ROW);
        h8 al = tr_frag(st + WG_IMG + a_off[0], st + WG_IMG + a_off[0] + 4 * WG_ROW);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            h8 ahn = ah, aln = al;
            if (mt < 3) {
                ahn = tr_frag(st + a_off[mt + 1], st + a_off[mt + 1] + 4 * WG_ROW);
                aln = tr_frag(st + WG_IMG + a_off[mt + 1], st + WG_IMG + a_off[mt + 1] + 4 * WG_ROW);
            }
#pragma unroll
            for (int sub = 0; sub < 3; ++sub) {
#if TT_MUTATE_DROP_LO & 16
                if (sub == 0)
#endif
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sub == 1 ? al : ah, sub == 2 ? bl[nt] : bh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 5 + BJ; ++c) // the 5 + BJ pieces of the fill spread evenly over the 12 runs
                    if (c * 12 / (5 + BJ) == 3 * mt + sub)
                        chunk(nx, k0, c);
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x222, WG_OTHER_PER_MFMA, 0); // vector ALU / memory read / LDS write
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            ah = ahn;
            al = aln;
        }
    };
    if (kb < ke) { // tile kb converted into stage 0, tile kb + 32 requested, the row indices of tile kb + 64 on their way
        fetch_idx(kb);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            ra[j] = fetch_a(kb, j);
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            rb[j] = fetch_b(j);
        fetch_idx(kb + WG_KT);
#pragma unroll
        for (int c = 0; c < 12; ++c)
            chunk(lds, kb - WG_KT, c);
    }
    __syncthreads();
#ifdef TT_WG_DBG
    unsigned long long tm[3], tacc[2] = {0, 0};
#endif
    int cur = 0;
    for (int k0 = kb; k0 < ke; k0 += WG_KT) {
        WG_T(0);
        fused(lds + cur * WG_STAGE, lds + (cur ^ 1) * WG_STAGE, k0);
        WG_T(1);
        __syncthreads();
        WG_T(2);
#ifdef TT_WG_DBG
        tacc[0] += tm[1] - tm[0];
        tacc[1] += tm[2] - tm[1];
#endif
        cur ^= 1;
    }
#ifdef TT_WG_DBG
    if (NT == 5 && ke - kb > 1024 && lane == 0) {
        atomicAdd(&wg_dbg[0], tacc[0]);
        atomicAdd(&wg_dbg[1], tacc[1]);
        if (w == 0) {
            atomicAdd(&wg_dbg[2], (unsigned long long)((ke - kb) / WG_KT));
            atomicAdd(&wg_dbg[3], __builtin_amdgcn_s_memtime() - t_start);
            atomicAdd(&wg_dbg[4], 1ull);
        }
    }
#endif
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

bool wgrad_tiled() // (tt_common.h: false in the product build; the comparison build reads it at every call)
{
    return TT_AB_SWITCH(TT_WGRAD_TILED, 0) != 0;
}

} // namespace

// shapes this kernel takes: output rows a multiple of 256, N a multiple of 4, operand rows 16-byte aligned, A rows direct (no
// a_map), and a B operand of b_rows source rows that 32-bit byte offsets and 24-bit multiplies can address
bool tt_wgrad16_supported(int M, int N, int64_t lda, int64_t ldb, int64_t b_rows)
{
    return !wgrad_tiled() && M > 0 && M % WG_MT == 0 && N >= 64 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldb > 0 &&
           ldb < (1 << 22) && b_rows < (1 << 24) && b_rows * ldb < (int64_t(1) << 30);
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
    if (p.a_map || ((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15)) // (the caller checked tt_wgrad16_supported with B's row count)
        return TT_ERR_UNSUPPORTED;
#ifdef TT_WG_DBG
    static int calls = 0;
    if (++calls % 16 == 0) {
        unsigned long long h[8];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(wg_dbg), sizeof h) == hipSuccess && h[2])
            fprintf(stderr, "wgdbg NT=5: clocks per wave and tile: multiply + fill %.0f, barrier %.0f; %llu workgroups, %.1f tiles each, %.0f clocks from kernel entry to the end of the token loop\n",
                    h[0] / 8.0 / h[2], h[1] / 8.0 / h[2], h[4], (double)h[2] / h[4], (double)h[3] / h[4]);
    }
#endif
    const bool wide = p.N > 256 && p.N <= 320;
    const int NW = wide ? 160 : 128;
    const int n_ntiles = (p.N + NW - 1) / NW;
    int kchunk = (p.K + nslabs - 1) / nslabs;
    kchunk = (kchunk + WG_KT - 1) / WG_KT * WG_KT;
    const dim3 grid((unsigned)((p.M / WG_MT) * n_ntiles * ((nslabs + 7) / 8 * 8)));
    // four instantiations: column tiles per wave (5 / 4) x B rows direct or through b_map
    using Kern = void (*)(SgemmParams, int, int, int);
    static const Kern kerns[4] = {wgrad16_kernel<4, false>, wgrad16_kernel<4, true>, wgrad16_kernel<5, false>, wgrad16_kernel<5, true>};
    static bool attr_done = false;
    if (!attr_done) {
        for (Kern k : kerns)
            TT_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS));
        attr_done = true;
    }
    const Kern k = kerns[(wide ? 2 : 0) + (p.b_map ? 1 : 0)];
    hipLaunchKernelGGL(k, grid, dim3(512), WG_LDS, st, p, n_ntiles, kchunk, nslabs);
    TT_LAUNCH_CHECK();
    return TT_OK;
}
