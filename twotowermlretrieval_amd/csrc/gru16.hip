// K2 (fast form): the GRU recurrence with the hidden-state GEMM on the f16 matrix pipes at fp32 accuracy.
//
// The fp32-MFMA recurrence (encoder.hip: gru_seq_kernel) spends 24.6k CU-cycles per step on
// h[16,H] x W_hh^T[H,3H] (v_mfma_f32_16x16x4_f32 runs at 1/16 of the f16 rate) and streams all of W_hh (786 KB at
// H=256) from L2 every step.  Here both operands are SPLIT into two fp16 numbers each,
//     x * 2^s = hi + lo,   hi = fp16(x 2^s),  lo = fp16(x 2^s - hi)        (|x 2^s - hi - lo| <= 2^-24 |x 2^s|),
// and  h W = (h_hi W_hi + h_lo W_hi + h_hi W_lo) 2^-(s_h + s_W)  runs as three v_mfma_f32_16x16x32_f16 with fp32
// accumulation: fp16 x fp16 products are exact in fp32, the dropped h_lo W_lo term is <= 2^-24 of the product, so every
// product is good to ~3 * 2^-24 relative -- the error of ONE fp32 rounding, which the fp32 chain makes 256 times
// anyway.  The power-of-two scales (h: 2^10, since |h| < 1; W_hh: chosen from max|W_hh| so that the largest element
// lands in [2^13, 2^14)) keep hi AND lo in fp16's normal range for every element that matters; they are exact and
// are undone once per step on the fp32 accumulator.  3/16 of the fp32 MFMA time.
//
// With the arithmetic 5x cheaper the W_hh stream from L2 would bind (786 KB per step and workgroup at <= 64 B/clk per
// CU), so a wave keeps part of ITS slice of W_hh (the 96 gate columns of its 32 hidden units) resident for the whole
// sequence: R fragments in VGPRs, NL fragments in LDS, and only the remaining NS fragments are streamed per step
// through an NR-deep register ring that runs ahead across step boundaries.  Fragment = one MFMA B operand
// (32 k x 16 columns of hi or lo, 1 KiB per wave).  Which fragment lives where is a compile-time plan that spreads
// the streamed ones evenly over the step.
//
// One workgroup = 16 batch rows (one MFMA M tile) x all H units, H/32 waves, persistent over its rows' steps; h lives
// in LDS as two fp16 images (hi, lo), double-buffered: one barrier per step.  Gate math, length mask, training stash
// and outputs are those of gru_seq_kernel.
#include "encoder.h"
#include "sgemm.h"
#include "pack16.h"

#include <type_traits>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// One B fragment (1 KiB per wave) through a buffer load: SGPR resource + ONE VGPR lane offset + a constant that the
// compiler puts into the scalar / immediate offset fields.  (With flat 64-bit addresses hipcc hoists the ~60
// per-fragment addresses of the unrolled step out of the loop: 120 VGPRs of pointers, and spills.)
// Cache policy of the forward recurrence's per-step W_hh stream (tools/experiments/gru16_waux_ab.py, profiles/r05_k_gru16_waux.log,
// interleaved on one box each): nt (aux 2) is 37-42 % SLOWER (1.11 -> 1.52 ms at 8 192 passages: the fragments must STAY in L2,
// every workgroup of the XCD re-reads them every step); sc0 (1), sc1 (16: past the CU's L1) and sc0|sc1 (17) are within 1 % of the
// default on a second box (a first box had sc1 4 % ahead with the variants in another order): no robust gain, default kept.
#ifndef TT_G16_W_AUX
#define TT_G16_W_AUX 0
#endif
__device__ __forceinline__ h8 frag_load(__amdgpu_buffer_rsrc_t rsrc, int lane_off, int byte_off)
{
    return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, byte_off, 0));
}
// the per-step stream alone (the resident fragments are loaded once with the default policy)
__device__ __forceinline__ h8 frag_stream(__amdgpu_buffer_rsrc_t rsrc, int lane_off, int byte_off)
{
    return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, byte_off, TT_G16_W_AUX));
}

#ifndef TT_G16_R
#define TT_G16_R 21
#endif
constexpr int H_SHIFT = 10; // h is scaled by 2^10 before the split (|h| < 1)

enum { K_REG = 0, K_LDS = 1, K_STR = 2 };

// RT = row tiles of 16 batch rows per workgroup.  RT = 1: the form described above.  RT = 2 (big batches: four rounds of one-tile
// workgroups or more, e.g. evaluators.embed_corpus's 16 384 passages): a streamed W_hh fragment is multiplied with BOTH tiles' h
// fragments.  The accumulators, h registers and gi values of the second tile take the registers of 20 resident fragments (R = 1;
// 7 spills 10 - 27 registers wherever the gi loads go; 11 in LDS behind the second pair of h images, 84 streamed: 42 per 16 rows
// instead of 60) and the gi values are asked for late in the multiply instead of up front.  A row's arithmetic is the same
// instruction sequence on the same operands: results are BIT-IDENTICAL to RT = 1 (tests/test_encoder_gpu.py,
// tests/test_bench_size_gpu.py).
// What it buys is small, and why (round 4, profiles/r04_x_gru16_two_tiles.log, r04_x_gru16_ring_depth.log): document-tower call
// 4.10 -> 3.98 ms at 16 384 passages, 8.00 -> 7.56 at 32 768 (+3 - 6 %); SLOWER below two rounds of two-tile workgroups (8 192:
// equal; 4 096: 1.28 -> 1.58 ms, half the CUs idle), hence the threshold.  The W_hh stream is not what bounds the step: with 6
// more fragments streamed per step (R = 15) the 8 192-passage call goes from 2.15 to 2.20 ms, a ring of 8, 11, 12 or 18 loads in
// flight instead of 6 (at the expense of resident fragments) is 1 - 9 % SLOWER at every size -- the step (8.7 us at 8 192
// passages for 2.2 us of matrix work) is its instruction stream (144 MFMAs + ~70 loads + the gate math of 8 elements per lane, two
// waves per SIMD) at the clock the chip holds with every CU multiplying, and two tiles double that stream per workgroup.
template <int H, int RT = 1>
struct G16 {
    static constexpr int NW = H / 32;       // waves: wave w owns hidden units [32w, 32w+32) of all three gates
    static constexpr int NK = H / 32;       // k-steps of 32
    static constexpr int NF = 12 * NK;      // B fragments per wave and step: NK x 6 column tiles x {hi, lo}
    static constexpr int LDH = H + 8;       // fp16 elements per row of an h image (row stride = 4 banks: A reads spread)
    static constexpr int IMG = 16 * LDH * 2;
    static constexpr int H_BYTES = RT * 4 * IMG; // [row tile][buffer 2][hi, lo]
    static constexpr int LDS_CAP = ((160 * 1024 - H_BYTES) / NW) / 1024;
#ifndef TT_G16_R2
#define TT_G16_R2 1
#endif
    static constexpr int R = RT == 2 ? (H == 256 ? TT_G16_R2 : 16) : (H == 256 ? TT_G16_R : 16);
    static constexpr int NL = NF - R < LDS_CAP ? NF - R : LDS_CAP;
    static constexpr int NS = NF - R - NL;
#ifndef TT_G16_NR
#define TT_G16_NR 6
#endif
    static constexpr int NR = NS == 0 ? 1 : (H == 256 && RT == 1 ? TT_G16_NR : 6); // streamed fragments in flight per wave
    static constexpr int LDS_BYTES = H_BYTES + NW * NL * 1024;
#ifndef TT_G16_RT2_ROUNDS
// two tiles per workgroup from this many rounds of one-tile workgroups up.  4 until round 5; with the projected table (no Gi buffer
// to write and read back: the W_hh stream is all the recurrence waits for) the shared fragments pay earlier and more
// (tools/experiments/gru16_rt.py, profiles/r05_l_gru16_rt.log, one vs two tiles): 6 144 passages 1.00 vs 1.13 ms, 8 192: 1.21 vs
// 1.16, 12 288: 1.66 vs 1.56, 16 384: 2.16 vs 1.85, 32 768: 4.12 vs 3.56 ms (round 4, projecting calls: +3-6 % from 16 384 up)
#define TT_G16_RT2_ROUNDS 2
#endif
#ifndef TT_G16_GI_Q0
#define TT_G16_GI_Q0 14
#define TT_G16_GI_Q1 20
#endif
    static constexpr int GI_Q0 = TT_G16_GI_Q0, GI_Q1 = TT_G16_GI_Q1; // RT = 2: the multiply groups (of NK * 3 = 24) behind which the two tiles' gi loads go out
    static_assert(NS % NR == 0, "the ring must come round once per step");
    static_assert(NF <= 96, "plan tables are sized for H <= 256");
};

struct Plan {
    signed char kind[96];
    short idx[96];   // index within its kind, in consumption order
    short sfrag[96]; // fragment number of the i-th streamed fragment
};

// Spread the three kinds over the consumption order (largest-deficit-first), so that streamed fragments are
// consumed at an even pace and the ring's NR loads in flight cover the L2 latency.
template <int NF, int R, int NL>
constexpr Plan make_plan()
{
    Plan p{};
    const int total[3] = {R, NL, NF - R - NL};
    int done[3] = {0, 0, 0};
    for (int f = 0; f < NF; ++f) {
        int best = -1;
        long best_def = -(1L << 60);
        for (int k = 2; k >= 0; --k) {
            if (done[k] >= total[k])
                continue;
            const long def = (long)total[k] * (f + 1) - (long)done[k] * NF; // scaled deficit
            if (def > best_def) {
                best_def = def;
                best = k;
            }
        }
        p.kind[f] = (signed char)best;
        p.idx[f] = (short)done[best];
        if (best == K_STR)
            p.sfrag[done[best]] = (short)f;
        ++done[best];
    }
    return p;
}

template <class C>
struct PlanOf {
    static constexpr Plan value = make_plan<C::NF, C::R, C::NL>();
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// W_hh -> packed fragment order (pack16.h)
__global__ __launch_bounds__(256) void pack_whh16_kernel(const float *__restrict__ W, int H, const unsigned *__restrict__ absmax,
                                                         _Float16 *__restrict__ wp16)
{
    pack_whh16_body(W, H, absmax, wp16, (int)blockIdx.x, (int)gridDim.x);
}

__device__ __forceinline__ float fast_sigmoid16(float x) { return tt_fast_sigmoid(x); }
__device__ __forceinline__ float fast_tanh16(float x) { return tt_fast_tanh(x); }

#ifdef TT_G16_DBG // a measuring build: s_memtime clocks per phase of the step (wave 0 and wave 4 of every workgroup), printed by gru16_launch
__device__ unsigned long long g16_dbg[16];
#define G16_T(i) do { __builtin_amdgcn_sched_barrier(0); tm[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define G16_T(i) do { } while (0)
#endif
// GATHER: the projections come from the projected table (GruParams::gi_ids): one more dependent load per row and step, the
// id, asked for well ahead of the row loads that need it -- RT = 1: during the previous step (4 registers across the step
// boundary); RT = 2: GI_AHEAD multiply groups in front of the tile's projection loads (no registers to spare at the boundary).
// The arithmetic does not change: outputs are bit-identical to the same rows read from a [tokens][3H] buffer.
template <int H, int RT, bool GATHER>
__global__ __launch_bounds__(H / 32 * 64) void gru_seq16_kernel(GruParams p)
{
    using C = G16<H, RT>;
    using P = PlanOf<C>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const GruDir d = p.dir[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * ENC_RB * RT;

    int len_e[RT][4], off_e[RT][4];
    int steps = 0;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int br = row0 + rt * ENC_RB + kq * 4 + e;
            const int rid = br < p.B ? p.perm[br] : -1;
            len_e[rt][e] = rid >= 0 ? p.len[rid] : 0;
            off_e[rt][e] = rid >= 0 ? p.tok_off[rid] : 0;
            steps = max(steps, len_e[rt][e]); // block-wide max length
        }
    steps = max(steps, __shfl_xor(steps, 16));
    steps = max(steps, __shfl_xor(steps, 32));

    const int ex = tt_pow2_exponent(*d.wmax);
    const float up = ldexpf(1.0f, H_SHIFT + ex), down = ldexpf(1.0f, -(H_SHIFT + ex));
    int unit[2];
    float bias[3][2]; // b_hh scaled like the products: the accumulators start there
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        unit[ct] = 32 * w + 16 * ct + j;
#pragma unroll
        for (int g = 0; g < 3; ++g)
            bias[g][ct] = d.b_hh[g * H + unit[ct]] * up;
    }
    for (int i = threadIdx.x; i < C::H_BYTES / 4; i += C::NW * 64)
        ((int *)lds)[i] = 0; // h_0 = 0 in both buffers, both parts
    float hreg[RT][2][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                hreg[rt][ct][e] = 0.0f;

    // ---- this wave's slice of W_hh: resident fragments into VGPRs / LDS, the ring's first NR streamed ones ----
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((const char *)d.wp + (size_t)w * C::NF * 1024), 0, C::NF * 1024, 0x00020000);
    const int loff = lane * 16;
    char *wlds = lds + C::H_BYTES + w * C::NL * 1024 + lane * 16;
    h8 wreg[C::R];
    h8 ring[C::NR];
    static_for<0, C::NF>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        constexpr int kind = P::value.kind[f], idx = P::value.idx[f];
        if constexpr (kind == K_REG)
            wreg[idx] = frag_load(wsrc, loff, f * 1024);
        else if constexpr (kind == K_LDS)
            *(h8 *)(wlds + idx * 1024) = frag_load(wsrc, loff, f * 1024);
    });
    if constexpr (C::NS > 0)
        static_for<0, C::NR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int off = P::value.sfrag[i] * 1024;
            ring[i] = frag_stream(wsrc, loff, off);
        });
    __syncthreads();

    const int H3 = 3 * H;
    int cur = 0;
    // GATHER: grow = the projected-table rows of the step whose projections are loaded next
    constexpr int GI_AHEAD = 5;
    int grow[RT][4];
    auto tok_at = [&](int rt, int e, int ss) {
        const int t = d.reverse ? len_e[rt][e] - 1 - ss : ss;
        return off_e[rt][e] + (ss < len_e[rt][e] ? t : 0);
    };
    auto load_ids = [&](auto rtc, int ss) {
        constexpr int rt = decltype(rtc)::value;
        if constexpr (GATHER) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                grow[rt][e] = p.gi_ids[tok_at(rt, e, ss)];
        }
    };
    if constexpr (GATHER && RT == 1)
        load_ids(std::integral_constant<int, 0>{}, 0);
#ifdef TT_G16_DBG
    unsigned long long tm[5], tacc[4] = {0, 0, 0, 0};
#endif
    for (int s = 0; s < steps; ++s) {
        G16_T(0);
        float giv[RT][3][2][4];
        // this step's token of row e of tile rt (a valid token even when the row is done); RT = 2 recomputes it where it is used
        // instead of keeping it through the multiply
        auto token = [&](int rt, int e) { return tok_at(rt, e, s); };
        // the input projections of this step's tokens: RT = 1 asks for them up front (24 registers that wait through the whole
        // multiply); RT = 2 has no registers for 48 of them there and asks late in the multiply (tile 0 behind group GI_Q0, tile 1
        // behind GI_Q1: they arrive while the last groups run)
        auto load_gi = [&](auto rtc) {
            constexpr int rt = decltype(rtc)::value;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const size_t gr = GATHER ? (size_t)min((unsigned)grow[rt][e], p.gi_rows - 1u) : (size_t)token(rt, e);
                const float *row = d.gi + gr * H3 + unit[0];
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        giv[rt][g][ct][e] = row[g * H + 16 * ct];
            }
        };
        if constexpr (RT == 1) {
            load_gi(std::integral_constant<int, 0>{});
            load_ids(std::integral_constant<int, 0>{}, s + 1); // (GATHER) the next step's rows; s + 1 == steps reads a valid token's id
        }
        f32x4v acc[RT][6]; // tile t = 2 g + ct
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc[rt][2 * g + ct] = (f32x4v){bias[g][ct], bias[g][ct], bias[g][ct], bias[g][ct]};

        // row tile rt's images: [rt][buffer][hi, lo]
        const char *img = lds + cur * 2 * C::IMG + j * (C::LDH * 2) + kq * 16;
        // Consumption order: k-step s2, pair of column tiles, {hi t0, hi t1, lo t0, lo t1}.  Software pipeline, pinned
        // by scheduling fences (left alone, hipcc hoists every load of the step and spills): group q issues the LDS
        // reads of group q + 1 (its LDS-resident B fragments; the next k-step's A fragments one group early), then its
        // own six MFMAs per row tile, then refills the ring slots it consumed.
        G16_T(1);
        h8 a_hi[2][RT], a_lo[2][RT]; // by k-step parity
        h8 lbuf[2][4];               // LDS-resident B fragments of the current / next group
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            a_hi[0][rt] = *(const h8 *)(img + rt * 4 * C::IMG);
            a_lo[0][rt] = *(const h8 *)(img + rt * 4 * C::IMG + C::IMG);
        }
        static_for<0, 4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int kind = P::value.kind[i], idx = P::value.idx[i];
            if constexpr (kind == K_LDS)
                lbuf[0][i] = *(const h8 *)(wlds + idx * 1024);
        });
        static_for<0, C::NK * 3>([&](auto qc) {
            constexpr int q = decltype(qc)::value; // (k-step, pair)
            constexpr int s2 = q / 3, pair = q % 3, f0 = 4 * q;
            if constexpr (q + 1 < C::NK * 3) {
                static_for<0, 4>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, f = f0 + 4 + i;
                    constexpr int kind = P::value.kind[f], idx = P::value.idx[f];
                    if constexpr (kind == K_LDS)
                        lbuf[(q + 1) & 1][i] = *(const h8 *)(wlds + idx * 1024);
                });
                if constexpr (pair == 1 && s2 + 1 < C::NK) { // the next k-step's A fragments, 1.5 groups ahead
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        a_hi[(s2 + 1) & 1][rt] = *(const h8 *)(img + rt * 4 * C::IMG + (s2 + 1) * 64);
                        a_lo[(s2 + 1) & 1][rt] = *(const h8 *)(img + rt * 4 * C::IMG + C::IMG + (s2 + 1) * 64);
                    }
                }
            }
            h8 b[4];
            static_for<0, 4>([&](auto ic) {
                constexpr int i = decltype(ic)::value, f = f0 + i;
                constexpr int kind = P::value.kind[f], idx = P::value.idx[f];
#ifdef TT_G16_EXP_NO_OPERAND_WAIT // MEASUREMENT ONLY (wrong results): every B fragment is register 0 -- what the multiply costs without operand latency
                b[i] = wreg[0];
#else
                if constexpr (kind == K_REG)
                    b[i] = wreg[idx];
                else if constexpr (kind == K_LDS)
                    b[i] = lbuf[q & 1][i];
                else
                    b[i] = ring[idx % C::NR];
#endif
            });
            constexpr int t0 = 2 * pair, t1 = 2 * pair + 1;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                acc[rt][t0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[s2 & 1][rt], b[0], acc[rt][t0], 0, 0, 0);
                acc[rt][t1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[s2 & 1][rt], b[1], acc[rt][t1], 0, 0, 0);
#if !(TT_MUTATE_DROP_LO & 1)
                acc[rt][t0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[s2 & 1][rt], b[0], acc[rt][t0], 0, 0, 0);
                acc[rt][t1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[s2 & 1][rt], b[1], acc[rt][t1], 0, 0, 0);
                acc[rt][t0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[s2 & 1][rt], b[2], acc[rt][t0], 0, 0, 0);
                acc[rt][t1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[s2 & 1][rt], b[3], acc[rt][t1], 0, 0, 0);
#endif
            }
            // refill the ring slots this group consumed: the fragment NR streamed fragments further on (the next
            // step's first ones near the end of this step -- W_hh does not change)
            static_for<0, 4>([&](auto ic) {
                constexpr int i = decltype(ic)::value, f = f0 + i;
                constexpr int kind = P::value.kind[f], idx = P::value.idx[f];
                if constexpr (kind == K_STR) {
                    constexpr int off = P::value.sfrag[(idx + C::NR) % C::NS] * 1024;
#ifdef TT_G16_EXP_SKIP_STREAM // MEASUREMENT ONLY (wrong results): every TT_G16_EXP_SKIP_STREAM-th streamed fragment is not fetched
                    if constexpr (idx % TT_G16_EXP_SKIP_STREAM != 0)
#endif
                    ring[idx % C::NR] = frag_stream(wsrc, loff, off);
                }
            });
            if constexpr (RT == 2 && q == C::GI_Q0 - GI_AHEAD)
                load_ids(std::integral_constant<int, 0>{}, s);
            if constexpr (RT == 2 && q == C::GI_Q1 - GI_AHEAD)
                load_ids(std::integral_constant<int, RT - 1>{}, s);
            if constexpr (RT == 2 && q == C::GI_Q0)
                load_gi(std::integral_constant<int, 0>{});
            if constexpr (RT == 2 && q == C::GI_Q1)
                load_gi(std::integral_constant<int, RT - 1>{});
            __builtin_amdgcn_sched_barrier(0);
        });

        G16_T(2);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            char *nimg = lds + rt * 4 * C::IMG + (cur ^ 1) * 2 * C::IMG;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r = fast_sigmoid16(giv[rt][0][ct][e] + acc[rt][ct][e] * down);
                    const float z = fast_sigmoid16(giv[rt][1][ct][e] + acc[rt][2 + ct][e] * down);
                    const float ghn = acc[rt][4 + ct][e] * down;
                    const float n = fast_tanh16(giv[rt][2][ct][e] + r * ghn);
                    const float hn = (hreg[rt][ct][e] - n) * z + n;
                    if (s < len_e[rt][e]) {
                        hreg[rt][ct][e] = hn;
                        const size_t tok = (size_t)token(rt, e);
                        if (d.out_seq)
                            d.out_seq[tok * p.out_ld + d.out_col0 + unit[ct]] = hn;
                        if (d.gates) {
                            float *gs = d.gates + tok * 4 * H + unit[ct];
                            gs[0] = r;
                            gs[H] = z;
                            gs[2 * H] = n;
                            gs[3 * H] = ghn;
                        }
                    }
                    const float hs = hreg[rt][ct][e] * (float)(1 << H_SHIFT);
                    const _Float16 hi = (_Float16)hs;
                    const _Float16 lo = (_Float16)(hs - (float)hi);
                    _Float16 *dst = (_Float16 *)nimg + (kq * 4 + e) * C::LDH + unit[ct];
                    dst[0] = hi;
                    dst[C::IMG / 2] = lo;
                }
        }
        G16_T(3);
        __syncthreads();
        G16_T(4);
#ifdef TT_G16_DBG
        for (int i = 0; i < 4; ++i)
            tacc[i] += tm[i + 1] - tm[i];
#endif
        cur ^= 1;
    }
#ifdef TT_G16_DBG
    if (lane == 0 && (w == 0 || w == 4)) { // [setup, multiply, gate math + image writes, barrier] and the step count, per wave half
        for (int i = 0; i < 4; ++i)
            atomicAdd(&g16_dbg[8 * (w >> 2) + i], tacc[i]);
        atomicAdd(&g16_dbg[8 * (w >> 2) + 4], (unsigned long long)steps);
    }
#endif
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int br = row0 + rt * ENC_RB + kq * 4 + e;
                if (br < p.B)
                    d.h_final[(size_t)p.perm[br] * H + unit[ct]] = hreg[rt][ct][e];
            }
}

// ------------------------------------------------------------------ reverse-time recurrence (training)
// dh_{t-1} = dh_t z + dGh W_hh with dGh = [dr_pre, dz_pre, dn_pre r] (16 rows x 3H) on the same three-product
// fp16 split.  W_hh is fixed, so its scale and its resident / LDS / streamed fragments work as in the forward
// kernel (K = 3H: 24 k-steps, two column tiles per wave).  dGh are GRADIENTS: their magnitude is arbitrary and
// changes from row to row and step to step, so every batch row is scaled by its own power of two, taken from the
// row's largest |element| of THIS step (lane-local maxima -> 16-lane shuffle -> one LDS word per wave and row ->
// barrier -> every lane reads the 8 words of its rows), which puts the row maximum in [2^13, 2^14): hi and lo are then
// normal fp16 numbers for every element within 2^-27 of the row maximum, smaller ones err by <= 2^-25 in scaled units,
// i.e. <= 2^-38 of the row maximum.  The scales are exact and are undone on the fp32 accumulator.
template <int H>
struct B16 {
    static constexpr int NW = H / 32;
    static constexpr int NK = 3 * H / 32;   // k-steps of 32 over the 3H gate columns
    static constexpr int NF = 4 * NK;       // per k-step: {hi t0, hi t1, lo t0, lo t1}
    static constexpr int LDG = 3 * H + 8;   // fp16 elements per row of the dGh images
    static constexpr int IMG = 16 * LDG * 2;
    static constexpr int A_BYTES = 2 * IMG; // hi, lo (single buffer: two barriers per step)
    static constexpr int RM_BYTES = 2 * NW * 16 * 4; // per-wave row maxima, double-buffered
    static constexpr int LDS_CAP = ((160 * 1024 - A_BYTES - RM_BYTES) / NW) / 1024;
#ifndef TT_B16_R
#define TT_B16_R 15
#endif
    static constexpr int R = H == 256 ? TT_B16_R : 16; // (15: room for the bias sums and the operand maxima kept in registers)
    static constexpr int NL = NF - R < LDS_CAP ? NF - R : LDS_CAP;
    static constexpr int NS = NF - R - NL;
    static constexpr int NR = NS == 0 ? 1 : (NS % 6 == 0 ? 6 : (NS % 5 == 0 ? 5 : (NS % 4 == 0 ? 4 : 3)));
    static constexpr int LDS_BYTES = A_BYTES + RM_BYTES + NW * NL * 1024;
    static_assert(NS % NR == 0, "the ring must come round once per step");
    static_assert(NF <= 96, "plan tables are sized for H <= 256");
};

// Packed order for dh = dGh W_hh: wave w, fragment f = (s, within): s = f / 4 the k-step over gate rows,
// part = within >> 1, column tile t = within & 1.  Lane (n, kq) holds W_hh[32 s + 8 kq + i][32 w + 16 t + n], i < 8.
__global__ __launch_bounds__(256) void pack_whh16_t_kernel(const float *__restrict__ W, int H, const unsigned *__restrict__ absmax,
                                                           _Float16 *__restrict__ wtp16)
{
    const int NK = 3 * H / 32, NF = 4 * NK;
    const float sc = ldexpf(1.0f, tt_pow2_exponent(*absmax));
    const int n = (H / 32) * NF * 64;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int lane = i & 63;
        const int f = (i >> 6) % NF, w = (i >> 6) / NF;
        const int s = f / 4, within = f % 4, part = within >> 1, t = within & 1;
        const float *src = W + (size_t)(32 * s + 8 * (lane >> 4)) * H + 32 * w + 16 * t + (lane & 15);
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = src[(size_t)e * H] * sc;
            const _Float16 hi = (_Float16)x;
            o[e] = part ? (_Float16)(x - (float)hi) : hi;
        }
        *(h8 *)(wtp16 + (size_t)i * 8) = o;
    }
}

template <int H>
__global__ __launch_bounds__(B16<H>::NW * 64) void gru_bwd16_kernel(GruBwdParams p)
{
    const uint64_t drop_seed_v = (p.drop_p > 0.0f && p.drop_seed_ptr) ? *p.drop_seed_ptr : p.drop_seed;
    using C = B16<H>;
    using P = PlanOf<C>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const GruBwdDir d = p.dir[blockIdx.y];
    constexpr int H3 = 3 * H;
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
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        unit[ct] = 32 * w + 16 * ct + j;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            dh[ct][e] = (d.d_hfin && rid_e[e] >= 0) ? d.d_hfin[(size_t)rid_e[e] * H + unit[ct]] : 0.0f;
    }
    const int exw = tt_pow2_exponent(*d.wmax);

    char *const img = lds;                                  // [hi, lo][16][LDG] fp16
    float *const rmax = (float *)(lds + C::A_BYTES);        // [2][NW][16]
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((const char *)d.wtp + (size_t)w * C::NF * 1024), 0, C::NF * 1024, 0x00020000);
    const int loff = lane * 16;
    char *wlds = lds + C::A_BYTES + C::RM_BYTES + w * C::NL * 1024 + lane * 16;
    h8 wreg[C::R];
    h8 ring[C::NR];
    static_for<0, C::NF>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        constexpr int kind = P::value.kind[f], idx = P::value.idx[f];
        if constexpr (kind == K_REG)
            wreg[idx] = frag_load(wsrc, loff, f * 1024);
        else if constexpr (kind == K_LDS)
            *(h8 *)(wlds + idx * 1024) = frag_load(wsrc, loff, f * 1024);
    });
    if constexpr (C::NS > 0)
        static_for<0, C::NR>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int off = P::value.sfrag[i] * 1024;
            ring[i] = frag_load(wsrc, loff, off);
        });

    // The stash of a step (gates r,z,n,ghn, h_{t-1}, upstream d_seq) is loaded one step AHEAD, so the global
    // latency hides under the previous step's MFMA loop instead of sitting on the serial path of every step.
    struct Stash {
        float r[2][4], z[2][4], n[2][4], ghn[2][4], hp[2][4], dsv[2][4];
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
                st.r[ct][e] = st.z[ct][e] = st.n[ct][e] = st.ghn[ct][e] = st.hp[ct][e] = st.dsv[ct][e] = 0.0f;
                if (a) {
                    const float *gs = d.gates + tok * 4 * H + u;
                    st.r[ct][e] = gs[0];
                    st.z[ct][e] = gs[H];
                    st.n[ct][e] = gs[2 * H];
                    st.ghn[ct][e] = gs[3 * H];
                    if (s > 0)
                        st.hp[ct][e] = d.hseq[ptok * p.ld + d.col0 + u];
                    if (d.d_seq)
                        st.dsv[ct][e] = d.d_seq[tok * p.ld + d.col0 + u];
                }
            }
        }
    };
    Stash cur_st, next_st;
    load_stash(steps - 1, cur_st);
    int rb = 0;
    float bsum[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}; // column sums over this lane's rows and all steps: dr, dz, dn, dn r
    float mx_i = 0.0f, mx_h = 0.0f;                        // max |dGi|, max |dGh| over the same elements

    for (int s = steps - 1; s >= 0; --s) {
        float direct[2][4];
        float gv[3][2][4]; // dr_pre, dz_pre, dn_pre r: this lane's elements of dGh
        float mrow[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        bool act[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            act[e] = s < len_e[e];
            const int t = d.reverse ? len_e[e] - 1 - s : s;
            const size_t tok = (size_t)(off_e[e] + (act[e] ? t : 0));
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int u = unit[ct];
                float dr_pre = 0.0f, dz_pre = 0.0f, dghn_v = 0.0f;
                direct[ct][e] = 0.0f;
                if (act[e]) {
                    const float r = cur_st.r[ct][e], z = cur_st.z[ct][e], n = cur_st.n[ct][e], ghn = cur_st.ghn[ct][e];
                    const float hp = cur_st.hp[ct][e];
                    float dsv = cur_st.dsv[ct][e];
                    if (d.d_seq && p.drop_p > 0.0f)
                        dsv *= tt_dropout_scale(drop_seed_v, p.drop_layer,
                                                ((uint64_t)rid_e[e] * p.T + t) * p.ld + d.col0 + u, p.drop_p);
                    const float dhv = dh[ct][e] + dsv;
                    const float dn_pre = dhv * (1.0f - z) * (1.0f - n * n);
                    dz_pre = dhv * (hp - n) * z * (1.0f - z);
                    dr_pre = dn_pre * ghn * r * (1.0f - r);
                    dghn_v = dn_pre * r;
                    direct[ct][e] = dhv * z;
                    float *go = d.dgi + tok * H3 + u;
                    go[0] = dr_pre;
                    go[H] = dz_pre;
                    go[2 * H] = dn_pre;
                    float *gh = d.dghn + tok * H3 + u; // dGh whole (differs from dGi in the n column only)
                    gh[0] = dr_pre;
                    gh[H] = dz_pre;
                    gh[2 * H] = dghn_v;
                    bsum[0][ct] += dr_pre;
                    bsum[1][ct] += dz_pre;
                    bsum[2][ct] += dn_pre;
                    bsum[3][ct] += dghn_v;
                    const float m2 = fmaxf(fabsf(dr_pre), fabsf(dz_pre));
                    mx_i = fmaxf(mx_i, fmaxf(m2, fabsf(dn_pre)));
                    mx_h = fmaxf(mx_h, fmaxf(m2, fabsf(dghn_v)));
                }
                gv[0][ct][e] = dr_pre;
                gv[1][ct][e] = dz_pre;
                gv[2][ct][e] = dghn_v;
                mrow[e] = fmaxf(mrow[e], fmaxf(fmaxf(fabsf(dr_pre), fabsf(dz_pre)), fabsf(dghn_v)));
            }
        }
        // row maxima: the 16 lanes of a kq group hold the 32 units of this wave for rows 4 kq .. 4 kq + 3
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int off = 1; off < 16; off <<= 1)
                mrow[e] = fmaxf(mrow[e], __shfl_xor(mrow[e], off));
            if (j == 0)
                rmax[(rb * C::NW + w) * 16 + kq * 4 + e] = mrow[e];
        }
        load_stash(s - 1, next_st); // in flight during the MFMA loop below
        __syncthreads();            // B1: row maxima visible; every wave is done reading the previous step's images
        float down[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float m = 0.0f;
#pragma unroll
            for (int ww = 0; ww < C::NW; ++ww)
                m = fmaxf(m, rmax[(rb * C::NW + ww) * 16 + kq * 4 + e]);
            const int er = tt_pow2_exponent(__float_as_uint(m));
            const float upr = ldexpf(1.0f, er);
            down[e] = ldexpf(1.0f, -(er + exw));
            _Float16 *dst = (_Float16 *)img + (kq * 4 + e) * C::LDG;
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const float x = gv[g][ct][e] * upr;
                    const _Float16 hi = (_Float16)x;
                    dst[g * H + unit[ct]] = hi;
                    dst[C::IMG / 2 + g * H + unit[ct]] = (_Float16)(x - (float)hi);
                }
        }
        rb ^= 1;
        __syncthreads(); // B2: the dGh images are complete

        f32x4v acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        const char *arow = img + j * (C::LDG * 2) + kq * 16;
        h8 a_hi[2], a_lo[2]; // by k-step parity
        h8 lbuf[2][4];
        a_hi[0] = *(const h8 *)(arow);
        a_lo[0] = *(const h8 *)(arow + C::IMG);
        static_for<0, 4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int kind = P::value.kind[i], idx = P::value.idx[i];
            if constexpr (kind == K_LDS)
                lbuf[0][i] = *(const h8 *)(wlds + idx * 1024);
        });
        static_for<0, C::NK>([&](auto qc) {
            constexpr int q = decltype(qc)::value, f0 = 4 * q;
            if constexpr (q + 1 < C::NK) {
                static_for<0, 4>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, f = f0 + 4 + i;
                    constexpr int kind = P::value.kind[f], idx = P::value.idx[f];
                    if constexpr (kind == K_LDS)
                        lbuf[(q + 1) & 1][i] = *(const h8 *)(wlds + idx * 1024);
                });
                a_hi[(q + 1) & 1] = *(const h8 *)(arow + (q + 1) * 64);
                a_lo[(q + 1) & 1] = *(const h8 *)(arow + C::IMG + (q + 1) * 64);
            }
            h8 b[4];
            static_for<0, 4>([&](auto ic) {
                constexpr int i = decltype(ic)::value, f = f0 + i;
                constexpr int kind = P::value.kind[f], idx = P::value.idx[f];
                if constexpr (kind == K_REG)
                    b[i] = wreg[idx];
                else if constexpr (kind == K_LDS)
                    b[i] = lbuf[q & 1][i];
                else
                    b[i] = ring[idx % C::NR];
            });
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[q & 1], b[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[q & 1], b[1], acc[1], 0, 0, 0);
#if !(TT_MUTATE_DROP_LO & 2)
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[q & 1], b[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[q & 1], b[1], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[q & 1], b[2], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[q & 1], b[3], acc[1], 0, 0, 0);
#endif
            static_for<0, 4>([&](auto ic) {
                constexpr int i = decltype(ic)::value, f = f0 + i;
                constexpr int kind = P::value.kind[f], idx = P::value.idx[f];
                if constexpr (kind == K_STR) {
                    constexpr int off = P::value.sfrag[(idx + C::NR) % C::NS] * 1024;
                    ring[idx % C::NR] = frag_load(wsrc, loff, off);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (act[e])
                    dh[ct][e] = direct[ct][e] + acc[ct][e] * down[e];
        cur_st = next_st;
    }
    // bias gradients of this row group (rows live in the four kq lane groups) and the operand maxima
    if (d.bias_slab) {
        float *slab = d.bias_slab + (size_t)blockIdx.x * 2 * H3;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = bsum[g][ct];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                bsum[g][ct] = v;
            }
            if (kq == 0) {
                slab[unit[ct]] = bsum[0][ct];              // d b_ih = colsum(dGi) = [dr, dz, dn]
                slab[H + unit[ct]] = bsum[1][ct];
                slab[2 * H + unit[ct]] = bsum[2][ct];
                slab[H3 + unit[ct]] = bsum[0][ct];         // d b_hh = colsum(dGh) = [dr, dz, dn r]
                slab[H3 + H + unit[ct]] = bsum[1][ct];
                slab[H3 + 2 * H + unit[ct]] = bsum[3][ct];
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mx_i = fmaxf(mx_i, __shfl_xor(mx_i, off));
            mx_h = fmaxf(mx_h, __shfl_xor(mx_h, off));
        }
        if (lane == 0) { // non-negative floats order like their bit patterns
            atomicMax(d.mx_dgi, __float_as_uint(mx_i));
            atomicMax(d.mx_dghn, __float_as_uint(mx_h));
        }
    }
}

// Two row tiles per workgroup when the launch is more than one round of one-tile workgroups anyway (one workgroup per CU: the
// weights fill the LDS) -- fewer, fatter workgroups then cost nothing in parallelism and halve the W_hh stream per row.  Below
// that every row group gets a CU of its own and finishes sooner alone.
inline bool gru16_two_tiles(int B, int ndir)
{
    const int mode = TT_AB_SWITCH(TT_GRU16_RT, 2); // comparison build: 1 = never, 3 = always (tests, tools/experiments/gru16_rt.py)
    if (mode != 2)
        return mode > 2;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return false;
    return (long)((B + ENC_RB - 1) / ENC_RB) * ndir >= (long)TT_G16_RT2_ROUNDS * cus;
}

template <int H, int RT, bool GATHER>
int launch16g(const GruParams &gp, int ndir, hipStream_t st)
{
    using C = G16<H, RT>;
    TT_HIP_CHECK(hipFuncSetAttribute((const void *)gru_seq16_kernel<H, RT, GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    hipLaunchKernelGGL((gru_seq16_kernel<H, RT, GATHER>), dim3((gp.B + ENC_RB * RT - 1) / (ENC_RB * RT), ndir), dim3(C::NW * 64), C::LDS_BYTES, st, gp);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

template <int H, int RT>
int launch16(const GruParams &gp, int ndir, hipStream_t st)
{
    if (gp.gi_ids && gp.gi_rows == 0)
        return tt_fail(TT_ERR_BAD_SHAPE, "gru16_launch: projected table without rows");
    return gp.gi_ids ? launch16g<H, RT, true>(gp, ndir, st) : launch16g<H, RT, false>(gp, ndir, st);
}

} // namespace

bool gru16_supported(int H) { return H == 256 || H == 128; }

// W_hh [3H][H] fp32 -> absmax word + packed fp16 hi/lo fragments (3H*H*4 bytes: the size of the fp32 matrix)
int gru16_pack(const float *W_hh, int H, unsigned *absmax /*zeroed by the caller on the stream*/, void *wp16, hipStream_t st)
{
    TT_RC_CHECK(tt_absmax(W_hh, (int64_t)3 * H * H, absmax, st));
    hipLaunchKernelGGL(pack_whh16_kernel, dim3(96), dim3(256), 0, st, W_hh, H, (const unsigned *)absmax, (_Float16 *)wp16);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

int gru16_launch(const GruParams &gp, int ndir, hipStream_t st)
{
#ifdef TT_G16_DBG
    static int calls = 0;
    if (++calls % 4 == 0) {
        unsigned long long h[16];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g16_dbg), sizeof h) == hipSuccess)
            for (int k = 0; k < 2; ++k)
                if (h[8 * k + 4])
                    fprintf(stderr, "g16dbg waves %d: clocks per step: setup %.0f, multiply %.0f, gates + images %.0f, barrier %.0f (steps %llu)\n", 4 * k,
                            (double)h[8 * k] / h[8 * k + 4], (double)h[8 * k + 1] / h[8 * k + 4], (double)h[8 * k + 2] / h[8 * k + 4],
                            (double)h[8 * k + 3] / h[8 * k + 4], h[8 * k + 4]);
        unsigned long long z[16] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g16_dbg), z, sizeof z);
    }
#endif
    const bool two = gru16_two_tiles(gp.B, ndir);
    if (gp.H == 256)
        return two ? launch16<256, 2>(gp, ndir, st) : launch16<256, 1>(gp, ndir, st);
    if (gp.H == 128)
        return launch16<128, 1>(gp, ndir, st);
    return tt_fail(TT_ERR_UNSUPPORTED, "gru16_launch: H=%d", gp.H);
}

// absmax: max |W_hh| as the forward's gru16_pack left it (same weights, same workspace)
int gru16_pack_t(const float *W_hh, int H, const unsigned *absmax, void *wtp16, hipStream_t st)
{
    hipLaunchKernelGGL(pack_whh16_t_kernel, dim3(96), dim3(256), 0, st, W_hh, H, (const unsigned *)absmax, (_Float16 *)wtp16);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

template <int H>
static int launch_bwd16(const GruBwdParams &bp, int ndir, hipStream_t st)
{
    using C = B16<H>;
    TT_HIP_CHECK(hipFuncSetAttribute((const void *)gru_bwd16_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    hipLaunchKernelGGL(gru_bwd16_kernel<H>, dim3((bp.B + ENC_RB - 1) / ENC_RB, ndir), dim3(C::NW * 64), C::LDS_BYTES, st, bp);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

int gru16_bwd_launch(const GruBwdParams &bp, int ndir, hipStream_t st)
{
    if (bp.H == 256)
        return launch_bwd16<256>(bp, ndir, st);
    if (bp.H == 128)
        return launch_bwd16<128>(bp, ndir, st);
    return tt_fail(TT_ERR_UNSUPPORTED, "gru16_bwd_launch: H=%d", bp.H);
}
