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

#include <type_traits>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// One B fragment (1 KiB per wave) through a buffer load: SGPR resource + ONE VGPR lane offset + a constant that the
// compiler puts into the scalar / immediate offset fields.  (With flat 64-bit addresses hipcc hoists the ~60
// per-fragment addresses of the unrolled step out of the loop: 120 VGPRs of pointers, and spills.)
__device__ __forceinline__ h8 frag_load(__amdgpu_buffer_rsrc_t rsrc, int lane_off, int byte_off)
{
    return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, byte_off, 0));
}

#ifndef TT_G16_R
#define TT_G16_R 21
#endif
constexpr int H_SHIFT = 10; // h is scaled by 2^10 before the split (|h| < 1)

enum { K_REG = 0, K_LDS = 1, K_STR = 2 };

template <int H>
struct G16 {
    static constexpr int NW = H / 32;       // waves: wave w owns hidden units [32w, 32w+32) of all three gates
    static constexpr int NK = H / 32;       // k-steps of 32
    static constexpr int NF = 12 * NK;      // B fragments per wave and step: NK x 6 column tiles x {hi, lo}
    static constexpr int LDH = H + 8;       // fp16 elements per row of an h image (row stride = 4 banks: A reads spread)
    static constexpr int IMG = 16 * LDH * 2;
    static constexpr int H_BYTES = 4 * IMG; // [buffer 2][hi, lo]
    static constexpr int LDS_CAP = ((160 * 1024 - H_BYTES) / NW) / 1024;
    static constexpr int R = H == 256 ? TT_G16_R : 16;
    static constexpr int NL = NF - R < LDS_CAP ? NF - R : LDS_CAP;
    static constexpr int NS = NF - R - NL;
    static constexpr int NR = NS == 0 ? 1 : 6;
    static constexpr int LDS_BYTES = H_BYTES + NW * NL * 1024;
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
template <int H>
constexpr Plan make_plan()
{
    using C = G16<H>;
    Plan p{};
    const int total[3] = {C::R, C::NL, C::NS};
    int done[3] = {0, 0, 0};
    for (int f = 0; f < C::NF; ++f) {
        int best = -1;
        long best_def = -(1L << 60);
        for (int k = 2; k >= 0; --k) {
            if (done[k] >= total[k])
                continue;
            const long def = (long)total[k] * (f + 1) - (long)done[k] * C::NF; // scaled deficit
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

template <int H>
struct PlanOf {
    static constexpr Plan value = make_plan<H>();
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// exponent e with max|W| 2^e in [2^13, 2^14) (0 for an all-zero or non-finite matrix)
__host__ __device__ inline int gru16_exponent(unsigned absmax_bits)
{
    const int ex = (int)((absmax_bits >> 23) & 0xff);
    if (ex == 0 || ex == 255)
        return 0;
    int e = 13 - (ex - 127);
    e = e > 100 ? 100 : e;
    e = e < -100 ? -100 : e;
    return e;
}

__global__ __launch_bounds__(256) void whh_absmax_kernel(const float *__restrict__ W, int n, unsigned *__restrict__ out)
{
    float m = 0.0f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        m = fmaxf(m, fabsf(W[i]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0)
        atomicMax(out, __float_as_uint(m)); // non-negative floats order like their bit patterns
}

// Packed order: wave w, fragment f = (s, pair, within): s = f / 12 the k-step, pair = (f % 12) / 4 the pair of column
// tiles {2 pair, 2 pair + 1}, within = f % 4 -> part = within >> 1 (0 hi, 1 lo), tile t = 2 pair + (within & 1),
// gate g = t >> 1, ct = t & 1.  Lane (n = lane & 15, kq = lane >> 4) holds the 8 fp16 of
//   W_hh[g H + 32 w + 16 ct + n][32 s + 8 kq .. + 7]  (scaled by 2^e; hi or lo part):  16 bytes at
//   wp16[((w NF + f) 64 + lane) 8 ...].
__global__ __launch_bounds__(256) void pack_whh16_kernel(const float *__restrict__ W, int H, const unsigned *__restrict__ absmax,
                                                         _Float16 *__restrict__ wp16)
{
    const int NK = H / 32, NF = 12 * NK;
    const float sc = ldexpf(1.0f, gru16_exponent(*absmax));
    const int n = (H / 32) * NF * 64; // (wave, fragment, lane) triples
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int lane = i & 63;
        const int f = (i >> 6) % NF, w = (i >> 6) / NF;
        const int s = f / 12, pair = (f % 12) / 4, within = f % 4;
        const int part = within >> 1, t = 2 * pair + (within & 1), g = t >> 1, ct = t & 1;
        const float *src = W + (size_t)(g * H + 32 * w + 16 * ct + (lane & 15)) * H + 32 * s + 8 * (lane >> 4);
        const f32x4v a = *(const f32x4v *)src, b = *(const f32x4v *)(src + 4);
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (e < 4 ? a[e] : b[e - 4]) * sc;
            const _Float16 hi = (_Float16)x;
            o[e] = part ? (_Float16)(x - (float)hi) : hi;
        }
        *(h8 *)(wp16 + (size_t)i * 8) = o;
    }
}

__device__ __forceinline__ float fast_sigmoid16(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh16(float x)
{
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.0f;
}

template <int H>
__global__ __launch_bounds__(G16<H>::NW * 64) void gru_seq16_kernel(GruParams p)
{
    using C = G16<H>;
    using P = PlanOf<H>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const GruDir d = p.dir[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * ENC_RB;

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

    const int ex = gru16_exponent(*d.wmax);
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
    float hreg[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};

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
            ring[i] = frag_load(wsrc, loff, off);
        });
    __syncthreads();

    const int H3 = 3 * H;
    int cur = 0;
    for (int s = 0; s < steps; ++s) {
        bool act[4];
        size_t tok[4];
        float giv[3][2][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            act[e] = s < len_e[e];
            const int t = d.reverse ? len_e[e] - 1 - s : s;
            tok[e] = (size_t)(off_e[e] + (act[e] ? t : 0));
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    giv[g][ct][e] = d.gi[tok[e] * H3 + g * H + unit[ct]]; // (a valid token even when the row is done)
        }
        f32x4v acc[6]; // tile t = 2 g + ct
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                acc[2 * g + ct] = (f32x4v){bias[g][ct], bias[g][ct], bias[g][ct], bias[g][ct]};

        const char *img = lds + cur * 2 * C::IMG + j * (C::LDH * 2) + kq * 16;
        // Consumption order: k-step s2, pair of column tiles, {hi t0, hi t1, lo t0, lo t1}.  Software pipeline, pinned
        // by scheduling fences (left alone, hipcc hoists every load of the step and spills): group q issues the LDS
        // reads of group q + 1 (its LDS-resident B fragments; the next k-step's A fragments one group early), then its
        // own six MFMAs, then refills the ring slots it consumed.
        h8 a_hi[2], a_lo[2]; // by k-step parity
        h8 lbuf[2][4];       // LDS-resident B fragments of the current / next group
        a_hi[0] = *(const h8 *)(img);
        a_lo[0] = *(const h8 *)(img + C::IMG);
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
                    a_hi[(s2 + 1) & 1] = *(const h8 *)(img + (s2 + 1) * 64);
                    a_lo[(s2 + 1) & 1] = *(const h8 *)(img + C::IMG + (s2 + 1) * 64);
                }
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
            constexpr int t0 = 2 * pair, t1 = 2 * pair + 1;
            acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[s2 & 1], b[0], acc[t0], 0, 0, 0);
            acc[t1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[s2 & 1], b[1], acc[t1], 0, 0, 0);
            acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[s2 & 1], b[0], acc[t0], 0, 0, 0);
            acc[t1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[s2 & 1], b[1], acc[t1], 0, 0, 0);
            acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[s2 & 1], b[2], acc[t0], 0, 0, 0);
            acc[t1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[s2 & 1], b[3], acc[t1], 0, 0, 0);
            // refill the ring slots this group consumed: the fragment NR streamed fragments further on (the next
            // step's first ones near the end of this step -- W_hh does not change)
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

        char *nimg = lds + (cur ^ 1) * 2 * C::IMG;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float r = fast_sigmoid16(giv[0][ct][e] + acc[ct][e] * down);
                const float z = fast_sigmoid16(giv[1][ct][e] + acc[2 + ct][e] * down);
                const float ghn = acc[4 + ct][e] * down;
                const float n = fast_tanh16(giv[2][ct][e] + r * ghn);
                const float hn = (hreg[ct][e] - n) * z + n;
                if (act[e]) {
                    hreg[ct][e] = hn;
                    if (d.out_seq)
                        d.out_seq[tok[e] * p.out_ld + d.out_col0 + unit[ct]] = hn;
                    if (d.gates) {
                        float *gs = d.gates + tok[e] * 4 * H + unit[ct];
                        gs[0] = r;
                        gs[H] = z;
                        gs[2 * H] = n;
                        gs[3 * H] = ghn;
                    }
                }
                const float hs = hreg[ct][e] * (float)(1 << H_SHIFT);
                const _Float16 hi = (_Float16)hs;
                const _Float16 lo = (_Float16)(hs - (float)hi);
                _Float16 *dst = (_Float16 *)nimg + (kq * 4 + e) * C::LDH + unit[ct];
                dst[0] = hi;
                dst[C::IMG / 2] = lo;
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

template <int H>
int launch16(const GruParams &gp, int ndir, hipStream_t st)
{
    using C = G16<H>;
    TT_HIP_CHECK(hipFuncSetAttribute((const void *)gru_seq16_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    hipLaunchKernelGGL(gru_seq16_kernel<H>, dim3((gp.B + ENC_RB - 1) / ENC_RB, ndir), dim3(C::NW * 64), C::LDS_BYTES, st, gp);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

} // namespace

bool gru16_supported(int H) { return H == 256 || H == 128; }

// W_hh [3H][H] fp32 -> absmax word + packed fp16 hi/lo fragments (3H*H*4 bytes: the size of the fp32 matrix)
int gru16_pack(const float *W_hh, int H, unsigned *absmax /*zeroed by the caller on the stream*/, void *wp16, hipStream_t st)
{
    hipLaunchKernelGGL(whh_absmax_kernel, dim3(48), dim3(256), 0, st, W_hh, 3 * H * H, absmax);
    hipLaunchKernelGGL(pack_whh16_kernel, dim3(96), dim3(256), 0, st, W_hh, H, (const unsigned *)absmax, (_Float16 *)wp16);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

int gru16_launch(const GruParams &gp, int ndir, hipStream_t st)
{
    if (gp.H == 256)
        return launch16<256>(gp, ndir, st);
    if (gp.H == 128)
        return launch16<128>(gp, ndir, st);
    return tt_fail(TT_ERR_UNSUPPORTED, "gru16_launch: H=%d", gp.H);
}
