// K1r: the input projections Gi = X W_ih^T + b_ih of a whole packed batch (model.py:59-62: nn.GRU's first half; X rows
// are GloVe table rows gathered by token id) as a TOKEN-STATIONARY GEMM on the f16 matrix pipes.
//
// Same arithmetic as tt_sgemm16 (sgemm.hip): both operands split into fp16 hi + lo parts after an exact power-of-two
// scaling, product = hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation.  What changes is who
// stays put.  The tiled kernel (128 x 128 output tile, both operands through LDS, two barriers per 32-wide k-tile)
// re-converts every token tile once per 128-column block (6x at N = 768) and hides one tile's worth of load latency
// behind 24 MFMAs; measured 1.45 ms for 573 k tokens x 768 x 300 where the MFMAs need 0.34 ms.  Here
//   * a workgroup owns 64 tokens and ALL N output columns: the 64 x K token block is gathered, scaled, split ONCE into
//     two fp16 LDS images (64 x 312 halves each: 78 KiB per workgroup, two workgroups per CU) and never moves again;
//   * W_ih arrives pre-split in MFMA-fragment order (tt_pack_frag16: 1 KiB per fragment, lane-linear), streamed from
//     L2 straight into registers by buffer loads through a 4-deep register ring -- no LDS, no barrier in the main loop;
//   * each of the 4 waves walks its own N/4 columns in passes of 64 (2 x 2 accumulator tiles of 32 x 32 against the
//     64 tokens), re-reading the token fragments from LDS (42 B/clk per CU) and storing C^T-oriented accumulators as
//     16-byte non-temporal stores at the end of each pass.
// One barrier per workgroup (after the fill); the second resident workgroup's fill overlaps the first one's MFMAs.
// Bounds per 573 k-token launch: MFMA 0.34 ms, L2 -> CU fragment stream 8.4 GB (0.4 ms at the ~87 GB/s per CU measured
// for gru16), output 1.76 GB (0.3 ms of HBM writes), all overlapped.
#include "sgemm.h"
#include "pack16.h"

#include <type_traits>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x16v __attribute__((ext_vector_type(16)));

#define RS_MFMA(wf, tf, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(tf, wf, c, 0, 0, 0)
constexpr int RS_ROWS = 64;                     // tokens per workgroup
// Two forms.  NARROW (K <= 304: the embedding layer, a stacked unidirectional layer): 4 waves, 2 column tiles (64 columns) per
// pass, 78 KiB of token images -> two workgroups per CU.  WIDE (K = 512: the input of a stacked BIDIRECTIONAL layer,
// config.json's default shape): the images are 130 KiB, so ONE workgroup of 8 waves per CU, one 32-column tile per wave and
// pass (24 chunks of N = 768 over 8 waves: 3 passes each; 64-column chunks would leave half the waves a pass short).
template <int NKS, bool WIDE>
struct RSCfg {
    static constexpr int WAVES = WIDE ? 8 : 4;
    static constexpr int CT = WIDE ? 1 : 2;              // 32-column tiles per pass
    static constexpr int CHUNK = 32 * CT;                // columns per (wave, pass)
    static constexpr int FRAGS = 2 * CT;                 // 1-KiB fragments per k-step: hi, lo of each tile
    static constexpr int LDH = WIDE ? NKS * 16 + 8 : 312; // halves per image row: 156 / 260 dwords = 28 / 4 mod 32 banks
    static constexpr int IMG = RS_ROWS * LDH * 2;        // bytes per image (hi | lo)
    static constexpr int LDS = 2 * IMG + RS_ROWS * 4;    // + per-token "down" factors: 80 128 B (two per CU) / 133 376 B (one per CU)
    static constexpr int TPR = WAVES * 64 / RS_ROWS;     // threads per token row in the fill
    static constexpr int NJ = (NKS * 4 + TPR - 1) / TPR; // 16-byte pieces per thread
};
#ifndef TT_ROWS_NR
#define TT_ROWS_NR 4
#endif
constexpr int RS_NR = TT_ROWS_NR;                        // ring depth in k-steps (4 fragments = 16 VGPRs each)

__device__ __forceinline__ h8 frag_load(__amdgpu_buffer_rsrc_t rsrc, int lane_off, int byte_off)
{
    return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, byte_off, 0));
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// NKS = ceil(K / 16) k-steps (compile time: the k loop is fully unrolled and software-pipelined).  The column chunks go
// round the waves: wave w takes chunks w, w + W, ..., one per PASS (runtime loop).

// The TAIL SPLIT.  A workgroup owns 64 tokens and ALL columns, and `slots` of them are resident at a time (two per CU): the
// last round of a launch is as long as any other however few token blocks it holds -- 1 098 blocks on 512 slots (the train
// step's document call) are 2.14 rounds of work in the time of 3, a serving-size query batch is one round at 10 % occupancy.
// When every wave makes the same number of passes and that number is a multiple of RS_SPLIT, the token blocks beyond the last
// full round -- if there are at most slots / RS_SPLIT of them -- are each given to RS_SPLIT workgroups that fill the same
// block and take a third of every wave's passes.  M may only be known on the device (m_dyn), so every workgroup derives its
// role from the block index itself; the host launches enough blocks for either outcome.
constexpr int RS_SPLIT = 3;

template <int NKS, bool WIDE>
__global__ __launch_bounds__(WIDE ? 512 : 256, WIDE ? 1 : 2) void gemm_rows16_kernel(SgemmParams p, int slots)
{
    using C = RSCfg<NKS, WIDE>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform: the buffer resource below must live in SGPRs
    const int M = p.m_dyn ? min(p.M, *p.m_dyn) : p.M;
    const int nchunks = p.N / C::CHUNK;
    const int nblocks = (M + RS_ROWS - 1) / RS_ROWS, nfull = slots > 0 ? nblocks / slots * slots : nblocks, ntail = nblocks - nfull;
    const bool split = slots > 0 && nchunks % (C::WAVES * RS_SPLIT) == 0 && ntail > 0 && ntail * RS_SPLIT <= slots;
    int tb = blockIdx.x, part = 0, nparts = 1; // token block, and which share of every wave's passes
    if (split && tb >= nfull) {
        tb = nfull + (blockIdx.x - nfull) / RS_SPLIT;
        part = (blockIdx.x - nfull) % RS_SPLIT;
        nparts = RS_SPLIT;
    }
    const int row0 = tb * RS_ROWS;
    if (row0 >= M)
        return;
    const int ea = p.a_absmax ? tt_pow2_exponent(*p.a_absmax) : p.a_exp;
    const int eb = p.b_absmax ? tt_pow2_exponent(*p.b_absmax) : p.b_exp;
    const int i = lane & 31, h = lane >> 5;
    float *const tdown = (float *)(lds + 2 * C::IMG); // [RS_ROWS]: 2^-(e_row + e_B), what a token's accumulators are multiplied by
    const int npass_all = rs_chunks_of(nchunks, w, C::WAVES);
    const int npass = npass_all / nparts, pass0 = part * npass; // this workgroup's passes of the wave: pass0 .. pass0 + npass - 1
    int first = pass0; // this wave's first pass in the (wave-major) fragment stream
    for (int j = 0; j < w; ++j)
        first += rs_chunks_of(nchunks, j, C::WAVES);
    // The bias (added when a pass's accumulators are scaled back) is fetched one pass ahead: a load issued in the epilogue
    // would have to be waited for with vmcnt(0), draining the fragment ring at every pass boundary.
    float bnext[C::CT], bcur[C::CT]; // [ct]: column (W pass + w) CHUNK + 32 ct + i
    auto load_bias = [&](int pass) {
#pragma unroll
        for (int ct = 0; ct < C::CT; ++ct)
            bnext[ct] = (p.bias && pass < npass) ? p.bias[((pass0 + pass) * C::WAVES + w) * C::CHUNK + 32 * ct + i] : 0.0f;
    };
    load_bias(0);

    // ---- this wave's fragment stream (FRAGS KiB per k-step: hi/lo of each column tile): ring slot of k-step s of any pass is
    // s % RS_NR, so a pass is PER = NKS rounded up to a multiple of RS_NR ring turns, the last PER - NKS of them refill-only.
    // Loads past the end of the stream (the prefetch for a pass that does not exist) return zeros: buffer bounds. ----
    constexpr int PER = (NKS + RS_NR - 1) / RS_NR * RS_NR;
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((const char *)p.b_hi16 + (size_t)first * NKS * C::FRAGS * 1024), 0, npass * NKS * C::FRAGS * 1024, 0x00020000);
    const int loff = lane * 16;
    h8 ring[RS_NR][C::FRAGS];
    static_for<0, RS_NR>([&](auto ic) {
        constexpr int k = decltype(ic)::value;
        static_for<0, C::FRAGS>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            ring[k][j] = frag_load(wsrc, loff, (k * C::FRAGS + j) * 1024);
        });
    });

    // ---- fill: thread -> (token tid / TPR, 16-byte pieces q + TPR j of its row: TPR lanes read 16 TPR contiguous bytes) ----
    {
        const int r = tid / C::TPR, q = tid % C::TPR, row = row0 + r;
        const float *src = nullptr;
        if (row < M)
            src = p.A + (size_t)(p.a_map ? (int64_t)p.a_map[row] : (int64_t)row) * p.lda;
        f32x4v v[C::NJ];
        float mx = 0.0f;
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) {
            v[j] = (f32x4v){0, 0, 0, 0};
            if (src && 4 * (q + C::TPR * j) < p.K)
                v[j] = *(const f32x4v *)(src + 4 * (q + C::TPR * j));
#pragma unroll
            for (int e = 0; e < 4; ++e)
                mx = fmaxf(mx, fabsf(v[j][e]));
        }
        // the row's largest element: its TPR threads are adjacent lanes
#pragma unroll
        for (int off = 1; off < C::TPR; off <<= 1)
            mx = fmaxf(mx, __shfl_xor(mx, off));
        const int ea_row = p.a_row_scale ? tt_pow2_exponent(__float_as_uint(mx)) : ea;
        const float sa = ldexpf(1.0f, ea_row);
        if (q == 0)
            tdown[r] = ldexpf(1.0f, -(ea_row + eb));
        if (p.a_absmax_out && part == 0) { // one atomic per wave (non-negative floats order like their bit patterns)
            float mw = mx;
#pragma unroll
            for (int off = C::TPR; off < 64; off <<= 1)
                mw = fmaxf(mw, __shfl_xor(mw, off));
            if (lane == 0)
                atomicMax(p.a_absmax_out, __float_as_uint(mw));
        }
        char *dst = lds + r * (C::LDH * 2) + q * 8;
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) {
            if (4 * (q + C::TPR * j) >= NKS * 16) // (only when NKS * 4 is not a multiple of TPR)
                continue;
            h4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = v[j][e] * sa;
                const _Float16 hv = (_Float16)x;
                hi[e] = hv;
                lo[e] = (_Float16)(x - (float)hv);
            }
            *(h4 *)(dst + j * (8 * C::TPR)) = hi;
            *(h4 *)(dst + C::IMG + j * (8 * C::TPR)) = lo;
        }
    }
    __syncthreads();

    // ---- main loop: token fragments one k-step ahead (LDS), W fragments RS_NR k-steps ahead (L2 -> registers) ----
    const char *abase = lds + i * (C::LDH * 2) + h * 16;
    f32x16v acc[C::CT][2]; // [ct][rt]
#pragma unroll
    for (int a = 0; a < C::CT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                acc[a][b][e] = 0.0f;
    h8 ahi[2][2], alo[2][2]; // [parity of the k-step][rt]
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        ahi[0][rt] = *(const h8 *)(abase + rt * 32 * (C::LDH * 2));
        alo[0][rt] = *(const h8 *)(abase + C::IMG + rt * 32 * (C::LDH * 2));
    }
    // lane <-> column, registers <-> tokens: a store instruction writes two whole 128-byte lines.  (The C^T orientation's
    // 16-byte stores put 32 bytes into each of 32 lines per instruction: the launch ran at the 1.9 TB/s those writes
    // reached, 0.9 ms of its 1.5; whole lines: 0.35 ms.  Plain stores: non-temporal ones are no faster for whole lines
    // even at 1.8 GB, and an output that fits the MALL is read back from it by the recurrence.)
    float *const cbase = p.C + (size_t)(row0 + 4 * h) * p.ldc + w * C::CHUNK + i;
    const int mrem = M - row0 - 4 * h; // token offset t of this lane's base row is stored iff t < mrem
    for (int pass = 0; pass < npass; ++pass) {
        const int sbase = pass * NKS * C::FRAGS * 1024; // byte offset of this pass in the wave's stream
#pragma unroll
        for (int ct = 0; ct < C::CT; ++ct)
            bcur[ct] = bnext[ct];
        static_for<0, PER>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            // A double buffer: k-step s of a pass reads buffer s & 1 and fetches k-step s + 1 into the other one.  The
            // fragments of the next pass's k-step 0 go into buffer 0: from the last k-step when NKS is even (it reads
            // buffer 1), from the first refill-only turn when NKS is odd (the last k-step is still reading buffer 0).
            constexpr int par = s & 1;
            if constexpr (s == NKS && (NKS & 1)) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    ahi[0][rt] = *(const h8 *)(abase + rt * 32 * (C::LDH * 2));
                    alo[0][rt] = *(const h8 *)(abase + C::IMG + rt * 32 * (C::LDH * 2));
                }
            }
            if constexpr (s < NKS) {
                if constexpr (s + 1 < NKS || !(NKS & 1)) {
                    constexpr int s1 = (s + 1) % NKS;
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        ahi[par ^ 1][rt] = *(const h8 *)(abase + rt * 32 * (C::LDH * 2) + s1 * 32);
                        alo[par ^ 1][rt] = *(const h8 *)(abase + C::IMG + rt * 32 * (C::LDH * 2) + s1 * 32);
                    }
                }
                // ring slot: [2 ct] hi of column tile ct, [2 ct + 1] lo of it
                h8(&b)[C::FRAGS] = ring[s % RS_NR];
#pragma unroll
                for (int ct = 0; ct < C::CT; ++ct)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        acc[ct][rt] = RS_MFMA(b[2 * ct], ahi[par][rt], acc[ct][rt]);
#if !(TT_MUTATE_DROP_LO & 4)
#pragma unroll
                for (int ct = 0; ct < C::CT; ++ct)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        acc[ct][rt] = RS_MFMA(b[2 * ct], alo[par][rt], acc[ct][rt]);
#pragma unroll
                for (int ct = 0; ct < C::CT; ++ct)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        acc[ct][rt] = RS_MFMA(b[2 * ct + 1], ahi[par][rt], acc[ct][rt]);
#endif
            }
            // refill the slot just consumed (or, on a refill-only turn, the slot of this turn) with the k-step RS_NR
            // turns on: k-step s + RS_NR of this pass, or k-step s + RS_NR - PER of the next one
            {
                constexpr int tgt = s + RS_NR;
                if constexpr (tgt < NKS) {
                    static_for<0, C::FRAGS>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        ring[s % RS_NR][j] = frag_load(wsrc, loff + sbase, (tgt * C::FRAGS + j) * 1024);
                    });
                } else if constexpr (tgt >= PER) {
                    static_for<0, C::FRAGS>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        ring[s % RS_NR][j] = frag_load(wsrc, loff + sbase, ((NKS + tgt - PER) * C::FRAGS + j) * 1024);
                    });
                }
            }
            if constexpr (s == 0)
                load_bias(pass + 1);
            __builtin_amdgcn_sched_barrier(0);
        });
        // end of a pass: 64 tokens x CHUNK columns out
        float *cp = cbase + (pass0 + pass) * (C::WAVES * C::CHUNK);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                // registers 4 rq .. 4 rq + 3 hold tokens 4 h + rt 32 + 8 rq + (0..3): their four down factors in one read
                const f32x4v dn = *(const f32x4v *)(tdown + 4 * h + rt * 32 + 8 * rq);
#pragma unroll
                for (int ct = 0; ct < C::CT; ++ct)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int r = 4 * rq + rr, t = rt * 32 + 8 * rq + rr;
                        const float v = fmaf(acc[ct][rt][r], dn[rr], bcur[ct]);
                        acc[ct][rt][r] = 0.0f;
                        if (t < mrem)
                            cp[(size_t)t * p.ldc + 32 * ct] = v;
                    }
            }
    }
}

// W [N][K] fp32 -> the fragment stream (pack16.h)
__global__ __launch_bounds__(256) void pack_frag16_kernel(const float *__restrict__ W, int N, int K, int nks, int waves, int ctn,
                                                          const unsigned *__restrict__ absmax, _Float16 *__restrict__ out)
{
    pack_frag16_body(W, N, K, nks, waves, ctn, absmax, out, (int)blockIdx.x, (int)gridDim.x);
}

template <int NKS, bool WIDE>
int launch_rows16(const SgemmParams &p, hipStream_t st)
{
    using C = RSCfg<NKS, WIDE>;
    static bool attr_done = false; // (idempotent: a race sets the same value twice)
    if (!attr_done) {
        TT_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_rows16_kernel<NKS, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         C::LDS));
        attr_done = true;
    }
    // resident workgroups (the kernel's rounds): two per CU for the narrow form, one for the wide one; the comparison build's TT_ROWS_SPLIT=0: no tail split
    static const int resident = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
            (void)hipGetLastError();
            cus = 256;
        }
        return (WIDE ? 1 : 2) * cus;
    }();
    const int slots = TT_AB_SWITCH(TT_ROWS_SPLIT, 1) ? resident : 0; // (tt_common.h: a constant in the product build)
    // (a split tail has at most slots / RS_SPLIT token blocks, each RS_SPLIT workgroups: (RS_SPLIT - 1) slots / RS_SPLIT extra)
    const unsigned blocks = (unsigned)((p.M + RS_ROWS - 1) / RS_ROWS) + (unsigned)(slots - slots / RS_SPLIT);
    hipLaunchKernelGGL((gemm_rows16_kernel<NKS, WIDE>), dim3(blocks), dim3(C::WAVES * 64), C::LDS, st, p, slots);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

bool rows16_wide(int K) { return (K + 15) / 16 > 19; }

} // namespace

bool tt_gemm_rows16_supported(int N, int K, int64_t lda, int64_t ldc)
{
    const int nks = (K + 15) / 16;
    if (K % 4 || lda % 4 || ldc % 4 || N < 256)
        return false;
    return rows16_wide(K) ? (nks == 32 && N % 32 == 0) : (N % 64 == 0 && (nks == 19 || nks == 16 || nks == 13));
}

int tt_pack_frag16(const float *W, int N, int K, const unsigned *absmax, void *out, hipStream_t st)
{
    const int nks = (K + 15) / 16;
    const int waves = rows16_wide(K) ? 8 : 4, ctn = rows16_wide(K) ? 1 : 2;
    const int total = N / 32 * nks * 64;
    hipLaunchKernelGGL(pack_frag16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, N, K, nks, waves, ctn, absmax,
                       (_Float16 *)out);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

int tt_gemm_rows16(const SgemmParams &p, hipStream_t st)
{
    if (p.M <= 0)
        return TT_OK;
    if (!tt_gemm_rows16_supported(p.N, p.K, p.lda, p.ldc) || !p.b_hi16 || p.accumulate || p.k_dyn || p.b_map)
        return TT_ERR_UNSUPPORTED;
    switch ((p.K + 15) / 16) {
    case 19: return launch_rows16<19, false>(p, st); // E = 300 (GloVe 6B.300d: the north-star tower)
    case 16: return launch_rows16<16, false>(p, st); // 256: a stacked layer's input
    case 13: return launch_rows16<13, false>(p, st); // E = 200 (config.json's embedding width)
    case 32: return launch_rows16<32, true>(p, st);  // 512: a stacked bidirectional layer's input (config.json's default)
    }
    return TT_ERR_UNSUPPORTED;
}
