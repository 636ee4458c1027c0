// Generic LDS-tiled fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, gfx950).
// Used by the encoder for the input projections (K1), the weight-gradient products and
// the input-gradient products (K7).  Not the product's roofline kernel: these GEMMs are
// < 10 % of an encoder step; the recurrence dominates.
#pragma once
#include "tt_common.h"

struct SgemmParams {
    // C[M,N] (+ bias[N]) (+= when accumulate) = opA(A) * opB(B)
    //   A_T == false: A is [M,K] row-major (lda), row m read from source row a_map ? a_map[m] : m
    //   A_T == true : A is [K,M] row-major (lda) (the product sums over A's ROWS), source row of
    //                 reduction index k is a_map ? a_map[k] : k
    //   B_T == false: B is [N,K] row-major (ldb)  ("weights" layout: C = A * B^T)
    //   B_T == true : B is [K,N] row-major (ldb), source row of k is b_map ? b_map[k] : k
    const float *A;
    const float *B;
    float *C;
    const float *bias;
    const int32_t *a_map;
    const int32_t *b_map;
    const int *m_dyn; // optional device scalar: effective M = min(M, *m_dyn)
    const int *k_dyn; // optional device scalar: effective K = min(K, *k_dyn)
    int M, N, K;
    int64_t lda, ldb, ldc;
    int64_t slab_stride; // split-K: slice z writes C + z*slab_stride (0 when gridDim.z == 1)
    int accumulate;      // C += result (gridDim.z must be 1)
    // tt_sgemm16 only: power-of-two operand scales applied before the fp16 hi/lo split (exact; undone on the fp32
    // accumulator).  exponent = *_absmax ? tt_pow2_exponent(*_absmax) (largest element -> [2^13, 2^14)) : *_exp
    const unsigned *a_absmax, *b_absmax; // device: bit pattern of max |element| (nullable)
    int a_exp, b_exp;
    // tt_sgemm16 with B_T == false only: B given ALREADY split (tt_pack_rows16: [N][ldb16] fp16 hi and lo images, k padded
    // with zeros to a multiple of 32, scaled by *b_absmax's power of two).  A weight matrix is then converted once per
    // call instead of once per workgroup that touches a tile of it.  Null: B is split on the fly like A.
    const void *b_hi16, *b_lo16;
    int64_t ldb16;
    // tt_gemm_rows16 only.  a_row_scale != 0: every A row (an embedding vector gathered through a_map) is scaled by ITS OWN
    // power of two (row maximum -> [2^13, 2^14)) before the split, undone on that row's accumulators: fp32-grade for tables of
    // any magnitude, and -- unlike a batch-wide scale -- a row's result does not depend on which other rows share the launch.
    // a_absmax_out (nullable): receives atomicMax of the bit pattern of max |A element| over the rows of this launch (the
    // weight-gradient product over the same rows takes its X scale from it).
    int a_row_scale = 0;
    unsigned *a_absmax_out = nullptr;
};

// exponent e with max|x| 2^e in [2^13, 2^14) (0 for an all-zero or non-finite tensor)
__host__ __device__ static inline int tt_pow2_exponent(unsigned absmax_bits)
{
    const int ex = (int)((absmax_bits >> 23) & 0xff);
    if (ex == 0 || ex == 255)
        return 0;
    int e = 13 - (ex - 127);
    e = e > 100 ? 100 : e;
    e = e < -100 ? -100 : e;
    return e;
}

// Launch: grid (ceil(N/128), ceil(M/128), splits), block 256.
int tt_sgemm(const SgemmParams &p, bool a_t, bool b_t, int splits, hipStream_t st);

// The same products on the f16 matrix pipes: both operands are split into fp16 hi + lo parts while they are staged into LDS
// and the product is taken as hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_f16 with fp32 accumulation -- every product good to
// ~3 * 2^-24 relative, i.e. one fp32 rounding, at 3/16 of the fp32-MFMA time (csrc/gru16.hip has the error argument).
// A_T = B_T = false: C = A * B^T, A rows optionally gathered through a_map; A_T = false, B_T = true: C = A * B (B [K][N]);
// ... and with both operands stored [K][rows] (C = A^T * B summed over their rows: the weight-gradient products over
// all tokens), split-K like tt_sgemm.  Same four [row][k] LDS images; only the staging differs.
int tt_sgemm16(const SgemmParams &p, bool a_t, bool b_t, int splits, hipStream_t st);

// W [N][K] fp32 -> hi / lo fp16 images [N][Kp] (Kp = K rounded up to 32, zero padded), W scaled by the power of two
// that *absmax implies (tt_pow2_exponent)
int tt_pack_rows16(const float *W, int N, int K, const unsigned *absmax, void *hi16, void *lo16, hipStream_t st);

// Token-stationary form of the same product for the input projections (csrc/gemm_rows16.hip): C = A * B^T with A's rows
// gathered through a_map, K <= 304, N a multiple of 256.  p.b_hi16 = tt_pack_frag16's fragment stream (p.b_lo16 / ldb16
// unused); a_exp / a_absmax, b_absmax, bias, m_dyn as for tt_sgemm16.
bool tt_gemm_rows16_supported(int N, int K, int64_t lda, int64_t ldc);
int tt_pack_frag16(const float *W, int N, int K, const unsigned *absmax, void *out, hipStream_t st);
int tt_gemm_rows16(const SgemmParams &p, hipStream_t st);

// The weight-gradient products C[M][N] = A^T B summed over tokens (csrc/wgrad16.hip): p as for tt_sgemm16's (a_t, b_t) = (true,
// true) form with C = the split-K slabs [z][M][N] (ldc = N, slab_stride = M N); writes nslabs slabs = tt_wgrad16_slabs(...),
// which the caller reduces (tt_slab_reduce).  tt_wgrad16_supported: M a multiple of 256 (and, in the comparison build, TT_WGRAD_TILED unset).
bool tt_wgrad16_supported(int M, int N, int64_t lda, int64_t ldb, int64_t b_rows);
int tt_wgrad16_slabs(int M, int N, int cus, int max_slabs);
int tt_wgrad16(const SgemmParams &p, int nslabs, hipStream_t st);

// bit pattern of max |x| over n floats -> *out (atomicMax; the caller zeroes *out on the stream first)
int tt_absmax(const float *x, int64_t n, unsigned *out, hipStream_t st);
int tt_absmax2(const float *x0, int64_t n0, unsigned *out0, const float *x1, int64_t n1, unsigned *out1, hipStream_t st);

// the same over the gathered rows x[map[r]][0..K), r < min(M, *m_dyn) (K, ld multiples of 4)
int tt_absmax_rows(const float *x, int64_t ld, int K, const int32_t *map, int M, const int *m_dyn, unsigned *out, hipStream_t st);

// out[i] (+)= sum_z slabs[z][i], fixed order (deterministic split-K reduction)
int tt_slab_reduce(const float *slabs, int nslab, int64_t n, float *out, int accumulate, hipStream_t st);
