#include "sgemm.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDT = 20; // LDS row stride 20 floats: conflict-free b128 reads

// One operand tile [128 rows][BK] (k contiguous in LDS), staged in two halves so the global loads of
// tile t+1 are in flight while tile t is multiplied: tile_fetch() -> registers, tile_commit() -> LDS.
//  TR == false: source [rows][K]: thread -> (row t>>1, 8 consecutive k)
//  TR == true : source [K][rows]: thread -> (k t>>4, 8 consecutive rows), scattered into LDS
template <bool TR>
__device__ __forceinline__ void tile_fetch(f32x4 &v0, f32x4 &v1, const float *__restrict__ src, int64_t ld,
                                           const int32_t *__restrict__ map, int row0, int rows_eff, int k0, int k_end,
                                           int tid)
{
    v0 = (f32x4){0, 0, 0, 0};
    v1 = (f32x4){0, 0, 0, 0};
    if (!TR) {
        const int r = tid >> 1, kc = (tid & 1) * 8;
        const int row = row0 + r;
        if (row < rows_eff) {
            const int64_t srow = map ? (int64_t)map[row] : (int64_t)row;
            const float *p = src + srow * ld + k0 + kc;
            if (k0 + kc < k_end)
                v0 = *(const f32x4 *)p;
            if (k0 + kc + 4 < k_end)
                v1 = *(const f32x4 *)(p + 4);
        }
    } else {
        const int kk = tid >> 4, rc = (tid & 15) * 8;
        const int k = k0 + kk;
        if (k < k_end) {
            const int64_t srow = map ? (int64_t)map[k] : (int64_t)k;
            const float *p = src + srow * ld + row0 + rc;
            if (row0 + rc < rows_eff)
                v0 = *(const f32x4 *)p;
            if (row0 + rc + 4 < rows_eff)
                v1 = *(const f32x4 *)(p + 4);
        }
    }
}

template <bool TR>
__device__ __forceinline__ void tile_commit(float *__restrict__ tile, const f32x4 &v0, const f32x4 &v1, int tid)
{
    if (!TR) {
        const int r = tid >> 1, kc = (tid & 1) * 8;
        *(f32x4 *)(tile + r * LDT + kc) = v0;
        *(f32x4 *)(tile + r * LDT + kc + 4) = v1;
    } else {
        const int kk = tid >> 4, rc = (tid & 15) * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tile[(rc + e) * LDT + kk] = v0[e];
            tile[(rc + 4 + e) * LDT + kk] = v1[e];
        }
    }
}

template <bool A_T, bool B_T>
__global__ __launch_bounds__(256) void sgemm_kernel(SgemmParams p)
{
    __shared__ __attribute__((aligned(16))) float As[BM * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = p.m_dyn ? min(p.M, *p.m_dyn) : p.M;
    const int K = p.k_dyn ? min(p.K, *p.k_dyn) : p.K;
    // Workgroup -> tile.  Consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2; the
    // n-blocks of one m-block re-read the same A rows (for K1: rows gathered from a 480 MB embedding table), so
    // they are placed on ONE XCD, back to back: flat id L -> xcd = L % 8, slot = L / 8; the slots of an XCD walk
    // (m-block, n-block) with n fastest and m-blocks dealt out 8 apart.
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int nbx = gridDim.x, nby = gridDim.y;
        const int L = by * nbx + bx, xcd = L & 7, slot = L >> 3;
        const int mb = (slot / nbx) * 8 + xcd, nb = slot % nbx;
        const int full = (nby / 8) * 8; // m-blocks beyond the last full group of 8 keep the plain mapping
        if (by < full && mb < full) {
            by = mb;
            bx = nb;
        }
    }
    const int m0 = by * BM, n0 = bx * BN;
    if (m0 >= M)
        return;
    // split-K range (multiples of BK)
    const int splits = gridDim.z;
    int kchunk = (K + splits - 1) / splits;
    kchunk = (kchunk + BK - 1) / BK * BK;
    const int kb = blockIdx.z * kchunk, ke = min(kb + kchunk, K);
    float *C = p.C + (size_t)blockIdx.z * p.slab_stride;

    const int wr = wave >> 1, wc = wave & 1, i = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[a][b][r] = 0.0f;

    f32x4 ra0, ra1, rb0, rb1;
    tile_fetch<A_T>(ra0, ra1, p.A, p.lda, p.a_map, m0, M, kb, ke, tid);
    tile_fetch<B_T>(rb0, rb1, p.B, p.ldb, p.b_map, n0, p.N, kb, ke, tid);
    for (int k0 = kb; k0 < ke; k0 += BK) {
        tile_commit<A_T>(As, ra0, ra1, tid);
        tile_commit<B_T>(Bs, rb0, rb1, tid);
        __syncthreads();
        if (k0 + BK < ke) { // next tile's global loads fly under this tile's MFMAs
            tile_fetch<A_T>(ra0, ra1, p.A, p.lda, p.a_map, m0, M, k0 + BK, ke, tid);
            tile_fetch<B_T>(rb0, rb1, p.B, p.ldb, p.b_map, n0, p.N, k0 + BK, ke, tid);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = *(const f32x4 *)(As + (64 * wr + 32 * t + i) * LDT + 8 * c + 4 * h);
                b[t] = *(const f32x4 *)(Bs + (64 * wc + 32 * t + i) * LDT + 8 * c + 4 * h);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][e], b[nt][e], acc[mt][nt], 0, 0, 0);
        }
        __syncthreads();
    }

#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = n0 + 64 * wc + 32 * nt + i;
            if (col >= p.N)
                continue;
            const float bv = (p.bias && blockIdx.z == 0) ? p.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 64 * wr + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < M) {
                    float *dst = C + (size_t)row * p.ldc + col;
                    const float v = acc[mt][nt][r] + bv;
                    *dst = p.accumulate ? *dst + v : v;
                }
            }
        }
}

// ------------------------------------------------------------------ C = A * B^T on the f16 pipes, fp32-grade
// 128 x 128 x 32 tiles, 4 waves, 2 x 2 accumulators of 32 x 32 per wave (v_mfma_f32_32x32x16_f16).  The fp32 operands
// are fetched as in sgemm_kernel (16 consecutive k per thread, next tile in flight under the MFMAs), scaled by their
// power of two, split into fp16 hi / lo and committed to FOUR LDS images of 64-byte rows (32 KB in all: five workgroups per
// CU -- the kernel hides its gather latency with occupancy).  The four 16-byte chunks of a row are XOR-swizzled with
// f(row) = ((row >> 2) + (row >> 1)) & 3: the ds_read_b128 fragment reads of a 16-lane group fall into 16 different
// bank groups, the ds_write_b128 of rows r and r + 2 (same bank base) into different chunks, and the transposed
// staging's packed dword writes stay 2-way (free for ds_write_b32).
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
constexpr int BK16 = 32, LDT16 = 32;
__device__ __forceinline__ int swz16(int row) { return ((row >> 2) + (row >> 1)) & 3; }

//  TR == false: source [rows][K]: thread -> (row t>>1, 16 consecutive k)
//  TR == true : source [K][rows] (weight-gradient products: K = tokens): thread -> (token pair p = t & 15, row quads
//               4c .. 4c+3 and 64 + 4c .. 64 + 4c + 3, c = t >> 4): a wave's four c values read 64 contiguous bytes of
//               each token row; commit packs the two tokens of a row into one dword per image (2-way LDS conflicts,
//               which ds_write_b32 absorbs) -- the [row][k] images are the same as for TR == false.
template <bool TR>
__device__ __forceinline__ void fetch16(f32x4 (&v)[4], const float *__restrict__ src, int64_t ld,
                                        const int32_t *__restrict__ map, int row0, int rows_eff, int k0, int k_end, int tid)
{
#pragma unroll
    for (int q = 0; q < 4; ++q)
        v[q] = (f32x4){0, 0, 0, 0};
    if (!TR) {
        const int r = tid >> 1, kc = (tid & 1) * 16;
        const int row = row0 + r;
        if (row < rows_eff) {
            const int64_t srow = map ? (int64_t)map[row] : (int64_t)row;
            const float *p = src + srow * ld + k0 + kc;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (k0 + kc + 4 * q < k_end)
                    v[q] = *(const f32x4 *)(p + 4 * q);
        }
    } else {
        const int pp = tid & 15, c = tid >> 4;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k = k0 + 2 * pp + t;
            if (k < k_end) {
                const int64_t srow = map ? (int64_t)map[k] : (int64_t)k;
#pragma unroll
                for (int g = 0; g < 2; ++g) // v[2 g + t]: rows 64 g + 4 c .. + 3 of token k
                    if (row0 + 64 * g + 4 * c < rows_eff)
                        v[2 * g + t] = *(const f32x4 *)(src + srow * ld + row0 + 64 * g + 4 * c);
            }
        }
    }
}

template <bool TR>
__device__ __forceinline__ void commit16(_Float16 *__restrict__ hi_img, _Float16 *__restrict__ lo_img, const f32x4 (&v)[4],
                                         float scale, int tid)
{
    if (!TR) {
        const int r = tid >> 1;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            h8v hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = v[2 * half + (e >> 2)][e & 3] * scale;
                const _Float16 hv = (_Float16)x;
                hi[e] = hv;
                lo[e] = (_Float16)(x - (float)hv);
            }
            const int o = r * LDT16 + ((((tid & 1) * 2 + half) ^ swz16(r)) << 3);
            *(h8v *)(hi_img + o) = hi;
            *(h8v *)(lo_img + o) = lo;
        }
    } else {
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        const int pp = tid & 15, c = tid >> 4;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h2v hi, lo;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float x = v[2 * g + t][e] * scale;
                    const _Float16 hv = (_Float16)x;
                    hi[t] = hv;
                    lo[t] = (_Float16)(x - (float)hv);
                }
                const int row = 64 * g + 4 * c + e;
                const int o = row * LDT16 + ((((pp >> 2) ^ swz16(row)) << 3) | ((2 * pp) & 7));
                *(h2v *)(hi_img + o) = hi;
                *(h2v *)(lo_img + o) = lo;
            }
    }
}

// B operand already split (tt_pack_rows16): 16 halves of hi and of lo per thread, straight into the images
__device__ __forceinline__ void fetch16_pre(h8v (&v)[4], const _Float16 *__restrict__ hi, const _Float16 *__restrict__ lo,
                                            int64_t ld, int row0, int rows_eff, int k0, int tid)
{
    const int r = tid >> 1, kc = (tid & 1) * 16;
    const int row = row0 + r;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        v[q] = (h8v){0, 0, 0, 0, 0, 0, 0, 0};
    if (row < rows_eff) {
        const _Float16 *ph = hi + (int64_t)row * ld + k0 + kc, *pl = lo + (int64_t)row * ld + k0 + kc;
        v[0] = *(const h8v *)ph;
        v[1] = *(const h8v *)(ph + 8);
        v[2] = *(const h8v *)pl;
        v[3] = *(const h8v *)(pl + 8);
    }
}
__device__ __forceinline__ void commit16_pre(_Float16 *__restrict__ hi_img, _Float16 *__restrict__ lo_img, const h8v (&v)[4],
                                             int tid)
{
    const int r = tid >> 1;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int o = r * LDT16 + ((((tid & 1) * 2 + half) ^ swz16(r)) << 3);
        *(h8v *)(hi_img + o) = v[half];
        *(h8v *)(lo_img + o) = v[2 + half];
    }
}

template <bool A_T, bool B_T, bool B_PRE = false>
__global__ __launch_bounds__(256) void sgemm16_kernel(SgemmParams p)
{
    __shared__ __attribute__((aligned(16))) _Float16 img[4][BM * LDT16]; // A hi, A lo, B hi, B lo
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = p.m_dyn ? min(p.M, *p.m_dyn) : p.M;
    const int K = p.k_dyn ? min(p.K, *p.k_dyn) : p.K;
    int bx = blockIdx.x, by = blockIdx.y; // the n-blocks of an m-block on one XCD (see sgemm_kernel)
    {
        const int nbx = gridDim.x, nby = gridDim.y;
        const int L = by * nbx + bx, xcd = L & 7, slot = L >> 3;
        const int mb = (slot / nbx) * 8 + xcd, nb = slot % nbx;
        const int full = (nby / 8) * 8;
        if (by < full && mb < full) {
            by = mb;
            bx = nb;
        }
    }
    const int m0 = by * BM, n0 = bx * BN;
    if (m0 >= M)
        return;
    // split-K range (multiples of BK16)
    const int splits = gridDim.z;
    int kchunk = (K + splits - 1) / splits;
    kchunk = (kchunk + BK16 - 1) / BK16 * BK16;
    const int kb = blockIdx.z * kchunk, ke = min(kb + kchunk, K);
    float *C = p.C + (size_t)blockIdx.z * p.slab_stride;
    const int ea = p.a_absmax ? tt_pow2_exponent(*p.a_absmax) : p.a_exp;
    const int eb = p.b_absmax ? tt_pow2_exponent(*p.b_absmax) : p.b_exp;
    const float sa = ldexpf(1.0f, ea), sb = ldexpf(1.0f, eb), down = ldexpf(1.0f, -(ea + eb));

    const int wr = wave >> 1, wc = wave & 1, i = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[a][b][r] = 0.0f;

    f32x4 ra[4], rb[4];
    h8v rbp[4];
    const _Float16 *bhi = (const _Float16 *)p.b_hi16, *blo = (const _Float16 *)p.b_lo16;
    fetch16<A_T>(ra, p.A, p.lda, p.a_map, m0, M, kb, ke, tid);
    if constexpr (B_PRE)
        fetch16_pre(rbp, bhi, blo, p.ldb16, n0, p.N, kb, tid);
    else
        fetch16<B_T>(rb, p.B, p.ldb, p.b_map, n0, p.N, kb, ke, tid);
    for (int k0 = kb; k0 < ke; k0 += BK16) {
        commit16<A_T>(img[0], img[1], ra, sa, tid);
        if constexpr (B_PRE)
            commit16_pre(img[2], img[3], rbp, tid);
        else
            commit16<B_T>(img[2], img[3], rb, sb, tid);
        __syncthreads();
        if (k0 + BK16 < ke) { // next tile's global loads fly under this tile's MFMAs
            fetch16<A_T>(ra, p.A, p.lda, p.a_map, m0, M, k0 + BK16, ke, tid);
            if constexpr (B_PRE)
                fetch16_pre(rbp, bhi, blo, p.ldb16, n0, p.N, k0 + BK16, tid);
            else
                fetch16<B_T>(rb, p.B, p.ldb, p.b_map, n0, p.N, k0 + BK16, ke, tid);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8v ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ar = 64 * wr + 32 * t + i, br = 64 * wc + 32 * t + i;
                const int ao = ar * LDT16 + (((2 * ks + h) ^ swz16(ar)) << 3);
                const int bo = br * LDT16 + (((2 * ks + h) ^ swz16(br)) << 3);
                ah[t] = *(const h8v *)(img[0] + ao);
                al[t] = *(const h8v *)(img[1] + ao);
                bh[t] = *(const h8v *)(img[2] + bo);
                bl[t] = *(const h8v *)(img[3] + bo);
            }
            // The B-image fragment is the MFMA's A operand and vice versa: the accumulator tile is C^T (rows = output
            // columns, columns = output rows), so a lane holds FOUR CONSECUTIVE output columns of one output row in
            // registers 4g .. 4g+3 and the epilogue stores 16 bytes at a time (with the natural orientation it issued 64
            // dword stores per lane -- 1.8 GB of input projections went out 256 bytes per instruction).
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nt], ah[mt], acc[mt][nt], 0, 0, 0);
#if !(TT_MUTATE_DROP_LO & 8)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nt], al[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[nt], ah[mt], acc[mt][nt], 0, 0, 0);
#endif
        }
        __syncthreads();
    }

#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = m0 + 64 * wr + 32 * mt + i; // lane i <-> output row; registers 4g .. 4g+3 <-> 4 consecutive columns
        if (row >= M)
            continue;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int col = n0 + 64 * wc + 32 * nt + 8 * g4 + 4 * h;
                if (col >= p.N) // N is a multiple of 4
                    continue;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = acc[mt][nt][4 * g4 + e] * down;
                if (p.bias && blockIdx.z == 0)
                    v += *(const f32x4 *)(p.bias + col);
                f32x4 *dst = (f32x4 *)(C + (size_t)row * p.ldc + col);
                if (p.accumulate)
                    *dst = *dst + v;
                else // write-once output (1.8 GB of input projections per index-build call): keep it from evicting the
                    __builtin_nontemporal_store(v, dst); // gathered A rows the other n-blocks of this m-block re-read from L2

            }
    }
}

// W [N][K] fp32 -> hi / lo [N][Kp] fp16 (zero padded), one thread per 8 consecutive k
__global__ __launch_bounds__(256) void pack_rows16_kernel(const float *__restrict__ W, int N, int K, int Kp,
                                                          const unsigned *__restrict__ absmax, _Float16 *__restrict__ hi,
                                                          _Float16 *__restrict__ lo)
{
    const float sc = ldexpf(1.0f, tt_pow2_exponent(*absmax));
    const int per_row = Kp / 8;
    const int64_t total = (int64_t)N * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int row = (int)(i / per_row), k0 = (int)(i % per_row) * 8;
        h8v vh, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (k0 + e < K) ? W[(size_t)row * K + k0 + e] * sc : 0.0f;
            const _Float16 hv = (_Float16)x;
            vh[e] = hv;
            vl[e] = (_Float16)(x - (float)hv);
        }
        *(h8v *)(hi + (size_t)row * Kp + k0) = vh;
        *(h8v *)(lo + (size_t)row * Kp + k0) = vl;
    }
}

// max|x| into *out (atomicMax on the bit pattern: non-negative floats order like their bits).  ONE atomic per workgroup and
// at most 64 workgroups: a thousand same-address atomics serialise in L2 (the 230 k-element weight matrices took 13-14 us
// with one atomic per wave of 256 workgroups, 4 us of it the reads).
static __device__ __forceinline__ void absmax_body(const float *__restrict__ x, int64_t n, unsigned *__restrict__ out)
{
    __shared__ float part[4];
    float m = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    if ((((uintptr_t)x) & 15) == 0) {
        const int64_t n4 = n >> 2;
        const f32x4 *x4 = (const f32x4 *)x;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            const f32x4 v = x4[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
            m = fmaxf(m, fabsf(x[i]));
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
            m = fmaxf(m, fabsf(x[i]));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(out, __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));
}

__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, int64_t n, unsigned *__restrict__ out)
{
    absmax_body(x, n, out);
}
__global__ __launch_bounds__(256) void absmax2_kernel(const float *__restrict__ x0, int64_t n0, unsigned *__restrict__ out0,
                                                      const float *__restrict__ x1, int64_t n1, unsigned *__restrict__ out1)
{
    if (blockIdx.y == 0)
        absmax_body(x0, n0, out0);
    else
        absmax_body(x1, n1, out1);
}

// the same over the rows map[0 .. min(M, *m_dyn)) of a [.][ld] table, K (a multiple of 4) elements each: the embedding vectors
// of a packed batch (shapes the token-stationary K1, which finds this maximum on the way, does not take)
__global__ __launch_bounds__(256) void absmax_rows_kernel(const float *__restrict__ x, int64_t ld, int K, const int32_t *__restrict__ map,
                                                          int M, const int *__restrict__ m_dyn, unsigned *__restrict__ out)
{
    __shared__ float part[4];
    const int m_eff = m_dyn ? min(M, *m_dyn) : M;
    const int kq = K >> 2;
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)m_eff * kq; i += (int64_t)gridDim.x * 256) {
        const int row = (int)(i / kq), c = (int)(i % kq);
        const f32x4 v = *(const f32x4 *)(x + (size_t)map[row] * ld + 4 * c);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(out, __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));
}

// four consecutive elements per thread (n % 4 == 0, 16-byte aligned slabs / out): 16-byte loads, 8 slabs in flight
__global__ __launch_bounds__(256) void slab_reduce4_kernel(const float *__restrict__ slabs, int nslab, int64_t n4,
                                                           float *out, int accumulate)
{
    const f32x4 *src = (const f32x4 *)slabs;
    f32x4 *dst = (f32x4 *)out;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 8
        for (int z = 0; z < nslab; ++z) {
            const f32x4 v = src[(size_t)z * n4 + i];
            s.x += v.x;
            s.y += v.y;
            s.z += v.z;
            s.w += v.w;
        }
        if (accumulate) {
            const f32x4 o = dst[i];
            s.x += o.x;
            s.y += o.y;
            s.z += o.z;
            s.w += o.w;
        }
        dst[i] = s;
    }
}

__global__ void slab_reduce_kernel(const float *__restrict__ slabs, int nslab, int64_t n, float *out, int accumulate)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.0f;
        for (int z = 0; z < nslab; ++z)
            s += slabs[(size_t)z * n + i];
        out[i] = accumulate ? out[i] + s : s;
    }
}

// The same sum for SHORT vectors and many slabs (bias gradients: n = G*H columns, 256 row slices): one thread per
// element would walk 256 dependent loads in a handful of workgroups (60-100 us for 768 elements).  Here a workgroup
// takes 32 elements x 8 slab groups; every group sums its slabs in order, the 8 partial sums are combined in a fixed
// order through LDS: still deterministic, ~30 loads deep.
__global__ __launch_bounds__(256) void slab_reduce_wide_kernel(const float *__restrict__ slabs, int nslab, int n,
                                                               float *out, int accumulate)
{
    __shared__ float part[8][32];
    const int col = blockIdx.x * 32 + (threadIdx.x & 31), zg = threadIdx.x >> 5;
    const int per = (nslab + 7) / 8;
    float s = 0.0f;
    if (col < n)
        for (int z = zg * per; z < min(zg * per + per, nslab); ++z)
            s += slabs[(size_t)z * n + col];
    part[zg][threadIdx.x & 31] = s;
    __syncthreads();
    if (zg == 0 && col < n) {
        float t = part[0][threadIdx.x];
#pragma unroll
        for (int g = 1; g < 8; ++g)
            t += part[g][threadIdx.x];
        out[col] = accumulate ? out[col] + t : t;
    }
}

} // namespace

int tt_sgemm(const SgemmParams &p, bool a_t, bool b_t, int splits, hipStream_t st)
{
    if (p.M <= 0 || p.N <= 0 || p.K <= 0)
        return TT_OK;
    // vector loads run along K for a non-transposed operand and along M / N for a transposed one
    if (((!a_t || !b_t) && (p.K & 3)) || (a_t && (p.M & 3)) || (b_t && (p.N & 3)) || (p.lda & 3) || (p.ldb & 3))
        return tt_fail(TT_ERR_UNSUPPORTED, "tt_sgemm: dims must be multiples of 4 (M=%d N=%d K=%d)", p.M, p.N, p.K);
    if (splits < 1)
        splits = 1;
    if (splits > 1 && p.accumulate)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_sgemm: accumulate with split-K");
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, splits);
    if (!a_t && !b_t)
        hipLaunchKernelGGL((sgemm_kernel<false, false>), grid, dim3(256), 0, st, p);
    else if (!a_t && b_t)
        hipLaunchKernelGGL((sgemm_kernel<false, true>), grid, dim3(256), 0, st, p);
    else if (a_t && !b_t)
        hipLaunchKernelGGL((sgemm_kernel<true, false>), grid, dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((sgemm_kernel<true, true>), grid, dim3(256), 0, st, p);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

int tt_sgemm16(const SgemmParams &p, bool a_t, bool b_t, int splits, hipStream_t st)
{
    if (p.M <= 0 || p.N <= 0 || p.K <= 0)
        return TT_OK;
    if (a_t && !b_t)
        return tt_fail(TT_ERR_UNSUPPORTED, "tt_sgemm16: A^T * B^T is not built");
    if ((!a_t && (p.K & 3)) || (a_t && (p.M & 3)) || (p.N & 3) || (p.lda & 3) || (p.ldb & 3) || (p.ldc & 3) ||
        ((uintptr_t)p.C & 15) || (p.bias && ((uintptr_t)p.bias & 15)))
        return tt_fail(TT_ERR_UNSUPPORTED, "tt_sgemm16: dims / leading dimensions must be multiples of 4 and C, bias 16-byte "
                                           "aligned (M=%d N=%d K=%d)", p.M, p.N, p.K);
    if (splits < 1)
        splits = 1;
    if (splits > 1 && p.accumulate)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_sgemm16: accumulate with split-K");
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, splits);
    if (!a_t && p.b_hi16 && p.b_lo16) {
        if (!p.b_absmax || (p.ldb16 & 31) || p.ldb16 < p.K)
            return tt_fail(TT_ERR_BAD_SHAPE, "tt_sgemm16: pre-split B needs b_absmax and ldb16 >= K, a multiple of 32");
        hipLaunchKernelGGL((sgemm16_kernel<false, false, true>), grid, dim3(256), 0, st, p);
    } else if (!a_t && b_t) // C = A * B with B stored [K][N] (the input-gradient product dX = dGi * W_ih)
        hipLaunchKernelGGL((sgemm16_kernel<false, true>), grid, dim3(256), 0, st, p);
    else if (!a_t)
        hipLaunchKernelGGL((sgemm16_kernel<false, false>), grid, dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((sgemm16_kernel<true, true>), grid, dim3(256), 0, st, p);
    TT_LAUNCH_CHECK();
    return TT_OK;
}


int tt_pack_rows16(const float *W, int N, int K, const unsigned *absmax, void *hi16, void *lo16, hipStream_t st)
{
    if (N <= 0 || K <= 0)
        return TT_OK;
    const int Kp = (K + 31) / 32 * 32;
    const int64_t want = ((int64_t)N * (Kp / 8) + 255) / 256;
    hipLaunchKernelGGL(pack_rows16_kernel, dim3((unsigned)(want > 1024 ? 1024 : want)), dim3(256), 0, st, W, N, K, Kp, absmax,
                       (_Float16 *)hi16, (_Float16 *)lo16);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

// two tensors in one launch (blockIdx.y picks): the two weight matrices of a GRU layer
int tt_absmax2(const float *x0, int64_t n0, unsigned *out0, const float *x1, int64_t n1, unsigned *out1, hipStream_t st)
{
    if (n0 <= 0 || n1 <= 0)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_absmax2: n0=%lld n1=%lld", (long long)n0, (long long)n1);
    const int64_t big = n0 > n1 ? n0 : n1, want = (big + 1023) / 1024;
    hipLaunchKernelGGL(absmax2_kernel, dim3((unsigned)(want > 64 ? 64 : want), 2), dim3(256), 0, st, x0, n0, out0, x1, n1, out1);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

int tt_absmax(const float *x, int64_t n, unsigned *out, hipStream_t st)
{
    if (n <= 0)
        return TT_OK;
    const int64_t want = (n + 1023) / 1024; // one 16-byte load per thread and pass
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(want > 64 ? 64 : want)), dim3(256), 0, st, x, n, out);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

int tt_absmax_rows(const float *x, int64_t ld, int K, const int32_t *map, int M, const int *m_dyn, unsigned *out, hipStream_t st)
{
    if (M <= 0 || K <= 0 || (K & 3) || (ld & 3))
        return TT_OK;
    const int64_t want = ((int64_t)M * (K >> 2) + 1023) / 1024;
    hipLaunchKernelGGL(absmax_rows_kernel, dim3((unsigned)(want > 64 ? 64 : (want < 1 ? 1 : want))), dim3(256), 0, st, x, ld, K, map, M,
                       m_dyn, out);
    TT_LAUNCH_CHECK();
    return TT_OK;
}

int tt_slab_reduce(const float *slabs, int nslab, int64_t n, float *out, int accumulate, hipStream_t st)
{
    if (n <= 0)
        return TT_OK;
    if (n <= 8192 && nslab >= 64) {
        hipLaunchKernelGGL(slab_reduce_wide_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, slabs, nslab, (int)n,
                           out, accumulate);
        TT_LAUNCH_CHECK();
        return TT_OK;
    }
    if ((n & 3) == 0 && ((((uintptr_t)slabs) | ((uintptr_t)out)) & 15) == 0 && n >= 65536) { // (same sums, same order)
        int blocks4 = (int)((n / 4 + 255) / 256);
        hipLaunchKernelGGL(slab_reduce4_kernel, dim3(blocks4 > 2048 ? 2048 : blocks4), dim3(256), 0, st, slabs, nslab, n / 4,
                           out, accumulate);
        TT_LAUNCH_CHECK();
        return TT_OK;
    }
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048)
        blocks = 2048;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, slabs, nslab, n, out, accumulate);
    TT_LAUNCH_CHECK();
    return TT_OK;
}
