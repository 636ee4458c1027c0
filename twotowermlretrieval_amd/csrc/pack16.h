// The two weight conversions of a GRU layer as device functions, so that ONE launch can do both (encoder.hip: pack2_kernel):
// W_ih -> gemm_rows16's fragment stream, W_hh -> gru16's packed fragment order.  `bid` / `nblk`: this workgroup's index among the
// workgroups that share the job (256 threads each).
#pragma once
#include "tt_common.h"
#include "sgemm.h"

typedef _Float16 pk_h8 __attribute__((ext_vector_type(8)));
typedef float pk_f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ static inline int rs_chunks_of(int nchunks, int w, int W) { return (nchunks - w + W - 1) / W; }

// W [N][K] fp32 -> the fragment stream gemm_rows16_kernel reads: 1-KiB blocks ordered (wave, pass, k-step, column tile,
// hi | lo), lane l of a block = 8 halves of column (W pass + wave) CHUNK + 32 ct + (l & 31) at k = 16 s + 8 (l >> 5) ..
static __device__ __forceinline__ void pack_frag16_body(const float *__restrict__ W, int N, int K, int nks, int waves, int ctn,
                                                        const unsigned *__restrict__ absmax, _Float16 *__restrict__ out, int bid, int nblk)
{
    const float sc = ldexpf(1.0f, tt_pow2_exponent(*absmax));
    const int chunk_cols = 32 * ctn, nchunks = N / chunk_cols;
    const int total = nchunks * nks * ctn * 64; // (chunk, s, ct, lane)
    for (int t = bid * 256 + threadIdx.x; t < total; t += nblk * 256) {
        const int l = t & 63;
        int rest = t >> 6;
        const int ct = rest % ctn;
        rest /= ctn;
        const int s = rest % nks, chunk = rest / nks;
        const int w = chunk % waves, pass = chunk / waves;
        int first = 0;
        for (int j = 0; j < w; ++j)
            first += rs_chunks_of(nchunks, j, waves);
        const int col = chunk * chunk_cols + 32 * ct + (l & 31), k0 = 16 * s + 8 * (l >> 5);
        pk_h8 vh, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (k0 + e < K) ? W[(size_t)col * K + k0 + e] * sc : 0.0f;
            const _Float16 hv = (_Float16)x;
            vh[e] = hv;
            vl[e] = (_Float16)(x - (float)hv);
        }
        _Float16 *blk = out + ((size_t)(((first + pass) * nks + s) * ctn + ct) * 2) * 512 + l * 8;
        *(pk_h8 *)blk = vh;
        *(pk_h8 *)(blk + 512) = vl;
    }
}

// Packed order of W_hh: wave w, fragment f = (s, pair, within): s = f / 12 the k-step, pair = (f % 12) / 4 the pair of column
// tiles {2 pair, 2 pair + 1}, within = f % 4 -> part = within >> 1 (0 hi, 1 lo), tile t = 2 pair + (within & 1),
// gate g = t >> 1, ct = t & 1.  Lane (n = lane & 15, kq = lane >> 4) holds the 8 fp16 of
//   W_hh[g H + 32 w + 16 ct + n][32 s + 8 kq .. + 7]  (scaled by 2^e; hi or lo part):  16 bytes at
//   wp16[((w NF + f) 64 + lane) 8 ...].
static __device__ __forceinline__ void pack_whh16_body(const float *__restrict__ W, int H, const unsigned *__restrict__ absmax,
                                                       _Float16 *__restrict__ wp16, int bid, int nblk)
{
    const int NK = H / 32, NF = 12 * NK;
    const float sc = ldexpf(1.0f, tt_pow2_exponent(*absmax));
    const int n = (H / 32) * NF * 64; // (wave, fragment, lane) triples
    for (int i = bid * 256 + threadIdx.x; i < n; i += nblk * 256) {
        const int lane = i & 63;
        const int f = (i >> 6) % NF, w = (i >> 6) / NF;
        const int s = f / 12, pair = (f % 12) / 4, within = f % 4;
        const int part = within >> 1, t = 2 * pair + (within & 1), g = t >> 1, ct = t & 1;
        const float *src = W + (size_t)(g * H + 32 * w + 16 * ct + (lane & 15)) * H + 32 * s + 8 * (lane >> 4);
        const pk_f32x4 a = *(const pk_f32x4 *)src, b = *(const pk_f32x4 *)(src + 4);
        pk_h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (e < 4 ? a[e] : b[e - 4]) * sc;
            const _Float16 hi = (_Float16)x;
            o[e] = part ? (_Float16)(x - (float)hi) : hi;
        }
        *(pk_h8 *)(wp16 + (size_t)i * 8) = o;
    }
}
