// Workspace layout shared by the encoder forward (encoder.hip) and backward (encoder_bwd.hip).
#pragma once
#include "tt_common.h"

constexpr int ENC_MAX_LAYERS = 4;

// RNN_TYPE of the reference (getattr(nn, rnn_type.upper()), backend/model.py:30).  Gate rows of W_ih / W_hh: GRU r,z,n;
// LSTM i,f,g,o (h_n is the output, model.py:59-60); RNN one (tanh).
enum { CELL_GRU = 0, CELL_LSTM = 1, CELL_RNN = 2 };
static inline int enc_gates(int cell) { return cell == CELL_LSTM ? 4 : (cell == CELL_RNN ? 1 : 3); }
constexpr int ENC_RB = 16; // batch rows per recurrence block (one 16x16x4 MFMA M-tile)

// All offsets are bytes from the start of the caller's workspace.  Token-indexed buffers use
// the PACKED token order: row b's position t lives at tok_off[b] + t (t < len[b]); their
// capacity is the upper bound MT = B*T because the valid count is only known on the device.
struct EncLayout {
    int B, T, E, H, L, ndir, train;
    int64_t MT;
    size_t len, tok_off, perm, ids, flag;        // int32 metadata (tok_off has B+1 entries)
    size_t x[ENC_MAX_LAYERS + 1];                // x[l+1] = output sequence of layer l: [MT][ndir*H]
    size_t xd[ENC_MAX_LAYERS + 1];               // train + dropout: x[l+1] after the inter-layer dropout mask
    size_t gates[ENC_MAX_LAYERS][2];             // train: [MT][4][H] = r, z, n, W_hn h + b_hn (GRU) / i, f, g, o (LSTM)
    size_t cseq[ENC_MAX_LAYERS][2];              // train, LSTM: cell state after each token [MT+1][H] (row MT = zeros)
    size_t hfin;                                 // [ndir][B][H] final hidden of the LAST layer
    size_t hid;                                  // [B][H] head output before normalisation
    int ng;                                      // gate rows / H: 3 GRU, 4 LSTM, 1 RNN
    size_t gi[2];                                // scratch: input projections [MT][ng H] per direction
    size_t wp[2];                                // scratch: W_hh packed for the MFMA B operand
    size_t wih16[2];                             // scratch: W_ih split into fp16 hi | lo images [ng H][Kp] each
    size_t xch;                                  // column-split recurrence (gru16x4.hip): the teams' hand-off granules; 0 = none
    size_t fwd_end;
    // backward scratch (train only)
    size_t d_hfin;                               // [ndir][B][H]
    size_t d_hid;                                // [B][H] gradient w.r.t. the head output before normalisation
    size_t prevmap[2];                           // [MT] int32: packed index of the token one step earlier (MT = none)
    size_t dgi[2];                               // [MT][3H] per direction
    size_t dghn[2];                              // GRU: dGh = [dr_pre, dz_pre, dn_pre * r], [MT][3H] (the hidden-side pre-activation gradients)
    size_t dx[2];                                // ping-pong [MT][ndir*H]: gradient w.r.t. a layer's input
    size_t wtp[2];                               // W_hh packed for dh_prev = dGh * W_hh
    size_t xchb;                                 // column-split backward recurrence: hand-off granules; 0 = none
    size_t slabs;                                // split-K partial products
    size_t dx0;                                  // train == 2 (trainable table): gradient w.r.t. the gathered vectors [MT][E]
    size_t total;
};

// Word of the 256-byte status block (EncLayout::flag) that holds the bit pattern of max |x| over the embedding vectors of
// the call's valid tokens: written by the training forward (K1's fill, or tt_absmax_rows), read by the backward as the X
// scale of dW_ih = dGi^T X.  (Other words: 0 error flags, 16.. max|W_hh| forward, 32.. backward, 40.. max|W_ih|, 48..63
// max|dGi| / max|dGh|.)
constexpr int ENC_FLAG_XMAX = 8;

#ifndef TT_ENC_SPLITK
#define TT_ENC_SPLITK 64
#endif
constexpr int ENC_SPLITK = TT_ENC_SPLITK;

// The mask is a counter-based hash of (seed, layer, padded element index): no storage, the backward
// pass regenerates it.  Identical to oracle/tt_oracle.c:o_dropout_scale.
__host__ __device__ static inline float tt_dropout_scale(uint64_t seed, int layer, uint64_t idx, float p)
{
    uint64_t x = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(layer + 1);
    x ^= idx * 0xD1342543DE82EF95ull;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    const uint32_t u = (uint32_t)(x >> 32);
    const double t = (double)p * 4294967296.0;
    const uint32_t thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    return u >= thresh ? 1.0f / (1.0f - p) : 0.0f;
}

// csrc/gru16x4.hip: the GRU recurrence (H = 256) with a row group's gate columns split over four CUs; bit-identical to
// gru16_launch.  gru16x4_xch_bytes: hand-off scratch for B rows (0 when H != 256; host-only arithmetic, no GPU call).
size_t gru16x4_xch_bytes(int B, int H, int ndir);
size_t gru16x4_bwd_xch_bytes(int B, int H, int ndir);
bool gru16x4_usable(int B, int H, int ndir); // this device has a CU for every member of every team of ONE launch (gru16x4_launches > 0)
int gru16x4_launches(int B, int H, int ndir); // 1: both directions in one grid, 2: one launch per direction, 0: does not fit
bool gru16x4_bwd_usable(int B, int H, int ndir);

// projected: an inference call whose layer 0 reads its input projections from the projected table (encoder.hip): a
// one-layer model then needs no [tokens][3H] scratch at all (11 GB at 32 768 passages)
static inline EncLayout enc_layout(int B, int T, int E, int H, int L, int bidir, int train, int dropout = 0,
                                   int cell = CELL_GRU, bool projected = false)
{
    EncLayout lo;
    lo.B = B; lo.T = T; lo.E = E; lo.H = H; lo.L = L; lo.ndir = bidir ? 2 : 1; lo.train = train;
    const int ng = enc_gates(cell);
    lo.ng = ng;
    lo.MT = (int64_t)B * T;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = tt_align_up(off + bytes, 256); return o; };
    lo.len = take(sizeof(int32_t) * B);
    lo.tok_off = take(sizeof(int32_t) * (B + 1));
    lo.perm = take(sizeof(int32_t) * B);
    lo.ids = take(sizeof(int32_t) * lo.MT);
    lo.flag = take(256);
    // train: one extra, all-zero row at index MT stands for "h before the first step"
    const size_t seq = sizeof(float) * (lo.MT + (train ? 1 : 0)) * lo.ndir * H;
    lo.x[0] = 0;
    for (int l = 0; l <= ENC_MAX_LAYERS; ++l)
        lo.xd[l] = 0;
    for (int l = 0; l < L; ++l) {
        if (train && dropout && l + 1 < L)
            lo.xd[l + 1] = take(seq);
        if (train)
            lo.x[l + 1] = take(seq);
        else if (l + 1 < L)
            lo.x[l + 1] = l < 2 ? take(seq) : lo.x[l - 1]; // inference: two ping-pong buffers
        else
            lo.x[l + 1] = 0; // last layer's sequence is not needed for inference
        for (int d = 0; d < 2; ++d) {
            lo.gates[l][d] = (train && d < lo.ndir && cell != CELL_RNN) ? take(sizeof(float) * lo.MT * 4 * H) : 0;
            lo.cseq[l][d] = (train && d < lo.ndir && cell == CELL_LSTM) ? take(sizeof(float) * (lo.MT + 1) * H) : 0;
        }
    }
    lo.hfin = take(sizeof(float) * lo.ndir * B * H);
    lo.hid = take(sizeof(float) * B * H);
    for (int d = 0; d < 2; ++d) {
        lo.gi[d] = d < lo.ndir ? take((projected && !train && L == 1) ? 0 : sizeof(float) * lo.MT * ng * H) : 0;
        lo.wp[d] = d < lo.ndir ? take(sizeof(float) * ng * H * H) : 0;
        {
            const int in_w = E > lo.ndir * H ? E : lo.ndir * H;
            lo.wih16[d] = d < lo.ndir ? take((size_t)2 * sizeof(uint16_t) * ng * H * ((in_w + 31) / 32 * 32)) : 0;
        }
    }
    {
        const size_t xb = cell == CELL_GRU ? gru16x4_xch_bytes(B, H, lo.ndir) : 0;
        lo.xch = xb ? take(xb) : 0;
    }
    lo.fwd_end = off;
    lo.d_hfin = lo.d_hid = lo.slabs = lo.xchb = 0;
    for (int d = 0; d < 2; ++d)
        lo.dgi[d] = lo.dghn[d] = lo.dx[d] = lo.wtp[d] = lo.prevmap[d] = 0;
    if (train) {
        lo.d_hfin = take(sizeof(float) * lo.ndir * B * H);
        lo.d_hid = take(sizeof(float) * B * H);
        for (int d = 0; d < lo.ndir; ++d)
            lo.prevmap[d] = take(sizeof(int32_t) * lo.MT);
        for (int d = 0; d < lo.ndir; ++d) {
            lo.dgi[d] = take(sizeof(float) * lo.MT * ng * H);
            lo.dghn[d] = take(sizeof(float) * lo.MT * 3 * H);
            lo.wtp[d] = take(sizeof(float) * ng * H * H);
        }
        if (L > 1)
            for (int i = 0; i < 2; ++i)
                lo.dx[i] = take(seq);
        {
            const size_t xb = cell == CELL_GRU ? gru16x4_bwd_xch_bytes(B, H, lo.ndir) : 0;
            lo.xchb = xb ? take(xb) : 0;
        }
        const size_t in_max = (size_t)(E > lo.ndir * H ? E : lo.ndir * H);
        lo.slabs = take(sizeof(float) * ENC_SPLITK * (ng < 3 ? 3 : ng) * H * (in_max > (size_t)H ? in_max : (size_t)H));
    }
    // last, so that the layouts of train == 1 and train == 2 agree on everything before it
    lo.dx0 = train == 2 ? take(sizeof(float) * lo.MT * E) : 0;
    lo.total = off;
    return lo;
}

// Weights in the forms the forward kernels read (tt_encoder_prepare_f32): a 256-byte header of scale words
// (word 2 (2 l + d): bit pattern of max |W_ih|, word 2 (2 l + d) + 1: of max |W_hh|) and, per layer and direction, the
// W_ih images (fragment stream or [rows][Kp] hi | lo) and the packed W_hh
struct EncPrepared {
    size_t wih[ENC_MAX_LAYERS][2], wp[ENC_MAX_LAYERS][2];
    size_t total;
};
static inline EncPrepared enc_prepared_layout(int E, int H, int L, int bidir, int cell)
{
    EncPrepared pl;
    const int ng = enc_gates(cell), ndir = bidir ? 2 : 1;
    size_t off = 256;
    auto take = [&](size_t bytes) { size_t o = off; off = tt_align_up(off + bytes, 256); return o; };
    for (int l = 0; l < ENC_MAX_LAYERS; ++l)
        for (int d = 0; d < 2; ++d) {
            pl.wih[l][d] = pl.wp[l][d] = 0;
            if (l < L && d < ndir) {
                const int in_w = l == 0 ? E : ndir * H;
                pl.wih[l][d] = take((size_t)2 * sizeof(uint16_t) * ng * H * ((in_w + 31) / 32 * 32));
                pl.wp[l][d] = take(sizeof(float) * ng * H * H);
            }
        }
    pl.total = off;
    return pl;
}

// ------------------------------------------------------------------ gate nonlinearities (forward and backward kernels)
// v_exp_f32 / v_rcp_f32 based.  sigmoid has no cancellation (values near 1/2 for small x).  tanh as 2/(1+e^-2x) - 1 has
// an ABSOLUTE error of ~1e-7 (one rounding at magnitude 1), i.e. a relative error of 1e-7/|x| for small pre-activations --
// a model whose hidden state is tiny everywhere would see it amplified by F.normalize -- so |x| < 1/8 takes the odd
// polynomial x (1 - x^2/3 + 2x^4/15 - 17x^6/315), truncation error < 0.022 x^8 relative (1.3e-9 at the switch point).
#ifdef __HIPCC__
__device__ __forceinline__ float tt_fast_sigmoid(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tt_fast_tanh(float x)
{
    const float big = 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.0f;
    const float t = x * x;
    const float small = x * fmaf(t, fmaf(t, fmaf(t, -17.0f / 315.0f, 2.0f / 15.0f), -1.0f / 3.0f), 1.0f);
    return fabsf(x) < 0.125f ? small : big;
}
#endif

// ------------------------------------------------------------------ K2 recurrence: shared launch parameters
struct GruDir {
    const float *gi;      // [M][3H] packed tokens
    const float *wp;      // packed W_hh (fp32 MFMA order, or fp16 hi/lo fragments for the f16-split kernel)
    const unsigned *wmax; // f16-split kernel: bit pattern of max|W_hh| (the scale exponent derives from it)
    const float *b_hh;    // [3H]
    float *out_seq;       // nullable, [M][out_ld]
    float *gates;         // nullable, [M][4][H]
    float *cseq;          // nullable (LSTM training): [M+1][H] cell state after each token
    float *h_final;       // [B][H]
    int out_col0;
    int reverse;
};

struct GruParams {
    GruDir dir[2];
    const int32_t *len, *tok_off, *perm;
    int B, H, out_ld;
    int slots; // workgroups resident at once (2 per CU): within one such round, long row groups pair with short ones
    // Projected table (inference, layer 0; tt_encoder_project_table_f32): dir[].gi is then P[gi_rows][3H] = table W_ih^T + b_ih
    // of EVERY vocabulary row and a token's projections are row gi_ids[packed token] of it (clamped to gi_rows - 1: the packed
    // ids of a zero-length row are never written).  Null: dir[].gi is indexed by the packed token itself.  (f16-split GRU kernels only.)
    const int32_t *gi_ids = nullptr;
    unsigned gi_rows = 0;
};

struct GruBwdDir {
    const float *gates;   // [M][4][H]
    const float *cseq;    // LSTM: cell state after each token [M+1][H] (row M = zeros: c before the first step)
    const float *hseq;    // this layer's output sequence, packed [M+1][ld]
    const float *d_seq;   // nullable: gradient w.r.t. that sequence, [M][ld]
    const float *d_hfin;  // nullable: [B][H]
    const float *wtp;     // packed W_hh for dh_prev = dGh W_hh (fp32 MFMA order, or fp16 hi/lo fragments)
    const unsigned *wmax; // f16-split kernel: bit pattern of max|W_hh|
    float *dgi;           // [M][3H]
    float *dghn;          // GRU: dGh [M][3H] = [dr_pre, dz_pre, dn_pre * r]
    // f16-split kernel: per-workgroup column sums of dGi | dGh ([row groups][2][3H]: the bias gradients, reduced in fixed order
    // afterwards) and the bit patterns of max |dGi|, max |dGh| (atomicMax; zeroed by the caller): the scales of the
    // weight-gradient products -- what two column-sum passes over the 2 x 218 MB computed before
    float *bias_slab;
    unsigned *mx_dgi, *mx_dghn;
    int col0, reverse;
};

struct GruBwdParams {
    GruBwdDir dir[2];
    const int32_t *len, *tok_off, *perm;
    int B, H, ld;
    // inter-layer dropout on this layer's OUTPUT: d_seq is the gradient w.r.t. the dropped sequence
    float drop_p;
    uint64_t drop_seed;
    const uint64_t *drop_seed_ptr = nullptr; // TT_ENC_SEED_ON_DEVICE: the seed is read from here by the kernel (include/tt.h)
    int drop_layer, T;
};

// csrc/gru16.hip: the recurrence on the f16 matrix pipes (fp16 hi/lo split of both operands, fp32 accuracy)
bool gru16_supported(int H);
int gru16_pack(const float *W_hh, int H, unsigned *absmax, void *wp16, hipStream_t st);
int gru16_launch(const GruParams &gp, int ndir, hipStream_t st);
int gru16_pack_t(const float *W_hh, int H, const unsigned *absmax, void *wtp16, hipStream_t st);
int gru16_bwd_launch(const GruBwdParams &bp, int ndir, hipStream_t st);
int gru16x4_launch(const GruParams &gp, int ndir, void *xch, int32_t *status, hipStream_t st, bool xch_zeroed = false);
// the reverse-time recurrence on four CUs per row group (reduction split; deterministic, not bit-identical to gru16_bwd_launch)
int gru16x4_bwd_launch(const GruBwdParams &bp, int ndir, void *xch, int32_t *status, hipStream_t st);
