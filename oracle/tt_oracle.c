/*
 * tt_oracle.c -- CPU restatement of the two-tower retrieval hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product path
 * (twotowermlretrieval_amd/) never links, imports or falls back to it.
 *
 * Parity pin: the reference project ships no tests or golden vectors for
 * this path, so this restatement is pinned against
 * outputs of the reference itself, generated in the dev container by
 * importing /root/reference/backend/model.py (tests/golden/gen_golden.py)
 * and committed as tests/golden/ (npz files).  tests/test_oracle_golden.py checks
 * every function here against those vectors.
 *
 * Each function cites the reference file:line whose arithmetic it restates
 * (paths relative to the reference repository root).  All arithmetic is
 * IEEE fp32 with a DEFINED operation order, so results are reproducible.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define O_OK 0
#define O_ERR_BAD_SHAPE 1
#define O_ERR_BAD_INDEX 2
#define O_ERR_ZERO_LENGTH 3
#define O_ERR_NOMEM 4

/* ------------------------------------------------------------------ */
/* small helpers                                                      */
/* ------------------------------------------------------------------ */
static inline float o_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

/* y[n] = b[n] + sum_k x[k] * W[n*K + k]   (W row-major [N,K]; b may be NULL) */
static void o_affine(const float *x, const float *W, const float *b, int N,
                     int K, float *y)
{
    for (int n = 0; n < N; ++n) {
        float acc = b ? b[n] : 0.0f;
        const float *w = W + (size_t)n * K;
        for (int k = 0; k < K; ++k)
            acc += x[k] * w[k];
        y[n] = acc;
    }
}

/*
 * Sequence lengths.  backend/model.py:52  lengths = (x != 0).sum(dim=1)
 * A length is the COUNT of non-zero ids, not the position of the last one:
 * an id 0 inside a sentence shortens the row.
 * Rows of length 0 make pack_padded_sequence raise (model.py:55-57):
 * returned as O_ERR_ZERO_LENGTH.
 */
int o_lengths(const int64_t *ids, int B, int T, int32_t *len)
{
    if (B <= 0 || T <= 0)
        return O_ERR_BAD_SHAPE;
    int rc = O_OK;
    for (int b = 0; b < B; ++b) {
        int c = 0;
        for (int t = 0; t < T; ++t)
            c += ids[(size_t)b * T + t] != 0;
        len[b] = c;
        if (c == 0)
            rc = O_ERR_ZERO_LENGTH;
    }
    return rc;
}

/* Embedding row gather.  backend/model.py:49  embedded = self.embedding(x)
 * Row 0 is gathered like any other row (it is the GloVe word "the", not a
 * zero vector, once pretrained embeddings are copied in: model.py:25-27). */
int o_gather(const int64_t *ids, int B, int T, const float *table, int64_t V,
             int E, float *out)
{
    for (size_t i = 0; i < (size_t)B * T; ++i) {
        int64_t id = ids[i];
        if (id < 0 || id >= V)
            return O_ERR_BAD_INDEX;
        memcpy(out + i * E, table + (size_t)id * E, sizeof(float) * E);
    }
    return O_OK;
}

/*
 * One GRU direction over a padded batch with per-row lengths
 * (torch.nn.GRU on a PackedSequence; backend/model.py:31-37,55-62).
 * Gate order r,z,n; b_hn sits inside the r*(.) term:
 *   r = s(W_ir x + b_ir + W_hr h + b_hr)
 *   z = s(W_iz x + b_iz + W_hz h + b_hz)
 *   n = tanh(W_in x + b_in + r * (W_hn h + b_hn))
 *   h' = (h - n) * z + n
 * Row b runs over positions 0..len[b]-1 (reverse: len[b]-1..0) and then
 * stops updating.  out_seq (nullable) [B,T,H] gets h after each visited
 * position, zeros at padded positions (pad_packed_sequence semantics).
 * stash (nullable) [B,T,5,H] receives r,z,n,ghn(=W_hn h+b_hn),h_prev for
 * the backward pass.
 */
int o_gru_layer(const float *x, int B, int T, int I, const int32_t *len,
                const float *W_ih, const float *W_hh, const float *b_ih,
                const float *b_hh, int H, int reverse, float *out_seq,
                float *h_final, float *stash)
{
    float *gi = (float *)malloc(sizeof(float) * 3 * H);
    float *gh = (float *)malloc(sizeof(float) * 3 * H);
    float *h = (float *)malloc(sizeof(float) * H);
    if (!gi || !gh || !h) {
        free(gi); free(gh); free(h);
        return O_ERR_NOMEM;
    }
    if (out_seq)
        memset(out_seq, 0, sizeof(float) * (size_t)B * T * H);
    for (int b = 0; b < B; ++b) {
        memset(h, 0, sizeof(float) * H);
        int L = len[b];
        for (int s = 0; s < L; ++s) {
            int t = reverse ? (L - 1 - s) : s;
            const float *xt = x + ((size_t)b * T + t) * I;
            o_affine(xt, W_ih, b_ih, 3 * H, I, gi);
            o_affine(h, W_hh, b_hh, 3 * H, H, gh);
            float *st = stash ? stash + ((size_t)b * T + t) * 5 * H : NULL;
            for (int u = 0; u < H; ++u) {
                float r = o_sigmoid(gi[u] + gh[u]);
                float z = o_sigmoid(gi[H + u] + gh[H + u]);
                float n = tanhf(gi[2 * H + u] + r * gh[2 * H + u]);
                if (st) {
                    st[u] = r;
                    st[H + u] = z;
                    st[2 * H + u] = n;
                    st[3 * H + u] = gh[2 * H + u];
                    st[4 * H + u] = h[u];
                }
                h[u] = (h[u] - n) * z + n;
            }
            if (out_seq)
                memcpy(out_seq + ((size_t)b * T + t) * H, h, sizeof(float) * H);
        }
        memcpy(h_final + (size_t)b * H, h, sizeof(float) * H);
    }
    free(gi); free(gh); free(h);
    return O_OK;
}

/*
 * One LSTM direction (torch.nn.LSTM on a PackedSequence; the reference builds it with
 * getattr(nn, rnn_type.upper()), backend/model.py:30-37, and keeps h_n: model.py:59-60).
 * Gate order i,f,g,o over the rows of W_ih [4H,I] / W_hh [4H,H]:
 *   i = s(.)  f = s(.)  g = tanh(.)  o = s(.)   of   W_i* x + b_i* + W_h* h + b_h*
 *   c' = f c + i g ;  h' = o tanh(c')
 * stash (nullable) [B,T,7,H]: i, f, g, o, c_prev, c', h_prev.
 */
int o_lstm_layer(const float *x, int B, int T, int I, const int32_t *len,
                 const float *W_ih, const float *W_hh, const float *b_ih,
                 const float *b_hh, int H, int reverse, float *out_seq,
                 float *h_final, float *stash)
{
    float *gi = (float *)malloc(sizeof(float) * 4 * H);
    float *gh = (float *)malloc(sizeof(float) * 4 * H);
    float *h = (float *)malloc(sizeof(float) * H);
    float *c = (float *)malloc(sizeof(float) * H);
    if (!gi || !gh || !h || !c) {
        free(gi); free(gh); free(h); free(c);
        return O_ERR_NOMEM;
    }
    if (out_seq)
        memset(out_seq, 0, sizeof(float) * (size_t)B * T * H);
    for (int b = 0; b < B; ++b) {
        memset(h, 0, sizeof(float) * H);
        memset(c, 0, sizeof(float) * H);
        int L = len[b];
        for (int s = 0; s < L; ++s) {
            int t = reverse ? (L - 1 - s) : s;
            const float *xt = x + ((size_t)b * T + t) * I;
            o_affine(xt, W_ih, b_ih, 4 * H, I, gi);
            o_affine(h, W_hh, b_hh, 4 * H, H, gh);
            float *st = stash ? stash + ((size_t)b * T + t) * 7 * H : NULL;
            for (int u = 0; u < H; ++u) {
                float ig = o_sigmoid(gi[u] + gh[u]);
                float fg = o_sigmoid(gi[H + u] + gh[H + u]);
                float gg = tanhf(gi[2 * H + u] + gh[2 * H + u]);
                float og = o_sigmoid(gi[3 * H + u] + gh[3 * H + u]);
                float cn = fg * c[u] + ig * gg;
                if (st) {
                    st[u] = ig;
                    st[H + u] = fg;
                    st[2 * H + u] = gg;
                    st[3 * H + u] = og;
                    st[4 * H + u] = c[u];
                    st[5 * H + u] = cn;
                    st[6 * H + u] = h[u];
                }
                c[u] = cn;
                h[u] = og * tanhf(cn);
            }
            if (out_seq)
                memcpy(out_seq + ((size_t)b * T + t) * H, h, sizeof(float) * H);
        }
        memcpy(h_final + (size_t)b * H, h, sizeof(float) * H);
    }
    free(gi); free(gh); free(h); free(c);
    return O_OK;
}

/*
 * One vanilla-RNN direction (torch.nn.RNN, default nonlinearity tanh; rnn_type "RNN",
 * backend/model.py:30,61-62):  h' = tanh(W_ih x + b_ih + W_hh h + b_hh).
 * stash (nullable) [B,T,2,H]: h', h_prev.
 */
int o_rnn_layer(const float *x, int B, int T, int I, const int32_t *len,
                const float *W_ih, const float *W_hh, const float *b_ih,
                const float *b_hh, int H, int reverse, float *out_seq,
                float *h_final, float *stash)
{
    float *gi = (float *)malloc(sizeof(float) * H);
    float *gh = (float *)malloc(sizeof(float) * H);
    float *h = (float *)malloc(sizeof(float) * H);
    if (!gi || !gh || !h) {
        free(gi); free(gh); free(h);
        return O_ERR_NOMEM;
    }
    if (out_seq)
        memset(out_seq, 0, sizeof(float) * (size_t)B * T * H);
    for (int b = 0; b < B; ++b) {
        memset(h, 0, sizeof(float) * H);
        int L = len[b];
        for (int s = 0; s < L; ++s) {
            int t = reverse ? (L - 1 - s) : s;
            const float *xt = x + ((size_t)b * T + t) * I;
            o_affine(xt, W_ih, b_ih, H, I, gi);
            o_affine(h, W_hh, b_hh, H, H, gh);
            float *st = stash ? stash + ((size_t)b * T + t) * 2 * H : NULL;
            for (int u = 0; u < H; ++u) {
                float hn = tanhf(gi[u] + gh[u]);
                if (st) {
                    st[u] = hn;
                    st[H + u] = h[u];
                }
                h[u] = hn;
            }
            if (out_seq)
                memcpy(out_seq + ((size_t)b * T + t) * H, h, sizeof(float) * H);
        }
        memcpy(h_final + (size_t)b * H, h, sizeof(float) * H);
    }
    free(gi); free(gh); free(h);
    return O_OK;
}

/* cell: 0 GRU, 1 LSTM, 2 RNN(tanh) -- the reference's RNN_TYPE (config.json; model.py:30) */
static int o_cell_gates(int cell) { return cell == 1 ? 4 : (cell == 2 ? 1 : 3); }
static int o_cell_stash(int cell) { return cell == 1 ? 7 : (cell == 2 ? 2 : 5); }

static int o_cell_layer(int cell, const float *x, int B, int T, int I, const int32_t *len,
                        const float *W_ih, const float *W_hh, const float *b_ih,
                        const float *b_hh, int H, int reverse, float *out_seq,
                        float *h_final, float *stash)
{
    if (cell == 1)
        return o_lstm_layer(x, B, T, I, len, W_ih, W_hh, b_ih, b_hh, H, reverse, out_seq, h_final, stash);
    if (cell == 2)
        return o_rnn_layer(x, B, T, I, len, W_ih, W_hh, b_ih, b_hh, H, reverse, out_seq, h_final, stash);
    return o_gru_layer(x, B, T, I, len, W_ih, W_hh, b_ih, b_hh, H, reverse, out_seq, h_final, stash);
}

/* F.normalize(hidden, p=2, dim=1), eps 1e-12.  backend/model.py:73-74 */
void o_l2_normalize(const float *x, int B, int H, float *y)
{
    for (int b = 0; b < B; ++b) {
        float ss = 0.0f;
        for (int u = 0; u < H; ++u)
            ss += x[(size_t)b * H + u] * x[(size_t)b * H + u];
        float nrm = sqrtf(ss);
        if (nrm < 1e-12f)
            nrm = 1e-12f;
        for (int u = 0; u < H; ++u)
            y[(size_t)b * H + u] = x[(size_t)b * H + u] / nrm;
    }
}

/*
 * Inter-layer dropout (nn.GRU(dropout=p), backend/model.py:31-37; config.json DROPOUT): in train mode
 * the output sequence of every layer but the last is multiplied by a Bernoulli(1-p) mask / (1-p)
 * before it feeds the next layer.  torch draws the mask from its own RNG stream, which cannot be
 * matched; the build DEFINES the mask as a counter-based hash of (seed, layer, b, t, column) so the
 * HIP kernels and this oracle agree bit for bit and the backward pass can regenerate it.
 */
static inline float o_dropout_scale(uint64_t seed, int layer, uint64_t idx, float p)
{
    uint64_t x = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(layer + 1);
    x ^= idx * 0xD1342543DE82EF95ull;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    uint32_t u = (uint32_t)(x >> 32);
    double t = (double)p * 4294967296.0;
    uint32_t thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    return u >= thresh ? 1.0f / (1.0f - p) : 0.0f;
}

/* mask values for elements 0..n-1 of one layer boundary (test hook for the mask's statistics) */
void o_dropout_mask(uint64_t seed, int layer, int64_t n, float p, float *out)
{
    for (int64_t i = 0; i < n; ++i)
        out[i] = o_dropout_scale(seed, layer, (uint64_t)i, p);
}

/*
 * RNNEncoder.forward, GRU flavour.  backend/model.py:48-75
 * w holds 4 pointers per (layer, direction): index (layer*ndir + dir)*4 +
 * {0:W_ih [3H,I], 1:W_hh [3H,H], 2:b_ih [3H], 3:b_hh [3H]}; layer-0 I = E,
 * deeper layers I = ndir*H.  Bidirectional: hidden = cat(h_n[-2], h_n[-1])
 * -> Linear(2H,H) (model.py:65-69).  Inter-layer dropout is identity (eval
 * mode / DROPOUT 0): parity is defined per call in eval mode.
 */
int o_encoder_forward_cell(int cell, const int64_t *ids, int B, int T, const float *table,
                           int64_t V, int E, int H, int num_layers, int bidir,
                           const float *const *w, const float *proj_w,
                           const float *proj_b, int normalize, float dropout_p,
                           uint64_t dropout_seed, float *out);

int o_encoder_forward(const int64_t *ids, int B, int T, const float *table,
                      int64_t V, int E, int H, int num_layers, int bidir,
                      const float *const *w, const float *proj_w,
                      const float *proj_b, int normalize, float dropout_p,
                      uint64_t dropout_seed, float *out)
{
    return o_encoder_forward_cell(0, ids, B, T, table, V, E, H, num_layers, bidir, w, proj_w, proj_b, normalize,
                                  dropout_p, dropout_seed, out);
}

/* The same for any RNN_TYPE the reference accepts (cell: 0 GRU, 1 LSTM -- h_n is used, model.py:59-60 --, 2 RNN);
 * W_ih / W_hh / biases then have 4H (LSTM) or H (RNN) rows. */
int o_encoder_forward_cell(int cell, const int64_t *ids, int B, int T, const float *table,
                           int64_t V, int E, int H, int num_layers, int bidir,
                           const float *const *w, const float *proj_w,
                           const float *proj_b, int normalize, float dropout_p,
                           uint64_t dropout_seed, float *out)
{
    if (B <= 0 || T <= 0 || num_layers < 1 || cell < 0 || cell > 2)
        return O_ERR_BAD_SHAPE;
    int ndir = bidir ? 2 : 1;
    int32_t *len = (int32_t *)malloc(sizeof(int32_t) * B);
    if (!len)
        return O_ERR_NOMEM;
    int rc = o_lengths(ids, B, T, len);
    if (rc != O_OK) {
        free(len);
        return rc;
    }
    size_t in_w = (size_t)(E > ndir * H ? E : ndir * H);
    float *cur = (float *)malloc(sizeof(float) * (size_t)B * T * in_w);
    float *nxt = (float *)malloc(sizeof(float) * (size_t)B * T * ndir * H);
    float *seq = (float *)malloc(sizeof(float) * (size_t)B * T * H);
    float *hfin = (float *)malloc(sizeof(float) * (size_t)B * ndir * H);
    float *hid = (float *)malloc(sizeof(float) * (size_t)B * H);
    if (!cur || !nxt || !seq || !hfin || !hid) {
        rc = O_ERR_NOMEM;
        goto done;
    }
    rc = o_gather(ids, B, T, table, V, E, cur);
    if (rc != O_OK)
        goto done;
    int I = E;
    for (int l = 0; l < num_layers; ++l) {
        for (int d = 0; d < ndir; ++d) {
            const float *const *p = w + ((size_t)l * ndir + d) * 4;
            rc = o_cell_layer(cell, cur, B, T, I, len, p[0], p[1], p[2], p[3], H, d,
                              seq, hfin + (size_t)d * B * H, NULL);
            if (rc != O_OK)
                goto done;
            for (size_t i = 0; i < (size_t)B * T; ++i)
                memcpy(nxt + i * ndir * H + (size_t)d * H, seq + i * H,
                       sizeof(float) * H);
        }
        I = ndir * H;
        if (dropout_p > 0.0f && l + 1 < num_layers)
            for (size_t i = 0; i < (size_t)B * T * I; ++i)
                nxt[i] *= o_dropout_scale(dropout_seed, l, i, dropout_p);
        memcpy(cur, nxt, sizeof(float) * (size_t)B * T * I);
    }
    if (bidir) {
        float *cat = (float *)malloc(sizeof(float) * 2 * H);
        if (!cat) {
            rc = O_ERR_NOMEM;
            goto done;
        }
        for (int b = 0; b < B; ++b) {
            memcpy(cat, hfin + (size_t)b * H, sizeof(float) * H);
            memcpy(cat + H, hfin + (size_t)B * H + (size_t)b * H,
                   sizeof(float) * H);
            o_affine(cat, proj_w, proj_b, H, 2 * H, hid + (size_t)b * H);
        }
        free(cat);
    } else {
        memcpy(hid, hfin, sizeof(float) * (size_t)B * H);
    }
    if (normalize)
        o_l2_normalize(hid, B, H, out);
    else
        memcpy(out, hid, sizeof(float) * (size_t)B * H);
done:
    free(len); free(cur); free(nxt); free(seq); free(hfin); free(hid);
    return rc;
}

/* ------------------------------------------------------------------ */
/* brute-force scoring + top-k                                         */
/* ------------------------------------------------------------------ */

/* (score desc, index asc) strict ordering: 1 if a ranks before b. */
static inline int o_before(float sa, int64_t ia, float sb, int64_t ib)
{
    return sa > sb || (sa == sb && ia < ib);
}

/*
 * torch.matmul(q, D.t()) then torch.topk(s, k).
 * backend/evaluators.py:185-186, :269-272; backend/trainer.py:62-65
 * Score = fp32 FMA chain over the feature index in ascending order,
 *   acc = fmaf(q[j], d[j], acc), acc0 = 0
 * which is exactly what a chain of v_mfma_f32_32x32x2_f32 computes, so the
 * HIP kernel is bit-identical to this function.  torch.topk leaves the order
 * of tied scores unspecified; this restatement DEFINES it as
 * (score descending, index ascending).  Indices are global: idx_offset + row.
 * When N < k the tail is filled with (-inf, -1).
 */
int o_score_topk(const float *Q, int B, int d, const float *D, int64_t N,
                 int k, int64_t idx_offset, float *out_val, int64_t *out_idx)
{
    if (B < 0 || d <= 0 || k <= 0 || N < 0)
        return O_ERR_BAD_SHAPE;
    for (int b = 0; b < B; ++b) {
        float *v = out_val + (size_t)b * k;
        int64_t *ix = out_idx + (size_t)b * k;
        int cnt = 0;
        const float *q = Q + (size_t)b * d;
        for (int64_t n = 0; n < N; ++n) {
            const float *row = D + (size_t)n * d;
            float acc = 0.0f;
            for (int j = 0; j < d; ++j)
                acc = fmaf(q[j], row[j], acc);
            int64_t gi = idx_offset + n;
            if (cnt == k && !o_before(acc, gi, v[k - 1], ix[k - 1]))
                continue;
            int p = cnt < k ? cnt : k - 1;
            while (p > 0 && o_before(acc, gi, v[p - 1], ix[p - 1])) {
                v[p] = v[p - 1];
                ix[p] = ix[p - 1];
                --p;
            }
            v[p] = acc;
            ix[p] = gi;
            if (cnt < k)
                ++cnt;
        }
        for (int p = cnt; p < k; ++p) {
            v[p] = -INFINITY;
            ix[p] = -1;
        }
    }
    return O_OK;
}

/*
 * Merge of per-shard / per-tile partial top-k lists into the global top-k
 * (new in the build).  in_val/in_idx are [B, M]
 * candidates in any order; entries with idx < 0 are padding.  Output order
 * (score desc, index asc).
 */
int o_topk_merge(const float *in_val, const int64_t *in_idx, int B, int M,
                 int k, float *out_val, int64_t *out_idx)
{
    for (int b = 0; b < B; ++b) {
        float *v = out_val + (size_t)b * k;
        int64_t *ix = out_idx + (size_t)b * k;
        int cnt = 0;
        for (int m = 0; m < M; ++m) {
            float s = in_val[(size_t)b * M + m];
            int64_t gi = in_idx[(size_t)b * M + m];
            if (gi < 0)
                continue;
            if (cnt == k && !o_before(s, gi, v[k - 1], ix[k - 1]))
                continue;
            int p = cnt < k ? cnt : k - 1;
            while (p > 0 && o_before(s, gi, v[p - 1], ix[p - 1])) {
                v[p] = v[p - 1];
                ix[p] = ix[p - 1];
                --p;
            }
            v[p] = s;
            ix[p] = gi;
            if (cnt < k)
                ++cnt;
        }
        for (int p = cnt; p < k; ++p) {
            v[p] = -INFINITY;
            ix[p] = -1;
        }
    }
    return O_OK;
}

/*
 * Rank of one designated document per query (1-based), as BatchEvaluator
 * derives it from a full descending sort.  backend/evaluators.py:58-65
 * rank = 1 + #{n : n ranks before target} under (score desc, index asc).
 */
int o_score_rank(const float *Q, int B, int d, const float *D, int64_t N,
                 const int64_t *target, int64_t *rank)
{
    for (int b = 0; b < B; ++b) {
        const float *q = Q + (size_t)b * d;
        int64_t tg = target[b];
        if (tg < 0 || tg >= N)
            return O_ERR_BAD_INDEX;
        float st = 0.0f;
        for (int j = 0; j < d; ++j)
            st = fmaf(q[j], D[(size_t)tg * d + j], st);
        int64_t r = 1;
        for (int64_t n = 0; n < N; ++n) {
            float acc = 0.0f;
            for (int j = 0; j < d; ++j)
                acc = fmaf(q[j], D[(size_t)n * d + j], acc);
            if (n != tg && o_before(acc, n, st, tg))
                ++r;
        }
        rank[b] = r;
    }
    return O_OK;
}

/* ------------------------------------------------------------------ */
/* triplet loss                                                        */
/* ------------------------------------------------------------------ */

/* F.cosine_similarity(a, b) for one row, eps 1e-8: each norm is clamped
 * separately (ATen: x / max(||x||, eps)).  backend/model.py:112-113
 * Returns cos; fills unit vectors ah, bh and clamped norms. */
static float o_cos_row(const float *a, const float *b, int H, float *na_out,
                       float *nb_out)
{
    float sa = 0.0f, sb = 0.0f;
    for (int u = 0; u < H; ++u) {
        sa += a[u] * a[u];
        sb += b[u] * b[u];
    }
    float na = sqrtf(sa), nb = sqrtf(sb);
    if (na < 1e-8f) na = 1e-8f;
    if (nb < 1e-8f) nb = 1e-8f;
    float dot = 0.0f;
    for (int u = 0; u < H; ++u)
        dot += (a[u] / na) * (b[u] / nb);
    *na_out = na;
    *nb_out = nb;
    return dot;
}

/*
 * triplet_loss_cosine forward + gradient w.r.t. q, p, n.
 * backend/model.py:109-114
 *   loss = mean_b max(0, cos(q,n) - cos(q,p) + margin)
 * clamp(min=0) passes gradient where its argument >= 0 (ATen clamp_min
 * backward mask).  d cos(a,b)/da = (bh - cos*ah) / |a|  (norm above eps).
 * dq/dp/dn may be NULL (forward only).
 */
int o_triplet_loss(const float *q, const float *p, const float *n, int B,
                   int H, float margin, float *loss, float *dq, float *dp,
                   float *dn)
{
    if (B <= 0)
        return O_ERR_BAD_SHAPE;
    float total = 0.0f;
    for (int b = 0; b < B; ++b) {
        const float *qb = q + (size_t)b * H, *pb = p + (size_t)b * H,
                    *nb = n + (size_t)b * H;
        float nq1, np_, nq2, nn_;
        float cp = o_cos_row(qb, pb, H, &nq1, &np_);
        float cn = o_cos_row(qb, nb, H, &nq2, &nn_);
        float v = cn - cp + margin;
        float act = v >= 0.0f ? 1.0f : 0.0f;
        total += v > 0.0f ? v : 0.0f;
        if (dq && dp && dn) {
            float g = act / (float)B;
            for (int u = 0; u < H; ++u) {
                float qh = qb[u] / nq1, ph = pb[u] / np_, nh = nb[u] / nn_;
                /* dL/dq = g * (dcn/dq - dcp/dq) */
                dq[(size_t)b * H + u] =
                    g * ((nh - cn * qh) / nq1 - (ph - cp * qh) / nq1);
                dp[(size_t)b * H + u] = -g * (qh - cp * ph) / np_;
                dn[(size_t)b * H + u] = g * (qh - cn * nh) / nn_;
            }
        }
    }
    *loss = total / (float)B;
    return O_OK;
}

/* ------------------------------------------------------------------ */
/* encoder backward (BPTT), restating what loss.backward() computes    */
/* through backend/model.py:48-75 (backend/main.py:254).               */
/* ------------------------------------------------------------------ */

/* Backward of F.normalize: y = x/max(|x|,eps); dx = (dy - y*(y.dy))/|x| */
static void o_l2_normalize_bwd(const float *x, const float *dy, int B, int H,
                               float *dx)
{
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (size_t)b * H, *dyb = dy + (size_t)b * H;
        float ss = 0.0f;
        for (int u = 0; u < H; ++u)
            ss += xb[u] * xb[u];
        float nrm = sqrtf(ss);
        if (nrm < 1e-12f) {
            for (int u = 0; u < H; ++u)
                dx[(size_t)b * H + u] = dyb[u] / 1e-12f;
            continue;
        }
        float dot = 0.0f;
        for (int u = 0; u < H; ++u)
            dot += (xb[u] / nrm) * dyb[u];
        for (int u = 0; u < H; ++u)
            dx[(size_t)b * H + u] = (dyb[u] - (xb[u] / nrm) * dot) / nrm;
    }
}

/*
 * Backward of one GRU direction.  Inputs: x, len, weights, the stash written
 * by o_gru_layer, d_out_seq (nullable [B,T,H]: gradient w.r.t. the emitted
 * sequence) and d_h_final [B,H].  Accumulates into gW_ih,gW_hh,gb_ih,gb_hh
 * and writes dx (nullable [B,T,I], ACCUMULATED so two directions can share
 * it; caller zeroes).
 */
static int o_gru_layer_bwd(const float *x, int B, int T, int I,
                           const int32_t *len, const float *W_ih,
                           const float *W_hh, int H, int reverse,
                           const float *stash, const float *d_out_seq,
                           const float *d_h_final, float *gW_ih, float *gW_hh,
                           float *gb_ih, float *gb_hh, float *dx)
{
    float *dh = (float *)malloc(sizeof(float) * H);
    float *dgi = (float *)malloc(sizeof(float) * 3 * H);
    float *dgh = (float *)malloc(sizeof(float) * 3 * H);
    float *dhp = (float *)malloc(sizeof(float) * H);
    if (!dh || !dgi || !dgh || !dhp) {
        free(dh); free(dgi); free(dgh); free(dhp);
        return O_ERR_NOMEM;
    }
    for (int b = 0; b < B; ++b) {
        int L = len[b];
        memcpy(dh, d_h_final + (size_t)b * H, sizeof(float) * H);
        for (int s = L - 1; s >= 0; --s) {
            int t = reverse ? (L - 1 - s) : s;
            const float *st = stash + ((size_t)b * T + t) * 5 * H;
            const float *xt = x + ((size_t)b * T + t) * I;
            if (d_out_seq)
                for (int u = 0; u < H; ++u)
                    dh[u] += d_out_seq[((size_t)b * T + t) * H + u];
            for (int u = 0; u < H; ++u) {
                float r = st[u], z = st[H + u], n = st[2 * H + u],
                      ghn = st[3 * H + u], hp = st[4 * H + u];
                /* h' = (hp - n) z + n */
                float dz = dh[u] * (hp - n);
                float dn = dh[u] * (1.0f - z);
                dhp[u] = dh[u] * z;
                float dn_pre = dn * (1.0f - n * n);
                float dz_pre = dz * z * (1.0f - z);
                float dr = dn_pre * ghn;
                float dr_pre = dr * r * (1.0f - r);
                dgi[u] = dr_pre;
                dgi[H + u] = dz_pre;
                dgi[2 * H + u] = dn_pre;
                dgh[u] = dr_pre;
                dgh[H + u] = dz_pre;
                dgh[2 * H + u] = dn_pre * r;
            }
            for (int g = 0; g < 3 * H; ++g) {
                gb_ih[g] += dgi[g];
                gb_hh[g] += dgh[g];
                float a = dgi[g], c = dgh[g];
                float *wi = gW_ih + (size_t)g * I;
                float *wh = gW_hh + (size_t)g * H;
                for (int k = 0; k < I; ++k)
                    wi[k] += a * xt[k];
                const float *hp = st + 4 * H;
                for (int k = 0; k < H; ++k)
                    wh[k] += c * hp[k];
                const float *whr = W_hh + (size_t)g * H;
                for (int k = 0; k < H; ++k)
                    dhp[k] += c * whr[k];
                if (dx) {
                    const float *wir = W_ih + (size_t)g * I;
                    float *dxt = dx + ((size_t)b * T + t) * I;
                    for (int k = 0; k < I; ++k)
                        dxt[k] += a * wir[k];
                }
            }
            memcpy(dh, dhp, sizeof(float) * H);
        }
    }
    free(dh); free(dgi); free(dgh); free(dhp);
    return O_OK;
}

/* Backward of one LSTM direction (stash of o_lstm_layer); same contract as o_gru_layer_bwd. */
static int o_lstm_layer_bwd(const float *x, int B, int T, int I,
                            const int32_t *len, const float *W_ih,
                            const float *W_hh, int H, int reverse,
                            const float *stash, const float *d_out_seq,
                            const float *d_h_final, float *gW_ih, float *gW_hh,
                            float *gb_ih, float *gb_hh, float *dx)
{
    float *dh = (float *)malloc(sizeof(float) * H);
    float *dc = (float *)malloc(sizeof(float) * H);
    float *dg = (float *)malloc(sizeof(float) * 4 * H);
    float *dhp = (float *)malloc(sizeof(float) * H);
    if (!dh || !dc || !dg || !dhp) {
        free(dh); free(dc); free(dg); free(dhp);
        return O_ERR_NOMEM;
    }
    for (int b = 0; b < B; ++b) {
        int L = len[b];
        memcpy(dh, d_h_final + (size_t)b * H, sizeof(float) * H);
        memset(dc, 0, sizeof(float) * H);
        for (int s = L - 1; s >= 0; --s) {
            int t = reverse ? (L - 1 - s) : s;
            const float *st = stash + ((size_t)b * T + t) * 7 * H;
            const float *xt = x + ((size_t)b * T + t) * I;
            if (d_out_seq)
                for (int u = 0; u < H; ++u)
                    dh[u] += d_out_seq[((size_t)b * T + t) * H + u];
            for (int u = 0; u < H; ++u) {
                float ig = st[u], fg = st[H + u], gg = st[2 * H + u], og = st[3 * H + u];
                float cp = st[4 * H + u], cn = st[5 * H + u];
                float tc = tanhf(cn);
                float d_o = dh[u] * tc;
                float dct = dc[u] + dh[u] * og * (1.0f - tc * tc);
                dg[u] = dct * gg * ig * (1.0f - ig);
                dg[H + u] = dct * cp * fg * (1.0f - fg);
                dg[2 * H + u] = dct * ig * (1.0f - gg * gg);
                dg[3 * H + u] = d_o * og * (1.0f - og);
                dc[u] = dct * fg;
                dhp[u] = 0.0f;
            }
            const float *hp = st + 6 * H;
            for (int g = 0; g < 4 * H; ++g) {
                float a = dg[g];
                gb_ih[g] += a;
                gb_hh[g] += a;
                float *wi = gW_ih + (size_t)g * I;
                float *wh = gW_hh + (size_t)g * H;
                for (int k = 0; k < I; ++k)
                    wi[k] += a * xt[k];
                for (int k = 0; k < H; ++k)
                    wh[k] += a * hp[k];
                const float *whr = W_hh + (size_t)g * H;
                for (int k = 0; k < H; ++k)
                    dhp[k] += a * whr[k];
                if (dx) {
                    const float *wir = W_ih + (size_t)g * I;
                    float *dxt = dx + ((size_t)b * T + t) * I;
                    for (int k = 0; k < I; ++k)
                        dxt[k] += a * wir[k];
                }
            }
            memcpy(dh, dhp, sizeof(float) * H);
        }
    }
    free(dh); free(dc); free(dg); free(dhp);
    return O_OK;
}

/* Backward of one vanilla-RNN direction (stash of o_rnn_layer). */
static int o_rnn_layer_bwd(const float *x, int B, int T, int I,
                           const int32_t *len, const float *W_ih,
                           const float *W_hh, int H, int reverse,
                           const float *stash, const float *d_out_seq,
                           const float *d_h_final, float *gW_ih, float *gW_hh,
                           float *gb_ih, float *gb_hh, float *dx)
{
    float *dh = (float *)malloc(sizeof(float) * H);
    float *dg = (float *)malloc(sizeof(float) * H);
    float *dhp = (float *)malloc(sizeof(float) * H);
    if (!dh || !dg || !dhp) {
        free(dh); free(dg); free(dhp);
        return O_ERR_NOMEM;
    }
    for (int b = 0; b < B; ++b) {
        int L = len[b];
        memcpy(dh, d_h_final + (size_t)b * H, sizeof(float) * H);
        for (int s = L - 1; s >= 0; --s) {
            int t = reverse ? (L - 1 - s) : s;
            const float *st = stash + ((size_t)b * T + t) * 2 * H;
            const float *xt = x + ((size_t)b * T + t) * I;
            if (d_out_seq)
                for (int u = 0; u < H; ++u)
                    dh[u] += d_out_seq[((size_t)b * T + t) * H + u];
            for (int u = 0; u < H; ++u) {
                dg[u] = dh[u] * (1.0f - st[u] * st[u]);
                dhp[u] = 0.0f;
            }
            const float *hp = st + H;
            for (int g = 0; g < H; ++g) {
                float a = dg[g];
                gb_ih[g] += a;
                gb_hh[g] += a;
                float *wi = gW_ih + (size_t)g * I;
                float *wh = gW_hh + (size_t)g * H;
                for (int k = 0; k < I; ++k)
                    wi[k] += a * xt[k];
                for (int k = 0; k < H; ++k)
                    wh[k] += a * hp[k];
                const float *whr = W_hh + (size_t)g * H;
                for (int k = 0; k < H; ++k)
                    dhp[k] += a * whr[k];
                if (dx) {
                    const float *wir = W_ih + (size_t)g * I;
                    float *dxt = dx + ((size_t)b * T + t) * I;
                    for (int k = 0; k < I; ++k)
                        dxt[k] += a * wir[k];
                }
            }
            memcpy(dh, dhp, sizeof(float) * H);
        }
    }
    free(dh); free(dg); free(dhp);
    return O_OK;
}

static int o_cell_layer_bwd(int cell, const float *x, int B, int T, int I,
                            const int32_t *len, const float *W_ih,
                            const float *W_hh, int H, int reverse,
                            const float *stash, const float *d_out_seq,
                            const float *d_h_final, float *gW_ih, float *gW_hh,
                            float *gb_ih, float *gb_hh, float *dx)
{
    if (cell == 1)
        return o_lstm_layer_bwd(x, B, T, I, len, W_ih, W_hh, H, reverse, stash, d_out_seq, d_h_final, gW_ih, gW_hh,
                                gb_ih, gb_hh, dx);
    if (cell == 2)
        return o_rnn_layer_bwd(x, B, T, I, len, W_ih, W_hh, H, reverse, stash, d_out_seq, d_h_final, gW_ih, gW_hh,
                               gb_ih, gb_hh, dx);
    return o_gru_layer_bwd(x, B, T, I, len, W_ih, W_hh, H, reverse, stash, d_out_seq, d_h_final, gW_ih, gW_hh,
                           gb_ih, gb_hh, dx);
}

/*
 * Gradient of the encoder output w.r.t. every trainable tensor, given
 * d_out [B,H] (gradient w.r.t. RNNEncoder.forward's return value).
 * With GloVe vectors the embedding table is frozen (model.py:25-27) and gets
 * no gradient: pass g_table = NULL.  Without them the reference trains the
 * table (model.py:23, nn.Embedding(..., padding_idx=0)): g_table [V,E] then
 * receives the sum, over the valid positions holding each id, of the
 * gradient w.r.t. that position's input vector; row 0 stays zero
 * (padding_idx).  g has the same layout as w (4 pointers per (layer,dir));
 * all gradient buffers are OVERWRITTEN.  g_proj_w/g_proj_b only when bidir.
 */
int o_encoder_backward_cell(int cell, const int64_t *ids, int B, int T, const float *table,
                            int64_t V, int E, int H, int num_layers, int bidir,
                            const float *const *w, const float *proj_w,
                            const float *proj_b, int normalize, float dropout_p,
                            uint64_t dropout_seed, const float *d_out,
                            float *const *g, float *g_proj_w, float *g_proj_b,
                            float *g_table);

int o_encoder_backward(const int64_t *ids, int B, int T, const float *table,
                       int64_t V, int E, int H, int num_layers, int bidir,
                       const float *const *w, const float *proj_w,
                       const float *proj_b, int normalize, float dropout_p,
                       uint64_t dropout_seed, const float *d_out,
                       float *const *g, float *g_proj_w, float *g_proj_b,
                       float *g_table)
{
    return o_encoder_backward_cell(0, ids, B, T, table, V, E, H, num_layers, bidir, w, proj_w, proj_b, normalize,
                                   dropout_p, dropout_seed, d_out, g, g_proj_w, g_proj_b, g_table);
}

/* The same for any RNN_TYPE (cell: 0 GRU, 1 LSTM, 2 RNN); weight / gradient tensors have 4H or H rows. */
int o_encoder_backward_cell(int cell, const int64_t *ids, int B, int T, const float *table,
                            int64_t V, int E, int H, int num_layers, int bidir,
                            const float *const *w, const float *proj_w,
                            const float *proj_b, int normalize, float dropout_p,
                            uint64_t dropout_seed, const float *d_out,
                            float *const *g, float *g_proj_w, float *g_proj_b,
                            float *g_table)
{
    if (cell < 0 || cell > 2)
        return O_ERR_BAD_SHAPE;
    const int NG = o_cell_gates(cell), SW = o_cell_stash(cell);
    int ndir = bidir ? 2 : 1;
    int rc = O_OK;
    int32_t *len = (int32_t *)malloc(sizeof(int32_t) * B);
    if (!len)
        return O_ERR_NOMEM;
    rc = o_lengths(ids, B, T, len);
    if (rc != O_OK) {
        free(len);
        return rc;
    }
    size_t BT = (size_t)B * T;
    /* per-layer inputs and stashes */
    float **xin = (float **)calloc(num_layers + 1, sizeof(float *));
    float **stash = (float **)calloc((size_t)num_layers * ndir, sizeof(float *));
    float *seq = (float *)malloc(sizeof(float) * BT * H);
    float *hfin = (float *)malloc(sizeof(float) * (size_t)B * ndir * H);
    float *hid = (float *)malloc(sizeof(float) * (size_t)B * H);
    float *dhid = (float *)malloc(sizeof(float) * (size_t)B * H);
    float *dhfin = (float *)calloc((size_t)B * ndir * H, sizeof(float));
    float *dseq = NULL, *dseq_next = NULL, *dtmp = NULL;
    if (!xin || !stash || !seq || !hfin || !hid || !dhid || !dhfin) {
        rc = O_ERR_NOMEM;
        goto done;
    }
    xin[0] = (float *)malloc(sizeof(float) * BT * E);
    if (!xin[0]) { rc = O_ERR_NOMEM; goto done; }
    rc = o_gather(ids, B, T, table, V, E, xin[0]);
    if (rc != O_OK)
        goto done;
    int I = E;
    for (int l = 0; l < num_layers; ++l) {
        xin[l + 1] = (float *)malloc(sizeof(float) * BT * ndir * H);
        if (!xin[l + 1]) { rc = O_ERR_NOMEM; goto done; }
        for (int d = 0; d < ndir; ++d) {
            const float *const *p = w + ((size_t)l * ndir + d) * 4;
            stash[l * ndir + d] = (float *)malloc(sizeof(float) * BT * SW * H);
            if (!stash[l * ndir + d]) { rc = O_ERR_NOMEM; goto done; }
            rc = o_cell_layer(cell, xin[l], B, T, I, len, p[0], p[1], p[2], p[3], H,
                              d, seq, hfin + (size_t)d * B * H,
                              stash[l * ndir + d]);
            if (rc != O_OK)
                goto done;
            for (size_t i = 0; i < BT; ++i)
                memcpy(xin[l + 1] + i * ndir * H + (size_t)d * H, seq + i * H,
                       sizeof(float) * H);
        }
        I = ndir * H;
        /* layer l+1 consumes the DROPPED sequence; the recurrence's own h (in the stash) is not dropped */
        if (dropout_p > 0.0f && l + 1 < num_layers)
            for (size_t i = 0; i < BT * I; ++i)
                xin[l + 1][i] *= o_dropout_scale(dropout_seed, l, i, dropout_p);
    }
    /* head */
    if (bidir) {
        float *cat = (float *)malloc(sizeof(float) * 2 * H);
        if (!cat) { rc = O_ERR_NOMEM; goto done; }
        for (int b = 0; b < B; ++b) {
            memcpy(cat, hfin + (size_t)b * H, sizeof(float) * H);
            memcpy(cat + H, hfin + (size_t)B * H + (size_t)b * H,
                   sizeof(float) * H);
            o_affine(cat, proj_w, proj_b, H, 2 * H, hid + (size_t)b * H);
        }
        free(cat);
    } else {
        memcpy(hid, hfin, sizeof(float) * (size_t)B * H);
    }
    if (normalize)
        o_l2_normalize_bwd(hid, d_out, B, H, dhid);
    else
        memcpy(dhid, d_out, sizeof(float) * (size_t)B * H);
    if (bidir) {
        memset(g_proj_w, 0, sizeof(float) * (size_t)H * 2 * H);
        memset(g_proj_b, 0, sizeof(float) * H);
        for (int b = 0; b < B; ++b)
            for (int o = 0; o < H; ++o) {
                float gd = dhid[(size_t)b * H + o];
                g_proj_b[o] += gd;
                for (int d = 0; d < 2; ++d)
                    for (int u = 0; u < H; ++u) {
                        float hv = hfin[(size_t)d * B * H + (size_t)b * H + u];
                        g_proj_w[(size_t)o * 2 * H + d * H + u] += gd * hv;
                        dhfin[(size_t)d * B * H + (size_t)b * H + u] +=
                            gd * proj_w[(size_t)o * 2 * H + d * H + u];
                    }
            }
    } else {
        memcpy(dhfin, dhid, sizeof(float) * (size_t)B * H);
    }
    /* layers, top down.  h_n of lower layers gets no direct gradient. */
    dseq = NULL; /* gradient w.r.t. xin[l+1] (= concat of dir outputs) */
    for (int l = num_layers - 1; l >= 0; --l) {
        int Il = l == 0 ? E : ndir * H;
        dseq_next = NULL;
        if (l > 0 || g_table) {
            dseq_next = (float *)calloc(BT * Il, sizeof(float));
            if (!dseq_next) { rc = O_ERR_NOMEM; goto done; }
        }
        for (int d = 0; d < ndir; ++d) {
            const float *const *p = w + ((size_t)l * ndir + d) * 4;
            float *const *gp = g + ((size_t)l * ndir + d) * 4;
            memset(gp[0], 0, sizeof(float) * (size_t)NG * H * Il);
            memset(gp[1], 0, sizeof(float) * (size_t)NG * H * H);
            memset(gp[2], 0, sizeof(float) * NG * H);
            memset(gp[3], 0, sizeof(float) * NG * H);
            const float *dos = NULL;
            if (dseq) {
                dtmp = (float *)malloc(sizeof(float) * BT * H);
                if (!dtmp) { rc = O_ERR_NOMEM; goto done; }
                for (size_t i = 0; i < BT; ++i)
                    for (int u = 0; u < H; ++u) {
                        size_t ix = i * ndir * H + (size_t)d * H + u;
                        float gv = dseq[ix]; /* gradient w.r.t. the dropped sequence */
                        if (dropout_p > 0.0f)
                            gv *= o_dropout_scale(dropout_seed, l, ix, dropout_p);
                        dtmp[i * H + u] = gv;
                    }
                dos = dtmp;
            }
            float *zero_hf = NULL;
            const float *dhf;
            if (l == num_layers - 1) {
                dhf = dhfin + (size_t)d * B * H;
            } else {
                zero_hf = (float *)calloc((size_t)B * H, sizeof(float));
                if (!zero_hf) { rc = O_ERR_NOMEM; goto done; }
                dhf = zero_hf;
            }
            rc = o_cell_layer_bwd(cell, xin[l], B, T, Il, len, p[0], p[1], H, d,
                                  stash[l * ndir + d], dos, dhf, gp[0], gp[1],
                                  gp[2], gp[3], dseq_next);
            free(zero_hf);
            free(dtmp);
            dtmp = NULL;
            if (rc != O_OK)
                goto done;
        }
        free(dseq);
        dseq = dseq_next;
        dseq_next = NULL;
    }
    if (g_table) { /* dseq = gradient w.r.t. the gathered vectors [B,T,E]; nn.Embedding backward, padding_idx 0 */
        memset(g_table, 0, sizeof(float) * (size_t)V * E);
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < len[b]; ++t) {
                int64_t id = ids[(size_t)b * T + t];
                if (id == 0)
                    continue;
                const float *dx = dseq + ((size_t)b * T + t) * E;
                float *gr = g_table + (size_t)id * E;
                for (int k = 0; k < E; ++k)
                    gr[k] += dx[k];
            }
    }
done:
    free(len);
    if (xin)
        for (int l = 0; l <= num_layers; ++l)
            free(xin[l]);
    if (stash)
        for (int i = 0; i < num_layers * ndir; ++i)
            free(stash[i]);
    free(xin); free(stash); free(seq); free(hfin); free(hid); free(dhid);
    free(dhfin); free(dseq); free(dseq_next); free(dtmp);
    return rc;
}

/* ------------------------------------------------------------------ */
/* optimiser step                                                      */
/* ------------------------------------------------------------------ */

/*
 * torch.nn.utils.clip_grad_norm_(params, max_norm) followed by
 * torch.optim.Adam(lr, betas=(b1,b2), eps, weight_decay=0).step()
 * over ONE flat fp32 buffer.  backend/main.py:257,259 (Adam built :222)
 *   total = ||g||_2 ; coef = min(1, max_norm / (total + 1e-6)) ; g *= coef
 *   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
 *   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * max_norm <= 0 disables clipping.  step is the 1-based step count.
 * grads are scaled in place, like the reference.
 */
int o_clip_adam_step(float *p, float *g, float *m, float *v, int64_t n,
                     int64_t step, float lr, float b1, float b2, float eps,
                     float max_norm, float *total_norm_out)
{
    double ss = 0.0;
    for (int64_t i = 0; i < n; ++i)
        ss += (double)g[i] * (double)g[i];
    float total = (float)sqrt(ss);
    if (total_norm_out)
        *total_norm_out = total;
    if (max_norm > 0.0f) {
        float coef = max_norm / (total + 1e-6f);
        if (coef > 1.0f)
            coef = 1.0f;
        for (int64_t i = 0; i < n; ++i)
            g[i] *= coef;
    }
    float bc1 = 1.0f - powf(b1, (float)step);
    float bc2 = 1.0f - powf(b2, (float)step);
    float step_size = lr / bc1;
    float bc2_sqrt = sqrtf(bc2);
    for (int64_t i = 0; i < n; ++i) {
        m[i] = m[i] + (g[i] - m[i]) * (1.0f - b1); /* lerp, as ATen */
        v[i] = v[i] * b2 + (1.0f - b2) * g[i] * g[i];
        float denom = sqrtf(v[i]) / bc2_sqrt + eps;
        p[i] -= step_size * (m[i] / denom);
    }
    return O_OK;
}
