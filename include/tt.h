/*
 * tt.h -- C ABI of libtt.so: the MI355X (gfx950) implementation of the
 * two-tower retrieval hot path of jpe17/TwoTowerMLRetrieval.
 *
 * The reference is pure Python and has no FFI seam: the
 * arithmetic of this path is stock PyTorch calls.  Each entry point below
 * replaces one of those call sites (cited as reference file:line, paths
 * relative to the reference repository root); INTEGRATION.md shows the
 * ctypes stub a reference maintainer would add to bind them.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *    the parameter comment says "host".  All matrices are row-major, dense.
 *  - the library never allocates or frees user-visible memory and keeps no
 *    reference past return: outputs and workspace are caller-owned; workspace
 *    size comes from the matching *_workspace_bytes() query.
 *  - every call is asynchronous on `stream` (a hipStream_t; NULL = the null
 *    stream), issues no hipDeviceSynchronize and may be captured in a hipGraph.
 *  - every call returns an int status (TT_OK = 0) and never throws;
 *    tt_last_error() returns a thread-local message for the last failure.
 *  - re-entrant: no global mutable state besides that thread-local string.
 */
#ifndef TT_H
#define TT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *tt_stream_t; /* hipStream_t */

#define TT_TOPK_INVALID_INDEX (1ll << 62) /* see tt_score_topk_f32 */

enum tt_status {
    TT_OK = 0,
    TT_ERR_BAD_SHAPE = 1,   /* -> ValueError / RuntimeError in the Python shim */
    TT_ERR_BAD_INDEX = 2,   /* token id outside [0,V): IndexError (g10_errors.json) */
    TT_ERR_ZERO_LENGTH = 3, /* a row with no non-zero id: RuntimeError (model.py:55-57) */
    TT_ERR_UNSUPPORTED = 4, /* shape/dtype outside what the kernels are built for */
    TT_ERR_WORKSPACE = 5,   /* workspace NULL or too small */
    TT_ERR_HIP = 6          /* a HIP runtime call failed; message has hipGetErrorString */
};

const char *tt_version(void);
const char *tt_last_error(void);

/* hipEvent wrappers for hosts without HIP headers: used with the `prof_events` arguments (a HOST array
 * of two events recorded on the stream right before / after an entry point's dominant kernel). */
int tt_event_create(void **ev);
int tt_event_destroy(void *ev);
int tt_event_elapsed_ms(void *start, void *stop, float *ms); /* blocks until `stop` has happened */

/* ------------------------------------------------------------------ */
/* Brute-force scoring + top-k                                         */
/* ------------------------------------------------------------------ */

/*
 * Replaces  torch.matmul(q, D.t()) ; torch.topk(scores, k)
 *   backend/evaluators.py:185-186, :269-272 ; backend/trainer.py:62-65
 * Q [B,d] f32, D [N,d] f32 -> out_val [B,k] f32 (descending), out_idx [B,k]
 * int64 = idx_offset + row of D.  The [B,N] score matrix is never formed.
 * Score = fp32 FMA chain over the feature index in ascending order (exactly
 * oracle/tt_oracle.c:o_score_topk).  Ties: (score desc, index asc).  When
 * N < k the tail is (-inf, -1).  Supported: d in {32,64,96,128,192,256} (32-query tiles, 32x32x2 MFMA) and
 * {320,384,448,512} (16-query tiles, 16x16x4 MFMA: the same chain, bit for bit), 1 <= k <= 64,
 * N < 2^31 - 64 per call (shard larger corpora; idx_offset makes indices global).
 * Inputs must be finite (a NaN score is never selected).
 * Every wait between waves inside the kernels is bounded.  One of them cannot be skipped without losing documents (a wave
 * waiting ~2 us for a neighbour's draw from the shared tile pool, B >= 96 only): should its budget (~4 ms) ever run out, the
 * wave marks its partial lists (+inf, index >= TT_TOPK_INVALID_INDEX) and tt_score_topk_f32 itself redoes the affected
 * 32-query tiles, on the device and on the stream, with the static split of the corpus (no pool, nobody to wait for): the
 * results are complete and exact either way, and no caller has to look for the marker.  (Never observed outside the test that
 * forces it.  tt_score_topk_partials_f32 hands its partial lists out as they are, marker included.)
 */
/* Diagnostic: byte offset, in the workspace of a finished tt_score_topk_f32 call of this shape, of one int32 per 32-query tile
 * that is non-zero when the tile was done again for that reason; (size_t)-1 = the shape never draws from a shared pool. */
size_t tt_score_topk_redo_flags_offset(int B, int64_t N, int d, int k);
/* Diagnostic: byte offset, in the workspace of a finished exact search of this shape, of an int32 counting the waves whose
 * chunk-pacing wait timed out (the pacing counters are coherent only among waves on one XCD: a launch whose chunks straddle
 * XCDs -- partition modes, tiny grids -- still returns exact results, each wave ~0.15 ms late once); (size_t)-1 = shape not paced. */
size_t tt_score_topk_pace_timeouts_offset(int B, int64_t N, int d, int k);
size_t tt_score_topk_workspace_bytes(int B, int64_t N, int d, int k);
int tt_score_topk_f32(const float *Q, int B, int d, const float *D, int64_t N, int k,
                      int64_t idx_offset, float *out_val, int64_t *out_idx, void *workspace,
                      size_t workspace_bytes, tt_stream_t stream);

/*
 * First half of tt_score_topk_f32 alone: the streaming score + per-tile top-k kernel.
 * Leaves [B, *part_m] unordered candidates (idx -1 = empty) in the workspace and returns
 * host pointers-to-device-pointers to them; follow with tt_topk_merge.  Exposed so a
 * sharded caller can gather candidates itself and so bench.py can time the kernel alone.
 */
int tt_score_topk_partials_f32(const float *Q, int B, int d, const float *D, int64_t N, int k,
                               int64_t idx_offset, void *workspace, size_t workspace_bytes,
                               const float **part_val /*host out*/, const int64_t **part_idx /*host out*/,
                               int *part_m /*host out*/, void *const *prof_events /*NULL, or 2 tt events
                               recorded on `stream` right before/after the main-pass launch*/,
                               tt_stream_t stream);

/*
 * Screened exact search, the fast path at every batch size: SAME result as tt_score_topk_f32 (bit-exact
 * scores, same tie order), computed as an fp16-MFMA screen over an fp16 shadow copy of the corpus
 * with a rigorous error bound, followed by exact fp32 rescoring of the survivors
 * (csrc/screen.hip header; DESIGN.md "K4s", "K4t").  B <= 64 takes the streaming form (bound by HBM streaming
 * of the shadow copy, N x 512 B), larger batches the shared-tile form (bound by the fp16 MFMA rate).  tt_index_build_f16 makes the shadow copy D16 [N,d]
 * (2 bytes/element) and stats[2] = {largest row L2 norm, largest |element|} (device floats; the
 * caller reads stats[0] once at index-build time and passes it as dmax_norm).  If the screen cannot
 * guarantee exactness for some query (tie cluster too large, |q| beyond fp16) it sets
 * fallback_flag[query / 32] (device int32 array of ceil(B/32) entries) and the exact kernel,
 * predicated on those flags on the device, recomputes the flagged 32-query tiles -- no host
 * synchronisation.  Supported: d = 256, k <= 64, dmax_norm < 6e4.
 */
int tt_index_build_f16(const float *D, int64_t N, int d, void *D16, float *stats, tt_stream_t stream);
/* Same for a block of a bf16 corpus (BASELINE configs[4]: passages kept as bf16 in pinned host DRAM and
 * streamed through the GPU): D_bf16 [N,d] -> D32 [N,d] fp32 (exact widening) and, when D16 is not NULL,
 * the fp16 shadow; stats (nullable) accumulates the two maxima across calls unless reset_stats != 0. */
int tt_index_build_from_bf16(const void *D_bf16, int64_t N, int d, float *D32, void *D16, float *stats,
                             int reset_stats, tt_stream_t stream);
size_t tt_score_topk_screened_workspace_bytes(int B, int64_t N, int d, int k);
/* Byte offset, inside the workspace of a finished screened search of this (B, N, k), of its per-query statistics:
 * int32 [B][2] = (candidates pooled over the whole corpus, survivors rescored exactly).  Diagnostic: how much the filter
 * let through on the caller's data; read it after the call, before the workspace is reused. */
size_t tt_score_topk_screened_stats_offset(int B, int64_t N, int d, int k);
int tt_score_topk_screened_f32(const float *Q, int B, int d, const float *D32, const void *D16, int64_t N, int k,
                               float dmax_norm, int64_t idx_offset, float *out_val, int64_t *out_idx,
                               int32_t *fallback_flag, void *workspace, size_t workspace_bytes,
                               void *const *prof_events /*host, nullable*/, tt_stream_t stream);

/*
 * The same search in two calls around an exchange of the seed thresholds -- for a ROW-SHARDED corpus, where a
 * shard's own sample gives a much weaker seed than the whole corpus would: ≈ 8 candidates per workgroup-tile on a 1.25M-row
 * shard against 1 on 10M rows, and the append path, not the matrix pipes, then sets the pace of the screen.
 *   tt_score_topk_screened_seed_f32       query image + sample pass; seed[q] = the k_seed-th largest sample maximum of THIS
 *                                         shard (k_seed = the k of the FINAL answer, <= k = the per-shard list length)
 *   tt_score_topk_screened_seed_list_f32  the same pass, but seed_list[q][0..k_seed) = this shard's k_seed LARGEST sample
 *                                         maxima (unordered; -3e38 / -inf entries where the shard has no such sample)
 *   [caller]                              ONE all-gather of the ranks' seed lists (k_seed floats per query and rank: 40 KB per
 *                                         rank at B = 1024, k_seed = 10 -- tt_allgather_topk moves any byte block), then
 *   tt_seed_union_f32                     seed[q] = the kth-th largest of the world * list_len gathered values (kth = the FINAL
 *                                         k; list_len = k_seed of the lists, normally kth too -- a job so wide that
 *                                         world * kth > 512 lists fewer per rank, list_len = 512 / world >= kth / world).  They are
 *                                         approximate scores of DISTINCT documents (one per 32-document sample tile, the shards'
 *                                         rows are disjoint), so kth documents of the whole corpus reach seed[q]: it bounds the
 *                                         GLOBAL kth-th best approximate score from below -- the seed the unsharded search
 *                                         would take from a sample of the same total size
 *   tt_score_topk_screened_seeded_f32     the screen with thresholds seed[q] - 2 eps_q, pooling + exact rescoring, predicated exact
 *                                         kernels; writes this shard's documents above the global threshold (up to k, best first,
 *                                         the rest padded with -inf / -1): merged over the shards they contain the exact global
 *                                         top-k_seed (any valid lower bound works as seed[]: a single shard's own, or the union's)
 * Both calls take the SAME (B, N, k), flags and workspace; nothing else may use the workspace in between.
 */
int tt_score_topk_screened_seed_f32(const float *Q, int B, int d, const void *D16, int64_t N, int k, int k_seed, float dmax_norm,
                                    int32_t *fallback_flag, float *seed /*[B] out*/, void *workspace, size_t workspace_bytes,
                                    tt_stream_t stream);
int tt_score_topk_screened_seed_list_f32(const float *Q, int B, int d, const void *D16, int64_t N, int k, int k_seed,
                                         float dmax_norm, int32_t *fallback_flag, float *seed_list /*[B][k_seed] out*/,
                                         void *workspace, size_t workspace_bytes, tt_stream_t stream);
int tt_seed_union_f32(const float *lists /*[world][B][list_len]*/, int world, int B, int list_len, int kth,
                      float *seed /*[B] out*/, tt_stream_t stream);
int tt_score_topk_screened_seeded_f32(const float *Q, int B, int d, const float *D32, const void *D16, int64_t N, int k,
                                      float dmax_norm, int64_t idx_offset, float *out_val, int64_t *out_idx,
                                      int32_t *fallback_flag, const float *seed /*[B]*/, void *workspace, size_t workspace_bytes,
                                      void *const *prof_events /*host, nullable*/, tt_stream_t stream);

/*
 * Merge of partial top-k lists (per tile, per shard after the RCCL all-gather:
 * SURVEY 8e) into the global top-k: in_val/in_idx [B,M] candidates in any
 * order, idx < 0 = padding; out [B,k], (score desc, index asc), tail (-inf,-1).
 * New in the build (the reference is single-device); oracle: o_topk_merge.
 */
int tt_topk_merge(const float *in_val, const int64_t *in_idx, int B, int M, int k, float *out_val,
                  int64_t *out_idx, tt_stream_t stream);

/*
 * The same merge over the per-shard lists of a row-sharded index, read IN PLACE from the all-gather's
 * receive buffer (north_star: "per-shard top-k merged via RCCL all-gather"): rank r's block starts at
 * gathered + r*rank_stride and holds vals f32 [B,kp] at byte 0 and idx int64 [B,kp] at byte
 * idx_byte_offset (both 8-byte aligned).  Result: global top-k per row, same order rule.
 */
int tt_topk_merge_shards(const void *gathered, int world, size_t rank_stride, size_t idx_byte_offset,
                         int B, int kp, int k, float *out_val, int64_t *out_idx, tt_stream_t stream);

/*
 * Rank (1-based) of one designated document per query under (score desc,
 * index asc), what BatchEvaluator extracts from a full sort per row.
 *   backend/evaluators.py:50,58-65
 * Q [B,d], D [N,d], target [B] int64 (row of D) -> rank [B] int64.
 */
int tt_score_rank_f32(const float *Q, int B, int d, const float *D, int64_t N, const int64_t *target,
                      int64_t *rank, tt_stream_t stream);

/*
 * Every score of every query, for callers that blend the dense score of ALL documents with another signal:
 *   dense_scores = cosine_similarity([query_emb], self.doc_embeddings)[0]     backend/simple_hybrid.py:53-54
 * Q [B,d], D [N,d] -> S [B,N] f32, S[b][n] = the same ascending-index fp32 FMA chain tt_score_topk_f32 selects from
 * (for unit-norm rows that is the cosine).  Materialises B*N floats: meant for the small corpora that idiom is
 * used on; the top-k entry points above never form this matrix.  d multiple of 4, <= 512.
 */
int tt_score_all_f32(const float *Q, int B, int d, const float *D, int64_t N, float *S, tt_stream_t stream);

/* ------------------------------------------------------------------ */
/* Encoder tower (GloVe gather -> GRU -> L2-normalise)                 */
/* ------------------------------------------------------------------ */

/*
 * Replaces RNNEncoder.forward                              backend/model.py:48-75
 *   embedded = self.embedding(x)                           :49
 *   lengths = (x != 0).sum(dim=1)  (count of non-zero ids) :52
 *   pack_padded_sequence + getattr(nn, rnn_type.upper())   :30,55-62
 *     rnn_type 0 = GRU (gate rows r,z,n), 1 = LSTM (i,f,g,o; h_n is kept, :59-60), 2 = RNN (tanh):
 *     weight_ih / weight_hh / biases have G*H rows, G = 3 / 4 / 1
 *   h_n[-1], or cat(h_n[-2], h_n[-1]) -> Linear(2H,H)      :65-71
 *   F.normalize(p=2, dim=1, eps 1e-12)                     :73-74
 * ids [B,T] int64 right-padded with 0; table [V,E] f32 (row 0 is a real word vector and IS
 * used when id 0 occurs inside the first `length` positions); out [B,H] f32.
 * weights: HOST array of 4*num_layers*ndir DEVICE pointers, index (layer*ndir + dir)*4 +
 * {0: weight_ih [GH,I], 1: weight_hh [GH,H], 2: bias_ih [GH], 3: bias_hh [GH]} = the reference's
 * state_dict tensors rnn.{weight_ih,weight_hh,bias_ih,bias_hh}_l{layer}[_reverse]; I = E for
 * layer 0, ndir*H above.  proj_w [H,2H], proj_b [H] only when bidirectional.
 * Inter-layer dropout (nn.GRU(dropout=p), config.json DROPOUT) is applied when train != 0,
 * dropout_p > 0 and num_layers > 1: the output sequence of every layer but the last is multiplied by
 * a Bernoulli(1-p) mask / (1-p).  torch's RNG stream cannot be matched, so the mask is DEFINED as a
 * counter-based hash of (dropout_seed, layer, element) -- the same function in oracle/tt_oracle.c --
 * which the backward pass regenerates (pass the same dropout_p / dropout_seed to it).
 * Lengths are computed on the device; nothing synchronises with the host.  Data errors cannot be
 * returned synchronously, so they are reported through `status` (device int32, nullable), written
 * on the stream: bit 0 = a row with no non-zero id (the reference raises RuntimeError), bit 1 = an
 * id outside [0,V) (IndexError).  Such rows produce finite garbage, never a fault.  bit 2 = the column-split GRU
 * recurrence (H = 256, B <= 1024: a row group's gate columns on four workgroups that hand the hidden state to each other
 * every step, csrc/gru16x4.hip) gave up waiting for a partner workgroup: its waits are bounded, so a workgroup that is
 * never scheduled ends the call with this bit instead of a hang; the outputs are then invalid: call again with
 * TT_ENC_ONE_WORKGROUP, which keeps every recurrence on the one-workgroup kernels (no hand-off between workgroups, nothing to
 * wait for; results bit-identical to the column-split forward's).  The column-split kernels need every member of a row
 * group's team on a CU at the same time: ONE such launch always fits (checked against the device's CU count), a host that
 * keeps SEVERAL encoder calls in flight on different streams orders their recurrences with events (tt_enc_sync_t below: what
 * trainer.train_step does), or passes TT_ENC_ONE_WORKGROUP to all but the largest -- or relies on the bounded wait and the
 * retry above.
 * train: 0 inference, 1 training, 2 training with a trainable table, optionally | TT_ENC_ONE_WORKGROUP.  train != 0 keeps the
 * activations the backward pass needs inside the workspace: the SAME workspace (sized with the same train value) must then be
 * passed, untouched, to tt_encoder_backward_f32.  The library reads no environment variable on any call.
 * Supported: H multiple of 32 in [32,512], E multiple of 4, 1 <= num_layers <= 4.
 */
/*
 * Ordering the recurrences of calls that are in flight on DIFFERENT streams (nullable argument of the training calls).  A
 * column-split recurrence launch takes one CU per workgroup and needs all of them resident at once; two such launches on two
 * streams can ask for more CUs than the device has (query tower 128 + document tower 256 on 256 CUs).  Instead of hoping that
 * dispatch order lets complete teams form, the host passes events (hipEvent_t, created by the host: tt_event_create serves hosts
 * without a HIP binding):
 *   record_after_recurrence  recorded on `stream` right behind the call's LAST recurrence launch
 *   wait_before_recurrence   `stream` waits for it right in front of the call's FIRST recurrence launch
 * so call A's recurrences have drained before call B's start while everything else of the two calls (gathers, projections,
 * weight gradients) still overlaps.  The host must ISSUE the recording call before the waiting call (a wait on an event
 * whose record has not been issued is a no-op).  Both are ordinary stream operations: asynchronous, capturable.
 */
typedef struct tt_enc_sync {
    void *wait_before_recurrence;  /* hipEvent_t or NULL */
    void *record_after_recurrence; /* hipEvent_t or NULL */
} tt_enc_sync_t;

#define TT_ENC_TRAIN_MASK 0xff
#define TT_ENC_ONE_WORKGROUP 0x100 /* option bit of `train` / `opts`: recurrences on one workgroup per 16-row group */
/* tt_encoder_forward_f32 in two halves, so that a host can put ANOTHER call's launches between them: the same arguments and
 * workspace twice, once with TT_ENC_PHASE_BEGIN (input checks, lengths, weight conversion, layer 0's input projection: everything
 * in front of the first recurrence launch; `out` is not written) and once with TT_ENC_PHASE_FINISH (the recurrences and the
 * head).  With tt_enc_sync_t a host orders call B's recurrences behind call A's without delaying B's projection: B BEGIN, A
 * (records), B FINISH (waits) -- what trainer.train_step does with the document tower (B) and the query tower (A). */
#define TT_ENC_PHASE_BEGIN 0x200
#define TT_ENC_PHASE_FINISH 0x400
/* Option bit of `train` (forward) and `opts` (backward): `dropout_seed` is the ADDRESS of a device uint64 that holds the seed; the
 * kernels read it when they run.  A host that captures a train step into a HIP graph writes a new seed there before every replay
 * (trainer.GraphedTrainStep); passing the value instead would freeze one mask into the graph.  Forward and backward of a step must
 * see the same value. */
#define TT_ENC_SEED_ON_DEVICE 0x800
/* Option bit of `train` (forward), `opts` (prepared forward, backward): the REFERENCE'S OWN ARITHMETIC.  By default the GRU towers
 * with H = 128 / 256 take every matrix product as three f16 MFMAs on fp16 hi/lo splits of the fp32 operands (fp32-grade: ~4e-7 on
 * unit-norm outputs; csrc/gru16.hip has the error argument).  With TT_ENC_F32 every product of the call runs on the fp32-MFMA
 * kernels instead -- plain fp32 multiply-adds, as nn.GRU computes them (backend/model.py:31-37, :59-62) -- at several times the
 * cost.  A training step passes it to the forward AND the backward (the backward reads the workspace the forward filled);
 * tt_encoder_forward_prepared_f32 accepts it and then derives the fp32 kernels' weight forms per call (the prepared images are
 * the f16-split kernels'); no call with this bit is column-split (tt_encoder_split_workgroups counts without it), and the
 * projected table does not exist for it. */
#define TT_ENC_F32 0x1000
/* CUs the column-split recurrence of one call of this shape occupies (one workgroup each, all resident at once; a bidirectional
 * call whose two directions do not fit together runs them one launch after the other and this is ONE direction's count); 0 = the
 * call runs the one-workgroup kernels whatever `train` says.  A host with several calls in flight keeps the sum within the device's
 * CU count by passing TT_ENC_ONE_WORKGROUP to the smaller ones. */
int tt_encoder_split_workgroups(int B, int H, int bidirectional, int rnn_type);
size_t tt_encoder_workspace_bytes(int B, int T, int E, int H, int num_layers, int bidirectional, int rnn_type,
                                  int train /* as tt_encoder_forward_f32's (the option bits do not change the size) */,
                                  int dropout /* train && dropout_p > 0 && num_layers > 1 */);
int tt_encoder_forward_f32(const int64_t *ids, int B, int T, const float *table, int64_t V, int E, int H,
                           int num_layers, int bidirectional, int rnn_type, const float *const *weights /*host array*/,
                           const float *proj_w, const float *proj_b, int normalize, int train, float dropout_p,
                           uint64_t dropout_seed, float *out, void *workspace, size_t workspace_bytes,
                           int32_t *status, const tt_enc_sync_t *sync /*nullable*/, tt_stream_t stream);

/*
 * Inference with the weights converted ONCE.  tt_encoder_forward_f32 re-derives, on every call, what its kernels read
 * instead of the nn.GRU tensors (W_ih as an fp16 hi/lo MFMA-fragment stream, W_hh in packed fragment order, one
 * power-of-two scale word each): six small launches, ~40 us -- a quarter of a query-tower call.  A serving host
 * (backend/query_inferencer.py:51-75, frontend/main.py:150-156: weights loaded once, never changed) prepares them once:
 *   tt_encoder_prepared_bytes(...)  size of the caller-owned device buffer `prepared` (256-B aligned)
 *   tt_encoder_prepare_f32(...)     fills it from `weights` on `stream`; call again after the weights change
 *   tt_encoder_forward_prepared_f32 = tt_encoder_forward_f32 with train = 0 | opts reading `prepared` (same results, bit for
 *                                     bit); opts: 0, TT_ENC_ONE_WORKGROUP, TT_ENC_F32
 */
size_t tt_encoder_prepared_bytes(int E, int H, int num_layers, int bidirectional, int rnn_type);
int tt_encoder_prepare_f32(int E, int H, int num_layers, int bidirectional, int rnn_type,
                           const float *const *weights /*host array*/, void *prepared, size_t prepared_bytes,
                           tt_stream_t stream);
int tt_encoder_forward_prepared_f32(const int64_t *ids, int B, int T, const float *table, int64_t V, int E, int H,
                                    int num_layers, int bidirectional, int rnn_type,
                                    const float *const *weights /*host array*/, const void *prepared,
                                    const float *proj_w, const float *proj_b, int normalize, int opts, float *out,
                                    void *workspace, size_t workspace_bytes, int32_t *status, tt_stream_t stream);

/*
 * Inference without the input projection (the PROJECTED TABLE).  With GloVe loaded the embedding table is frozen
 * (backend/model.py:25-27) and at inference W_ih / b_ih are fixed too (the premise of tt_encoder_prepare_f32), so layer 0's
 * Gi = table[id] W_ih^T + b_ih of a token (model.py:49, :59-62: nn.Embedding, then nn.GRU's first half) depends on its id
 * alone.  A serving / index-building host computes it ONCE for every vocabulary row:
 *   tt_encoder_projected_bytes(...)    size of the caller-owned device buffer `projected` (256-B aligned): V x 3H floats per
 *                                      direction, 1.23 GB for V = 400 003, H = 256; 0 = this configuration has none
 *                                      (f16-split GRU only: H = 128 or 256)
 *   tt_encoder_project_table_f32(...)  fills it: the launch tt_encoder_forward_prepared_f32 makes for layer 0, over the rows
 *                                      0 .. V-1 in order (needs `prepared`; call again after the table or the weights change)
 *   tt_encoder_forward_projected_f32   = tt_encoder_forward_prepared_f32 without that launch: the recurrence kernels gather a
 *                                      step's projections from projected[id].  Same results, bit for bit.  No `table`
 *                                      argument: the call never reads it.
 * Workspace: tt_encoder_workspace_bytes(..., train = 0 | TT_ENC_PROJECTED, 0) (a one-layer model then needs no
 * [tokens][3H] scratch; the size for train = 0 is enough too).
 */
#define TT_ENC_PROJECTED 0x2000 /* tt_encoder_workspace_bytes only: size the workspace for tt_encoder_forward_projected_f32 */
size_t tt_encoder_projected_bytes(int64_t V, int E, int H, int bidirectional, int rnn_type);
int tt_encoder_project_table_f32(const float *table, int64_t V, int E, int H, int num_layers, int bidirectional, int rnn_type,
                                 const float *const *weights /*host array*/, const void *prepared, void *projected,
                                 size_t projected_bytes, tt_stream_t stream);
int tt_encoder_forward_projected_f32(const int64_t *ids, int B, int T, const void *projected, int64_t V, int E, int H,
                                     int num_layers, int bidirectional, int rnn_type, const float *const *weights /*host array*/,
                                     const void *prepared, const float *proj_w, const float *proj_b, int normalize, int opts,
                                     float *out, void *workspace, size_t workspace_bytes, int32_t *status, tt_stream_t stream);

/*
 * Two padded id batches as one: out [Ba + Bb][T] (T >= max(Ta, Tb)), rows of a then rows of b, padded with id 0
 * (padding_idx).  The train step of backend/main.py:244-259 runs positives and negatives through the SAME tower
 * (model.py:96-106): as one 2B-row call the recurrence fills twice the CUs and the weight-gradient products run once; this is
 * the host's torch.zeros + two slice copies in one launch.
 */
int tt_concat_ids_i64(const int64_t *a, int Ba, int Ta, const int64_t *b, int Bb, int Tb, int64_t *out, int T,
                      tt_stream_t stream);

/*
 * Replaces loss.backward() through RNNEncoder.forward     backend/main.py:254 over model.py:48-75
 * d_out [B,H] = gradient w.r.t. the forward's `out`.  `workspace` is the buffer a forward call with
 * train=1 and the SAME ids/weights filled.  grads: HOST array of DEVICE pointers laid out like
 * `weights`; every gradient buffer is OVERWRITTEN (the caller accumulates across calls, as autograd
 * does).  With GloVe vectors the embedding table is frozen (model.py:25-27) and gets no gradient: g_table = NULL.
 * Without them the reference trains it (nn.Embedding(V, E, padding_idx=0), model.py:23): pass g_table [V,E]
 * (overwritten: dense gradient, row 0 = padding_idx stays zero) and size / run the forward with train = 2.
 * opts: 0 or TT_ENC_ONE_WORKGROUP (the reverse-time recurrence on the one-workgroup kernel; any combination with the forward's
 * choice is valid -- the stash is the same bits either way).
 * status (nullable): device int32 word into which the call ORs bit 2 (value 4) when its column-split recurrence gave up
 * waiting for a partner workgroup (the gradients are then invalid; the bias gradients are NaN).  The word is ONLY OR-ed into,
 * never cleared: pass the forward call's status word (one read then covers both calls) or a word zeroed beforehand.  The
 * call writes to no other caller memory than grads / g_* / workspace / status.
 */
int tt_encoder_backward_f32(const int64_t *ids, int B, int T, const float *table, int64_t V, int E, int H,
                            int num_layers, int bidirectional, int rnn_type, const float *const *weights /*host array*/,
                            const float *proj_w, const float *proj_b, int normalize, float dropout_p,
                            uint64_t dropout_seed, const float *d_out, float *const *grads /*host array*/,
                            float *g_proj_w, float *g_proj_b, float *g_table /*nullable*/, void *workspace,
                            size_t workspace_bytes, int opts, int32_t *status /*nullable*/,
                            const tt_enc_sync_t *sync /*nullable*/, tt_stream_t stream);

/* ------------------------------------------------------------------ */
/* Training step pieces                                                */
/* ------------------------------------------------------------------ */

/*
 * Replaces triplet_loss_cosine((q,p,n), margin)            backend/model.py:109-114
 *   loss = mean(clamp(cos(q,n) - cos(q,p) + margin, min=0)), F.cosine_similarity eps 1e-8
 * and its gradient: loss [1]; dq/dp/dn [B,H] = d loss / d{q,p,n} (all three or none, nullable);
 * scratch_rows [B] f32.  Deterministic (fixed-order reduction).
 */
int tt_triplet_loss_f32(const float *q, const float *p, const float *n, int B, int H, float margin, float *loss,
                        float *dq, float *dp, float *dn, float *scratch_rows, tt_stream_t stream);

/*
 * Replaces torch.nn.utils.clip_grad_norm_(params, max_norm) ; torch.optim.Adam(...).step()
 *   backend/main.py:257,259 (optimizer built at :222; betas (0.9,0.999), eps 1e-8, no weight decay)
 * over ONE flat fp32 buffer of n elements (params, grads, exp_avg, exp_avg_sq).  grads are first
 * multiplied by grad_scale (1/world_size after a summing all-reduce: SURVEY 8e), then clipped by the
 * global L2 norm (max_norm <= 0: no clipping) and left in place, then Adam step number `step` (1-based).
 * total_norm_out (nullable, device) receives the pre-clip norm.  scratch: tt_clip_adam_scratch_bytes().
 */
size_t tt_clip_adam_scratch_bytes(void);
int tt_clip_adam_step_f32(float *params, float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, int64_t step,
                          float lr, float beta1, float beta2, float eps, float max_norm, float grad_scale,
                          float *total_norm_out, void *scratch, tt_stream_t stream);

/*
 * The same step as a COLLECTIVE DECISION of a data-parallel job, with nothing baked into the launch that changes from step to
 * step (so the whole train step can sit in one hipGraph):
 *   gate (nullable): device float [TT_STEP_GATE_WORDS], the tail of the flat gradient bucket -- the host all-reduces
 *     n + TT_STEP_GATE_WORDS floats (tt_allreduce_grads), so after the reduction every rank holds the SAME words:
 *     gate[b] = how many encoder calls, over all ranks, raised status bit b this step (tt_step_gate_f32 below).  If any word
 *     is non-zero NO rank applies the step: params, grads, moments and the step counter stay untouched (total_norm_out is
 *     still written).  The reference's step is single-process: a bad batch raises and the run stops
 *     (backend/main.py:244-259); here every rank reads the same reduced words and raises the same exception, instead of one
 *     rank leaving the step while its peers wait in the all-reduce.
 *   step_counter: device int64, the number of steps applied so far; incremented (by one thread, after the norm pass) iff
 *     the step is applied; the bias corrections 1 - beta^t are computed on the device from it, in double precision as
 *     torch.optim.Adam does on the host.
 */
#define TT_STEP_GATE_WORDS 4
int tt_clip_adam_step_gated_f32(float *params, float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                                int64_t *step_counter, float lr, float beta1, float beta2, float eps, float max_norm,
                                float grad_scale, float *total_norm_out, const float *gate /*nullable*/, void *scratch,
                                tt_stream_t stream);
/*
 * gate[b] = number of the n_status status words (HOST array of DEVICE int32 pointers, any number -- eight per launch --, null entries skipped)
 * that have bit b set, b = 0 .. 2 (tt_encoder_forward_f32's bits; the backward ORs bit 2 into the same words); gate[3] = 0.
 * One launch, on `stream`, behind the encoder calls it looks at.
 */
int tt_step_gate_f32(const int32_t *const *status_words /*host array*/, int n_status, float *gate, tt_stream_t stream);

/* ------------------------------------------------------------------ */
/* Collectives of the sharded path (RCCL over xGMI), SURVEY 8e          */
/* ------------------------------------------------------------------ */

/*
 * New in the build (the reference is single-device).  `comm` is an ncclComm_t CREATED BY THE HOST (one per rank and
 * GPU; e.g. torch.distributed's NCCL process group, or tt_comm_init_rank below) and passed as an opaque handle;
 * libtt.so does not link RCCL but binds, at first use, to the RCCL instance already mapped into the process
 * (tt_comm_library() names it), so the handle and the calls belong to the same library.  Both calls are
 * asynchronous on `stream`; as with any NCCL collective, every rank must issue them in the same order.
 *
 * tt_allgather_topk -- the ONE exchange of a row-sharded search, after the per-shard
 *   torch.matmul + torch.topk (backend/evaluators.py:185-186) that tt_score_topk*_f32 replaces:
 *   every rank contributes its block (vals f32 [B,kp] | idx int64 [B,kp], block_bytes bytes, written in place by
 *   the local search) and receives all `world` blocks in rank order in recv_blocks (world * block_bytes bytes) --
 *   exactly the layout tt_topk_merge_shards reads in place.
 * tt_allreduce_grads -- data-parallel training: the summing all-reduce of the flat fp32 gradient buffer, in place,
 *   between loss.backward() and clip_grad_norm_ (backend/main.py:254 -> :257); tt_clip_adam_step_f32 then
 *   applies grad_scale = 1/world before the norm, so every rank clips the same averaged gradient.
 */
int tt_allgather_topk(void *comm, const void *send_block, void *recv_blocks, size_t block_bytes, tt_stream_t stream);
int tt_allreduce_grads(void *comm, float *flat_grads, int64_t n, tt_stream_t stream);

/* Helpers for hosts without an RCCL binding of their own: rank 0 obtains a 128-byte id (HOST buffer), ships it to the
 * other ranks by any means, then every rank calls tt_comm_init_rank on its GPU (hipSetDevice first). */
const char *tt_comm_library(void);                 /* path of the RCCL libtt.so bound to ("" if none) */
int tt_comm_unique_id(void *id_bytes /*host, 128 bytes*/);
int tt_comm_init_rank(void **comm /*host out*/, int world, const void *id_bytes /*host*/, int rank);
int tt_comm_info(void *comm, int *world /*host out*/, int *rank /*host out*/);
int tt_comm_destroy(void *comm);

/* ------------------------------------------------------------------ */
/* Host-side text front end (no GPU): tokenise -> ids -> padded batch.  */
/* ------------------------------------------------------------------ */

/*
 * PretrainedTokenizer.encode for a batch of texts, natively and on several threads, so that the index
 * build (document tower at ~1e8 tokens/s) is not bound by a Python loop.
 *   backend/tokenizer.py:41-43   re.findall(r"\w+|[.,!?;]", str(s).lower()) -> word2idx.get(tok, unk)
 *   backend/main.py:50-56        pad_sequence(batch_first=True, padding_value=0)
 * tt_tok_create copies the vocabulary (n_words UTF-8 keys back to back in words_blob, word_off[n_words+1],
 * their ids).  tt_tok_encode tokenises text i = text_blob[text_off[i], text_off[i+1]) into
 * ragged_ids[text_off[i] ...] (a text never has more tokens than bytes), lens[i] tokens; ASCII semantics only:
 * a text with a byte >= 0x80 gets status[i] = 1, lens[i] = 0 and is left to the caller's Python path.
 * tt_tok_pad writes the right-padded [n_texts, width] int64 batch (pad id 0).  All buffers are host memory.
 */
int tt_tok_create(const char *words_blob, const int64_t *word_off, const int64_t *word_ids, int64_t n_words,
                  int64_t unk_id, void **handle);
void tt_tok_destroy(void *handle);
int tt_tok_encode(const void *handle, const char *text_blob, const int64_t *text_off, int64_t n_texts,
                  int64_t *ragged_ids, int32_t *lens, int32_t *status, int n_threads);
/* tt_tok_encode for texts joined into one blob with the byte `sep` between them (n_texts - 1 separators; fails with
 * TT_ERR_BAD_SHAPE when the count differs, i.e. a text contains `sep`): the offsets are found here (text_off_out [n_texts + 1],
 * text i = blob[text_off_out[i], text_off_out[i + 1] - 1)), which saves a Python host its per-text length pass. */
int tt_tok_encode_sep(const void *handle, const char *text_blob, int64_t blob_len, char sep, int64_t n_texts,
                      int64_t *text_off_out, int64_t *ragged_ids, int32_t *lens, int32_t *status, int n_threads);
/* tt_tok_encode for texts that lie wherever the host keeps them: texts[i] points at text_len[i] bytes (no terminator needed).
 * text_off_out [n_texts + 1] receives the running sum of the lengths; text i's ids go to ragged_ids[text_off_out[i] ...]
 * (capacity: the sum of the lengths), which is the layout tt_tok_pad reads.  A host whose strings are separate objects (CPython
 * str, Go string, Java byte[]) builds no blob: twotowermlretrieval_amd/csrc/pytext.c collects the pointers of a list of str. */
int tt_tok_encode_ptrs(const void *handle, const char *const *texts, const int64_t *text_len, int64_t n_texts,
                       int64_t *text_off_out, int64_t *ragged_ids, int32_t *lens, int32_t *status, int n_threads);
/* tt_tok_pad with 4-byte ids (half the bytes to copy to the device; TT_ERR_BAD_INDEX when an id does not fit an int32). */
int tt_tok_pad_i32(const int64_t *ragged_ids, const int64_t *text_off, const int32_t *lens, int64_t n_texts,
                   int64_t width, int32_t *out, int n_threads);
/* Text beyond ASCII, tokenised here with the HOST's Unicode tables so that the ids are the host's by construction.
 * tt_tok_set_unicode: low[cp] = the code point's lower-case code point (0xffffffff: no context-free single-code-point answer --
 * U+0130, U+03A3 in CPython -- a text that holds it gets status 1), cls[cp] = 0 other / 1 word (\w) / 2 one of .,!?; -- n_cp
 * entries each (at most 0x110000), copied.  tt_tok_encode_units = tt_tok_encode_ptrs for texts of 1-, 2- or 4-byte code units (one
 * code point per unit: CPython's compact str kinds): unit_bytes[i] = 0: text_len[i] ASCII bytes; 1 / 2 / 4: units of that size.
 * A token's key is the UTF-8 encoding of its lower-cased code points (what tt_tok_create's keys are). */
int tt_tok_set_unicode(void *handle, const uint32_t *low, const uint8_t *cls, int64_t n_cp);
int tt_tok_encode_units(const void *handle, const void *const *texts, const int64_t *text_len, const uint8_t *unit_bytes,
                        int64_t n_texts, int64_t *text_off_out, int64_t *ragged_ids, int32_t *lens, int32_t *status, int n_threads);
int tt_tok_pad(const int64_t *ragged_ids, const int64_t *text_off, const int32_t *lens, int64_t n_texts,
               int64_t width, int64_t *out, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* TT_H */
