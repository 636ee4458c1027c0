"""Metric shells of the reference's evaluators (backend/evaluators.py) on top of the HIP kernels.

The three classes main.py imports (backend/main.py:29: BatchEvaluator, CorpusEvaluator, TestEvaluator) keep their
constructor arguments, their evaluate(...) signatures and their use of `random` (same draws in the same order, so a
seeded run samples the same queries as the reference); the arithmetic inside is the HIP path:

  * BatchEvaluator (evaluators.py:18-79): validation loss + Recall@{1,5,10} + MRR with the positive of
    query i at index i.  The reference builds the full [Nq,Nq] score matrix (:50) and SORTS every row
    (:62) to find one rank; here `score_rank` counts the documents ranking before the positive in one
    streaming pass (ties: index ascending; the reference's torch.sort leaves tie order unspecified).
  * corpus_recall_hit (evaluators.py:177-209): Recall@k / Hit@k of one query against a document-
    embedding matrix, via the fused `score_topk`.
  * embed_documents (evaluators.py:162-175, 240-250; main.py:125-138): batches of strings ->
    [N,H] document embeddings through tokenizer.encode_batch + model.encode_document.
  * CorpusEvaluator (evaluators.py:83-209): unique queries / documents of the validation triplets, candidate and query
    sampling as in the reference, document embeddings in batches of 64, then ONE fused score + top-k over all sampled
    queries (the reference: one matmul + topk per query, :185-186) and the same Recall@k / Hit@k bookkeeping.
  * TestEvaluator (evaluators.py:212-283): the qualitative print-out (top-k documents per sampled query with their
    scores and whether they are ground-truth positives), :269-272's matmul + topk replaced the same way.
"""
from __future__ import annotations

import random
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np
import torch

from .index import score_rank, score_topk
from .model import triplet_loss_cosine


class BatchEvaluator:
    def __init__(self, top_k: List[int] = [1, 5, 10]):
        self.top_k = top_k

    def evaluate(self, model, val_loader: Iterable[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]],
                 device: torch.device, config: Dict):
        model.eval()
        q_all, d_all, losses = [], [], []
        with torch.no_grad():
            for queries, pos_docs, neg_docs in val_loader:
                queries, pos_docs, neg_docs = queries.to(device), pos_docs.to(device), neg_docs.to(device)
                q = model.encode_query(queries)
                d = model.encode_document(pos_docs)
                n = model.encode_document(neg_docs)
                losses.append(triplet_loss_cosine((q, d, n), margin=config.get("MARGIN", 0.2)))
                q_all.append(q)
                d_all.append(d)
        if not q_all:
            return {}, 0
        q_embs, d_embs = torch.cat(q_all), torch.cat(d_all)
        rank = score_rank(q_embs, d_embs, torch.arange(q_embs.shape[0], device=q_embs.device)).to(torch.float64)
        metrics = {f"Recall@{k}": float((rank <= k).double().mean().item()) for k in self.top_k}
        metrics["MRR"] = float((1.0 / rank).mean().item())
        return metrics, float(torch.stack(losses).mean().item())


def embed_documents(model, tokenizer, documents: Sequence[str], device: torch.device, batch_size: int = 64) -> torch.Tensor:
    """[len(documents), H] fp32 on the device, row i <-> documents[i] (the document_embeddings.npy layout).
    Same batching as the reference (backend/main.py:125-138, BATCH_SIZE documents per encoder call); for a large
    corpus use embed_corpus, which pipelines the host front end against the GPU."""
    out = []
    with torch.no_grad():
        for i in range(0, len(documents), batch_size):
            ids = tokenizer.encode_batch(documents[i:i + batch_size]).to(device)
            out.append(model.encode_document(ids))
    return torch.cat(out) if out else torch.empty((0, 0), device=device)


def embed_corpus(model, tokenizer, documents: Sequence[str], device: torch.device, batch_size: int = 32768,
                 prefetch: int = 2, out: torch.Tensor = None, producers: int = 0, stats: Dict = None,
                 threads_per_producer: int = 0, copy_ahead: int = 2) -> torch.Tensor:
    """Index build: the same rows as embed_documents, with the host front end off the critical path.
    `producers` host threads tokenise and pad batches natively (tt_tok_encode / tt_tok_pad release the GIL; each thread
    reuses its scratch arrays) into pinned memory while the GPU encodes; batches are consumed in document order whatever
    order they finish in; ids cross PCIe with a non-blocking copy on a side stream; embeddings land in one preallocated
    [N, H] matrix.  Rows are independent, so neither the batch size nor the number of producers changes the result.

    producers = 0: one, two from 32 cores of the host's share (cgroup quota respected) up.  History, 16-core share, GloVe-size
    vocabulary: rounds 1-3 ONE producer that page-locked a fresh tensor per batch and tokenised through a four-array hash table:
    93-115 M tokens/s against the document tower's ~240 M; round 4 (a): staging ring, single-pass tokenizer, one join per batch:
    two producers x 8 threads 160-234 M tokens/s depending on the box (0.66-0.95 of the GPU's rate); round 4 (b): pipelined
    table lookups, 16-byte slots on huge pages, the texts read in place (tt_tok_encode_ptrs: no join under the GIL): the native
    part does 590 M tokens/s on 8 threads and 1 050 M on 16 (tools/experiments/tok_harness.sh), every producer setting from
    1 x 16 to 4 x 4 builds at the GPU's rate (3.3-3.8 M passages/s, profiles/r04_w_index_build.log), and one producer with all
    the threads is the fastest of them (fewer Python threads trading the GIL with the consumer).  Round 5: with the projected
    table the document tower consumes 9-10 M passages/s and the front end binds again; 1 x 16 builds at 5.0-5.5 M passages/s,
    2 x 16 (the default from 12 cores up) at 6.7-7.7 M, 3 x 10 7.6 M, 4 x 8 6.5-6.7 M, 2 x 24 6.3 M (profiles/r05_g_index_build.log).
    batch_size 32 768: per-batch costs (launches, the copy's place in the queue, the recurrence's tail, two row tiles per workgroup
    from 16 384 passages up) amortise -- 1 M passages 3.55-3.67 M passages/s at 16 384 per batch, 3.89-3.92 M at 32 768, 3.83-3.90 at
    49 152 / 65 536 (tools/experiments/index_build_batch.py); 377 MB of pinned staging, ~10 GB of workspace.
    stats (optional dict): receives what the build used."""
    import collections
    import os
    from concurrent.futures import ThreadPoolExecutor
    n = len(documents)
    if n == 0:
        return torch.empty((0, 0), device=device)
    from .tokenizer import host_cores
    cores = host_cores()
    if producers <= 0:
        producers = 1 if cores < 12 else 2
    # (two producers on a 16-core share take 16 threads EACH: a producer's native section uses the whole share while the other
    #  is in its Python section -- text pointers, slicing, the hand-over)
    nt = threads_per_producer if threads_per_producer > 0 else max(1, min(16, cores if producers <= 2 else 2 * cores // producers))
    # (a quarter- and a half-size first batch, so that the GPU starts sooner, was measured and dropped: 0.844 -> 0.848 of the
    #  GPU-only rate; what is left of the gap is the copies running beside the kernels -- 62 ms resident, 67 ms with the pinned
    #  batches copied on the side stream, 74 ms with the tokenising: tools/experiments/index_build_gap.py)
    starts = [(i, min(i + batch_size, n)) for i in range(0, n, batch_size)]
    window = producers + max(1, prefetch)          # batches tokenised or being tokenised ahead of the GPU
    if stats is not None:
        stats.update(producers=producers, threads_per_producer=nt, host_cores=cores, batch_size=batch_size)

    # pinned staging buffers, kept across calls: page-locking a fresh 15-30 MB block per batch (what a pinned torch.empty does
    # whenever the host allocator has no free block of that size) costs more than tokenising the batch.  A call OWNS its buffers:
    # they are taken out of the process-wide pool for the call's duration (two builds at once -- or one that follows a build that
    # died -- never stage into the same block) and handed back only when no producer and no copy can still touch them.
    import queue
    cap = batch_size * 160                      # int64 ids: ~21 MB per buffer at the default batch size
    ring_key = (str(device), cap)
    mine = _take_staging(ring_key, window + 6, cap)
    free: "queue.SimpleQueue" = queue.SimpleQueue()
    for buf in mine:
        free.put(buf)

    def make(i):
        buf = free.get(timeout=120)             # (at most `window` jobs are outstanding and 3 batches in flight behind them; the
        #                                         time-out only ends a job whose consumer died: the pool's threads outlive the call)
        if buf is None:                         # the call is over (it failed): nothing to stage into
            raise RuntimeError("embed_corpus: the build this batch belonged to has ended")
        return tokenizer.encode_batch(documents[i[0]:i[1]], pin=True, n_threads=nt, out=buf, ids32=True), buf

    inflight = collections.deque()
    copy_stream = torch.cuda.Stream(device=device)
    cur = torch.cuda.current_stream(device)
    res = out
    # the producer threads are kept across calls: their per-thread scratch arrays (tokenizer._scratch: ~50 MB of ragged ids per
    # batch) would otherwise be page-faulted in again by every call's fresh threads -- ~0.09 s per call, more than a 400 k-passage
    # build takes
    pool = _PRODUCER_POOLS.get(producers)
    if pool is None:
        pool = _PRODUCER_POOLS[producers] = ThreadPoolExecutor(max_workers=producers, thread_name_prefix="tt-tok")
    with torch.no_grad():
        pending = collections.deque()
        staged = collections.deque()     # batches whose copy to the device has been issued: (rows, ids, copy event, pinned, buf)
        nxt = 0
        try:
            while nxt < len(starts) and len(pending) < window:
                pending.append((starts[nxt], pool.submit(make, starts[nxt])))
                nxt += 1

            def stage(block: bool) -> bool:
                """Issue the copy of the next tokenised batch (block: wait for its producer; else only if it is ready)."""
                nonlocal nxt
                if not pending or not (block or pending[0][1].done()):
                    return False
                i, fut = pending.popleft()
                ids_host, buf = fut.result()         # (re-raises a producer's exception here)
                if nxt < len(starts):
                    pending.append((starts[nxt], pool.submit(make, starts[nxt])))
                    nxt += 1
                with torch.cuda.stream(copy_stream):
                    ids = ids_host.to(device, non_blocking=True)   # (int32 when the vocabulary allows: on this platform the copy
                    #                                               is a shader kernel whose time adds to the encoder's)
                    ev_c = torch.cuda.Event()
                    ev_c.record(copy_stream)
                staged.append((i, ids, ev_c, ids_host, buf))
                return True

            while pending or staged:
                if not staged:
                    stage(True)
                # A copy issued BEHIND an encode call's launches does not start before they have run (0.25 ms per batch that
                # then adds to the build: tools/experiments/copy_order.py -- 67.6 ms against 64.4 with the copy one batch
                # ahead, 63.4 with the ids resident), so the next batches' copies go out IN FRONT of this batch's launches
                while len(staged) < 1 + copy_ahead and stage(False):
                    pass
                i, ids, ev_c, ids_host, buf = staged.popleft()
                cur.wait_event(ev_c)
                ids.record_stream(cur)
                if ids.dtype != torch.int64:
                    ids = ids.to(torch.int64)                      # widened on the device: ~10 us
                emb = model.encode_document(ids)
                if res is None:
                    res = torch.empty((n, emb.shape[1]), dtype=torch.float32, device=device)
                res[i[0]:i[0] + emb.shape[0]].copy_(emb)
                # a pinned batch must outlive its async copy; keep two batches in flight so the GPU never waits for the host
                ev = torch.cuda.Event()
                ev.record(cur)
                inflight.append((ids_host, ev, buf))
                if len(inflight) > 2:
                    done = inflight.popleft()
                    done[1].synchronize()
                    free.put(done[2])
        finally:
            # Hand the staging buffers back only when nothing can write or read them any more: producer jobs that have not started
            # are cancelled, the ones that are waiting for a buffer are released with a sentinel, the running ones are waited for
            # (fut.cancel() does not stop a running job: it would go on writing into a block the next call already owns), and
            # the copies / encoder calls queued on the device are drained.
            for _, fut in pending:
                fut.cancel()
            for _ in pending:
                free.put(None)
            for _, fut in pending:
                if not fut.cancelled():
                    try:
                        fut.result()
                    except BaseException:  # noqa: BLE001 -- (its own failure, or the sentinel's: the call is ending anyway)
                        pass
            if pending or staged:          # (only after a failure: a completed build has consumed everything)
                torch.cuda.synchronize(device)
            while inflight:
                inflight.popleft()[1].synchronize()
            _give_staging(ring_key, mine)
    return res


_STAGING_KEEP_BYTES = 768 << 20   # pinned host memory kept between builds (the default batch size's ring is 377 MB)


def _take_staging(key, count: int, cap: int) -> list:
    """`count` pinned int64 buffers of `cap` elements that belong to the caller until _give_staging: from the pool, else fresh."""
    with _STAGING_LOCK:
        pool = _PINNED_RINGS.setdefault(key, [])
        mine = [pool.pop() for _ in range(min(count, len(pool)))]
    while len(mine) < count:
        mine.append(torch.empty(cap, dtype=torch.int64, pin_memory=True))
    return mine


def _give_staging(key, bufs: list) -> None:
    """Back into the pool, most recently used size first; buffers beyond _STAGING_KEEP_BYTES (other batch sizes' first) are freed."""
    with _STAGING_LOCK:
        pool = _PINNED_RINGS.pop(key, [])
        pool.extend(bufs)
        _PINNED_RINGS[key] = pool          # (re-inserted: most recently used key last)
        total = sum(b.numel() * 8 for p_ in _PINNED_RINGS.values() for b in p_)
        for k in list(_PINNED_RINGS):
            p_ = _PINNED_RINGS[k]
            while p_ and total > _STAGING_KEEP_BYTES:
                total -= p_.pop().numel() * 8
            if not p_ and k != key:
                del _PINNED_RINGS[k]


_PINNED_RINGS: Dict = {}
_STAGING_LOCK = __import__("threading").Lock()
_PRODUCER_POOLS: Dict = {}


def corpus_recall_hit(query_emb: torch.Tensor, doc_embeddings: torch.Tensor, positives: Sequence[int],
                      top_k: Sequence[int] = (1, 5, 10)) -> Dict[str, float]:
    """Recall@k = found positives / available positives, Hit@k = any positive found (evaluators.py:197-207).
    `positives` are row indices of doc_embeddings."""
    pos = set(int(p) for p in positives)
    if not pos:
        return {}
    _, idx = score_topk(query_emb.reshape(1, -1), doc_embeddings, max(top_k))
    top = [int(i) for i in idx[0].tolist()]
    res = {}
    for k in top_k:
        found = len([i for i in top[:k] if i in pos])
        res[f"Recall@{k}"] = found / len(pos)
        res[f"Hit@{k}"] = 1 if found else 0
    return res


def save_inference_artifacts(output_dir, model, config: Dict, tokenizer, documents, device=None, tfidf: bool = True) -> np.ndarray:
    """backend/main.py:92-149: model.pth, config.json (+VOCAB_SIZE, EMBED_DIM), word_to_idx.pkl, documents.pkl,
    document_embeddings.npy ([N,H] fp32, C order, row i <-> documents[i]) and tfidf_artifacts.pkl ({'vectorizer', 'matrix'}:
    sklearn on the CPU, exactly the reference's call; skipped with tfidf=False or without sklearn).

    `documents`: the reference's fifth argument -- `datasets`, a dict split -> [(query, pos_doc, neg_doc)] whose unique
    documents are collected as main.py:115-121 does -- or a plain sequence of document strings.  `device`: where the towers
    run; default: the device of the model's parameters (the reference reads the caller-set `model.device`)."""
    import json
    import pickle
    from pathlib import Path
    out = Path(output_dir)
    out.mkdir(parents=True, exist_ok=True)
    torch.save(model.state_dict(), out / "model.pth")
    cfg = dict(config)
    cfg["VOCAB_SIZE"] = tokenizer.vocab_size()
    cfg["EMBED_DIM"] = model.query_encoder.embedding.embedding_dim
    (out / "config.json").write_text(json.dumps(cfg, indent=4))
    with open(out / "word_to_idx.pkl", "wb") as f:
        pickle.dump({w: i for w, i in tokenizer.word2idx.items()}, f)
    if isinstance(documents, dict):
        all_docs = set()
        for split_data in documents.values():
            for _, pos_doc, neg_doc in split_data:
                all_docs.add(pos_doc)
                all_docs.add(neg_doc)
        docs = list(all_docs)
    else:
        docs = list(documents)
    if device is None:
        device = getattr(model, "device", None) or next(model.parameters()).device
    model.eval()
    # same rows either way (rows are independent); the pipelined build is the fast path for a real corpus
    if len(docs) > 4096 and torch.device(device).type == "cuda":
        emb = embed_corpus(model, tokenizer, docs, device).cpu().numpy()
    else:
        emb = embed_documents(model, tokenizer, docs, device, config.get("BATCH_SIZE", 64)).cpu().numpy()
    with open(out / "documents.pkl", "wb") as f:
        pickle.dump(docs, f)
    np.save(out / "document_embeddings.npy", emb)
    if tfidf and docs:
        try:
            from sklearn.feature_extraction.text import TfidfVectorizer
        except ImportError:
            TfidfVectorizer = None
        if TfidfVectorizer is not None:
            vec = TfidfVectorizer(stop_words="english", max_features=20000)  # main.py:141-142
            try:
                mat = vec.fit_transform(docs)
            except ValueError:  # (only stop words / empty vocabulary: nothing to save)
                mat = None
            if mat is not None:
                with open(out / "tfidf_artifacts.pkl", "wb") as f:
                    pickle.dump({"vectorizer": vec, "matrix": mat}, f)
    return emb


def _encode_texts(model, tokenizer, texts: Sequence[str], device: torch.device, which: str, batch_size: int = 64) -> torch.Tensor:
    """texts -> [n,H] embeddings in batches of `batch_size` rows padded to the batch maximum (evaluators.py:162-175)."""
    enc = model.encode_document if which == "doc" else model.encode_query
    out = []
    with torch.no_grad():
        for i in range(0, len(texts), batch_size):
            rows = [tokenizer.encode(t) for t in texts[i:i + batch_size]]
            width = max((len(r) for r in rows), default=0)
            ids = torch.zeros((len(rows), width), dtype=torch.long)
            for j, r in enumerate(rows):
                ids[j, :len(r)] = torch.as_tensor(r, dtype=torch.long)
            out.append(enc(ids.to(device)))
    return torch.cat(out) if out else torch.empty((0, 0), device=device)


class CorpusEvaluator:
    """Full-corpus evaluation with several positives per query (backend/evaluators.py:83-209)."""

    def __init__(self, top_k: List[int] = [1, 5, 10], max_candidates: int = 1000, max_queries: int = 50):
        self.top_k = top_k
        self.max_candidates = max_candidates
        self.max_queries = max_queries

    def evaluate(self, model, val_data: List[Tuple[str, str, str]], tokenizer, device: torch.device) -> Dict[str, float]:
        model.eval()
        query_to_positives: Dict[str, set] = {}
        all_docs = set()
        for query, pos_doc, neg_doc in val_data:
            query_to_positives.setdefault(query, set()).add(pos_doc)
            all_docs.add(pos_doc)
            all_docs.add(neg_doc)
        unique_queries = list(query_to_positives.keys())
        unique_docs = list(all_docs)
        if len(unique_docs) > self.max_candidates:  # (same draw as the reference: evaluators.py:124-126)
            unique_docs = random.sample(unique_docs, self.max_candidates)
            print(f"  Using {self.max_candidates} candidate documents for evaluation")
        print(f"\nCorpus evaluation: {len(unique_queries)} unique queries against {len(unique_docs)} documents...")
        doc_embeddings = _encode_texts(model, tokenizer, unique_docs, device, "doc")
        sample_queries = random.sample(unique_queries, min(self.max_queries, len(unique_queries)))  # evaluators.py:138
        metrics: Dict[str, List[float]] = {f"Recall@{k}": [] for k in self.top_k}
        metrics.update({f"Hit@{k}": [] for k in self.top_k})
        if sample_queries and len(unique_docs):
            doc_pos = {doc: i for i, doc in enumerate(unique_docs)}
            # one query at a time through the tower, like the reference (a padded batch would feed the same rows: padding
            # never reaches the recurrence), then one fused score + top-k for all of them
            q_embs = torch.cat([_encode_texts(model, tokenizer, [q], device, "query") for q in sample_queries])
            kmax = min(max(self.top_k), len(unique_docs))
            _, top = score_topk(q_embs, doc_embeddings, kmax)
            top = top.tolist()
            for qi, query in enumerate(sample_queries):
                known = query_to_positives[query]
                available = [doc for doc in known if doc in doc_pos]
                if not available:
                    continue  # no positive among the candidates: the reference skips the query (:191-192)
                pos_idx = {doc_pos[doc] for doc in available}
                for k in self.top_k:
                    found = len([i for i in top[qi][:k] if i in pos_idx])
                    metrics[f"Recall@{k}"].append(found / len(available))
                    metrics[f"Hit@{k}"].append(1 if found else 0)
        return {name: (float(np.mean(vals)) if vals else 0.0) for name, vals in metrics.items()}


class TestEvaluator:
    """Qualitative print-out on the test triplets (backend/evaluators.py:212-283); returns nothing, like the reference."""

    __test__ = False  # (not a pytest class)

    def __init__(self, num_examples: int = 10, top_k: int = 5):
        self.num_examples = num_examples
        self.top_k = top_k

    def evaluate(self, model, test_data: List[Tuple[str, str, str]], tokenizer, device: torch.device) -> None:
        model.eval()
        all_queries = {t[0] for t in test_data}
        all_docs = {t[1] for t in test_data}.union({t[2] for t in test_data})
        ground_truth: Dict[str, set] = {}
        for query, pos_doc, _ in test_data:
            ground_truth.setdefault(query, set()).add(pos_doc)
        unique_queries, unique_docs = list(all_queries), list(all_docs)
        print(f"\nTest evaluation: {len(unique_queries)} queries, {len(unique_docs)} documents...")
        doc_embs = _encode_texts(model, tokenizer, unique_docs, device, "doc")
        sample_queries = random.sample(unique_queries, min(self.num_examples, len(unique_queries)))
        print("\n" + "=" * 80 + f"\nQUALITATIVE EXAMPLES (Top {self.top_k})\n" + "=" * 80)
        if not sample_queries or not len(unique_docs):
            return
        q_embs = torch.cat([_encode_texts(model, tokenizer, [q], device, "query") for q in sample_queries])
        vals, idx = score_topk(q_embs, doc_embs, min(self.top_k, len(unique_docs)))
        vals, idx = vals.tolist(), idx.tolist()
        for i, query in enumerate(sample_queries):
            print(f"\n--- Example {i + 1}/{len(sample_queries)} ---\nQuery: {query}\n\nTop {self.top_k} retrieved documents:")
            truth = ground_truth.get(query, set())
            hits = 0
            for rank, (score, di) in enumerate(zip(vals[i], idx[i])):
                doc = unique_docs[di]
                good = doc in truth
                hits += int(good)
                print(f"  {rank + 1}. {'[+]' if good else '[-]'} {doc[:100]}... (Score: {score:.4f})")
            print(f"\nSummary: found {hits}/{len(truth)} ground truth positives in Top {self.top_k}.")
