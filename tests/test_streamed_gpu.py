"""BASELINE configs[4] path: bf16 corpus in pinned host memory streamed through the GPU in blocks
(double-buffered hipMemcpyAsync on a copy stream overlapped with scoring).  Results must equal the
oracle's exact top-k over the bf16 corpus widened to fp32."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,N,block,k,screen", [(5, 30000, 4096, 10, True), (130, 50001, 16384, 10, True),
                                                (33, 9000, 100000, 5, False), (200, 20000, 3000, 16, True)])
def test_streamed_equals_oracle_on_widened_corpus(oracle, B, N, block, k, screen):
    import twotowermlretrieval_amd as tt
    D = torch.from_numpy(synth.unit_rows(50 + N, N, 256)).to(torch.bfloat16)       # the corpus IS bf16
    Q = synth.unit_rows(60 + B, B, 256)
    ix = tt.StreamedIndex(D, block_docs=block, idx_offset=1000, screen=screen)
    v, i = ix.search(torch.from_numpy(Q).cuda(), k)
    torch.cuda.synchronize()
    ov, oi = oracle.score_topk(Q, D.to(torch.float32).numpy(), k, idx_offset=1000)
    assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(v.cpu().numpy(), ov)
    v2, i2 = ix.search(torch.from_numpy(Q).cuda(), k)                               # buffers are reusable
    assert torch.equal(i2, i) and torch.equal(v2, v)
    assert abs(ix.dmax_norm - float(D.to(torch.float32).norm(dim=1).max())) < 1e-5
    rv, ri = ix.resident().search(torch.from_numpy(Q).cuda(), k)                    # widened once into HBM: same answer
    assert torch.equal(ri, i) and torch.equal(rv, v)


def test_streamed_index_at_shard_scale_4m_rows(oracle):
    """configs[4] near its own size: a 4.2M-row bf16 corpus (2.1 GB) in PINNED host memory, streamed
    in 1M-row blocks (the last one ragged).  streamed == resident (widened once into HBM) == the plain fp32 kernel for every
    query of a large and of a serving-size batch, and == the CPU oracle over all rows for four of them."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    import twotowermlretrieval_amd as tt
    dev = torch.device("cuda:0")
    N = 4_200_003
    host = torch.empty((N, 256), dtype=torch.bfloat16).pin_memory()
    g = torch.Generator(device=dev).manual_seed(17)
    for lo in range(0, N, 1_000_000):
        hi = min(N, lo + 1_000_000)
        x = torch.randn((hi - lo, 256), device=dev, generator=g)
        x /= x.norm(dim=1, keepdim=True)
        host[lo:hi].copy_(x.to(torch.bfloat16))
    torch.cuda.synchronize()
    Q = torch.randn((200, 256), device=dev, generator=g)
    Q /= Q.norm(dim=1, keepdim=True)
    Q[3] = host[4_100_000].to(dev).to(torch.float32)          # a document of the LAST (ragged) block as a query
    Q[3] /= Q[3].norm()
    ix = tt.StreamedIndex(host, block_docs=1 << 20, idx_offset=7)
    assert ix.host.is_pinned() and ix._d16[0] is not None
    res = ix.resident()
    full32 = res.docs                                          # the corpus widened to fp32 (exact), resident
    for B in (200, 32):
        q = Q[:B].contiguous()
        sv, si = ix.search(q, 10)
        rv, ri = res.search(q, 10)
        ev, ei = tt.score_topk(q, full32, 10, idx_offset=7)
        torch.cuda.synchronize()
        assert torch.equal(si, ri) and torch.equal(sv, rv), B
        assert torch.equal(si, ei) and torch.equal(sv, ev), B
    assert int(si[3, 0]) == 4_100_000 + 7
    rows = [0, 3, 17, 31]
    d_np = full32.cpu().numpy()
    with ThreadPoolExecutor(min(4, len(os.sched_getaffinity(0)))) as ex:
        got = list(ex.map(lambda r: oracle.score_topk(Q[r:r + 1].cpu().numpy(), d_np, 10, idx_offset=7), rows))
    for r, (ov, oi) in zip(rows, got):
        assert np.array_equal(si[r].cpu().numpy(), oi[0]) and np.array_equal(sv[r].cpu().numpy(), ov[0]), r


def test_streamed_shard_at_configs4_size_12_5m_rows(oracle):
    """BASELINE configs[4] at ITS size for one GPU: 100M x 256 bf16 over 8 GPUs = a 12.5M-row shard (6.4 GB) in PINNED host
    memory, streamed in 1M-row blocks (13 of them, the last ragged), here as rank 5 of the 8 (row offset 62.5M).  For a bench
    batch (B = 1024) and a serving batch (B = 32): streamed == the same shard widened once into HBM == the plain fp32 kernel,
    bit for bit; the shard's list as the sharded search exchanges it (ShardedIndex over the streamed shard, per-shard
    top-50 -> top-10) gives the same top-10; four queries (two of them copies of documents of the first and of the last,
    ragged block) == the CPU oracle over all 12.5M rows."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    import twotowermlretrieval_amd as tt
    dev = torch.device("cuda:0")
    N, OFF = 12_500_000, 62_500_000
    host = torch.empty((N, 256), dtype=torch.bfloat16).pin_memory()
    g = torch.Generator(device=dev).manual_seed(23)
    for lo in range(0, N, 1_000_000):
        hi = min(N, lo + 1_000_000)
        x = torch.randn((hi - lo, 256), device=dev, generator=g)
        x /= x.norm(dim=1, keepdim=True)
        host[lo:hi].copy_(x.to(torch.bfloat16))
    del x
    torch.cuda.synchronize()
    Q = torch.randn((1024, 256), device=dev, generator=g)
    Q /= Q.norm(dim=1, keepdim=True)
    for r, doc in ((3, 12_499_990), (17, 5)):
        Q[r] = host[doc].to(dev).to(torch.float32)
        Q[r] /= Q[r].norm()
    ix = tt.StreamedIndex(host, block_docs=1 << 20, idx_offset=OFF)
    assert ix.host.is_pinned() and ix._d16[0] is not None and ix.host.data_ptr() == host.data_ptr()
    sh = tt.ShardedIndex(ix, OFF, shard_k=50)                 # one rank, no process group: the exchange is a copy
    assert sh.streamed and sh._seed_exchange is False
    res = ix.resident()
    full32 = res.docs
    kept = {}
    for B in (1024, 32):
        q = Q[:B].contiguous()
        sv, si = ix.search(q, 10)
        rv, ri = res.search(q, 10)
        ev, ei = tt.score_topk(q, full32, 10, idx_offset=OFF)
        hv, hi_ = sh.search(q, 10)
        torch.cuda.synchronize()
        assert torch.equal(si, ri) and torch.equal(sv, rv), B
        assert torch.equal(si, ei) and torch.equal(sv, ev), B
        assert torch.equal(hi_, si) and torch.equal(hv, sv), B
        assert int(si[3, 0]) == OFF + 12_499_990 and int(si[17, 0]) == OFF + 5
        kept[B] = (sv.cpu().numpy(), si.cpu().numpy())
    pv, pi = sh.submit(Q, 10).result()                        # the pipelined form over a streamed shard
    torch.cuda.synchronize()
    assert np.array_equal(pi.cpu().numpy(), kept[1024][1]) and np.array_equal(pv.cpu().numpy(), kept[1024][0])
    rows = [0, 3, 17, 1023]
    d_np = full32.cpu().numpy()
    del res, full32, ix, sh
    torch.cuda.empty_cache()
    with ThreadPoolExecutor(min(4, len(os.sched_getaffinity(0)))) as ex:
        got = list(ex.map(lambda r: oracle.score_topk(Q[r:r + 1].cpu().numpy(), d_np, 10, idx_offset=OFF), rows))
    for r, (ov, oi) in zip(rows, got):
        assert np.array_equal(kept[1024][1][r], oi[0]) and np.array_equal(kept[1024][0][r], ov[0]), r
        if r < 32:
            assert np.array_equal(kept[32][1][r], oi[0]) and np.array_equal(kept[32][0][r], ov[0]), r


def test_streamed_search_rejects_what_it_cannot_answer():
    import twotowermlretrieval_amd as tt
    D = torch.from_numpy(synth.unit_rows(5, 300, 256)).to(torch.bfloat16)
    ix = tt.StreamedIndex(D, block_docs=128)
    with pytest.raises(ValueError):
        ix.search(torch.zeros((2, 128), device="cuda"), 5)          # other width
    with pytest.raises(RuntimeError):
        ix.search(torch.zeros((2, 256)), 5)                         # CPU queries: no fallback
    with pytest.raises(ValueError):
        tt.ShardedIndex(ix, 9)                                      # row numbering of the index and of the shard disagree
    empty = tt.StreamedIndex(D[:0], block_docs=128, idx_offset=4)   # a rank whose shard is empty lists only padding
    v, i = empty.search(torch.from_numpy(synth.unit_rows(6, 3, 256)).cuda(), 4)
    assert bool((i == -1).all()) and bool(torch.isinf(v).all())


def test_streamed_and_sharded_streamed_on_random_shapes(oracle):
    """Fifteen seeded random (rows, block size, batch, k, shard-list length) combinations -- blocks of one row, blocks larger than the
    corpus, k larger than a block, k' < k, batches on both sides of the two screen forms: StreamedIndex and a ShardedIndex over it
    equal the oracle over the widened rows, bit for bit."""
    import twotowermlretrieval_amd as tt
    rs = np.random.RandomState(77)
    for trial in range(15):
        N = int(rs.choice([1, 7, 300, 4097, 20000, 70001]))
        block = int(rs.choice([1, 5, 256, 4096, 65536, 200000])) if N < 5000 else int(rs.choice([4096, 16384, 65536, 200000]))
        B = int(rs.choice([1, 3, 32, 33, 64, 65, 130]))
        k = int(rs.choice([1, 5, 10, 50, 64]))
        shard_k = int(rs.choice([1, 10, 50]))
        off = int(rs.randint(0, 10 ** 9))
        D = torch.from_numpy(synth.unit_rows(300 + trial, N, 256)).to(torch.bfloat16)
        Q = synth.unit_rows(400 + trial, B, 256)
        ix = tt.StreamedIndex(D, block_docs=block, idx_offset=off)
        v, i = ix.search(torch.from_numpy(Q).cuda(), k)
        sv, si = tt.ShardedIndex(ix, off, shard_k=shard_k).search(torch.from_numpy(Q).cuda(), k)
        torch.cuda.synchronize()
        ov, oi = oracle.score_topk(Q, D.to(torch.float32).numpy(), k, idx_offset=off)
        what = (trial, N, block, B, k, shard_k)
        assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(v.cpu().numpy(), ov), what
        assert np.array_equal(si.cpu().numpy(), oi) and np.array_equal(sv.cpu().numpy(), ov), what
