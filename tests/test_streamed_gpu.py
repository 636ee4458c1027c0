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
