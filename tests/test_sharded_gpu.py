"""ShardedIndex on the GPU with a real RCCL communicator (single rank: the collective, the byte
packing and the merge kernel all run; multi-rank equality is covered on CPU over gloo in
tests/test_host_cpu.py and by construction: every rank merges the same gathered candidates)."""
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

import synth

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def nccl_group():
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("screen,B", [(False, 37), (True, 160)])
def test_sharded_index_through_rccl(nccl_group, oracle, screen, B):
    import twotowermlretrieval_amd as tt
    Q = synth.unit_rows(3, B, 256)
    D = synth.unit_rows(4, 7001, 256)
    idx = tt.ShardedIndex.from_global(torch.from_numpy(D).cuda(), shard_k=50, screen=screen)
    v, i = idx.search(torch.from_numpy(Q).cuda(), k=10)
    torch.cuda.synchronize()
    ov, oi = oracle.score_topk(Q, D, 10)
    assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(v.cpu().numpy(), ov)


def test_fused_optimizer_allreduce_through_rccl(nccl_group):
    from twotowermlretrieval_amd.trainer import FusedClipAdam
    p = [torch.nn.Parameter(torch.randn(50, 7, device="cuda")), torch.nn.Parameter(torch.randn(9, device="cuda"))]
    ref = [x.detach().clone() for x in p]
    opt = FusedClipAdam(p, lr=1e-2, max_norm=1.0)
    opt.zero_grad()
    for x in p:
        x.grad.add_(torch.ones_like(x))
    opt.step()
    torch.cuda.synchronize()
    assert all((a.detach() - b).abs().max() > 0 for a, b in zip(p, ref))
    assert abs(opt.total_norm.item() - (50 * 7 + 9) ** 0.5) < 1e-3
