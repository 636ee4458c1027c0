"""The screened search on a corpus it can LOSE on (backend/evaluators.py:185-186 over the outputs of a
trained tower, backend/model.py:71-74): MS MARCO has near-duplicate passages and a trained encoder clusters them, so the 2-eps
slack of the fp16 filter covers a whole cluster, and a group of EXACT duplicates larger than k puts more tied documents in front
than any list can hold.  bench.make_clustered_corpus: 1 333 clustered centres (~150 rows each that agree to ~1e-3; two random rows
have cosine ~0.6) + 30 groups of 1 100 exact duplicates; a quarter of the queries are copies of duplicated rows.

What must hold whatever the filter does: values and indices bit-identical to the CPU oracle, tied scores in ascending index
order, for both forms of the screen.  What the filter does is asserted too: the duplicate-group queries overflow the survivor
list (1 100 documents tied at the top > SURV_MAX = 1 024) and their 32-query tiles are recomputed by the exact kernel -- predicated on the
device flag, no host round trip; a tile falls back EXACTLY when one of its queries overflowed a capacity limit of the proof (a
plain query whose cluster happens to host a duplicate group does too), never otherwise -- a cluster of ~150 near-duplicates alone
(more than round 3's 256-entry survivor list could take together with a small duplicate group) is rescored by the screen itself."""
import numpy as np
import pytest
import torch

from test_encoder_corpus_gpu import _oracle_topk

pytestmark = pytest.mark.gpu


def test_clustered_corpus_with_duplicate_groups_is_bit_identical_to_the_oracle(oracle):
    import bench
    import twotowermlretrieval_amd as tt
    dev = torch.device("cuda:0")
    N, B, k = 200_000, 160, 10
    D, Q, members = bench.make_clustered_corpus(N, B, dev, seed=5, n_centres=1333, dup_groups=30, dup=1100)
    samp = D[torch.randint(0, N, (2048,), device=dev)]
    mean_cos = float(((samp @ samp.t()).sum() - 2048) / (2048 * 2047))
    assert mean_cos >= 0.5, mean_cos
    n_dupq = min(members.shape[0], B // 4)                    # queries 0 .. n_dupq-1 are copies of duplicated rows
    ix = tt.BruteForceIndex(D, screen=True)
    assert ix.docs16 is not None
    ix.keep_stats = True
    Dn, Qn = D.cpu().numpy(), Q.cpu().numpy()
    ov, oi = _oracle_topk(oracle, Qn, Dn, k)
    # the duplicate groups ARE the top of their queries' lists: k tied scores, lowest indices of the group first
    for qi in range(n_dupq):
        grp = np.sort(members[qi].cpu().numpy())
        assert len(set(ov[qi])) == 1 and list(oi[qi]) == list(grp[:k]), qi
    report = {}
    for nq in (B, 40):          # shared-tile form (B > 64) and streaming form (B <= 64)
        v, i = ix.search(Q[:nq].contiguous(), k)
        torch.cuda.synchronize()
        vi, ii = v.cpu().numpy(), i.cpu().numpy()
        assert np.array_equal(ii, oi[:nq]) and np.array_equal(vi, ov[:nq])
        same = vi[:, 1:] == vi[:, :-1]
        assert (ii[:, 1:][same] > ii[:, :-1][same]).all()     # ties: index ascending
        st = ix.search_stats().cpu().numpy()
        flags = ix.fallback_flags.cpu().numpy() != 0
        report[nq] = dict(pooled_mean=float(st[:, 0].mean()), pooled_max=int(st[:, 0].max()), surv_mean=float(st[:, 1].mean()),
                          surv_max=int(st[:, 1].max()), fallback_tiles=int(flags.sum()), tiles=len(flags))
        # a tile falls back exactly when one of its queries overflowed a capacity limit of the proof
        over = (st[:, 1] > 1024) | (st[:, 0] > 8192)
        want = np.array([over[t * 32:(t + 1) * 32].any() for t in range(len(flags))])
        assert np.array_equal(flags, want), report
        assert flags[: (min(nq, n_dupq) + 31) // 32].all(), report      # every tile that holds a duplicate-group query
        assert st[:, 1].min() >= k, report
    ev, ei = tt.score_topk(Q, D, k)                          # the plain fp32 kernel agrees too
    assert np.array_equal(ei.cpu().numpy(), oi) and np.array_equal(ev.cpu().numpy(), ov)
    print("clustered-corpus filter statistics:", report)
