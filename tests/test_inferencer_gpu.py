"""QueryInferencer drop-in (backend/query_inferencer.py:20-82) against the reference's recorded
outputs on a synthetic artifacts directory (tests/golden/g9_inferencer.npz)."""
import json
import pickle

import numpy as np
import pytest
import torch

import synth
from conftest import assert_fwd_close

pytestmark = pytest.mark.gpu


def _artifacts(tmp_path, g):
    from twotowermlretrieval_amd.model import TwoTowerModel
    V, E, H, seed = [int(x) for x in g["dims"]]
    vocab = json.loads(str(g["vocab_json"]))
    with open(tmp_path / "word_to_idx.pkl", "wb") as f:
        pickle.dump(vocab, f)
    cfg = {"HIDDEN_DIM": H, "RNN_TYPE": "GRU", "NUM_LAYERS": 1, "BIDIRECTIONAL": False, "DROPOUT": 0.0,
           "NORMALIZE_OUTPUT": True, "EMBED_DIM": E, "VOCAB_SIZE": V}
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    table = synth.make_table(seed, V, E)
    m = TwoTowerModel(dict(cfg), table)
    sd = {}
    for i, tower in enumerate(("query_encoder.", "doc_encoder.")):
        sd[tower + "embedding.weight"] = torch.from_numpy(table)
        sd.update({k: torch.from_numpy(v) for k, v in synth.make_encoder_state(seed + 10 + i, E, H, prefix=tower).items()})
    m.load_state_dict(sd)
    torch.save(m.state_dict(), tmp_path / "model.pth")  # the artifact format of backend/main.py:98
    return tmp_path


def test_query_inferencer_matches_reference(tmp_path, golden):
    from twotowermlretrieval_amd.query_inferencer import QueryInferencer
    g = golden("g9_inferencer.npz")
    inf = QueryInferencer(str(_artifacts(tmp_path, g)))
    for q, want in zip(g["queries"], g["embs"]):
        e = inf.get_query_embedding(str(q))
        assert e.shape == want.shape and e.dtype == np.float32
        assert_fwd_close(e, want)
        assert abs(np.linalg.norm(e) - 1.0) < 1e-5          # the reference's own self-check (:98)
    z = inf.get_query_embedding("")
    assert z.shape == g["empty"].shape and not z.any()       # un-tokenisable query -> zero vector (:66-69)
    assert str(g["the_the_error"]).startswith("RuntimeError")
    with pytest.raises(RuntimeError, match="Length of all samples"):
        inf.get_query_embedding("the the")                   # ids [0,0]: the reference raises too
    batch = inf.get_query_embeddings([str(q) for q in g["queries"]] + [""])
    assert_fwd_close(batch[:-1].cpu().numpy(), g["embs"])
    assert not batch[-1].any()


def test_concurrent_callers_share_an_encoder_and_an_index(tmp_path):
    """The reference serves /search from a thread pool (frontend/main.py:103): eight threads, each on its own HIP
    stream, hammer ONE query tower and ONE index; every answer must equal the single-threaded one."""
    import threading
    import twotowermlretrieval_amd as tt
    from twotowermlretrieval_amd import index as _index
    _index.SCREEN_MIN_DOCS = 0
    V, E, H = 300, 20, 256
    torch.manual_seed(0)
    m = tt.TwoTowerModel({"HIDDEN_DIM": H, "VOCAB_SIZE": V, "EMBED_DIM": E}, synth.make_table(5, V, E)).cuda().eval()
    D = torch.from_numpy(synth.unit_rows(7, 70000, 256)).cuda()
    ix = tt.BruteForceIndex(D, screen=True)
    ids = [torch.from_numpy(synth.make_ids(100 + t, 3, 4 + t, V)).cuda() for t in range(8)]
    with torch.no_grad():
        want = []
        for x in ids:
            q = m.encode_query(x)
            want.append(tuple(t.clone() for t in ix.search(q, 10)))
    torch.cuda.synchronize()
    errs = []

    def work(t):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s), torch.no_grad():
                for _ in range(20):
                    q = m.encode_query(ids[t])
                    v, i = ix.search(q, 10)
                    s.synchronize()
                    if not (torch.equal(v, want[t][0]) and torch.equal(i, want[t][1])):
                        errs.append(t)
                        return
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
