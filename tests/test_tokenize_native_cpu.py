"""The native batch tokenizer (csrc/tokenize.hip: tt_tok_encode / tt_tok_pad) must give exactly the ids of the
Python path, which is pinned to the reference by tests/golden/g8_tokenizer.json."""
import json
import random
from pathlib import Path

import numpy as np

from twotowermlretrieval_amd.tokenizer import PretrainedTokenizer

GOLD = Path(__file__).parent / "golden"


def _tok(n_words=3000):
    words = ["the", ",", ".", "of", "and", "!", "?", ";"] + [f"w{i}" for i in range(8, n_words)] + \
            ["don", "t", "e", "mail", "naïve", "straße", "x_y", "42", "The"]
    return PretrainedTokenizer(word2idx={w: i for i, w in enumerate(words)}), words


def test_native_equals_python_on_the_golden_strings():
    g = json.loads((GOLD / "g8_tokenizer.json").read_text())
    tok = PretrainedTokenizer(word2idx=g["vocab"])
    texts = [c["text"] for c in g["cases"]]
    a = tok.encode_batch(texts, native=False)
    b = tok.encode_batch(texts, native=True)
    assert a.shape == b.shape and a.tolist() == b.tolist()
    for c, row in zip(g["cases"], b.tolist()):
        assert row[:len(c["ids"])] == c["ids"] and not any(row[len(c["ids"]):])


def test_native_equals_python_on_random_text():
    tok, words = _tok()
    rs = random.Random(5)
    seps = [" ", "  ", "\t", "\n", "-", "'", "/", ", ", ". ", "! ", "?", ";", "(", ")", "\"", "…", "é", " — "]
    texts = []
    for i in range(3000):
        parts = []
        for _ in range(rs.randint(0, 60)):
            w = rs.choice(words)
            if rs.random() < 0.3:
                w = w.upper() if rs.random() < 0.5 else w.capitalize()
            if rs.random() < 0.05:
                w = "zzz" + w  # unknown word
            parts.append(w + rs.choice(seps))
        texts.append("".join(parts))
    texts += ["", " ", None, 12345, "İstanbul ǅ ß", "a" * 5000, "...!!!", "_", "x_y__z 9_9"]
    a = tok.encode_batch(texts, native=False)
    for threads in (1, 3):
        b = tok.encode_batch(texts, native=True, n_threads=threads)
        assert a.shape == b.shape and bool((a == b).all())


def test_empty_batches():
    tok, _ = _tok(50)
    assert tuple(tok.encode_batch([]).shape) == (0, 0)
    assert tuple(tok.encode_batch(["", ""]).shape) == (2, 0)
    assert tok.encode_batch(["", "the of"]).tolist() == [[0, 0], [0, 3]]


def test_fast_path_separator_form_equals_python():
    """All-ASCII str batches take the one-join form (tt_tok_encode_sep: boundaries found natively); a text that holds the
    separator byte, a non-str element or a non-ASCII character sends the whole batch through the offsets form.  Same ids
    either way; words longer than the tokenizer's 64-byte stack buffer; a caller-owned output buffer."""
    import torch
    tok, words = _tok()
    rs = random.Random(9)
    ascii_texts = [" ".join(rs.choice(words[:-9]).upper() if rs.random() < 0.2 else rs.choice(words[:-9]) for _ in range(rs.randint(0, 40)))
                   + rs.choice(["", ".", " !?", ";;"]) for _ in range(2000)]
    ascii_texts += ["", "q" * 300 + " the " + "W" * 70, "a,b.c!d?e;f", "   "]
    want = tok.encode_batch(ascii_texts, native=False)
    got = tok.encode_batch(ascii_texts)
    assert got.shape == want.shape and bool((got == want).all())
    with_nul = list(ascii_texts)
    with_nul[7] = "the\x00of and"                     # \x00 is not a word character: two tokens either side of it
    want = tok.encode_batch(with_nul, native=False)
    got = tok.encode_batch(with_nul)
    assert got.shape == want.shape and bool((got == want).all())
    buf = torch.empty(len(ascii_texts) * 400, dtype=torch.int64)
    view = tok.encode_batch(ascii_texts, out=buf)
    assert view.data_ptr() == buf.data_ptr() and bool((view == tok.encode_batch(ascii_texts, native=False)).all())
    small = torch.empty(8, dtype=torch.int64)           # too small: a fresh tensor, same ids
    assert tok.encode_batch(ascii_texts, out=small).data_ptr() != small.data_ptr()


def test_ids_beyond_int32_and_duplicate_keys():
    """The native table keeps ids inside its 16-byte slots when they fit an int32 and in a side array when they do not; a
    vocabulary that lists a key twice (a dict cannot, a pickled list-built table can) resolves to the later id, as a dict
    built in order would."""
    big = {"the": 0, "alpha": 5_000_000_000, "beta": -7, "gamma": 3}
    tok = PretrainedTokenizer(word2idx=big)
    texts = ["alpha beta gamma delta the", "GAMMA alpha"]
    assert tok.encode_batch(texts).tolist() == tok.encode_batch(texts, native=False).tolist()
    assert tok.encode_batch(texts).tolist()[0][0] == 5_000_000_000


def test_host_cores_is_positive_and_at_most_the_affinity():
    import os
    from twotowermlretrieval_amd.tokenizer import host_cores
    assert 1 <= host_cores() <= len(os.sched_getaffinity(0))
