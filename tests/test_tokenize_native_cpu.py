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


def _long_word_vocab(rs, n=4000):
    """Words of 1 .. 40 bytes; many share their first 8 or 16 bytes (the native table keeps 8 key bytes in a slot and compares
    the rest in the key blob; the scan takes words of up to 16 bytes from a 16-byte window and longer ones byte by byte)."""
    alphabet = "abcdefghijklmnopqrstuvwxyz0123456789_"
    stems = ["".join(rs.choice(alphabet) for _ in range(L)) for L in (8, 8, 16, 16, 7, 15, 9, 17)]
    words = {}
    while len(words) < n:
        L = rs.choice([1, 2, 3, 5, 7, 8, 9, 12, 15, 16, 17, 18, 24, 31, 32, 33, 40])
        w = "".join(rs.choice(alphabet) for _ in range(L))
        if rs.random() < 0.4:
            w = (rs.choice(stems) + w)[:max(L, 9)]
        words.setdefault(w, len(words))
    for p in ".,!?;":
        words.setdefault(p, len(words))
    return words


def test_long_keys_shared_prefixes_and_every_window_alignment(monkeypatch):
    from twotowermlretrieval_amd import tokenizer as T
    rs = random.Random(11)
    vocab = _long_word_vocab(rs)
    vocab["big" + "x" * 20] = 7_000_000_000            # a long key whose id needs 64 bits
    tok = PretrainedTokenizer(word2idx=vocab)
    keys = list(vocab)
    texts = []
    for i in range(1500):
        parts = [" " * rs.randint(0, 17)]               # every start offset inside a 16-byte window
        for _ in range(rs.randint(0, 30)):
            w = rs.choice(keys)
            r = rs.random()
            if r < 0.25:
                w = w.upper()
            elif r < 0.35:
                w = w[:-1] + "Q" if len(w) > 1 else w   # near miss: same length, same prefix
            elif r < 0.45:
                w = w + "z"                             # near miss: one byte longer
            parts.append(w + rs.choice([" ", "  ", ",", ". ", "-", "\t", "!?", ";", " " * 15, " " * 16, " " * 33]))
        texts.append("".join(parts)[:rs.randint(0, 400)] if rs.random() < 0.3 else "".join(parts))
    texts += ["a" * 16, "a" * 17, "a" * 16 + " " + "b" * 16, "x" * 31, "x" * 32, "x" * 33, "_" * 8, "9" * 9]
    want = tok.encode_batch(texts, native=False)
    got = tok.encode_batch(texts)                        # list of ASCII str: pointer form when csrc/pytext.c is built
    assert got.shape == want.shape and bool((got == want).all())
    assert bool((tok.encode_batch(tuple(texts), n_threads=3) == want).all())
    monkeypatch.setattr(T, "_GATHER", None)              # the one-join form
    assert bool((tok.encode_batch(texts) == want).all())
    mixed = texts + [None, "é " + texts[3]]              # the offsets form (a non-str element, a non-ASCII text)
    assert bool((tok.encode_batch(mixed) == tok.encode_batch(mixed, native=False)).all())


def test_pointer_form_is_the_one_taken_for_lists_of_str():
    """csrc/pytext.c is built by twotowermlretrieval_amd.build: lists and tuples of str are read in place (no join): where the
    code units lie, how many, and their size (0 = ASCII bytes, 1 / 2 / 4 = Latin-1 / UCS-2 / UCS-4)."""
    from twotowermlretrieval_amd import tokenizer as T
    gather = T._pytext_gather()
    assert gather is not None, "twotowermlretrieval_amd/_pytext*.so is missing: python -m twotowermlretrieval_amd.build"
    ptrs, lens, units = np.zeros(5, np.uint64), np.zeros(5, np.int64), np.zeros(5, np.uint8)
    a = (ptrs.ctypes.data, lens.ctypes.data, units.ctypes.data)
    assert gather(["ab", "", "cde"], *a) == (3, 5, 0) and lens[:3].tolist() == [2, 0, 3] and units[:3].tolist() == [0, 0, 0]
    assert gather(["ab", "é", "\u2019x", "\U0001F600"], *a) == (4, 6, 3) and units[:4].tolist() == [0, 1, 2, 4]
    assert gather(("ab", 7), *a)[0] == 1                               # stops at the first item that is not a str
    import pytest
    with pytest.raises(TypeError):
        gather("ab", *a)


def test_text_beyond_ascii_is_tokenised_natively_with_the_interpreters_own_tables():
    """Non-ASCII texts no longer send a batch through Python: tt_tok_encode_units reads CPython's 1-, 2- and 4-byte code units and
    classifies / lower-cases with tables made from THIS interpreter's str.lower() and str.isalnum() (tokenizer._unicode_tables), so
    the ids equal re.findall(r"\\w+|[.,!?;]", text.lower()) -> word2idx by construction -- checked here on Latin-1, Greek (capital
    sigma: context-dependent lower case -> that text goes through Python), Cyrillic, CJK, digits of other scripts, combining
    marks, typographic punctuation, U+0130 (lower-cases to two code points -> Python), astral code points, lone surrogates."""
    import re
    from twotowermlretrieval_amd import tokenizer as T
    low, cls = T._unicode_tables()
    rs = random.Random(21)
    for cp in rs.sample(range(0x110000), 30000) + list(range(0x2000)):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        ch = chr(cp)
        lw = ch.lower()
        assert (low[cp] == 0xFFFFFFFF) == (len(lw) != 1 or cp == 0x3A3)
        assert bool(cls[cp] == 1) == bool(re.fullmatch(r"\w", ch)), hex(cp)
    pieces = ["the", "Straße", "STRASSE", "naïve", "NAÏVE", "Ελληνικά", "ΟΔΟΣ", "σίγμα", "Привет", "МИР", "日本語", "テスト", "한국어",
              "٣٤٥", "x²", "İstanbul", "ǅ", "ǆ", "ẞ", "ﬁ", "e\u0301", "don\u2019t", "\u201cquoted\u201d", "a\u2013b", "😀", "𝔘𝔫𝔦", "𐐀𐐨",
              "_under_", "MiXeD123", "…", "¿qué?", "50%", "a.b,c!d?e;f", "\ud800", "w" * 20, "Ω" * 9, "é" * 17, "\u00a0", "\u3000"]
    vocab = {}
    for w in pieces + ["don", "t", "quoted", "a", "b", "qué", "50", "c", "d", "e", "f", "e\u0301".lower(), "😀"]:
        for tok in re.findall(r"\w+|[.,!?;]", w.lower()):
            vocab.setdefault(tok, len(vocab))
    vocab = {k: v for k, v in vocab.items() if "\ud800" not in k}
    tok = PretrainedTokenizer(word2idx=vocab)
    seps = [" ", "  ", ", ", ".", "!", "\t", "\u2014", "\u00a0", "-", "/", "(", ")", "\n", "\u3000", "?"]
    texts = []
    for i in range(3000):
        texts.append("".join(rs.choice(pieces) + rs.choice(seps) for _ in range(rs.randint(0, 25))))
    texts += ["", "é", "Σ", "ΑΣ ΑΣΑ", "İ", "\U0001F600", "ascii only text, here."]
    want = tok.encode_batch(texts, native=False)
    got = tok.encode_batch(texts)
    assert got.shape == want.shape and bool((got == want).all())
    assert bool((tok.encode_batch(tuple(texts), n_threads=3) == want).all())
    assert getattr(tok, "_tok_unicode", False)                           # the native Unicode path was taken
    mixed = texts + [None]                                               # a non-str element: the general form, same ids
    assert bool((tok.encode_batch(mixed) == tok.encode_batch(mixed, native=False)).all())


def test_a_key_listed_twice_resolves_to_the_later_id():
    """tt_tok_create takes (keys, ids) arrays: a key that occurs twice -- a dict cannot hold one, a table built from a word list
    can -- resolves to the LATER id, as a dict filled in order would; for short keys (id in the slot) and long ones (side table)."""
    import ctypes as C
    from twotowermlretrieval_amd import _lib
    L = _lib.lib()
    keys = [b"the", b"averyveryverylongword", b"mid_length_k", b"the", b"averyveryverylongword", b"mid_length_k", b"x"]
    ids = np.asarray([1, 2, 3, 40, 50, 60, 7], dtype=np.int64)
    off = np.zeros(len(keys) + 1, dtype=np.int64)
    np.cumsum([len(k) for k in keys], out=off[1:])
    blob = b"".join(keys)
    h = C.c_void_p()
    _lib.check(L.tt_tok_create(blob, off.ctypes.data, ids.ctypes.data, len(keys), 99, C.byref(h)))
    text = b"The averyveryverylongword mid_length_k x averyveryverylongworD2 mid_length_"
    toff = np.asarray([0, len(text)], dtype=np.int64)
    ragged = np.zeros(len(text), dtype=np.int64)
    n, st = np.zeros(1, np.int32), np.zeros(1, np.int32)
    _lib.check(L.tt_tok_encode(h, text, toff.ctypes.data, 1, ragged.ctypes.data, n.ctypes.data, st.ctypes.data, 1))
    L.tt_tok_destroy(h)
    assert st[0] == 0 and ragged[:n[0]].tolist() == [40, 50, 60, 7, 99, 99]


def test_int32_batches_in_the_callers_block():
    """encode_batch(out=pinned block, ids32=True): the padded batch as int32 in the caller's block (half the bytes for the copy
    to the device; evaluators.embed_corpus widens it there); a vocabulary with an id beyond int32 gets the int64 batch."""
    import torch
    tok = PretrainedTokenizer(word2idx={"the": 0, "a": 1, "cat": 2})
    buf = torch.zeros(64, dtype=torch.int64)
    texts = ["the cat", "a a a cat zz", ""]
    t = tok.encode_batch(texts, out=buf, ids32=True)
    assert t.dtype == torch.int32 and t.data_ptr() == buf.data_ptr() and t.tolist() == tok.encode_batch(texts).tolist()
    big = PretrainedTokenizer(word2idx={"the": 0, "a": 5_000_000_000})
    t = big.encode_batch(["the a"], out=buf, ids32=True)
    assert t.dtype == torch.int64 and t.tolist() == [[0, 5_000_000_000]]


def test_concurrent_producers_share_one_tokenizer():
    """Several threads take a FRESH tokenizer through its first calls at once (evaluators.embed_corpus's producers): one native
    table and one Unicode hand-over, same ids as the Python path from every thread, and nobody deadlocks."""
    import threading
    vocab = {"the": 0, "café": 1, "a": 2}
    texts = ["the café a", "a the", "café’s"] * 200
    want = PretrainedTokenizer(word2idx=vocab).encode_batch(texts, native=False).tolist()
    tok = PretrainedTokenizer(word2idx=vocab)
    wrong = []

    def work():
        for _ in range(10):
            if tok.encode_batch(texts, n_threads=2).tolist() != want:
                wrong.append(1)
    threads = [threading.Thread(target=work, daemon=True) for _ in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert not any(t.is_alive() for t in threads) and not wrong
