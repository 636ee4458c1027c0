"""Host-side text front end: the reference's PretrainedTokenizer semantics (backend/tokenizer.py:6-71)
plus the pad-to-batch step of its collate_fn (backend/main.py:50-56), producing the right-padded
int64 id batches the encoder kernels consume.  Plain CPU string work; no GPU code here.
`encode_batch` runs natively (libtt.so: tt_tok_encode / tt_tok_pad, multi-threaded, GIL released) for ASCII
texts and falls back to the Python `encode` for any text with a non-ASCII character, so the ids are identical
to the reference's by construction (the index build must not be bound by a Python loop).

Semantics kept exactly (tests/golden/g8_tokenizer.json):
  * tokens = re.findall(r"\\w+|[.,!?;]", str(text).lower())      -- str(None) == "none" is tokenised too
  * ids = word2idx.get(token, unk_id); '<UNK>' is appended at index len(vocab) when the pickle lacks it
  * no truncation, no padding inside encode(); pad value is 0 -- which is ALSO the id of the GloVe word
    "the" (SURVEY 8a quirk): the encoder counts non-zero ids as the length.
"""
from __future__ import annotations

import pickle
import re
import threading
from typing import Dict, Iterable, List, Sequence

import numpy as np

_TOKEN_RE = re.compile(r"\w+|[.,!?;]")
UNK = "<UNK>"

_SCRATCH = threading.local()


def host_cores() -> int:
    """CPU cores this process may really use: the scheduler affinity, capped by the cgroup's CPU quota (a container on a
    256-thread host with a 16-core share reports 256 in sched_getaffinity; 64 tokenizer threads on 16 cores thrash)."""
    import os
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


_GATHER = False


def _pytext_gather():
    """csrc/pytext.c's gather (built by twotowermlretrieval_amd.build next to libtt.so), or None when it was not built: the batch
    front end then joins the texts into one blob per batch, which gives the same ids."""
    global _GATHER
    if _GATHER is False:
        try:
            from . import _pytext
            _GATHER = _pytext.gather
        except ImportError:
            _GATHER = None
    return _GATHER


_UNICODE_TABLES = None
_UNICODE_LOCK = threading.RLock()   # (re-entrant: encode_batch takes it around _native(), which takes it too)


def _unicode_tables():
    """(low uint32 [0x110000], cls uint8 [0x110000]) -- THIS interpreter's answers, made once per process (~0.7 s), for
    tt_tok_set_unicode: low[cp] = ord(chr(cp).lower()) (0xffffffff where str.lower() has no context-free single-code-point
    answer: U+0130 lower-cases to two code points, U+03A3's lower case depends on its position in the word), cls[cp] = 1 where
    re's \\w matches chr(cp) (str.isalnum() or '_': sre's definition), 2 for the five marks of the pattern .,!?; , else 0.
    The native tokenizer then classifies exactly as re.findall(r"\\w+|[.,!?;]", text.lower()) does."""
    global _UNICODE_TABLES
    if _UNICODE_TABLES is None:
        n = 0x110000
        low = np.arange(n, dtype=np.uint32)
        cls = np.zeros(n, dtype=np.uint8)
        for cp in range(n):
            if 0xD800 <= cp <= 0xDFFF:
                continue   # (a lone surrogate: no case, no class)
            ch = chr(cp)
            lw = ch.lower()
            if len(lw) == 1:
                if lw != ch:
                    low[cp] = ord(lw)
            else:
                low[cp] = 0xFFFFFFFF
            if ch.isalnum():
                cls[cp] = 1
        low[0x3A3] = 0xFFFFFFFF          # capital sigma: 'σ' or 'ς' by context (Final_Sigma)
        cls[ord("_")] = 1
        for ch in ".,!?;":
            cls[ord(ch)] = 2
        _UNICODE_TABLES = (low, cls)
    return _UNICODE_TABLES


def _scratch(name: str, n: int, dtype) -> np.ndarray:
    """A per-thread array of at least n elements that is REUSED from call to call (grown geometrically).  The ragged id
    buffer of one 16k-document batch is ~50 MB; as a fresh np.empty every call it is ~12k first-touch page faults, which cost
    more than the tokenising and do not scale over threads (they serialise in the kernel): 145 ms cold against 10 ms warm for
    the same native call with 8 threads."""
    a = getattr(_SCRATCH, name, None)
    if a is None or a.shape[0] < n:
        a = np.empty(max(int(n * 1.25), 1024), dtype=dtype)
        setattr(_SCRATCH, name, a)
    return a


class PretrainedTokenizer:
    def __init__(self, word_to_idx_path: str = None, word2idx: Dict[str, int] = None):
        if word2idx is None:
            with open(word_to_idx_path, "rb") as f:
                word2idx = pickle.load(f)
        self.word2idx = dict(word2idx)
        self.unk_token = UNK
        if UNK not in self.word2idx:
            self.word2idx[UNK] = len(self.word2idx)
        self.unk_token_id = self.word2idx[UNK]
        self.idx2word = {i: w for w, i in self.word2idx.items()}

    # -- reference API ---------------------------------------------------------------
    def encode(self, sentence) -> List[int]:
        get, unk = self.word2idx.get, self.unk_token_id
        return [get(tok, unk) for tok in _TOKEN_RE.findall(str(sentence).lower())]

    def decode(self, token_ids: Iterable[int]) -> str:
        return " ".join(self.idx2word.get(int(i), UNK) for i in token_ids)

    def vocab_size(self) -> int:
        return len(self.word2idx)

    def get_word_index(self, word: str) -> int:
        return self.word2idx.get(word, -1)

    def get_index_word(self, index: int) -> str:
        return self.idx2word.get(index, UNK)

    def contains_word(self, word: str) -> bool:
        return word in self.word2idx

    # -- batch front end (pad_sequence(batch_first=True, padding_value=0), main.py:50-56) ---------
    def _native(self):
        """Handle of the native vocabulary table (built once, on first use)."""
        if getattr(self, "_tok_handle", None) is None:
            with _UNICODE_LOCK:   # (one table per tokenizer even when several producer threads get here together)
                if getattr(self, "_tok_handle", None) is None:
                    self._tok_handle = self._make_native()
        return self._tok_handle

    def _make_native(self):
        import ctypes as C
        from . import _lib
        words = list(self.word2idx.keys())
        enc = [w.encode("utf-8") if isinstance(w, str) else str(w).encode("utf-8") for w in words]
        off = np.zeros(len(enc) + 1, dtype=np.int64)
        np.cumsum([len(b) for b in enc], out=off[1:])
        blob = b"".join(enc)
        ids = np.asarray([self.word2idx[w] for w in words], dtype=np.int64)
        h = C.c_void_p()
        _lib.check(_lib.lib().tt_tok_create(blob, off.ctypes.data, ids.ctypes.data, len(enc), int(self.unk_token_id), C.byref(h)))
        return h

    def __del__(self):
        h = getattr(self, "_tok_handle", None)
        if h is not None:
            try:
                from . import _lib
                _lib.lib().tt_tok_destroy(h)
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass

    def encode_batch(self, texts: Sequence, pin: bool = False, native: bool = True, n_threads: int = 0, out=None, ids32: bool = False):
        """texts -> right-padded int64 tensor [B, max_len] (at least one column when B > 0 ... zero columns for
        all-empty batches, like pad_sequence).
        out: a 1-D int64 tensor (typically pinned, reused by the caller: evaluators.embed_corpus keeps a ring of them) that
        receives the batch when it is large enough; the result is then a view of it.  A fresh pinned tensor per batch costs a
        page-locking allocation whenever the host allocator has no free block of that size.
        ids32 (with `out`, fast forms only): the batch as int32 in the caller's block when every id fits (else int64 as usual)."""
        import torch
        if not native:
            rows = [self.encode(t) for t in texts]
            width = max((len(r) for r in rows), default=0)
            out = np.zeros((len(rows), width), dtype=np.int64)
            for i, r in enumerate(rows):
                out[i, :len(r)] = r
            t = torch.from_numpy(out)
            return t.pin_memory() if pin else t
        import os
        from . import _lib
        L = _lib.lib()
        n = len(texts)
        nt = n_threads or min(16, host_cores())
        # Fast form (every text a str, all ASCII, none holding a NUL -- the common case): ONE join + encode under the GIL, the
        # text boundaries are found natively (tt_tok_encode_sep).  What Python does per batch is then ~3 ms for 16k passages, which
        # is what bounds several producer threads (evaluators.embed_corpus) once the native part is spread over enough cores.
        fast = None
        gather = _pytext_gather()
        if n and gather is not None and type(texts) in (list, tuple):
            # Fastest form: the texts are read where the interpreter keeps them (csrc/pytext.c collects one pointer, one
            # length and the code-unit size per str, ~0.25 ms per 16 k passages under the GIL; tt_tok_encode_units does the rest
            # without it).  The str objects stay referenced by this frame's tuple for the duration of the call.  Texts beyond
            # ASCII are tokenised natively too, with this interpreter's own Unicode tables (_unicode_tables); the few that
            # hold a code point whose lower case depends on context come back with status 1 and go through self.encode.
            ptrs = _scratch("ptrs", n, np.uint64)
            tlen = _scratch("tlen", n, np.int64)
            units = _scratch("units", n, np.uint8)
            if type(texts) is list:
                texts = tuple(texts)   # (the pointers stay valid even if another thread edits the caller's list meanwhile: ~40 us)
            n_ok, total, beyond = gather(texts, ptrs.ctypes.data, tlen.ctypes.data, units.ctypes.data)
            if n_ok == n:
                if beyond and not getattr(self, "_tok_unicode", False):
                    with _UNICODE_LOCK:   # (producer threads share a tokenizer: the tables are handed over once, by one of them)
                        if not getattr(self, "_tok_unicode", False):
                            low, cls = _unicode_tables()
                            _lib.check(L.tt_tok_set_unicode(self._native(), low.ctypes.data, cls.ctypes.data, low.shape[0]))
                            self._tok_unicode = True
                off = _scratch("off", n + 1, np.int64)
                ragged = _scratch("ragged", total + 1, np.int64)   # (a text never has more tokens than code points)
                lens = _scratch("lens", n, np.int32)
                status = _scratch("status", n, np.int32)
                _lib.check(L.tt_tok_encode_units(self._native(), ptrs.ctypes.data, tlen.ctypes.data, units.ctypes.data, n,
                                                 off.ctypes.data, ragged.ctypes.data, lens.ctypes.data, status.ctypes.data, nt))
                if beyond:
                    for i in np.flatnonzero(status[:n]):   # context-dependent lower case: Python's own str.lower() and re
                        ids = self.encode(texts[i])
                        lens[i] = len(ids)
                        ragged[off[i]:off[i] + len(ids)] = ids
                fast = (off, ragged, lens)
        if n and fast is None:
            try:
                blob = "\x00".join(texts).encode("ascii")
                off = _scratch("off", n + 1, np.int64)
                ragged = _scratch("ragged", len(blob) + 1, np.int64)   # (a text never has more tokens than bytes)
                lens = _scratch("lens", n, np.int32)
                status = _scratch("status", n, np.int32)
                rc = L.tt_tok_encode_sep(self._native(), blob, len(blob), b"\x00", n, off.ctypes.data, ragged.ctypes.data,
                                         lens.ctypes.data, status.ctypes.data, nt)
                if rc == _lib.TT_OK:
                    fast = (off, ragged, lens)
                elif rc != _lib.TT_ERR_BAD_SHAPE:   # (BAD_SHAPE: a text contains the separator -- the general form below)
                    _lib.check(rc)
            except (TypeError, UnicodeEncodeError):
                pass
        if fast is not None:
            off, ragged, lens = fast
            width = int(lens[:n].max())
            if ids32 and width and out is not None and out.dtype == torch.int64 and out.dim() == 1 and 2 * out.numel() >= n * width:
                # 4-byte ids into the caller's (pinned) block: half the bytes for the copy to the device (embed_corpus widens
                # them there); a vocabulary with ids beyond int32 takes the 8-byte form below
                t32 = out.view(torch.int32)[:n * width].view(n, width)
                rc = L.tt_tok_pad_i32(ragged.ctypes.data, off.ctypes.data, lens.ctypes.data, n, width, t32.data_ptr(), nt)
                if rc == _lib.TT_OK:
                    return t32
                if rc != _lib.TT_ERR_BAD_INDEX:
                    _lib.check(rc)
            if out is not None and out.numel() >= n * width and out.dtype == torch.int64 and out.dim() == 1:
                t = out[:n * width].view(n, width)
            else:
                t = torch.empty((n, width), dtype=torch.int64, pin_memory=bool(pin) and n * width > 0)
            if width:
                _lib.check(L.tt_tok_pad(ragged.ctypes.data, off.ctypes.data, lens.ctypes.data, n, width, t.data_ptr(), nt))
            return t
        strs = [t if type(t) is str else str(t) for t in texts]
        off = np.zeros(n + 1, dtype=np.int64)
        try:
            # all-ASCII batch: ONE encode of the joined text, byte lengths = character lengths
            blob = "".join(strs).encode("ascii")
            if n:
                np.cumsum(np.fromiter(map(len, strs), dtype=np.int64, count=n), out=off[1:])
        except UnicodeEncodeError:
            enc = [t.encode("utf-8", "surrogatepass") for t in strs]   # (any byte >= 0x80 sends the text to self.encode anyway)
            if n:
                np.cumsum([len(b) for b in enc], out=off[1:])
            blob = b"".join(enc)
        total = int(off[-1])
        ragged = _scratch("ragged", max(total, 1), np.int64)   # (a text never has more tokens than bytes)
        lens = _scratch("lens", max(n, 1), np.int32)
        status = _scratch("status", max(n, 1), np.int32)
        lens[:max(n, 1)] = 0
        status[:max(n, 1)] = 0
        _lib.check(L.tt_tok_encode(self._native(), blob, off.ctypes.data, n, ragged.ctypes.data, lens.ctypes.data,
                                   status.ctypes.data, nt))
        slow = np.flatnonzero(status[:n])
        extra = {}
        for i in slow:  # non-ASCII text: Python's Unicode \w and lower()
            ids = self.encode(texts[i])
            extra[int(i)] = ids
            lens[i] = len(ids)
        width = int(lens[:n].max()) if n else 0
        for i, ids in extra.items():  # a text never has more tokens than UTF-8 bytes: its ragged slot is large enough
            ragged[off[i]:off[i] + len(ids)] = ids
        t = torch.empty((n, width), dtype=torch.int64, pin_memory=bool(pin) and n * width > 0)
        if n and width:
            _lib.check(L.tt_tok_pad(ragged.ctypes.data, off.ctypes.data, lens.ctypes.data, n, width, t.data_ptr(), nt))
        return t
