// Host-side text front end of the index build: tokenise -> ids -> right-padded int64 batch, so that
// feeding the document tower (about 100 M tokens/s on one MI355X) is not bound by a Python loop.  No device code.
//
// Semantics: exactly PretrainedTokenizer.encode of the reference (backend/tokenizer.py:41-43) for ASCII text:
//     tokens = re.findall(r"\w+|[.,!?;]", str(text).lower());  ids = word2idx.get(token, unk_id)
// For ASCII, lower() maps A-Z to a-z and \w is [A-Za-z0-9_].  A text holding any byte >= 0x80 is NOT tokenised
// here (Unicode \w and case mapping are Python's business): its status is set to 1 and the Python caller encodes
// that one text itself, so results are identical by construction (tests/test_tokenize_native_cpu.py).
#include "tt_common.h"

#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// One 16-byte slot per key: everything a probe needs sits in the slot's cache line (tag = high hash bits, length, where the
// key's bytes are, the id) -- a hit costs the slot's line plus the key's bytes, a miss on an occupied slot only the slot (the
// first table kept word numbers in the slots and looked length, bytes and id up in three more arrays: four dependent cache
// misses per token on a 400 k-word GloVe vocabulary).
struct TokSlot {
    uint32_t tag;   // hash >> 32
    uint32_t off;   // offset of the key in blob
    int32_t len;    // key length in bytes, -1 = empty slot
    int32_t val;    // the id (ids32) or the word number (index into ids)
};

struct TokTable {
    std::vector<char> blob;          // all words, back to back
    std::vector<int64_t> ids;        // id of word i (only read when !ids32)
    std::vector<TokSlot> slots;      // open addressing, linear probing
    uint64_t mask = 0;
    int64_t unk = 0;
    bool ids32 = true;               // every id fits an int32: ids live in the slots
};

constexpr uint64_t FNV_BASIS = 1469598103934665603ull, FNV_PRIME = 1099511628211ull;
inline uint64_t fnv_finish(uint64_t h) { return h ^ (h >> 29); }
inline uint64_t fnv1a(const char *p, size_t n)
{
    uint64_t h = FNV_BASIS;
    for (size_t i = 0; i < n; ++i) {
        h ^= (unsigned char)p[i];
        h *= FNV_PRIME;
    }
    return fnv_finish(h);
}

inline int64_t lookup_hashed(const TokTable &t, const char *p, size_t n, uint64_t h)
{
    const uint32_t tag = (uint32_t)(h >> 32);
    uint64_t s = h & t.mask;
    for (;;) {
        const TokSlot &sl = t.slots[s];
        if (sl.len < 0)
            return t.unk;
        if (sl.tag == tag && (size_t)sl.len == n && std::memcmp(t.blob.data() + sl.off, p, n) == 0)
            return t.ids32 ? (int64_t)sl.val : t.ids[(size_t)sl.val];
        s = (s + 1) & t.mask;
    }
}

inline bool is_word(unsigned char c)
{
    return (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9') || c == '_' || (c >= 'A' && c <= 'Z');
}
inline bool is_punct(unsigned char c) { return c == '.' || c == ',' || c == '!' || c == '?' || c == ';'; }

// one text -> ids at out[0..]; returns the token count, or -1 when the text is not pure ASCII
int64_t encode_one(const TokTable &t, const char *s, size_t n, int64_t *out, std::vector<char> &lower)
{
    for (size_t i = 0; i < n; ++i)
        if ((unsigned char)s[i] >= 0x80)
            return -1;
    int64_t cnt = 0;
    size_t i = 0;
    char small[64];
    while (i < n) {
        const unsigned char c = (unsigned char)s[i];
        if (is_word(c)) {
            // lower-case copy and hash in ONE pass over the word (a stack buffer for words of up to 64 bytes)
            size_t j = i;
            while (j < n && is_word((unsigned char)s[j]))
                ++j;
            const size_t len = j - i;
            char *buf = small;
            if (len > sizeof small) {
                lower.resize(len);
                buf = lower.data();
            }
            uint64_t h = FNV_BASIS;
            for (size_t q = 0; q < len; ++q) {
                const char ch = s[i + q];
                const char lc = (ch >= 'A' && ch <= 'Z') ? (char)(ch + 32) : ch;
                buf[q] = lc;
                h ^= (unsigned char)lc;
                h *= FNV_PRIME;
            }
            out[cnt++] = lookup_hashed(t, buf, len, fnv_finish(h));
            i = j;
        } else if (is_punct(c)) {
            out[cnt++] = lookup_hashed(t, s + i, 1, fnv1a(s + i, 1));
            ++i;
        } else {
            ++i;
        }
    }
    return cnt;
}

template <class F>
void parallel_for(int64_t n, int n_threads, F &&f)
{
    if (n_threads <= 1 || n < 2 * n_threads) {
        f(0, n);
        return;
    }
    std::vector<std::thread> th;
    const int64_t per = (n + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
        const int64_t lo = t * per, hi = lo + per < n ? lo + per : n;
        if (lo >= hi)
            break;
        th.emplace_back([&f, lo, hi] { f(lo, hi); });
    }
    for (auto &x : th)
        x.join();
}

} // namespace

TT_EXPORT int tt_tok_create(const char *words_blob, const int64_t *word_off, const int64_t *word_ids, int64_t n_words,
                            int64_t unk_id, void **handle)
{
    if (!handle || n_words < 0 || (n_words > 0 && (!words_blob || !word_off || !word_ids)) || n_words > (1ll << 30) ||
        (n_words > 0 && word_off[n_words] > (int64_t)0xffffffffll))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_tok_create: n_words=%lld (or more than 4 GiB of keys)", (long long)n_words);
    TokTable *t = new TokTable;
    t->unk = unk_id;
    t->ids.assign(word_ids, word_ids + n_words);
    t->blob.assign(words_blob, words_blob + (n_words ? word_off[n_words] : 0));
    for (int64_t w = 0; w < n_words; ++w)
        if (word_ids[w] < INT32_MIN || word_ids[w] > INT32_MAX)
            t->ids32 = false;
    uint64_t cap = 16;
    while (cap < (uint64_t)n_words * 2 + 1)
        cap <<= 1;
    t->mask = cap - 1;
    t->slots.assign(cap, TokSlot{0u, 0u, -1, 0});
    for (int64_t w = 0; w < n_words; ++w) {
        const char *p = t->blob.data() + word_off[w];
        const size_t n = (size_t)(word_off[w + 1] - word_off[w]);
        const uint64_t h = fnv1a(p, n);
        const TokSlot fresh{(uint32_t)(h >> 32), (uint32_t)word_off[w], (int32_t)n, t->ids32 ? (int32_t)word_ids[w] : (int32_t)w};
        uint64_t s = h & t->mask;
        bool dup = false;
        while (t->slots[s].len >= 0) {
            const TokSlot &o = t->slots[s];
            if (o.tag == fresh.tag && (size_t)o.len == n && std::memcmp(t->blob.data() + o.off, p, n) == 0) {
                dup = true; // the same key twice: the later entry wins, as in a dict built in order
                t->slots[s] = fresh;
                break;
            }
            s = (s + 1) & t->mask;
        }
        if (!dup)
            t->slots[s] = fresh;
    }
    *handle = t;
    return TT_OK;
}

TT_EXPORT void tt_tok_destroy(void *handle) { delete (TokTable *)handle; }

TT_EXPORT int tt_tok_encode(const void *handle, const char *text_blob, const int64_t *text_off, int64_t n_texts,
                            int64_t *ragged_ids, int32_t *lens, int32_t *status, int n_threads)
{
    if (!handle || n_texts < 0 || (n_texts > 0 && (!text_blob || !text_off || !ragged_ids || !lens || !status)))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_tok_encode: n_texts=%lld", (long long)n_texts);
    const TokTable &t = *(const TokTable *)handle;
    parallel_for(n_texts, n_threads, [&](int64_t lo, int64_t hi) {
        std::vector<char> lower;
        lower.reserve(64);
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t c = encode_one(t, text_blob + text_off[i], (size_t)(text_off[i + 1] - text_off[i]),
                                         ragged_ids + text_off[i], lower);
            status[i] = c < 0 ? 1 : 0;
            lens[i] = c < 0 ? 0 : (int32_t)c;
        }
    });
    return TT_OK;
}

// The same for texts handed over as ONE blob with a separator byte between them (n_texts - 1 separators; the caller checks
// nothing: a text that contains the separator makes the count differ and the call fails with TT_ERR_BAD_SHAPE, and the caller
// takes the offsets form).  Saves the host the per-text length pass: "\0".join(texts).encode() is all the Python it needs.
// text_off_out [n_texts + 1]: start of text i in the blob; text i ends one byte (the separator) before text_off_out[i + 1],
// and ragged_ids / tt_tok_pad use these offsets as tt_tok_encode's.
TT_EXPORT int tt_tok_encode_sep(const void *handle, const char *text_blob, int64_t blob_len, char sep, int64_t n_texts,
                                int64_t *text_off_out, int64_t *ragged_ids, int32_t *lens, int32_t *status, int n_threads)
{
    if (!handle || n_texts < 0 || blob_len < 0 || (n_texts > 0 && (!text_blob || !text_off_out || !ragged_ids || !lens || !status)))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_tok_encode_sep: n_texts=%lld", (long long)n_texts);
    if (n_texts == 0)
        return blob_len == 0 ? TT_OK : tt_fail(TT_ERR_BAD_SHAPE, "tt_tok_encode_sep: %lld bytes for no text", (long long)blob_len);
    int64_t k = 0;
    text_off_out[k++] = 0;
    for (const char *q = text_blob, *end = text_blob + blob_len; q < end;) {
        const char *hit = (const char *)memchr(q, sep, (size_t)(end - q));
        if (!hit)
            break;
        if (k >= n_texts)
            return tt_fail(TT_ERR_BAD_SHAPE, "tt_tok_encode_sep: more than %lld separators (a text contains the separator byte)",
                           (long long)(n_texts - 1));
        text_off_out[k++] = (int64_t)(hit - text_blob) + 1;
        q = hit + 1;
    }
    if (k != n_texts)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_tok_encode_sep: %lld separators for %lld texts", (long long)(k - 1), (long long)n_texts);
    text_off_out[n_texts] = blob_len + 1; // (as if a separator followed the last text)
    const TokTable &t = *(const TokTable *)handle;
    parallel_for(n_texts, n_threads, [&](int64_t lo, int64_t hi) {
        std::vector<char> lower;
        lower.reserve(64);
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t c = encode_one(t, text_blob + text_off_out[i], (size_t)(text_off_out[i + 1] - 1 - text_off_out[i]),
                                         ragged_ids + text_off_out[i], lower);
            status[i] = c < 0 ? 1 : 0;
            lens[i] = c < 0 ? 0 : (int32_t)c;
        }
    });
    return TT_OK;
}

TT_EXPORT int tt_tok_pad(const int64_t *ragged_ids, const int64_t *text_off, const int32_t *lens, int64_t n_texts,
                         int64_t width, int64_t *out, int n_threads)
{
    if (n_texts < 0 || width < 0 || (n_texts > 0 && (!ragged_ids || !text_off || !lens || (width > 0 && !out))))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_tok_pad: n_texts=%lld width=%lld", (long long)n_texts, (long long)width);
    for (int64_t i = 0; i < n_texts; ++i)
        if (lens[i] > width)
            return tt_fail(TT_ERR_BAD_SHAPE, "tt_tok_pad: row %lld has %d tokens > width %lld", (long long)i, lens[i], (long long)width);
    parallel_for(n_texts, n_threads, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            int64_t *row = out + i * width;
            std::memcpy(row, ragged_ids + text_off[i], sizeof(int64_t) * (size_t)lens[i]);
            std::memset(row + lens[i], 0, sizeof(int64_t) * (size_t)(width - lens[i]));
        }
    });
    return TT_OK;
}
