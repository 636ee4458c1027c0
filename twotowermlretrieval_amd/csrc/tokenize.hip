// Host-side text front end of the index build: tokenise -> ids -> right-padded int64 batch, so that
// feeding the document tower (about 100 M tokens/s on one MI355X) is not bound by a Python loop.  No device code.
//
// Semantics: exactly PretrainedTokenizer.encode of the reference (backend/tokenizer.py:41-43):
//     tokens = re.findall(r"\w+|[.,!?;]", str(text).lower());  ids = word2idx.get(token, unk_id)
// For ASCII, lower() maps A-Z to a-z and \w is [A-Za-z0-9_]: the byte entry points (tt_tok_encode, _sep, _ptrs) tokenise ASCII
// text and hand a text with any byte >= 0x80 back (status 1: the Python caller encodes that one text itself).  Beyond ASCII the
// answers are the host interpreter's: tt_tok_set_unicode takes ITS lower-case and word-class tables and tt_tok_encode_units reads
// its strings' 1-, 2- or 4-byte code units in place -- identical ids by construction either way (tests/test_tokenize_native_cpu.py).
#include "tt_common.h"

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#if defined(__SSE2__) && !defined(TT_TOK_NO_SIMD)
#include <emmintrin.h>
#endif
#include <condition_variable>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <thread>
#include <vector>

// Build-time knobs (tools/experiments/tok_harness.sh times the variants on the host in question; none is read at run time)
#ifndef TT_TOK_LAG
#define TT_TOK_LAG 16        // tokens between a slot's prefetch and its read (4 ... 48 within 8 % of each other where measured)
#endif
#ifndef TT_TOK_HUGEPAGES
#define TT_TOK_HUGEPAGES 1   // ask for transparent huge pages under the slot table
#endif

namespace {

// One 16-byte slot per key, four per cache line: tag_len = the hash's high bits with the key length (capped at 31) in the low five,
// the key's first 8 bytes, and the id.  A token of up to 8 bytes -- most of a text -- is resolved from the slot alone; a longer
// one whose tag, length and first 8 bytes agree is compared with the key's remaining bytes through `longs`.  A 400 k-word
// vocabulary is a 16 MB table.  (The first table kept word numbers in the slots and looked length, bytes and id up in three more
// arrays: four dependent cache misses per token; the second kept 16-byte slots and every key in the blob: two.)
struct TokSlot {
    uint32_t tag_len; // 0 = empty slot (no key has length 0)
    int32_t val;      // len <= 8: the id (ids32) or the word number (index into ids); len > 8: index into longs
    uint64_t k0;      // the key's bytes 0 .. 7 (little-endian, zero-padded)
};
static_assert(sizeof(TokSlot) == 16, "four slots per cache line");
struct LongKey {
    uint32_t off, len; // the key in blob
    int64_t id;
};

// The slot table on transparent huge pages where the kernel grants them (madvise mode included): with 4 KB pages nearly every
// probe of a 16 MB table also misses the TLB, and a software prefetch that has to walk the page table first is no prefetch.
template <class T>
struct HugeArray {
    T *p = nullptr;
    size_t n = 0;
    HugeArray() = default;
    HugeArray(const HugeArray &) = delete;
    HugeArray &operator=(const HugeArray &) = delete;
    ~HugeArray() { std::free(p); }
    bool assign(size_t count, const T &v)
    {
        std::free(p);
        p = nullptr;
        n = 0;
        const size_t huge = (size_t)2 << 20, raw = count * sizeof(T);
        const bool big = raw >= huge;             // (a toy vocabulary gets an ordinary allocation, not a 2 MB page)
        const size_t align = big ? huge : 64, bytes = (raw + align - 1) / align * align;
        void *q = nullptr;
        if (posix_memalign(&q, align, bytes ? bytes : align) != 0)
            return false;
#if defined(MADV_HUGEPAGE) && TT_TOK_HUGEPAGES
        if (big)
            madvise(q, bytes, MADV_HUGEPAGE); // (advice: refused or unsupported -> ordinary pages)
#endif
        p = (T *)q;
        n = count;
        for (size_t i = 0; i < count; ++i)
            p[i] = v;
        return true;
    }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

struct TokTable {
    std::vector<char> blob;          // all words, back to back
    std::vector<int64_t> ids;        // id of word i (only read when !ids32)
    HugeArray<TokSlot> slots;        // open addressing, linear probing
    std::vector<LongKey> longs;      // keys of more than 8 bytes
    uint64_t mask = 0;
    int64_t unk = 0;
    bool ids32 = true;               // every id fits an int32: ids live in the slots
    // Unicode (tt_tok_set_unicode: the host's OWN tables, so that the ids are the host's by construction): code point -> its
    // lower-case code point (TOK_NO_LOWER: no single-code-point answer without context -- U+0130, U+03A3: the text is handed back)
    // and the class (CH_WORD / CH_PUNCT / CH_OTHER) of every code point.  Empty until set: non-ASCII texts are handed back.
    std::vector<uint32_t> ulow;
    std::vector<unsigned char> ucls;
};
constexpr uint32_t TOK_NO_LOWER = 0xffffffffu;

// The hash of a key: its bytes as little-endian 64-bit words (the last one zero-padded), one multiply per word -- a word of up
// to 8 characters costs one round (a byte-serial FNV-1a was a chain of one 64-bit multiply PER BYTE).
constexpr uint64_t HASH_SEED = 0x243f6a8885a308d3ull, HASH_MUL = 0x9e3779b97f4a7c15ull, HASH_MUL2 = 0xd6e8feb86659fd93ull;
inline uint64_t load_le(const char *p, size_t n) // n <= 8 bytes, zero-padded
{
    uint64_t v = 0;
    std::memcpy(&v, p, n);
    return v;
}
inline uint64_t hash_round(uint64_t h, uint64_t w)
{
    h = (h ^ w) * HASH_MUL;
    return h ^ (h >> 32);
}
inline uint64_t hash_finish(uint64_t h, size_t n)
{
    h = (h ^ (uint64_t)n) * HASH_MUL2;
    return h ^ (h >> 29);
}
inline uint64_t hash_key(const char *p, size_t n)
{
    uint64_t h = HASH_SEED;
    size_t i = 0;
    for (; i + 8 <= n; i += 8)
        h = hash_round(h, load_le(p + i, 8));
    if (i < n)
        h = hash_round(h, load_le(p + i, n - i));
    return hash_finish(h, n);
}

// character classes of the reference's pattern  \w+|[.,!?;]  on lower-cased ASCII text (backend/tokenizer.py:41-43)
enum : unsigned char { CH_OTHER = 0, CH_WORD = 1, CH_PUNCT = 2, CH_HIGH = 3 };
struct CharTables {
    unsigned char cls[256];
    char low[256];
    CharTables()
    {
        for (int c = 0; c < 256; ++c) {
            const bool word = (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9') || c == '_' || (c >= 'A' && c <= 'Z');
            const bool punct = c == '.' || c == ',' || c == '!' || c == '?' || c == ';';
            cls[c] = c >= 0x80 ? CH_HIGH : word ? CH_WORD : punct ? CH_PUNCT : CH_OTHER;
            low[c] = (c >= 'A' && c <= 'Z') ? (char)(c + 32) : (char)c;
        }
    }
};
const CharTables g_chars;

inline uint32_t tag_len_of(uint64_t h, size_t n) { return ((uint32_t)(h >> 32) & ~31u) | (uint32_t)(n < 31 ? n : 31); }
inline int64_t short_value(const TokTable &t, const TokSlot &sl) { return t.ids32 ? (int64_t)sl.val : t.ids[(size_t)sl.val]; }

// a key of more than 16 bytes (p: the whole lower-cased key)
int64_t lookup_long(const TokTable &t, const char *p, size_t n, uint64_t h)
{
    const uint32_t tl = tag_len_of(h, n);
    const uint64_t k0 = load_le(p, 8);
    for (uint64_t s = h & t.mask;; s = (s + 1) & t.mask) {
        const TokSlot &sl = t.slots[s];
        if (sl.tag_len == 0)
            return t.unk;
        if (sl.tag_len == tl && sl.k0 == k0) {
            const LongKey &lk = t.longs[(size_t)sl.val];
            if (lk.len == n && std::memcmp(t.blob.data() + lk.off + 8, p + 8, n - 8) == 0)
                return lk.id;
        }
    }
}

// Tokens of up to 16 bytes are looked up LATE: the scan computes a token's hash, asks for its slot's cache line and goes on
// scanning; the slot is read LAG tokens later, when the line has arrived.  A 400 k-word vocabulary is a 16 MB table and MS MARCO
// text hits it all over: looked up in place, every token waited ~70 ns for memory (14 M tokens/s per thread) while the core
// could have ten such misses in flight.
struct Pending {
    uint64_t h, k0, k1;
    int64_t *dst;
    uint32_t len;
};
struct Lookups {
    static constexpr unsigned RING = 64, LAG = TT_TOK_LAG;
    const TokTable &t;
    Pending ring[RING];
    unsigned head = 0, tail = 0;
    explicit Lookups(const TokTable &tab) : t(tab) {}
    void resolve(const Pending &p)
    {
        const uint32_t tl = tag_len_of(p.h, p.len);
        for (uint64_t s = p.h & t.mask;; s = (s + 1) & t.mask) {
            const TokSlot &sl = t.slots[s];
            if (sl.tag_len == 0) {
                *p.dst = t.unk;
                return;
            }
            if (sl.tag_len == tl && sl.k0 == p.k0) {
                if (p.len <= 8) {
                    *p.dst = short_value(t, sl);
                    return;
                }
                const LongKey &lk = t.longs[(size_t)sl.val];
                if (lk.len == p.len && load_le(t.blob.data() + lk.off + 8, p.len - 8) == p.k1) {
                    *p.dst = lk.id;
                    return;
                }
            }
        }
    }
    void push(uint64_t h, uint64_t k0, uint64_t k1, uint32_t len, int64_t *dst)
    {
        __builtin_prefetch(&t.slots[h & t.mask]);
        ring[head++ % RING] = Pending{h, k0, k1, dst, len};
        if (head - tail > LAG)
            resolve(ring[tail++ % RING]);
    }
    void flush()
    {
        while (tail != head)
            resolve(ring[tail++ % RING]);
    }
};

// one text -> ids at out[0..] (the short tokens' ids arrive by lk.flush() at the latest); returns the token count, or -1 when the
// text is not pure ASCII (whatever was written to out is then meaningless)
inline int64_t encode_tail(const TokTable &t, Lookups &lk, const char *s, size_t i, size_t n, int64_t *out, int64_t cnt, std::vector<char> &lower)
{
    const unsigned char *cls = g_chars.cls;
    const char *low = g_chars.low;
    while (i < n) {
        const unsigned char c = (unsigned char)s[i];
        const unsigned char k = cls[c];
        if (k == CH_WORD) {
            alignas(8) char buf[16] = {0};
            size_t j = i;
            while (j < n && j - i < 16 && cls[(unsigned char)s[j]] == CH_WORD) {
                buf[j - i] = low[(unsigned char)s[j]];
                ++j;
            }
            if (j < n && j - i == 16 && cls[(unsigned char)s[j]] == CH_WORD) { // a long word: lower-cased copy, looked up in place
                while (j < n && cls[(unsigned char)s[j]] == CH_WORD)
                    ++j;
                const size_t len = j - i;
                lower.resize(len);
                for (size_t q = 0; q < len; ++q)
                    lower[q] = low[(unsigned char)s[i + q]];
                out[cnt++] = lookup_long(t, lower.data(), len, hash_key(lower.data(), len));
            } else {
                const size_t len = j - i;
                const uint64_t k0 = load_le(buf, 8), k1 = load_le(buf + 8, 8);
                uint64_t h = hash_round(HASH_SEED, k0);
                if (len > 8)
                    h = hash_round(h, k1);
                lk.push(hash_finish(h, len), k0, k1, (uint32_t)len, out + cnt++);
            }
            i = j;
        } else if (k == CH_PUNCT) {
            lk.push(hash_finish(hash_round(HASH_SEED, (uint64_t)c), 1), (uint64_t)c, 0, 1u, out + cnt++);
            ++i;
        } else if (k == CH_HIGH) {
            return -1;
        } else {
            ++i;
        }
    }
    return cnt;
}

inline uint64_t lower8(uint64_t x) // A-Z -> a-z in each byte < 0x80 (a byte >= 0x80 can only disturb the bytes ABOVE it)
{
    const uint64_t up = ((x + 0x3f3f3f3f3f3f3f3full) & ~(x + 0x2525252525252525ull)) & 0x8080808080808080ull;
    return x | (up >> 2);
}
inline uint64_t load8(const char *p)
{
    uint64_t v;
    std::memcpy(&v, p, 8);
    return v;
}

int64_t encode_one(const TokTable &t, Lookups &lk, const char *s, size_t n, int64_t *out, std::vector<char> &lower)
{
    int64_t cnt = 0;
    size_t i = 0;
#if defined(__SSE2__) && !defined(TT_TOK_NO_SIMD)
    // Windows of 16 bytes while 32 are left (a token's 16 key bytes are then readable wherever it starts in the window).  The
    // classes of a window as bit masks; every token that ENDS inside the window is taken from the masks (no byte loop, no
    // data-dependent load address between windows); a word that touches the window's end restarts the window at its first byte.
    // A window starts at a token boundary: behind a full step its predecessor byte was no word byte.
    const __m128i v_20 = _mm_set1_epi8(0x20), v_a1 = _mm_set1_epi8('a' - 1), v_z1 = _mm_set1_epi8('z' + 1), v_01 = _mm_set1_epi8('0' - 1),
                  v_91 = _mm_set1_epi8('9' + 1), v_us = _mm_set1_epi8('_');
    while (i + 32 <= n) {
        const __m128i v = _mm_loadu_si128((const __m128i *)(s + i));
        const __m128i lo = _mm_or_si128(v, v_20);  // (only the letter test reads it; signed compares: bytes >= 0x80 are negative)
        const __m128i alpha = _mm_and_si128(_mm_cmpgt_epi8(lo, v_a1), _mm_cmpgt_epi8(v_z1, lo));
        const __m128i digit = _mm_and_si128(_mm_cmpgt_epi8(v, v_01), _mm_cmpgt_epi8(v_91, v));
        const __m128i wordv = _mm_or_si128(_mm_or_si128(alpha, digit), _mm_cmpeq_epi8(v, v_us));
        const __m128i punct = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, _mm_set1_epi8('.')), _mm_cmpeq_epi8(v, _mm_set1_epi8(','))),
                                           _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, _mm_set1_epi8('!')), _mm_cmpeq_epi8(v, _mm_set1_epi8('?'))),
                                                        _mm_cmpeq_epi8(v, _mm_set1_epi8(';'))));
        const unsigned W = (unsigned)_mm_movemask_epi8(wordv), P = (unsigned)_mm_movemask_epi8(punct);
        if (_mm_movemask_epi8(v))
            return -1;
        unsigned T = (W & ~(W << 1)) | P; // first bytes of the window's tokens
        size_t step = 16;
        while (T) {
            const unsigned b = (unsigned)__builtin_ctz(T);
            T &= T - 1;
            const char *p = s + i + b;
            if (P >> b & 1u) {
                const uint64_t c = (unsigned char)*p;
                lk.push(hash_finish(hash_round(HASH_SEED, c), 1), c, 0, 1u, out + cnt++);
                continue;
            }
            const unsigned len = (unsigned)__builtin_ctz(~(W >> b)); // <= 16 - b: the bits above the window are clear in W
            if (b + len == 16) {      // touches the end of the window (it is the window's last token)
                if (b) {
                    step = b;
                    break;
                }
                if (g_chars.cls[(unsigned char)s[i + 16]] == CH_WORD) { // 17 bytes or more: lower-cased copy, looked up in place
                    size_t j = i + 16;
                    while (j < n && g_chars.cls[(unsigned char)s[j]] == CH_WORD)
                        ++j;
                    const size_t wl = j - i;
                    lower.resize(wl);
                    for (size_t q = 0; q < wl; ++q)
                        lower[q] = g_chars.low[(unsigned char)s[i + q]];
                    out[cnt++] = lookup_long(t, lower.data(), wl, hash_key(lower.data(), wl));
                    step = wl;
                    break;
                }
            }
            uint64_t k0 = lower8(load8(p)), k1 = 0;
            if (len < 8)
                k0 &= ~0ull >> (64 - 8 * len);
            else if (len > 8) {
                k1 = lower8(load8(p + 8));
                if (len < 16)
                    k1 &= ~0ull >> (128 - 8 * len);
            }
            uint64_t h = hash_round(HASH_SEED, k0);
            if (len > 8)
                h = hash_round(h, k1);
            lk.push(hash_finish(h, len), k0, k1, len, out + cnt++);
        }
        i += step;
    }
#endif
    return encode_tail(t, lk, s, i, n, out, cnt, lower);
}

// A text with code points beyond ASCII, as the host's string holds it: one code point per 1-, 2- or 4-byte unit (CPython's
// compact str kinds: Latin-1, UCS-2, UCS-4).  Per code point: the host's lower-case mapping, then the host's class of the result
// -- the reference lower-cases the text before it matches \w+|[.,!?;] (backend/tokenizer.py:41-43) -- and a token's key is the
// UTF-8 encoding of its lower-cased code points (what the vocabulary's keys are).  Returns the token count, or -1 when the text
// holds a code point whose lower case depends on context (the caller tokenises that text itself).
template <class U>
int64_t encode_units(const TokTable &t, Lookups &lk, const U *s, size_t n, int64_t *out, std::vector<char> &tok)
{
    const size_t n_cp = t.ulow.size();
    if (n_cp == 0)
        return -1;
    int64_t cnt = 0;
    tok.clear();
    auto flush = [&] {
        const size_t len = tok.size();
        if (len == 0)
            return;
        if (len <= 16) {
            char buf[16] = {0};
            std::memcpy(buf, tok.data(), len);
            const uint64_t k0 = load_le(buf, 8), k1 = load_le(buf + 8, 8);
            uint64_t h = hash_round(HASH_SEED, k0);
            if (len > 8)
                h = hash_round(h, k1);
            lk.push(hash_finish(h, len), k0, k1, (uint32_t)len, out + cnt++);
        } else {
            out[cnt++] = lookup_long(t, tok.data(), len, hash_key(tok.data(), len));
        }
        tok.clear();
    };
    for (size_t i = 0; i < n; ++i) {
        const uint32_t cp = (uint32_t)s[i];
        if (cp >= n_cp) { // (not a code point: no class)
            flush();
            continue;
        }
        const uint32_t lw = t.ulow[cp];
        if (lw == TOK_NO_LOWER)
            return -1;
        const unsigned char c = lw < n_cp ? t.ucls[lw] : (unsigned char)CH_OTHER;
        if (c == CH_WORD) {
            if (lw < 0x80) {
                tok.push_back((char)lw);
            } else if (lw < 0x800) {
                tok.push_back((char)(0xc0 | (lw >> 6)));
                tok.push_back((char)(0x80 | (lw & 0x3f)));
            } else if (lw < 0x10000) {
                tok.push_back((char)(0xe0 | (lw >> 12)));
                tok.push_back((char)(0x80 | ((lw >> 6) & 0x3f)));
                tok.push_back((char)(0x80 | (lw & 0x3f)));
            } else {
                tok.push_back((char)(0xf0 | (lw >> 18)));
                tok.push_back((char)(0x80 | ((lw >> 12) & 0x3f)));
                tok.push_back((char)(0x80 | ((lw >> 6) & 0x3f)));
                tok.push_back((char)(0x80 | (lw & 0x3f)));
            }
        } else {
            flush();
            if (c == CH_PUNCT) {
                const uint64_t k0 = (uint64_t)lw; // (the five punctuation marks are ASCII)
                lk.push(hash_finish(hash_round(HASH_SEED, k0), 1), k0, 0, 1u, out + cnt++);
            }
        }
    }
    flush();
    return cnt;
}

// Workers that outlive the call.  A batch is two parallel loops (encode, pad) of ~1 ms each; sixteen std::thread constructions and
// joins per loop were a quarter of that.  One pool per process, grown on demand, leaked on purpose (its detached workers wait on
// its condition variable until the process ends); a loop that finds the pool busy -- two producers at once -- spawns threads as
// before; a forked child starts with no pool.
class WorkerPool {
    std::mutex m;
    std::condition_variable work_cv, done_cv;
    const std::function<void(int)> *job = nullptr;
    int chunks = 0, next_chunk = 0, unfinished = 0, workers = 0;
    uint64_t generation = 0;
    std::mutex busy; // one loop at a time

    void worker()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            work_cv.wait(lk, [&] { return generation != seen && next_chunk < chunks; });
            seen = generation;
            while (next_chunk < chunks) {
                const int c = next_chunk++;
                const std::function<void(int)> *j = job;
                lk.unlock();
                (*j)(c);
                lk.lock();
                if (--unfinished == 0)
                    done_cv.notify_all();
            }
        }
    }

public:
    // runs fn(0 .. k - 1), the caller taking chunks too; false = the pool is in use (the caller spawns its own threads)
    bool run(int k, const std::function<void(int)> &fn)
    {
        std::unique_lock<std::mutex> one(busy, std::try_to_lock);
        if (!one.owns_lock())
            return false;
        std::unique_lock<std::mutex> lk(m);
        while (workers < k - 1 && workers < 63) {
            std::thread([this] { worker(); }).detach();
            ++workers;
        }
        job = &fn;
        chunks = k;
        next_chunk = 0;
        unfinished = k;
        ++generation;
        work_cv.notify_all();
        while (next_chunk < chunks) { // the caller works as well
            const int c = next_chunk++;
            lk.unlock();
            fn(c);
            lk.lock();
            --unfinished;
        }
        done_cv.wait(lk, [&] { return unfinished == 0; });
        job = nullptr;
        chunks = 0;
        return true;
    }
};

WorkerPool *g_pool = nullptr;
std::once_flag g_pool_atfork;
std::mutex g_pool_make;

WorkerPool *worker_pool()
{
    std::lock_guard<std::mutex> lk(g_pool_make);
    if (!g_pool) {
        std::call_once(g_pool_atfork, [] { pthread_atfork(nullptr, nullptr, [] { g_pool = nullptr; }); });
        g_pool = new WorkerPool; // (leaked: see above)
    }
    return g_pool;
}

template <class F>
void parallel_for(int64_t n, int n_threads, F &&f)
{
    if (n_threads <= 1 || n < 2 * n_threads) {
        f(0, n);
        return;
    }
    const int64_t per = (n + n_threads - 1) / n_threads;
    const int k = (int)((n + per - 1) / per);
    const std::function<void(int)> chunk = [&](int t) {
        const int64_t lo = t * per, hi = lo + per < n ? lo + per : n;
        if (lo < hi)
            f(lo, hi);
    };
    if (worker_pool()->run(k, chunk))
        return;
    std::vector<std::thread> th;
    for (int t = 0; t < k; ++t)
        th.emplace_back([&chunk, t] { chunk(t); });
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
    if (!t->slots.assign(cap, TokSlot{0u, 0, 0ull})) {
        delete t;
        return tt_fail(TT_ERR_WORKSPACE, "tt_tok_create: no memory for %llu slots", (unsigned long long)cap);
    }
    for (int64_t w = 0; w < n_words; ++w) {
        const char *p = t->blob.data() + word_off[w];
        const size_t n = (size_t)(word_off[w + 1] - word_off[w]);
        if (n == 0)
            continue; // (the empty string is no token of any text)
        const uint64_t h = hash_key(p, n);
        const uint32_t tl = tag_len_of(h, n);
        const uint64_t k0 = load_le(p, n < 8 ? n : 8);
        uint64_t s = h & t->mask;
        bool dup = false;
        for (; t->slots[s].tag_len != 0; s = (s + 1) & t->mask) {
            TokSlot &o = t->slots[s];
            if (o.tag_len != tl || o.k0 != k0)
                continue;
            if (n <= 8) { // the same key twice: the later entry wins, as in a dict built in order
                o.val = t->ids32 ? (int32_t)word_ids[w] : (int32_t)w;
                dup = true;
                break;
            }
            LongKey &lk = t->longs[(size_t)o.val];
            if (lk.len == n && std::memcmp(t->blob.data() + lk.off, p, n) == 0) {
                lk.id = word_ids[w];
                dup = true;
                break;
            }
        }
        if (dup)
            continue;
        if (n <= 8)
            t->slots[s] = TokSlot{tl, t->ids32 ? (int32_t)word_ids[w] : (int32_t)w, k0};
        else {
            t->slots[s] = TokSlot{tl, (int32_t)t->longs.size(), k0};
            t->longs.push_back(LongKey{(uint32_t)word_off[w], (uint32_t)n, word_ids[w]});
        }
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
        Lookups lk(t);
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t c = encode_one(t, lk, text_blob + text_off[i], (size_t)(text_off[i + 1] - text_off[i]),
                                         ragged_ids + text_off[i], lower);
            status[i] = c < 0 ? 1 : 0;
            lens[i] = c < 0 ? 0 : (int32_t)c;
        }
        lk.flush();
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
        Lookups lk(t);
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t c = encode_one(t, lk, text_blob + text_off_out[i], (size_t)(text_off_out[i + 1] - 1 - text_off_out[i]),
                                         ragged_ids + text_off_out[i], lower);
            status[i] = c < 0 ? 1 : 0;
            lens[i] = c < 0 ? 0 : (int32_t)c;
        }
        lk.flush();
    });
    return TT_OK;
}

// The same for texts that lie wherever the host keeps them (one pointer and one byte length per text: a host whose strings are
// separate objects -- CPython's str, Go's string, a Java byte[] -- hands them over without building a blob; what bounded several
// Python producer threads was the join + encode under the interpreter lock, 2.7 ms per 16 k passages).  text_off_out [n_texts + 1]
// receives the running sum of the lengths: text i's ids go to ragged_ids[text_off_out[i] ...] (capacity: the sum of the lengths),
// the layout tt_tok_pad reads.
TT_EXPORT int tt_tok_encode_ptrs(const void *handle, const char *const *texts, const int64_t *text_len, int64_t n_texts,
                                 int64_t *text_off_out, int64_t *ragged_ids, int32_t *lens, int32_t *status, int n_threads)
{
    if (!handle || n_texts < 0 || (n_texts > 0 && (!texts || !text_len || !text_off_out || !ragged_ids || !lens || !status)))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_tok_encode_ptrs: n_texts=%lld", (long long)n_texts);
    int64_t run = 0;
    for (int64_t i = 0; i < n_texts; ++i) {
        if (text_len[i] < 0 || (text_len[i] > 0 && !texts[i]))
            return tt_fail(TT_ERR_BAD_SHAPE, "tt_tok_encode_ptrs: text %lld: length %lld", (long long)i, (long long)text_len[i]);
        text_off_out[i] = run;
        run += text_len[i];
    }
    if (n_texts > 0)
        text_off_out[n_texts] = run;
    const TokTable &t = *(const TokTable *)handle;
    parallel_for(n_texts, n_threads, [&](int64_t lo, int64_t hi) {
        std::vector<char> lower;
        lower.reserve(64);
        Lookups lk(t);
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t c = encode_one(t, lk, texts[i], (size_t)text_len[i], ragged_ids + text_off_out[i], lower);
            status[i] = c < 0 ? 1 : 0;
            lens[i] = c < 0 ? 0 : (int32_t)c;
        }
        lk.flush();
    });
    return TT_OK;
}

// tt_tok_pad with 4-byte ids: half the bytes for the host-to-device copy of a batch (on this platform that copy is a shader
// kernel whose time adds to the encoder's; the device widens the batch again in ~10 us).  Fails with
// TT_ERR_BAD_INDEX when an id does not fit an int32 (the caller takes tt_tok_pad).
TT_EXPORT int tt_tok_pad_i32(const int64_t *ragged_ids, const int64_t *text_off, const int32_t *lens, int64_t n_texts,
                             int64_t width, int32_t *out, int n_threads)
{
    if (n_texts < 0 || width < 0 || (n_texts > 0 && (!ragged_ids || !text_off || !lens || (width > 0 && !out))))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_tok_pad_i32: n_texts=%lld width=%lld", (long long)n_texts, (long long)width);
    for (int64_t i = 0; i < n_texts; ++i)
        if (lens[i] > width)
            return tt_fail(TT_ERR_BAD_SHAPE, "tt_tok_pad_i32: row %lld has %d tokens > width %lld", (long long)i, lens[i], (long long)width);
    std::vector<int> bad((size_t)(n_threads > 1 ? n_threads : 1) + 1, 0);
    const int64_t per = n_threads > 1 ? (n_texts + n_threads - 1) / n_threads : n_texts;
    parallel_for(n_texts, n_threads, [&](int64_t lo, int64_t hi) {
        int wide = 0;
        for (int64_t i = lo; i < hi; ++i) {
            int32_t *row = out + i * width;
            const int64_t *src = ragged_ids + text_off[i];
            for (int32_t k = 0; k < lens[i]; ++k) {
                wide |= src[k] < INT32_MIN || src[k] > INT32_MAX;
                row[k] = (int32_t)src[k];
            }
            std::memset(row + lens[i], 0, sizeof(int32_t) * (size_t)(width - lens[i]));
        }
        bad[(size_t)(per > 0 ? lo / per : 0)] = wide;
    });
    for (int b : bad)
        if (b)
            return tt_fail(TT_ERR_BAD_INDEX, "tt_tok_pad_i32: an id does not fit 32 bits");
    return TT_OK;
}

// The host's Unicode tables (see TokTable): low [n_cp] uint32 (0xffffffff = context-dependent: hand the text back), cls [n_cp]
// (0 other, 1 word, 2 one of .,!?;) -- copied.
TT_EXPORT int tt_tok_set_unicode(void *handle, const uint32_t *low, const uint8_t *cls, int64_t n_cp)
{
    if (!handle || n_cp < 128 || n_cp > 0x110000 || !low || !cls)
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_tok_set_unicode: n_cp=%lld", (long long)n_cp);
    TokTable *t = (TokTable *)handle;
    t->ulow.assign(low, low + n_cp);
    t->ucls.assign(cls, cls + n_cp);
    return TT_OK;
}

// tt_tok_encode_ptrs for texts of 1-, 2- or 4-byte CODE UNITS (one code point each): unit_bytes[i] = 0: text i is text_len[i]
// ASCII bytes (the fast scan); 1 / 2 / 4: text_len[i] units of that many bytes (needs tt_tok_set_unicode; without it, or for a text
// with a context-dependent lower case, status[i] = 1 and the caller tokenises it).  ragged_ids / text_off_out as tt_tok_encode_ptrs
// (a text never has more tokens than units).
TT_EXPORT int tt_tok_encode_units(const void *handle, const void *const *texts, const int64_t *text_len, const uint8_t *unit_bytes,
                                  int64_t n_texts, int64_t *text_off_out, int64_t *ragged_ids, int32_t *lens, int32_t *status,
                                  int n_threads)
{
    if (!handle || n_texts < 0 || (n_texts > 0 && (!texts || !text_len || !unit_bytes || !text_off_out || !ragged_ids || !lens || !status)))
        return tt_fail(TT_ERR_BAD_SHAPE, "tt_tok_encode_units: n_texts=%lld", (long long)n_texts);
    int64_t run = 0;
    for (int64_t i = 0; i < n_texts; ++i) {
        const int ub = unit_bytes[i];
        if (text_len[i] < 0 || (text_len[i] > 0 && !texts[i]) || !(ub == 0 || ub == 1 || ub == 2 || ub == 4))
            return tt_fail(TT_ERR_BAD_SHAPE, "tt_tok_encode_units: text %lld: length %lld, unit %d", (long long)i, (long long)text_len[i], ub);
        text_off_out[i] = run;
        run += text_len[i];
    }
    if (n_texts > 0)
        text_off_out[n_texts] = run;
    const TokTable &t = *(const TokTable *)handle;
    parallel_for(n_texts, n_threads, [&](int64_t lo, int64_t hi) {
        std::vector<char> lower;
        lower.reserve(64);
        Lookups lk(t);
        for (int64_t i = lo; i < hi; ++i) {
            int64_t *dst = ragged_ids + text_off_out[i];
            const size_t n = (size_t)text_len[i];
            int64_t c;
            switch (unit_bytes[i]) {
            case 0: c = encode_one(t, lk, (const char *)texts[i], n, dst, lower); break;
            case 1: c = encode_units(t, lk, (const unsigned char *)texts[i], n, dst, lower); break;
            case 2: c = encode_units(t, lk, (const uint16_t *)texts[i], n, dst, lower); break;
            default: c = encode_units(t, lk, (const uint32_t *)texts[i], n, dst, lower); break;
            }
            status[i] = c < 0 ? 1 : 0;
            lens[i] = c < 0 ? 0 : (int32_t)c;
        }
        lk.flush();
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
