#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <random>
#include <cstdint>
extern "C" {
int tt_tok_create(const char *, const int64_t *, const int64_t *, int64_t, int64_t, void **);
void tt_tok_destroy(void *);
int tt_tok_set_unicode(void *, const uint32_t *, const uint8_t *, int64_t);
int tt_tok_encode_units(const void *, const void *const *, const int64_t *, const uint8_t *, int64_t, int64_t *, int64_t *, int32_t *, int32_t *, int);
int tt_tok_pad(const int64_t *, const int64_t *, const int32_t *, int64_t, int64_t, int64_t *, int);
}
int main()
{
    std::mt19937_64 rng(7);
    const char alpha[] = "abcXYZ019_ .,!?;-\t\n'\"()";
    std::vector<std::string> words; std::string blob; std::vector<int64_t> off{0}, ids;
    for (int i = 0; i < 3000; ++i) { std::string w; int L = 1 + rng() % 40; for (int k = 0; k < L; ++k) w += "abcdefghij0123_"[rng() % 15]; words.push_back(w); blob += w; off.push_back((int64_t)blob.size()); ids.push_back(i); }
    void *h; tt_tok_create(blob.data(), off.data(), ids.data(), (int64_t)words.size(), 99999, &h);
    std::vector<uint32_t> low(0x110000); std::vector<uint8_t> cls(0x110000, 0);
    for (uint32_t c = 0; c < 0x110000; ++c) low[c] = (c >= 'A' && c <= 'Z') ? c + 32 : c;
    for (int c = 0; c < 128; ++c) cls[c] = (isalnum(c) || c == '_') ? 1 : (strchr(".,!?;", c) && c ? 2 : 0);
    for (uint32_t c = 0xc0; c < 0x3000; ++c) cls[c] = 1;
    low[0x130] = 0xffffffffu;
    tt_tok_set_unicode(h, low.data(), cls.data(), 0x110000);
    long checksum = 0;
    for (int round = 0; round < 200; ++round) {
        const int n = 64;
        // every text in its OWN exact-size heap block so that an over-read trips the sanitizer
        std::vector<void *> bufs(n); std::vector<const void *> ptrs(n); std::vector<int64_t> len(n); std::vector<uint8_t> ub(n);
        int64_t total = 0;
        for (int i = 0; i < n; ++i) {
            const int units = rng() % 90, kind = (int[]){0, 0, 1, 2, 4}[rng() % 5];
            len[i] = units; ub[i] = (uint8_t)kind; total += units;
            const int bytes = units * (kind ? kind : 1);
            bufs[i] = malloc(bytes ? bytes : 1);
            for (int k = 0; k < units; ++k) {
                uint32_t cp = rng() % 4 ? (uint32_t)alpha[rng() % (sizeof alpha - 1)] : (uint32_t)words[rng() % words.size()][0];
                if (kind && rng() % 5 == 0) cp = kind == 1 ? 0xe9 : (kind == 2 ? 0x2019 + rng() % 40 : 0x1f600);
                if (kind == 0 || kind == 1) ((unsigned char *)bufs[i])[k] = (unsigned char)(kind == 0 ? cp & 0x7f : cp & 0xff);
                else if (kind == 2) ((uint16_t *)bufs[i])[k] = (uint16_t)cp;
                else ((uint32_t *)bufs[i])[k] = cp;
            }
            // splice vocabulary words in (ASCII kinds)
            if (kind == 0 && units > 45) { const std::string &w = words[rng() % words.size()]; memcpy(bufs[i], w.data(), w.size() < (size_t)units ? w.size() : units); }
            ptrs[i] = bufs[i];
        }
        std::vector<int64_t> toff(n + 1), ragged(total + 1), padded((size_t)n * 96); std::vector<int32_t> lens(n), st(n);
        tt_tok_encode_units(h, ptrs.data(), len.data(), ub.data(), n, toff.data(), ragged.data(), lens.data(), st.data(), 1 + round % 3);
        int w = 0; for (int i = 0; i < n; ++i) w = lens[i] > w ? lens[i] : w;
        tt_tok_pad(ragged.data(), toff.data(), lens.data(), n, w, padded.data(), 2);
        for (int i = 0; i < n; ++i) { checksum += lens[i] + st[i]; free(bufs[i]); }
    }
    tt_tok_destroy(h);
    printf("asan harness done, checksum %ld\n", checksum);
}
