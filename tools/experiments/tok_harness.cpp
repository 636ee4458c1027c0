// Timing harness for the host tokenizer (csrc/tokenize.hip) WITHOUT Python: 16 384 synthetic passages of 40 - 99 uniformly random
// words of a 400 k-word vocabulary (the cache-unfriendliest text there is), tt_tok_encode_ptrs with N threads.
// tools/experiments/tok_harness.sh builds it against variants of the source (-DTT_TOK_LAG, -DTT_TOK_HUGEPAGES, -DTT_TOK_NO_SIMD).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <vector>
#include <random>
#include <cstdint>
extern "C" {
int tt_tok_create(const char *, const int64_t *, const int64_t *, int64_t, int64_t, void **);
int tt_tok_encode_ptrs(const void *, const char *const *, const int64_t *, int64_t, int64_t *, int64_t *, int32_t *, int32_t *, int);
}
int main(int argc, char **argv)
{
    const int V = 400000, n = 16384;
    std::vector<std::string> words;
    std::string blob; std::vector<int64_t> off{0}, ids;
    for (int i = 0; i < V; ++i) { words.push_back("w" + std::to_string(i)); blob += words.back(); off.push_back((int64_t)blob.size()); ids.push_back(i); }
    void *h; tt_tok_create(blob.data(), off.data(), ids.data(), V, V, &h);
    std::mt19937_64 rng(1);
    std::vector<std::string> docs(n);
    int64_t total = 0, ntok = 0;
    for (auto &d : docs) { int L = 40 + rng() % 60; for (int k = 0; k < L; ++k) { d += words[rng() % V]; d += ' '; } total += d.size(); ntok += L; }
    std::vector<const char *> ptrs(n); std::vector<int64_t> len(n), toff(n + 1), ragged(total + 1); std::vector<int32_t> lens(n), st(n);
    for (int i = 0; i < n; ++i) { ptrs[i] = docs[i].data(); len[i] = docs[i].size(); }
    for (int a = 1; a < (argc > 1 ? argc : 2); ++a) {
        const int nt = argc > 1 ? atoi(argv[a]) : 1;
        double best = 1e9;
        for (int rep = 0; rep < 7; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            tt_tok_encode_ptrs(h, ptrs.data(), len.data(), n, toff.data(), ragged.data(), lens.data(), st.data(), nt);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            best = dt < best ? dt : best;
        }
        printf("threads %2d: %.2f ms  %.1f ns/token/thread  %.1f M tokens/s\n", nt, best * 1e3, best / ntok * 1e9 * nt, ntok / best / 1e6);
    }
}
