#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <thread>
#include <random>
#include <cstdint>
extern "C" {
int tt_tok_create(const char *, const int64_t *, const int64_t *, int64_t, int64_t, void **);
int tt_tok_encode_ptrs(const void *, const char *const *, const int64_t *, int64_t, int64_t *, int64_t *, int32_t *, int32_t *, int);
int tt_tok_pad_i32(const int64_t *, const int64_t *, const int32_t *, int64_t, int64_t, int32_t *, int);
}
int main()
{
    const int V = 5000, n = 2000;
    std::vector<std::string> words; std::string blob; std::vector<int64_t> off{0}, ids;
    for (int i = 0; i < V; ++i) { words.push_back("w" + std::to_string(i)); blob += words.back(); off.push_back((int64_t)blob.size()); ids.push_back(i); }
    void *h; tt_tok_create(blob.data(), off.data(), ids.data(), V, V, &h);
    std::mt19937_64 rng(1);
    std::vector<std::string> docs(n);
    int64_t total = 0;
    for (auto &d : docs) { int L = 5 + rng() % 40; for (int k = 0; k < L; ++k) { d += words[rng() % V]; d += ' '; } total += d.size(); }
    std::vector<const char *> ptrs(n); std::vector<int64_t> len(n);
    for (int i = 0; i < n; ++i) { ptrs[i] = docs[i].data(); len[i] = docs[i].size(); }
    auto worker = [&](int nt, int reps, long *sum) {
        std::vector<int64_t> toff(n + 1), ragged(total + 1); std::vector<int32_t> lens(n), st(n), out((size_t)n * 64);
        long s = 0;
        for (int r = 0; r < reps; ++r) {
            tt_tok_encode_ptrs(h, ptrs.data(), len.data(), n, toff.data(), ragged.data(), lens.data(), st.data(), nt);
            tt_tok_pad_i32(ragged.data(), toff.data(), lens.data(), n, 64, out.data(), nt);
            for (int i = 0; i < n; ++i) s += lens[i] + out[(size_t)i * 64];
        }
        *sum = s;
    };
    long s1 = 0, s2 = 0, s3 = 0, s0 = 0;
    worker(1, 1, &s0);
    std::thread a(worker, 4, 30, &s1), b(worker, 3, 30, &s2), c(worker, 8, 30, &s3);
    a.join(); b.join(); c.join();
    printf("%ld %ld %ld %ld %s\n", s0 * 30, s1, s2, s3, (s1 == s0 * 30 && s2 == s1 && s3 == s1) ? "consistent" : "MISMATCH");
}
