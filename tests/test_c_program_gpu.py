"""libtt.so without Python or torch: a plain HIP host program (hipMalloc + the C ABI of include/tt.h) scores a small
corpus and checks the result against its own sequential fmaf loop -- the binding a non-Python caller would write."""
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

SRC = r"""
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "tt.h"
#define CK(x) do { if ((x) != hipSuccess) { printf("hip error line %d\n", __LINE__); return 10; } } while (0)
int main()
{
    const int B = 5, d = 256, k = 10;
    const int64_t N = 70000;
    std::vector<float> Q(B * d), D(N * d);
    srand(3);
    for (auto &x : Q) x = rand() / (float)RAND_MAX - 0.5f;
    for (auto &x : D) x = rand() / (float)RAND_MAX - 0.5f;
    float *dQ, *dD, *dV; int64_t *dI; void *ws;
    CK(hipMalloc(&dQ, Q.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMalloc(&dV, B * k * 4)); CK(hipMalloc(&dI, B * k * 8));
    CK(hipMemcpy(dQ, Q.data(), Q.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dD, D.data(), D.size() * 4, hipMemcpyHostToDevice));
    const size_t need = tt_score_topk_workspace_bytes(B, N, d, k);
    CK(hipMalloc(&ws, need));
    hipStream_t st; CK(hipStreamCreate(&st));
    if (tt_score_topk_f32(dQ, B, d, dD, N, k, 100, dV, dI, ws, need, st) != 0) { printf("tt: %s\n", tt_last_error()); return 11; }
    CK(hipStreamSynchronize(st));
    std::vector<float> V(B * k); std::vector<int64_t> I(B * k);
    CK(hipMemcpy(V.data(), dV, V.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(I.data(), dI, I.size() * 8, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b) {
        std::vector<float> s(N);
        for (int64_t n = 0; n < N; ++n) {
            float acc = 0.0f;
            for (int x = 0; x < d; ++x) acc = fmaf(Q[b * d + x], D[n * d + x], acc);
            s[n] = acc;
        }
        for (int r = 0; r < k; ++r) {              /* k rounds of arg-best with (score desc, index asc) */
            int64_t best = -1;
            for (int64_t n = 0; n < N; ++n)
                if (s[n] > -INFINITY && (best < 0 || s[n] > s[best])) best = n;
            if (I[b * k + r] != best + 100 || V[b * k + r] != s[best]) { printf("mismatch b=%d r=%d\n", b, r); return 12; }
            s[best] = -INFINITY;
        }
    }
    printf("ok %s\n", tt_version());
    return 0;
}
"""


def test_plain_hip_program_through_the_c_abi(tmp_path):
    from twotowermlretrieval_amd import build as b
    lib = b.build()
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = tmp_path / "prog.hip"
    src.write_text(SRC)
    exe = tmp_path / "prog"
    obj = tmp_path / "prog.o"
    r = subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", f"-I{b.PKG.parent / 'include'}", "-c", str(src), "-o", str(obj)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([hipcc, "--offload-arch=gfx950", str(obj), str(lib), f"-Wl,-rpath,{lib.parent}", "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.returncode, r.stdout, r.stderr[-500:])
