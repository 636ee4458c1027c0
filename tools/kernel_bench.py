#!/usr/bin/env python3
"""Time the K4 scoring kernel alone (HIP events on the launch stream) over a (B, N, k) sweep.
   python tools/kernel_bench.py [--n 10000000] [--iters 5]"""
import argparse, ctypes as C, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from twotowermlretrieval_amd import _lib

def time_partials(Q, D, k, iters=5, warm=2):
    L = _lib.lib()
    B, d = Q.shape; N = D.shape[0]
    ws = torch.empty(L.tt_score_topk_workspace_bytes(B, N, d, k), dtype=torch.uint8, device=Q.device)
    pv, pi, pm = C.c_void_p(), C.c_void_p(), C.c_int()
    st = torch.cuda.current_stream().cuda_stream
    def call():
        _lib.check(L.tt_score_topk_partials_f32(Q.data_ptr(), B, d, D.data_ptr(), N, k, 0, ws.data_ptr(), ws.numel(),
                                                C.byref(pv), C.byref(pi), C.byref(pm), None, st))
    for _ in range(warm): call()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); call(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts)//2], pm.value

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--cases", default="1:10,16:10,32:10,32:50,64:10,256:10,1024:10")
    a = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(1)
    D = torch.randn(a.n, 256, device="cuda", generator=g); D /= D.norm(dim=1, keepdim=True)
    for case in a.cases.split(","):
        B, k = map(int, case.split(":"))
        Q = torch.randn(B, 256, device="cuda", generator=g); Q /= Q.norm(dim=1, keepdim=True)
        ms, pm = time_partials(Q, D, k, a.iters)
        byts = a.n * 1024 + B * 1024 + B * k * 12
        fl = 2.0 * (-(-B // 32) * 32) * a.n * 256
        # merge kernel time via full call
        import twotowermlretrieval_amd as tt
        tt.score_topk(Q, D, k); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); tt.score_topk(Q, D, k); e1.record(); torch.cuda.synchronize()
        print(json.dumps(dict(B=B, k=k, N=a.n, kernel_ms=round(ms, 4), full_ms=round(e0.elapsed_time(e1), 4), part_m=pm,
                              GBps=round(byts / ms / 1e6, 1), hbm_frac=round(byts / ms / 1e6 / 8000, 4),
                              TFLOPs_padded=round(fl / ms / 1e9, 2), qps=round(B / ms * 1e3, 1))), flush=True)
def screened():
    import twotowermlretrieval_amd as tt
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    a, _ = ap.parse_known_args()
    g = torch.Generator(device="cuda").manual_seed(1)
    D = torch.randn(a.n, 256, device="cuda", generator=g); D /= D.norm(dim=1, keepdim=True)
    ix = tt.BruteForceIndex(D, screen=True)
    for B in (128, 256, 512, 1024, 2048, 4096):
        Q = torch.randn(B, 256, device="cuda", generator=g); Q /= Q.norm(dim=1, keepdim=True)
        for _ in range(2): ix.search(Q, 10)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ix.search(Q, 10)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(json.dumps(dict(path="screened", B=B, N=a.n, ms=round(ms, 4), qps=round(B / ms * 1e3, 1),
                              TFLOPs=round(2.0 * B * a.n * 256 / ms / 1e9, 1), f16_GBps=round(a.n * 512 / ms / 1e6 * ((B + 511) // 512), 1),
                              flags=int(ix.fallback_flags.ne(0).sum().item()))), flush=True)

if __name__ == "__main__":
    if "--screened" in sys.argv:
        sys.argv.remove("--screened"); screened()
    else:
        main()
