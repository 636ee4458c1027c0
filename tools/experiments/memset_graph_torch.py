#!/usr/bin/env python3
"""Second half of the memset-node reproducer (see memset_graph.hip): the round-1 symptom appeared under
torch.cuda.graph capture with the cleared block allocated DURING capture from the graph's private pool.  Captures
{flags = torch.empty(n) (pool allocation); hipMemsetAsync(flags, 0); seen.copy_(flags); flags.fill_(7)} and replays."""
import ctypes as C
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
dev = torch.device("cuda:0")
fails = 0
for n_i32 in (1, 3, 4, 8, 16, 64):
    seen = torch.zeros(n_i32, dtype=torch.int32, device=dev)
    log = torch.zeros(8, dtype=torch.int32, device=dev)
    rep = torch.zeros(1, dtype=torch.int64, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            warm = torch.empty(n_i32, dtype=torch.int32, device=dev)
            warm.fill_(7)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        flags = torch.empty(n_i32, dtype=torch.int32, device=dev)
        rc = hip.hipMemsetAsync(flags.data_ptr(), 0, n_i32 * 4, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        seen.copy_(flags)
        log.scatter_add_(0, rep.clamp(max=7), (seen != 0).any().to(torch.int32).reshape(1))
        rep += 1
        flags.fill_(7)
    for _ in range(6):
        g.replay()
    torch.cuda.synchronize()
    bad = [i for i, v in enumerate(log.tolist()) if v]
    print(f"{n_i32 * 4:4d} bytes from the capture pool: dirty seen in replays {bad or 'none'}")
    fails += bool(bad)
print(f"RESULT: {fails} of 6 cases failed")
