import torch, time
x = torch.empty(440_000_000, dtype=torch.float32, device="cuda")  # 1.76 GB
y = torch.empty_like(x)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
a = t(lambda: x.fill_(1.0)); print("fill 1.76GB ms", a*1e3, "TB/s", 1.76e-3/a)
b = t(lambda: y.copy_(x)); print("copy ms", b*1e3, "TB/s (r+w)", 3.52e-3/b)
c = t(lambda: x.sum()); print("read ms", c*1e3, "TB/s", 1.76e-3/c)
