import torch, time
x = torch.empty(420*1024*1024//4, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
for name, fn in (("fill", lambda: x.fill_(1.0)), ("copy", lambda: y.copy_(x)), ("read-sum", lambda: x.sum())):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    nbytes = x.numel() * 4 * (2 if name == "copy" else 1)
    print(name, round(ms * 1e3, 1), "us", round(nbytes / ms / 1e9, 2), "TB/s")
