import torch, time
x = torch.empty(2 * 1024**3, dtype=torch.float64, device="cuda")   # 16 GiB
for fn, name in ((lambda: x.zero_(), "zero_"), (lambda: x.fill_(1.5), "fill_")):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): fn()
    b.record(); torch.cuda.synchronize()
    print(name, 5 * x.numel() * 8 / (a.elapsed_time(b) * 1e-3) / 1e9, "GB/s")
