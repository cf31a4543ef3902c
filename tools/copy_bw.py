import torch, time
n = 8 * 1024**3 // 4
a = torch.empty(n, dtype=torch.float32, device="cuda").fill_(1.0)
b = torch.empty(n, dtype=torch.float32, device="cuda")
for name, fn, bytes_ in (("copy", lambda: b.copy_(a), 2 * n * 4), ("fill", lambda: b.fill_(2.0), n * 4), ("sum(read)", lambda: a.sum(), n * 4)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%s: %.3f ms, %.2f TB/s" % (name, ms, bytes_ / ms / 1e9), flush=True)
