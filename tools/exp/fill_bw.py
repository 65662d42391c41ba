import torch, time
x = torch.empty(268435456 // 4, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
for fn, name, nbytes in ((lambda: x.fill_(1.0), "fill 268 MB", 268435456), (lambda: y.copy_(x), "copy 268 MB (read + write)", 2 * 268435456)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("%s: %.1f us = %.2f TB/s" % (name, us, nbytes / us / 1e6))
