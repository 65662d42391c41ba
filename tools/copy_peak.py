"""Achievable HBM bandwidth of this MI355X as seen from plain device code: a read-only pass (torch sum over
4 GiB) and a device copy (read + write 2 x 4 GiB), best of several runs.  The 8 TB/s in the roofline is
the vendor peak; this is the number a streaming kernel can actually reach on the box."""
import time
import torch
dev = torch.device("cuda", 0)
n = 1 << 30
x = torch.empty(n, dtype=torch.float32, device=dev).normal_()
y = torch.empty_like(x)
for name, fn, nbytes in (("read (sum)", lambda: x.sum(), 4 * n), ("copy", lambda: y.copy_(x), 8 * n),
                         ("fill (write)", lambda: y.fill_(1.0), 4 * n)):
    best = 1e9
    for _ in range(10):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e-3)
    print("%-13s %6.2f GiB in %7.3f ms = %6.2f TB/s" % (name, nbytes / 2**30, best * 1e3, nbytes / best / 1e12))
