"""Known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box:
out = x + 1 over 1 GiB of fp32 reads exactly 2^30 bytes and writes exactly 2^30 bytes per launch
(16 B/lane vectorised elementwise kernel).  Run under `rocprofv3 --pmc <COUNTER>`."""
import torch

n = (1 << 30) // 4
x = torch.ones(n, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
for _ in range(6):
    torch.add(x, 1.0, out=y)
torch.cuda.synchronize()
print("calib done", float(y[0]))
