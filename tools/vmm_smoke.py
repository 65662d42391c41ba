import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from gamma_amd import api
t0 = time.time()
g = api.GammaHip(0)
g.raw_init(128)
x = np.random.default_rng(0).integers(0, 255, size=(30000, 128)).astype(np.float32)
g.raw_append(x)
print("append 1", g.raw_stats(), round(time.time() - t0, 2), flush=True)
for i in range(10):
    g.raw_append(x + 1000.0 * (i + 1))
print("append 11", g.raw_stats(), round(time.time() - t0, 2), flush=True)
D, I = g.flat_search(x[:8] + 1.0, 1, api.SearchArgs(metric=api.METRIC_L2, min_score=-3e38, max_score=3e38))
print(I[:, 0], D[:, 0], flush=True)
g.close()
print("ok", round(time.time() - t0, 2))
