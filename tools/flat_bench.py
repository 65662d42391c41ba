"""C2-shaped flat search timing (1M x 128, 1024 queries per call, k = 100), device buffers."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gamma_amd import api, synth
N, d, nq, k = int(os.environ.get("FLAT_BENCH_N", "1000000")), 128, int(os.environ.get("FLAT_BENCH_NQ", "1024")), int(sys.argv[1]) if len(sys.argv) > 1 else 100
base = synth.sift_like(N, d=d, seed=1234)
q = synth.sift_like(nq, d=d, seed=4321)
g = api.GammaHip(0)
g.raw_init(d)
if os.environ.get("FLAT_BENCH_NO_TIES"):
    g.set_exact_ties(False)
g.raw_append(base)
dev = torch.device("cuda", 0)
dq = torch.from_numpy(q).to(dev)
D = torch.empty((nq, k), dtype=torch.float32, device=dev)
I = torch.empty((nq, k), dtype=torch.int64, device=dev)
args = api.SearchArgs(metric=api.METRIC_L2, min_score=0.0, max_score=1e30)
for _ in range(2):
    g.flat_search_device(dq.data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
g.synchronize()
t0 = time.perf_counter()
n = int(os.environ.get("FLAT_BENCH_CALLS", "5"))
for _ in range(n):
    g.flat_search_device(dq.data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
g.synchronize()
dt = (time.perf_counter() - t0) / n
print("flat %d x 128, %d queries, k=%d: %.2f ms per call = %.0f queries/s" % (N, nq, k, dt * 1e3, nq / dt))
