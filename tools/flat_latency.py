"""Flat (brute-force) search, small calls: per-call latency at C1 size (10 k x 128) and at 1 M x 128, k = 10,
device buffers, synchronised per call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gamma_amd import api, synth
d, k = 128, 10
dev = torch.device("cuda", 0)
for N in (10000, 1000000):
    base = synth.sift_like(N, d=d, seed=1234)
    q = synth.sift_like(256, d=d, seed=4321)
    g = api.GammaHip(0)
    g.raw_init(d)
    g.raw_append(base)
    dq = torch.from_numpy(q).to(dev)
    D = torch.empty((256, k), dtype=torch.float32, device=dev)
    I = torch.empty((256, k), dtype=torch.int64, device=dev)
    args = api.SearchArgs(metric=api.METRIC_L2, min_score=0.0, max_score=1e30)
    for nq in (1, 16, 64):
        ts = []
        for i in range(120):
            t0 = time.perf_counter()
            g.flat_search_device(dq[(i * nq) % 128:].data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
            g.synchronize()
            ts.append(time.perf_counter() - t0)
        ts = np.sort(np.array(ts[20:])) * 1e6
        print("flat %d x %d, %d queries per call, k=%d: median %.1f us, p99 %.1f us" % (N, d, nq, k, np.median(ts), ts[98]), flush=True)
    g.close()
