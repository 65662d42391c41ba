"""Closed-loop single-query clients on one handle (the reference's tools/perf.cc pattern): T threads, each
issuing host-buffer Search calls of one query, C3 index.  Reports aggregate queries/s and per-call latency
with request combining on (default) and off (GAMMA_HIP_NO_COMBINE=1 in the environment)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gamma_amd import api, synth
dev = torch.device("cuda", 0)
N, d, nlist, M, P, R, k = 1000000, 128, 4096, 16, 32, 200, 10
base = synth.sift_like(N, d=d, seed=1234)
cc, pq = api.train_ivfpq(base[:nlist * 40], nlist, M)
g = api.GammaHip(0)
g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, bucket_init_size=700)
g.ivfpq_set_trained(cc, pq, None)
g.raw_init(d)
for i0 in range(0, N, 200000):
    g.raw_append(base[i0:i0 + 200000])
    g.add(base[i0:i0 + 200000], i0)
q = synth.sift_like(4096, d=d, seed=4321)
args = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=R, has_rank=True, min_score=-1e30, max_score=1e30)
g.ivfpq_search(q[:1], k, args)
for T in (1, 8, 32, 128):
    calls = max(200, 4000 // T)
    lat = [None] * T

    def client(t):
        ts = np.empty(calls)
        for i in range(calls):
            j = (t * calls + i) % 4096
            t0 = time.perf_counter()
            g.ivfpq_search(q[j:j + 1], k, args)
            ts[i] = time.perf_counter() - t0
        lat[t] = ts

    th = [threading.Thread(target=client, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    al = np.sort(np.concatenate(lat)) * 1e6
    print("%3d client threads x 1 query: %8.0f queries/s, latency median %.0f us, p99 %.0f us" % (
        T, T * calls / dt, np.median(al), al[int(0.99 * len(al))]), flush=True)
