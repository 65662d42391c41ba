"""Experiment: two handles (own streams and workspaces) on one GPU searching half batches concurrently."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gamma_amd import api, synth
N, d, nlist, M = 1000000, 128, 4096, 16
base = synth.sift_like(N, d=d, seed=1234)
dev = torch.device("cuda", 0)
cc, pq = api.train_ivfpq(base[:nlist * 64], nlist, M)
def mk():
    g = api.GammaHip(0)
    g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, bucket_init_size=1000)
    g.ivfpq_set_trained(cc, pq, None)
    g.add(base, 0)
    g.raw_init(d); g.raw_append(base)
    return g
hs = [mk() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2)]
args = api.SearchArgs(metric=api.METRIC_L2, nprobe=32, recall_num=200, has_rank=True, min_score=0.0, max_score=1e30)
tot = 8192
for nh in range(1, len(hs) + 1):
    nq = tot // nh
    q = synth.sift_like(tot, d=d, seed=4321)
    dq = torch.from_numpy(q).to(dev)
    outs = [(torch.empty((nq, 10), dtype=torch.float32, device=dev), torch.empty((nq, 10), dtype=torch.int64, device=dev)) for _ in range(nh)]
    steps = 60
    def run(i):
        g = hs[i]; D, I = outs[i]; x = dq[i * nq:(i + 1) * nq]
        for _ in range(steps):
            g.ivfpq_search_device(x.data_ptr(), nq, 10, args, D.data_ptr(), I.data_ptr())
        g.synchronize()
    for i in range(nh): run(i)   # warm
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i,)) for i in range(nh)]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("%d concurrent handle(s) x %d queries: %.3f ms per %d queries = %.0f queries/s" % (nh, nq, dt / steps * 1e3, tot, tot * steps / dt))
