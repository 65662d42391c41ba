import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gamma_amd import api, synth
N, d, nlist, M, P, k, nq = 1000000, 128, 4096, 16, 32, 10, 16384
base = synth.sift_like(N, d=d, seed=1234)
g = api.GammaHip(0)
g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, bucket_init_size=1000)
cc, pq = api.train_ivfpq(base[:nlist * 64], nlist, M)
g.ivfpq_set_trained(cc, pq, None)
g.raw_init(d); g.raw_append(base); g.add(base, 0)
dev = torch.device("cuda", 0)
q = torch.from_numpy(synth.sift_like(nq, d=d, seed=4321)).to(dev)
D = torch.empty((nq, k), dtype=torch.float32, device=dev); I = torch.empty((nq, k), dtype=torch.int64, device=dev)
for R in (200, 300, 400, 512):
    a = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=R, has_rank=True, min_score=0.0, max_score=1e30)
    for _ in range(3): g.ivfpq_search_device(q.data_ptr(), nq, k, a, D.data_ptr(), I.data_ptr())
    g.synchronize(); g.profile_enable(True); g.profile_reset()
    t0 = time.perf_counter()
    for _ in range(10): g.ivfpq_search_device(q.data_ptr(), nq, k, a, D.data_ptr(), I.data_ptr())
    g.synchronize(); dt = (time.perf_counter() - t0) / 10
    pr = g.profile(); g.profile_enable(False)
    print("R=%d: %.3f ms/step, stage us %s, scan frac %.3f, checksum %d" % (R, dt * 1e3, {n: round(pr[n][0] / 10 * 1e3) for n in ("coarse","tables","scan","select","rerank")},
          pr["scan_bytes"] / 10 / (pr["scan"][0] / 10 / 1e3) / 8e12, int(I.sum().item())))
