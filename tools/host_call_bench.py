"""Host-buffer Search calls at the C3 shape (what the plugin boundary hands over): median ms per call for several batch
sizes, against the device-pointer call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gamma_amd import api, synth
N, d, nlist, M = 1000000, 128, 4096, 16
base = synth.sift_like(N, d=d, seed=1234)
g = api.GammaHip(0)
cc, pq = g.ivfpq_train(base[:nlist * 64], nlist, M)
g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, bucket_init_size=1000)
g.ivfpq_set_trained(cc, pq, None)
g.raw_init(d)
for i0 in range(0, N, 250000):
    g.raw_append(base[i0:i0 + 250000])
    g.add(base[i0:i0 + 250000], i0)
q = synth.sift_like(32768, d=d, seed=4321)
dev = torch.device("cuda", 0)
dq = torch.from_numpy(q).to(dev)
k = 10
args = api.SearchArgs(metric=api.METRIC_L2, nprobe=32, recall_num=200, has_rank=True, min_score=0.0, max_score=1e30)
for nq in (1024, 4096, 8192, 16384, 32768):
    D = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    qh = np.ascontiguousarray(q[:nq])
    td, th = [], []
    for i in range(25):
        t0 = time.perf_counter()
        g.ivfpq_search_device(dq.data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
        g.synchronize()
        td.append(time.perf_counter() - t0)
    for i in range(25):
        t0 = time.perf_counter()
        g.ivfpq_search(qh, k, args)
        th.append(time.perf_counter() - t0)
    md, mh = np.median(td[5:]) * 1e3, np.median(th[5:]) * 1e3
    print("nq %6d: device-pointer call %.3f ms, host-buffer call %.3f ms (+%.3f) = %.2f M queries/s" % (nq, md, mh, mh - md, nq / mh / 1e3))
